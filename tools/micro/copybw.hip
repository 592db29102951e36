// Micro-benchmark behind k_ref_dws's roofline: what does the chip sustain for the refinement's traffic shape?
// A refinement map is [B,H,W,32] float32 = 128 B per pixel (134 MB at 8 x 256x512); a depthwise-separable block reads one
// map and writes another.  Measures, on three rotating 134 MB buffers (as the ping-pong workspace does):
//   write-only, read-only, dense copy, and copies whose workgroups own 8 x 16 pixels of ONE dilation phase (the pixels of a
//   tile are d lines apart, as in k_ref_dws) for d = 1, 2, 4, 8, 16, with and without the 1-pixel halo re-read.
//   hipcc --offload-arch=gfx950 -O3 -o copybw tools/micro/copybw.hip && ./copybw [B=8]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

constexpr int H = 256, W = 512;

__global__ __launch_bounds__(256) void k_write(float4 *out, size_t n4)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) out[i] = make_float4(1.f, 2.f, 3.f, (float)threadIdx.x);
}

__global__ __launch_bounds__(256) void k_read(const float4 *in, float *sink, size_t n4)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float4 v = i < n4 ? in[i] : make_float4(0, 0, 0, 0);
    if (v.x + v.y + v.z + v.w == 12345.678f) sink[0] = 1.f;
}

__global__ __launch_bounds__(256) void k_copy(const float4 *in, float4 *out, size_t n4)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) out[i] = in[i];
}

// phase-grid tile copy: workgroup = TYv x TXv pixels of one phase (y mod d, x mod d); thread = (pixel, 4-channel group);
// HALO: also read (and discard) the 1-pixel halo ring of the phase grid, as a 3x3 dilated stencil does
template <bool HALO, int TYv, int TXv, int NTv>
__global__ __launch_bounds__(NTv) void k_phase(const float4 *in, float4 *out, float *sink, int d, int nbx, int nby)
{
    int bid = blockIdx.x;
    const int d2 = d * d, phase = bid % d2;
    bid /= d2;
    const int bx = bid % nbx;
    bid /= nbx;
    const int by = bid % nby, b = bid / nby;
    const int Y0 = by * TYv * d + phase / d, X0 = bx * TXv * d + phase % d;
    const int c4 = threadIdx.x & 7;
    const float4 *inb = in + (size_t)b * H * W * 8;
    float4 acc = make_float4(0, 0, 0, 0);
    constexpr int HXv = TXv + 2, NP = HALO ? (TYv + 2) * HXv : TYv * TXv, PPI = NTv / 8, IT = (NP + PPI - 1) / PPI;
    constexpr int OT = TYv * TXv / PPI;
    float4 v[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int hp = (threadIdx.x >> 3) + PPI * i;
        int gy, gx;
        if (HALO) {
            const int hy = hp / HXv, hx = hp - hy * HXv;
            gy = Y0 + (hy - 1) * d;
            gx = X0 + (hx - 1) * d;
        } else {
            gy = Y0 + (hp / TXv) * d;
            gx = X0 + (hp % TXv) * d;
        }
        const bool ok = hp < NP && gy >= 0 && gy < H && gx >= 0 && gx < W;
        v[i] = inb[ok ? ((size_t)gy * W + gx) * 8 + c4 : 0];
    }
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        acc.x += v[i].x;
        acc.y += v[i].y;
        acc.z += v[i].z;
        acc.w += v[i].w;
    }
    float4 *outb = out + (size_t)b * H * W * 8;
#pragma unroll
    for (int i = 0; i < OT; ++i) {
        const int p = (threadIdx.x >> 3) + PPI * i;
        const int gy = Y0 + (p / TXv) * d, gx = X0 + (p % TXv) * d;
        if (gy < H && gx < W) outb[((size_t)gy * W + gx) * 8 + c4] = HALO ? acc : v[i];
    }
    if (acc.x == 12345.678f) sink[0] = 1.f;
}


// persistent form of the halo tile copy: a fixed grid of workgroups walks the tile list and issues the NEXT tile's loads before
// it stores the current tile (twice the bytes in flight per workgroup, no launch / drain per tile)
template <int TYv, int TXv, int NTv>
__global__ __launch_bounds__(NTv) void k_phase_persist(const float4 *in, float4 *out, float *sink, int d, int nbx, int nby, int ntiles)
{
    const int c4 = threadIdx.x & 7;
    constexpr int HXv = TXv + 2, NP = (TYv + 2) * HXv, PPI = NTv / 8, IT = (NP + PPI - 1) / PPI, OT = TYv * TXv / PPI;
    const int d2 = d * d;
    auto coords = [&](int bid, int &b, int &Y0, int &X0) {
        const int phase = bid % d2;
        bid /= d2;
        const int bx = bid % nbx;
        bid /= nbx;
        const int by = bid % nby;
        b = bid / nby;
        Y0 = by * TYv * d + phase / d;
        X0 = bx * TXv * d + phase % d;
    };
    auto load = [&](int bid, float4 (&v)[IT]) {
        int b, Y0, X0;
        coords(bid, b, Y0, X0);
        const float4 *inb = in + (size_t)b * H * W * 8;
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int hp = (threadIdx.x >> 3) + PPI * i;
            const int hy = hp / HXv, hx = hp - hy * HXv;
            const int gy = Y0 + (hy - 1) * d, gx = X0 + (hx - 1) * d;
            const bool ok = hp < NP && gy >= 0 && gy < H && gx >= 0 && gx < W;
            v[i] = inb[ok ? ((size_t)gy * W + gx) * 8 + c4 : 0];
        }
    };
    float4 cur[IT], nxt[IT];
    int t = blockIdx.x;
    if (t >= ntiles) return;
    load(t, cur);
    float4 total = make_float4(0, 0, 0, 0);
    for (; t < ntiles; t += gridDim.x) {
        const int tn = t + gridDim.x;
        if (tn < ntiles) load(tn, nxt);
        float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            acc.x += cur[i].x;
            acc.y += cur[i].y;
            acc.z += cur[i].z;
            acc.w += cur[i].w;
        }
        int b, Y0, X0;
        coords(t, b, Y0, X0);
        float4 *outb = out + (size_t)b * H * W * 8;
#pragma unroll
        for (int i = 0; i < OT; ++i) {
            const int p = (threadIdx.x >> 3) + PPI * i;
            const int gy = Y0 + (p / TXv) * d, gx = X0 + (p % TXv) * d;
            if (gy < H && gx < W) outb[((size_t)gy * W + gx) * 8 + c4] = acc;
        }
        total.x += acc.x;
#pragma unroll
        for (int i = 0; i < IT; ++i) cur[i] = nxt[i];
    }
    if (total.x == 12345.678f) sink[0] = 1.f;
}

template <class F>
static float timeit(F f, int reps)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f(i);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f(i);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}

int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 8;
    const size_t n4 = (size_t)B * H * W * 8, bytes = n4 * 16;
    float4 *buf[3];
    float *sink;
    for (auto &p : buf) {
        (void)hipMalloc(&p, bytes);
        (void)hipMemset(p, 0, bytes);
    }
    (void)hipMalloc(&sink, 4);
    const unsigned nb = (unsigned)((n4 + 255) / 256);
    const double MB = bytes / 1e6;
    printf("map = %d x %dx%d x 32 ch float32 = %.1f MB; three buffers rotate\n", B, H, W, MB);
    float us;
    us = timeit([&](int i) { k_write<<<nb, 256>>>(buf[i % 3], n4); }, 30);
    printf("write only           : %7.1f us  %5.2f TB/s\n", us, MB / us);
    us = timeit([&](int i) { k_read<<<nb, 256>>>(buf[i % 3], sink, n4); }, 30);
    printf("read only            : %7.1f us  %5.2f TB/s\n", us, MB / us);
    us = timeit([&](int i) { k_copy<<<nb, 256>>>(buf[i % 3], buf[(i + 1) % 3], n4); }, 30);
    printf("dense copy           : %7.1f us  %5.2f TB/s (read + write)\n", us, 2 * MB / us);
    for (int d : {1, 2, 4, 8, 16}) {
        const int nbx = (W + 16 * d - 1) / (16 * d), nby = (H + 8 * d - 1) / (8 * d);
        const unsigned g = (unsigned)(nbx * nby * d * d * B);
        us = timeit([&](int i) { k_phase<false, 8, 16, 256><<<g, 256>>>(buf[i % 3], buf[(i + 1) % 3], sink, d, nbx, nby); }, 30);
        printf("phase-grid copy d=%-2d : %7.1f us  %5.2f TB/s (algorithmic read + write)\n", d, us, 2 * MB / us);
        us = timeit([&](int i) { k_phase<true, 8, 16, 256><<<g, 256>>>(buf[i % 3], buf[(i + 1) % 3], sink, d, nbx, nby); }, 30);
        printf("  ... with halo  d=%-2d : %7.1f us  %5.2f TB/s (algorithmic; the halo re-reads 1.41x)\n", d, us, 2 * MB / us);
    }
    // larger phase-grid tiles with the halo: 8 x 32 (halo 1.33x) and 16 x 16 (1.27x), 256 and 512 threads
    for (int d : {1, 2, 8, 16}) {
        {
            const int nbx = (W + 32 * d - 1) / (32 * d), nby = (H + 8 * d - 1) / (8 * d);
            const unsigned g = (unsigned)(nbx * nby * d * d * B);
            us = timeit([&](int i) { k_phase<true, 8, 32, 256><<<g, 256>>>(buf[i % 3], buf[(i + 1) % 3], sink, d, nbx, nby); }, 30);
            printf("tile  8x32 halo d=%-2d 256 thr: %7.1f us  %5.2f TB/s\n", d, us, 2 * MB / us);
            us = timeit([&](int i) { k_phase<true, 8, 32, 512><<<g, 512>>>(buf[i % 3], buf[(i + 1) % 3], sink, d, nbx, nby); }, 30);
            printf("tile  8x32 halo d=%-2d 512 thr: %7.1f us  %5.2f TB/s\n", d, us, 2 * MB / us);
        }
        {
            const int nbx = (W + 16 * d - 1) / (16 * d), nby = (H + 16 * d - 1) / (16 * d);
            const unsigned g = (unsigned)(nbx * nby * d * d * B);
            us = timeit([&](int i) { k_phase<true, 16, 16, 256><<<g, 256>>>(buf[i % 3], buf[(i + 1) % 3], sink, d, nbx, nby); }, 30);
            printf("tile 16x16 halo d=%-2d 256 thr: %7.1f us  %5.2f TB/s\n", d, us, 2 * MB / us);
            us = timeit([&](int i) { k_phase<true, 16, 16, 512><<<g, 512>>>(buf[i % 3], buf[(i + 1) % 3], sink, d, nbx, nby); }, 30);
            printf("tile 16x16 halo d=%-2d 512 thr: %7.1f us  %5.2f TB/s\n", d, us, 2 * MB / us);
        }
    }
    // persistent, prefetching form of the 8 x 16 halo tile copy (grid = 256 CUs x k workgroups)
    for (int d : {1, 2, 8, 16}) {
        const int nbx = (W + 16 * d - 1) / (16 * d), nby = (H + 8 * d - 1) / (8 * d);
        const int nt = nbx * nby * d * d * B;
        for (int k : {4, 6, 8}) {
            us = timeit([&](int i) { k_phase_persist<8, 16, 256><<<256 * k, 256>>>(buf[i % 3], buf[(i + 1) % 3], sink, d, nbx, nby, nt); }, 30);
            printf("persistent 8x16 halo d=%-2d %d WG/CU: %7.1f us  %5.2f TB/s\n", d, k, us, 2 * MB / us);
        }
    }
    return 0;
}
