// Micro-benchmark: can a packed-float32 VALU form of the 8 -> 8 Conv3D layer beat k_conv3d_mid8q's MFMA form?
// v_mfma_f32_4x4x1_16B_f32 issues at ~10 cycles per instruction = 51 useful FLOP/clk/SIMD (tools/micro/mfma4x4_bcast.hip);
// v_pk_fma_f32 is 2 FMAs x 64 lanes per 4-cycle issue = 64 FLOP/clk/SIMD on paper, every FLOP useful, bit-identical to the
// scalar fmaf chain.  Priced here before any kernel is written:
//   (1) rate:  independent v_pk_fma_f32 chains, all-VGPR operands and with one SGPR-pair operand (the weights of a conv are
//              wave-uniform), at 1 / 2 / 4 / 8 waves per SIMD;
//   (2) conv:  the real inner loop -- thread = V voxels of an LDS tile (channels-last, two float4 half-planes per voxel as in
//              k_conv3d_mid8q), 27 taps x 8 input channels x 8 output channels, weights streamed through the scalar cache
//              (wave-uniform addresses -> s_load), accumulators = 4 register pairs per voxel.
// Prints cycles per instruction per SIMD and useful FLOP/clk/SIMD (s_memtime = shader clock).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o pkfma_conv tools/micro/pkfma_conv.hip && ./pkfma_conv
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned long long clk()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

template <bool SGPR>
__global__ __launch_bounds__(1024) void k_rate(float *out, unsigned long long *cyc, const float *w, int iters)
{
    f2 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f2){(float)threadIdx.x * 1e-3f, 1.0f};
    f2 x = {1.0f + (float)threadIdx.x * 1e-6f, 0.5f};
    f2 ws[8];
    for (int i = 0; i < 8; ++i) ws[i] = (f2){w[2 * i], w[2 * i + 1]};          // uniform address: s_load -> SGPR pair
    const unsigned long long t0 = clk();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_elementwise_fma(x, SGPR ? ws[i] : acc[(i + 1) & 7] * 0.0f + ws[i], acc[i]);
    }
    const unsigned long long t1 = clk();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

// conv-shaped loop.  LDS: sA[half][voxel] float4 (half = input channels 0-3 / 4-7), halo tile HD x HY x HX with the lane's
// voxel at (d, y, x): a wave owns 64 consecutive x of V rows.  Weights: [tap][cin][cout] floats, read through uniform addresses.
template <int V>
__global__ __launch_bounds__(256) void k_conv(const float *__restrict__ w, float *out, unsigned long long *cyc, int reps)
{
    constexpr int HX = 66, HY = 2 * V + 2 + 6, HD = 3, NV = HD * HY * HX;
    __shared__ float4 sA[2][NV];
    for (int i = threadIdx.x; i < 2 * NV; i += 256) (&sA[0][0])[i] = make_float4(0.001f * i, 1.0f, 0.5f, 0.25f);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f2 acc[V][4];
    for (int v = 0; v < V; ++v)
        for (int c = 0; c < 4; ++c) acc[v][c] = (f2){0.0f, 0.0f};
    const unsigned long long t0 = clk();
    for (int r = 0; r < reps; ++r) {
#pragma unroll 1
        for (int kd = 0; kd < 3; ++kd)
#pragma unroll 1
            for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const float *wt = w + (((kd * 3 + kh) * 3 + kw) * 64);          // [cin][cout] of this tap: uniform
                    float4 a[V][2];
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        const int vox = (kd * HY + (wave * V + v + kh)) * HX + lane + kw;
                        a[v][0] = sA[0][vox];
                        a[v][1] = sA[1][vox];
                    }
#pragma unroll
                    for (int cin = 0; cin < 8; ++cin) {
#pragma unroll
                        for (int cp = 0; cp < 4; ++cp) {
                            const f2 ww = {wt[cin * 8 + 2 * cp], wt[cin * 8 + 2 * cp + 1]};
#pragma unroll
                            for (int v = 0; v < V; ++v) {
                                const float4 q = a[v][cin >> 2];
                                const float av = (cin & 3) == 0 ? q.x : (cin & 3) == 1 ? q.y : (cin & 3) == 2 ? q.z : q.w;
                                acc[v][cp] = __builtin_elementwise_fma((f2){av, av}, ww, acc[v][cp]);
                            }
                        }
                    }
                }
            }
    }
    const unsigned long long t1 = clk();
    float s = 0;
    for (int v = 0; v < V; ++v)
        for (int c = 0; c < 4; ++c) s += acc[v][c].x + acc[v][c].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

static double median(std::vector<unsigned long long> v)
{
    std::sort(v.begin(), v.end());
    return (double)v[v.size() / 2];
}

int main()
{
    float *out, *w;
    unsigned long long *cyc;
    (void)hipMalloc(&out, 256 * 4096 * 4);
    (void)hipMalloc(&cyc, 65536 * 8);
    std::vector<float> hw(27 * 64);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0.001f * (float)(i % 97) - 0.03f;
    (void)hipMalloc(&w, hw.size() * 4);
    (void)hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    const int iters = 2000;
    for (int sg = 0; sg < 2; ++sg)
        for (int wps : {1, 2, 4, 8}) {
            const int threads = 256 * wps > 1024 ? 1024 : 256 * wps, blocks = 256 * (256 * wps / threads);
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0);
            (void)hipEventCreate(&e1);
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(e0, 0);
                if (sg) k_rate<true><<<blocks, threads>>>(out, cyc, w, iters * 10);
                else k_rate<false><<<blocks, threads>>>(out, cyc, w, iters * 10);
                (void)hipEventRecord(e1, 0);
                (void)hipDeviceSynchronize();
                (void)hipEventElapsedTime(&ms, e0, e1);
            }
            // wall clock: every thread runs iters*10 x 8 packed FMAs (the all-VGPR form: + a packed multiply and add each)
            printf("      wall: %.3f ms for %.3g packed FMAs = %.1f TFLOP/s in FMAs alone\n", ms,
                   (double)blocks * threads * iters * 10 * 8, (double)blocks * threads * iters * 10 * 8 * 4 / (ms * 1e-3) / 1e12);
            for (int rep = 0; rep < 2; ++rep) {
                if (sg) k_rate<true><<<blocks, threads>>>(out, cyc, w, iters);
                else k_rate<false><<<blocks, threads>>>(out, cyc, w, iters);
                (void)hipDeviceSynchronize();
            }
            const int nw = blocks * threads / 64;
            std::vector<unsigned long long> h(nw);
            (void)hipMemcpy(h.data(), cyc, nw * 8, hipMemcpyDeviceToHost);
            const double c = median(h);
            const double per_simd = c / ((double)iters * 8 * wps);      // cycles per v_pk_fma_f32 per SIMD
            printf("rate  %s  %d waves/SIMD: %.2f cycles per v_pk_fma_f32 per SIMD = %.1f FLOP/clk/SIMD\n",
                   sg ? "SGPR-pair weight operand" : "all-VGPR operands       ", wps, per_simd, 256.0 / per_simd);
        }
    for (int V : {1, 2}) {
        for (int wgs_per_cu : {1, 2, 4}) {
            const int reps = 40, blocks = 256 * wgs_per_cu;
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0);
            (void)hipEventCreate(&e1);
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(e0, 0);
                if (V == 1) k_conv<1><<<blocks, 256>>>(w, out, cyc, reps * 10);
                else k_conv<2><<<blocks, 256>>>(w, out, cyc, reps * 10);
                (void)hipEventRecord(e1, 0);
                (void)hipDeviceSynchronize();
                (void)hipEventElapsedTime(&ms, e0, e1);
            }
            printf("      wall: %.3f ms = %.1f useful TFLOP/s over the whole chip (k_conv3d_mid8q at B=8: 100)\n", ms,
                   (double)blocks * 256 * reps * 10 * 27.0 * 64.0 * V * 2.0 / (ms * 1e-3) / 1e12);
            for (int rep = 0; rep < 2; ++rep) {
                if (V == 1) k_conv<1><<<blocks, 256>>>(w, out, cyc, reps);
                else k_conv<2><<<blocks, 256>>>(w, out, cyc, reps);
                (void)hipDeviceSynchronize();
            }
            std::vector<unsigned long long> h(blocks * 4);
            (void)hipMemcpy(h.data(), cyc, blocks * 4 * 8, hipMemcpyDeviceToHost);
            const double c = median(h);
            // per SIMD: wgs_per_cu waves (one wave of each workgroup per SIMD), each reps x 27 x 8 x 8 x V x 64 FMAs = x2 FLOP
            const double flop = (double)wgs_per_cu * reps * 27.0 * 64.0 * V * 64.0 * 2.0;
            printf("conv  V=%d voxels per lane, %d waves/SIMD: %.0f cycles per wave for %d layers-worth; %.1f useful FLOP/clk/SIMD "
                   "(k_conv3d_mid8q's 4x4x1 MFMA: 51)\n", V, wgs_per_cu, c, reps, flop / c);
        }
    }
    return 0;
}
