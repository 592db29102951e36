// Micro-test: does a kernel find the PREVIOUS kernel's output in its XCD's L2 when the same XCD produced it?
// k_write: workgroup b writes slice b (SLICE KB) of a buffer; k_read: workgroup b reads slice (b + shift) % n and sums it.
// Under round-robin dispatch workgroup b runs on XCD b % 8, so
//   shift 0    the reader of a slice is on the producer's XCD (and, typically, CU)
//   shift 8    the producer's XCD, another CU
//   shift 1    another XCD: the data must come over the fabric (Infinity Cache / HBM)
// for a buffer that fits the eight L2s (16 MB) and one that does not (128 MB); write-back and write-through (sc0 sc1) stores.
// If shift 0 / 8 are much faster than shift 1 for the small buffer, an XCD-consistent tile map across consecutive layers
// (a consumer tile on the XCD that produced its input) would turn fabric reads into L2 hits.
//   hipcc --offload-arch=gfx950 -O3 -o l2_reuse tools/micro/l2_reuse.hip && ./l2_reuse
#include <hip/hip_runtime.h>
#include <stdio.h>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("%s failed: %s\n", #x, hipGetErrorString(e_));                  \
            return 1;                                                              \
        }                                                                          \
    } while (0)

constexpr int SLICE_F4 = 1024;   // float4 per slice = 16 KB, 256 threads x 4 float4

__global__ __launch_bounds__(256) void k_write(float4 *buf, float v, int wt)
{
    float4 *p = buf + (size_t)blockIdx.x * SLICE_F4 + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 x = make_float4(v, v + 1.f, v + 2.f, v + (float)i);
        if (wt) {
            typedef float fx4_ __attribute__((ext_vector_type(4)));
            const fx4_ r = {x.x, x.y, x.z, x.w};
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p + 256 * i), "v"(r) : "memory");
        } else {
            p[256 * i] = x;
        }
    }
}

__global__ __launch_bounds__(256) void k_read(const float4 *buf, float *out, int shift)
{
    const int b = (blockIdx.x + shift) % gridDim.x;
    const float4 *p = buf + (size_t)b * SLICE_F4 + threadIdx.x;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 x = p[256 * i];
        s += x.x + x.y + x.z + x.w;
    }
    if (s == -1.f) out[0] = s;     // never true: keeps the loads
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    float *out;
    CK(hipMalloc(&out, 64));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0));
    CK(hipEventCreate(&t1));
    for (int mb : {16, 128}) {
        const int n = mb * 1024 / 16;            // slices
        float4 *buf;
        CK(hipMalloc(&buf, (size_t)n * SLICE_F4 * 16));
        for (int wt = 0; wt < 2; ++wt)
            for (int shift : {0, 8, 1, 3, 16}) {
                float best = 1e30f;
                for (int rep = 0; rep < 20; ++rep) {
                    hipLaunchKernelGGL(k_write, dim3(n), dim3(256), 0, s, buf, (float)rep, wt);
                    CK(hipEventRecord(t0, s));
                    hipLaunchKernelGGL(k_read, dim3(n), dim3(256), 0, s, buf, out, shift);
                    CK(hipEventRecord(t1, s));
                    CK(hipStreamSynchronize(s));
                    float ms = 0;
                    CK(hipEventElapsedTime(&ms, t0, t1));
                    if (ms < best) best = ms;
                }
                printf("%4d MB  %s stores  reader shift %2d: k_read %7.2f us = %6.2f TB/s\n", mb, wt ? "write-through" : "write-back   ", shift,
                       1e3 * best, (double)n * SLICE_F4 * 16 / (best * 1e-3) / 1e12);
            }
        CK(hipFree(buf));
    }
    return 0;
}
