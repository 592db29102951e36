// Micro-benchmark: issue rate of v_mfma_f32_4x4x1_16B_f32 (16 blocks of 4x4, K = 1: 512 FLOP) against
// v_mfma_f32_16x16x4_f32 (2048 FLOP), one wave per SIMD, 1 / 2 / 4 independent accumulator chains.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int KIND, int CH>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    floatx4 a[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) a[c] = (floatx4){0, 0, 0, 0};
    float x = (float)threadIdx.x * 1e-3f, y = 1.0f + x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (KIND == 0) a[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a[c], 0, 0, 0);
                else a[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a[c], 0, 0, 0);
            }
    }
    float r = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) r += a[c][0] + a[c][1] + a[c][2] + a[c][3];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int KIND, int CH>
static void run(float *out)
{
    const int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    k<KIND, CH><<<256, 256>>>(out, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<KIND, CH><<<256, 256>>>(out, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double n = 256.0 * 4 * iters * 16 * CH, flop = n * (KIND == 0 ? 2048 : 512);
    printf("%s chains=%d: %.3f ms, %.1f TF, %.1f cycles per instruction per SIMD at 2.4 GHz\n", KIND == 0 ? "16x16x4   " : "4x4x1_16B ", CH,
           ms, flop / ms * 1e-9, ms * 1e-3 * 2.4e9 / (iters * 16.0 * CH));
}

int main()
{
    float *out;
    (void)hipMalloc(&out, 256 * 256 * 4);
    run<0, 1>(out); run<0, 2>(out); run<0, 4>(out);
    run<1, 1>(out); run<1, 2>(out); run<1, 4>(out); run<1, 8>(out);
    return 0;
}
