// Micro-benchmark: do fp32 MFMA and fp32 VALU FMA co-issue on one SIMD at full rate each?
// WG = 4 "matrix" waves (one per SIMD) + NV "vector" waves; each matrix wave issues NM MFMAs (4 independent chains),
// each vector wave NF v_fma (8 independent chains, scalar multiplier).  Modes: 1 = matrix only, 2 = vector only, 3 = both.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int NV>
__global__ __launch_bounds__(256 + 64 * NV) void k(float *out, const float *sw, int mode, int iters)
{
    const int wave = threadIdx.x >> 6;
    if (wave < 4) {
        if (!(mode & 1)) return;
        floatx4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        float x = (float)threadIdx.x * 1e-3f, y = 1.0f + x;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, y, a3, 0, 0, 0);
            }
        }
        out[blockIdx.x * 1024 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
    } else {
        if (!(mode & 2)) return;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = (float)j;
        float v = (float)threadIdx.x * 1e-3f;
        for (int i = 0; i < iters; ++i) {
            // 32 MFMAs above = 1024 cycles; give the vector waves 16 scalar weights x 8 chains = 128 fma per iteration
            const float *w = sw + (i & 15) * 16;
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const float s = w[u];     // wave-uniform -> s_load
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(v, s, acc[j]);
            }
        }
        float r = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) r += acc[j];
        out[blockIdx.x * 1024 + threadIdx.x] = r;
    }
}

template <int NV>
static void run(float *out, float *sw, int iters)
{
    for (int mode = 1; mode <= 3; ++mode) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        k<NV><<<256, 256 + 64 * NV>>>(out, sw, mode, iters);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<NV><<<256, 256 + 64 * NV>>>(out, sw, mode, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double mf = (mode & 1) ? 256.0 * 4 * iters * 32 * 2048 : 0, vf = (mode & 2) ? 256.0 * NV * iters * 128 * 128 : 0;
        printf("NV=%d mode=%d: %.3f ms  matrix %.1f TF  vector %.1f TF  sum %.1f TF\n", NV, mode, ms, mf / ms * 1e-9, vf / ms * 1e-9,
               (mf + vf) / ms * 1e-9);
    }
}

int main()
{
    float *out, *sw;
    hipMalloc(&out, 256 * 1024 * 4);
    hipMalloc(&sw, 1024);
    hipMemset(sw, 0, 1024);
    const int iters = 2000;
    run<4>(out, sw, iters);
    run<8>(out, sw, iters);
    run<12>(out, sw, iters);
    return 0;
}
