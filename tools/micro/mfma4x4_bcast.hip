// Micro-test + micro-benchmark for k_conv3d_mid8q's building block: v_mfma_f32_4x4x1_16B_f32 with the A-block broadcast
// (CBSZ = 4, ABID = k).  (1) semantics: D_b[i][j] (register i of lane 4b + j) += A_k[i] * B_b[j] for every block b, with
// A_k[i] taken from lane 4k + i; a single fma per element (K = 1).  (2) issue rate with 2 accumulator chains per wave at
// 1 / 2 / 4 / 6 waves per SIMD, against the same loop without broadcast and against v_mfma_f32_16x16x4_f32.
//   hipcc --offload-arch=gfx950 -O3 -o mfma4x4_bcast tools/micro/mfma4x4_bcast.hip && ./mfma4x4_bcast
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
typedef float floatx4 __attribute__((ext_vector_type(4)));

__global__ void k_sem(const float *ab, float *out)     // ab: [2][64] operands prepared on the host (no device rounding of them)
{
    const int l = threadIdx.x;
    const float a = ab[l], b = ab[64 + l];
    floatx4 c0 = {0.5f, 0.25f, 0.125f, 1.0f}, c5 = c0, c15 = c0;
    c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 4, 0, 0);
    c5 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c5, 4, 5, 0);
    c15 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c15, 4, 15, 0);
    for (int i = 0; i < 4; ++i) {
        out[(0 * 64 + l) * 4 + i] = c0[i];
        out[(1 * 64 + l) * 4 + i] = c5[i];
        out[(2 * 64 + l) * 4 + i] = c15[i];
    }
}

template <int KIND>
__global__ __launch_bounds__(1024) void k_rate(float *out, int iters)
{
    floatx4 lo = {0, 0, 0, 0}, hi = {0, 0, 0, 0};
    float x = (float)threadIdx.x * 1e-3f, y = 1.0f + x;
    for (int i = 0; i < iters; ++i) {
#define Q(k)                                                                 \
    if (KIND == 0) {                                                         \
        lo = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, lo, 4, 2 * (k), 0);    \
        hi = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, hi, 4, 2 * (k) + 1, 0);\
    } else if (KIND == 1) {                                                  \
        lo = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, lo, 0, 0, 0);          \
        hi = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, hi, 0, 0, 0);          \
    } else {                                                                 \
        lo = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, lo, 0, 0, 0);        \
        hi = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, hi, 0, 0, 0);        \
    }
        Q(0) Q(1) Q(2) Q(3) Q(4) Q(5) Q(6) Q(7)
#undef Q
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = lo[0] + lo[1] + lo[2] + lo[3] + hi[0] + hi[1] + hi[2] + hi[3];
}

template <int KIND>
static void run(float *out, int waves_per_simd)
{
    const int iters = 4000, threads = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    dim3 grid(256), block(threads > 1024 ? 1024 : threads);
    if (threads > 1024) grid = dim3(256 * threads / 1024);       // 6 waves/SIMD: 1.5 blocks of 1024 per CU -> use 512-thread blocks
    if (waves_per_simd == 6) { grid = dim3(768); block = dim3(512); }
    k_rate<KIND><<<grid, block>>>(out, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k_rate<KIND><<<grid, block>>>(out, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double n_simd = (double)iters * 16 * waves_per_simd;        // instructions per SIMD
    const double flop = n_simd * 1024 * (KIND == 2 ? 2048.0 : 512.0);
    printf("%-22s waves/SIMD=%d: %.3f ms, %6.1f TF, %5.2f cycles per instruction per SIMD at 2.4 GHz\n",
           KIND == 0 ? "4x4x1_16B cbsz=4 abid" : KIND == 1 ? "4x4x1_16B" : "16x16x4", waves_per_simd, ms, flop / ms * 1e-9,
           ms * 1e-3 * 2.4e9 / n_simd);
}

int main()
{
    float *out, host[3 * 64 * 4];
    (void)hipMalloc(&out, 1 << 22);
    float ab[128], *dab;
    for (int l = 0; l < 64; ++l) {
        ab[l] = 1.0f + 0.37f * (float)l + 1e-3f * (float)(l * l % 7);
        ab[64 + l] = 2.0f - 0.011f * (float)l + 3e-4f * (float)(l % 5);
    }
    (void)hipMalloc(&dab, sizeof(ab));
    (void)hipMemcpy(dab, ab, sizeof(ab), hipMemcpyHostToDevice);
    k_sem<<<1, 64>>>(dab, out);
    (void)hipMemcpy(host, out, sizeof(host), hipMemcpyDeviceToHost);
    const int ids[3] = {0, 5, 15};
    const float cinit[4] = {0.5f, 0.25f, 0.125f, 1.0f};
    int bad = 0;
    for (int t = 0; t < 3; ++t)
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 4; ++i) {
                const float want = fmaf(ab[4 * ids[t] + i], ab[64 + l], cinit[i]);      // ONE rounding
                if (host[(t * 64 + l) * 4 + i] != want) {
                    if (bad < 8) printf("MISMATCH abid=%d lane=%d reg=%d: got %.9g want %.9g\n", ids[t], l, i, host[(t * 64 + l) * 4 + i], want);
                    ++bad;
                }
            }
    printf("semantics (D[i] of lane l = fmaf(A[lane 4*abid + i], B[lane l], C[i]), bit-exact): %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
    for (int w : {1, 2, 4, 6}) run<0>(out, w);
    for (int w : {1, 4}) run<1>(out, w);
    for (int w : {1, 4}) run<2>(out, w);
    return bad ? 1 : 0;
}
