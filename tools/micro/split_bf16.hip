// Pricing VERDICT r2's optional item 8 before building anything: a split-bf16 ("bf16x3") form of the 32 -> 32 Conv3D contraction.
// Every float32 operand x is split into three bf16 values hi + mid + lo (24 mantissa bits in all); the product a*b is
// approximated by the six cross terms of total order <= 2 (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid), each exact in float32,
// accumulated by v_mfma_f32_16x16x32_bf16 (16 cycles per 16x16x32 against 32 cycles per 16x16x4 for the f32-input MFMA).
//   (1) accuracy: one 16 x 16 output tile with K = 27 taps x 32 channels = 864 (a stage-1 middle layer), ReLU'd activations,
//       Kaiming weights: |split-bf16 - float64| and |float32 fma chain - float64| (the contract the kernels and the oracle share);
//   (2) issue rate: 6 bf16 MFMAs per (tap, 32-channel block) against 8 f32 MFMAs, operands in registers, 1 and 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o split_bf16 tools/micro/split_bf16.hip && ./split_bf16
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__host__ __device__ static inline uint16_t f2bf(float x)      // round to nearest even (finite inputs)
{
    union { float f; uint32_t u; } v;
    v.f = x;
    return (uint16_t)((v.u + 0x7FFFu + ((v.u >> 16) & 1u)) >> 16);
}
__host__ __device__ static inline float bf2f(uint16_t h)
{
    union { float f; uint32_t u; } v;
    v.u = (uint32_t)h << 16;
    return v.f;
}
__host__ __device__ static inline void split3(float x, uint16_t &hi, uint16_t &mid, uint16_t &lo)
{
    hi = f2bf(x);
    const float r1 = x - bf2f(hi);
    mid = f2bf(r1);
    const float r2 = r1 - bf2f(mid);
    lo = f2bf(r2);
}

constexpr int K = 864, KB = K / 32;

// one wave: A [16][K] weights, B [K][16] activations -> out32 (fma chain, k ascending), outbf (split-bf16 through MFMA)
__global__ void k_acc(const float *A, const float *B, float *out32, float *outbf)
{
    const int l = threadIdx.x, rc = l & 15, g = l >> 4;
    // (a) the contract: v_mfma_f32_16x16x4_f32 == a k-ordered fma chain
    floatx4 c = {0.f, 0.f, 0.f, 0.f};
    for (int k4 = 0; k4 < K / 4; ++k4) c = __builtin_amdgcn_mfma_f32_16x16x4f32(A[rc * K + 4 * k4 + g], B[(4 * k4 + g) * 16 + rc], c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) out32[(4 * g + i) * 16 + rc] = c[i];
    // (b) split-bf16: per 32-deep K block six MFMAs into ONE accumulator, largest terms last within a block? no: hi*hi first
    floatx4 d = {0.f, 0.f, 0.f, 0.f};
    for (int kb = 0; kb < KB; ++kb) {
        bf16x8 ah, am, al, bh, bm, bl;
        for (int j = 0; j < 8; ++j) {
            uint16_t h, m, lo_;
            split3(A[rc * K + kb * 32 + 8 * g + j], h, m, lo_);
            ah[j] = (short)h; am[j] = (short)m; al[j] = (short)lo_;
            split3(B[(kb * 32 + 8 * g + j) * 16 + rc], h, m, lo_);
            bh[j] = (short)h; bm[j] = (short)m; bl[j] = (short)lo_;
        }
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, d, 0, 0, 0);      // smallest terms first
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, d, 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) outbf[(4 * g + i) * 16 + rc] = d[i];
}

template <int KIND>
__global__ __launch_bounds__(512) void k_rate(float *out, int iters)
{
    floatx4 acc[4];
    for (auto &a : acc) a = (floatx4){0.f, 0.f, 0.f, 0.f};
    const float x = 1e-3f * threadIdx.x, y = 1.0f + x;
    bf16x8 p, q;
    for (int j = 0; j < 8; ++j) { p[j] = (short)f2bf(x + j); q[j] = (short)f2bf(y - j); }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {        // 4 independent accumulators (rows x cout tiles of one wave)
            if (KIND == 0) {
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[t], 0, 0, 0);
            } else {
#pragma unroll
                for (int u = 0; u < 6; ++u) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p, q, acc[t], 0, 0, 0);
            }
        }
    }
    float r = 0;
    for (auto &a : acc) r += a[0] + a[1] + a[2] + a[3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND>
static double rate(float *out, int wps)
{
    const int iters = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    k_rate<KIND><<<256, 256 * wps>>>(out, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k_rate<KIND><<<256, 256 * wps>>>(out, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3 * 2.4e9 / ((double)iters * 4 * wps);     // cycles at 2.4 GHz per (tap, 32-channel block, accumulator) per SIMD
}

int main()
{
    float *hA = (float *)malloc(16 * K * 4), *hB = (float *)malloc(K * 16 * 4);
    srand(7);
    auto rnd = [] { return (rand() + 0.5) / ((double)RAND_MAX + 1.0); };
    auto gauss = [&] { return sqrt(-2.0 * log(rnd())) * cos(6.283185307179586 * rnd()); };
    for (int i = 0; i < 16 * K; ++i) hA[i] = (float)(gauss() * sqrt(2.0 / K));                 // Kaiming normal, fan_in = 864
    for (int i = 0; i < K * 16; ++i) hB[i] = (float)fmax(gauss() * 1.5 + 0.3, 0.0);              // BN + ReLU'd activations
    float *dA, *dB, *d32, *dbf, *sink;
    (void)hipMalloc(&dA, 16 * K * 4); (void)hipMalloc(&dB, K * 16 * 4); (void)hipMalloc(&d32, 1024); (void)hipMalloc(&dbf, 1024);
    (void)hipMalloc(&sink, 256 * 512 * 4);
    (void)hipMemcpy(dA, hA, 16 * K * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dB, hB, K * 16 * 4, hipMemcpyHostToDevice);
    k_acc<<<1, 64>>>(dA, dB, d32, dbf);
    float o32[256], obf[256];
    (void)hipMemcpy(o32, d32, 1024, hipMemcpyDeviceToHost);
    (void)hipMemcpy(obf, dbf, 1024, hipMemcpyDeviceToHost);
    double e32 = 0, ebf = 0, m32 = 0, mbf = 0, scale = 0, dmax = 0;
    int chain_ok = 1;
    for (int r = 0; r < 16; ++r)
        for (int c = 0; c < 16; ++c) {
            double ref = 0;
            float chain = 0.f;
            for (int k = 0; k < K; ++k) {
                ref += (double)hA[r * K + k] * (double)hB[k * 16 + c];
                chain = fmaf(hA[r * K + k], hB[k * 16 + c], chain);
            }
            if (chain != o32[r * 16 + c]) chain_ok = 0;
            const double a = fabs(o32[r * 16 + c] - ref), b = fabs(obf[r * 16 + c] - ref);
            e32 = fmax(e32, a); ebf = fmax(ebf, b); m32 += a / 256; mbf += b / 256;
            scale = fmax(scale, fabs(ref));
            dmax = fmax(dmax, fabs((double)obf[r * 16 + c] - (double)o32[r * 16 + c]));
        }
    printf("K = %d, output scale (max |ref|) %.3f; f32 MFMA == host fmaf chain bit for bit: %s\n", K, scale, chain_ok ? "yes" : "NO");
    printf("float32 fma chain   vs float64: max %.3e  mean %.3e\n", e32, m32);
    printf("split-bf16 (6 terms) vs float64: max %.3e  mean %.3e   (vs the float32 chain: max %.3e)\n", ebf, mbf, dmax);
    for (int w : {1, 2}) {
        const double c0 = rate<0>(sink, w), c1 = rate<1>(sink, w);
        printf("waves/SIMD=%d: f32 16x16x4 x 8 = %.1f cycles, bf16 16x16x32 x 6 = %.1f cycles per (tap, 32-channel block): %.2fx\n", w, c0, c1, c0 / c1);
    }
    return 0;
}
