// Device math shared by the regression kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lws {

// XCD-aware block -> tile map: blocks b and b+8 share an XCD (round-robin dispatch), so give every XCD a
// contiguous run of tiles; neighbouring tiles then find each other's halo voxels in the same L2.
// Speed only: any placement is correct.
__device__ __forceinline__ int xcd_tile(int b, int nb)
{
    const int q = nb >> 3, r = nb & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

// ---- split-bf16 ("bf16x3") helpers of the opt-in numerics mode (k_conv3d_mid16x, k_ref_conv64x) -------------------------
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// float32 -> bf16 bits, round to nearest even (finite values; the activations are BN + ReLU outputs)
__host__ __device__ __forceinline__ uint32_t f2bf_bits(float x)
{
    union { float f; uint32_t u; } v;
    v.f = x;
    return (v.u + 0x7FFFu + ((v.u >> 16) & 1u)) >> 16;
}
__host__ __device__ __forceinline__ float bf_bits2f(uint32_t h)
{
    union { float f; uint32_t u; } v;
    v.u = h << 16;
    return v.f;
}
// x -> (hi, mid, lo) bf16 bit patterns; every subtraction is exact in float32
__host__ __device__ __forceinline__ void split_bf16x3(float x, uint32_t &hi, uint32_t &mid, uint32_t &lo)
{
    hi = f2bf_bits(x);
    const float r1 = x - bf_bits2f(hi);
    mid = f2bf_bits(r1);
    const float r2 = r1 - bf_bits2f(mid);
    lo = f2bf_bits(r2);
}

// device form for two values at once: v_cvt_pk_bf16_f32 (round to nearest even, the same rounding as f2bf_bits) -- 11
// instructions per pair instead of ~40 for the bit arithmetic; result dwords hold (x0's, x1's) bf16 in (low, high) halves
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b)
{
    typedef float f2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 b2_ __attribute__((ext_vector_type(2)));
    const f2_ v = {a, b};
    const b2_ r = __builtin_convertvector(v, b2_);
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ void split_bf16x3_pair(float x0, float x1, uint32_t &h, uint32_t &m, uint32_t &l)
{
    h = pack_bf16x2(x0, x1);
    float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    m = pack_bf16x2(r0, r1);
    r0 = r0 - __uint_as_float(m << 16);
    r1 = r1 - __uint_as_float(m & 0xffff0000u);
    l = pack_bf16x2(r0, r1);
}

__device__ __forceinline__ bf16x8 as_bf16x8(const uint4 &u)
{
    union { uint4 u4; bf16x8 v; } cv;
    cv.u4 = u;
    return cv.v;
}
// One term of  acc += W * X  for a 16 x 16 tile and a K = 32 block, W and X given as their (hi, mid, lo) bf16 fragments: the six
// cross terms of total order <= 2, smallest first -- T = 0 lo*hi, 1 hi*lo, 2 mid*mid, 3 mid*hi, 4 hi*mid, 5 hi*hi -- each
// product exact in float32 and summed by v_mfma_f32_16x16x32_bf16.  Callers run T in the OUTER loop over their accumulators
// (T a compile-time constant after unrolling): consecutive instructions then belong to different accumulators and issue every
// 16 cycles; the six terms of one accumulator back to back are paced by the dependency (18-25 cycles each, tools/stamps.py).
__device__ __forceinline__ floatx4 mfma_split_bf16_term(floatx4 a, const uint4 (&w)[3], const uint4 (&x)[3], int T)
{
    const int wi = T == 0 ? 2 : (T == 2 || T == 3) ? 1 : 0;
    const int xi = T == 1 ? 2 : (T == 2 || T == 4) ? 1 : 0;
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(w[wi]), as_bf16x8(x[xi]), a, 0, 0, 0);
}

// ---- packed float32 math (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two independent IEEE operations per lane and
// instruction, the way the chip reaches its vector-FP32 peak).  Bit-identical to the scalar operation on each half, so a
// kernel may pack any two INDEPENDENT chains (two output channels, two pixels) without changing a bit.
typedef float f2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2_t pk_fma(f2_t a, f2_t b, f2_t c) { return __builtin_elementwise_fma(a, b, c); }
// acc.{x,y,z,w} = fmaf(a.{x,..}, w.{x,..}, acc.{x,..}) as two packed instructions
__device__ __forceinline__ void fma4(float4 &acc, const float4 &a, const float4 &w)
{
    const f2_t lo = pk_fma((f2_t){a.x, a.y}, (f2_t){w.x, w.y}, (f2_t){acc.x, acc.y});
    const f2_t hi = pk_fma((f2_t){a.z, a.w}, (f2_t){w.z, w.w}, (f2_t){acc.z, acc.w});
    acc = make_float4(lo.x, lo.y, hi.x, hi.y);
}
// acc.{x,..} = fmaf(v, w.{x,..}, acc.{x,..}) (one scalar against four weights)
__device__ __forceinline__ void fma4s(float4 &acc, float v, const float4 &w)
{
    const f2_t vv = {v, v};
    const f2_t lo = pk_fma(vv, (f2_t){w.x, w.y}, (f2_t){acc.x, acc.y});
    const f2_t hi = pk_fma(vv, (f2_t){w.z, w.w}, (f2_t){acc.z, acc.w});
    acc = make_float4(lo.x, lo.y, hi.x, hi.y);
}
// max(fmaf(x, s, t), 0) on four channels: BatchNorm(eval) + ReLU
__device__ __forceinline__ float4 bn_relu4(const float4 &x, const float4 &s, const float4 &t)
{
    const f2_t z = {0.0f, 0.0f};
    const f2_t lo = __builtin_elementwise_max(pk_fma((f2_t){x.x, x.y}, (f2_t){s.x, s.y}, (f2_t){t.x, t.y}), z);
    const f2_t hi = __builtin_elementwise_max(pk_fma((f2_t){x.z, x.w}, (f2_t){s.z, s.w}, (f2_t){t.z, t.w}), z);
    return make_float4(lo.x, lo.y, hi.x, hi.y);
}

// exp(x) for x <= 0 from IEEE mul / fma / rint only (Cephes expf polynomial), so the result is a
// pure function of the float32 input on any IEEE machine; returns 0 below -80 (e^-80 ~ 1.8e-35).
__device__ __forceinline__ float lws_expf(float x)
{
    if (x < -80.0f) return 0.0f;
    float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float y = fmaf(p, r2, r) + 1.0f;
    int32_t bits = __float_as_int(y) + (((int32_t)n) << 23);
    return __int_as_float(bits);
}

// Bilinear source index of F.interpolate(align_corners=False) -- the ONE place every resize of the path gets its taps from
// (k_volume_l1_warp's prologue, k_upsample_add, k_softargmin_upsample, DeferredMap and through it the fused consumers).
// `off` carries lws_config.interp_align_mode: 0.5f = align_mode 0, half-pixel centres, src = ratio * (dst + 0.5) - 0.5 (what
// the oracle bets Paddle 2.0rc0 does); 0.0f = align_mode 1, src = ratio * dst (Paddle 1.x / 2.0-beta) --
// the same expression, since x + 0 and x - 0 are exact.  /root/reference/models/models.py:119,146,154,161.
__device__ __forceinline__ void src_index(int dst, float ratio, int in, int &i0, int &i1, float &l0, float &l1, float off)
{
    float s = ratio * ((float)dst + off) - off;
    if (s < 0.0f) s = 0.0f;
    int a = (int)s;
    if (a > in - 1) a = in - 1;
    i0 = a;
    i1 = (a < in - 1) ? a + 1 : a;
    l1 = s - (float)a;
    l0 = 1.0f - l1;
}

// A "deferred" full-resolution disparity map: pred(Y,X) = upsample(low * H / h)(Y,X) + prev(Y,X) evaluated on demand
// with exactly the operations of k_upsample_add (models.py:145-148,153-156), so a consumer can read the map before
// the kernel that materialises it has run.  low == nullptr: the map is materialised, read prev directly.
// Two levels (round 5): when prev == nullptr and low0 != nullptr the previous map is itself deferred and has no
// predecessor -- prev(Y,X) = upsample(low0 * H / h0)(Y,X), stage 1's map seen from stage 3; prev == nullptr and
// low0 == nullptr: there is no previous map (stage 1's map seen from stage 2), nothing is added.
struct DeferredMap {
    const float *low;     // [h,w] of this image, or nullptr
    const float *prev;    // [H,W] of this image (the previous stage's map; the materialised map if low == nullptr), or nullptr
    int h, w;
    float mul_a, mul_b;   // (float)H, 1/(float)h
    const float *low0 = nullptr;   // [h0,w0]: the previous map's own low-resolution source (see above)
    int h0 = 0, w0 = 0;
    float mul_b0 = 0.0f;           // 1/(float)h0
    float off = 0.5f;              // src_index's offset (lws_config.interp_align_mode)
};

// upsample(low * mul_a * mul_b) at N points, every load issued before the first use: k_upsample_add's operations
template <int N>
__device__ __forceinline__ void upsample_at_n(const float *low, int h, int w, float mul_a, float mul_b, const int (&ys)[N],
                                              const int (&xs)[N], int H, int W, float (&out)[N], float off)
{
    const float rh = (float)h / (float)H, rw = (float)w / (float)W;
    float hy0[N], hy1[N], wx0[N], wx1[N], t[N][4];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        int y0, y1, x0, x1;
        src_index(ys[i], rh, h, y0, y1, hy0[i], hy1[i], off);
        src_index(xs[i], rw, w, x0, x1, wx0[i], wx1[i], off);
        t[i][0] = low[y0 * w + x0];
        t[i][1] = low[y0 * w + x1];
        t[i][2] = low[y1 * w + x0];
        t[i][3] = low[y1 * w + x1];
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float p00 = (t[i][0] * mul_a) * mul_b;
        const float p01 = (t[i][1] * mul_a) * mul_b;
        const float p10 = (t[i][2] * mul_a) * mul_b;
        const float p11 = (t[i][3] * mul_a) * mul_b;
        const float top = p00 * wx0[i] + p01 * wx1[i];
        const float bot = p10 * wx0[i] + p11 * wx1[i];
        out[i] = hy0[i] * top + hy1[i] * bot;
    }
}

// N points of the same map at once: the operations of k_upsample_add per point, but every load of every point is issued
// before the first use (four back-to-back single-point evaluations cost four dependent memory round trips: 8.0k of the 10.4k
// cycles a k_volume_l1_warp workgroup lived, r03).  prev_out (optional): the previous map's values at the points (what a
// consumer writes out when that map is deferred too).
template <int N>
__device__ __forceinline__ void deferred_at_n(const DeferredMap &m, const int (&ys)[N], const int (&xs)[N], int H, int W,
                                              float (&out)[N], float *prev_out = nullptr)
{
    float pv[N];
    const bool have_prev = m.prev != nullptr || m.low0 != nullptr;      // (kernel-uniform)
    if (m.prev != nullptr) {
#pragma unroll
        for (int i = 0; i < N; ++i) pv[i] = m.prev[(int64_t)ys[i] * W + xs[i]];
    } else if (m.low0 != nullptr) {
        upsample_at_n<N>(m.low0, m.h0, m.w0, m.mul_a, m.mul_b0, ys, xs, H, W, pv, m.off);
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) pv[i] = 0.0f;
    }
    if (prev_out != nullptr) {
#pragma unroll
        for (int i = 0; i < N; ++i) prev_out[i] = pv[i];
    }
    if (m.low == nullptr) {
#pragma unroll
        for (int i = 0; i < N; ++i) out[i] = pv[i];
        return;
    }
    float v[N];
    upsample_at_n<N>(m.low, m.h, m.w, m.mul_a, m.mul_b, ys, xs, H, W, v, m.off);
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = have_prev ? v[i] + pv[i] : v[i];
}

// sum_k softmax_k(-c) * (start + k) over D values c[k*stride]: max-subtracted, S summed ascending,
// p_k = e_k / S (IEEE division), expectation summed ascending.  e_k is recomputed in the third
// pass instead of being kept in a D-sized register array (it is a pure function, so identical).
__device__ __forceinline__ float softargmin_pixel(const float *c, int64_t stride, int D, float start)
{
    float m = -c[0];
    for (int k = 1; k < D; ++k) m = fmaxf(m, -c[(int64_t)k * stride]);
    float S = 0.0f;
    for (int k = 0; k < D; ++k) S = S + lws_expf(-c[(int64_t)k * stride] - m);
    float acc = 0.0f;
    for (int k = 0; k < D; ++k) {
        float pk = lws_expf(-c[(int64_t)k * stride] - m) / S;
        acc = acc + pk * (start + (float)k);
    }
    return acc;
}

}  // namespace lws
