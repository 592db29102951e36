/* TEST INFRASTRUCTURE ONLY -- deterministic C restatement of the LWSNet disparity
 * hot path.  PARITY UNPINNED: the reference is Python on PaddlePaddle 2.0.0rc0,
 * which cannot be installed here, and it holds no golden vectors for this path
 * (SURVEY.md section 8c).  This file pins the *arithmetic* -- every float32
 * operation and its order -- so that the HIP kernels can be compared bit for
 * bit; oracle/lws_oracle.py is the literal op-by-op restatement this file is
 * itself validated against (tests/test_oracle_cpu.py).
 *
 * Nothing under lwsnet_amd/ links or loads this file.  Build:
 *   gcc -O2 -fopenmp -ffp-contract=off -mfma -shared -fPIC lws_oracle.c -o _build/liblws_oracle.so -lm
 * (-ffp-contract=off: no implicit FMAs; every fused multiply-add below is an
 * explicit fmaf(), which is what v_fma_f32 / v_mfma_f32_16x16x4_f32 compute.)
 *
 * Reference lines followed (all under /root/reference/models/):
 *   lwso_volume_l1_shift   models.py:58-76
 *   lwso_resize_bilinear   models.py:119-121 (F.interpolate; align_mode: lwso_set_align_mode) + the two scalar scales
 *   lwso_volume_l1_warp    models.py:78-104 + warp :28-55 (grid_sample, align_corners=True, zeros)
 *   lwso_bnrelu_conv3d     submodules.py:190-204 (BatchNorm3D -> ReLU -> Conv3D k3 s1 p1, no bias)
 *   lwso_softargmin        models.py:142,151-152,167-179
 *   lwso_upsample_add      models.py:145-148,153-156
 */
#include <immintrin.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#define LWSO_API __attribute__((visibility("default")))

/* exp(x) for x <= 0, float32, built only from IEEE mul/fma/rint so that the
 * GPU kernel (same formula) is bit-identical.  Cephes expf polynomial. */
static inline float lwso_expf(float x)
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
    union { float f; int32_t i; } u;
    u.f = y;
    u.i += ((int32_t)n) << 23;
    return u.f;
}

/* BASELINE config 5 (not in the reference): round every value to fp16 (round-to-nearest-even) and back. */
LWSO_API void lwso_round_fp16(const float *x, float *y, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) y[i] = _cvtsh_ss(_cvtss_sh(x[i], _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC));
}

LWSO_API void lwso_expf_array(const float *x, float *y, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) y[i] = lwso_expf(x[i]);
}

/* cost[b,d,y,x] = sum_c |L[b,c,y,x] - (x>=d ? R[b,c,y,x-d] : 0)|, c ascending. */
LWSO_API void lwso_volume_l1_shift(const float *L, const float *R, float *cost,
                                   int B, int C, int h, int w, int D)
{
    const int64_t plane = (int64_t)h * w;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int d = 0; d < D; ++d)
            for (int y = 0; y < h; ++y)
                for (int x = 0; x < w; ++x) {
                    float acc = 0.0f;
                    for (int c = 0; c < C; ++c) {
                        int64_t o = ((int64_t)b * C + c) * plane + (int64_t)y * w;
                        float l = L[o + x];
                        float r = (x >= d) ? R[o + x - d] : 0.0f;
                        acc = acc + fabsf(l - r);
                    }
                    cost[((int64_t)b * D + d) * plane + (int64_t)y * w + x] = acc;
                }
}

/* Which source index F.interpolate(mode="bilinear", align_corners=False) uses (models.py:119,146,154,161) is a reading of
 * Paddle 2.0rc0's defaults that cannot be checked offline, so it is a switch here exactly as it is
 * in the product (lws_config.interp_align_mode) and in oracle/lws_oracle.py (VARIANT["align_mode"], which c_oracle.py
 * forwards before every resize):  0 = half-pixel centres, src = ratio * (dst + 0.5) - 0.5;  1 = src = ratio * dst
 * (Paddle's bilinear_interp with align_mode = 1: y_n = int(ratio * k), d_n = ratio * k - y_n).  One expression serves
 * both: with off = 0 the additions are exact. */
static float lwso_align_off = 0.5f;
LWSO_API void lwso_set_align_mode(int mode) { lwso_align_off = mode == 1 ? 0.0f : 0.5f; }
LWSO_API int lwso_get_align_mode(void) { return lwso_align_off == 0.0f ? 1 : 0; }

static inline void lwso_src(int dst, float ratio, int in, int *i0, int *i1, float *l0, float *l1)
{
    const float off = lwso_align_off;
    float s = ratio * ((float)dst + off) - off;
    if (s < 0.0f) s = 0.0f;
    int a = (int)s;
    if (a > in - 1) a = in - 1;
    *i0 = a;
    *i1 = (a < in - 1) ? a + 1 : a;
    *l1 = s - (float)a;
    *l0 = 1.0f - *l1;
}

static inline float lwso_bilerp(const float *p, int win, int y0, int y1, int x0, int x1,
                                float hy0, float hy1, float wx0, float wx1)
{
    float top = p[(int64_t)y0 * win + x0] * wx0 + p[(int64_t)y0 * win + x1] * wx1;
    float bot = p[(int64_t)y1 * win + x0] * wx0 + p[(int64_t)y1 * win + x1] * wx1;
    return hy0 * top + hy1 * bot;
}

/* out = bilinear_resize(in) * mul_a * mul_b  (each product rounded to float32).
 * Stage prologue: mul_a = float(h), mul_b = float32(1/H)  (models.py:119-121). */
LWSO_API void lwso_resize_bilinear(const float *in, float *out, int N, int hin, int win,
                                   int hout, int wout, float mul_a, float mul_b)
{
    const float rh = (float)hin / (float)hout, rw = (float)win / (float)wout;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int y = 0; y < hout; ++y) {
            int y0, y1;
            float hy0, hy1;
            lwso_src(y, rh, hin, &y0, &y1, &hy0, &hy1);
            const float *p = in + (int64_t)n * hin * win;
            for (int x = 0; x < wout; ++x) {
                int x0, x1;
                float wx0, wx1;
                lwso_src(x, rw, win, &x0, &x1, &wx0, &wx1);
                float v = lwso_bilerp(p, win, y0, y1, x0, x1, hy0, hy1, wx0, wx1);
                v = v * mul_a;
                v = v * mul_b;
                out[((int64_t)n * hout + y) * wout + x] = v;
            }
        }
}

/* Residual volume: k = 0..2m-2, s_k = k-(m-1); R sampled by the reference's
 * normalise -> grid_sample round trip, in float32, op by op. */
LWSO_API void lwso_volume_l1_warp(const float *L, const float *R, const float *wflow, float *cost,
                                  int B, int C, int h, int w, int m)
{
    const int K = 2 * m - 1;
    const int64_t plane = (int64_t)h * w;
    const float rw = 1.0f / (float)(w - 1 > 1 ? w - 1 : 1);
    const float rh = 1.0f / (float)(h - 1 > 1 ? h - 1 : 1);
    const float fw1 = (float)(w - 1), fh1 = (float)(h - 1);
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int k = 0; k < K; ++k) {
            const float sk = (float)(k - (m - 1));
            for (int y = 0; y < h; ++y)
                for (int x = 0; x < w; ++x) {
                    float delta = wflow[(int64_t)b * plane + (int64_t)y * w + x] - sk;   /* :93  */
                    float vx = (float)x - delta;                                          /* :45  */
                    float gx = (2.0f * vx) * rw - 1.0f;                                   /* :47  */
                    float gy = (2.0f * (float)y) * rh - 1.0f;                             /* :48  */
                    /* grid_sample un-normalise.  The CUDA kernels compute ((g+1)/2)*(size-1) and the
                     * CPU kernels (g+1)*((size-1)/2); halving is exact in binary floating point, so
                     * both round to the same float32 and one formula covers both. */
                    float ix = ((gx + 1.0f) / 2.0f) * fw1;
                    float iy = ((gy + 1.0f) / 2.0f) * fh1;
                    float fx0 = floorf(ix), fy0 = floorf(iy);
                    float fx1 = fx0 + 1.0f, fy1 = fy0 + 1.0f;
                    float w_nw = (fx1 - ix) * (fy1 - iy);
                    float w_ne = (ix - fx0) * (fy1 - iy);
                    float w_sw = (fx1 - ix) * (iy - fy0);
                    float w_se = (ix - fx0) * (iy - fy0);
                    /* clamp before the int cast: far out-of-range samples are all-zero anyway */
                    float cx = fx0 < -2.0f ? -2.0f : (fx0 > (float)w ? (float)w : fx0);
                    float cy = fy0 < -2.0f ? -2.0f : (fy0 > (float)h ? (float)h : fy0);
                    int x0 = (int)cx, y0 = (int)cy, x1 = x0 + 1, y1 = y0 + 1;
                    int vx0 = (x0 >= 0 && x0 < w), vx1 = (x1 >= 0 && x1 < w);
                    int vy0 = (y0 >= 0 && y0 < h), vy1 = (y1 >= 0 && y1 < h);
                    float acc = 0.0f;
                    for (int c = 0; c < C; ++c) {
                        const float *rp = R + ((int64_t)b * C + c) * plane;
                        float s = 0.0f;
                        if (vy0 && vx0) s = s + rp[(int64_t)y0 * w + x0] * w_nw;
                        if (vy0 && vx1) s = s + rp[(int64_t)y0 * w + x1] * w_ne;
                        if (vy1 && vx0) s = s + rp[(int64_t)y1 * w + x0] * w_sw;
                        if (vy1 && vx1) s = s + rp[(int64_t)y1 * w + x1] * w_se;
                        float l = L[((int64_t)b * C + c) * plane + (int64_t)y * w + x];
                        acc = acc + fabsf(l - s);                                         /* :101 */
                    }
                    cost[((int64_t)b * K + k) * plane + (int64_t)y * w + x] = acc;
                }
        }
}

/* One [BatchNorm3D -> ReLU -> Conv3D 3x3x3 s1 p1] layer, NCDHW.
 *   a = max(fmaf(x, bn_s[ci], bn_t[ci]), 0)        (eval BN as scale/shift, see weights.bn_scale_shift)
 *   y[co] = fmaf-chain over taps (kd,kh,kw) outer, ci inner, ascending, from 0;
 *   zero padding is applied to `a` (out-of-range taps are skipped: fmaf(0,w,acc)==acc).
 * If skip != NULL (single-channel), y += skip  (models.py:137). */
LWSO_API void lwso_bnrelu_conv3d(const float *x, const float *wgt, const float *bn_s, const float *bn_t,
                                 const float *skip, float *y, int B, int Cin, int Cout, int D, int h, int w)
{
    const int64_t plane = (int64_t)h * w, vol = (int64_t)D * plane;
#pragma omp parallel for collapse(3) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Cout; ++co)
            for (int d = 0; d < D; ++d)
                for (int yy = 0; yy < h; ++yy)
                    for (int xx = 0; xx < w; ++xx) {
                        float acc = 0.0f;
                        for (int kd = 0; kd < 3; ++kd) {
                            int zd = d + kd - 1;
                            if (zd < 0 || zd >= D) continue;
                            for (int kh = 0; kh < 3; ++kh) {
                                int zy = yy + kh - 1;
                                if (zy < 0 || zy >= h) continue;
                                for (int kw = 0; kw < 3; ++kw) {
                                    int zx = xx + kw - 1;
                                    if (zx < 0 || zx >= w) continue;
                                    int64_t off = (int64_t)zd * plane + (int64_t)zy * w + zx;
                                    int tap = (kd * 3 + kh) * 3 + kw;
                                    for (int ci = 0; ci < Cin; ++ci) {
                                        float v = x[((int64_t)b * Cin + ci) * vol + off];
                                        float a = fmaxf(fmaf(v, bn_s[ci], bn_t[ci]), 0.0f);
                                        acc = fmaf(a, wgt[((int64_t)co * Cin + ci) * 27 + tap], acc);
                                    }
                                }
                            }
                        }
                        int64_t o = ((int64_t)b * Cout + co) * vol + (int64_t)d * plane + (int64_t)yy * w + xx;
                        if (skip) acc = acc + skip[(int64_t)b * vol + (int64_t)d * plane + (int64_t)yy * w + xx];
                        y[o] = acc;
                    }
}

/* d[b,y,x] = sum_k softmax_k(-cost[b,:,y,x]) * (start + k): max-subtracted softmax,
 * e_k = expf(m - cost_k) with m = max_k(-cost_k); S = sum e_k ascending;
 * p_k = e_k / S; acc += p_k * v_k ascending. */
LWSO_API void lwso_softargmin(const float *cost, float *out, int B, int D, int h, int w, float start)
{
    const int64_t plane = (int64_t)h * w;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)B * plane; ++i) {
        int64_t b = i / plane, p = i % plane;
        const float *c = cost + b * D * plane + p;
        float m = -c[0];
        for (int k = 1; k < D; ++k) m = fmaxf(m, -c[(int64_t)k * plane]);
        float e[64];
        float S = 0.0f;
        for (int k = 0; k < D; ++k) {
            e[k] = lwso_expf(-c[(int64_t)k * plane] - m);
            S = S + e[k];
        }
        float acc = 0.0f;
        for (int k = 0; k < D; ++k) {
            float pk = e[k] / S;
            acc = acc + pk * (start + (float)k);
        }
        out[i] = acc;
    }
}

/* out = bilinear_resize(low * mul_a * mul_b -> [H,W]) (+ prev).
 * mul_a = float(H), mul_b = float32(1/h)  (models.py:145-146,153-156). */
LWSO_API void lwso_upsample_add(const float *low, const float *prev, float *out,
                                int B, int h, int w, int H, int W, float mul_a, float mul_b)
{
    const float rh = (float)h / (float)H, rw = (float)w / (float)W;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y) {
            int y0, y1;
            float hy0, hy1;
            lwso_src(y, rh, h, &y0, &y1, &hy0, &hy1);
            const float *p = low + (int64_t)b * h * w;
            for (int x = 0; x < W; ++x) {
                int x0, x1;
                float wx0, wx1;
                lwso_src(x, rw, w, &x0, &x1, &wx0, &wx1);
                float p00 = (p[(int64_t)y0 * w + x0] * mul_a) * mul_b;
                float p01 = (p[(int64_t)y0 * w + x1] * mul_a) * mul_b;
                float p10 = (p[(int64_t)y1 * w + x0] * mul_a) * mul_b;
                float p11 = (p[(int64_t)y1 * w + x1] * mul_a) * mul_b;
                float top = p00 * wx0 + p01 * wx1;
                float bot = p10 * wx0 + p11 * wx1;
                float v = hy0 * top + hy1 * bot;
                int64_t o = ((int64_t)b * H + y) * W + x;
                if (prev) v = v + prev[o];
                out[o] = v;
            }
        }
}

/* =====================================================================================
 * 2D networks: feature extractor
 * (submodules.py:5-33,35-109,113-188) and refinement (submodules.py:223-327).
 * Arithmetic contract for every 2D convolution: ONE fmaf chain from 0 per output,
 * taps (kh,kw) outer ascending, input channel inner ascending; zero padding (skipped taps).
 * ===================================================================================== */

/* Conv2D, NCHW, square kernel k (1 or 3), groups in {1, Cin (depthwise, Cout == Cin)}.
 * If pre_s != NULL the input is first mapped through a = max(fmaf(x, pre_s[ci], pre_t[ci]), 0)
 * (BatchNorm2D -> ReLU in front of the convolution, submodules.py:223-280); padding stays zero. */
LWSO_API void lwso_conv2d(const float *x, const float *wgt, const float *pre_s, const float *pre_t, float *y,
                          int B, int Cin, int Cout, int H, int W, int Ho, int Wo, int k, int stride, int pad,
                          int dil, int depthwise)
{
    const int64_t plane = (int64_t)H * W, oplane = (int64_t)Ho * Wo;
    const int cin_g = depthwise ? 1 : Cin;
#pragma omp parallel for collapse(3) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Cout; ++co)
            for (int oy = 0; oy < Ho; ++oy)
                for (int ox = 0; ox < Wo; ++ox) {
                    float acc = 0.0f;
                    for (int kh = 0; kh < k; ++kh) {
                        int iy = oy * stride - pad + kh * dil;
                        if (iy < 0 || iy >= H) continue;
                        for (int kw = 0; kw < k; ++kw) {
                            int ix = ox * stride - pad + kw * dil;
                            if (ix < 0 || ix >= W) continue;
                            for (int cg = 0; cg < cin_g; ++cg) {
                                int ci = depthwise ? co : cg;
                                float v = x[((int64_t)b * Cin + ci) * plane + (int64_t)iy * W + ix];
                                if (pre_s) v = fmaxf(fmaf(v, pre_s[ci], pre_t[ci]), 0.0f);
                                acc = fmaf(v, wgt[(((int64_t)co * cin_g + cg) * k + kh) * k + kw], acc);
                            }
                        }
                    }
                    y[((int64_t)b * Cout + co) * oplane + (int64_t)oy * Wo + ox] = acc;
                }
}

/* Conv2DTranspose k=3, stride=2, padding=1, output_padding=1 (submodules.py:25-32): Ho = 2H, Wo = 2W.
 * Weight layout [Cin][Cout][3][3].  out[oy][ox] gathers in[iy][ix] with oy = 2*iy - 1 + kh. */
LWSO_API void lwso_deconv2d_s2(const float *x, const float *wgt, float *y, int B, int Cin, int Cout, int H, int W)
{
    const int Ho = 2 * H, Wo = 2 * W;
    const int64_t plane = (int64_t)H * W, oplane = (int64_t)Ho * Wo;
#pragma omp parallel for collapse(3) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Cout; ++co)
            for (int oy = 0; oy < Ho; ++oy)
                for (int ox = 0; ox < Wo; ++ox) {
                    float acc = 0.0f;
                    for (int kh = 0; kh < 3; ++kh) {
                        int ty = oy + 1 - kh;
                        if (ty < 0 || (ty & 1)) continue;
                        int iy = ty >> 1;
                        if (iy >= H) continue;
                        for (int kw = 0; kw < 3; ++kw) {
                            int tx = ox + 1 - kw;
                            if (tx < 0 || (tx & 1)) continue;
                            int ix = tx >> 1;
                            if (ix >= W) continue;
                            for (int ci = 0; ci < Cin; ++ci) {
                                float v = x[((int64_t)b * Cin + ci) * plane + (int64_t)iy * W + ix];
                                acc = fmaf(v, wgt[(((int64_t)ci * Cout + co) * 3 + kh) * 3 + kw], acc);
                            }
                        }
                    }
                    y[((int64_t)b * Cout + co) * oplane + (int64_t)oy * Wo + ox] = acc;
                }
}

/* y = fmaf(x, s[c], t[c]) (if s != NULL); y += add (if add != NULL); y = max(y, 0) (if relu).  NCHW, in place ok. */
LWSO_API void lwso_bn_add_relu(const float *x, const float *s, const float *t, const float *add, float *y,
                               int B, int C, int64_t plane, int relu)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) {
            const int64_t o = ((int64_t)b * C + c) * plane;
            for (int64_t i = 0; i < plane; ++i) {
                float v = x[o + i];
                if (s) v = fmaf(v, s[c], t[c]);
                if (add) v = v + add[o + i];
                if (relu) v = fmaxf(v, 0.0f);
                y[o + i] = v;
            }
        }
}
