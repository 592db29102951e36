// Cost-volume kernels (HBM/L2-bound; float32; compiled with -ffp-contract=off so that every
// operation below is the IEEE operation written -- the C oracle performs the same sequence).
//
//   k_volume_l1_shift  <- LWSNet._build_volume_2d   /root/reference/models/models.py:58-76
//   k_volume_l1_warp   <- forward() glue :119-121 + _build_volume_2d3 :78-104 + warp :28-55
#include <hip/hip_fp16.h>

#include "lws_common.h"
#include "lws_device_math.h"

namespace lws {

LWS_DEFINE_STAMPS(volume)

// BASELINE config 5: feature values pass through fp16 (round-to-nearest-even) where the volume kernels read them.
template <bool QH>
__device__ __forceinline__ float qf(float x)
{
    return QH ? __half2float(__float2half_rn(x)) : x;
}

// ---------------------------------------------------------------------------------------------
// Stage-1 volume.  One workgroup = one image row segment of 64 pixels; the right-feature row
// segment [x0-(D-1), x0+63] of all C channels is staged once in LDS (zero for x < 0: the
// reference treats the occluded columns as |L - 0|, models.py:71), the left features of a
// pixel live in registers, and the 4 waves split the D hypotheses.  Global reads and writes
// are coalesced along W.
// ---------------------------------------------------------------------------------------------
template <int C, bool QH>
__global__ __launch_bounds__(256) void k_volume_l1_shift(const float *__restrict__ L,
                                                         const float *__restrict__ R,
                                                         float *__restrict__ cost, int h, int w, int D)
{
    extern __shared__ float sR[];   // [C][64 + D - 1]
    const int x0 = blockIdx.x * 64, y = blockIdx.y, b = blockIdx.z;
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int span = 64 + D - 1;
    const int64_t plane = (int64_t)h * w;
    const float *Rb = R + (int64_t)b * C * plane + (int64_t)y * w;
    // wave ty stages channels ty, ty+4, ...; positions p = tx and tx + 64 (span <= 127): all C/2 loads of a thread are
    // unconditional (clamped) and in flight together
    {
        const int xa = x0 - (D - 1) + tx, xb = xa + 64;
        const bool oka = xa >= 0 && xa < w, okb = tx + 64 < span && xb >= 0 && xb < w;
        float va[C / 4], vb[C / 4];
#pragma unroll
        for (int j = 0; j < C / 4; ++j) {
            const float *rc = Rb + (int64_t)(ty + 4 * j) * plane;
            va[j] = rc[oka ? xa : 0];
            vb[j] = rc[okb ? xb : 0];
        }
#pragma unroll
        for (int j = 0; j < C / 4; ++j) {
            sR[(ty + 4 * j) * span + tx] = oka ? qf<QH>(va[j]) : 0.0f;
            if (tx + 64 < span) sR[(ty + 4 * j) * span + tx + 64] = okb ? qf<QH>(vb[j]) : 0.0f;
        }
    }
    __syncthreads();
    const int x = x0 + tx;
    if (x >= w) return;
    float l[C];
    const float *Lb = L + (int64_t)b * C * plane + (int64_t)y * w + x;
#pragma unroll
    for (int c = 0; c < C; ++c) l[c] = qf<QH>(Lb[(int64_t)c * plane]);
    float *out = cost + (int64_t)b * D * plane + (int64_t)y * w + x;
    for (int d = ty; d < D; d += 4) {
        const float *r = sR + tx + (D - 1) - d;
        float acc = 0.0f;
#pragma unroll
        for (int c = 0; c < C; ++c) acc = acc + fabsf(l[c] - r[c * span]);
        out[(int64_t)d * plane] = acc;
    }
}

int launch_volume_l1_shift(const float *L, const float *R, float *cost, int B, int C, int h, int w, int D,
                           hipStream_t st, bool q16)
{
    dim3 grid(cdiv(w, 64), h, B), block(64, 4);
    size_t lds = (size_t)C * (64 + D - 1) * sizeof(float);
#define LWS_VS(CC)                                                                                            \
    if (q16) hipLaunchKernelGGL((k_volume_l1_shift<CC, true>), grid, block, lds, st, L, R, cost, h, w, D);     \
    else hipLaunchKernelGGL((k_volume_l1_shift<CC, false>), grid, block, lds, st, L, R, cost, h, w, D)
    switch (C) {
        case 8: LWS_VS(8); break;
        case 16: LWS_VS(16); break;
        case 32: LWS_VS(32); break;
        default: set_error("volume_l1_shift: unsupported channel count %d (8, 16, 32)", C); return LWS_ERR_INVALID;
    }
#undef LWS_VS
    LWS_LAUNCH_CHECK();
    return LWS_OK;
}

// ---------------------------------------------------------------------------------------------
// Stage-2/3 residual volume (round 4: the design north_star describes).  One workgroup = one row
// segment of 64 pixels of one image, three waves.
//   phase A  every wave loads the left features of its pixel into registers (C coalesced loads);
//            wave 0 also evaluates the flow of the 64 pixels -- the previous disparity resized on
//            the fly, models.py:119-121 -- ONCE per pixel, and from it the tile's column window
//            [min x0, max x0 + 1] (x0 is monotone in the hypothesis k, so the bounds come from
//            k = 0 and k = 2m-2) with a wave reduction;
//   phase B  the right-feature window of all C channels, rows fy0 (and fy0 + 1 when iy is not an
//            integer: a row property), is staged in LDS channel-innermost with ZEROS outside the
//            image -- grid_sample's zero padding -- so the taps need no masks (s + 0*w == s up to
//            the sign of zero, which |l - s| discards);
//   phase C  wave j computes hypotheses k = j, j+3, ...: two (four) 16-byte LDS reads per four
//            channels, the lerp and |l - s| on packed float32 pairs (v_pk_mul_f32 / v_pk_add_f32;
//            -ffp-contract=off: no fused multiply-add), the channel sum sequential and ascending.
// The 9x expansion of L, R and disp the reference materialises (models.py:85-99) is index
// arithmetic.  The float32 coordinate round trip (normalise :45-48, grid_sample un-normalise) is
// replayed op for op.  A tile whose flow range needs more than WARP_NCMAX columns falls back
// (workgroup-uniform) to gathering the taps from global memory with the same arithmetic.
// ---------------------------------------------------------------------------------------------
constexpr int WARP_TX = 64;        // pixels per workgroup
constexpr int WARP_NW = 3;         // waves per workgroup (hypotheses k = wave, wave + 3, ...)
constexpr int WARP_NCMAX = 160;    // LDS window columns (64 + flow range of the tile + 2m + 1)

struct WarpRow {      // row quantities of grid_sample (depend on y only)
    float wy0, wy1;   // fy1 - iy, iy - fy0
    int y0;           // clamped floor(iy)
    bool south;       // iy is not an integer: rows y0 and y0 + 1 both contribute
};

__device__ __forceinline__ WarpRow warp_row(int y, int h)
{
    const float rh = 1.0f / (float)(h - 1 > 1 ? h - 1 : 1);
    const float fh1 = (float)(h - 1);
    const float gy = (2.0f * (float)y) * rh - 1.0f;           // models.py:48
    const float iy = ((gy + 1.0f) / 2.0f) * fh1;              // grid_sample un-normalise (align_corners=True)
    const float fy0 = floorf(iy), fy1 = fy0 + 1.0f;
    const float cy = fy0 < -2.0f ? -2.0f : (fy0 > (float)h ? (float)h : fy0);
    WarpRow r;
    r.wy0 = fy1 - iy;
    r.wy1 = iy - fy0;
    r.y0 = (int)cy;
    r.south = (iy - fy0) != 0.0f;                             // else w_sw == w_se == 0 exactly
    return r;
}

// column quantities of one (pixel, hypothesis): x0 = clamped floor(ix), wx0 = fx1 - ix, wx1 = ix - fx0
__device__ __forceinline__ void warp_col(float wf, int k, int m, int x, int w, float rw, float fw1, int &x0, float &wx0,
                                         float &wx1)
{
    const float sk = (float)(k - (m - 1));
    const float delta = wf - sk;                              // models.py:93
    const float vx = (float)x - delta;                        // :45
    const float gx = (2.0f * vx) * rw - 1.0f;                 // :47
    const float ix = ((gx + 1.0f) / 2.0f) * fw1;              // grid_sample un-normalise
    const float fx0 = floorf(ix), fx1 = fx0 + 1.0f;
    wx0 = fx1 - ix;
    wx1 = ix - fx0;
    const float cx = fx0 < -2.0f ? -2.0f : (fx0 > (float)w ? (float)w : fx0);
    x0 = (int)cx;
}

template <int C, bool QH>
__global__ __launch_bounds__(WARP_TX *WARP_NW) void k_volume_l1_warp(const float *__restrict__ L,
                                                                      const float *__restrict__ R,
                                                                      const float *__restrict__ prev,
                                                                      float *__restrict__ cost,
                                                                      float *__restrict__ wflow_out, int h, int w,
                                                                      int H, int W, int m, float mul_a, float mul_b,
                                                                      const float *__restrict__ plow, int ph, int pw,
                                                                      float *__restrict__ pmat, int force_gather,
                                                                      const float *__restrict__ plow0, int ph0, int pw0,
                                                                      float *__restrict__ pmat0, float ioff)
{
    // plow != nullptr: the previous stage's full-resolution map has not been materialised (no k_upsample_add launch):
    // it is evaluated on demand as upsample(plow [ph,pw]) + prev (prev = the map of the stage before it), see
    // DeferredMap.  With H == 2h, W == 2w the four taps of pixel (y,x) are exactly the 2 x 2 block (2y..2y+1,
    // 2x..2x+1) of that map, so wave 0 also writes it out (pmat) -- the map is an output of the path.
    // prev == nullptr (round 5): the map before it is not in memory either.  plow0 != nullptr: it is stage 1's map,
    // upsample(plow0 [ph0,pw0]) with nothing added; this kernel evaluates it at the same four pixels and writes it to pmat0
    // too.  plow0 == nullptr: there is no map before it (stage 2 reading stage 1's deferred map; nothing is written).
    constexpr int CP = C + 4;                                  // LDS floats per column: conflict-free 16-byte reads
    __shared__ float4 sR4[2 * WARP_NCMAX * CP / 4];            // [row][column][CP]
    __shared__ float sWf[WARP_TX];
    __shared__ int sMeta[2];                                   // xlo, columns
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = blockIdx.x * WARP_TX + lane, y = blockIdx.y, b = blockIdx.z;
    const int nk = 2 * m - 1;
    const int64_t plane = (int64_t)h * w;
    const bool live = x < w;
    LWS_STAMPK(10, 0);

    // phase A: left features of this thread's pixel (independent of the flow: in flight beside it)
    f2_t l2[C / 2];
    {
        const float *Lp = L + (int64_t)b * C * plane + (int64_t)y * w + (live ? x : 0);
#pragma unroll
        for (int c = 0; c < C; c += 2) {
            l2[c / 2].x = qf<QH>(Lp[(int64_t)c * plane]);
            l2[c / 2].y = qf<QH>(Lp[(int64_t)(c + 1) * plane]);
        }
    }
    const float rw = 1.0f / (float)(w - 1 > 1 ? w - 1 : 1);
    const float fw1 = (float)(w - 1);
    if (wave == 0) {
        // wflow = resize(prev)[y,x] * float(h) * float32(1/H)            (models.py:119-121)
        float wf = 0.0f;
        int lo = 0x7fffffff, hi = -0x7fffffff;
        if (live) {
            const float rh_ = (float)H / (float)h, rw_ = (float)W / (float)w;
            int y0, y1, x0, x1;
            float hy0, hy1, wx0, wx1;
            src_index(y, rh_, H, y0, y1, hy0, hy1, ioff);
            src_index(x, rw_, W, x0, x1, wx0, wx1, ioff);
            DeferredMap dm{plow != nullptr ? plow + (int64_t)b * ph * pw : nullptr,
                           prev != nullptr ? prev + (int64_t)b * H * W : nullptr, ph, pw, (float)H, 1.0f / (float)(ph > 0 ? ph : 1)};
            dm.off = ioff;
            if (plow0 != nullptr) {
                dm.low0 = plow0 + (int64_t)b * ph0 * pw0;
                dm.h0 = ph0;
                dm.w0 = pw0;
                dm.mul_b0 = 1.0f / (float)(ph0 > 0 ? ph0 : 1);
            }
            const int tys[4] = {y0, y0, y1, y1}, txs[4] = {x0, x1, x0, x1};
            float q[4], q0[4];
            deferred_at_n<4>(dm, tys, txs, H, W, q, q0);
            if (pmat != nullptr) {
                float *pm = pmat + (int64_t)b * H * W;
                pm[(int64_t)y0 * W + x0] = q[0];
                pm[(int64_t)y0 * W + x1] = q[1];
                pm[(int64_t)y1 * W + x0] = q[2];
                pm[(int64_t)y1 * W + x1] = q[3];
            }
            if (pmat0 != nullptr) {
                float *pm = pmat0 + (int64_t)b * H * W;
                pm[(int64_t)y0 * W + x0] = q0[0];
                pm[(int64_t)y0 * W + x1] = q0[1];
                pm[(int64_t)y1 * W + x0] = q0[2];
                pm[(int64_t)y1 * W + x1] = q0[3];
            }
            const float top = q[0] * wx0 + q[1] * wx1;
            const float bot = q[2] * wx0 + q[3] * wx1;
            wf = hy0 * top + hy1 * bot;
            wf = wf * mul_a;
            wf = wf * mul_b;
            if (wflow_out != nullptr) wflow_out[(int64_t)b * plane + (int64_t)y * w + x] = wf;
            float u0, u1;
            warp_col(wf, 0, m, x, w, rw, fw1, lo, u0, u1);
            warp_col(wf, nk - 1, m, x, w, rw, fw1, hi, u0, u1);
            hi += 1;
        }
        sWf[lane] = wf;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo = min(lo, __shfl_xor(lo, o));
            hi = max(hi, __shfl_xor(hi, o));
        }
        if (lane == 0) {
            sMeta[0] = lo;
            sMeta[1] = hi - lo + 1;
        }
    }
    __syncthreads();
    LWS_STAMPK(10, 1);
    const float wf = sWf[lane];
    const int xlo = sMeta[0], ncols = sMeta[1];
    const WarpRow row = warp_row(y, h);
    float *out = cost + (int64_t)b * nk * plane + (int64_t)y * w + x;
    const float *Rb = R + (int64_t)b * C * plane;

    if (force_gather || ncols > WARP_NCMAX) {
        // fallback: the taps gathered from global memory (L2), loaded UNCONDITIONALLY from clamped addresses so that all
        // of a channel's gathers are in flight together; invalid taps are skipped as the reference's zero padding does
        if (!live) return;
        const int y0 = row.y0, y1 = y0 + 1;
        const bool vy0 = (y0 >= 0 && y0 < h), vy1 = (y1 >= 0 && y1 < h);
        for (int k = wave; k < nk; k += WARP_NW) {
            int x0;
            float wx0, wx1;
            warp_col(wf, k, m, x, w, rw, fw1, x0, wx0, wx1);
            const int x1 = x0 + 1;
            const bool vx0 = (x0 >= 0 && x0 < w), vx1 = (x1 >= 0 && x1 < w);
            const float w_nw = wx0 * row.wy0, w_ne = wx1 * row.wy0, w_sw = wx0 * row.wy1, w_se = wx1 * row.wy1;
            const bool b_nw = vy0 && vx0, b_ne = vy0 && vx1, b_sw = row.south && vy1 && vx0, b_se = row.south && vy1 && vx1;
            const int o_nw = b_nw ? y0 * w + x0 : 0, o_ne = b_ne ? y0 * w + x1 : 0;
            const int o_sw = b_sw ? y1 * w + x0 : 0, o_se = b_se ? y1 * w + x1 : 0;
            float acc = 0.0f;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float *rp = Rb + (int64_t)c * plane;
                const float r_nw = qf<QH>(rp[o_nw]), r_ne = qf<QH>(rp[o_ne]);
                float r_sw = 0.0f, r_se = 0.0f;
                if (row.south) {
                    r_sw = qf<QH>(rp[o_sw]);
                    r_se = qf<QH>(rp[o_se]);
                }
                const float l = (c & 1) ? l2[c / 2].y : l2[c / 2].x;
                float s = 0.0f;
                if (b_nw) s = s + r_nw * w_nw;
                if (b_ne) s = s + r_ne * w_ne;
                if (b_sw) s = s + r_sw * w_sw;
                if (b_se) s = s + r_se * w_se;
                acc = acc + fabsf(l - s);                    // :101
            }
            out[(int64_t)k * plane] = acc;
        }
        return;
    }

    // phase B: stage the window.  item = (column, 4-channel group): four coalesced plane loads, one 16-byte LDS write
    {
        const int nrows = row.south ? 2 : 1;
        const int per_row = ncols * (C / 4);
        for (int it = threadIdx.x; it < nrows * per_row; it += WARP_TX * WARP_NW) {
            const int r = it >= per_row ? 1 : 0;
            const int i = it - r * per_row;
            const int cg = i / ncols, col = i - cg * ncols;
            const int xg = xlo + col, yr = row.y0 + r;
            const bool ok = xg >= 0 && xg < w && yr >= 0 && yr < h;
            const float *rp = Rb + (int64_t)(4 * cg) * plane + (ok ? yr * w + xg : 0);
            float4 v;
            v.x = rp[0];
            v.y = rp[plane];
            v.z = rp[2 * plane];
            v.w = rp[3 * plane];
            if (ok) {
                v.x = qf<QH>(v.x);
                v.y = qf<QH>(v.y);
                v.z = qf<QH>(v.z);
                v.w = qf<QH>(v.w);
            } else {
                v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
            sR4[((r * WARP_NCMAX + col) * CP + 4 * cg) / 4] = v;
        }
    }
    __syncthreads();
    LWS_STAMPK(10, 2);
    if (!live) return;

    // phase C
    for (int k = wave; k < nk; k += WARP_NW) {
        int x0;
        float wx0, wx1;
        warp_col(wf, k, m, x, w, rw, fw1, x0, wx0, wx1);
        const float w_nw = wx0 * row.wy0, w_ne = wx1 * row.wy0;
        const float4 *p0 = sR4 + ((x0 - xlo) * CP) / 4;        // column x0, row y0; column x0 + 1 follows CP floats later
        const f2_t Wnw = {w_nw, w_nw}, Wne = {w_ne, w_ne};
        float acc = 0.0f;
        if (!row.south) {
#pragma unroll
            for (int cg = 0; cg < C / 4; ++cg) {
                const float4 a = p0[cg], e = p0[CP / 4 + cg];
                const f2_t a0 = {a.x, a.y}, a1 = {a.z, a.w}, e0 = {e.x, e.y}, e1 = {e.z, e.w};
                const f2_t s0 = a0 * Wnw + e0 * Wne, s1 = a1 * Wnw + e1 * Wne;
                const f2_t d0 = l2[2 * cg] - s0, d1 = l2[2 * cg + 1] - s1;
                acc = acc + fabsf(d0.x);                     // :101, channels ascending
                acc = acc + fabsf(d0.y);
                acc = acc + fabsf(d1.x);
                acc = acc + fabsf(d1.y);
            }
        } else {
            const float w_sw = wx0 * row.wy1, w_se = wx1 * row.wy1;
            const f2_t Wsw = {w_sw, w_sw}, Wse = {w_se, w_se};
            const float4 *p1 = p0 + (WARP_NCMAX * CP) / 4;
#pragma unroll
            for (int cg = 0; cg < C / 4; ++cg) {
                const float4 a = p0[cg], e = p0[CP / 4 + cg], f = p1[cg], g = p1[CP / 4 + cg];
                const f2_t a0 = {a.x, a.y}, a1 = {a.z, a.w}, e0 = {e.x, e.y}, e1 = {e.z, e.w};
                const f2_t f0 = {f.x, f.y}, f1 = {f.z, f.w}, g0 = {g.x, g.y}, g1 = {g.z, g.w};
                const f2_t s0 = ((a0 * Wnw + e0 * Wne) + f0 * Wsw) + g0 * Wse;
                const f2_t s1 = ((a1 * Wnw + e1 * Wne) + f1 * Wsw) + g1 * Wse;
                const f2_t d0 = l2[2 * cg] - s0, d1 = l2[2 * cg + 1] - s1;
                acc = acc + fabsf(d0.x);
                acc = acc + fabsf(d0.y);
                acc = acc + fabsf(d1.x);
                acc = acc + fabsf(d1.y);
            }
        }
        out[(int64_t)k * plane] = acc;
    }
    LWS_STAMPK(10, 3);
}

int launch_volume_l1_warp(const float *L, const float *R, const float *prev, float *cost, float *wflow_out,
                          int B, int C, int h, int w, int H, int W, int m, hipStream_t st, bool q16, const float *plow,
                          int ph, int pw, float *pmat, int form, const float *plow0, int ph0, int pw0, float *pmat0, float ioff)
{
    if ((pmat != nullptr || pmat0 != nullptr) && (plow == nullptr || H != 2 * h || W != 2 * w)) {
        set_error("volume_l1_warp: the deferred map can only be written out at exactly half resolution");
        return LWS_ERR_INVALID;
    }
    if (prev == nullptr && plow == nullptr) {
        set_error("volume_l1_warp: no previous map (neither materialised nor deferred)");
        return LWS_ERR_INVALID;
    }
    if ((plow0 != nullptr && prev != nullptr) || (pmat0 != nullptr && plow0 == nullptr)) {
        set_error("volume_l1_warp: the second deferred level stands in for a missing `prev`");
        return LWS_ERR_INVALID;
    }
    dim3 grid(cdiv(w, WARP_TX), h, B), block(WARP_TX * WARP_NW);
    const float mul_a = (float)h, mul_b = 1.0f / (float)H;
    const int force_gather = form == 0 ? 1 : 0;
#define LWS_VW(CC)                                                                                                   \
    if (q16)                                                                                                          \
        hipLaunchKernelGGL((k_volume_l1_warp<CC, true>), grid, block, 0, st, L, R, prev, cost, wflow_out, h, w, H, W, m,  \
                           mul_a, mul_b, plow, ph, pw, pmat, force_gather, plow0, ph0, pw0, pmat0, ioff);                 \
    else                                                                                                              \
        hipLaunchKernelGGL((k_volume_l1_warp<CC, false>), grid, block, 0, st, L, R, prev, cost, wflow_out, h, w, H, W, m, \
                           mul_a, mul_b, plow, ph, pw, pmat, force_gather, plow0, ph0, pw0, pmat0, ioff)
    switch (C) {
        case 8: LWS_VW(8); break;
        case 16: LWS_VW(16); break;
        default: set_error("volume_l1_warp: unsupported channel count %d (8, 16)", C); return LWS_ERR_INVALID;
    }
#undef LWS_VW
    LWS_LAUNCH_CHECK();
    return LWS_OK;
}

}  // namespace lws
