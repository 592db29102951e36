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
// Stage-2/3 residual volume.  One thread = one (pixel, hypothesis k); grid.y = k so that a
// wave reads 64 consecutive pixels of one row.  The 9x expansion of L, R and disp the
// reference materialises (models.py:85-99) is pure index arithmetic here.  The previous
// disparity is resized on the fly (4 taps); the right features are gathered straight from
// L2/L1: neighbouring lanes sample neighbouring columns because the flow is smooth.
// The south taps are only touched when iy is not an exact integer (a wave-uniform branch,
// exact because x + 0*v == x for finite v).
// ---------------------------------------------------------------------------------------------
template <int C, bool QH>
__global__ __launch_bounds__(256) void k_volume_l1_warp(const float *__restrict__ L,
                                                        const float *__restrict__ R,
                                                        const float *__restrict__ prev,
                                                        float *__restrict__ cost,
                                                        float *__restrict__ wflow_out, int h, int w,
                                                        int H, int W, int m, float mul_a, float mul_b,
                                                        const float *__restrict__ plow, int ph, int pw,
                                                        float *__restrict__ pmat)
{
    // plow != nullptr: the previous stage's full-resolution map has not been materialised (no k_upsample_add launch):
    // it is evaluated on demand as upsample(plow [ph,pw]) + prev (prev = the map of the stage before it), see
    // DeferredMap.  With H == 2h, W == 2w the four taps of pixel (y,x) are exactly the 2 x 2 block (2y..2y+1,
    // 2x..2x+1) of that map, so the k == 0 threads also write it out (pmat) -- the map is an output of the path.
    const int pix = blockIdx.x * 256 + threadIdx.x;
    const int k = blockIdx.y, b = blockIdx.z;
    const int64_t plane = (int64_t)h * w;
    if (pix >= plane) return;
    const int y = pix / w, x = pix - y * w;
    LWS_STAMPK(10, 0);

    // wflow = resize(prev)[y,x] * float(h) * float32(1/H)            (models.py:119-121)
    float wf;
    {
        const float rh = (float)H / (float)h, rw = (float)W / (float)w;
        int y0, y1, x0, x1;
        float hy0, hy1, wx0, wx1;
        src_index(y, rh, H, y0, y1, hy0, hy1);
        src_index(x, rw, W, x0, x1, wx0, wx1);
        const DeferredMap dm{plow != nullptr ? plow + (int64_t)b * ph * pw : nullptr, prev + (int64_t)b * H * W, ph, pw,
                             (float)H, 1.0f / (float)(ph > 0 ? ph : 1)};
        const int tys[4] = {y0, y0, y1, y1}, txs[4] = {x0, x1, x0, x1};
        float q[4];
        deferred_at_n<4>(dm, tys, txs, H, W, q);
        const float q00 = q[0], q01 = q[1], q10 = q[2], q11 = q[3];
        if (pmat != nullptr && k == 0) {
            float *pm = pmat + (int64_t)b * H * W;
            pm[(int64_t)y0 * W + x0] = q00;
            pm[(int64_t)y0 * W + x1] = q01;
            pm[(int64_t)y1 * W + x0] = q10;
            pm[(int64_t)y1 * W + x1] = q11;
        }
        float top = q00 * wx0 + q01 * wx1;
        float bot = q10 * wx0 + q11 * wx1;
        wf = hy0 * top + hy1 * bot;
        wf = wf * mul_a;
        wf = wf * mul_b;
    }
    if (wflow_out != nullptr && k == 0) wflow_out[(int64_t)b * plane + pix] = wf;
    LWS_STAMPK(10, 1);

    const float rw = 1.0f / (float)(w - 1 > 1 ? w - 1 : 1);
    const float rh = 1.0f / (float)(h - 1 > 1 ? h - 1 : 1);
    const float fw1 = (float)(w - 1), fh1 = (float)(h - 1);
    const float sk = (float)(k - (m - 1));
    float delta = wf - sk;                              // models.py:93
    float vx = (float)x - delta;                        // :45
    float gx = (2.0f * vx) * rw - 1.0f;                 // :47
    float gy = (2.0f * (float)y) * rh - 1.0f;           // :48
    float ix = ((gx + 1.0f) / 2.0f) * fw1;              // grid_sample un-normalise (align_corners=True)
    float iy = ((gy + 1.0f) / 2.0f) * fh1;
    float fx0 = floorf(ix), fy0 = floorf(iy);
    float fx1 = fx0 + 1.0f, fy1 = fy0 + 1.0f;
    float w_nw = (fx1 - ix) * (fy1 - iy);
    float w_ne = (ix - fx0) * (fy1 - iy);
    float w_sw = (fx1 - ix) * (iy - fy0);
    float w_se = (ix - fx0) * (iy - fy0);
    float cx = fx0 < -2.0f ? -2.0f : (fx0 > (float)w ? (float)w : fx0);
    float cy = fy0 < -2.0f ? -2.0f : (fy0 > (float)h ? (float)h : fy0);
    const int x0 = (int)cx, y0 = (int)cy, x1 = x0 + 1, y1 = y0 + 1;
    const bool vx0 = (x0 >= 0 && x0 < w), vx1 = (x1 >= 0 && x1 < w);
    const bool vy0 = (y0 >= 0 && y0 < h), vy1 = (y1 >= 0 && y1 < h);
    const bool south = (iy - fy0) != 0.0f;              // else w_sw == w_se == 0 exactly

    const float *Lp = L + (int64_t)b * C * plane + pix;
    const float *Rp = R + (int64_t)b * C * plane;
    // Taps are loaded UNCONDITIONALLY from clamped addresses (an invalid tap reads element 0 and is masked below), so
    // that all 2-4 gathers and the L load of every channel are in flight together instead of one branch per tap.
    const bool b_nw = vy0 && vx0, b_ne = vy0 && vx1, b_sw = south && vy1 && vx0, b_se = south && vy1 && vx1;
    const int o_nw = b_nw ? y0 * w + x0 : 0, o_ne = b_ne ? y0 * w + x1 : 0;
    const int o_sw = b_sw ? y1 * w + x0 : 0, o_se = b_se ? y1 * w + x1 : 0;
    float acc = 0.0f;
    if (!south) {                                   // wave-uniform in practice (iy depends on the row only)
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float *rp = Rp + (int64_t)c * plane;
            const float r_nw = qf<QH>(rp[o_nw]), r_ne = qf<QH>(rp[o_ne]);
            const float l = qf<QH>(Lp[(int64_t)c * plane]);
            float s = 0.0f;
            if (b_nw) s = s + r_nw * w_nw;
            if (b_ne) s = s + r_ne * w_ne;
            acc = acc + fabsf(l - s);                    // :101
        }
    } else {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float *rp = Rp + (int64_t)c * plane;
            const float r_nw = qf<QH>(rp[o_nw]), r_ne = qf<QH>(rp[o_ne]), r_sw = qf<QH>(rp[o_sw]), r_se = qf<QH>(rp[o_se]);
            const float l = qf<QH>(Lp[(int64_t)c * plane]);
            float s = 0.0f;
            if (b_nw) s = s + r_nw * w_nw;
            if (b_ne) s = s + r_ne * w_ne;
            if (b_sw) s = s + r_sw * w_sw;
            if (b_se) s = s + r_se * w_se;
            acc = acc + fabsf(l - s);                    // :101
        }
    }
    LWS_STAMPK(10, 2);
    cost[((int64_t)b * (2 * m - 1) + k) * plane + pix] = acc;
    LWS_STAMPK(10, 3);
}

int launch_volume_l1_warp(const float *L, const float *R, const float *prev, float *cost, float *wflow_out,
                          int B, int C, int h, int w, int H, int W, int m, hipStream_t st, bool q16, const float *plow,
                          int ph, int pw, float *pmat)
{
    if (pmat != nullptr && (plow == nullptr || H != 2 * h || W != 2 * w)) {
        set_error("volume_l1_warp: the deferred map can only be written out at exactly half resolution");
        return LWS_ERR_INVALID;
    }
    dim3 grid(cdiv(h * w, 256), 2 * m - 1, B), block(256);
    const float mul_a = (float)h, mul_b = 1.0f / (float)H;
#define LWS_VW(CC)                                                                                                   \
    if (q16)                                                                                                          \
        hipLaunchKernelGGL((k_volume_l1_warp<CC, true>), grid, block, 0, st, L, R, prev, cost, wflow_out, h, w, H, W, m,  \
                           mul_a, mul_b, plow, ph, pw, pmat);                                                                             \
    else                                                                                                              \
        hipLaunchKernelGGL((k_volume_l1_warp<CC, false>), grid, block, 0, st, L, R, prev, cost, wflow_out, h, w, H, W, m, \
                           mul_a, mul_b, plow, ph, pw, pmat)
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
