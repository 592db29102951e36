// Soft-argmin regression and disparity upsampling (HBM-bound, float32, -ffp-contract=off).
//
//   k_softargmin    <- F.softmax(-cost, axis=1) + disparity_regression
//                      /root/reference/models/models.py:142,151-152,167-179
//   k_upsample_add  <- models.py:145-148,153-156
#include "lws_common.h"
#include "lws_device_math.h"

namespace lws {

// One thread per pixel.  The D costs of a pixel are D coalesced plane reads, all issued before the first use
// (DT = compile-time D keeps them in registers: one memory round trip instead of three dependent passes);
// DT = 0 is the generic fallback that re-reads through L1.  p_k = e_k / S is a correctly rounded division, as in
// the literal softmax followed by the expectation.
template <int DT>
__global__ __launch_bounds__(64) void k_softargmin(const float *__restrict__ cost, float *__restrict__ low,
                                                   int64_t plane, int D, float start)
{
    const int64_t p = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int b = blockIdx.y;
    if (p >= plane) return;
    const float *c = cost + (int64_t)b * D * plane + p;
    float r;
    if (DT > 0) {
        float v[DT > 0 ? DT : 1];
#pragma unroll
        for (int k = 0; k < DT; ++k) v[k] = c[(int64_t)k * plane];
        float m = -v[0];
#pragma unroll
        for (int k = 1; k < DT; ++k) m = fmaxf(m, -v[k]);
        float S = 0.0f;
#pragma unroll
        for (int k = 0; k < DT; ++k) {
            v[k] = lws_expf(-v[k] - m);
            S = S + v[k];
        }
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < DT; ++k) {
            float pk = v[k] / S;
            acc = acc + pk * (start + (float)k);
        }
        r = acc;
    } else {
        r = softargmin_pixel(c, plane, D, start);
    }
    low[(int64_t)b * plane + p] = r;
}

int launch_softargmin(const float *cost, float *low, int B, int D, int h, int w, float start, hipStream_t st)
{
    const int64_t plane = (int64_t)h * w;
    dim3 grid((unsigned)((plane + 63) / 64), B), block(64);   // 64-thread blocks: the stage-1 map has only h*w = 2048 pixels
    switch (D) {
        case 9: hipLaunchKernelGGL(k_softargmin<9>, grid, block, 0, st, cost, low, plane, D, start); break;
        case 24: hipLaunchKernelGGL(k_softargmin<24>, grid, block, 0, st, cost, low, plane, D, start); break;
        case 32: hipLaunchKernelGGL(k_softargmin<32>, grid, block, 0, st, cost, low, plane, D, start); break;
        default: hipLaunchKernelGGL(k_softargmin<0>, grid, block, 0, st, cost, low, plane, D, start); break;
    }
    LWS_LAUNCH_CHECK();
    return LWS_OK;
}

// One thread per full-resolution pixel; the low-resolution map is tiny and L2-resident.
__global__ __launch_bounds__(256) void k_upsample_add(const float *__restrict__ low,
                                                      const float *__restrict__ prev,
                                                      float *__restrict__ out, int h, int w, int H, int W,
                                                      float mul_a, float mul_b, float ioff)
{
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    const int b = blockIdx.z;
    if (x >= W || y >= H) return;
    const float rh = (float)h / (float)H, rw = (float)w / (float)W;
    int y0, y1, x0, x1;
    float hy0, hy1, wx0, wx1;
    src_index(y, rh, h, y0, y1, hy0, hy1, ioff);
    src_index(x, rw, w, x0, x1, wx0, wx1, ioff);
    const float *p = low + (int64_t)b * h * w;
    float p00 = (p[(int64_t)y0 * w + x0] * mul_a) * mul_b;
    float p01 = (p[(int64_t)y0 * w + x1] * mul_a) * mul_b;
    float p10 = (p[(int64_t)y1 * w + x0] * mul_a) * mul_b;
    float p11 = (p[(int64_t)y1 * w + x1] * mul_a) * mul_b;
    float top = p00 * wx0 + p01 * wx1;
    float bot = p10 * wx0 + p11 * wx1;
    float v = hy0 * top + hy1 * bot;
    const int64_t o = ((int64_t)b * H + y) * W + x;
    if (prev != nullptr) v = v + prev[o];
    out[o] = v;
}

int launch_upsample_add(const float *low, const float *prev, float *out, int B, int h, int w, int H, int W,
                        hipStream_t st, float ioff)
{
    dim3 grid(cdiv(W, 64), cdiv(H, 4), B), block(64, 4);
    hipLaunchKernelGGL(k_upsample_add, grid, block, 0, st, low, prev, out, h, w, H, W, (float)H,
                       1.0f / (float)h, ioff);
    LWS_LAUNCH_CHECK();
    return LWS_OK;
}

// ---------------------------------------------------------------------------------------------
// Soft-argmin + rescale + bilinear upsample (+ previous stage) in one launch (stage 1, where the tile of the last
// Conv3D layer cannot span D): a workgroup computes the low-resolution disparities of a 4 x 8 pixel tile plus a
// one-pixel ring into LDS, then writes the (4*s) x (8*s) full-resolution pixels they determine (s = H/h).
// ---------------------------------------------------------------------------------------------
// Tile: 4 x 8 low-resolution pixels, or 2 x 4 when that leaves most CUs without a workgroup (one pair at 256x512: 64 tiles).
template <int DT, int SU_TY, int SU_TX>
__global__ __launch_bounds__(256) void k_softargmin_upsample(const float *__restrict__ cost,
                                                             const float *__restrict__ prev, float *__restrict__ out,
                                                             float *__restrict__ low_out, int D, int h, int w, int H,
                                                             int W, float start, float mul_a, float mul_b, float ioff)
{
    constexpr int SU_HY = SU_TY + 2, SU_HX = SU_TX + 2;
    __shared__ float sLow[SU_HY * SU_HX];
    const int tid = threadIdx.x, b = blockIdx.z;
    const int ly0 = blockIdx.y * SU_TY, lx0 = blockIdx.x * SU_TX;
    const int64_t plane = (int64_t)h * w;
    if (tid < SU_HY * SU_HX) {
        const int hy = tid / SU_HX, hx = tid - hy * SU_HX;
        const int y = ly0 + hy - 1, x = lx0 + hx - 1;
        float r = 0.0f;
        if (y >= 0 && y < h && x >= 0 && x < w) {
            const float *c = cost + (int64_t)b * D * plane + (int64_t)y * w + x;
            if (DT > 0) {
                float v[DT > 0 ? DT : 1];
#pragma unroll
                for (int k = 0; k < DT; ++k) v[k] = c[(int64_t)k * plane];
                float m = -v[0];
#pragma unroll
                for (int k = 1; k < DT; ++k) m = fmaxf(m, -v[k]);
                float S = 0.0f;
#pragma unroll
                for (int k = 0; k < DT; ++k) {
                    v[k] = lws_expf(-v[k] - m);
                    S = S + v[k];
                }
                float acc = 0.0f;
#pragma unroll
                for (int k = 0; k < DT; ++k) {
                    float pk = v[k] / S;
                    acc = acc + pk * (start + (float)k);
                }
                r = acc;
            } else {
                r = softargmin_pixel(c, plane, D, start);
            }
            if (low_out != nullptr && hy >= 1 && hy <= SU_TY && hx >= 1 && hx <= SU_TX)
                low_out[(int64_t)b * plane + (int64_t)y * w + x] = r;
        }
        sLow[tid] = r;
    }
    __syncthreads();
    const int sy = H / h, sx = W / w;                     // integer upsampling factors (8, 4, 2)
    const int oh = SU_TY * sy, ow = SU_TX * sx;
    const float rh = (float)h / (float)H, rw = (float)w / (float)W;
    for (int i = tid; i < oh * ow; i += 256) {
        const int oy = i / ow, ox = i - oy * ow;
        const int y = ly0 * sy + oy, x = lx0 * sx + ox;
        if (y >= H || x >= W) continue;
        int y0, y1, x0, x1;
        float hy0, hy1, wx0, wx1;
        src_index(y, rh, h, y0, y1, hy0, hy1, ioff);
        src_index(x, rw, w, x0, x1, wx0, wx1, ioff);
        const float *p = sLow + (1 - ly0) * SU_HX + (1 - lx0);      // low-res (y,x) -> sLow[(y-ly0+1)*HX + x-lx0+1]
        float p00 = (p[y0 * SU_HX + x0] * mul_a) * mul_b;
        float p01 = (p[y0 * SU_HX + x1] * mul_a) * mul_b;
        float p10 = (p[y1 * SU_HX + x0] * mul_a) * mul_b;
        float p11 = (p[y1 * SU_HX + x1] * mul_a) * mul_b;
        float top = p00 * wx0 + p01 * wx1;
        float bot = p10 * wx0 + p11 * wx1;
        float v = hy0 * top + hy1 * bot;
        const int64_t o = ((int64_t)b * H + y) * W + x;
        if (prev != nullptr) v = v + prev[o];
        out[o] = v;
    }
}

int launch_softargmin_upsample(const float *cost, const float *prev, float *out, float *low_out, int B, int D, int h,
                               int w, int H, int W, float start, hipStream_t st, float ioff)
{
    const bool small = (long)cdiv(w, 8) * cdiv(h, 4) * B < 256;
    dim3 grid(cdiv(w, small ? 4 : 8), cdiv(h, small ? 2 : 4), B), block(256);
    const float mul_a = (float)H, mul_b = 1.0f / (float)h;
#define LWS_SU(DT)                                                                                                              \
    if (small)                                                                                                                  \
        hipLaunchKernelGGL((k_softargmin_upsample<DT, 2, 4>), grid, block, 0, st, cost, prev, out, low_out, D, h, w, H, W, start, \
                           mul_a, mul_b, ioff);                                                                                  \
    else                                                                                                                        \
        hipLaunchKernelGGL((k_softargmin_upsample<DT, 4, 8>), grid, block, 0, st, cost, prev, out, low_out, D, h, w, H, W, start, \
                           mul_a, mul_b, ioff)
    switch (D) {
        case 9: LWS_SU(9); break;
        case 24: LWS_SU(24); break;
        case 32: LWS_SU(32); break;
        default: LWS_SU(0); break;
    }
#undef LWS_SU
    LWS_LAUNCH_CHECK();
    return LWS_OK;
}

}  // namespace lws
