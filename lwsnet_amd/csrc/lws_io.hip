// The two per-pixel steps of the reference's I/O harness that sit directly on either side of LWSNet.forward, as device kernels
// (SURVEY.md section 8f row 4): the input transform of /root/reference/inference.py:83-85,102-103 (ToTensor + Normalize) and
// the output mapping of :114-115 (`.astype(np.uint8)` + cv2.applyColorMap).  Both are HBM-bound byte shuffles (4 / 16 bytes per
// pixel); what they buy is host time: in the pipelined directory mode of lwsnet_amd/inference.py the host workers are left with
// PNG decode and PNG encode only, and the uploads / downloads shrink from float32 planes to uint8 pixels.
// Arithmetic contract: the float32 operations numpy performs in lwsnet_amd/imageio.py (to_input, disparity_to_color), one IEEE
// operation each, built with correctly rounded division -- tests/test_gpu_parity.py::test_io_kernels_match_the_host_pipeline.
#include "lws_common.h"

namespace lws {

// rgb [B,H,W,3] uint8 -> out [B,3,H,W] float32:  ((v / 255) - mean[c]) / std[c]
__global__ __launch_bounds__(256) void k_preprocess_rgb8(const unsigned char *__restrict__ rgb, float *__restrict__ out, int64_t plane,
                                                         float m0, float m1, float m2, float s0, float s1, float s2)
{
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (p >= plane) return;
    const unsigned char *q = rgb + ((int64_t)b * plane + p) * 3;
    float *o = out + (int64_t)b * 3 * plane + p;
    const float r = (float)q[0] / 255.0f, g = (float)q[1] / 255.0f, bl = (float)q[2] / 255.0f;
    o[0] = (r - m0) / s0;
    o[plane] = (g - m1) / s1;
    o[2 * plane] = (bl - m2) / s2;
}

// disp [N] float32 -> rgb [N,3] uint8 = lut[(uint8)(int64)disp]: the C cast of numpy's .astype(np.uint8) (truncation toward zero,
// wrap-around outside 0..255; values no int64 holds, and NaN, become INT64_MIN there: low byte 0)
__global__ __launch_bounds__(256) void k_apply_lut8(const float *__restrict__ disp, const unsigned char *__restrict__ lut,
                                                    unsigned char *__restrict__ rgb, int64_t n)
{
    __shared__ unsigned char sLut[768];
    for (int i = threadIdx.x; i < 768; i += 256) sLut[i] = lut[i];
    __syncthreads();
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const float v = disp[p];
    unsigned idx = 0;
    if (fabsf(v) < 9.2233715e18f) idx = (unsigned)((long long)v & 0xFF);
    unsigned char *o = rgb + p * 3;
    o[0] = sLut[idx * 3 + 0];
    o[1] = sLut[idx * 3 + 1];
    o[2] = sLut[idx * 3 + 2];
}

}  // namespace lws

using namespace lws;

extern "C" {

int lws_preprocess_rgb8(const uint8_t *rgb, float *out, int B, int H, int W, const float *mean, const float *std, void *stream)
{
    LWS_CHECK_ARG(rgb && out && mean && std, "preprocess_rgb8: null pointer");
    LWS_CHECK_ARG(B >= 1 && H >= 1 && W >= 1, "preprocess_rgb8: bad shape B=%d %dx%d", B, H, W);
    for (int c = 0; c < 3; ++c) LWS_CHECK_ARG(std[c] != 0.0f, "preprocess_rgb8: std[%d] is zero", c);
    const int64_t plane = (int64_t)H * W;
    dim3 grid((unsigned)((plane + 255) / 256), B), block(256);
    hipLaunchKernelGGL(k_preprocess_rgb8, grid, block, 0, (hipStream_t)stream, rgb, out, plane, mean[0], mean[1], mean[2], std[0], std[1],
                       std[2]);
    LWS_LAUNCH_CHECK();
    return LWS_OK;
}

int lws_apply_lut8(const float *disp, const uint8_t *lut, uint8_t *rgb, int64_t n, void *stream)
{
    LWS_CHECK_ARG(disp && lut && rgb, "apply_lut8: null pointer");
    LWS_CHECK_ARG(n >= 1, "apply_lut8: bad size");
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    hipLaunchKernelGGL(k_apply_lut8, grid, block, 0, (hipStream_t)stream, disp, lut, rgb, n);
    LWS_LAUNCH_CHECK();
    return LWS_OK;
}

}  // extern "C"
