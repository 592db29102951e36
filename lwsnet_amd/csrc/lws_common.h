// Internal declarations shared by the HIP translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <string>
#include <vector>

#include "lwsnet_hip.h"

namespace lws {

void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

#define LWS_CHECK_ARG(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            ::lws::set_error(__VA_ARGS__);       \
            return LWS_ERR_INVALID;              \
        }                                        \
    } while (0)

#define LWS_HIP(call)                                                                           \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            ::lws::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__,   \
                             __LINE__);                                                         \
            return LWS_ERR_HIP;                                                                 \
        }                                                                                       \
    } while (0)

#define LWS_LAUNCH_CHECK() LWS_HIP(hipGetLastError())

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// One BatchNorm3D -> ReLU -> Conv3D layer, device side.
struct Conv3dLayer {
    int cin = 0, cout = 0;
    float *w = nullptr;      // packed weights (layout depends on the kernel that consumes them)
    float *bn_s = nullptr;   // [cin] scale of THIS layer's BatchNorm (applied to its input)
    float *bn_t = nullptr;   // [cin] shift
};

struct Stage3d {
    int c3 = 0;
    std::vector<Conv3dLayer> layers;   // layers_3d + 2
};

}  // namespace lws

struct lws_prof_rec {
    int kc;
    hipEvent_t t0, t1;
};

struct lws_ctx {
    lws_config cfg;
    bool prof_on = false;
    std::vector<lws_prof_rec> prof;          // records of the current session
    std::vector<hipEvent_t> evt_pool;        // events available for reuse
    int device = 0;
    bool finalized = false;
    std::map<std::string, std::vector<float>> host;        // state dict as given
    std::map<std::string, std::vector<int64_t>> shapes;
    std::map<std::string, std::vector<int64_t>> spec;      // accepted keys -> shapes (from cfg)
    float *params = nullptr;                                // one device slab for all packed params
    size_t params_bytes = 0;
    lws::Stage3d stage[3];
    // activation workspace (grown by lws_reserve / on demand)
    float *ws = nullptr;
    size_t ws_bytes = 0;
};

namespace lws {

// ---- kernel launchers (lws_volume.hip, lws_regress.hip, lws_conv3d.hip) ----
int launch_volume_l1_shift(const float *L, const float *R, float *cost, int B, int C, int h, int w, int D,
                           hipStream_t st);
int launch_volume_l1_warp(const float *L, const float *R, const float *prev, float *cost, float *wflow_out,
                          int B, int C, int h, int w, int H, int W, int m, hipStream_t st);
int launch_softargmin(const float *cost, float *low, int B, int D, int h, int w, float start, hipStream_t st);
int launch_upsample_add(const float *low, const float *prev, float *out, int B, int h, int w, int H, int W,
                        hipStream_t st);

// conv3d stack pieces; activations are channels-last [B,D,h,w,C3]
int launch_conv3d_first(const Stage3d &s, const float *cost, float *act_out, int B, int D, int h, int w,
                        hipStream_t st);
int launch_conv3d_mid(const Stage3d &s, int layer, const float *act_in, float *act_out, int B, int D, int h,
                      int w, hipStream_t st);
int launch_conv3d_last(const Stage3d &s, const float *act_in, const float *cost_skip, float *cost_out, int B,
                       int D, int h, int w, hipStream_t st);

// host-side weight packing used by lws_finalize
size_t packed_mid_weight_floats(int c3);
void pack_mid_weights(const float *w /*[c3][c3][27]*/, int c3, float *out);

}  // namespace lws
