// Internal declarations shared by the HIP translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <atomic>
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

// A cross-stream fork without a marker packet on the producer's queue: an event bound to the producer kernel's OWN completion
// signal -- hipExtLaunchKernelGGL(..., stopEvent) -- instead of hipEventRecord behind it (tools/micro/event_cost.hip;
// profiles/NOTES.md, "fork cost").  Protocol: a StopArm in the caller's scope arms the event, the caller calls a launcher, and
// StopArm::finish records the event the ordinary way if the launcher's kernel did not take it -- so a launch path that does not
// know about stop events stays correct; the guard's destructor disarms on every exit path, so an error return can never leave a
// stale event armed for the next call on this thread.  Launch sites that honour it use LWS_LAUNCH_STOP instead of
// hipLaunchKernelGGL.  Thread-local: handles are driven from their own host threads (lws_pool).
// A kernel-bound event is NOT a capture-time record: under hipGraph capture (use_ext = false) the guard never arms and
// finish() records the event with hipEventRecord, which is what pulls the waiting stream into the capture.
extern thread_local hipEvent_t tl_stop_event;
static inline void stop_event_arm(hipEvent_t e) { tl_stop_event = e; }
static inline hipEvent_t stop_event_take()
{
    hipEvent_t e = tl_stop_event;
    tl_stop_event = nullptr;
    return e;
}
struct StopArm {
    hipEvent_t e;
    hipStream_t st;
    bool ext;
    StopArm(hipEvent_t e_, hipStream_t st_, bool use_ext) : e(e_), st(st_), ext(use_ext && e_ != nullptr)
    {
        if (ext) stop_event_arm(e);
    }
    StopArm(const StopArm &) = delete;
    StopArm &operator=(const StopArm &) = delete;
    // after the launcher returned rc: the event must be complete once that kernel is
    hipError_t finish(int rc)
    {
        const bool taken = ext && stop_event_take() == nullptr;      // the launcher's kernel carries it
        if (e == nullptr || taken || rc != 0) return hipSuccess;
        return hipEventRecord(e, st);
    }
    ~StopArm() { if (ext) (void)stop_event_take(); }
};
#define LWS_LAUNCH_STOP(kernel, grid, block, lds, st, ...)                                                  \
    do {                                                                                                    \
        hipEvent_t se_ = ::lws::stop_event_take();                                                          \
        if (se_ != nullptr)                                                                                 \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, st, nullptr, se_, 0, __VA_ARGS__);              \
        else                                                                                                \
            hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                                  \
    } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (function, device).  One mask per launch site, one bit per
// device; handles on different host threads reach a launch site concurrently (lws_pool, bench.py's pipelined mode), so the
// mask is atomic -- a lost race only repeats the idempotent call.  Devices >= 64 set the attribute on every launch.
static inline int ensure_dyn_lds(std::atomic<uint64_t> &done, const void *fn, int bytes)
{
    int dev = 0;
    LWS_HIP(hipGetDevice(&dev));
    const uint64_t bit = dev >= 0 && dev < 64 ? (uint64_t)1 << dev : 0;
    if (bit != 0 && (done.load(std::memory_order_acquire) & bit)) return LWS_OK;
    LWS_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    if (bit != 0) done.fetch_or(bit, std::memory_order_release);
    return LWS_OK;
}

// 16-byte activation store.  wt (wave-uniform) selects a write-through store (sc0 sc1): the line does not stay dirty in
// the XCD's L2, so the end-of-kernel release has less to write back.  It pays for small outputs and costs for large ones
// (profiles/NOTES.md, "write-through stores"), so launchers enable it through use_wt_stores.
__device__ __forceinline__ void store_act4(float *p, float4 v, int wt)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (wt) {
        typedef float fx4_ __attribute__((ext_vector_type(4)));
        const fx4_ r = {v.x, v.y, v.z, v.w};
        // (s_nop 1 inside the string: hipcc must not overwrite the data registers before the store has read them)
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(r) : "memory");
        return;
    }
#endif
    *reinterpret_cast<float4 *>(p) = v;
}
static inline int use_wt_stores(size_t out_bytes) { return out_bytes <= ((size_t)40 << 20) ? 1 : 0; }

// Diagnostic builds only (LWS_EXTRA_FLAGS="-DLWS_STAMPS=<kernel id>"; tools/stamps.py): every workgroup of the selected
// kernel stores s_memtime stamps of its phases (slots 0..5) and the s_memrealtime (100 MHz) of its first and latest stamp
// (slots 6, 7: the in-kernel clock = d s_memtime / d s_memrealtime x 100 MHz) in a per-translation-unit buffer.  The shipped library compiles
// LWS_STAMPK to nothing.  Kernel ids: 1 mid16, 3 conv3d_last, 4 conv3d_first, 5 ref_dws, 6 ref_conv64,
// 7 conv2d_nchw, 8 ref_first, 9 ref_last, 10 volume_warp, 11 volume_shift, 12 softargmin_upsample, 13..16 conv2d_pair (dres0, dres1, conv1+2, conv3+4), 18 conv3d_mid8q, 19 conv3d_mid16x, 20 conv3d_mid8x, 21 ref_conv64x.
#ifdef LWS_STAMPS
#define LWS_DEFINE_STAMPS(tu)                                                                                      \
    __device__ unsigned long long g_stamps_##tu[4096 * 8];                                                         \
    static __device__ __forceinline__ void stamp_(int i)                                                           \
    {                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        unsigned long long t_, r_;                                                                                 \
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "=s"(r_)::"memory");  \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        const unsigned bid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);                       \
        if (threadIdx.x == 0 && threadIdx.y == 0 && bid < 4096) {                                                  \
            g_stamps_##tu[bid * 8 + i] = t_;                                                                       \
            g_stamps_##tu[bid * 8 + (i == 0 ? 6 : 7)] = r_;   /* 100 MHz wall clock at the first / latest stamp */  \
            if (i == 0) {   /* slot 5: where the workgroup runs -- HW_ID (cu 11:8, sh 12, se 15:13) | XCC_ID << 32 */      \
                unsigned hw_, xcc_;                                                                                \
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw_), "=s"(xcc_)); \
                g_stamps_##tu[bid * 8 + 5] = ((unsigned long long)(xcc_ & 15u) << 32) | hw_;                        \
            }                                                                                                      \
        }                                                                                                          \
    }                                                                                                              \
    extern "C" int lws_debug_read_stamps_##tu(unsigned long long *out, int n)                                      \
    {                                                                                                              \
        return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps_##tu), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -1; \
    }
#define LWS_STAMPK(k, i)                 \
    do {                                 \
        if ((k) == LWS_STAMPS) stamp_(i); \
    } while (0)
#else
#define LWS_DEFINE_STAMPS(tu)
#define LWS_STAMPK(k, i) do {} while (0)
#endif

// One BatchNorm3D -> ReLU -> Conv3D layer, device side.
struct Conv3dLayer {
    int cin = 0, cout = 0;
    float *w = nullptr;      // packed weights (layout depends on the kernel that consumes them)
    float *bn_s = nullptr;   // [cin] scale of THIS layer's BatchNorm (applied to its input)
    float *bn_t = nullptr;   // [cin] shift
    float *w_mfma = nullptr; // first layer only: MFMA A fragments, [9][64] (c3 = 8, k_conv3d_first8) or [c3/16][7][64]
};

struct Stage3d {
    int c3 = 0;
    bool mid8_split = false;           // 8 -> 8 layers: false = k_conv3d_mid8q (4x4x1_16B, the oracle's chain), true = k_conv3d_mid8x (split-bf16, not bit-exact)
    bool mid16_split = false;          // 32 -> 32 layers: false = k_conv3d_mid16 (f32 MFMA, the oracle's chain), true = k_conv3d_mid16x (split-bf16, not bit-exact)
    int mid8_balance = 1;              // k_conv3d_mid8q on small grids: the small tile whenever that is the fullest CU's shorter schedule (0 in lws_pool workers)
    int cu_count = 0;                  // compute units of the handle's device (0 = unknown: 256)
    unsigned long long *clk = nullptr; // lws_clock_stamp: k_conv3d_mid16 leaves its shader / wall clocks here (64 x 4 values)
    std::vector<Conv3dLayer> layers;   // layers_3d + 2
};

// One 3x3 layer of the 2D feature extractor (NCHW planes): conv or stride-2 transposed conv,
// then BatchNorm (bn_s != nullptr) -> + residual -> ReLU.
struct Conv2dLayer {
    int cin = 0, cout = 0, stride = 1, pad = 1, dil = 1;
    bool transposed = false, relu = false;
    float *w = nullptr, *bn_s = nullptr, *bn_t = nullptr;
    float *w_pair = nullptr;   // [tap][pair_groups][cin][cout/pair_groups] for k_conv2d_pair (layers 0..7 only)
    float *w_mfma = nullptr;   // [tap][lane][cin/4] A fragments for k_conv2d_pair_mfma (16-output-channel layers 4..7)
    int pair_groups = 0;
};

// Refinement: BatchNorm(32) -> ReLU -> depthwise 3x3 (dil) -> pointwise 32->32
struct RefDws {
    int dil = 1;
    float *bn_s = nullptr, *bn_t = nullptr;
    float *dw = nullptr;   // [tap][32]
    float *pw = nullptr;   // MFMA A fragments [q][mt][lane][4]
};

struct RefConv64 {
    float *bn_s = nullptr, *bn_t = nullptr;   // [64]
    float *w = nullptr;                       // MFMA A fragments [tap][qq][mt][lane][4]
    float *wx = nullptr;                      // split-bf16 A fragments [step][mt][variant][lane][8 bf16] (k_ref_conv64x)
    int form = 0;                             // 0 = k_ref_conv64 (f32 MFMA, the oracle's chain), 1 = k_ref_conv64x (not bit-exact)
};

struct Net2d {
    Conv2dLayer fe[12];          // feature extractor, in execution order
    float *r1_first[2] = {nullptr, nullptr};   // refinement1_left / _disp first conv, [tap][cin][32]
    float *r1_first_mfma[2] = {nullptr, nullptr};   // ... as A fragments for first_conv_mfma, [mt][j][lane]
    RefDws r1[2][4];             // refinement1_{left,disp} blocks 1..4
    RefConv64 r2_first;
    RefDws r2[4];
    float *r2_last = nullptr;    // [tap][32]
};

}  // namespace lws

struct lws_prof_rec {
    int kc;
    hipEvent_t t0, t1;
};

struct lws_ctx {
    lws_config cfg;
    // schedule options (lws_set_option): every setting computes the same bits, only the launch plan differs
    struct {
        int fuse_first = 3;        // bit 0: refinement1_disp's 1 -> 32, bit 1: refinement1_left's 3 -> 32 convolution inside their first depthwise blocks
        int defer_upsample = 1;    // batches <= 2: consumers evaluate the stage-2/3 maps (no k_upsample_add launches)
        int side_streams = 1;      // 0: no handle-owned side stream, the whole forward on the caller's stream (lws_pool workers)
        int split_bf16 = 0;        // bit mask of the MFMA convolutions on split-bf16 operands (NOT bit-exact): 1 = Conv3D 32 -> 32, 2 = Conv3D 8 -> 8, 4 = refinement2[0]
        int ref_pipe = -1;         // refinement chunks alternating over two streams: -1 = from four chunks up, 0 = never, 1 = from two chunks up
        int warp_form = 1;         // residual volumes: 1 = right-feature window of a 64-pixel row segment staged in LDS, 0 = every tap gathered from global memory
        int fuse_last1 = 1;        // batches <= 2: stage 1's last Conv3D layer + soft-argmin in one launch, pred1 evaluated by its consumers
        int fork2_after = -1;      // the second fork: 0 = behind stage 1's last Conv3D layer, k = behind its k-th middle layer, -1 = automatic (the last middle layer)
        int fuse_ref_last = -1;    // refinement2's last block + the 32 -> 1 convolution + pred3 in one launch: -1 = batch 1 only, 0 / 1
        int ref_chunk_mb = 72;     // refinement in chunks of pairs whose maps are at most this many MB each (0 = one chunk); see refine_chunk
    } opt;
    unsigned prof_mask = 0;                  // kernel classes being timed in the current call
    unsigned prof_mask_cfg = 0;              // ... as configured by lws_profile_enable
    int prof_every = 1;                      // lws_profile_sample: lws_forward records events on every n-th call only
    unsigned prof_calls = 0;
    std::vector<lws_prof_rec> prof;          // records of the current session
    std::vector<hipEvent_t> evt_pool;        // events available for reuse
    int device = 0;
    int cu_count = 0;                        // compute units of `device` (lws_finalize)
    bool finalized = false;
    std::map<std::string, std::vector<float>> host;        // state dict as given
    std::map<std::string, std::vector<int64_t>> shapes;
    std::map<std::string, std::vector<int64_t>> spec;      // accepted keys -> shapes (from cfg)
    float *params = nullptr;                                // one device slab for all packed params
    bool owns_params = true;                                // false: a clone (lws_clone) sharing its source's slab
    size_t params_bytes = 0;
    lws::Stage3d stage[3];
    lws::Net2d net2d;
    bool have_2d = false;                                   // all 2D tensors were given
    // activation workspace (grown by lws_reserve / on demand)
    float *ws = nullptr;
    size_t ws_bytes = 0;
    int mid8_balance = 1;                                   // (not an option: lws_pool sets 0 on its workers' clones)
    // side stream for the branches of the forward that run beside the stage loop (feature tail, refinement1_left)
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_feat[3] = {nullptr, nullptr, nullptr};   // f8 / f4 / f2 complete
    hipEvent_t ev_fork2 = nullptr;                          // second fork of lws_forward (beside the end of stage 1's Conv3D stack)
    unsigned long long *clk_buf = nullptr;                  // lws_clock_stamp's device buffer
};

namespace lws {

// ---- kernel launchers (lws_volume.hip, lws_regress.hip, lws_conv3d.hip) ----
// q16: round the feature values to fp16 where they are read (lws_config.feature_fp16)
int launch_volume_l1_shift(const float *L, const float *R, float *cost, int B, int C, int h, int w, int D,
                           hipStream_t st, bool q16 = false);
int launch_volume_l1_warp(const float *L, const float *R, const float *prev, float *cost, float *wflow_out,
                          int B, int C, int h, int w, int H, int W, int m, hipStream_t st, bool q16 = false,
                          const float *plow = nullptr, int ph = 0, int pw = 0, float *pmat = nullptr,     // deferred prev map
                          int form = 1,    // 1 = right-feature window staged in LDS, 0 = every tap gathered from global memory
                          const float *plow0 = nullptr, int ph0 = 0, int pw0 = 0, float *pmat0 = nullptr,    // prev == nullptr: its own deferred source
                          float ioff = 0.5f);   // src_index's offset: 0.5f = interp_align_mode 0, 0.0f = 1 (every `ioff` below)
int launch_softargmin(const float *cost, float *low, int B, int D, int h, int w, float start, hipStream_t st);
int launch_upsample_add(const float *low, const float *prev, float *out, int B, int h, int w, int H, int W,
                        hipStream_t st, float ioff = 0.5f);
int launch_softargmin_upsample(const float *cost, const float *prev, float *out, float *low_out, int B, int D, int h,
                               int w, int H, int W, float start, hipStream_t st, float ioff = 0.5f);

// conv3d stack pieces; activations are channels-last [B,D,h,w,C3]
bool shift_first_can_fuse(const Stage3d &s, int C);
int launch_shift_first(const Stage3d &s, const float *L, const float *R, float *cost, float *act_out, int B, int C, int D,
                       int h, int w, hipStream_t st, bool q16);
int launch_conv3d_first(const Stage3d &s, const float *cost, float *act_out, int B, int D, int h, int w,
                        hipStream_t st);
// e0/e1 (optional, C3 % 16 == 0 only): events stamped with the kernel's own begin / end (hipExtLaunchKernelGGL)
int launch_conv3d_mid(const Stage3d &s, int layer, const float *act_in, float *act_out, int B, int D, int h,
                      int w, hipStream_t st, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
int launch_conv3d_last(const Stage3d &s, const float *act_in, const float *cost_skip, float *cost_out, int B,
                       int D, int h, int w, hipStream_t st);

// 2D networks (lws_conv2d.hip)
int launch_conv2d_nchw(const Conv2dLayer &l, const float *in, const float *res, float *out, int N, int H, int W,
                       hipStream_t st, const float *in2 = nullptr, int n1 = 0);
int conv2d_pair_groups(int layer);
void pack_pair_mfma(const float *w, int cin, float *out);
int launch_conv2d_pair(const Conv2dLayer &a, const Conv2dLayer &b, const float *in, const float *res, float *out, int N,
                       int H, int W, hipStream_t st, const float *in2 = nullptr, int n1 = 0);
int launch_ref_first(const float *in, int cin, const float *w, float *out, int B, int H, int W, hipStream_t st);
int launch_ref_dws(const RefDws &l, const float *in, float *out, int B, int H, int W, hipStream_t st);
bool ref_first_dws_can_fuse(const RefDws &l, int cin);
int launch_ref_first_dws(const RefDws &l, const float *img, int cin, const float *wfrag, float *out, int B, int H, int W,
                         hipStream_t st, const float *plow = nullptr, int ph = 0, int pw = 0, float *pmat = nullptr,
                         float ioff = 0.5f);
int packed_first_mfma_floats(int cin);
void pack_first_mfma(const float *w /*[32][cin][3][3]*/, int cin, float *out);
int launch_ref_conv64(const RefConv64 &l, const float *inL, const float *inD, float *out, int B, int H, int W,
                      hipStream_t st);
int launch_ref_last(const float *in, const float *w, const float *pred3, float *out, int B, int H, int W, hipStream_t st);
bool ref_dws_last_can_fuse(const RefDws &l);
int launch_ref_dws_last(const RefDws &l, const float *in, const float *wlast, const float *pred3, float *out, int B, int H, int W,
                        hipStream_t st);
void pack_conv2d_mfma(const float *w, int cin, int ktaps, float *out);
size_t packed_conv64x_floats();
void pack_conv64_bf16x3(const float *w, float *out);   // [32][64][3][3] -> k_ref_conv64x fragments

bool conv3d_last_can_fuse(const Stage3d &s, int D);
int launch_conv3d_last_softargmin(const Stage3d &s, const float *act_in, const float *cost_skip, float *cost_out,
                                  float *low, float start, int B, int D, int h, int w, hipStream_t st);

// host-side weight packing used by lws_finalize
size_t packed_mid_weight_floats(int c3);
void pack_mid_weights(const float *w /*[c3][c3][27]*/, int c3, float *out);
void pack_first8_weights(const float *w /*[8][27]*/, float *out /*[9][64]*/);
void pack_first16_weights(const float *w /*[c3][27]*/, int c3, float *out /*[c3/16][7][64]*/);

}  // namespace lws
