/* lwsnet_hip.h -- C ABI of the MI355X (gfx950) LWSNet disparity hot path.
 *
 * The reference (PrinceVictor/LWSNet) has no FFI/plugin layer: the path is Python
 * calling PaddlePaddle ops (models/models.py).  This header is the boundary a
 * maintainer binds with ctypes from models/models.py; every
 * entry point names the reference lines it replaces.  Plain pointers and sizes
 * only; all tensors are float32, contiguous, NCHW, in DEVICE memory unless the
 * parameter is called `host`.  `stream` is a hipStream_t passed as void*
 * (NULL = the default stream).  Calls are asynchronous on that stream.
 *
 * Every function returns 0 on success or a negative lws_status; the message of
 * the last failure on the calling thread is lws_last_error().
 * A handle is NOT thread-safe; use one handle per (process, device, stream); distinct handles may be driven from
 * distinct host threads concurrently (lws_clone, lws_pool).
 * A handle belongs to the HIP device that was current when lws_create ran (or lws_set_option(h, "device", n) before
 * anything was allocated): every call that touches the GPU through it returns LWS_ERR_INVALID unless that device is the
 * calling thread's current device.  The library never changes the caller's current device.
 */
#ifndef LWSNET_HIP_H
#define LWSNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LWS_ABI_VERSION 8

typedef enum {
    LWS_OK = 0,
    LWS_ERR_INVALID = -1,      /* bad argument / unsupported shape (Python shim raises ValueError) */
    LWS_ERR_HIP = -2,          /* a HIP runtime call failed (RuntimeError) */
    LWS_ERR_STATE = -3,        /* missing tensor / not finalized (RuntimeError) */
    LWS_ERR_NOMEM = -4
} lws_status;

/* Constructor arguments of LWSNet(args): models/models.py:8-14, defaults inference.py:23-26. */
typedef struct {
    int32_t maxdisplist[3];   /* {24,5,5}: stage-1 hypotheses, stage-2/3 residual half-range m (D = 2m-1) */
    int32_t layers_3d;        /* 4 */
    int32_t channels_3d;      /* 8 */
    int32_t growth_rate[3];   /* {4,1,1}: c3 of stage i = channels_3d * growth_rate[i] */
    int32_t feature_fp16;     /* 0 (reference behaviour).  1 = BASELINE config 5: the three feature maps are rounded to
                                 fp16 (round-to-nearest-even) where the volume kernels read them; everything else,
                                 including the soft-argmin, stays float32.  Not part of the reference; cannot meet the
                                 1e-3 px tolerance, judged on 3-px error. */
    int32_t interp_align_mode; /* 0 (default) / 1: which source index F.interpolate(mode="bilinear") uses in the reference's four
                                 resizes (models/models.py:119,146,154,161; Paddle's align_corners=False, align_mode).
                                 0 = half-pixel centres, src = ratio * (dst + 0.5) - 0.5 -- what the oracle bets Paddle 2.0rc0
                                 does; 1 = src = ratio * dst, the Paddle 1.x / 2.0-beta default.  The
                                 reference cannot be run here (no Paddle wheel), so the bet is a switch, not a constant: every
                                 resize of the path takes its taps from one helper (src_index, lws_device_math.h), the C oracle
                                 has the same switch (lwso_set_align_mode) and both values are bit-exact against it
                                 (tests/test_gpu_parity.py::test_interp_align_mode_*).  The per-op entry points without a
                                 handle (lws_volume_l1_warp, lws_upsample_add) compute mode 0.  (ABI v8) */
} lws_config;

typedef struct lws_ctx *lws_handle;

int lws_abi_version(void);
const char *lws_last_error(void);

/* Number of HIP devices visible / name of device `dev` (diagnostics only). */
int lws_device_count(void);

/* ---- model object: models/models.py:8-26 -------------------------------------------- */
int lws_create(const lws_config *cfg, lws_handle *out);
int lws_destroy(lws_handle h);

/* model.set_state_dict (inference.py:45): one call per state-dict entry, HOST pointer.
 * Keys are the Paddle structured names, e.g. "volume_postprocess.0.1.2.weight"
 * (lwsnet_amd/weights.py lists all 226).  Keys outside the hot path are stored too
 * (used by lws_forward once the 2D networks run natively) and unknown keys are an error. */
int lws_set_tensor(lws_handle h, const char *key, const float *host, const int64_t *shape, int ndim);

/* Folds eval-mode BatchNorm3D into (scale, shift) pairs, packs the Conv3D weights into
 * MFMA fragment order and uploads them.  Must be called after the last lws_set_tensor
 * and before any function that takes a handle + stage. */
int lws_finalize(lws_handle h);

/* Pre-allocates the activation workspace for batches up to B pairs of H x W and creates the handle's side stream
 * and cross-stream events (unless option "side_streams" is 0), so that later calls allocate nothing (required before
 * hipGraph capture). */
int lws_reserve(lws_handle h, int B, int H, int W);

/* ---- per-op entry points (each is one kernel launch) --------------------------------- */

/* LWSNet._build_volume_2d, models/models.py:58-76 (stride 1).
 * cost[b,d,y,x] = sum_c |L[b,c,y,x] - (x>=d ? R[b,c,y,x-d] : 0)|;  L,R [B,C,h,w] -> cost [B,D,h,w]. */
int lws_volume_l1_shift(const float *L, const float *R, float *cost,
                        int B, int C, int h, int w, int D, void *stream);

/* forward() glue models/models.py:119-121 + LWSNet._build_volume_2d3 :78-104 + warp :28-55.
 * prev_disp [B,1,H,W] is the previous stage's full-resolution disparity; the kernel
 * resizes it to [h,w] (half-pixel bilinear), scales by h/H, and for k = 0..2m-2 samples R at
 * x - wflow + (k-(m-1)) through the reference's normalise/grid_sample float32 round trip.
 * L,R [B,C,h,w] -> cost [B,2m-1,h,w].  If wflow_out != NULL the resized flow [B,h,w] is stored. */
int lws_volume_l1_warp(const float *L, const float *R, const float *prev_disp, float *cost,
                       float *wflow_out, int B, int C, int h, int w, int H, int W, int m, void *stream);

/* volume_postprocess[stage](cost) + cost, models/models.py:136-138 with post_3dconvs,
 * models/submodules.py:190-221: 6 x (BatchNorm3D(eval) -> ReLU -> Conv3D 3x3x3 s1 p1) + skip.
 * cost_in, cost_out [B,D,h,w] (may not alias). */
int lws_conv3d_stack(lws_handle h, int stage, const float *cost_in, float *cost_out,
                     int B, int D, int hh, int ww, void *stream);

/* F.softmax(-cost, axis=1) + disparity_regression, models/models.py:142,151-152,167-179.
 * cost [B,D,h,w] -> disp_low [B,h,w]; hypothesis values are start, start+1, ... */
int lws_softargmin(const float *cost, float *disp_low, int B, int D, int h, int w, float start, void *stream);

/* models/models.py:145-148,153-156: out = bilinear_resize(disp_low * H / h -> [H,W]) (+ prev).
 * disp_low [B,h,w]; prev (may be NULL) and out [B,1,H,W]. */
int lws_upsample_add(const float *disp_low, const float *prev, float *out,
                     int B, int h, int w, int H, int W, void *stream);

/* ---- whole path: the body of `for scale in range(3)`, models/models.py:115-156 ------ */
/* featsL/featsR: the three feature maps of feature_extraction.  With H2 = ceil(H/2), W2 = ceil(W/2) (the stem
 * convolution is k3 s2 dil2 pad2, models/submodules.py:118-125; both must be divisible by 4):
 * 1/8: [B,16,H2/4,W2/4], 1/4: [B,16,H2/2,W2/2], 1/2: [B,8,H2,W2] -- for H = 8k these are H/8, H/4, H/2, for the equally
 * legal H = 8k-1 they are NOT floor(H/8) ... (63 rows -> 8, 16, 32).  pred_out[s] [B,1,H,W] for s = 0..2. */
int lws_disparity_stages(lws_handle h, const float *const featsL[3], const float *const featsR[3],
                         int B, int H, int W, float *const pred_out[3], void *stream);

/* ---- the 2D networks around the path ---------- */
/* feature_extraction, models/submodules.py:113-188 (+ hourglass :35-109): img [N,3,H,W] ->
 * f8 [N,16,H2/4,W2/4], f4 [N,16,H2/2,W2/2], f2 [N,8,H2,W2] with H2 = ceil(H/2), W2 = ceil(W/2) as above. */
int lws_feature_extraction(lws_handle h, const float *img, int N, int H, int W, float *f8, float *f4, float *f2,
                           void *stream);

/* models/models.py:158-162 with refinement1/refinement2, models/submodules.py:223-327:
 * pred4 = pred3 + refinement2(concat(refinement1_left(left), refinement1_disp(pred3))).
 * left [B,3,H,W]; pred3, pred4 [B,1,H,W].  Any H, W > 0 (the refinement has no stride, hence no size rule). */
int lws_refine(lws_handle h, const float *left, const float *pred3, int B, int H, int W, float *pred4, void *stream);

/* LWSNet.forward, models/models.py:106-164: left, right [B,3,H,W] -> pred_out[0..3] [B,1,H,W]. */
int lws_forward(lws_handle h, const float *left, const float *right, int B, int H, int W, float *const pred_out[4],
                void *stream);

/* ---- the per-pixel steps of the I/O harness on either side of forward (ABI v8) ------------- */
/* inference.py:83-85,102-103: ToTensor + Normalize of the (already cropped) image.  rgb [B,H,W,3] uint8, device memory ->
 * out [B,3,H,W] float32 = ((v / 255) - mean[c]) / std[c], one IEEE float32 operation each -- bit for bit what numpy computes in
 * lwsnet_amd/imageio.py:to_input.  mean, std: HOST pointers to 3 floats (the ImageNet constants of dataloader/dataloader.py:10-11). */
int lws_preprocess_rgb8(const uint8_t *rgb, float *out, int B, int H, int W, const float *mean, const float *std, void *stream);
/* inference.py:114-115: `.astype(np.uint8)` (C cast: truncation toward zero, wrap-around outside 0..255) + cv2.applyColorMap.
 * disp [n] float32 -> rgb [n,3] uint8 = lut[(uint8)(int64)disp]; lut: 256 x 3 bytes in device memory (lwsnet_amd.imageio.jet_lut). */
int lws_apply_lut8(const float *disp, const uint8_t *lut, uint8_t *rgb, int64_t n, void *stream);

/* Launch-plan options of lws_forward / lws_disparity_stages.  They change which kernels / streams carry the work, never
 * the arithmetic: every setting returns the same bits (tests/test_gpu_parity.py::test_forward_schedule_options) -- except
 * the opt-in numerics mode "split_bf16".  (ABI v8 removed the options two rounds of sweeps had retired: left_at, split_heads,
 * fuse_shift, conv3d_order, mid8_tile, mid8_balance, fork_ext, tail_at, and folded mid8_form / mid16_form / conv64_form into
 * "split_bf16"; what was measured against what is in profiles/NOTES.md.)
 *   "fuse_first"     bit mask, 3 (default): bit 0 = refinement1_disp's 1 -> 32 convolution, bit 1 = refinement1_left's 3 -> 32
 *                    convolution inside their first depthwise blocks (k_ref_dws<CIN>: the convolution recomputed on the halo
 *                    tile on fp32 MFMA; one launch and one 32-channel map less each)
 *   "defer_upsample" 1 (default) = at batches <= 2 the consumers evaluate the stage-2/3 maps
 *   "split_bf16"     0 (default) or a bit mask: 1 = the 32 -> 32 Conv3D layers (k_conv3d_mid16x), 2 = the 8 -> 8 Conv3D layers
 *                    of stages 2, 3 (k_conv3d_mid8x; samples under 256 tiles stay on the exact kernel), 4 = refinement2[0]
 *                    (k_ref_conv64x); 7 = all of them.  Split-bf16 MFMA: each float32 operand as three bf16 values, six exact
 *                    cross products accumulated in float32 -- ~2.5x the MFMA issue rate at float32-level accuracy, gated
 *                    against the float64 oracle by tests/test_gpu_parity.py::test_split_bf16_*, but NOT the oracle's bits: an
 *                    opt-in numerics mode, never what bench.py's headline measures
 *   "side_streams"   1 (default) = refinement1_left and the feature-extractor tail run on a handle-owned side stream;
 *                    0 = the whole forward on the caller's stream, no forks / joins (what lws_pool workers use)
 *   "ref_chunk_mb"   72 (default): the refinement runs in chunks of pairs whose [b,H,W,32] maps are at most this many MB
 *                    each, so that a chunk's maps stay in the 256 MiB Infinity Cache between layers (batch 8 at 256x512:
 *                    two chunks of 4; 368x1232: one pair per chunk); 0 = one chunk
 *   "ref_pipe"       -1 (default: on from four chunks up), 0 / 1: consecutive refinement chunks alternate between the caller's
 *                    stream and the handle's side stream, one chunk's memory-bound blocks beside the other's 64 -> 32 convolution
 *   "fork2_after"    -1 (default: behind the last middle layer) / 0 / k: where the second fork of lws_forward sits -- behind
 *                    stage 1's last Conv3D layer (0) or behind its k-th middle layer, so that the side branch starts beside the
 *                    end of the stage-1 stack
 *   "warp_form"      residual volumes of stages 2 and 3: 1 (default) = k_volume_l1_warp stages the right-feature window of a
 *                    64-pixel row segment (all channels, zero-filled outside the image) in LDS and computes the 2m - 1
 *                    hypotheses from it; 0 = every tap gathered from global memory (the form a tile falls back to when its
 *                    flow range needs more than 160 window columns)
 *   "fuse_last1"     1 (default) / 0: batches <= 2 (with "defer_upsample"): stage 1's last Conv3D layer and the soft-argmin run in
 *                    one launch (24 x 2 x 4 tiles spanning D) and NO launch materialises pred1: stage 2's warp kernel evaluates
 *                    the taps it needs from the 1/8 map, stage 3's warp kernel writes pred1 beside pred2;
 *                    0 = k_conv3d_last + k_softargmin_upsample
 *   "fuse_ref_last"  -1 (default: batch 1 only) / 0 / 1: refinement2's last depthwise-separable block (dilation 1), the 32 -> 1
 *                    convolution and "+ pred3" in one launch (k_ref_dws_last: the block recomputed on the one-pixel ring the
 *                    convolution needs) instead of k_ref_dws + k_ref_last
 *   "device"         the HIP device the handle belongs to; settable only before lws_finalize / lws_reserve allocate
 * Unknown names and out-of-range values return LWS_ERR_INVALID.
 * hipGraph capture: lws_reserve first (nothing may allocate while capturing), then capture lws_forward on a non-default
 * stream; the library sees the capture (hipStreamIsCapturing), records its forks as capture-time events instead of binding
 * them to kernel completion signals, and times nothing (tests/test_gpu_parity.py::test_graph_capture_replays_the_forward). */
int lws_set_option(lws_handle h, const char *name, int value);
int lws_get_option(lws_handle h, const char *name, int *value);

/* ---- measurement hooks (bench.py) ------------------------------------------------------ */
/* Kernel classes timed by the built-in profiler. */
typedef enum {
    LWS_KC_VOLUME_SHIFT = 0,   /* k_volume_l1_shift                      */
    LWS_KC_VOLUME_WARP = 1,    /* k_volume_l1_warp                       */
    LWS_KC_CONV3D_FIRST = 2,   /* k_conv3d_first  (1 -> C3)              */
    LWS_KC_CONV3D_MID16 = 3,   /* k_conv3d_mid16 / k_conv3d_mid16x (C3 -> C3, C3 % 16 == 0, MFMA) */
    LWS_KC_CONV3D_MID8 = 4,    /* k_conv3d_mid8q / k_conv3d_mid8x (8 -> 8, MFMA) */
    LWS_KC_CONV3D_LAST = 5,    /* k_conv3d_last   (C3 -> 1, + skip)      */
    LWS_KC_SOFTARGMIN = 6,     /* k_softargmin                           */
    LWS_KC_UPSAMPLE = 7,       /* k_upsample_add                         */
    LWS_KC_FEATURE2D = 8,      /* k_conv2d_nchw (feature extractor)      */
    LWS_KC_REF_FIRST = 9,      /* k_ref_first (only with option "fuse_first" < 3) */
    LWS_KC_REF_DWS = 10,       /* k_ref_dws                              */
    LWS_KC_REF_CONV64 = 11,    /* k_ref_conv64                           */
    LWS_KC_REF_LAST = 12,      /* k_ref_last                             */
    LWS_KC_COUNT = 13
} lws_kernel_class;

/* class_mask != 0: every launch whose kernel class bit (1 << lws_kernel_class) is set is bracketed from now
 * on by a hipEvent pair recorded on the launch stream (records are dropped, never blocking, beyond 65536
 * launches); -1 selects all classes.  class_mask == 0: stop.  Either way the accumulated records are cleared. */
int lws_profile_enable(lws_handle h, int class_mask);
/* After lws_profile_enable: lws_forward records events on every `every_n`-th call only (1 = every call).  Timing a
 * kernel with its own begin / end events keeps the next dispatch from being queued behind it, so bracketing every
 * launch of a latency-bound step perturbs the step; sampling bounds that cost.  Reset to 1 by lws_profile_enable. */
int lws_profile_sample(lws_handle h, int every_n);
/* Synchronises the recorded events and returns, per kernel class, the summed device time in
 * milliseconds and the number of launches.  Both arrays have LWS_KC_COUNT entries. */
int lws_profile_read(lws_handle h, double *total_ms, int64_t *launches);
/* The individual launches of one kernel class, in launch order: ms_out[0 .. min(*count, capacity)) receives their
 * durations in milliseconds, *count the number of recorded launches (bench.py separates the stage-2 from the stage-3
 * launches of k_conv3d_mid8 with it). */
int lws_profile_read_class(lws_handle h, int kernel_class, float *ms_out, int capacity, int *count);
const char *lws_kernel_class_name(int kernel_class);
/* The clock the dominant kernel really runs at (ABI v8).  lws_clock_stamp(h, 1): from now on every k_conv3d_mid16 launch made
 * through this handle (the four 32 -> 32 Conv3D layers of stage 1) has its first 64 workgroups leave s_memtime (shader clock)
 * and s_memrealtime (100 MHz) of their first and last instruction in a handle-owned buffer (a scalar branch in the kernel,
 * nothing when off); the latest launch wins.  lws_clock_read synchronises the device and returns the median over those
 * workgroups of d s_memtime / d s_memrealtime x 100 MHz for the latest stamped launch.  bench.py stamps a few forwards right
 * after its timed region -- same queue depth, same mix of kernels -- and reports roofline.clock_ghz (per rank for N > 1): a box
 * or rank that holds a lower clock shows up there, not as an unexplained slower step.  LWS_ERR_STATE when stage 1's C3 is 8. */
int lws_clock_stamp(lws_handle h, int enable);
int lws_clock_read(lws_handle h, double *ghz);

/* ---- several forwards in flight (no counterpart in the reference: inference.py:105-109 is one thread, one stream) ---- */
/* A second handle for the same model on the same device: shares src's (read-only) parameter slab, owns its workspace,
 * streams and options.  src must outlive it and must not be re-finalized while clones exist; a clone refuses
 * lws_set_tensor / lws_finalize.  Use: one clone per host thread / stream. */
int lws_clone(lws_handle src, lws_handle *out);

/* A pool of `workers` host threads, each with its own clone of `model` and ONE HIP stream.  A batch-1 forward is a chain
 * of ~35 dependent launches (launch-latency-bound, ~345 us of host time to issue); the pool keeps `workers` of them in
 * flight so that they overlap on the device and their host cost runs in parallel.  Results are bit-identical to
 * lws_forward.  flags: 0, or LWS_POOL_SIDE_STREAMS to let every worker also use its per-handle side stream
 * (2 streams per worker; more streams than hardware queues makes throughput depend on the stream -> queue mapping).
 * `model` must outlive the pool. */
typedef struct lws_pool *lws_pool_handle;
#define LWS_POOL_SIDE_STREAMS 1
int lws_pool_create(lws_handle model, int workers, int flags, lws_pool_handle *out);
int lws_pool_destroy(lws_pool_handle p);            /* runs what is queued, then joins the workers */
int lws_pool_workers(lws_pool_handle p);
/* Pre-allocates every worker's workspace (waits for the jobs in flight first). */
int lws_pool_reserve(lws_pool_handle p, int B, int H, int W);
/* Queues one lws_forward(left, right -> pred_out[0..3]) and returns its ticket.  The job starts behind everything
 * already queued on `after_stream` (the stream that produces left / right; NULL = the default stream).  The buffers
 * must stay valid, and pred_out unread, until lws_pool_wait(ticket) has returned.  At most 4 x workers jobs are kept
 * in flight: a further submit first waits for the oldest one.  Thread-safe. */
int lws_pool_submit(lws_pool_handle p, const float *left, const float *right, int B, int H, int W,
                    float *const pred_out[4], void *after_stream, int64_t *ticket);
/* Blocks the calling thread until the job's outputs are complete in device memory; returns the job's status
 * (lws_last_error() then holds the worker's message).  A ticket may be waited for any number of times while its slot is
 * live (the 4 x workers most recent tickets); for an older, recycled ticket the job is complete by construction and its own
 * status is gone: the call then returns the pool's STICKY status -- the status and message of the first job that failed since
 * the pool was created (or since lws_pool_clear_error) -- unless the ticket is older than the SMALLEST failed ticket (those
 * jobs ran to completion: LWS_OK), so a failed forward is never reported as success once its slot has been reused.  The same sticky
 * status is returned by lws_pool_wait_all and refuses further lws_pool_submit calls until it is cleared.
 * Options are FROZEN at lws_pool_create: the workers are clones taken then; a later lws_set_option on `model` does not reach
 * them (create a new pool after changing options). */
int lws_pool_wait(lws_pool_handle p, int64_t ticket);
int lws_pool_wait_all(lws_pool_handle p);
int lws_pool_clear_error(lws_pool_handle p);          /* ABI v6 */
/* Kernel timing inside the pool (ABI v7): lws_profile_enable(class_mask) + lws_profile_sample(every_n) on every worker's
 * clone, and the per-class sums over the workers (arrays of LWS_KC_COUNT entries, as lws_profile_read).  Call both with
 * nothing in flight (after lws_pool_wait_all).  bench.py prices `pipelined` with it: k_conv3d_mid16 as it runs beside the
 * other workers' kernels. */
int lws_pool_profile_enable(lws_pool_handle p, int class_mask, int every_n);
int lws_pool_profile_read(lws_pool_handle p, double *total_ms, int64_t *launches);

#ifdef __cplusplus
}
#endif
#endif /* LWSNET_HIP_H */
