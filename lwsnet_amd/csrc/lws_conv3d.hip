// 3D cost-volume filtering: 6 x (BatchNorm3D(eval) -> ReLU -> Conv3D 3x3x3, stride 1, pad 1, no bias)
// plus the outer skip   /root/reference/models/submodules.py:190-221, models/models.py:136-138.
//
// Data layout: the stack's internal activations are channels-last [B, D, h, w, C3] and hold the
// POST-activation values relu(bn_{j+1}(conv_j(.))) -- the BatchNorm+ReLU of layer j+1 is the epilogue
// of layer j, so the zero padding of the convolution is a literal zero, exactly as in the reference
// (padding is applied after BN+ReLU there too).  BN is y = fmaf(x, s, t) with (s, t) folded on the host.
//
// Arithmetic contract (identical in oracle/lws_oracle.c, checked bit for bit by tests/test_gpu_parity.py):
// every output is ONE float32 fma chain from 0 over taps (kd,kh,kw) outer, input channel inner, ascending.
// v_mfma_f32_16x16x4_f32 computes exactly such a k-ordered chain, so the MFMA kernels keep that contract.
//
//   k_conv3d_first8 / k_conv3d_first16   cin = 1 -> C3   fp32 MFMA (K = 27 taps)
//   k_conv3d_mid16  C3 % 16 == 0     fp32 MFMA implicit GEMM, M = cout tile, N = 16 voxels along x
//   k_conv3d_mid8q  C3 == 8          v_mfma_f32_4x4x1_16B_f32 with A-block broadcast: 4 couts x 64 voxels, no padding
//   k_conv3d_last   C3 -> 1 + skip   VALU (K = 27*C3)
#include <hip/hip_ext.h>
#include <hip/hip_fp16.h>
#include <stdlib.h>

#include "lws_common.h"
#include "lws_device_math.h"

namespace lws {


LWS_DEFINE_STAMPS(conv3d)

__device__ __forceinline__ float bn_relu(float x, float s, float t) { return fmaxf(fmaf(x, s, t), 0.0f); }

// tile index (already XCD-contiguous, xcd_tile) -> tile coordinates: d fastest, then x, then y -- the tiles that share halo
// planes along d (2 of the 5 planes a 3-deep tile reads) and along y are then neighbours in an XCD's run and co-resident, so
// their re-reads hit that XCD's L2 instead of the fabric.  The same map in every layer of a stack: a consumer tile finds its
// producer's output in the same L2.  Speed only.
__device__ __forceinline__ void tile_coords(int tile, int tiles_x, int tiles_d, int &tx, int &ty, int &td)
{
    td = tile % tiles_d;
    tile /= tiles_d;
    tx = tile % tiles_x;
    ty = tile / tiles_x;
}

// compile-time component select (j is always a constant after unrolling)
__device__ __forceinline__ float f4(const float4 &v, int j) { return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; }

// =============================================================================================
// Middle layers, C3 a multiple of 16 (stage 1: C3 = 32).
//
// Implicit GEMM  Out^T[cout, voxel] = sum_{tap, cin} W[cout, (tap,cin)] * X[(tap,cin), voxel]
// on v_mfma_f32_16x16x4_f32:  A (16 x 4) = weights of 16 output channels, B (4 x 16) = activations of 16
// consecutive voxels along x, K = 4 input channels of one tap per instruction.
//
// Workgroup = 4 waves = an output tile of TD x TY rows of 16 voxels.  The input halo tile
// (TD+2)(TY+2)(18) voxels is staged once in LDS, channels-last with a padded voxel stride of C3+4
// dwords (ds_read_b128 of 16 neighbouring voxels then spreads over the banks).  Inside every
// 16-channel group the LDS image is 4x4-transposed: dword 4g+j of the group holds channel 4j+g, so the
// ONE ds_read_b128 of lane (n, g) yields its B operand for the 4 MFMAs j = 0..3 of the group and MFMA j
// sums channels 4j..4j+3 in order -- the natural ascending-channel chain.  The transpose is free: the
// staging thread loads the 16 channels of a voxel as 4 float4 and writes 4 float4.
//
// Weights are pre-packed on the host in A-fragment order [tap][q][mt][lane][j] (1 KiB per wave load) and
// streamed from L2 straight into VGPRs, one (tap, 16-channel group) step ahead of the MFMAs; every wave
// of every workgroup reads the same 27*C3*C3*4 bytes, so they stay L2/L1-resident.
// Each wave owns TD*TY/4 rows x all C3/16 output-channel tiles: (TD*TY/4)*(C3/16) independent
// accumulator chains, which covers the 40-cycle dependent-issue latency of the 32-cycle MFMA.
// =============================================================================================
constexpr bool MID16_INTERLEAVE = true;   // see the inner loop of k_conv3d_mid16

template <int C3, int TD, int TY, int WR, int WM>
struct Mid16Cfg {
    static constexpr int MT = C3 / 16;          // output-channel tiles of the layer
    static constexpr int Q = C3 / 16;           // input-channel groups per tap
    static constexpr int NW = WR * WM;          // waves per workgroup
    static constexpr int NT = 64 * NW;
    static constexpr int ROWS = TD * TY;
    static constexpr int RW = ROWS / WR;        // rows per wave
    static constexpr int MTW = MT / WM;         // output-channel tiles per wave
    static constexpr int HD = TD + 2, HY = TY + 2, HX = 18;
    static constexpr int VS = C3 + 4;           // LDS voxel stride in dwords
    static constexpr int NVOX = HD * HY * HX;
    static constexpr int ITEMS = NVOX * Q;      // staging items: (voxel, 16-channel group)
    static constexpr int SITER = (ITEMS + NT - 1) / NT;
    static constexpr int LDS_BYTES = NVOX * VS * 4;
    static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");
    static_assert(ROWS % WR == 0 && MT % WM == 0, "tile must split evenly over the waves");
    static_assert(RW * MTW >= 2 || NW == 8, "need >= 2 independent accumulator chains per SIMD");
};

template <int C3, int TD, int TY, int WR, int WM>
__global__ __launch_bounds__(64 * WR * WM) void k_conv3d_mid16(const float *__restrict__ in,     // [B,D,h,w,C3]
                                                              const float4 *__restrict__ wpk,   // packed A fragments
                                                              const float *__restrict__ bn_s,   // next layer BN [C3]
                                                              const float *__restrict__ bn_t,
                                                              float *__restrict__ out, int D, int h, int w,
                                                              int tiles_x, int tiles_y, int wt, int tiles_d,
                                                              unsigned long long *__restrict__ clk)
{
    // clk != nullptr (lws_clock_stamp only; kernel-uniform, so a scalar branch): the first 64 workgroups leave the shader-clock
    // counter (s_memtime) and the 100 MHz wall clock (s_memrealtime) of their first and last instruction -- the clock this
    // kernel really ran at is d s_memtime / d s_memrealtime x 100 MHz (bench.py: roofline.clock_ghz)
    unsigned long long clk_c0 = 0, clk_r0 = 0;
    if (clk != nullptr)
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(clk_c0), "=s"(clk_r0)::"memory");
    using Cfg = Mid16Cfg<C3, TD, TY, WR, WM>;
    constexpr int MT = Cfg::MT, Q = Cfg::Q, RW = Cfg::RW, MTW = Cfg::MTW, HY = Cfg::HY, HX = Cfg::HX, VS = Cfg::VS;
    constexpr int NT = Cfg::NT, SITER = Cfg::SITER;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WM, wm = wave % WM;     // this wave's row group / output-channel group
    const int n = lane & 15, g = lane >> 4;
    int tx, ty, td;
    tile_coords(xcd_tile(blockIdx.x, gridDim.x), tiles_x, tiles_d, tx, ty, td);
    const int b = blockIdx.y;
    const int x0 = tx * 16, y0 = ty * TY, d0 = td * TD;
    const float *inb = in + (int64_t)b * D * h * w * C3;
    LWS_STAMPK(1, 0);

    // ---- stage the halo tile: one item = (voxel, 16-channel group) = 64 contiguous bytes.  All global loads
    //      of a thread are issued first (SITER x 4 float4 in flight), then transposed 4x4 and written to LDS --
    //      in two phases: the taps kd = 0 only read halo slices [0, TD), i.e. the items of iterations i < P1;
    //      the rest (iterations P1..SITER-1, still in flight) is written after the kd = 0 taps, under their MFMAs.
    constexpr int P1 = (TD * HY * HX * Q + NT - 1) / NT < SITER ? (TD * HY * HX * Q + NT - 1) / NT : SITER;
    float4 c[SITER][4];
    bool okv[SITER];
#pragma unroll
    for (int i = 0; i < SITER; ++i) {
        const int it = tid + i * NT;
        const int q = it % Q, v = it / Q;
        const int hx = v % HX, t2 = v / HX, hy = t2 % HY, hd = t2 / HY;
        const int gd = d0 + hd - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
        okv[i] = it < Cfg::ITEMS && gd >= 0 && gd < D && gy >= 0 && gy < h && gx >= 0 && gx < w;
        // unconditional loads (padding / surplus items read voxel 0 and are zeroed below): no branch per load
        const float4 *src = reinterpret_cast<const float4 *>(inb + (okv[i] ? (((int64_t)gd * h + gy) * w + gx) * C3 + q * 16 : 0));
        c[i][0] = src[0];
        c[i][1] = src[1];
        c[i][2] = src[2];
        c[i][3] = src[3];
    }
    auto stage_write = [&](int i) {
        const int it = tid + i * NT;
        if (it < Cfg::ITEMS) {
            const int q = it % Q, v = it / Q;
            float4 *dst = reinterpret_cast<float4 *>(lds + v * VS + q * 16);
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 c0 = okv[i] ? c[i][0] : z, c1 = okv[i] ? c[i][1] : z, c2 = okv[i] ? c[i][2] : z, c3 = okv[i] ? c[i][3] : z;
            dst[0] = make_float4(c0.x, c1.x, c2.x, c3.x);
            dst[1] = make_float4(c0.y, c1.y, c2.y, c3.y);
            dst[2] = make_float4(c0.z, c1.z, c2.z, c3.z);
            dst[3] = make_float4(c0.w, c1.w, c2.w, c3.w);
        }
    };
    // next layer's BatchNorm of this lane's output channels: loaded here, behind the halo loads, so that the epilogue
    // does not start with an exposed L2 round trip
    float4 es[MTW], et[MTW];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
        es[mt] = *reinterpret_cast<const float4 *>(bn_s + (wm * MTW + mt) * 16 + 4 * g);
        et[mt] = *reinterpret_cast<const float4 *>(bn_t + (wm * MTW + mt) * 16 + 4 * g);
    }
#pragma unroll
    for (int i = 0; i < P1; ++i) stage_write(i);

    // weights of this wave: [tap][q][mt][lane] float4, mt in [wm*MTW, (wm+1)*MTW); the packed array holds two
    // extra all-zero taps so that the two-taps-ahead prefetch never needs a bounds check
    const float4 *wp = wpk + (wm * MTW) * 64 + lane;
    float4 wbuf[3][Q][MTW];                         // ring over kw: tap kw of the current (kd,kh) lives in wbuf[kw]
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
            wbuf[0][q][mt] = wp[(q * MT + mt) * 64];
            wbuf[1][q][mt] = wp[((Q + q) * MT + mt) * 64];     // the stream runs two taps ahead of the MFMAs
        }
    __syncthreads();
    LWS_STAMPK(1, 1);

    floatx4 acc[RW][MTW];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) acc[r][mt] = (floatx4){0.f, 0.f, 0.f, 0.f};

    // per-row LDS base of this lane: voxel (rd, ry, n) of the halo tile, channel quad g
    const float *rptr[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int row = wr * RW + r;
        const int rd = row / TY, ry = row % TY;
        rptr[r] = lds + ((rd * HY + ry) * HX + n) * VS + 4 * g;
    }

    // Software pipeline without copies: 3 taps (kw) x Q channel groups = 3Q steps per (kd,kh) iteration; activation
    // fragments ping-pong between bbuf[0/1] by step parity (3Q even), weights rotate through wbuf[kw].  The prefetch
    // of step s+1 / tap t+1 is issued in small fenced slices BETWEEN the four 6-MFMA groups of step s, so that every
    // ds_read / global_load issues under a running MFMA instead of stalling the matrix pipe at step boundaries.
    // (Q == 1, i.e. C3 == 16, has an odd number of steps per iteration: it copies bbuf[1] -> bbuf[0] instead.)
    float4 bbuf[2][RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) bbuf[0][r] = *reinterpret_cast<const float4 *>(rptr[r]);

#pragma unroll 1
    for (int kdh = 0; kdh < 9; ++kdh) {
        const int kd = kdh / 3, kh = kdh - kd * 3;
        const int base = (kd * HY + kh) * HX * VS;
        if (P1 < SITER && kdh == 3) {
            // phase 2 of the staging: halo slices TD, TD+1 (first read by kd = 1); the fragment prefetched at the end
            // of iteration 2 may predate these writes, so it is read again
#pragma unroll
            for (int i = P1; i < SITER; ++i) stage_write(i);
            __syncthreads();
#pragma unroll
            for (int r = 0; r < RW; ++r) bbuf[0][r] = *reinterpret_cast<const float4 *>(rptr[r] + base);
        }
        // first fragment offset of the NEXT (kd,kh) iteration (clamped for the last one: the prefetch is unused)
        const int kn = kdh < 8 ? kdh + 1 : 8;
        const int base_n = ((kn / 3) * HY + (kn % 3)) * HX * VS;
        const float4 *wtap = wp + (size_t)(kdh * 3) * Q * MT * 64;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const int s = kw * Q + q;                      // step inside the iteration (compile time)
                const int cur = (Q % 2 == 0) ? (s & 1) : 0;
                float4 *bc = bbuf[cur], *bn = bbuf[cur ^ 1];
                // where the next step reads its activation fragments
                const bool same_tap = q + 1 < Q;
                const int off_n = same_tap ? base + kw * VS + (q + 1) * 16 : (kw < 2 ? base + (kw + 1) * VS : base_n);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int r = 0; r < RW; ++r)
#pragma unroll
                        for (int mt = 0; mt < MTW; ++mt)
                            acc[r][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4(wbuf[kw][q][mt], j), f4(bc[r], j), acc[r][mt], 0, 0, 0);
                    if (!MID16_INTERLEAVE) __builtin_amdgcn_sched_barrier(0);
                    // prefetch slice j: one activation fragment per slice, then (on the first channel group of a tap)
                    // the weights of the next tap
                    if (j < RW) bn[j] = *reinterpret_cast<const float4 *>(rptr[j] + off_n);
                    if (j == 3) {
#pragma unroll
                        for (int r = 4; r < RW; ++r) bn[r] = *reinterpret_cast<const float4 *>(rptr[r] + off_n);
                    }
                    if (q == 0 && j >= 1) {
                        // tap t+2 -> wbuf[(kw+2)%3] (the slot of tap t-1): two taps = 4 steps = ~3k cycles of lead over
                        // the L2 latency; Q*MTW loads spread over slices 1..3
                        constexpr int NL = Q * MTW;
#pragma unroll
                        for (int l = 0; l < NL; ++l)
                            if (l % 3 == j - 1) {
                                const int q2 = l / MTW, mt2 = l % MTW;
                                wbuf[(kw + 2) % 3][q2][mt2] = wtap[((kw + 2) * Q + q2) * MT * 64 + mt2 * 64];
                            }
                    }
                    if (MID16_INTERLEAVE) {
                        // one prefetch instruction per MFMA gap instead of all of them behind the group: an MFMA leaves
                        // ~24 issue cycles free, a ds_read plus two global loads plus their address adds need ~50
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // VALU (address)
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (Q % 2 != 0) {
#pragma unroll
                    for (int r = 0; r < RW; ++r) bbuf[0][r] = bbuf[1][r];
                }
            }
        }
    }

    LWS_STAMPK(1, 2);
    // ---- epilogue: D[i][j]: row i = 4*(lane>>4) + reg = output channel in the tile, col j = lane&15 = voxel.
    //      Apply the NEXT layer's BatchNorm + ReLU and store 4 consecutive channels of one voxel (16 B).
    float *outb = out + (int64_t)b * D * h * w * C3;
    const int gx = x0 + n;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int row = wr * RW + r;
        const int gd = d0 + row / TY, gy = y0 + row % TY;
        if (gd < D && gy < h && gx < w) {
            float *o = outb + (((int64_t)gd * h + gy) * w + gx) * C3;
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) {
                const int cb = (wm * MTW + mt) * 16 + 4 * g;
                const float4 s = es[mt], t = et[mt];
                float4 v;
                v.x = bn_relu(acc[r][mt][0], s.x, t.x);
                v.y = bn_relu(acc[r][mt][1], s.y, t.y);
                v.z = bn_relu(acc[r][mt][2], s.z, t.z);
                v.w = bn_relu(acc[r][mt][3], s.w, t.w);
                store_act4(o + cb, v, wt);
            }
        }
    }
    LWS_STAMPK(1, 3);
    if (clk != nullptr) {
        unsigned long long c1, r1;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
        const unsigned bid = blockIdx.x + gridDim.x * blockIdx.y;
        if (threadIdx.x == 0 && bid < 64) {
            clk[4 * bid + 0] = clk_c0;
            clk[4 * bid + 1] = clk_r0;
            clk[4 * bid + 2] = c1;
            clk[4 * bid + 3] = r1;
        }
    }
}

// =============================================================================================
// Middle layers, C3 == 32, split-bf16 form (k_conv3d_mid16x; option "split_bf16" bit 0, NOT the default and never what
// bench.py's headline measures: it is not bit-exact against the oracle chain).
//
// The f32-input MFMA runs at the float32 vector rate (157 TF); the bf16 MFMA at 16x that.  Every float32 operand is split
// into three bf16 values x = hi + mid + lo (8 + 8 + 8 mantissa bits) and a product a*b is replaced by the six cross terms of
// total order <= 2 -- lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi, each exact in float32 -- accumulated in float32 by
// v_mfma_f32_16x16x32_bf16 (K = the 32 input channels of one tap per instruction).  Six 16-cycle instructions replace eight
// 32-cycle ones: 2.5-2.6x the issue rate (tools/micro/split_bf16.hip: 107-114 against 277-285 cycles per (tap, accumulator)),
// at float32-level accuracy: on a K = 864 contraction of ReLU'd activations with Kaiming weights the result sits 3.2e-6 (max) /
// 5.3e-7 (mean) from float64, the float32 fma chain 2.7e-6 / 4.9e-7 (same file).  What it cannot be is bit-identical to that
// chain, so it is gated by the float64 noise-floor tests (tests/test_gpu_parity.py::test_split_bf16_*) instead of the oracle.
//
// Same tile, same interface and same epilogue as k_conv3d_mid16 (float32 channels-last in and out): the split happens
// while the halo tile is staged into LDS ([voxel][variant hi/mid/lo][32 channels] bf16, 208-byte voxel stride), the
// weights are pre-split on the host into A fragments [tap][mt][variant][lane][8] and streamed one tap ahead.
// =============================================================================================
template <int TD, int TY>
struct Mid16xCfg {
    static constexpr int C3 = 32, MT = 2, NW = 4, NT = 256;
    static constexpr int ROWS = TD * TY, RW = ROWS / NW;
    static constexpr int HD = TD + 2, HY = TY + 2, HX = 18;
    static constexpr int VSB = 208;                  // LDS voxel stride in bytes: 3 variants x 64 B + 16 B (ds_read_b128 of 16 neighbouring voxels spreads over the banks)
    static constexpr int NVOX = HD * HY * HX;
    static constexpr int ITEMS = NVOX * 2;           // (voxel, 16-channel group)
    static constexpr int SITER = (ITEMS + NT - 1) / NT;
    static constexpr int LDS_BYTES = NVOX * VSB;
    static constexpr int TAP_U4 = MT * 3 * 64;       // uint4 (8 bf16) per tap in the packed weights
    static_assert(ROWS % NW == 0, "rows must split evenly over the waves");
};

template <int TD, int TY>
__global__ __launch_bounds__(256) void k_conv3d_mid16x(const float *__restrict__ in,      // [B,D,h,w,32]
                                                       const uint4 *__restrict__ wpk,     // [29][2][3][64] x 8 bf16
                                                       const float *__restrict__ bn_s,    // next layer BN [32]
                                                       const float *__restrict__ bn_t,
                                                       float *__restrict__ out, int D, int h, int w,
                                                       int tiles_x, int tiles_y, int tiles_d)
{
    using Cfg = Mid16xCfg<TD, TY>;
    constexpr int C3 = 32, MT = 2, RW = Cfg::RW, HY = Cfg::HY, HX = Cfg::HX, VSB = Cfg::VSB, NT = Cfg::NT, SITER = Cfg::SITER;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    unsigned char *ldsb = reinterpret_cast<unsigned char *>(lds);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    int tx, ty, td;
    tile_coords(xcd_tile(blockIdx.x, gridDim.x), tiles_x, tiles_d, tx, ty, td);
    const int b = blockIdx.y;
    const int x0 = tx * 16, y0 = ty * TY, d0 = td * TD;
    const float *inb = in + (int64_t)b * D * h * w * C3;
    LWS_STAMPK(19, 0);

    // ---- stage: item = (voxel, 16-channel group) = 64 B of float32 -> 3 x 32 B of bf16 (hi / mid / lo)
    float4 c[SITER][4];
    bool okv[SITER];
#pragma unroll
    for (int i = 0; i < SITER; ++i) {
        const int it = tid + i * NT;
        const int q = it & 1, v = it >> 1;
        const int hx = v % HX, t2 = v / HX, hy = t2 % HY, hd = t2 / HY;
        const int gd = d0 + hd - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
        okv[i] = it < Cfg::ITEMS && gd >= 0 && gd < D && gy >= 0 && gy < h && gx >= 0 && gx < w;
        const float4 *src = reinterpret_cast<const float4 *>(inb + (okv[i] ? (((int64_t)gd * h + gy) * w + gx) * C3 + q * 16 : 0));
        c[i][0] = src[0];
        c[i][1] = src[1];
        c[i][2] = src[2];
        c[i][3] = src[3];
    }
    // next layer's BatchNorm of this lane's output channels, and the first tap's weights, behind the halo loads
    float4 es[MT], et[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        es[mt] = *reinterpret_cast<const float4 *>(bn_s + mt * 16 + 4 * g);
        et[mt] = *reinterpret_cast<const float4 *>(bn_t + mt * 16 + 4 * g);
    }
    const uint4 *wp = wpk + lane;
    uint4 wa[3][MT][3];                             // ring over kw: tap (kd,kh,kw) lives in slot kw; the stream runs TWO taps ahead
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            wa[0][mt][t] = wp[(mt * 3 + t) * 64];
            wa[1][mt][t] = wp[Cfg::TAP_U4 + (mt * 3 + t) * 64];
        }
    // two phases as in k_conv3d_mid16: the kd = 0 taps only read halo planes [0, TD), i.e. the items of iterations i < P1
    constexpr int P1 = (TD * HY * HX * 2 + NT - 1) / NT < SITER ? (TD * HY * HX * 2 + NT - 1) / NT : SITER;
    auto stage_write = [&](int i) {
        const int it = tid + i * NT;
        if (it < Cfg::ITEMS) {
            const int q = it & 1, v = it >> 1;
            uint32_t pk[3][8];                      // per variant: 16 bf16 = 8 dwords
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float xs[4] = {c[i][k].x, c[i][k].y, c[i][k].z, c[i][k].w};
#pragma unroll
                for (int e = 0; e < 4; e += 2)
                    split_bf16x3_pair(okv[i] ? xs[e] : 0.f, okv[i] ? xs[e + 1] : 0.f, pk[0][2 * k + e / 2], pk[1][2 * k + e / 2],
                                      pk[2][2 * k + e / 2]);
            }
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                uint4 *dst = reinterpret_cast<uint4 *>(ldsb + v * VSB + t * 64 + q * 32);
                dst[0] = make_uint4(pk[t][0], pk[t][1], pk[t][2], pk[t][3]);
                dst[1] = make_uint4(pk[t][4], pk[t][5], pk[t][6], pk[t][7]);
            }
        }
    };
#pragma unroll
    for (int i = 0; i < P1; ++i) stage_write(i);
    __syncthreads();
    LWS_STAMPK(19, 1);

    floatx4 acc[RW][MT];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[r][mt] = (floatx4){0.f, 0.f, 0.f, 0.f};
    // per-row LDS base of this lane: voxel (rd, ry, n) of the halo tile, channels 8 g .. 8 g + 7 of each variant
    const unsigned char *rptr[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int row = wave * RW + r;
        rptr[r] = ldsb + ((row / TY * HY + row % TY) * HX + n) * VSB + g * 16;
    }
    // activation fragments of one tap: RW rows x (hi, mid, lo), double-buffered by tap parity (the tap loop is fully unrolled,
    // so every buffer index and LDS offset is a compile-time constant)
    uint4 bb[2][RW][3];
    auto load_b_row = [&](uint4 (&dst)[RW][3], int r, int off) {
#pragma unroll
        for (int t = 0; t < 3; ++t) dst[r][t] = *reinterpret_cast<const uint4 *>(rptr[r] + off + t * 64);
    };
#pragma unroll
    for (int r = 0; r < RW; ++r) load_b_row(bb[0], r, 0);

    // One tap = 6 terms x (RW x MT = 6 accumulators) = 36 MFMAs of 16 cycles, of which each holds the vector issue port for 8.
    // The 15 prefetches of a tap ride in those gaps ONE at a time, fenced behind every second MFMA: the next tap's fragments
    // (9 ds_read_b128), then the weights of tap + 2 into the ring slot of tap - 1 (6 global loads).  Measured r03
    // (tools/stamps.py mid16x, cycles per MFMA over the MFMA phase): all prefetches in one block ahead of the tap's MFMAs 22.6,
    // one slice of 3 behind each term group 20.0, one at a time 19.0 -- and 15.9 / 14.8 / 15.1 us per launch at B = 1: the last
    // step came back as a lower clock (1.99 -> 1.95 GHz in-kernel), not as time (MI355X_MICROARCH.md, DVFS give-back).
    static_assert(2 * (3 * RW + 3 * MT) <= 6 * RW * MT, "one prefetch behind every second MFMA");
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
        const int kw = tap % 3, kdh = tap / 3;
        const int cb = tap & 1, nb = cb ^ 1;
        const int tn = tap < 26 ? tap + 1 : 26;                          // (clamped: the last prefetch is unused)
        const int off_n = (((tn / 9) * HY + (tn / 3) % 3) * HX + tn % 3) * VSB;
        if (P1 < SITER && tap == 9) {
            // phase 2 of the staging (halo planes TD, TD + 1, first read by kd = 1); the fragments prefetched during tap 8
            // may predate these writes: read again
#pragma unroll
            for (int i = P1; i < SITER; ++i) stage_write(i);
            __syncthreads();
#pragma unroll
            for (int r = 0; r < RW; ++r) load_b_row(bb[cb], r, ((kdh / 3) * HY + kdh % 3) * HX * VSB);
        }
        // prefetch op k of this tap: k < 3 RW = one ds_read_b128 (row k / 3, variant k % 3) of the next tap's fragments, then
        // one global load each of the weights of tap + 2 (cout tile, variant) into the ring slot of tap - 1
        auto prefetch = [&](int k) {
            if (k < 3 * RW)
                bb[nb][k / 3][k % 3] = *reinterpret_cast<const uint4 *>(rptr[k / 3] + off_n + (k % 3) * 64);
            else if (k < 3 * RW + 3 * MT)
                wa[(kw + 2) % 3][(k - 3 * RW) / 3][(k - 3 * RW) % 3] = wp[(size_t)(tap + 2) * Cfg::TAP_U4 + (k - 3 * RW) * 64];
        };
#pragma unroll
        for (int T = 0; T < 6; ++T)
#pragma unroll
            for (int r = 0; r < RW; ++r)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    acc[r][mt] = mfma_split_bf16_term(acc[r][mt], wa[kw][mt], bb[cb][r], T);
                    const int m = (T * RW + r) * MT + mt;               // one prefetch behind every second MFMA
                    if (m % 2 == 0 && m / 2 < 3 * RW + 3 * MT) {
                        __builtin_amdgcn_sched_barrier(0);
                        prefetch(m / 2);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
    }

    LWS_STAMPK(19, 2);
    // ---- epilogue (as k_conv3d_mid16): row i = 4 (lane >> 4) + reg = output channel in the tile, col = lane & 15 = voxel
    float *outb = out + (int64_t)b * D * h * w * C3;
    const int gx = x0 + n;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int row = wave * RW + r;
        const int gd = d0 + row / TY, gy = y0 + row % TY;
        if (gd < D && gy < h && gx < w) {
            float *o = outb + (((int64_t)gd * h + gy) * w + gx) * C3;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const float4 s = es[mt], t = et[mt];
                float4 v;
                v.x = bn_relu(acc[r][mt][0], s.x, t.x);
                v.y = bn_relu(acc[r][mt][1], s.y, t.y);
                v.z = bn_relu(acc[r][mt][2], s.z, t.z);
                v.w = bn_relu(acc[r][mt][3], s.w, t.w);
                store_act4(o + mt * 16 + 4 * g, v, 0);
            }
        }
    }
    LWS_STAMPK(19, 3);
}

// =============================================================================================
// Middle layers, C3 == 8, split-bf16 form (k_conv3d_mid8x; option "split_bf16" bit 1, NOT the default and never what bench.py's
// headline measures: not bit-exact against the oracle chain, see k_conv3d_mid16x).
// A parity-row tile -- MFMA row i = 8 xpar + cout, column n = voxel pair, output x = x0 + 2 n + xpar (both parities read the
// same activations, so all 16 rows of the 8-channel layer work; the scheme k_conv3d_first8 uses with one input channel) -- with
// K = 32 = 4 x-offsets t x 8 input channels: ONE v_mfma_f32_16x16x32_bf16 contracts a whole (kd, kh) row of taps, lane
// (n, g) supplying the 8 channels of voxel x0 + 2 n + g - 1 (one ds_read_b128 per variant) against
// W[cout][cin][kd][kh][g - xpar] (zero where g - xpar is outside 0..2).  9 steps x 6 cross products = 54 MFMAs of 16 cycles
// per row of 32 outputs, against 216 x 10 cycles on the 4x4x1 form.  LDS: three variant planes [voxel][8 x bf16] with the
// 16-byte voxel slots XOR-swizzled by bit 4 of the voxel index, so that the 16 lanes of a ds_read_b128 (voxels c, c + 2, ...,
// c + 30) cover 16 different slots mod 256 B for every c.  Weights pre-split on the host ([step][variant][lane][8]) and
// streamed two steps ahead; staging in two phases.
// =============================================================================================
template <int TD, int TY>
struct Mid8xCfg {
    static constexpr int NW = 4, NT = 256;
    static constexpr int ROWS = TD * TY, RW = ROWS / NW;
    static constexpr int HD = TD + 2, HY = TY + 2, HX = 34;
    static constexpr int NVOX = HD * HY * HX;
    static constexpr int PLANE = ((NVOX + 15) & ~15) * 16;      // bytes per variant plane (whole 16-voxel swizzle groups)
    static constexpr int LDS_BYTES = 3 * PLANE;
    static constexpr int SITER = (NVOX + NT - 1) / NT;           // item = voxel (8 channels, 32 B of float32)
    static constexpr int STEP_U4 = 3 * 64;                       // uint4 per (kd, kh) step of the packed weights
    static_assert(ROWS % NW == 0, "rows must split evenly over the waves");
};
constexpr size_t MID8X_PACK_FLOATS = (size_t)11 * 3 * 64 * 4;   // 9 steps + two all-zero steps

__device__ __forceinline__ int mid8x_slot(int v) { return (v ^ ((v >> 4) & 1)) * 16; }

template <int TD, int TY>
__global__ __launch_bounds__(256) void k_conv3d_mid8x(const float *__restrict__ in,      // [B,D,h,w,8]
                                                      const uint4 *__restrict__ wpk,     // [11][3][64] x 8 bf16
                                                      const float *__restrict__ bn_s,    // next layer BN [8]
                                                      const float *__restrict__ bn_t,
                                                      float *__restrict__ out, int D, int h, int w,
                                                      int tiles_x, int tiles_y, int tiles_d)
{
    using Cfg = Mid8xCfg<TD, TY>;
    constexpr int RW = Cfg::RW, HY = Cfg::HY, HX = Cfg::HX, NT = Cfg::NT, SITER = Cfg::SITER, PLANE = Cfg::PLANE;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    unsigned char *ldsb = reinterpret_cast<unsigned char *>(lds);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    int tx, ty, td;
    tile_coords(xcd_tile(blockIdx.x, gridDim.x), tiles_x, tiles_d, tx, ty, td);
    const int b = blockIdx.y;
    const int x0 = tx * 32, y0 = ty * TY, d0 = td * TD;
    const float *inb = in + (int64_t)b * D * h * w * 8;
    LWS_STAMPK(20, 0);

    float4 c[SITER][2];
    bool okv[SITER];
#pragma unroll
    for (int i = 0; i < SITER; ++i) {
        const int v = tid + i * NT;
        const int hx = v % HX, t2 = v / HX, hy = t2 % HY, hd = t2 / HY;
        const int gd = d0 + hd - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
        okv[i] = v < Cfg::NVOX && gd >= 0 && gd < D && gy >= 0 && gy < h && gx >= 0 && gx < w;
        const float4 *src = reinterpret_cast<const float4 *>(inb + (okv[i] ? (((int64_t)gd * h + gy) * w + gx) * 8 : 0));
        c[i][0] = src[0];
        c[i][1] = src[1];
    }
    const int xpar = g >> 1, cb8 = 4 * (g & 1);
    const float4 es8 = *reinterpret_cast<const float4 *>(bn_s + cb8);
    const float4 et8 = *reinterpret_cast<const float4 *>(bn_t + cb8);
    const uint4 *wp = wpk + lane;
    uint4 wa[3][3];                                 // ring over kh: step (kd, kh) lives in slot kh; the stream runs TWO steps ahead
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        wa[0][t] = wp[t * 64];
        wa[1][t] = wp[Cfg::STEP_U4 + t * 64];
    }
    constexpr int P1 = (TD * HY * HX + NT - 1) / NT < SITER ? (TD * HY * HX + NT - 1) / NT : SITER;
    auto stage_write = [&](int i) {
        const int v = tid + i * NT;
        if (v < Cfg::NVOX) {
            uint32_t pk[3][4];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float xs[4] = {c[i][k].x, c[i][k].y, c[i][k].z, c[i][k].w};
#pragma unroll
                for (int e = 0; e < 4; e += 2)
                    split_bf16x3_pair(okv[i] ? xs[e] : 0.f, okv[i] ? xs[e + 1] : 0.f, pk[0][2 * k + e / 2], pk[1][2 * k + e / 2],
                                      pk[2][2 * k + e / 2]);
            }
            unsigned char *dst = ldsb + mid8x_slot(v);
#pragma unroll
            for (int t = 0; t < 3; ++t)
                *reinterpret_cast<uint4 *>(dst + t * PLANE) = make_uint4(pk[t][0], pk[t][1], pk[t][2], pk[t][3]);
        }
    };
#pragma unroll
    for (int i = 0; i < P1; ++i) stage_write(i);
    __syncthreads();
    LWS_STAMPK(20, 1);

    floatx4 acc[RW];
    int rbase[RW];                                  // halo voxel (rd, ry, 2 n + g) of this lane's row
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        acc[r] = (floatx4){0.f, 0.f, 0.f, 0.f};
        const int row = wave * RW + r;
        rbase[r] = (row / TY * HY + row % TY) * HX + 2 * n + g;
    }
    uint4 bb[2][RW][3];                             // double-buffered by step parity; the step loop is fully unrolled
    auto load_b_row = [&](uint4 (&dst)[RW][3], int r, int off) {
        const unsigned char *p = ldsb + mid8x_slot(rbase[r] + off);
#pragma unroll
        for (int t = 0; t < 3; ++t) dst[r][t] = *reinterpret_cast<const uint4 *>(p + t * PLANE);
    };
#pragma unroll
    for (int r = 0; r < RW; ++r) load_b_row(bb[0], r, 0);

    // One step = 6 terms x RW accumulators; the prefetches ride in the MFMA gaps one at a time (see k_conv3d_mid16x): the next
    // step's fragments, then the weights of step + 2
    static_assert(3 * RW + 3 <= 6 * RW, "one prefetch behind an MFMA");
#pragma unroll
    for (int step = 0; step < 9; ++step) {
        const int kd = step / 3, kh = step % 3;
        const int cb = step & 1, nb = cb ^ 1;
        const int sn = step < 8 ? step + 1 : 8;                          // (clamped: the last prefetch is unused)
        const int off_n = ((sn / 3) * HY + sn % 3) * HX;
        if (P1 < SITER && step == 3) {
            // phase 2 of the staging (halo planes TD, TD + 1); the fragments prefetched during step 2 may predate it
#pragma unroll
            for (int i = P1; i < SITER; ++i) stage_write(i);
            __syncthreads();
#pragma unroll
            for (int r = 0; r < RW; ++r) load_b_row(bb[cb], r, (kd * HY + kh) * HX);
        }
        auto prefetch = [&](int k) {                                     // as in k_conv3d_mid16x: one load per call
            if (k < 3 * RW)
                bb[nb][k / 3][k % 3] = *reinterpret_cast<const uint4 *>(ldsb + mid8x_slot(rbase[k / 3] + off_n) + (k % 3) * PLANE);
            else if (k < 3 * RW + 3)
                wa[(kh + 2) % 3][k - 3 * RW] = wp[(size_t)(step + 2) * Cfg::STEP_U4 + (k - 3 * RW) * 64];
        };
#pragma unroll
        for (int T = 0; T < 6; ++T)
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                acc[r] = mfma_split_bf16_term(acc[r], wa[kh], bb[cb][r], T);
                const int m = T * RW + r;                                // 18 MFMAs, 12 prefetches: one behind each of the first 12
                if (m < 3 * RW + 3) {
                    __builtin_amdgcn_sched_barrier(0);
                    prefetch(m);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
    }

    LWS_STAMPK(20, 2);
    // ---- epilogue: row i = 4*(lane>>4) + reg -> xpar = (lane>>4)>>1, cout = 4*((lane>>4)&1) + reg
    float *outb = out + (int64_t)b * D * h * w * 8;
    const int gx = x0 + 2 * n + xpar;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int row = wave * RW + r;
        const int gd = d0 + row / TY, gy = y0 + row % TY;
        if (gd < D && gy < h && gx < w) {
            float4 v;
            v.x = bn_relu(acc[r][0], es8.x, et8.x);
            v.y = bn_relu(acc[r][1], es8.y, et8.y);
            v.z = bn_relu(acc[r][2], es8.z, et8.z);
            v.w = bn_relu(acc[r][3], es8.w, et8.w);
            store_act4(outb + (((int64_t)gd * h + gy) * w + gx) * 8 + cb8, v, 0);
        }
    }
    LWS_STAMPK(20, 3);
}

// =============================================================================================
// Middle layers, C3 == 8, on v_mfma_f32_4x4x1_16B_f32 (k_conv3d_mid8q).
//
// With 8 output channels a 16x16x4 tile must pair two output positions per 16-row tile and pays 25 % of every instruction for
// structural zeros (their windows overlap in 2 of 3 taps; that form, k_conv3d_mid8, was superseded in round 3 and removed in
// round 6).  The multi-block MFMA has no such padding: one instruction is 16
// independent 4x4 outer products with K = 1,  D_b[i][j] += A_b[i] * B_b[j]  (b = block = lane / 4), and with the A-block
// broadcast (CBSZ = 4, ABID = k) all 16 blocks take block k's A.  So
//     lane l = voxel l of a 64-voxel group (2 rows x 32 along x)  ->  B operand = ONE activation value per lane,
//     A = W[4 cg .. 4 cg + 3][cin][tap] held by lanes 4k .. 4k+3 of a register, k = 2 cin + cg,
//     D register i of lane l = output channel 4 cg + i of voxel l:
// one instruction = 4 output channels x 64 voxels x 1 (tap, cin) term, every FLOP useful, and ONE register holds all
// 16 (cin, cg) weight blocks of a tap -- 27 A registers for the whole layer.  The chain per output is the
// contract's: taps (kd,kh,kw) outer, cin ascending, one fma each (K = 1).
// Wave = one 64-voxel group, two accumulators (cout 0-3, 4-7); workgroup = TD x TY x 32 voxels = TD*TY/2 waves; the
// halo tile sits in LDS channels-last as two half-planes [cin half][voxel] of float4 (one ds_read_b128 at lane base +
// immediate feeds 8 MFMAs; consecutive lanes read consecutive 16-byte slots: conflict-free).  Latency is hidden by the
// 5-6 co-resident waves per SIMD, not by software pipelining inside a wave (64 VGPRs).
// =============================================================================================
template <int TD, int TY>
struct Mid8qCfg {
    static_assert(TY % 2 == 0, "a wave owns two rows of 32 voxels");
    static constexpr int NW = TD * TY / 2, NT = 64 * NW;
    static constexpr int HD = TD + 2, HY = TY + 2, HX = 34;
    static constexpr int NVOX = HD * HY * HX;
    // half-plane stride in float4, == 4 (mod 8): the staging's ds_write_b128 of neighbouring lanes (alternating halves)
    // then fall on different banks
    static constexpr int NP = NVOX + ((4 - NVOX % 8) + 8) % 8;
    static constexpr int ITEMS = 2 * NVOX, SITER = (ITEMS + NT - 1) / NT;
    static constexpr int LDS_BYTES = 2 * NP * 16;
};

// one (tap, cin) term for both output-channel groups; ABID must be a literal
#define LWS_Q2(CIN, BV)                                                                    \
    lo = __builtin_amdgcn_mfma_f32_4x4x1f32(aw, (BV), lo, 4, 2 * (CIN), 0);                \
    hi = __builtin_amdgcn_mfma_f32_4x4x1f32(aw, (BV), hi, 4, 2 * (CIN) + 1, 0);

template <int TD, int TY>
__global__ __launch_bounds__((Mid8qCfg<TD, TY>::NT)) void k_conv3d_mid8q(const float *__restrict__ in,      // [B,D,h,w,8]
                                                                        const float *__restrict__ wpk,     // [7][64][4]
                                                                        const float *__restrict__ bn_s,    // next layer BN [8]
                                                                        const float *__restrict__ bn_t,
                                                                        float *__restrict__ out, int D, int h, int w,
                                                                        int tiles_x, int tiles_y, int wt, int tiles_d)
{
    using Cfg = Mid8qCfg<TD, TY>;
    constexpr int HY = Cfg::HY, HX = Cfg::HX, NP = Cfg::NP;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float4 *lds4 = reinterpret_cast<float4 *>(lds);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int tx, ty, td;
    tile_coords(xcd_tile(blockIdx.x, gridDim.x), tiles_x, tiles_d, tx, ty, td);
    const int b = blockIdx.y;
    const int x0 = tx * 32, y0 = ty * TY, d0 = td * TD;
    const float *inb = in + (int64_t)b * D * h * w * 8;
    LWS_STAMPK(18, 0);

    // ---- stage: item = (voxel, channel half) = 16 B, enumerated plane by plane; all global loads of a thread are in
    //      flight before its first LDS write; out-of-volume voxels become literal zeros (the convolution's padding).
    //      Two phases as in k_conv3d_mid16: iterations i < P1 cover halo planes [0, TD) -- all the kd = 0 taps read, for
    //      every wave -- and are written first; the rest lands in LDS after the kd = 0 taps, under their MFMAs.
    //      (Measured r03: a row-per-wave-instruction form with wave-uniform row arithmetic, 134 instead of 273 VALU
    //      instructions in front of the first MFMA, was SLOWER -- 105.8 vs 95.8 us at 8 x 9x128x256: a 34-voxel row is
    //      68 float4, i.e. a second, 4-lane load per row, 60 instead of 36 wave-loads per tile, and the staging is bound
    //      by the CU's fetch path (~10 B/cycle/CU from beyond L2), not by index arithmetic.)
    constexpr int NT = Cfg::NT, SITER = Cfg::SITER;
    constexpr int P1 = (TD * HY * HX * 2 + NT - 1) / NT < SITER ? (TD * HY * HX * 2 + NT - 1) / NT : SITER;
    float4 c[SITER];
    bool okv[SITER];
#pragma unroll
    for (int i = 0; i < SITER; ++i) {
        const int it = tid + i * NT;
        const int half = it & 1, v = it >> 1;
        const int hx = v % HX, t2 = v / HX, hy = t2 % HY, hd = t2 / HY;
        const int gd = d0 + hd - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
        okv[i] = it < Cfg::ITEMS && gd >= 0 && gd < D && gy >= 0 && gy < h && gx >= 0 && gx < w;
        c[i] = *reinterpret_cast<const float4 *>(inb + (okv[i] ? (((int64_t)gd * h + gy) * w + gx) * 8 + half * 4 : 0));
    }
    auto stage_write = [&](int i) {
        const int it = tid + i * NT;
        if (it < Cfg::ITEMS)
            lds4[(it & 1) * NP + (it >> 1)] = make_float4(okv[i] ? c[i].x : 0.f, okv[i] ? c[i].y : 0.f, okv[i] ? c[i].z : 0.f, okv[i] ? c[i].w : 0.f);
    };
    // the 27 A registers of this lane ([tap / 4][lane][tap % 4]) and the next layer's BatchNorm (wave-uniform)
    float wa[28];
#pragma unroll
    for (int s4 = 0; s4 < 7; ++s4) {
        const float4 v = reinterpret_cast<const float4 *>(wpk)[s4 * 64 + lane];
        wa[4 * s4 + 0] = v.x;
        wa[4 * s4 + 1] = v.y;
        wa[4 * s4 + 2] = v.z;
        wa[4 * s4 + 3] = v.w;
    }
    const float4 s_lo = *reinterpret_cast<const float4 *>(bn_s), s_hi = *reinterpret_cast<const float4 *>(bn_s + 4);
    const float4 t_lo = *reinterpret_cast<const float4 *>(bn_t), t_hi = *reinterpret_cast<const float4 *>(bn_t + 4);
#pragma unroll
    for (int i = 0; i < P1; ++i) stage_write(i);
    __syncthreads();
    LWS_STAMPK(18, 1);

    // ---- this wave's group: d-plane pd, rows 2 pr and 2 pr + 1, 32 voxels each
    const int pd = wave / (TY / 2), pr = wave % (TY / 2);
    const int ly = 2 * pr + (lane >> 5), lx = lane & 31;
    const float4 *bp = lds4 + (pd * HY + ly) * HX + lx;
    floatx4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) {
        if (kd == 1 && P1 < SITER) {
#pragma unroll
            for (int i = P1; i < SITER; ++i) stage_write(i);
            __syncthreads();
        }
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int off = (kd * HY + kh) * HX + kw;
                const float4 b0 = bp[off], b1 = bp[NP + off];
                const float aw = wa[(kd * 3 + kh) * 3 + kw];
                LWS_Q2(0, b0.x)
                LWS_Q2(1, b0.y)
                LWS_Q2(2, b0.z)
                LWS_Q2(3, b0.w)
                LWS_Q2(4, b1.x)
                LWS_Q2(5, b1.y)
                LWS_Q2(6, b1.z)
                LWS_Q2(7, b1.w)
            }
    }
    LWS_STAMPK(18, 2);

    // ---- epilogue: register i of lo / hi = output channel i / 4 + i of this lane's voxel; next layer's BN + ReLU;
    //      32 contiguous bytes per lane, consecutive lanes consecutive voxels
    const int gd = d0 + pd, gy = y0 + ly, gx = x0 + lx;
    if (gd < D && gy < h && gx < w) {
        float *o = out + (int64_t)b * D * h * w * 8 + (((int64_t)gd * h + gy) * w + gx) * 8;
        float4 v;
        v.x = bn_relu(lo[0], s_lo.x, t_lo.x);
        v.y = bn_relu(lo[1], s_lo.y, t_lo.y);
        v.z = bn_relu(lo[2], s_lo.z, t_lo.z);
        v.w = bn_relu(lo[3], s_lo.w, t_lo.w);
        store_act4(o, v, wt);
        v.x = bn_relu(hi[0], s_hi.x, t_hi.x);
        v.y = bn_relu(hi[1], s_hi.y, t_hi.y);
        v.z = bn_relu(hi[2], s_hi.z, t_hi.z);
        v.w = bn_relu(hi[3], s_hi.w, t_hi.w);
        store_act4(o + 4, v, wt);
    }
    LWS_STAMPK(18, 3);
}
#undef LWS_Q2

// =============================================================================================
// First layer, C3 == 8 (stages 2 and 3), on fp32 MFMA: the parity-row scheme with one input channel.  Rows =
// (x parity, cout), columns = voxel pairs, K = the four x positions t = 0..3 both parities read, so each (kd,kh) is ONE
// MFMA whose A operand is W[cout][kd][kh][t - xpar] (zero outside 0..2) and whose B operand is one ds_read_b32 of the
// BN0+ReLU'd cost halo tile: 9 MFMAs per row of 32 voxels, taps ascending -- the same chain as k_conv3d_first, which
// spends its time re-reading broadcast weights from LDS (54 ds_read_b128 per thread: LDS-issue-bound).
// =============================================================================================
template <int TD, int TY>
__global__ __launch_bounds__(256) void k_conv3d_first8(const float *__restrict__ cost,    // [B,D,h,w]
                                                       const float *__restrict__ wpk,     // [9][64] A fragments
                                                       const float *__restrict__ bn0_s, const float *__restrict__ bn0_t,
                                                       const float *__restrict__ bn_s,    // next layer BN [8]
                                                       const float *__restrict__ bn_t, float *__restrict__ out, int D,
                                                       int h, int w, int tiles_x, int tiles_y, int tiles_d)
{
    constexpr int RW = TD * TY / 4, HD = TD + 2, HY = TY + 2, HX = 34, NVOX = HD * HY * HX, SITER = (NVOX + 255) / 256;
    static_assert(TD * TY % 4 == 0, "rows must split over 4 waves");
    __shared__ float lds[NVOX];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    int tx, ty, td;
    tile_coords(xcd_tile(blockIdx.x, gridDim.x), tiles_x, tiles_d, tx, ty, td);
    const int b = blockIdx.y;
    const int x0 = tx * 32, y0 = ty * TY, d0 = td * TD;
    const float *cb = cost + (int64_t)b * D * h * w;
    float c[SITER];
    bool okv[SITER];
#pragma unroll
    for (int i = 0; i < SITER; ++i) {
        const int v = tid + i * 256;
        const int hx = v % HX, t2 = v / HX, hy = t2 % HY, hd = t2 / HY;
        const int gd = d0 + hd - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
        okv[i] = v < NVOX && gd >= 0 && gd < D && gy >= 0 && gy < h && gx >= 0 && gx < w;
        c[i] = cb[okv[i] ? ((int64_t)gd * h + gy) * w + gx : 0];
    }
    const float s0 = bn0_s[0], t0 = bn0_t[0];
    const int xpar = g >> 1, cbo = 4 * (g & 1);
    const float4 es = *reinterpret_cast<const float4 *>(bn_s + cbo);
    const float4 et = *reinterpret_cast<const float4 *>(bn_t + cbo);
    float wa[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wa[k] = wpk[k * 64 + lane];
#pragma unroll
    for (int i = 0; i < SITER; ++i) {
        const int v = tid + i * 256;
        if (v < NVOX) lds[v] = okv[i] ? bn_relu(c[i], s0, t0) : 0.0f;
    }
    __syncthreads();
    floatx4 acc[RW];
    int rbase[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        acc[r] = (floatx4){0.f, 0.f, 0.f, 0.f};
        const int row = wave * RW + r;
        rbase[r] = ((row / TY) * HY + row % TY) * HX + 2 * n + g;
    }
#pragma unroll
    for (int kd = 0; kd < 3; ++kd)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int r = 0; r < RW; ++r)
                acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[kd * 3 + kh], lds[rbase[r] + (kd * HY + kh) * HX], acc[r], 0, 0, 0);
    float *outb = out + (int64_t)b * D * h * w * 8;
    const int gx = x0 + 2 * n + xpar;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int row = wave * RW + r;
        const int gd = d0 + row / TY, gy = y0 + row % TY;
        if (gd < D && gy < h && gx < w) {
            float4 v;
            v.x = bn_relu(acc[r][0], es.x, et.x);
            v.y = bn_relu(acc[r][1], es.y, et.y);
            v.z = bn_relu(acc[r][2], es.z, et.z);
            v.w = bn_relu(acc[r][3], es.w, et.w);
            *reinterpret_cast<float4 *>(outb + (((int64_t)gd * h + gy) * w + gx) * 8 + cbo) = v;
        }
    }
}

// =============================================================================================
// First layer, C3 a multiple of 16 (stage 1), on fp32 MFMA: rows = 16 output channels per tile, columns = 16 voxels
// along x, K = 4 consecutive taps per instruction (27 taps + one zero tap = 7 MFMAs per output-channel tile).  Lane
// (n, g) of MFMA j supplies tap 4j+g of voxel n: one ds_read_b32 at a per-lane offset into the BN0+ReLU'd cost halo
// tile, shared by the C3/16 output-channel tiles.  Taps ascending: the same chain as k_conv3d_first.
// =============================================================================================
// SHIFT != 0 (stage 1 inside lws_forward / lws_disparity_stages): the kernel also BUILDS the volume
// (k_volume_l1_shift's arithmetic: cost = sum over c ascending of |L - R(x-d)|, occluded columns |L - 0|) -- the
// feature rows of the tile's halo go to LDS, the (TD+2)(TY+2)18 cost values are evaluated from them (2.8x
// recomputation of a cheap volume), the tile's own values are written out (`cost` is then the OUTPUT: the raw volume
// is the stack's skip input) and one launch disappears.  SHIFT == 2: features rounded to fp16 (BASELINE config 5).
template <int C3, int TD, int TY, int SHIFT>
__global__ __launch_bounds__(256) void k_conv3d_first16(float *__restrict__ cost,          // [B,D,h,w] (in; out if SHIFT)
                                                        const float *__restrict__ featL,   // SHIFT: [B,16,h,w]
                                                        const float *__restrict__ featR,
                                                        const float *__restrict__ wpk,     // [C3/16][7][64] A fragments
                                                        const float *__restrict__ bn0_s, const float *__restrict__ bn0_t,
                                                        const float *__restrict__ bn_s,    // next layer BN [C3]
                                                        const float *__restrict__ bn_t, float *__restrict__ out, int D,
                                                        int h, int w, int tiles_x, int tiles_y, int tiles_d)
{
    constexpr int MT = C3 / 16, RW = TD * TY / 4, HD = TD + 2, HY = TY + 2, HX = 18, NVOX = HD * HY * HX,
                  SITER = (NVOX + 255) / 256;
    static_assert(TD * TY % 4 == 0 && C3 % 16 == 0, "bad first16 geometry");
    __shared__ float lds[NVOX];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    int tx, ty, td;
    tile_coords(xcd_tile(blockIdx.x, gridDim.x), tiles_x, tiles_d, tx, ty, td);
    const int b = blockIdx.y;
    const int x0 = tx * 16, y0 = ty * TY, d0 = td * TD;
    float *cb = cost + (int64_t)b * D * h * w;
    constexpr int FC = 16, RWD = HX + HD - 1;                      // feature channels; width of the R row window
    constexpr int NL = FC * HY * HX, NR = FC * HY * RWD;
    __shared__ float sF[SHIFT ? NL + NR : 1];
    float c[SITER];
    bool okv[SITER];
    if (SHIFT) {
        // feature rows y0-1 .. y0+TY of L (columns x0-1 .. x0+16) and R (columns shifted by d0-1 .. d0+TD)
        constexpr int IL = (NL + 255) / 256, IR = (NR + 255) / 256;
        const int64_t plane = (int64_t)h * w;
        const float *Lb = featL + (int64_t)b * FC * plane, *Rb = featR + (int64_t)b * FC * plane;
        float vl[IL], vr[IR];
        bool okl[IL], okr[IR];
#pragma unroll
        for (int i = 0; i < IL; ++i) {
            const int j = tid + i * 256;
            const int ch = j / (HY * HX), hy = (j / HX) % HY, hx = j % HX;
            const int gy = y0 + hy - 1, gx = x0 + hx - 1;
            okl[i] = j < NL && gy >= 0 && gy < h && gx >= 0 && gx < w;
            vl[i] = Lb[okl[i] ? (int64_t)ch * plane + (int64_t)gy * w + gx : 0];
        }
#pragma unroll
        for (int i = 0; i < IR; ++i) {
            const int j = tid + i * 256;
            const int ch = j / (HY * RWD), hy = (j / RWD) % HY, jx = j % RWD;
            const int gy = y0 + hy - 1, gx = x0 - 1 - d0 - TD + jx;
            okr[i] = j < NR && gy >= 0 && gy < h && gx >= 0 && gx < w;
            vr[i] = Rb[okr[i] ? (int64_t)ch * plane + (int64_t)gy * w + gx : 0];
        }
#pragma unroll
        for (int i = 0; i < IL; ++i)
            if (tid + i * 256 < NL) sF[tid + i * 256] = okl[i] ? (SHIFT == 2 ? __half2float(__float2half_rn(vl[i])) : vl[i]) : 0.0f;
#pragma unroll
        for (int i = 0; i < IR; ++i)
            if (tid + i * 256 < NR) sF[NL + tid + i * 256] = okr[i] ? (SHIFT == 2 ? __half2float(__float2half_rn(vr[i])) : vr[i]) : 0.0f;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < SITER; ++i) {
            const int v = tid + i * 256, vc = v < NVOX ? v : 0;
            const int hx = vc % HX, t2 = vc / HX, hy = t2 % HY, hd = t2 / HY;
            const int gd = d0 + hd - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
            okv[i] = v < NVOX && gd >= 0 && gd < D && gy >= 0 && gy < h && gx >= 0 && gx < w;
            const float *lp = sF + hy * HX + hx, *rp = sF + NL + hy * RWD + hx - hd + TD + 1;
            float acc = 0.0f;
#pragma unroll
            for (int ch = 0; ch < FC; ++ch) acc = acc + fabsf(lp[ch * HY * HX] - rp[ch * HY * RWD]);
            c[i] = acc;
            if (okv[i] && hd >= 1 && hd <= TD && hy >= 1 && hy <= TY && hx >= 1 && hx <= 16)
                cb[((int64_t)gd * h + gy) * w + gx] = acc;
        }
    } else {
#pragma unroll
        for (int i = 0; i < SITER; ++i) {
            const int v = tid + i * 256;
            const int hx = v % HX, t2 = v / HX, hy = t2 % HY, hd = t2 / HY;
            const int gd = d0 + hd - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
            okv[i] = v < NVOX && gd >= 0 && gd < D && gy >= 0 && gy < h && gx >= 0 && gx < w;
            c[i] = cb[okv[i] ? ((int64_t)gd * h + gy) * w + gx : 0];
        }
    }
    const float s0 = bn0_s[0], t0 = bn0_t[0];
    float4 es[MT], et[MT];
    float wa[MT][7];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        es[mt] = *reinterpret_cast<const float4 *>(bn_s + mt * 16 + 4 * g);
        et[mt] = *reinterpret_cast<const float4 *>(bn_t + mt * 16 + 4 * g);
#pragma unroll
        for (int j = 0; j < 7; ++j) wa[mt][j] = wpk[(mt * 7 + j) * 64 + lane];
    }
    // halo offset of tap 4j+g (tap 27 does not exist: its weight is zero, any in-tile address will do)
    int toff[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int tap = 4 * j + g < 27 ? 4 * j + g : 26;
        const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
        toff[j] = (kd * HY + kh) * HX + kw + n;
    }
#pragma unroll
    for (int i = 0; i < SITER; ++i) {
        const int v = tid + i * 256;
        if (v < NVOX) lds[v] = okv[i] ? bn_relu(c[i], s0, t0) : 0.0f;
    }
    __syncthreads();
    floatx4 acc[RW][MT];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[r][mt] = (floatx4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 7; ++j)
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int row = wave * RW + r;
            const float bv = lds[((row / TY) * HY + row % TY) * HX + toff[j]];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[r][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[mt][j], bv, acc[r][mt], 0, 0, 0);
        }
    float *outb = out + (int64_t)b * D * h * w * C3;
    const int gx = x0 + n;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int row = wave * RW + r;
        const int gd = d0 + row / TY, gy = y0 + row % TY;
        if (gd < D && gy < h && gx < w) {
            float *o = outb + (((int64_t)gd * h + gy) * w + gx) * C3;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float4 v;
                v.x = bn_relu(acc[r][mt][0], es[mt].x, et[mt].x);
                v.y = bn_relu(acc[r][mt][1], es[mt].y, et[mt].y);
                v.z = bn_relu(acc[r][mt][2], es[mt].z, et[mt].z);
                v.w = bn_relu(acc[r][mt][3], es[mt].w, et[mt].w);
                *reinterpret_cast<float4 *>(o + mt * 16 + 4 * g) = v;
            }
        }
    }
}

// [c3][27] -> A fragments [c3/16][7][lane]: lane (m, g) of MFMA j holds W[16 mt + m][tap = 4j + g] (0 for tap 27)
void pack_first16_weights(const float *w, int c3, float *out)
{
    for (int mt = 0; mt < c3 / 16; ++mt)
        for (int j = 0; j < 7; ++j)
            for (int lane = 0; lane < 64; ++lane) {
                const int m = lane & 15, tap = 4 * j + (lane >> 4);
                out[(mt * 7 + j) * 64 + lane] = tap < 27 ? w[(16 * mt + m) * 27 + tap] : 0.0f;
            }
}

// [8][27] -> A fragments [kd*3+kh][lane]: lane (m, t): row m = 8*xpar + cout, W[cout][kd][kh][t - xpar] or 0
void pack_first8_weights(const float *w, float *out)
{
    for (int kdh = 0; kdh < 9; ++kdh)
        for (int lane = 0; lane < 64; ++lane) {
            const int m = lane & 15, t = lane >> 4, xpar = m >> 3, cout = m & 7, kw = t - xpar;
            out[kdh * 64 + lane] = (kw >= 0 && kw <= 2) ? w[cout * 27 + kdh * 3 + kw] : 0.0f;
        }
}

// =============================================================================================
// Last layer: act [B,D,h,w,C3] -> cost_out [B,D,h,w] = conv(act) + cost_in   (models.py:137), optionally
// followed in the same kernel by the soft-argmin over D (models.py:142,151-152,167-179) when the tile spans
// the whole disparity axis (stages 2/3: D = 9).
// Workgroup = TD x TY x TX output voxels, one thread each; the halo tile is staged in LDS (channels-last,
// padded voxel stride) with all global loads in flight at once; 27 x C3 fma chain per voxel; the weights
// [tap][cin] are indexed uniformly (scalar cache).
// =============================================================================================
template <int C3, int TD, int TY, int TX, bool FUSE>
struct LastCfg {
    static constexpr int NOUT = TD * TY * TX;
    static constexpr int NT = ((NOUT + 63) / 64) * 64;
    static constexpr int HD = TD + 2, HY = TY + 2, HX = TX + 2;
    static constexpr int VS = C3 + 4;
    static constexpr int NVOX = HD * HY * HX;
    static constexpr int ITEMS = NVOX * (C3 / 4);
    static constexpr int SITER = (ITEMS + NT - 1) / NT;
    static constexpr int LDS_FLOATS = NVOX * VS + (FUSE ? 2 * NOUT : 0);
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
};

template <int C3, int TD, int TY, int TX, bool FUSE>
__global__ __launch_bounds__((LastCfg<C3, TD, TY, TX, FUSE>::NT)) void k_conv3d_last(
    const float *__restrict__ act, const float *__restrict__ wgt,   // [27][C3]
    const float *__restrict__ skip, float *__restrict__ cost_out,  // may be nullptr when FUSE
    float *__restrict__ low, float start, int D, int h, int w, int tiles_x, int tiles_y, int tiles_d)
{
    using Cfg = LastCfg<C3, TD, TY, TX, FUSE>;
    constexpr int HY = Cfg::HY, HX = Cfg::HX, VS = Cfg::VS, NT = Cfg::NT, SITER = Cfg::SITER, C4 = C3 / 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    int tx_, ty_, td_;
    tile_coords(xcd_tile(blockIdx.x, gridDim.x), tiles_x, tiles_d, tx_, ty_, td_);
    const int b = blockIdx.y;
    const int x0 = tx_ * TX, y0 = ty_ * TY, d0 = td_ * TD;
    const int64_t vol = (int64_t)D * h * w;
    const float *ab = act + (int64_t)b * vol * C3;
    {
        float4 c[SITER];
#pragma unroll
        for (int i = 0; i < SITER; ++i) {
            const int it = tid + i * NT;
            const int c4 = it % C4, v = it / C4;
            const int hx = v % HX, t2 = v / HX, hy = t2 % HY, hd = t2 / HY;
            const int gd = d0 + hd - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
            const bool ok = it < Cfg::ITEMS && gd >= 0 && gd < D && gy >= 0 && gy < h && gx >= 0 && gx < w;
            c[i] = *reinterpret_cast<const float4 *>(ab + (ok ? (((int64_t)gd * h + gy) * w + gx) * C3 + c4 * 4 : 0));
            if (!ok) c[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < SITER; ++i) {
            const int it = tid + i * NT;
            if (it < Cfg::ITEMS) *reinterpret_cast<float4 *>(&lds[(it / C4) * VS + (it % C4) * 4]) = c[i];
        }
    }
    __syncthreads();
    const int lx = tid % TX, ly = (tid / TX) % TY, ld = tid / (TX * TY);
    const int gd = d0 + ld, gy = y0 + ly, gx = x0 + lx;
    const bool live = tid < Cfg::NOUT && gd < D && gy < h && gx < w;
    float val = 0.0f;
    if (tid < Cfg::NOUT) {
        float acc = 0.0f;
        const float *base = lds + ((ld * HY + ly) * HX + lx) * VS;
#pragma unroll 1
        for (int kd = 0; kd < 3; ++kd)
#pragma unroll 1
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const float *p = base + ((kd * HY + kh) * HX + kw) * VS;
                    const float *wt = wgt + ((kd * 3 + kh) * 3 + kw) * C3;
#pragma unroll
                    for (int c4 = 0; c4 < C4; ++c4) {
                        const float4 a = *reinterpret_cast<const float4 *>(p + c4 * 4);
                        acc = fmaf(a.x, wt[c4 * 4 + 0], acc);
                        acc = fmaf(a.y, wt[c4 * 4 + 1], acc);
                        acc = fmaf(a.z, wt[c4 * 4 + 2], acc);
                        acc = fmaf(a.w, wt[c4 * 4 + 3], acc);
                    }
                }
        if (live) {
            const int64_t o = (int64_t)b * vol + ((int64_t)gd * h + gy) * w + gx;
            val = acc + skip[o];
            if (cost_out != nullptr) cost_out[o] = val;
        }
    }
    if (FUSE) {
        // Soft-argmin over the TD == D costs of each of the tile's TY*TX pixels, all NOUT threads working (round 5; until then
        // the TY*TX threads of plane 0 each ran the three serial passes of softargmin_pixel -- 27 to 72 dependent lws_expf
        // calls at the end of every workgroup).  The operations and their order are softargmin_pixel's: m = max_k(-c_k);
        // e_k = lws_expf(-c_k - m) -- one per thread; S = e_0 + e_1 + ... ascending; t_k = (e_k / S) * (start + k) -- one per
        // thread; result = t_0 + t_1 + ... ascending.  Only the two sums are serial (D additions each).
        constexpr int P = TY * TX;
        float *sC = lds + Cfg::NVOX * VS;            // [TD][P]: the costs, later the terms t_k
        float *sE = sC + Cfg::NOUT;                  // [TD][P]: e_k
        const int pix = tid % P;                     // (tid = ld * P + pix; for tid >= NOUT nothing below is used)
        if (tid < Cfg::NOUT) sC[tid] = val;
        __syncthreads();
        float e = 0.0f;
        if (tid < Cfg::NOUT) {
            float m = -sC[pix];
#pragma unroll
            for (int k = 1; k < TD; ++k) m = fmaxf(m, -sC[k * P + pix]);
            e = lws_expf(-val - m);
            sE[tid] = e;
        }
        __syncthreads();
        if (tid < Cfg::NOUT) {
            float S = 0.0f;
#pragma unroll
            for (int k = 0; k < TD; ++k) S = S + sE[k * P + pix];
            const float pk = e / S;
            sC[tid] = pk * (start + (float)ld);
        }
        __syncthreads();
        if (tid < P && gy < h && gx < w) {           // ld == 0 for these threads: gd = d0 = 0, TD == D
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < TD; ++k) acc = acc + sC[k * P + tid];
            low[((int64_t)b * h + gy) * w + gx] = acc;
        }
    }
}

// =============================================================================================
// host side
// =============================================================================================
size_t packed_mid_weight_floats(int c3)
{
    if (c3 == 8) return 28 * 64 + MID8X_PACK_FLOATS;   // k_conv3d_mid8q's fragments, then k_conv3d_mid8x's
    // 27 taps + two all-zero taps (branch-free two-taps-ahead prefetch in k_conv3d_mid16); C3 == 32: + the split-bf16
    // fragments of k_conv3d_mid16x, 29 taps (two all-zero) x 2 cout tiles x 3 variants x 64 lanes x 8 bf16 = 16 B each
    return (size_t)29 * c3 * c3 + (c3 == 32 ? (size_t)29 * 2 * 3 * 64 * 4 : 0);
}

// w: [cout][cin][27] (Conv3D weight [Cout,Cin,3,3,3] flattened)
void pack_mid_weights(const float *w, int c3, float *out)
{
    if (c3 == 8) {
        // k_conv3d_mid8q: register `tap`, lane 4 (2 cin + cg) + i  ->  W[4 cg + i][cin][tap]; stored [tap / 4][lane][tap % 4]
        float *oq = out;
        for (int tap = 0; tap < 28; ++tap)
            for (int lane = 0; lane < 64; ++lane) {
                const int k = lane >> 2, i = lane & 3, cin = k >> 1, cg = k & 1;
                oq[((tap >> 2) * 64 + lane) * 4 + (tap & 3)] = tap < 27 ? w[((4 * cg + i) * 8 + cin) * 27 + tap] : 0.0f;
            }
        // k_conv3d_mid8x: lane l of (step = 3 kd + kh, variant) holds W[l & 7][cin = j][kd][kh][kw = (l >> 4) - ((l >> 3) & 1)],
        // j = 0..7, as bf16 bits (zero where kw is outside 0..2); two all-zero steps close the stream
        uint16_t *ox = reinterpret_cast<uint16_t *>(out + 28 * 64);
        for (int step = 0; step < 11; ++step)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int i = lane & 15, xpar = i >> 3, cout = i & 7, kw = (lane >> 4) - xpar;
                    uint32_t v[3] = {0, 0, 0};
                    if (step < 9 && kw >= 0 && kw <= 2) split_bf16x3(w[(cout * 8 + j) * 27 + step * 3 + kw], v[0], v[1], v[2]);
                    for (int t = 0; t < 3; ++t) ox[(((size_t)step * 3 + t) * 64 + lane) * 8 + j] = (uint16_t)v[t];
                }
        return;
    }
    const int Q = c3 / 16, MT = c3 / 16;
    if (c3 == 32) {
        // k_conv3d_mid16x: lane l of (tap, mt, variant) holds W[16 mt + (l & 15)][cin = 8 (l >> 4) + j][tap], j = 0..7, as bf16
        uint16_t *ox = reinterpret_cast<uint16_t *>(out + (size_t)29 * c3 * c3);
        for (int tap = 0; tap < 29; ++tap)
            for (int mt = 0; mt < 2; ++mt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int cout = 16 * mt + (lane & 15), cin = 8 * (lane >> 4) + j;
                        uint32_t v[3] = {0, 0, 0};
                        if (tap < 27) split_bf16x3(w[((size_t)cout * c3 + cin) * 27 + tap], v[0], v[1], v[2]);
                        for (int t = 0; t < 3; ++t) ox[((((size_t)tap * 2 + mt) * 3 + t) * 64 + lane) * 8 + j] = (uint16_t)v[t];
                    }
    }
    for (int tap = 0; tap < 27; ++tap)
        for (int q = 0; q < Q; ++q)
            for (int mt = 0; mt < MT; ++mt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 4; ++j) {
                        const int m = lane & 15, g = lane >> 4;
                        const int cout = 16 * mt + m, cin = 16 * q + 4 * j + g;
                        out[((((size_t)tap * Q + q) * MT + mt) * 64 + lane) * 4 + j] = w[((size_t)cout * c3 + cin) * 27 + tap];
                    }
}

bool shift_first_can_fuse(const Stage3d &s, int C)
{
    return C == 16 && (s.c3 == 16 || s.c3 == 32) && !s.layers.empty() && s.layers[0].w_mfma != nullptr;
}

// stage-1 volume (written to `cost`: the stack's skip input) + first Conv3D layer in one launch
int launch_shift_first(const Stage3d &s, const float *L, const float *R, float *cost, float *act_out, int B, int C, int D,
                       int h, int w, hipStream_t st, bool q16)
{
    if (!shift_first_can_fuse(s, C)) {
        set_error("shift_first: unsupported channels C=%d c3=%d", C, s.c3);
        return LWS_ERR_INVALID;
    }
    constexpr int TD = 3, TY = 4;
    const int tiles_x = cdiv(w, 16), tiles_y = cdiv(h, TY), tiles_d = cdiv(D, TD);
    dim3 grid(tiles_x * tiles_y * tiles_d, B), block(256);
#define LWS_SF(C3v, SH)                                                                                               \
    hipLaunchKernelGGL((k_conv3d_first16<C3v, TD, TY, SH>), grid, block, 0, st, cost, L, R, s.layers[0].w_mfma,        \
                       s.layers[0].bn_s, s.layers[0].bn_t, s.layers[1].bn_s, s.layers[1].bn_t, act_out, D, h, w,      \
                       tiles_x, tiles_y, tiles_d)
    if (s.c3 == 32) {
        if (q16) LWS_SF(32, 2);
        else LWS_SF(32, 1);
    } else {
        if (q16) LWS_SF(16, 2);
        else LWS_SF(16, 1);
    }
#undef LWS_SF
    LWS_LAUNCH_CHECK();
    return LWS_OK;
}

int launch_conv3d_first(const Stage3d &s, const float *cost, float *act_out, int B, int D, int h, int w,
                        hipStream_t st)
{
    if (s.c3 == 8 && s.layers[0].w_mfma != nullptr) {
        constexpr int TD = 3, TY = 4;
        const int tiles_x = cdiv(w, 32), tiles_y = cdiv(h, TY), tiles_d = cdiv(D, TD);
        dim3 grid(tiles_x * tiles_y * tiles_d, B), block(256);
        hipLaunchKernelGGL((k_conv3d_first8<TD, TY>), grid, block, 0, st, cost, s.layers[0].w_mfma, s.layers[0].bn_s,
                           s.layers[0].bn_t, s.layers[1].bn_s, s.layers[1].bn_t, act_out, D, h, w, tiles_x, tiles_y,
                           tiles_d);
        LWS_LAUNCH_CHECK();
        return LWS_OK;
    }
    if ((s.c3 == 16 || s.c3 == 32) && s.layers[0].w_mfma != nullptr) {
        constexpr int TD = 3, TY = 4;
        const int tiles_x = cdiv(w, 16), tiles_y = cdiv(h, TY), tiles_d = cdiv(D, TD);
        dim3 grid(tiles_x * tiles_y * tiles_d, B), block(256);
        float *cin = const_cast<float *>(cost);     // SHIFT = 0: read only
        const float *nofeat = nullptr;
        if (s.c3 == 32)
            hipLaunchKernelGGL((k_conv3d_first16<32, TD, TY, 0>), grid, block, 0, st, cin, nofeat, nofeat, s.layers[0].w_mfma,
                               s.layers[0].bn_s, s.layers[0].bn_t, s.layers[1].bn_s, s.layers[1].bn_t, act_out, D, h, w,
                               tiles_x, tiles_y, tiles_d);
        else
            hipLaunchKernelGGL((k_conv3d_first16<16, TD, TY, 0>), grid, block, 0, st, cin, nofeat, nofeat, s.layers[0].w_mfma,
                               s.layers[0].bn_s, s.layers[0].bn_t, s.layers[1].bn_s, s.layers[1].bn_t, act_out, D, h, w,
                               tiles_x, tiles_y, tiles_d);
        LWS_LAUNCH_CHECK();
        return LWS_OK;
    }
    set_error("conv3d_first: unsupported channel count %d (8, 16, 32) or weights not packed", s.c3);
    return LWS_ERR_INVALID;
}

template <int C3, int TD, int TY, int WR, int WM>
static int mid16_launch(const Stage3d &s, int layer, const float *in, float *out, int B, int D, int h, int w,
                        hipStream_t st, hipEvent_t e0, hipEvent_t e1)
{
    using Cfg = Mid16Cfg<C3, TD, TY, WR, WM>;
    static std::atomic<uint64_t> attr_done{0};
    {
        const int rc_ = ensure_dyn_lds(attr_done, reinterpret_cast<const void *>(&k_conv3d_mid16<C3, TD, TY, WR, WM>), Cfg::LDS_BYTES);
        if (rc_) return rc_;
    }
    const int tiles_x = cdiv(w, 16), tiles_y = cdiv(h, TY), tiles_d = cdiv(D, TD);
    dim3 grid(tiles_x * tiles_y * tiles_d, B), block(256);
    block = dim3(Cfg::NT);
    // write-through stores were measured a loss here (26.3 -> 27.6 us at B = 1): the next layer re-reads these
    // activations at once and finds them in the XCD's L2 only when they were stored write-back
    const int wt = 0;
    if (e0 != nullptr) {
        // profiler on: the events carry the kernel's own begin / end timestamps (no dispatch latency in between)
        hipExtLaunchKernelGGL((k_conv3d_mid16<C3, TD, TY, WR, WM>), grid, block, Cfg::LDS_BYTES, st, e0, e1, 0, in,
                              reinterpret_cast<const float4 *>(s.layers[layer].w), s.layers[layer + 1].bn_s,
                              s.layers[layer + 1].bn_t, out, D, h, w, tiles_x, tiles_y, wt, tiles_d, s.clk);
    } else {
        LWS_LAUNCH_STOP((k_conv3d_mid16<C3, TD, TY, WR, WM>), grid, block, Cfg::LDS_BYTES, st, in,
                        reinterpret_cast<const float4 *>(s.layers[layer].w), s.layers[layer + 1].bn_s,
                        s.layers[layer + 1].bn_t, out, D, h, w, tiles_x, tiles_y, wt, tiles_d, s.clk);
    }
    LWS_LAUNCH_CHECK();
    return LWS_OK;
}

template <int TD, int TY>
static int mid16x_launch(const Stage3d &s, int layer, const float *in, float *out, int B, int D, int h, int w, hipStream_t st,
                         hipEvent_t e0, hipEvent_t e1)
{
    using Cfg = Mid16xCfg<TD, TY>;
    static std::atomic<uint64_t> attr_done{0};
    {
        const int rc_ = ensure_dyn_lds(attr_done, reinterpret_cast<const void *>(&k_conv3d_mid16x<TD, TY>), Cfg::LDS_BYTES);
        if (rc_) return rc_;
    }
    const int tiles_x = cdiv(w, 16), tiles_y = cdiv(h, TY), tiles_d = cdiv(D, TD);
    dim3 grid(tiles_x * tiles_y * tiles_d, B), block(Cfg::NT);
    const uint4 *wx = reinterpret_cast<const uint4 *>(s.layers[layer].w + (size_t)29 * 32 * 32);
    if (e0 != nullptr)
        hipExtLaunchKernelGGL((k_conv3d_mid16x<TD, TY>), grid, block, Cfg::LDS_BYTES, st, e0, e1, 0, in, wx, s.layers[layer + 1].bn_s,
                              s.layers[layer + 1].bn_t, out, D, h, w, tiles_x, tiles_y, tiles_d);
    else
        hipLaunchKernelGGL((k_conv3d_mid16x<TD, TY>), grid, block, Cfg::LDS_BYTES, st, in, wx, s.layers[layer + 1].bn_s,
                           s.layers[layer + 1].bn_t, out, D, h, w, tiles_x, tiles_y, tiles_d);
    LWS_LAUNCH_CHECK();
    return LWS_OK;
}

constexpr int kCuLdsBytes = 160 * 1024;

template <int TD, int TY>
static int mid8q_launch(const Stage3d &s, int layer, const float *in, float *out, int B, int D, int h, int w, hipStream_t st)
{
    using Cfg = Mid8qCfg<TD, TY>;
    static std::atomic<uint64_t> attr_done{0};
    if (Cfg::LDS_BYTES > 48 * 1024) {
        const int rc_ = ensure_dyn_lds(attr_done, reinterpret_cast<const void *>(&k_conv3d_mid8q<TD, TY>), kCuLdsBytes);
        if (rc_) return rc_;
    }
    const int tiles_x = cdiv(w, 32), tiles_y = cdiv(h, TY), tiles_d = cdiv(D, TD);
    dim3 grid(tiles_x * tiles_y * tiles_d, B), block(Cfg::NT);
    hipLaunchKernelGGL((k_conv3d_mid8q<TD, TY>), grid, block, Cfg::LDS_BYTES, st, in, s.layers[layer].w,
                       s.layers[layer + 1].bn_s, s.layers[layer + 1].bn_t, out, D, h, w, tiles_x, tiles_y, /*wt=*/0, tiles_d);
    LWS_LAUNCH_CHECK();
    return LWS_OK;
}

template <int TD, int TY>
static int mid8x_launch(const Stage3d &s, int layer, const float *in, float *out, int B, int D, int h, int w, hipStream_t st)
{
    using Cfg = Mid8xCfg<TD, TY>;
    static std::atomic<uint64_t> attr_done{0};
    const int rc_ = ensure_dyn_lds(attr_done, reinterpret_cast<const void *>(&k_conv3d_mid8x<TD, TY>), Cfg::LDS_BYTES);
    if (rc_) return rc_;
    const int tiles_x = cdiv(w, 32), tiles_y = cdiv(h, TY), tiles_d = cdiv(D, TD);
    dim3 grid(tiles_x * tiles_y * tiles_d, B), block(Cfg::NT);
    hipLaunchKernelGGL((k_conv3d_mid8x<TD, TY>), grid, block, Cfg::LDS_BYTES, st, in,
                       reinterpret_cast<const uint4 *>(s.layers[layer].w + 28 * 64), s.layers[layer + 1].bn_s,
                       s.layers[layer + 1].bn_t, out, D, h, w, tiles_x, tiles_y, tiles_d);
    LWS_LAUNCH_CHECK();
    return LWS_OK;
}

// layer = 1 .. layers_3d (the C3 -> C3 convolutions)
int launch_conv3d_mid(const Stage3d &s, int layer, const float *act_in, float *act_out, int B, int D, int h,
                      int w, hipStream_t st, hipEvent_t e0, hipEvent_t e1)
{
    switch (s.c3) {
        case 8: {
            // split-bf16 (NOT bit-exact) only where its grid fills the chip; the choice depends on the per-SAMPLE geometry, never on
            // B: in this mode -- the only one whose two candidate kernels differ in bits -- a pair must get the same bits at every
            // batch size, so that the sharded / pooled / batched results stay equal to each other
            if (s.mid8_split && (long)cdiv(w, 32) * cdiv(h, 4) * cdiv(D, 3) >= 256)
                return mid8x_launch<3, 4>(s, layer, act_in, act_out, B, D, h, w, st);
            // k_conv3d_mid8q's tile: 3 x 8 x 32 voxels (12 waves, 54 KB of LDS, halo 2.21x) once there are enough of them to fill
            // the chip and they balance, else 3 x 2 x 32 (3 waves, 21.8 KB, halo 3.5x).  The staging bounds the kernel, so the
            // large tile's smaller halo wins wherever the grid is large; a launch lasts as long as its fullest CU, so a grid of
            // at most four small tiles per CU takes the small tile whenever that is the shorter schedule of the fullest CU
            // (`mid8_balance`; lws_pool workers run without it: other forwards fill the CUs).  History and numbers:
            // profiles/NOTES.md, "k_conv3d_mid8q tile choice".
            const long big_tiles = (long)cdiv(w, 32) * cdiv(h, 8) * cdiv(D, 3) * B;
            const long small_tiles = (long)cdiv(w, 32) * cdiv(h, 4) * cdiv(D, 3) * B;
            const long ncu = s.cu_count > 0 ? s.cu_count : 256;
            const long ks = (small_tiles + ncu - 1) / ncu, kl = (big_tiles + ncu - 1) / ncu;
            const bool small_wins = s.mid8_balance != 0 && ks <= 4 && ks * 384 < kl * 768;
            if (big_tiles >= 192 && !small_wins) return mid8q_launch<3, 8>(s, layer, act_in, act_out, B, D, h, w, st);
            return mid8q_launch<3, 2>(s, layer, act_in, act_out, B, D, h, w, st);
        }
        case 16: return mid16_launch<16, 3, 4, 4, 1>(s, layer, act_in, act_out, B, D, h, w, st, e0, e1);
        case 32: {
            if (s.mid16_split) return mid16x_launch<3, 4>(s, layer, act_in, act_out, B, D, h, w, st, e0, e1);
            return mid16_launch<32, 3, 4, 4, 1>(s, layer, act_in, act_out, B, D, h, w, st, e0, e1);
        }
        default: set_error("conv3d: unsupported channel count %d (8, 16, 32)", s.c3); return LWS_ERR_INVALID;
    }
}

template <int C3, int TD, int TY, int TX, bool FUSE>
static int last_launch(const Stage3d &s, const float *act, const float *skip, float *cost_out, float *low, float start,
                       int B, int D, int h, int w, hipStream_t st)
{
    using Cfg = LastCfg<C3, TD, TY, TX, FUSE>;
    static std::atomic<uint64_t> attr_done{0};
    if (Cfg::LDS_BYTES > 48 * 1024) {
        const int rc_ = ensure_dyn_lds(attr_done, reinterpret_cast<const void *>(&k_conv3d_last<C3, TD, TY, TX, FUSE>), Cfg::LDS_BYTES);
        if (rc_) return rc_;
    }
    const int tiles_x = cdiv(w, TX), tiles_y = cdiv(h, TY), tiles_d = cdiv(D, TD);
    dim3 grid(tiles_x * tiles_y * tiles_d, B), block(Cfg::NT);
    LWS_LAUNCH_STOP((k_conv3d_last<C3, TD, TY, TX, FUSE>), grid, block, Cfg::LDS_BYTES, st, act, s.layers.back().w,
                    skip, cost_out, low, start, D, h, w, tiles_x, tiles_y, tiles_d);
    LWS_LAUNCH_CHECK();
    return LWS_OK;
}

int launch_conv3d_last(const Stage3d &s, const float *act_in, const float *cost_skip, float *cost_out, int B,
                       int D, int h, int w, hipStream_t st)
{
    switch (s.c3) {
        case 8: return last_launch<8, 3, 4, 16, false>(s, act_in, cost_skip, cost_out, nullptr, 0.f, B, D, h, w, st);
        case 16: return last_launch<16, 3, 4, 16, false>(s, act_in, cost_skip, cost_out, nullptr, 0.f, B, D, h, w, st);
        case 32: return last_launch<32, 3, 4, 16, false>(s, act_in, cost_skip, cost_out, nullptr, 0.f, B, D, h, w, st);
        default: set_error("conv3d: unsupported channel count %d (8, 16, 32)", s.c3); return LWS_ERR_INVALID;
    }
}

// Last layer + soft-argmin in one launch; available when the tile can span the disparity axis.
// (C3 = 32 with D = 24 / 32 -- stage 1 of the default and of the maxdisp-256 configuration -- since round 5: a 24 x 2 x 4
// tile, 26 x 4 x 6 halo voxels = 90 KB of LDS, one workgroup per CU; one 256x512 pair is exactly 256 such tiles)
bool conv3d_last_can_fuse(const Stage3d &s, int D)
{
    return (D == 9 && (s.c3 == 8 || s.c3 == 16)) || (s.c3 == 32 && (D == 24 || D == 32));
}

int launch_conv3d_last_softargmin(const Stage3d &s, const float *act_in, const float *cost_skip, float *cost_out,
                                  float *low, float start, int B, int D, int h, int w, hipStream_t st)
{
    if (D == 9 && s.c3 == 8) return last_launch<8, 9, 2, 16, true>(s, act_in, cost_skip, cost_out, low, start, B, D, h, w, st);
    if (D == 9 && s.c3 == 16) return last_launch<16, 9, 2, 16, true>(s, act_in, cost_skip, cost_out, low, start, B, D, h, w, st);
    if (D == 24 && s.c3 == 32) return last_launch<32, 24, 2, 4, true>(s, act_in, cost_skip, cost_out, low, start, B, D, h, w, st);
    if (D == 32 && s.c3 == 32) return last_launch<32, 32, 2, 4, true>(s, act_in, cost_skip, cost_out, low, start, B, D, h, w, st);
    set_error("conv3d_last_softargmin: no fused variant for c3=%d D=%d", s.c3, D);
    return LWS_ERR_INVALID;
}

}  // namespace lws
