// C ABI of the library (include/lwsnet_hip.h): handle, state-dict ingest, BN folding / weight packing,
// workspace, and the stage loop of LWSNet.forward (/root/reference/models/models.py:115-156).
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <stdlib.h>

#include "lws_common.h"

namespace lws {

static thread_local char g_err[512] = "";
thread_local hipEvent_t tl_stop_event = nullptr;

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- the state-dict contract (mirrors lwsnet_amd/weights.py:state_dict_spec) -------------------
typedef std::map<std::string, std::vector<int64_t>> Spec;

static void spec_bn(Spec &sp, const std::string &p, int c)
{
    for (const char *s : {".weight", ".bias", "._mean", "._variance"}) sp[p + s] = {c};
}

static Spec build_spec(const lws_config &cfg)
{
    Spec sp;
    const std::string fe = "feature_extraction.";
    struct CB { const char *name; int co, ci; };
    for (CB e : {CB{"dres0.0", 4, 3}, CB{"dres0.2", 8, 4}, CB{"dres1.0", 4, 8}, CB{"dres1.2", 8, 4},
                 CB{"dres2.conv1.0", 16, 8}, CB{"dres2.conv2.0", 16, 16}, CB{"dres2.conv3.0", 16, 16},
                 CB{"dres2.conv4.0", 16, 16}, CB{"classif1.0", 8, 8}}) {
        sp[fe + e.name + ".0.weight"] = {e.co, e.ci, 3, 3};
        spec_bn(sp, fe + e.name + ".1", e.co);
    }
    sp[fe + "dres2.conv5.0.weight"] = {16, 16, 3, 3};
    spec_bn(sp, fe + "dres2.conv5.1", 16);
    sp[fe + "dres2.conv6.0.weight"] = {16, 8, 3, 3};
    spec_bn(sp, fe + "dres2.conv6.1", 8);
    sp[fe + "classif1.2.weight"] = {8, 8, 3, 3};
    for (int i = 0; i < 3; ++i) {
        const int c3 = cfg.channels_3d * cfg.growth_rate[i];
        for (int j = 0; j < cfg.layers_3d + 2; ++j) {
            const int ci = j == 0 ? 1 : c3, co = j == cfg.layers_3d + 1 ? 1 : c3;
            const std::string p = "volume_postprocess." + std::to_string(i) + "." + std::to_string(j);
            spec_bn(sp, p + ".0", ci);
            sp[p + ".2.weight"] = {co, ci, 3, 3, 3};
        }
    }
    for (const char *name : {"refinement1_left", "refinement1_disp"}) {
        const int cin = strcmp(name, "refinement1_left") == 0 ? 3 : 1;
        sp[std::string(name) + ".0.weight"] = {32, cin, 3, 3};
        for (int k = 1; k <= 4; ++k) {
            const std::string p = std::string(name) + "." + std::to_string(k);
            spec_bn(sp, p + ".0", 32);
            sp[p + ".2.weight"] = {32, 1, 3, 3};
            sp[p + ".3.weight"] = {32, 32, 1, 1};
        }
    }
    spec_bn(sp, "refinement2.0.0", 64);
    sp["refinement2.0.2.weight"] = {32, 64, 3, 3};
    for (int k = 1; k <= 4; ++k) {
        const std::string p = "refinement2." + std::to_string(k);
        spec_bn(sp, p + ".0", 32);
        sp[p + ".2.weight"] = {32, 1, 3, 3};
        sp[p + ".3.weight"] = {32, 32, 1, 1};
    }
    sp["refinement2.5.weight"] = {1, 32, 3, 3};
    return sp;
}

// Eval BatchNorm as y = fmaf(x, s, t): s = gamma / sqrt(var + eps), t = beta - mean*s, float32 steps
// (the same sequence as lwsnet_amd/weights.py:bn_scale_shift; this file is built with -ffp-contract=off).
static void fold_bn(const lws_ctx *h, const std::string &p, std::vector<float> &s, std::vector<float> &t)
{
    const std::vector<float> &g = h->host.at(p + ".weight"), &b = h->host.at(p + ".bias"),
                             &m = h->host.at(p + "._mean"), &v = h->host.at(p + "._variance");
    s.resize(g.size());
    t.resize(g.size());
    for (size_t i = 0; i < g.size(); ++i) {
        float sd = sqrtf(v[i] + 1e-5f);
        s[i] = g[i] / sd;
        float ms = m[i] * s[i];
        t[i] = b[i] - ms;
    }
}

struct WsLayout {
    size_t act_a, act_b, cost_raw, cost_out, low[3], total;   // float offsets (hot path); low[s]: stage s low-res disparity
    // 2D networks: feature-extractor maps for N = 2B images, refinement ping-pong maps
    size_t fe_o, fe_o2, fe_pre, fe_f8, fe_f4, fe_o3, fe_cls, fe_f2;
    size_t r_a, r_b, r_c;
    size_t total_all;
};

// Feature-map sizes: the stem convolution (submodules.py:118-125: k3 s2 dil2 pad2) gives ceil(H/2); the hourglass halves
// twice more (check_size guarantees ceil(H/2) % 4 == 0).  H = 8k-1 is therefore as legal as H = 8k.
static inline int half_up(int v) { return (v + 1) / 2; }

// src_index's offset for every resize of the path (lws_device_math.h): lws_config.interp_align_mode
static inline float ioff_of(const lws_ctx *h) { return h->cfg.interp_align_mode == 1 ? 0.0f : 0.5f; }

static bool stage_dims(const lws_ctx *h, int s, int H, int W, int &D, int &hh, int &ww)
{
    const int div = 4 >> s;
    hh = half_up(H) / div;
    ww = half_up(W) / div;
    D = s == 0 ? h->cfg.maxdisplist[0] : 2 * h->cfg.maxdisplist[s] - 1;
    return true;
}

static WsLayout ws_layout(const lws_ctx *h, int B, int H, int W)
{
    size_t max_act = 0, max_cost = 0;
    for (int s = 0; s < 3; ++s) {
        int D, hh, ww;
        stage_dims(h, s, H, W, D, hh, ww);
        size_t vox = (size_t)B * D * hh * ww;
        max_cost = std::max(max_cost, vox);
        max_act = std::max(max_act, vox * (size_t)h->stage[s].c3);
    }
    auto al = [](size_t n) { return (n + 63) & ~(size_t)63; };
    WsLayout L;
    L.act_a = 0;
    L.act_b = L.act_a + al(max_act);
    L.cost_raw = L.act_b + al(max_act);
    L.cost_out = L.cost_raw + al(max_cost);
    L.low[0] = L.cost_out + al(max_cost);
    const int H2 = half_up(H), W2 = half_up(W);
    const size_t N = 2 * (size_t)B, p2 = (size_t)H2 * W2, p4 = (size_t)(H2 / 2) * (W2 / 2), p8 = (size_t)(H2 / 4) * (W2 / 4);
    L.low[1] = L.low[0] + al((size_t)B * p8);
    L.low[2] = L.low[1] + al((size_t)B * p4);
    L.total = L.low[2] + al((size_t)B * p2);
    size_t o = L.total;
    auto take = [&](size_t n) { size_t r = o; o += al(n); return r; };
    L.fe_o = take(N * 8 * p2);
    L.fe_o2 = take(N * 8 * p2);
    L.fe_pre = take(N * 16 * p4);
    L.fe_f8 = take(N * 16 * p8);
    L.fe_f4 = take(N * 16 * p4);
    L.fe_o3 = take(N * 8 * p2);
    L.fe_cls = take(N * 8 * p2);
    L.fe_f2 = take(N * 8 * p2);
    L.r_a = take((size_t)B * H * W * 32);
    L.r_b = take((size_t)B * H * W * 32);
    L.r_c = take((size_t)B * H * W * 32);
    L.total_all = o;
    return L;
}

static int ensure_ws(lws_ctx *h, size_t floats)
{
    const size_t bytes = floats * sizeof(float);
    if (h->ws_bytes >= bytes) return LWS_OK;
    if (h->ws) LWS_HIP(hipFree(h->ws));
    h->ws = nullptr;
    h->ws_bytes = 0;
    LWS_HIP(hipMalloc(&h->ws, bytes));
    h->ws_bytes = bytes;
    return LWS_OK;
}

static int check_size(const lws_ctx *h, int B, int H, int W)
{
    LWS_CHECK_ARG(B >= 1, "batch must be >= 1 (got %d)", B);
    // the hourglass skip-adds (submodules.py:103,182) need ceil(H/2), ceil(W/2) divisible by 4; models.py:72 needs w/8 >= D1
    LWS_CHECK_ARG(H > 0 && W > 0 && half_up(H) % 4 == 0 && half_up(W) % 4 == 0,
                  "unsupported input size %dx%d: ceil(H/2) and ceil(W/2) must be divisible by 4", H, W);
    LWS_CHECK_ARG(half_up(W) / 4 >= h->cfg.maxdisplist[0],
                  "unsupported input size %dx%d: the 1/8 map is %d wide, must be >= maxdisplist[0] = %d", H, W,
                  half_up(W) / 4, h->cfg.maxdisplist[0]);
    return LWS_OK;
}

// ---- built-in profiler: one hipEvent pair per launch on the launch stream ---------------------
static const size_t kMaxProfRecords = 65536;

static hipEvent_t prof_event(lws_ctx *h)
{
    if (!h->evt_pool.empty()) {
        hipEvent_t e = h->evt_pool.back();
        h->evt_pool.pop_back();
        return e;
    }
    // timing events never publish data to the host: no system-scope fence when they complete
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess) return nullptr;
    return e;
}

struct ProfScope {
    lws_ctx *h;
    hipStream_t st;
    lws_prof_rec rec;
    bool live;
    ProfScope(lws_ctx *h_, int kc, hipStream_t st_) : h(h_), st(st_), live(false)
    {
        if (!((h->prof_mask >> kc) & 1u) || h->prof.size() >= kMaxProfRecords) return;
        rec.kc = kc;
        rec.t0 = prof_event(h);
        rec.t1 = prof_event(h);
        if (!rec.t0 || !rec.t1) return;
        live = hipEventRecord(rec.t0, st) == hipSuccess;
    }
    ~ProfScope()
    {
        if (!live) return;
        if (hipEventRecord(rec.t1, st) == hipSuccess) h->prof.push_back(rec);
    }
};

static void prof_clear(lws_ctx *h)
{
    for (lws_prof_rec &r : h->prof) {
        h->evt_pool.push_back(r.t0);
        h->evt_pool.push_back(r.t1);
    }
    h->prof.clear();
}

// low != nullptr: the soft-argmin is wanted too; *fused tells the caller whether it was done here
// last_stop (optional): an event that must be complete once the stack's last kernel is (stop_after = 0) or once its
// stop_after-th middle layer is -- bound to that kernel's completion signal (stop_ext, StopArm in lws_common.h) or recorded
// behind it
static int conv3d_stack(lws_ctx *h, int stage, const float *cost_in, float *cost_out, float *act_a, float *act_b,
                        int B, int D, int hh, int ww, hipStream_t st, float *low = nullptr, float start = 0.f,
                        bool *fused = nullptr, bool first_done = false, hipEvent_t last_stop = nullptr, int stop_after = 0,
                        bool stop_ext = false)
{
    const Stage3d &s = h->stage[stage];
    int rc = LWS_OK;
    if (!first_done) {     // (first_done: act_a already holds the first layer's output, see launch_shift_first)
        ProfScope p(h, LWS_KC_CONV3D_FIRST, st);
        rc = launch_conv3d_first(s, cost_in, act_a, B, D, hh, ww, st);
    }
    if (rc) return rc;
    float *src = act_a, *dst = act_b;
    for (int j = 1; j <= h->cfg.layers_3d; ++j) {
        StopArm stop(j == stop_after ? last_stop : nullptr, st, stop_ext);
        if (s.c3 != 8 && ((h->prof_mask >> LWS_KC_CONV3D_MID16) & 1u) && h->prof.size() < kMaxProfRecords) {
            // dominant kernel: timed by its own begin / end timestamps, not by events around the launch
            lws_prof_rec rec;
            rec.kc = LWS_KC_CONV3D_MID16;
            rec.t0 = prof_event(h);
            rec.t1 = prof_event(h);
            rc = launch_conv3d_mid(s, j, src, dst, B, D, hh, ww, st, rec.t0, rec.t1);
            if (rc == LWS_OK && rec.t0 && rec.t1) h->prof.push_back(rec);
        } else {
            ProfScope p(h, s.c3 == 8 ? LWS_KC_CONV3D_MID8 : LWS_KC_CONV3D_MID16, st);
            rc = launch_conv3d_mid(s, j, src, dst, B, D, hh, ww, st);
        }
        LWS_HIP(stop.finish(rc));
        if (rc) return rc;
        std::swap(src, dst);
    }
    {
        ProfScope p(h, LWS_KC_CONV3D_LAST, st);
        StopArm stop(stop_after == 0 ? last_stop : nullptr, st, stop_ext);
        if (low != nullptr && conv3d_last_can_fuse(s, D)) {
            *fused = true;
            rc = launch_conv3d_last_softargmin(s, src, cost_in, nullptr, low, start, B, D, hh, ww, st);
        } else {
            if (fused) *fused = false;
            rc = launch_conv3d_last(s, src, cost_in, cost_out, B, D, hh, ww, st);
        }
        LWS_HIP(stop.finish(rc));
    }
    return rc;
}


// ---- 2D networks: parameter slab entries and execution ---------------------------------------
struct SlabBuilder {
    std::vector<float> &slab;
    explicit SlabBuilder(std::vector<float> &s) : slab(s) {}
    size_t put(const std::vector<float> &v)
    {
        slab.resize((slab.size() + 63) & ~(size_t)63, 0.0f);
        size_t o = slab.size();
        slab.insert(slab.end(), v.begin(), v.end());
        return o;
    }
};

struct Net2dOffsets {
    struct L { size_t w, wp, wm, s, t; bool bn; };
    L fe[12];
    size_t r1_first[2], r1_first_mfma[2];
    struct D { size_t s, t, dw, pw; };
    D r1[2][4], r2[4];
    size_t r2f_s, r2f_t, r2f_w, r2f_wx, r2_last;
};

static bool have_all_2d(const lws_ctx *h)
{
    for (const auto &kv : h->spec)
        if (kv.first.compare(0, 19, "volume_postprocess.") != 0 && !h->host.count(kv.first)) return false;
    return true;
}

static const struct { const char *name; int cin, cout, stride, pad, dil; bool tr, bn, relu; } kFeLayers[12] = {
    {"dres0.0", 3, 4, 2, 2, 2, false, true, true},         {"dres0.2", 4, 8, 1, 4, 4, false, true, true},
    {"dres1.0", 8, 4, 1, 2, 2, false, true, true},         {"dres1.2", 4, 8, 1, 2, 2, false, true, false},
    {"dres2.conv1.0", 8, 16, 2, 1, 1, false, true, true},  {"dres2.conv2.0", 16, 16, 1, 1, 1, false, true, true},
    {"dres2.conv3.0", 16, 16, 2, 1, 1, false, true, true}, {"dres2.conv4.0", 16, 16, 1, 1, 1, false, true, true},
    {"dres2.conv5", 16, 16, 2, 1, 1, true, true, true},    {"dres2.conv6", 16, 8, 2, 1, 1, true, true, false},
    {"classif1.0", 8, 8, 1, 1, 1, false, true, true},      {"classif1.2", 8, 8, 1, 1, 1, false, false, false}};

static void pack_dws(const lws_ctx *h, SlabBuilder &sb, const std::string &p, Net2dOffsets::D &d)
{
    std::vector<float> s, t;
    fold_bn(h, p + ".0", s, t);
    d.s = sb.put(s);
    d.t = sb.put(t);
    const std::vector<float> &dw = h->host.at(p + ".2.weight");      // [32][1][3][3]
    std::vector<float> dwt(9 * 32);
    for (int c = 0; c < 32; ++c)
        for (int tap = 0; tap < 9; ++tap) dwt[tap * 32 + c] = dw[c * 9 + tap];
    d.dw = sb.put(dwt);
    std::vector<float> pw(2 * 2 * 64 * 4);
    pack_conv2d_mfma(h->host.at(p + ".3.weight").data(), 32, 1, pw.data());   // [32][32][1][1]
    d.pw = sb.put(pw);
}

static void build_net2d(const lws_ctx *h, std::vector<float> &slab, Net2dOffsets &o)
{
    SlabBuilder sb(slab);
    const std::string fe = "feature_extraction.";
    for (int i = 0; i < 12; ++i) {
        const auto &d = kFeLayers[i];
        const bool seq = !d.tr && d.bn;   // convbn inside Sequential: <name>.0.weight / <name>.1.*
        const std::string wkey = d.tr ? fe + d.name + ".0.weight" : (d.bn ? fe + d.name + ".0.weight" : fe + d.name + ".weight");
        (void)seq;
        {   // Conv2D [cout][cin][3][3] / Conv2DTranspose [cin][cout][3][3]  ->  [tap][wave][cin][cout/4]
            const std::vector<float> &w = h->host.at(wkey);
            const int cpt = d.cout / 4;
            std::vector<float> wt((size_t)9 * d.cin * d.cout);
            for (int co = 0; co < d.cout; ++co)
                for (int ci = 0; ci < d.cin; ++ci)
                    for (int tap = 0; tap < 9; ++tap)
                        wt[(((size_t)tap * 4 + co / cpt) * d.cin + ci) * cpt + co % cpt] =
                            d.tr ? w[((size_t)ci * d.cout + co) * 9 + tap] : w[((size_t)co * d.cin + ci) * 9 + tap];
            o.fe[i].w = sb.put(wt);
            o.fe[i].wp = 0;
            const int G = conv2d_pair_groups(i);      // second copy in the group count the pair kernel uses
            if (G > 0 && !d.tr) {
                const int cpg = d.cout / G;
                std::vector<float> wq((size_t)9 * d.cin * d.cout);
                for (int co = 0; co < d.cout; ++co)
                    for (int ci = 0; ci < d.cin; ++ci)
                        for (int tap = 0; tap < 9; ++tap)
                            wq[(((size_t)tap * G + co / cpg) * d.cin + ci) * cpg + co % cpg] = w[((size_t)co * d.cin + ci) * 9 + tap];
                o.fe[i].wp = sb.put(wq);
            }
            o.fe[i].wm = 0;
            if (i >= 4 && i <= 7) {                   // conv1..conv4: 16 output channels -> MFMA A fragments
                std::vector<float> wf((size_t)9 * 64 * (d.cin / 4));
                pack_pair_mfma(w.data(), d.cin, wf.data());
                o.fe[i].wm = sb.put(wf);
            }
        }
        o.fe[i].bn = d.bn;
        if (d.bn) {
            std::vector<float> s, t;
            fold_bn(h, fe + d.name + ".1", s, t);
            o.fe[i].s = sb.put(s);
            o.fe[i].t = sb.put(t);
        }
    }
    const char *r1n[2] = {"refinement1_left", "refinement1_disp"};
    for (int k = 0; k < 2; ++k) {
        const int cin = k == 0 ? 3 : 1;
        const std::vector<float> &w = h->host.at(std::string(r1n[k]) + ".0.weight");   // [32][cin][3][3]
        std::vector<float> wt(9 * cin * 32);
        for (int co = 0; co < 32; ++co)
            for (int ci = 0; ci < cin; ++ci)
                for (int tap = 0; tap < 9; ++tap) wt[(tap * cin + ci) * 32 + co] = w[(co * cin + ci) * 9 + tap];
        o.r1_first[k] = sb.put(wt);
        std::vector<float> wf(packed_first_mfma_floats(cin));
        pack_first_mfma(w.data(), cin, wf.data());
        o.r1_first_mfma[k] = sb.put(wf);
        for (int b = 0; b < 4; ++b) pack_dws(h, sb, std::string(r1n[k]) + "." + std::to_string(b + 1), o.r1[k][b]);
    }
    {
        std::vector<float> s, t;
        fold_bn(h, "refinement2.0.0", s, t);
        o.r2f_s = sb.put(s);
        o.r2f_t = sb.put(t);
        std::vector<float> wp(10 * 4 * 2 * 64 * 4, 0.0f);   // 9 taps + one all-zero tap (prefetch without bounds check)
        pack_conv2d_mfma(h->host.at("refinement2.0.2.weight").data(), 64, 9, wp.data());   // [32][64][3][3]
        o.r2f_w = sb.put(wp);
        std::vector<float> wx(packed_conv64x_floats(), 0.0f);
        pack_conv64_bf16x3(h->host.at("refinement2.0.2.weight").data(), wx.data());
        o.r2f_wx = sb.put(wx);
    }
    for (int b = 0; b < 4; ++b) pack_dws(h, sb, "refinement2." + std::to_string(b + 1), o.r2[b]);
    {
        const std::vector<float> &w = h->host.at("refinement2.5.weight");   // [1][32][3][3]
        std::vector<float> wt(9 * 32);
        for (int ci = 0; ci < 32; ++ci)
            for (int tap = 0; tap < 9; ++tap) wt[tap * 32 + ci] = w[ci * 9 + tap];
        o.r2_last = sb.put(wt);
    }
}

static void bind_net2d(lws_ctx *h, const Net2dOffsets &o)
{
    Net2d &n = h->net2d;
    for (int i = 0; i < 12; ++i) {
        const auto &d = kFeLayers[i];
        Conv2dLayer &l = n.fe[i];
        l.cin = d.cin; l.cout = d.cout; l.stride = d.stride; l.pad = d.pad; l.dil = d.dil;
        l.transposed = d.tr; l.relu = d.relu;
        l.w = h->params + o.fe[i].w;
        l.pair_groups = conv2d_pair_groups(i);
        l.w_pair = l.pair_groups > 0 ? h->params + o.fe[i].wp : nullptr;
        l.w_mfma = (i >= 4 && i <= 7) ? h->params + o.fe[i].wm : nullptr;
        l.bn_s = d.bn ? h->params + o.fe[i].s : nullptr;
        l.bn_t = d.bn ? h->params + o.fe[i].t : nullptr;
    }
    auto bind_dws = [&](RefDws &r, const Net2dOffsets::D &d, int dil) {
        r.dil = dil;
        r.bn_s = h->params + d.s; r.bn_t = h->params + d.t; r.dw = h->params + d.dw; r.pw = h->params + d.pw;
    };
    for (int k = 0; k < 2; ++k) {
        n.r1_first[k] = h->params + o.r1_first[k];
        n.r1_first_mfma[k] = h->params + o.r1_first_mfma[k];
        for (int b = 0; b < 4; ++b) bind_dws(n.r1[k][b], o.r1[k][b], 2 << b);      // dilation 2,4,8,16 (submodules.py:298)
    }
    n.r2_first.bn_s = h->params + o.r2f_s;
    n.r2_first.bn_t = h->params + o.r2f_t;
    n.r2_first.w = h->params + o.r2f_w;
    n.r2_first.wx = h->params + o.r2f_wx;
    for (int b = 0; b < 4; ++b) bind_dws(n.r2[b], o.r2[b], 8 >> b);                 // dilation 8,4,2,1 (submodules.py:316)
    n.r2_last = h->params + o.r2_last;
}

static int feature_tail(lws_ctx *h, int N, int H, int W, const WsLayout &L, float *f8, float *f4, float *f2,
                        hipStream_t st, hipEvent_t *ev, int part = 3);

// feature_extraction.forward (submodules.py:176-188) on N = nA + nB images read from two tensors (left batch, right batch) as
// ONE batch.  head_only: only the layers up to the 1/8 map run here (stage 1 needs nothing else); the caller runs feature_tail
// (conv5, conv6, classif1 -> f4, f2) on a side stream later, beside the volume stages.
// fork_ev (optional): complete once f8 / pre are -- bound to the last head kernel's completion signal (fork_ext) or recorded
static int feature_extraction(lws_ctx *h, const float *imgA, const float *imgB, int nA, int nB, int H, int W,
                              const WsLayout &L, float *f8, float *f4, float *f2, hipStream_t st,
                              bool head_only = false, hipEvent_t fork_ev = nullptr, bool fork_ext = false)
{
    const Net2d &n = h->net2d;
    const int N = nA + nB, H2 = half_up(H), W2 = half_up(W), H4 = H2 / 2, W4 = W2 / 2;
    float *ws = h->ws;
    float *o = ws + L.fe_o, *o2 = ws + L.fe_o2, *pre = ws + L.fe_pre;
    const float *img2 = nB > 0 ? imgB : nullptr;
    int rc;
#define LWS_FE(call)                                   \
    {                                                  \
        ProfScope p_(h, LWS_KC_FEATURE2D, st);         \
        rc = (call);                                   \
    }                                                  \
    if (rc) return rc;
    // dres0, dres1, hourglass conv1..conv4: consecutive layers run pairwise in one launch (k_conv2d_pair /
    // k_conv2d_pair_mfma, the same fma chains as one kernel per layer)
    LWS_FE(launch_conv2d_pair(n.fe[0], n.fe[1], imgA, nullptr, o, N, H, W, st, img2, nA));          // dres0
    LWS_FE(launch_conv2d_pair(n.fe[2], n.fe[3], o, o, o2, N, H2, W2, st));                           // dres1 + o (:179)
    LWS_FE(launch_conv2d_pair(n.fe[4], n.fe[5], o2, nullptr, pre, N, H2, W2, st));                   // conv1, conv2 -> pre
    {
        StopArm stop(fork_ev, st, fork_ext);
        ProfScope p_(h, LWS_KC_FEATURE2D, st);
        rc = launch_conv2d_pair(n.fe[6], n.fe[7], pre, nullptr, f8, N, H4, W4, st);                  // conv3, conv4 -> f8
        LWS_HIP(stop.finish(rc));
    }
    if (rc) return rc;
#undef LWS_FE
    if (head_only) return LWS_OK;
    return feature_tail(h, N, H, W, L, f8, f4, f2, st, nullptr);
}

// conv5 (-> f4), conv6, classif1 (-> f2), submodules.py:103-107,182-186.  ev (optional): ev[1] recorded after f4,
// ev[2] after f2, on st.  part: bit 0 = conv5 (f4), bit 1 = conv6 + classif1 (f2).
static int feature_tail(lws_ctx *h, int N, int H, int W, const WsLayout &L, float *f8, float *f4, float *f2,
                        hipStream_t st, hipEvent_t *ev, int part)
{
    const Net2d &n = h->net2d;
    const int H2 = half_up(H), W2 = half_up(W), H4 = H2 / 2, W4 = W2 / 2, H8 = H2 / 4, W8 = W2 / 4;
    float *ws = h->ws;
    float *o2 = ws + L.fe_o2, *pre = ws + L.fe_pre, *o3 = ws + L.fe_o3, *cls = ws + L.fe_cls;
    int rc;
#define LWS_FE(call)                                   \
    {                                                  \
        ProfScope p_(h, LWS_KC_FEATURE2D, st);         \
        rc = (call);                                   \
    }                                                  \
    if (rc) return rc;
    if (part & 1) {
        LWS_FE(launch_conv2d_nchw(n.fe[8], f8, pre, f4, N, H8, W8, st));                                    // relu(conv5 + pre) (:103)
        if (ev != nullptr) LWS_HIP(hipEventRecord(ev[1], st));
    }
    if (part & 2) {
        LWS_FE(launch_conv2d_nchw(n.fe[9], f4, o2, o3, N, H4, W4, st));                                     // conv6 + output (:106,:182)
        LWS_FE(launch_conv2d_nchw(n.fe[10], o3, nullptr, cls, N, H2, W2, st));                              // classif1.0
        LWS_FE(launch_conv2d_nchw(n.fe[11], cls, nullptr, f2, N, H2, W2, st));                              // classif1.2 -> f2
        if (ev != nullptr) LWS_HIP(hipEventRecord(ev[2], st));
    }
#undef LWS_FE
    return LWS_OK;
}

#define LWS_RF(kc, call)               \
    {                                  \
        ProfScope p_(h, kc, st);       \
        rc = (call);                   \
    }                                  \
    if (rc) return rc;

// refinement1_left(left) (models.py:158): depends on the left image only -> result in r_a (r_c is its scratch)
// The refinement maps are [B,H,W,32] float32 = 128 B per pixel (134 MB at 8 x 256x512) and every block reads one and writes
// another.  A chunk of pairs runs its whole layer chain before the next chunk starts, sized so that one map of the chunk is
// at most `ref_chunk_mb` MB: the three maps a block chain touches then stay in the 256 MiB Infinity Cache between a block's
// write and the next block's read (MI355X_MICROARCH.md: a table stays resident while it plus everything moved between two
// uses fits in ~256 MiB).  0 = one chunk.  Pairs are independent, so chunking cannot change a bit.
static int refine_chunk(const lws_ctx *h, int B, int H, int W)
{
    if (h->opt.ref_chunk_mb <= 0) return B;
    const double map_mb = (double)H * W * 128.0 / 1e6;
    int c = (int)((double)h->opt.ref_chunk_mb / map_mb);
    c = c < 1 ? 1 : c;
    return c < B ? c : B;
}

static int refine_left_chunk(lws_ctx *h, const float *left, int B, int H, int W, const WsLayout &L, hipStream_t st, int b0);

static int refine_left(lws_ctx *h, const float *left, int B, int H, int W, const WsLayout &L, hipStream_t st)
{
    const int CH = refine_chunk(h, B, H, W);
    for (int b0 = 0; b0 < B; b0 += CH) {
        const int rc = refine_left_chunk(h, left + (size_t)b0 * 3 * H * W, std::min(CH, B - b0), H, W, L, st, b0);
        if (rc) return rc;
    }
    return LWS_OK;
}

static int refine_left_chunk(lws_ctx *h, const float *left, int B, int H, int W, const WsLayout &L, hipStream_t st, int b0)
{
    const Net2d &n = h->net2d;
    // the result keeps its slice of r_a; the scratch map r_c is the SAME memory for every chunk (it stays cache-resident)
    float *ra = h->ws + L.r_a + (size_t)b0 * H * W * 32, *rc_ = h->ws + L.r_c;
    int rc;
    if ((h->opt.fuse_first & 2) && ref_first_dws_can_fuse(n.r1[0][0], 3)) {
        // the 3 -> 32 convolution is recomputed inside the first block's staging: one launch less, and the 32-channel map
        // it would write (and the block read back) never exists
        LWS_RF(LWS_KC_REF_DWS, launch_ref_first_dws(n.r1[0][0], left, 3, n.r1_first_mfma[0], rc_, B, H, W, st));
    } else {
        LWS_RF(LWS_KC_REF_FIRST, launch_ref_first(left, 3, n.r1_first[0], ra, B, H, W, st));
        LWS_RF(LWS_KC_REF_DWS, launch_ref_dws(n.r1[0][0], ra, rc_, B, H, W, st));
    }
    LWS_RF(LWS_KC_REF_DWS, launch_ref_dws(n.r1[0][1], rc_, ra, B, H, W, st));
    LWS_RF(LWS_KC_REF_DWS, launch_ref_dws(n.r1[0][2], ra, rc_, B, H, W, st));
    LWS_RF(LWS_KC_REF_DWS, launch_ref_dws(n.r1[0][3], rc_, ra, B, H, W, st));
    return LWS_OK;
}

// models.py:159-162: refinement1_disp(pred3), refinement2(concat), + pred3.  Needs refine_left's result in r_a.
// Deferred full-resolution maps: the fused last-layer + soft-argmin kernel of stage s (s = 1, 2) leaves the
// low-resolution disparity low[s]; instead of a k_upsample_add launch on the critical chain, the consumer of
// pred_out[s] (stage 3's warp kernel for s = 1, the refinement's first block for s = 2) evaluates
// upsample(low[s]) + pred_out[s-1] on demand (DeferredMap, lws_device_math.h: the same operations, bit for bit) and
// writes the map out as a by-product, since it is an output of the path.
struct DeferState {
    bool allow_last = false;                 // the caller will consume pred_out[2] through refine_rest
    // stage 1 (round 5, batches <= 2): the last Conv3D layer + soft-argmin leave low[0] (def[0]) and NO launch materialises
    // pred1: stage 2's warp kernel evaluates the four taps it needs from low[0]; stage 3's warp kernel, which evaluates and
    // writes the deferred stage-2 map pred2 = upsample(low[1]) + pred1 on its 2 x 2 blocks, evaluates pred1 =
    // upsample(low[0]) at the same pixels and writes it out too (two-level DeferredMap).  pred1_unwritten: nobody has yet.
    bool allow_first = false;                // the caller runs the whole forward (lws_forward)
    bool pred1_unwritten = false;
    hipEvent_t stage1_stop = nullptr;        // lws_forward's second fork: complete once stage 1's Conv3D stack is (conv3d_stack's last_stop)
    int stage1_stop_after = 0;               // ... or once its k-th middle layer is (option "fork2_after")
    bool stage1_stop_ext = false;            // bound to that kernel's completion signal (not under hipGraph capture)
    bool def[3] = {false, false, false};
    const float *low[3] = {nullptr, nullptr, nullptr};
    int lh[3] = {0, 0, 0}, lw[3] = {0, 0, 0};
};

static bool refine_can_defer(const lws_ctx *h);

static int refine_rest_chunk(lws_ctx *h, float *pred3, int B, int H, int W, const WsLayout &L, float *pred4, hipStream_t st,
                             const DeferState *ds, const float *pred2, int b0, int scratch_b0 = 0, hipEvent_t after_disp = nullptr,
                             bool fuse_last = false);

static int refine_rest(lws_ctx *h, float *pred3, int B, int H, int W, const WsLayout &L, float *pred4,
                       hipStream_t st, const DeferState *ds = nullptr, const float *pred2 = nullptr)
{
    const int CH = (ds != nullptr && ds->def[2]) ? B : refine_chunk(h, B, H, W);      // (deferred maps: batches <= 2, one chunk)
    // Option "ref_pipe": chunks alternate between the caller's stream and the side stream (idle by now), each starting once the
    // previous chunk has finished its disparity branch, so that one chunk's memory-bound blocks run beside the other's MFMA-bound
    // 64 -> 32 convolution; odd chunks use the second half of the (batch-sized) scratch maps.  Two chunks only add their
    // collisions, hence the automatic setting wants at least four.
    // refinement2[4] + refinement2[5] + pred3 in one launch (k_ref_dws_last): option "fuse_ref_last"; automatic = batch 1 only
    // (the fused launch takes as long as the two it replaces, so all it buys is one dispatch gap).
    // Numbers for both: profiles/NOTES.md, "refinement launch plan".
    const bool fuse_last = h->opt.fuse_ref_last >= 0 ? h->opt.fuse_ref_last != 0 : B <= 1;
    const int nchunks = (B + CH - 1) / CH;
    const int want = h->opt.ref_pipe >= 0 ? h->opt.ref_pipe : (nchunks >= 4 ? 1 : 0);
    const bool pipe = want != 0 && h->side != nullptr && h->opt.side_streams != 0 && nchunks >= 2 && 2 * CH <= B;
    if (pipe) {
        LWS_HIP(hipEventRecord(h->ev_fork, st));
        LWS_HIP(hipStreamWaitEvent(h->side, h->ev_fork, 0));
    }
    int k = 0;
    for (int b0 = 0; b0 < B; b0 += CH, ++k) {
        const size_t po = (size_t)b0 * H * W;
        hipStream_t cs = (pipe && (k & 1)) ? h->side : st;
        if (pipe && k > 0) LWS_HIP(hipStreamWaitEvent(cs, h->ev_feat[(k - 1) & 1], 0));
        const int rc = refine_rest_chunk(h, pred3 + po, std::min(CH, B - b0), H, W, L, pred4 + po, cs, ds,
                                         pred2 != nullptr ? pred2 + po : nullptr, b0, pipe ? (k & 1) * CH : 0,
                                         pipe ? h->ev_feat[k & 1] : nullptr, fuse_last);
        if (rc) return rc;
    }
    if (pipe) {
        LWS_HIP(hipEventRecord(h->ev_join, h->side));
        LWS_HIP(hipStreamWaitEvent(st, h->ev_join, 0));
    }
    return LWS_OK;
}

static int refine_rest_chunk(lws_ctx *h, float *pred3, int B, int H, int W, const WsLayout &L, float *pred4, hipStream_t st,
                             const DeferState *ds, const float *pred2, int b0, int scratch_b0, hipEvent_t after_disp, bool fuse_last)
{
    const Net2d &n = h->net2d;
    // r_a: this chunk's slice (refinement1_left's result); r_b, r_c: the same memory for every chunk
    float *ra = h->ws + L.r_a + (size_t)b0 * H * W * 32, *rb = h->ws + L.r_b + (size_t)scratch_b0 * H * W * 32,
          *rc_ = h->ws + L.r_c + (size_t)scratch_b0 * H * W * 32;
    int rc;
    if ((h->opt.fuse_first & 1) && ref_first_dws_can_fuse(n.r1[1][0], 1)) {
        // refinement1_disp: the 1 -> 32 convolution is recomputed inside the first block's staging (one launch less)
        if (ds != nullptr && ds->def[2]) {
            // pred3 = upsample(low[2]) + pred2 was not materialised: this kernel evaluates it and writes it out
            LWS_RF(LWS_KC_REF_DWS, launch_ref_first_dws(n.r1[1][0], pred2, 1, n.r1_first_mfma[1], rc_, B, H, W, st, ds->low[2],
                                                        ds->lh[2], ds->lw[2], pred3, ioff_of(h)));
        } else {
            LWS_RF(LWS_KC_REF_DWS, launch_ref_first_dws(n.r1[1][0], pred3, 1, n.r1_first_mfma[1], rc_, B, H, W, st));
        }
    } else {
        LWS_RF(LWS_KC_REF_FIRST, launch_ref_first(pred3, 1, n.r1_first[1], rb, B, H, W, st));
        LWS_RF(LWS_KC_REF_DWS, launch_ref_dws(n.r1[1][0], rb, rc_, B, H, W, st));
    }
    // refinement1_disp blocks 2..4 (dil 4, 8, 16; block 1 ran above): rc_ -> ... -> rb
    LWS_RF(LWS_KC_REF_DWS, launch_ref_dws(n.r1[1][1], rc_, rb, B, H, W, st));
    LWS_RF(LWS_KC_REF_DWS, launch_ref_dws(n.r1[1][2], rb, rc_, B, H, W, st));
    LWS_RF(LWS_KC_REF_DWS, launch_ref_dws(n.r1[1][3], rc_, rb, B, H, W, st));
    if (after_disp != nullptr) LWS_HIP(hipEventRecord(after_disp, st));
    LWS_RF(LWS_KC_REF_CONV64, launch_ref_conv64(n.r2_first, ra, rb, rc_, B, H, W, st));
    LWS_RF(LWS_KC_REF_DWS, launch_ref_dws(n.r2[0], rc_, ra, B, H, W, st));
    LWS_RF(LWS_KC_REF_DWS, launch_ref_dws(n.r2[1], ra, rc_, B, H, W, st));
    LWS_RF(LWS_KC_REF_DWS, launch_ref_dws(n.r2[2], rc_, ra, B, H, W, st));
    // refinement2[4] + refinement2[5] + pred3: one launch (k_ref_dws_last) or two (refine_rest decides)
    if (fuse_last && ref_dws_last_can_fuse(n.r2[3])) {
        LWS_RF(LWS_KC_REF_LAST, launch_ref_dws_last(n.r2[3], ra, n.r2_last, pred3, pred4, B, H, W, st));
        return LWS_OK;
    }
    LWS_RF(LWS_KC_REF_DWS, launch_ref_dws(n.r2[3], ra, rc_, B, H, W, st));
    LWS_RF(LWS_KC_REF_LAST, launch_ref_last(rc_, n.r2_last, pred3, pred4, B, H, W, st));
    return LWS_OK;
}
#undef LWS_RF

static bool refine_can_defer(const lws_ctx *h)
{
    return (h->opt.fuse_first & 1) && h->have_2d && ref_first_dws_can_fuse(h->net2d.r1[1][0], 1);
}

static int stages_impl(lws_ctx *h, const float *const featsL[3], const float *const featsR[3], int B, int H, int W,
                       float *const pred_out[3], const WsLayout &L, hipStream_t st, hipEvent_t *feat_ready = nullptr,
                       const std::function<int()> &after_stage1_stack = nullptr, DeferState *ds = nullptr);

}  // namespace lws

using namespace lws;

namespace lws {

// feat_ready (optional): events after which the stage-2 / stage-3 feature maps are complete (feat_ready[1], [2])
static int stages_impl(lws_ctx *h, const float *const featsL[3], const float *const featsR[3], int B, int H, int W,
                       float *const pred_out[3], const WsLayout &L, hipStream_t st, hipEvent_t *feat_ready,
                       const std::function<int()> &after_stage1_stack, DeferState *ds)
{
    int rc;
    float *act_a = h->ws + L.act_a, *act_b = h->ws + L.act_b, *raw = h->ws + L.cost_raw, *cost = h->ws + L.cost_out;
    const bool defer_up = h->opt.defer_upsample != 0;
    DeferState local;
    if (ds == nullptr) ds = &local;
    static const int feat_c[3] = {16, 16, 8};   // feature_extraction outputs, submodules.py:101,104,186
    // pred_out[0] as memory (k_upsample_add's `prev`): when stage 1 left its map deferred and the consumer that would have
    // written it out (stage 3's warp kernel on a deferred stage-2 map) is not coming, one k_upsample_add materialises it
    auto need_pred1 = [&]() -> int {
        if (!ds->pred1_unwritten) return LWS_OK;
        ds->pred1_unwritten = false;
        ProfScope p(h, LWS_KC_UPSAMPLE, st);
        return launch_upsample_add(ds->low[0], nullptr, pred_out[0], B, ds->lh[0], ds->lw[0], H, W, st, ioff_of(h));   // :145-148
    };
    for (int s = 0; s < 3; ++s) {
        int D, hh, ww;
        stage_dims(h, s, H, W, D, hh, ww);
        float *low = h->ws + L.low[s];
        if (s > 0 && feat_ready != nullptr && feat_ready[s] != nullptr) LWS_HIP(hipStreamWaitEvent(st, feat_ready[s], 0));
        bool first_done = false;
        if (s == 0 && shift_first_can_fuse(h->stage[0], feat_c[0])) {
            // stage-1 volume and the first Conv3D layer in one launch (the raw volume is still written: skip input)
            ProfScope p(h, LWS_KC_CONV3D_FIRST, st);
            rc = launch_shift_first(h->stage[0], featsL[0], featsR[0], raw, act_a, B, feat_c[0], D, hh, ww, st,
                                    h->cfg.feature_fp16 != 0);                                                   // :131
            first_done = true;
        } else if (s == 0) {
            ProfScope p(h, LWS_KC_VOLUME_SHIFT, st);
            rc = launch_volume_l1_shift(featsL[0], featsR[0], raw, B, feat_c[0], hh, ww, D, st, h->cfg.feature_fp16 != 0);   // :131
        } else if (ds->def[s - 1]) {
            // pred_out[s-1] is deferred: read it as upsample(low[s-1]) + pred_out[s-2] and (s == 2) write it out.  s == 1: stage
            // 1's map has no predecessor and is not written here; s == 2 with pred1 still unwritten: pred1 = upsample(low[0]) is
            // evaluated at the same pixels and written out as well
            const bool two = s == 2 && ds->pred1_unwritten;
            ProfScope p(h, LWS_KC_VOLUME_WARP, st);
            rc = launch_volume_l1_warp(featsL[s], featsR[s], (s == 1 || two) ? nullptr : pred_out[s - 2], raw, nullptr, B, feat_c[s],
                                       hh, ww, H, W, h->cfg.maxdisplist[s], st, h->cfg.feature_fp16 != 0, ds->low[s - 1],
                                       ds->lh[s - 1], ds->lw[s - 1], s == 2 ? pred_out[s - 1] : nullptr, h->opt.warp_form,
                                       two ? ds->low[0] : nullptr, two ? ds->lh[0] : 0, two ? ds->lw[0] : 0,
                                       two ? pred_out[0] : nullptr, ioff_of(h));
            if (two) ds->pred1_unwritten = false;
        } else {
            ProfScope p(h, LWS_KC_VOLUME_WARP, st);
            rc = launch_volume_l1_warp(featsL[s], featsR[s], pred_out[s - 1], raw, nullptr, B, feat_c[s], hh, ww, H, W,
                                       h->cfg.maxdisplist[s], st, h->cfg.feature_fp16 != 0, nullptr, 0, 0, nullptr,
                                       h->opt.warp_form, nullptr, 0, 0, nullptr, ioff_of(h));                 // :119-127
        }
        if (rc) return rc;
        const float start = s == 0 ? 0.0f : (float)(-h->cfg.maxdisplist[s] + 1);
        bool fused = false;
        // stage 1: the fused last layer only pays when the full-resolution map can leave the chain (below); otherwise the
        // k_softargmin_upsample launch does soft-argmin AND upsample in one kernel
        const bool defer_first = s == 0 && defer_up && h->opt.fuse_last1 != 0 && ds->allow_first && B <= 2 && H % 2 == 0 && W % 2 == 0;
        rc = conv3d_stack(h, s, raw, cost, act_a, act_b, B, D, hh, ww, st, (s > 0 || defer_first) ? low : nullptr, start, &fused,
                          first_done, s == 0 ? ds->stage1_stop : nullptr, s == 0 ? ds->stage1_stop_after : 0,
                          s == 0 && ds->stage1_stop_ext);                                                        // :136-138
        if (rc) return rc;
        if (s == 0 && after_stage1_stack) {
            rc = after_stage1_stack();
            if (rc) return rc;
        }
        if (s == 0 && fused) {
            ds->def[0] = true;
            ds->low[0] = low;
            ds->lh[0] = hh;
            ds->lw[0] = ww;
            ds->pred1_unwritten = true;
            continue;
        }
        const bool defer_this = fused && defer_up && B <= 2 && H % 2 == 0 && W % 2 == 0 && (s == 1 || (s == 2 && ds->allow_last));
        if (s >= 1 && !(s == 1 && defer_this)) {
            // a launch below reads pred_out[0] (s == 1), or the path is about to end (s == 2) with pred1 still unwritten
            rc = need_pred1();
            if (rc) return rc;
        }
        if (!fused) {
            // soft-argmin + rescale + upsample (+ previous stage) in one launch                             :142-148
            if (H % hh == 0 && W % ww == 0) {
                ProfScope p(h, LWS_KC_SOFTARGMIN, st);
                rc = launch_softargmin_upsample(cost, s == 0 ? nullptr : pred_out[s - 1], pred_out[s], nullptr, B, D, hh, ww,
                                                H, W, start, st, ioff_of(h));
            } else {
                // H or W = 8k-1: the resize ratio is not an integer, which the fused kernel's tile -> block map needs
                {
                    ProfScope p(h, LWS_KC_SOFTARGMIN, st);
                    rc = launch_softargmin(cost, low, B, D, hh, ww, start, st);
                }
                if (rc) return rc;
                ProfScope p(h, LWS_KC_UPSAMPLE, st);
                rc = launch_upsample_add(low, s == 0 ? nullptr : pred_out[s - 1], pred_out[s], B, hh, ww, H, W, st, ioff_of(h));
            }
            if (rc) return rc;
            continue;
        }
        // stage 2's map is consumed by stage 3's warp at exactly half resolution; stage 3's by the refinement (two launches
        // fewer on a batch-1 chain; from batch 4 up the heavier consumers cost more than the launches, so large batches keep
        // the separate k_upsample_add launches)
        // (only at exact 2x geometry: with odd H or W the four taps of a stage-3 pixel are not the 2x2 block it owns)
        if (defer_this) {
            ds->def[s] = true;
            ds->low[s] = low;
            ds->lh[s] = hh;
            ds->lw[s] = ww;
            continue;
        }
        {
            ProfScope p(h, LWS_KC_UPSAMPLE, st);
            rc = launch_upsample_add(low, s == 0 ? nullptr : pred_out[s - 1], pred_out[s], B, hh, ww, H, W, st, ioff_of(h));   // :145-156
        }
        if (rc) return rc;
    }
    return LWS_OK;
}

}  // namespace lws

// Every entry point that touches the GPU through a handle requires the calling thread's current HIP device to be the
// handle's (recorded by lws_create, or set with lws_set_option("device")): its parameter slab, workspace, streams and
// events live there, and a launch from another device would run on foreign pointers.  Checked, never switched: a C
// library that silently changes the caller's current device is worse than one that refuses.
static constexpr int kClockWgs = 64;      // workgroups of a stamped k_conv3d_mid16 launch that leave their clocks

static int check_device(const lws_ctx *h, const char *what)
{
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) {
        (void)hipGetLastError();
        cur = -1;
    }
    if (cur != h->device) {
        set_error("%s: the handle belongs to HIP device %d but the calling thread's current device is %d "
                  "(hipSetDevice(%d) first; -1 = no device)", what, h->device, cur, h->device);
        return LWS_ERR_INVALID;
    }
    return LWS_OK;
}
#define LWS_CHECK_DEVICE(h, what)              \
    do {                                       \
        int rc_dev_ = check_device((h), what); \
        if (rc_dev_) return rc_dev_;           \
    } while (0)

// side stream and cross-stream events of lws_forward: created by lws_reserve (which promises that later calls allocate
// nothing) or, for callers that never reserve, on the first forward.  (A CU-masked side stream was built, measured and removed
// in round 5: profiles/NOTES.md, "CU masks".)
static int ensure_streams(lws_ctx *h)
{
    if (h->side) return LWS_OK;
    const unsigned ef = hipEventDisableTiming | hipEventDisableSystemFence;
    LWS_HIP(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
    LWS_HIP(hipEventCreateWithFlags(&h->ev_fork, ef));
    LWS_HIP(hipEventCreateWithFlags(&h->ev_join, ef));
    for (int i = 0; i < 3; ++i) LWS_HIP(hipEventCreateWithFlags(&h->ev_feat[i], ef));
    LWS_HIP(hipEventCreateWithFlags(&h->ev_fork2, ef));
    return LWS_OK;
}

extern "C" {

int lws_abi_version(void) { return LWS_ABI_VERSION; }

const char *lws_last_error(void) { return g_err; }

int lws_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int lws_create(const lws_config *cfg, lws_handle *out)
{
    LWS_CHECK_ARG(cfg != nullptr && out != nullptr, "lws_create: null argument");
    LWS_CHECK_ARG(cfg->layers_3d >= 1 && cfg->layers_3d <= 16, "layers_3d must be in 1..16 (got %d)", cfg->layers_3d);
    LWS_CHECK_ARG(cfg->maxdisplist[0] >= 1 && cfg->maxdisplist[0] <= 64, "maxdisplist[0] must be in 1..64 (got %d)",
                  cfg->maxdisplist[0]);
    for (int i = 1; i < 3; ++i)
        LWS_CHECK_ARG(cfg->maxdisplist[i] >= 1 && 2 * cfg->maxdisplist[i] - 1 <= 64,
                      "maxdisplist[%d] must be in 1..32 (got %d)", i, cfg->maxdisplist[i]);
    for (int i = 0; i < 3; ++i) {
        const int c3 = cfg->channels_3d * cfg->growth_rate[i];
        LWS_CHECK_ARG(c3 == 8 || c3 == 16 || c3 == 32,
                      "stage %d: channels_3d*growth_rate = %d is not supported by the gfx950 kernels (8, 16, 32)", i, c3);
    }
    LWS_CHECK_ARG(cfg->feature_fp16 == 0 || cfg->feature_fp16 == 1, "feature_fp16 must be 0 or 1 (got %d)", cfg->feature_fp16);
    LWS_CHECK_ARG(cfg->interp_align_mode == 0 || cfg->interp_align_mode == 1, "interp_align_mode must be 0 or 1 (got %d)",
                  cfg->interp_align_mode);
    lws_ctx *h = new (std::nothrow) lws_ctx();
    if (!h) {
        set_error("out of host memory");
        return LWS_ERR_NOMEM;
    }
    h->cfg = *cfg;
    h->spec = build_spec(*cfg);
    for (int i = 0; i < 3; ++i) h->stage[i].c3 = cfg->channels_3d * cfg->growth_rate[i];
    if (hipGetDevice(&h->device) != hipSuccess) h->device = -1;   // no GPU: host-side calls still work
    (void)hipGetLastError();
    *out = h;
    return LWS_OK;
}

// copies the per-handle options into the per-layer structs the launchers read
static void apply_options(lws_ctx *h)
{
    for (int i = 0; i < 3; ++i) {
        h->stage[i].mid16_split = (h->opt.split_bf16 & 1) != 0;
        h->stage[i].mid8_split = (h->opt.split_bf16 & 2) != 0;
        h->stage[i].mid8_balance = h->mid8_balance;
        h->stage[i].cu_count = h->cu_count;
    }
    h->net2d.r2_first.form = (h->opt.split_bf16 & 4) ? 1 : 0;
}

static int *option_slot(lws_ctx *h, const char *name)
{
    struct { const char *name; int *slot; } tab[] = {{"fuse_first", &h->opt.fuse_first},
                                                     {"defer_upsample", &h->opt.defer_upsample},
                                                     {"side_streams", &h->opt.side_streams},
                                                     {"split_bf16", &h->opt.split_bf16},
                                                     {"ref_chunk_mb", &h->opt.ref_chunk_mb},
                                                     {"ref_pipe", &h->opt.ref_pipe},
                                                     {"warp_form", &h->opt.warp_form},
                                                     {"fuse_last1", &h->opt.fuse_last1},
                                                     {"fork2_after", &h->opt.fork2_after},
                                                     {"fuse_ref_last", &h->opt.fuse_ref_last},
                                                     {"device", &h->device}};
    for (auto &e : tab)
        if (strcmp(e.name, name) == 0) return e.slot;
    return nullptr;
}

int lws_set_option(lws_handle h, const char *name, int value)
{
    LWS_CHECK_ARG(h && name, "lws_set_option: null argument");
    int *slot = option_slot(h, name);
    LWS_CHECK_ARG(slot != nullptr, "lws_set_option: unknown option '%s'", name);
    if (strcmp(name, "split_bf16") == 0)
        // the opt-in numerics mode: a bit per MFMA convolution that has a split-bf16 form (NOT bit-exact); 7 = all of them
        LWS_CHECK_ARG(value >= 0 && value <= 7, "lws_set_option: split_bf16 is a bit mask in 0..7 (got %d)", value);
    else if (strcmp(name, "fuse_first") == 0)
        LWS_CHECK_ARG(value >= 0 && value <= 3, "lws_set_option: fuse_first is a bit mask in 0..3 (got %d)", value);
    else if (strcmp(name, "ref_pipe") == 0 || strcmp(name, "fuse_ref_last") == 0)
        LWS_CHECK_ARG(value >= -1 && value <= 1, "lws_set_option: %s is out of range (got %d)", name, value);
    else if (strcmp(name, "ref_chunk_mb") == 0)
        LWS_CHECK_ARG(value >= 0 && value <= 4096, "lws_set_option: ref_chunk_mb must be in 0..4096 (got %d)", value);
    else if (strcmp(name, "device") == 0) {
        LWS_CHECK_ARG(value >= 0, "lws_set_option: device must be >= 0 (got %d)", value);
        LWS_CHECK_ARG(h->params == nullptr && h->ws == nullptr && h->side == nullptr,
                      "lws_set_option: the device of a handle can only be changed before lws_finalize / lws_reserve "
                      "have allocated on device %d", h->device);
    }
    else if (strcmp(name, "fork2_after") == 0)
        LWS_CHECK_ARG(value >= -1 && value <= 16, "lws_set_option: fork2_after must be in -1..16 (got %d)", value);
    else
        LWS_CHECK_ARG(value == 0 || value == 1, "lws_set_option: %s must be 0 or 1 (got %d)", name, value);
    *slot = value;
    apply_options(h);
    return LWS_OK;
}

int lws_get_option(lws_handle h, const char *name, int *value)
{
    LWS_CHECK_ARG(h && name && value, "lws_get_option: null argument");
    int *slot = option_slot(h, name);
    LWS_CHECK_ARG(slot != nullptr, "lws_get_option: unknown option '%s'", name);
    *value = *slot;
    return LWS_OK;
}

int lws_profile_enable(lws_handle h, int class_mask)
{
    LWS_CHECK_ARG(h, "lws_profile_enable: null handle");
    prof_clear(h);
    h->prof_mask = h->prof_mask_cfg = (unsigned)class_mask;
    h->prof_every = 1;
    h->prof_calls = 0;
    return LWS_OK;
}

int lws_profile_sample(lws_handle h, int every_n)
{
    LWS_CHECK_ARG(h && every_n >= 1, "lws_profile_sample: every_n must be >= 1");
    h->prof_every = every_n;
    h->prof_calls = 0;
    return LWS_OK;
}

int lws_profile_read(lws_handle h, double *total_ms, int64_t *launches)
{
    LWS_CHECK_ARG(h && total_ms && launches, "lws_profile_read: null argument");
    LWS_CHECK_DEVICE(h, "lws_profile_read");
    for (int i = 0; i < LWS_KC_COUNT; ++i) {
        total_ms[i] = 0.0;
        launches[i] = 0;
    }
    for (lws_prof_rec &r : h->prof) {
        LWS_HIP(hipEventSynchronize(r.t1));
        float ms = 0.f;
        LWS_HIP(hipEventElapsedTime(&ms, r.t0, r.t1));
        total_ms[r.kc] += ms;
        launches[r.kc] += 1;
    }
    return LWS_OK;
}

int lws_profile_read_class(lws_handle h, int kernel_class, float *ms_out, int capacity, int *count)
{
    LWS_CHECK_ARG(h && count && (ms_out || capacity == 0) && capacity >= 0, "lws_profile_read_class: bad argument");
    LWS_CHECK_ARG(kernel_class >= 0 && kernel_class < LWS_KC_COUNT, "lws_profile_read_class: unknown kernel class %d", kernel_class);
    LWS_CHECK_DEVICE(h, "lws_profile_read_class");
    int n = 0;
    for (lws_prof_rec &r : h->prof) {
        if (r.kc != kernel_class) continue;
        if (n < capacity) {
            LWS_HIP(hipEventSynchronize(r.t1));
            float ms = 0.f;
            LWS_HIP(hipEventElapsedTime(&ms, r.t0, r.t1));
            ms_out[n] = ms;
        }
        ++n;
    }
    *count = n;
    return LWS_OK;
}

int lws_clock_stamp(lws_handle h, int enable)
{
    LWS_CHECK_ARG(h, "lws_clock_stamp: null handle");
    LWS_CHECK_DEVICE(h, "lws_clock_stamp");
    if (enable) {
        if (h->stage[0].c3 == 8) {
            set_error("lws_clock_stamp: stage 1 of this model has C3 = 8: no k_conv3d_mid16 launch to stamp");
            return LWS_ERR_STATE;
        }
        if (!h->clk_buf) LWS_HIP(hipMalloc(&h->clk_buf, kClockWgs * 4 * sizeof(unsigned long long)));
        LWS_HIP(hipMemset(h->clk_buf, 0, kClockWgs * 4 * sizeof(unsigned long long)));
    }
    h->stage[0].clk = enable ? h->clk_buf : nullptr;
    return LWS_OK;
}

int lws_clock_read(lws_handle h, double *ghz)
{
    LWS_CHECK_ARG(h && ghz, "lws_clock_read: null argument");
    LWS_CHECK_DEVICE(h, "lws_clock_read");
    if (!h->clk_buf) {
        set_error("lws_clock_read: lws_clock_stamp(h, 1) was never called");
        return LWS_ERR_STATE;
    }
    unsigned long long host[kClockWgs * 4];
    LWS_HIP(hipDeviceSynchronize());
    LWS_HIP(hipMemcpy(host, h->clk_buf, sizeof(host), hipMemcpyDeviceToHost));
    std::vector<double> v;
    for (int i = 0; i < kClockWgs; ++i) {
        const unsigned long long c0 = host[4 * i], r0 = host[4 * i + 1], c1 = host[4 * i + 2], r1 = host[4 * i + 3];
        if (r1 > r0 && c1 > c0) v.push_back((double)(c1 - c0) / (double)(r1 - r0) * 0.1);      // shader cycles per 10 ns tick -> GHz
    }
    if (v.empty()) {
        set_error("lws_clock_read: no stamped k_conv3d_mid16 launch since lws_clock_stamp(h, 1)");
        return LWS_ERR_STATE;
    }
    std::sort(v.begin(), v.end());
    *ghz = v[v.size() / 2];
    return LWS_OK;
}

const char *lws_kernel_class_name(int kc)
{
    static const char *names[LWS_KC_COUNT] = {"volume_l1_shift", "volume_l1_warp", "conv3d_first", "conv3d_mid16",
                                              "conv3d_mid8",     "conv3d_last",    "softargmin",   "upsample_add",
                                              "feature_conv2d",  "ref_first",      "ref_dws",      "ref_conv64",
                                              "ref_last"};
    return kc >= 0 && kc < LWS_KC_COUNT ? names[kc] : "?";
}

int lws_destroy(lws_handle h)
{
    if (!h) return LWS_OK;
    prof_clear(h);
    for (hipEvent_t e : h->evt_pool) (void)hipEventDestroy(e);
    if (h->side) {
        (void)hipStreamSynchronize(h->side);
        (void)hipStreamDestroy(h->side);
        (void)hipEventDestroy(h->ev_fork);
        (void)hipEventDestroy(h->ev_join);
        for (int i = 0; i < 3; ++i) (void)hipEventDestroy(h->ev_feat[i]);
        if (h->ev_fork2) (void)hipEventDestroy(h->ev_fork2);
    }
    if (h->clk_buf) (void)hipFree(h->clk_buf);
    if (h->params && h->owns_params) (void)hipFree(h->params);
    if (h->ws) (void)hipFree(h->ws);
    delete h;
    return LWS_OK;
}

int lws_set_tensor(lws_handle h, const char *key, const float *host, const int64_t *shape, int ndim)
{
    LWS_CHECK_ARG(h && key && host && shape && ndim >= 1 && ndim <= 5, "lws_set_tensor: bad argument");
    if (!h->owns_params) {
        set_error("lws_set_tensor: this handle is a clone and shares its source's parameters");
        return LWS_ERR_STATE;
    }
    auto it = h->spec.find(key);
    LWS_CHECK_ARG(it != h->spec.end(), "set_state_dict: unexpected key '%s'", key);
    std::vector<int64_t> shp(shape, shape + ndim);
    if (shp != it->second) {
        std::string want, got;
        for (int64_t d : it->second) want += std::to_string(d) + ",";
        for (int64_t d : shp) got += std::to_string(d) + ",";
        set_error("set_state_dict: shape mismatch for '%s': expected [%s] got [%s]", key, want.c_str(), got.c_str());
        return LWS_ERR_INVALID;
    }
    size_t n = 1;
    for (int64_t d : shp) n *= (size_t)d;
    h->host[key].assign(host, host + n);
    h->shapes[key] = shp;
    h->finalized = false;
    return LWS_OK;
}

int lws_finalize(lws_handle h)
{
    LWS_CHECK_ARG(h, "lws_finalize: null handle");
    if (!h->owns_params) {
        set_error("lws_finalize: this handle is a clone (lws_clone / lws_pool worker) and shares its source's parameters");
        return LWS_ERR_STATE;
    }
    LWS_CHECK_DEVICE(h, "lws_finalize");
    const int L = h->cfg.layers_3d + 2;
    // every hot-path tensor must be present
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < L; ++j) {
            const std::string p = "volume_postprocess." + std::to_string(i) + "." + std::to_string(j);
            for (const char *s : {".0.weight", ".0.bias", ".0._mean", ".0._variance", ".2.weight"})
                if (!h->host.count(p + s)) {
                    set_error("lws_finalize: state dict entry '%s%s' was never set", p.c_str(), s);
                    return LWS_ERR_STATE;
                }
        }
    // build one host slab, then upload
    std::vector<float> slab;
    struct Off { size_t w, s, t, wm; };
    std::vector<Off> offs[3];
    auto al = [&]() { slab.resize((slab.size() + 63) & ~(size_t)63, 0.0f); };
    for (int i = 0; i < 3; ++i) {
        const int c3 = h->stage[i].c3;
        for (int j = 0; j < L; ++j) {
            const std::string p = "volume_postprocess." + std::to_string(i) + "." + std::to_string(j);
            const std::vector<float> &w = h->host.at(p + ".2.weight");
            std::vector<float> s, t;
            fold_bn(h, p + ".0", s, t);
            Off o;
            al();
            o.w = slab.size();
            o.wm = 0;
            if (j == 0) {
                slab.insert(slab.end(), w.begin(), w.end());                    // [c3][27]
                al();                                                           // + MFMA A fragments
                o.wm = slab.size();
                if (c3 == 8) {
                    slab.resize(o.wm + 9 * 64);
                    pack_first8_weights(w.data(), slab.data() + o.wm);
                } else {
                    slab.resize(o.wm + (size_t)(c3 / 16) * 7 * 64);
                    pack_first16_weights(w.data(), c3, slab.data() + o.wm);
                }
            } else if (j == L - 1) {
                slab.resize(o.w + (size_t)27 * c3);                             // [27][c3]
                for (int ci = 0; ci < c3; ++ci)
                    for (int tap = 0; tap < 27; ++tap) slab[o.w + (size_t)tap * c3 + ci] = w[(size_t)ci * 27 + tap];
            } else {
                slab.resize(o.w + packed_mid_weight_floats(c3));
                pack_mid_weights(w.data(), c3, slab.data() + o.w);
            }
            al();
            o.s = slab.size();
            slab.insert(slab.end(), s.begin(), s.end());
            al();
            o.t = slab.size();
            slab.insert(slab.end(), t.begin(), t.end());
            offs[i].push_back(o);
        }
    }
    al();
    Net2dOffsets o2d;
    h->have_2d = have_all_2d(h);
    if (h->have_2d) build_net2d(h, slab, o2d);
    al();
    if (h->params) LWS_HIP(hipFree(h->params));
    h->params = nullptr;
    LWS_HIP(hipMalloc(&h->params, slab.size() * sizeof(float)));
    h->params_bytes = slab.size() * sizeof(float);
    LWS_HIP(hipMemcpy(h->params, slab.data(), h->params_bytes, hipMemcpyHostToDevice));
    for (int i = 0; i < 3; ++i) {
        const int c3 = h->stage[i].c3;
        h->stage[i].layers.assign(L, Conv3dLayer());
        for (int j = 0; j < L; ++j) {
            Conv3dLayer &l = h->stage[i].layers[j];
            l.cin = j == 0 ? 1 : c3;
            l.cout = j == L - 1 ? 1 : c3;
            l.w = h->params + offs[i][j].w;
            l.w_mfma = j == 0 ? h->params + offs[i][j].wm : nullptr;
            l.bn_s = h->params + offs[i][j].s;
            l.bn_t = h->params + offs[i][j].t;
        }
    }
    if (h->have_2d) bind_net2d(h, o2d);
    {
        int ncu = 0;
        if (h->device >= 0 && hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, h->device) == hipSuccess && ncu > 0)
            h->cu_count = ncu;
        (void)hipGetLastError();
    }
    apply_options(h);
    h->finalized = true;
    return LWS_OK;
}

int lws_reserve(lws_handle h, int B, int H, int W)
{
    LWS_CHECK_ARG(h, "lws_reserve: null handle");
    int rc = check_size(h, B, H, W);
    if (rc) return rc;
    LWS_CHECK_DEVICE(h, "lws_reserve");
    if (h->opt.side_streams != 0) {      // (single-stream plans never fork)
        rc = ensure_streams(h);
        if (rc) return rc;
    }
    return ensure_ws(h, ws_layout(h, B, H, W).total_all);
}

int lws_volume_l1_shift(const float *L, const float *R, float *cost, int B, int C, int h, int w, int D, void *stream)
{
    LWS_CHECK_ARG(L && R && cost, "volume_l1_shift: null pointer");
    LWS_CHECK_ARG(B >= 1 && h >= 1 && w >= 1 && D >= 1 && D <= 64, "volume_l1_shift: bad shape B=%d h=%d w=%d D=%d", B, h, w, D);
    // models.py:72: feat_l[:, :, :, i:] - feat_r[:, :, :, :-i] needs at least one column for every i < D
    LWS_CHECK_ARG(w >= D, "volume_l1_shift: width %d must be >= number of hypotheses %d", w, D);
    return launch_volume_l1_shift(L, R, cost, B, C, h, w, D, (hipStream_t)stream);
}

int lws_volume_l1_warp(const float *L, const float *R, const float *prev_disp, float *cost, float *wflow_out, int B,
                       int C, int h, int w, int H, int W, int m, void *stream)
{
    LWS_CHECK_ARG(L && R && prev_disp && cost, "volume_l1_warp: null pointer");
    LWS_CHECK_ARG(B >= 1 && h >= 1 && w >= 1 && H >= h && W >= w && m >= 1 && 2 * m - 1 <= 64,
                  "volume_l1_warp: bad shape B=%d h=%d w=%d H=%d W=%d m=%d", B, h, w, H, W, m);
    return launch_volume_l1_warp(L, R, prev_disp, cost, wflow_out, B, C, h, w, H, W, m, (hipStream_t)stream);
}

int lws_conv3d_stack(lws_handle h, int stage, const float *cost_in, float *cost_out, int B, int D, int hh, int ww,
                     void *stream)
{
    LWS_CHECK_ARG(h && cost_in && cost_out && cost_in != cost_out, "conv3d_stack: bad pointer");
    LWS_CHECK_ARG(stage >= 0 && stage < 3, "conv3d_stack: stage must be 0..2 (got %d)", stage);
    LWS_CHECK_ARG(B >= 1 && D >= 1 && hh >= 1 && ww >= 1, "conv3d_stack: bad shape");
    LWS_CHECK_DEVICE(h, "lws_conv3d_stack");
    if (!h->finalized) {
        set_error("conv3d_stack: lws_finalize has not been called");
        return LWS_ERR_STATE;
    }
    const size_t act = ((size_t)B * D * hh * ww * h->stage[stage].c3 + 63) & ~(size_t)63;
    int rc = ensure_ws(h, 2 * act);
    if (rc) return rc;
    (void)stop_event_take();
    return conv3d_stack(h, stage, cost_in, cost_out, h->ws, h->ws + act, B, D, hh, ww, (hipStream_t)stream);
}

int lws_softargmin(const float *cost, float *disp_low, int B, int D, int h, int w, float start, void *stream)
{
    LWS_CHECK_ARG(cost && disp_low, "softargmin: null pointer");
    LWS_CHECK_ARG(B >= 1 && D >= 1 && h >= 1 && w >= 1, "softargmin: bad shape");
    return launch_softargmin(cost, disp_low, B, D, h, w, start, (hipStream_t)stream);
}

int lws_upsample_add(const float *disp_low, const float *prev, float *out, int B, int h, int w, int H, int W,
                     void *stream)
{
    LWS_CHECK_ARG(disp_low && out, "upsample_add: null pointer");
    LWS_CHECK_ARG(B >= 1 && h >= 1 && w >= 1 && H >= h && W >= w, "upsample_add: bad shape");
    return launch_upsample_add(disp_low, prev, out, B, h, w, H, W, (hipStream_t)stream);
}

int lws_disparity_stages(lws_handle h, const float *const featsL[3], const float *const featsR[3], int B, int H,
                         int W, float *const pred_out[3], void *stream)
{
    LWS_CHECK_ARG(h && featsL && featsR && pred_out, "disparity_stages: null pointer");
    for (int s = 0; s < 3; ++s)
        LWS_CHECK_ARG(featsL[s] && featsR[s] && pred_out[s], "disparity_stages: null tensor for stage %d", s);
    int rc = check_size(h, B, H, W);
    if (rc) return rc;
    LWS_CHECK_DEVICE(h, "lws_disparity_stages");
    if (!h->finalized) {
        set_error("disparity_stages: lws_finalize has not been called");
        return LWS_ERR_STATE;
    }
    const WsLayout L = ws_layout(h, B, H, W);
    rc = ensure_ws(h, L.total);
    if (rc) return rc;
    (void)stop_event_take();
    return stages_impl(h, featsL, featsR, B, H, W, pred_out, L, (hipStream_t)stream);
}

int lws_feature_extraction(lws_handle h, const float *img, int N, int H, int W, float *f8, float *f4, float *f2,
                           void *stream)
{
    LWS_CHECK_ARG(h && img && f8 && f4 && f2, "feature_extraction: null pointer");
    LWS_CHECK_ARG(N >= 1 && H > 0 && W > 0 && half_up(H) % 4 == 0 && half_up(W) % 4 == 0,
                  "feature_extraction: unsupported size N=%d %dx%d (ceil(H/2), ceil(W/2) divisible by 4)", N, H, W);
    LWS_CHECK_DEVICE(h, "lws_feature_extraction");
    if (!h->finalized || !h->have_2d) {
        set_error("feature_extraction: the 2D network tensors were not all set before lws_finalize");
        return LWS_ERR_STATE;
    }
    // workspace is planned per pair: N images = ceil(N/2) pairs
    const WsLayout L = ws_layout(h, (N + 1) / 2, H, W);
    int rc = ensure_ws(h, L.total_all);
    if (rc) return rc;
    (void)stop_event_take();
    return feature_extraction(h, img, nullptr, N, 0, H, W, L, f8, f4, f2, (hipStream_t)stream);
}

int lws_refine(lws_handle h, const float *left, const float *pred3, int B, int H, int W, float *pred4, void *stream)
{
    LWS_CHECK_ARG(h && left && pred3 && pred4, "refine: null pointer");
    LWS_CHECK_ARG(B >= 1 && H > 0 && W > 0, "refine: unsupported size B=%d %dx%d", B, H, W);
    LWS_CHECK_DEVICE(h, "lws_refine");
    if (!h->finalized || !h->have_2d) {
        set_error("refine: the 2D network tensors were not all set before lws_finalize");
        return LWS_ERR_STATE;
    }
    const WsLayout L = ws_layout(h, B, H, W);
    int rc = ensure_ws(h, L.total_all);
    if (rc) return rc;
    (void)stop_event_take();
    rc = refine_left(h, left, B, H, W, L, (hipStream_t)stream);
    if (rc) return rc;
    return refine_rest(h, const_cast<float *>(pred3), B, H, W, L, pred4, (hipStream_t)stream);   // not written without a DeferState
}

int lws_forward(lws_handle h, const float *left, const float *right, int B, int H, int W, float *const pred_out[4],
                void *stream)
{
    LWS_CHECK_ARG(h && left && right && pred_out, "forward: null pointer");
    for (int s = 0; s < 4; ++s) LWS_CHECK_ARG(pred_out[s], "forward: null output for stage %d", s + 1);
    int rc = check_size(h, B, H, W);
    if (rc) return rc;
    LWS_CHECK_DEVICE(h, "lws_forward");
    if (!h->finalized || !h->have_2d) {
        set_error("forward: set_state_dict/lws_finalize must be called with the full state dict first");
        return LWS_ERR_STATE;
    }
    const WsLayout L = ws_layout(h, B, H, W);
    rc = ensure_ws(h, L.total_all);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    (void)stop_event_take();
    // Under hipGraph capture (tools/graph_pipeline.py; lws_reserve first, so that nothing allocates) the forks must be capture-
    // time records -- hipEventRecord on the capturing stream, which is what pulls the side stream into the graph; an event bound
    // to a kernel's completion signal is not one -- and nothing is profiled (timing events cannot be read back from a graph).
    bool capturing = false;
    if (st != nullptr) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        LWS_HIP(hipStreamIsCapturing(st, &cs));
        capturing = cs != hipStreamCaptureStatusNone;
    }
    // profiler sampling (lws_profile_sample): only every n-th forward call records events
    struct MaskGuard {
        lws_ctx *h;
        ~MaskGuard() { h->prof_mask = h->prof_mask_cfg; }
    } mask_guard{h};
    h->prof_mask = (!capturing && (h->prof_every <= 1 || h->prof_calls++ % (unsigned)h->prof_every == 0)) ? h->prof_mask_cfg : 0u;
    // Two independent branches run on the handle-owned side stream (speed only): the tail of the feature extractor and
    // refinement1_left, which depends on the left image only.  Option "side_streams" = 0 keeps everything on the caller's
    // stream: no forks, joins or event bubbles -- the plan lws_pool uses, where the kernels of OTHER forwards fill the CUs.
    // The plan (what was measured against what: profiles/NOTES.md, "launch plan of lws_forward"):
    //   caller's stream: feature head -> stage 1 -> stage 2 -> stage 3 -> refinement1_disp, refinement2
    //   fork 1, behind the feature head:                    conv5 (-> the 1/4 map of stage 2)
    //   fork 2, behind stage 1's last middle Conv3D layer:  conv6 + classif1 (-> the 1/2 map of stage 3), refinement1_left
    //   joins: the 1/4 map before stage 2, the 1/2 map before stage 3, refinement1_left before the refinement.
    // refinement1_left is HBM-bound work beside stages 2 and 3, whose MFMA kernels keep their weights in registers and do not
    // mind; beside stage 1 it would slow k_conv3d_mid16, which streams its weights from L2.
    const bool multi = h->opt.side_streams != 0;
    if (multi) {
        rc = ensure_streams(h);
        if (rc) return rc;
    }
    hipStream_t side = multi ? h->side : st;
    const bool ext = multi && !capturing;       // forks ride on their producer kernel's completion signal (StopArm)
    float *f8 = h->ws + L.fe_f8, *f4 = h->ws + L.fe_f4, *f2 = h->ws + L.fe_f2;
    rc = feature_extraction(h, left, right, B, B, H, W, L, f8, f4, f2, st, /*head_only=*/true,
                            multi ? h->ev_feat[0] : nullptr, ext);                                    // models.py:110-111
    if (rc) return rc;
    const size_t n2 = (size_t)B * 8 * half_up(H) * half_up(W), n4 = n2 / 2, n8 = n2 / 8;   // 8 / 16 / 16 channels
    const float *fl[3] = {f8, f4, f2};
    const float *fr[3] = {f8 + n8, f4 + n4, f2 + n2};
    if (multi) LWS_HIP(hipStreamWaitEvent(side, h->ev_feat[0], 0));
    rc = feature_tail(h, 2 * B, H, W, L, f8, f4, f2, side, multi ? h->ev_feat : nullptr, 1);          // conv5 -> f4
    if (rc) return rc;
    DeferState ds;
    ds.allow_last = refine_can_defer(h);
    ds.allow_first = true;
    // where the second fork sits: option "fork2_after" (k = behind stage 1's k-th middle layer, 0 = behind its last layer,
    // -1 = automatic: the last middle layer, which leaves every k_conv3d_mid16 launch undisturbed)
    const int L3 = h->cfg.layers_3d;
    int fork2 = h->opt.fork2_after;
    if (fork2 < 0) fork2 = L3;
    if (fork2 > L3) fork2 = 0;
    if (multi) {
        ds.stage1_stop = h->ev_fork2;
        ds.stage1_stop_after = fork2;
        ds.stage1_stop_ext = ext;
    }
    auto launch_tail = [&]() -> int {
        if (multi) LWS_HIP(hipStreamWaitEvent(side, h->ev_fork2, 0));     // (bound or recorded by conv3d_stack)
        int r2 = feature_tail(h, 2 * B, H, W, L, f8, f4, f2, side, multi ? h->ev_feat : nullptr, 2);   // conv6, classif1 -> f2
        if (r2) return r2;
        r2 = refine_left(h, left, B, H, W, L, side);                                                    // models.py:158
        if (r2) return r2;
        if (multi) LWS_HIP(hipEventRecord(h->ev_join, side));
        return LWS_OK;
    };
    hipEvent_t ready[3] = {nullptr, h->ev_feat[1], h->ev_feat[2]};        // joins: f4 before stage 2, f2 before stage 3
    rc = stages_impl(h, fl, fr, B, H, W, pred_out, L, st, multi ? ready : nullptr, launch_tail, &ds);   // :115-156
    if (rc) return rc;
    if (multi) LWS_HIP(hipStreamWaitEvent(st, h->ev_join, 0));
    return refine_rest(h, pred_out[2], B, H, W, L, pred_out[3], st, &ds, pred_out[1]);       // :159-162
}

}  // extern "C"
