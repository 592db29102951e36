// C ABI of the library (include/lwsnet_hip.h): handle, state-dict ingest, BN folding / weight packing,
// workspace, and the stage loop of LWSNet.forward (/root/reference/models/models.py:115-156).
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "lws_common.h"

namespace lws {

static thread_local char g_err[512] = "";

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
    size_t act_a, act_b, cost_raw, cost_out, low, total;   // float offsets
};

static bool stage_dims(const lws_ctx *h, int s, int H, int W, int &D, int &hh, int &ww)
{
    const int div = 8 >> s;
    hh = H / div;
    ww = W / div;
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
    L.low = L.cost_out + al(max_cost);
    L.total = L.low + al((size_t)B * (H / 2) * (W / 2));
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
    LWS_CHECK_ARG(H > 0 && W > 0 && H % 8 == 0 && W % 8 == 0,
                  "unsupported input size %dx%d: H and W must be multiples of 8 (ceil(H/2), ceil(W/2) divisible by 4)",
                  H, W);
    LWS_CHECK_ARG(W / 8 >= h->cfg.maxdisplist[0], "unsupported input size %dx%d: W/8 = %d must be >= maxdisplist[0] = %d",
                  H, W, W / 8, h->cfg.maxdisplist[0]);
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
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

struct ProfScope {
    lws_ctx *h;
    hipStream_t st;
    lws_prof_rec rec;
    bool live;
    ProfScope(lws_ctx *h_, int kc, hipStream_t st_) : h(h_), st(st_), live(false)
    {
        if (!h->prof_on || h->prof.size() >= kMaxProfRecords) return;
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

static int conv3d_stack(lws_ctx *h, int stage, const float *cost_in, float *cost_out, float *act_a, float *act_b,
                        int B, int D, int hh, int ww, hipStream_t st)
{
    const Stage3d &s = h->stage[stage];
    int rc;
    {
        ProfScope p(h, LWS_KC_CONV3D_FIRST, st);
        rc = launch_conv3d_first(s, cost_in, act_a, B, D, hh, ww, st);
    }
    if (rc) return rc;
    float *src = act_a, *dst = act_b;
    for (int j = 1; j <= h->cfg.layers_3d; ++j) {
        {
            ProfScope p(h, s.c3 == 8 ? LWS_KC_CONV3D_MID8 : LWS_KC_CONV3D_MID16, st);
            rc = launch_conv3d_mid(s, j, src, dst, B, D, hh, ww, st);
        }
        if (rc) return rc;
        std::swap(src, dst);
    }
    ProfScope p(h, LWS_KC_CONV3D_LAST, st);
    return launch_conv3d_last(s, src, cost_in, cost_out, B, D, hh, ww, st);
}

}  // namespace lws

using namespace lws;

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

int lws_profile_enable(lws_handle h, int on)
{
    LWS_CHECK_ARG(h, "lws_profile_enable: null handle");
    prof_clear(h);
    h->prof_on = on != 0;
    return LWS_OK;
}

int lws_profile_read(lws_handle h, double *total_ms, int64_t *launches)
{
    LWS_CHECK_ARG(h && total_ms && launches, "lws_profile_read: null argument");
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

const char *lws_kernel_class_name(int kc)
{
    static const char *names[LWS_KC_COUNT] = {"volume_l1_shift", "volume_l1_warp", "conv3d_first", "conv3d_mid16",
                                              "conv3d_mid8",     "conv3d_last",    "softargmin",   "upsample_add"};
    return kc >= 0 && kc < LWS_KC_COUNT ? names[kc] : "?";
}

int lws_destroy(lws_handle h)
{
    if (!h) return LWS_OK;
    prof_clear(h);
    for (hipEvent_t e : h->evt_pool) (void)hipEventDestroy(e);
    if (h->params) (void)hipFree(h->params);
    if (h->ws) (void)hipFree(h->ws);
    delete h;
    return LWS_OK;
}

int lws_set_tensor(lws_handle h, const char *key, const float *host, const int64_t *shape, int ndim)
{
    LWS_CHECK_ARG(h && key && host && shape && ndim >= 1 && ndim <= 5, "lws_set_tensor: bad argument");
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
    struct Off { size_t w, s, t; };
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
            if (j == 0) {
                slab.insert(slab.end(), w.begin(), w.end());                    // [c3][27]
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
            l.bn_s = h->params + offs[i][j].s;
            l.bn_t = h->params + offs[i][j].t;
        }
    }
    h->finalized = true;
    return LWS_OK;
}

int lws_reserve(lws_handle h, int B, int H, int W)
{
    LWS_CHECK_ARG(h, "lws_reserve: null handle");
    int rc = check_size(h, B, H, W);
    if (rc) return rc;
    return ensure_ws(h, ws_layout(h, B, H, W).total);
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
    if (!h->finalized) {
        set_error("conv3d_stack: lws_finalize has not been called");
        return LWS_ERR_STATE;
    }
    const size_t act = ((size_t)B * D * hh * ww * h->stage[stage].c3 + 63) & ~(size_t)63;
    int rc = ensure_ws(h, 2 * act);
    if (rc) return rc;
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
    if (!h->finalized) {
        set_error("disparity_stages: lws_finalize has not been called");
        return LWS_ERR_STATE;
    }
    const WsLayout L = ws_layout(h, B, H, W);
    rc = ensure_ws(h, L.total);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    float *act_a = h->ws + L.act_a, *act_b = h->ws + L.act_b, *raw = h->ws + L.cost_raw, *cost = h->ws + L.cost_out,
          *low = h->ws + L.low;
    static const int feat_c[3] = {16, 16, 8};   // feature_extraction outputs, submodules.py:101,104,186
    for (int s = 0; s < 3; ++s) {
        int D, hh, ww;
        stage_dims(h, s, H, W, D, hh, ww);
        if (s == 0) {
            ProfScope p(h, LWS_KC_VOLUME_SHIFT, st);
            rc = launch_volume_l1_shift(featsL[0], featsR[0], raw, B, feat_c[0], hh, ww, D, st);            // :131
        } else {
            ProfScope p(h, LWS_KC_VOLUME_WARP, st);
            rc = launch_volume_l1_warp(featsL[s], featsR[s], pred_out[s - 1], raw, nullptr, B, feat_c[s], hh, ww, H, W,
                                       h->cfg.maxdisplist[s], st);                                           // :119-127
        }
        if (rc) return rc;
        rc = conv3d_stack(h, s, raw, cost, act_a, act_b, B, D, hh, ww, st);                                  // :136-138
        if (rc) return rc;
        const float start = s == 0 ? 0.0f : (float)(-h->cfg.maxdisplist[s] + 1);
        {
            ProfScope p(h, LWS_KC_SOFTARGMIN, st);
            rc = launch_softargmin(cost, low, B, D, hh, ww, start, st);                                      // :142,151
        }
        if (rc) return rc;
        {
            ProfScope p(h, LWS_KC_UPSAMPLE, st);
            rc = launch_upsample_add(low, s == 0 ? nullptr : pred_out[s - 1], pred_out[s], B, hh, ww, H, W, st);   // :145-156
        }
        if (rc) return rc;
    }
    return LWS_OK;
}

}  // extern "C"
