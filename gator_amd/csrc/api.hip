// C-ABI entry points of libgator_hip.so (include/gator_hip.h): context creation from a reference-layout
// state_dict, stage dispatch, taps.  The arithmetic lives in basic_kernels.hip (bring-up) and fused_*.hip.
#include <hip/hip_runtime.h>

#include <cstring>

#include "internal.h"

namespace gator {

static thread_local char g_err[1024] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

StageTimer::StageTimer(gator_ctx* c_, const char* name, void* stream_) : c(c_), stream(stream_), idx(-1) {
    if (!c->profiling) return;
    hipEvent_t a = nullptr, b = nullptr;
    if (c->ev_pool.size() >= 2) {           // events are pooled: creating them per stage cost ~4 % of a 1.5 ms step
        a = (hipEvent_t)c->ev_pool.back(); c->ev_pool.pop_back();
        b = (hipEvent_t)c->ev_pool.back(); c->ev_pool.pop_back();
    } else if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
        return;
    }
    (void)hipEventRecord(a, (hipStream_t)stream);
    idx = (int)c->prof.size();
    c->prof.push_back({name, a, b});
}
StageTimer::~StageTimer() {
    if (idx >= 0) (void)hipEventRecord((hipEvent_t)c->prof[idx].stop, (hipStream_t)stream);
}

namespace {

struct Resolver {
    gator_ctx* c;
    int rc = GATOR_OK;
    const void* get(const std::string& name, int dtype, std::initializer_list<int64_t> shape) {
        if (rc != GATOR_OK) return nullptr;
        auto it = c->t.find(name);
        if (it == c->t.end()) {
            rc = fail(GATOR_EMISSING, "gator_create: tensor '%s' missing from the state_dict", name.c_str());
            return nullptr;
        }
        const TensorRef& r = it->second;
        bool ok = r.dtype == dtype && r.ndim == (int)shape.size();
        int i = 0;
        for (int64_t s : shape) ok = ok && (i < 4) && r.shape[i++] == s;
        if (!ok) {
            std::string want;
            for (int64_t s : shape) want += std::to_string(s) + ",";
            std::string got;
            for (int k = 0; k < r.ndim; ++k) got += std::to_string(r.shape[k]) + ",";
            rc = fail(GATOR_ESHAPE, "gator_create: tensor '%s' has dtype %d shape [%s], expected dtype %d shape [%s]",
                      name.c_str(), r.dtype, got.c_str(), dtype, want.c_str());
            return nullptr;
        }
        return r.data;
    }
    const float* f(const std::string& n, std::initializer_list<int64_t> s) { return (const float*)get(n, GATOR_F32, s); }
};

int resolve_weights(gator_ctx* c) {
    Resolver R{c};
    Weights& w = c->w;
    const int64_t J = c->J, D = c->D;
    const std::string g = c->prefix_gat, m = c->prefix_mdr;
    if (c->parts & GATOR_PART_GAT) {
    w.graph_adj = R.f(g + "graph_adj", {J, J});
    w.gl0_W = R.f(g + "GLinear.0.W", {64, 2});
    w.gl0_b = R.f(g + "GLinear.0.b", {64});
    w.gn_w = R.f(g + "GLinear.1.weight", {64});
    w.gn_b = R.f(g + "GLinear.1.bias", {64});
    w.gl3_W = R.f(g + "GLinear.3.W", {128, 64});
    w.gl3_b = R.f(g + "GLinear.3.b", {128});
    w.pos_id = R.f(g + "pos_id_embed.weight", {J + 1, 128});
    w.pos_num = R.f(g + "pos_num_embed.weight", {J, 128});
    w.hp_W = R.f(g + "get_hop_path_encoding.W", {8, J, J, D});
    w.hp_emb = R.f(g + "get_hop_path_encoding.spatial_pos_encoder.weight", {10, 8});
    w.hp_ew = R.f(g + "get_hop_path_encoding.edge_encoder.weight", {8 * J * J, J * J});
    w.hp_eb = R.f(g + "get_hop_path_encoding.edge_encoder.bias", {8 * J * J});
    w.norm_w = R.f(g + "norm.weight", {128});
    w.norm_b = R.f(g + "norm.bias", {128});
    w.lifter_w = R.f(g + "lifter.weight", {3 * J, 128 * J});
    w.lifter_b = R.f(g + "lifter.bias", {3 * J});
    for (int i = 0; i < kDepth; ++i) {
        const std::string b = g + "blocks." + std::to_string(i) + ".";
        GatBlockW& k = w.blk[i];
        k.n1w = R.f(b + "norm1.weight", {128});
        k.n1b = R.f(b + "norm1.bias", {128});
        k.qkv_w = R.f(b + "attn.qkv.weight", {384, 128});
        k.qkv_b = R.f(b + "attn.qkv.bias", {384});
        k.proj_w = R.f(b + "attn.proj.weight", {128, 128});
        k.proj_b = R.f(b + "attn.proj.bias", {128});
        k.gcn_W = R.f(b + "gcn.W", {2, 128, 128});
        k.gcn_M = R.f(b + "gcn.M", {J, 128});
        k.gcn_adj2 = R.f(b + "gcn.adj2", {J, J});
        k.gcn_bias = R.f(b + "gcn.bias", {128});
        k.xl0_w = R.f(b + "x_feat.linears.0.weight", {128, 128});
        k.xl0_b = R.f(b + "x_feat.linears.0.bias", {128});
        k.xl1_w = R.f(b + "x_feat.linears.1.weight", {16, 128});
        k.xl1_b = R.f(b + "x_feat.linears.1.bias", {16});
        k.xlb_w = R.f(b + "x_feat.linearback.weight", {128, 144});
        k.xlb_b = R.f(b + "x_feat.linearback.bias", {128});
        k.n2w = R.f(b + "norm2.weight", {128});
        k.n2b = R.f(b + "norm2.bias", {128});
        k.fc1_w = R.f(b + "mlp.fc1.weight", {512, 128});
        k.fc1_b = R.f(b + "mlp.fc1.bias", {512});
        k.fc2_w = R.f(b + "mlp.fc2.weight", {128, 512});
        k.fc2_b = R.f(b + "mlp.fc2.bias", {128});
    }
    w.sp = (const int64_t*)R.get("const.shortest_path", GATOR_I64, {J, J});
    w.edge_input = R.f("const.edge_input", {J, J, D});
    }
    if (!(c->parts & GATOR_PART_MDR)) return R.rc;
    w.v431 = R.f(m + "init_vertices", {kV, 3});
    w.v6890 = R.f(m + "init_vertices_6890", {kNV, 3});
    w.pos_j = R.f(m + "pos_j_id_embed.weight", {J + 1, 64});
    w.pos_v = R.f(m + "pos_v_id_embed.weight", {kV + 1, 64});
    w.jfeat_w = R.f(m + "get_joint_feature.weight", {64, 133});
    w.jfeat_b = R.f(m + "get_joint_feature.bias", {64});
    w.vfeat_w = R.f(m + "get_verts_feature.weight", {64, 6});
    w.vfeat_b = R.f(m + "get_verts_feature.bias", {64});
    const char* sfx[3] = {"", "_1", "_2"};
    for (int i = 0; i < 3; ++i) {
        const std::string e = m + "encoder" + sfx[i] + ".", sa = m + "selfatt" + sfx[i] + ".linears.", nm = m + "norm" + sfx[i] + ".";
        MdrLayerW& k = w.lay[i];
        k.n1w = R.f(e + "norm1.weight", {64});
        k.n1b = R.f(e + "norm1.bias", {64});
        k.wq = R.f(e + "attn.wq.weight", {64, 64});
        k.wk = R.f(e + "attn.wk.weight", {64, 64});
        k.wv = R.f(e + "attn.wv.weight", {64, 64});
        k.proj_w = R.f(e + "attn.proj.weight", {64, 64});
        k.proj_b = R.f(e + "attn.proj.bias", {64});
        k.n2w = R.f(e + "norm2.weight", {64});
        k.n2b = R.f(e + "norm2.bias", {64});
        k.fc1_w = R.f(e + "mlp.fc1.weight", {256, 64});
        k.fc1_b = R.f(e + "mlp.fc1.bias", {256});
        k.fc2_w = R.f(e + "mlp.fc2.weight", {64, 256});
        k.fc2_b = R.f(e + "mlp.fc2.bias", {64});
        k.a2 = R.f(nm + "a_2", {64});
        k.b2 = R.f(nm + "b_2", {64});
        for (int q = 0; q < 4; ++q) {
            k.sa_w[q] = R.f(sa + std::to_string(q) + ".weight", {64, 64});
            k.sa_b[q] = R.f(sa + std::to_string(q) + ".bias", {64});
        }
    }
    w.motion_w = R.f(m + "motion_linear.weight", {23, 64});
    w.motion_b = R.f(m + "motion_linear.bias", {23});
    w.biasl_w = R.f(m + "bias_linear.weight", {3, 64});
    w.biasl_b = R.f(m + "bias_linear.bias", {3});
    if (c->alpha) {   // lib/models/MDR.py:115-117
        w.bn_w = R.f(m + "bias_norm.weight", {3});
        w.bn_b = R.f(m + "bias_norm.bias", {3});
        w.scale_w = R.f(m + "scale_linear.weight", {1, 64});
        w.scale_b = R.f(m + "scale_linear.bias", {1});
        w.bn_mean = w.bn_var = nullptr;
    } else {          // lib/models/MDR.py:119
        w.bn_w = R.f(m + "bias_norm.weight", {kV});
        w.bn_b = R.f(m + "bias_norm.bias", {kV});
        w.bn_mean = R.f(m + "bias_norm.running_mean", {kV});
        w.bn_var = R.f(m + "bias_norm.running_var", {kV});
        w.scale_w = w.scale_b = nullptr;
    }
    w.bconv_w = R.f(m + "bias_conv1d.weight", {20, kV, 3});
    w.bconv_b = R.f(m + "bias_conv1d.bias", {20});
    w.up_w = R.f(m + "upsample_conv.weight", {kNV, kV, 3});
    w.up_b = R.f(m + "upsample_conv.bias", {kNV});
    w.vj = (const int32_t*)R.get("const.vj_relation", GATOR_I32, {kV});
    return R.rc;
}

size_t dtype_size(int dt) { return dt == GATOR_I64 ? 8 : 4; }

}  // namespace
}  // namespace gator

using namespace gator;

extern "C" const char* gator_last_error(void) { return g_err; }
extern "C" const char* gator_version(void) { return "gator-amd 0.2 (gfx950)"; }
extern "C" int gator_abi_version(void) { return GATOR_ABI_VERSION; }

static void prof_clear(gator_ctx* c, bool destroy);

extern "C" int gator_destroy(gator_ctx* c) {
    if (!c) return GATOR_OK;
    fused_destroy(c);
    prof_clear(c, true);
    if (c->ws) (void)hipFree(c->ws);
    if (c->arena) (void)hipFree(c->arena);
    if (c->status_host) (void)hipHostFree(c->status_host);
    delete c;
    return GATOR_OK;
}

extern "C" int gator_create(const gator_tensor* tensors, int32_t n, const gator_config* cfg_in, gator_ctx** out) {
    if (!tensors || n <= 0 || !cfg_in || !out) return fail(GATOR_EINVAL, "gator_create: null argument");
    // ABI 2: the caller states the size of ITS gator_config; later fields read as 0.  (An ABI-1 caller's first field is num_joint = 17 / 19.)
    const int32_t csz = cfg_in->struct_size;
    if (csz < 8 || csz > 1024 || (csz & 3))
        return fail(GATOR_EINVAL, "gator_create: gator_config.struct_size = %d; set it to sizeof(gator_config) (ABI %d: struct_size is the first field)", csz, GATOR_ABI_VERSION);
    gator_config cfg_local;
    memset(&cfg_local, 0, sizeof(cfg_local));
    memcpy(&cfg_local, cfg_in, std::min((size_t)csz, sizeof(cfg_local)));
    const gator_config* cfg = &cfg_local;
    if (cfg->num_joint != 17 && cfg->num_joint != 19)
        return fail(GATOR_EUNSUPPORTED, "gator_create: num_joint must be 17 or 19 (reference: lib/models/GAT.py:79-93), got %d", cfg->num_joint);
    if (cfg->arithmetic != GATOR_ARITH_DEFAULT && cfg->arithmetic != GATOR_ARITH_EXACT_SPLIT)
        return fail(GATOR_EINVAL, "gator_create: arithmetic must be GATOR_ARITH_DEFAULT (0) or GATOR_ARITH_EXACT_SPLIT (1), got %d", cfg->arithmetic);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(GATOR_EHIP, "gator_create: no HIP device available");
    gator_ctx* c = new gator_ctx();
    c->arithmetic = cfg->arithmetic;
    c->J = cfg->num_joint;
    c->alpha = cfg->alpha ? 1 : 0;
    c->impl = cfg->impl;
    c->subbatch_streams = cfg->subbatch_streams;
    c->parts = cfg->parts ? cfg->parts : (GATOR_PART_GAT | GATOR_PART_MDR);
    // a stand-alone GAT / MDR module has un-prefixed keys (its own state_dict), GATOR prefixes them
    const bool both = c->parts == (GATOR_PART_GAT | GATOR_PART_MDR);
    c->prefix_gat = both ? "pose_lifter." : "";
    c->prefix_mdr = both ? "pose2mesh." : "";
    (void)hipGetDevice(&c->device);
    {   // sticky device status word: pinned, device-visible host memory (no synchronisation needed to read it)
        void *hp = nullptr, *dp = nullptr;
        if (hipHostMalloc(&hp, 64, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess) {
            if (hp) (void)hipHostFree(hp);
            delete c;
            return fail(GATOR_ENOMEM, "gator_create: hipHostMalloc of the status word failed");
        }
        c->status_host = (unsigned*)hp;
        c->status_dev = (unsigned*)dp;
        *c->status_host = 0u;
    }
    // arena: one device allocation holding a private copy of every tensor (256-B aligned), + folded constants
    size_t total = 0;
    for (int i = 0; i < n; ++i) {
        const gator_tensor& t = tensors[i];
        if (!t.name || !t.data || t.ndim < 0 || t.ndim > 4) { delete c; return fail(GATOR_EINVAL, "gator_create: bad tensor #%d", i); }
        int64_t ne = 1;
        for (int k = 0; k < t.ndim; ++k) ne *= t.shape[k];
        total += ((size_t)ne * dtype_size(t.dtype) + 255) & ~(size_t)255;
    }
    const int J = c->J;
    const size_t folded = ((size_t)kH * J * J + kDepth * J + (size_t)kDepth * J * J + 2 * J * J + (size_t)J * kC) * sizeof(float) + 6 * 256;
    c->arena_bytes = total + folded;
    if (hipMalloc(&c->arena, c->arena_bytes) != hipSuccess) { delete c; return fail(GATOR_ENOMEM, "gator_create: hipMalloc(%zu) failed", c->arena_bytes); }
    size_t off = 0;
    for (int i = 0; i < n; ++i) {
        const gator_tensor& t = tensors[i];
        TensorRef r;
        r.dtype = t.dtype;
        r.ndim = t.ndim;
        r.numel = 1;
        for (int k = 0; k < t.ndim; ++k) { r.shape[k] = t.shape[k]; r.numel *= t.shape[k]; }
        const size_t bytes = (size_t)r.numel * dtype_size(t.dtype);
        r.data = c->arena + off;
        if (bytes) {
            hipError_t e = hipMemcpy(c->arena + off, t.data, bytes, t.is_host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice);
            if (e != hipSuccess) { int rc = fail(GATOR_EHIP, "gator_create: copying '%s' failed: %s", t.name, hipGetErrorString(e)); gator_destroy(c); return rc; }
        }
        off += (bytes + 255) & ~(size_t)255;
        c->t[t.name] = r;
    }
    auto carve = [&](size_t nfloat) { float* p = (float*)(c->arena + off); off += (nfloat * sizeof(float) + 255) & ~(size_t)255; return p; };
    c->hop_bias = carve((size_t)kH * J * J);
    c->adj_diag = carve((size_t)kDepth * J);
    c->adj_off = carve((size_t)kDepth * J * J);
    c->mask1 = carve((size_t)J * J);
    c->mask2 = carve((size_t)J * J);
    c->pos_embed = carve((size_t)J * kC);
    if (c->parts & GATOR_PART_GAT) {
        auto it = c->t.find("const.edge_input");
        if (it == c->t.end() || it->second.ndim != 3) { gator_destroy(c); return fail(GATOR_EMISSING, "gator_create: 'const.edge_input' [J,J,D] missing"); }
        c->D = (int)it->second.shape[2];
    }
    int rc = resolve_weights(c);
    if (rc == GATOR_OK && (c->parts & GATOR_PART_GAT)) rc = basic_fold_constants(c, nullptr);
    if (rc == GATOR_OK && cfg->max_batch > 0 && c->impl == GATOR_IMPL_BASIC) rc = ensure_workspace(c, cfg->max_batch);
    if (rc == GATOR_OK) rc = fused_create(c, nullptr);
    if (rc != GATOR_OK) { gator_destroy(c); return rc; }
    *out = c;
    return GATOR_OK;
}

// What a kernel left in the ctx's sticky status word (an EARLIER call's failure: nothing here synchronises).  Reported once, then cleared.
// `executed`: the report rides on an entry point that has queued its own work normally (finish_fwd): the caller's buffers are being written.
static int report_device_status(gator_ctx* c, unsigned st, const char* fn, bool executed) {
    const int code = executed ? GATOR_EDEVICE_DEFERRED : GATOR_EDEVICE;
    if (st) c->status_reason = (int)st;
    const char* tail = executed ? "  THIS call was queued normally; its outputs are valid unless the next call reports again." : "";
    if (st == DEV_PERSIST_INCOMPLETE) {
        fused_disable_persist(c);        // e.g. an XCD without workgroups (CU mask): its queue is never served.  The four-launch form has no such dependency.
        return fail(code, "%s: an earlier forward on this ctx did not complete its persistent MDR launch (a sample's stage tiles were never "
                                   "finished: an XCD without workgroups, or the hang guard); the vertices of that forward are NaN.  This ctx now uses the "
                                   "four-launch form of the MDR stages (same results); GATOR_MDR_PERSIST=0 selects it from the start.%s", fn, tail);
    }
    if (st == DEV_NONFINITE)
        return fail(code, "%s: an earlier forward on this ctx produced non-finite or out-of-range coarse vertices.  Either its input poses "
                                   "were not finite (the reference returns NaN for those too), or the weights drive an activation out of the default "
                                   "arithmetic's range (|vert431| must stay below 4094 m, and every activation that feeds a token-wise linear below "
                                   "4094: they travel as fp16 planes of 16 x value); a ctx created with gator_config.arithmetic = GATOR_ARITH_EXACT_SPLIT "
                                   "(model.arithmetic = 'exact') runs the bf16 forms without that range limit.%s", fn, tail);
    return GATOR_OK;
}
static unsigned take_status_word(gator_ctx* c) { return c->status_host ? __atomic_exchange_n(c->status_host, 0u, __ATOMIC_RELAXED) : 0u; }

extern "C" int gator_status_reason(gator_ctx* c) { return c ? c->status_reason : 0; }
extern "C" int gator_c3_state(gator_ctx* c, float* logit_bound) {
    if (!c) return fail(GATOR_EINVAL, "gator_c3_state: null ctx");
    return fused_c3_state(c, logit_bound);
}

extern "C" int gator_device_status(gator_ctx* c, int32_t sync) {
    if (!c) return fail(GATOR_EINVAL, "gator_device_status: null ctx");
    if (sync) {
        GATOR_HIP_CHECK(hipSetDevice(c->device));
        GATOR_HIP_CHECK(hipDeviceSynchronize());
    }
    unsigned st = c->deferred_status ? c->deferred_status : take_status_word(c);
    c->deferred_status = 0;
    return report_device_status(c, st, "gator_device_status", false);
}

// Entry of every forward.  An earlier call's device status does NOT stop this one (round-4 advice: a NaN input pose, for which the
// reference just returns NaN, must not make the next unrelated forward a no-op with stale output buffers, nor keep one rank of a
// sharded run out of its collective): the word is taken here -- the persistent-launch failure already switches the ctx to four
// launches for THIS call -- the call runs, and finish_fwd returns GATOR_EDEVICE_DEFERRED once it is queued.
static int check_fwd(gator_ctx* c, const void* a, const void* b, int B, const char* fn) {
    if (!c || !a || !b || B <= 0) return fail(GATOR_EINVAL, "%s: null pointer or batch <= 0", fn);
    if (const unsigned st = take_status_word(c)) {
        c->deferred_status = st;
        if (st == DEV_PERSIST_INCOMPLETE) fused_disable_persist(c);
    }
    c->profiling = c->prof_stride > 0 && (c->prof_calls++ % c->prof_stride) == 0;
    GATOR_HIP_CHECK(hipSetDevice(c->device));
    c->clear_taps();        // a tap never outlives the forward that produced it (workspaces may be re-allocated below)
    return c->impl == GATOR_IMPL_BASIC ? ensure_workspace(c, B) : GATOR_OK;     // the fused entry points size their own workspace
}
static int finish_fwd(gator_ctx* c, int rc, const char* fn) {
    if (rc != GATOR_OK || !c || !c->deferred_status) return rc;
    const unsigned st = c->deferred_status;
    c->deferred_status = 0;
    return report_device_status(c, st, fn, true);
}

static int gat_forward_f32_impl(gator_ctx* c, const float* pose2d, int32_t B, float* x_out, float* feat, void* stream) {
    int rc = check_fwd(c, pose2d, x_out, B, "gator_gat_forward_f32");
    if (rc) return rc;
    if (!feat) return fail(GATOR_EINVAL, "gator_gat_forward_f32: feat is null");
    if (!(c->parts & GATOR_PART_GAT)) return fail(GATOR_EUNSUPPORTED, "gator_gat_forward_f32: ctx was created without the GAT weights");
    c->last_batch = B;
    c->clear_taps();
    rc = c->impl == GATOR_IMPL_BASIC ? basic_gat_forward(c, pose2d, B, x_out, feat, stream) : fused_gat_forward(c, pose2d, B, x_out, feat, stream);
    if (rc == GATOR_OK) c->set_tap(TAP_FEAT, feat, (int64_t)B * c->J * kC);
    return rc;
}
extern "C" int gator_gat_forward_f32(gator_ctx* c, const float* pose2d, int32_t B, float* x_out, float* feat, void* stream) { return finish_fwd(c, gat_forward_f32_impl(c, pose2d, B, x_out, feat, stream), "gator_gat_forward_f32"); }

static int mdr_forward_f32_impl(gator_ctx* c, const float* pc, int32_t B, float* verts, void* stream) {
    int rc = check_fwd(c, pc, verts, B, "gator_mdr_forward_f32");
    if (rc) return rc;
    if (!(c->parts & GATOR_PART_MDR)) return fail(GATOR_EUNSUPPORTED, "gator_mdr_forward_f32: ctx was created without the MDR weights");
    c->last_batch = B;
    return c->impl == GATOR_IMPL_BASIC ? basic_mdr_forward(c, pc, B, verts, stream) : fused_mdr_forward(c, pc, B, verts, stream);
}
extern "C" int gator_mdr_forward_f32(gator_ctx* c, const float* pc, int32_t B, float* verts, void* stream) { return finish_fwd(c, mdr_forward_f32_impl(c, pc, B, verts, stream), "gator_mdr_forward_f32"); }

static int upsample_f32_impl(gator_ctx* c, const float* vert431, int32_t B, float* verts, void* stream) {
    int rc = check_fwd(c, vert431, verts, B, "gator_upsample_f32");
    if (rc) return rc;
    if (!(c->parts & GATOR_PART_MDR)) return fail(GATOR_EUNSUPPORTED, "gator_upsample_f32: ctx was created without the MDR weights");
    return c->impl == GATOR_IMPL_BASIC ? basic_upsample(c, vert431, B, verts, stream) : fused_upsample(c, vert431, B, verts, stream);
}
extern "C" int gator_upsample_f32(gator_ctx* c, const float* vert431, int32_t B, float* verts, void* stream) { return finish_fwd(c, upsample_f32_impl(c, vert431, B, verts, stream), "gator_upsample_f32"); }

static int forward_f32_impl(gator_ctx* c, const float* pose2d, int32_t B, float* verts, float* pose3d, void* stream) {
    int rc = check_fwd(c, pose2d, verts, B, "gator_forward_f32");
    if (rc) return rc;
    if (!pose3d) return fail(GATOR_EINVAL, "gator_forward_f32: pose3d is null");
    if (c->parts != (GATOR_PART_GAT | GATOR_PART_MDR)) return fail(GATOR_EUNSUPPORTED, "gator_forward_f32: ctx needs both GAT and MDR weights");
    c->last_batch = B;
    c->clear_taps();
    if (c->impl != GATOR_IMPL_BASIC) return fused_forward(c, pose2d, B, verts, pose3d, stream);
    const BasicLayout L = basic_layout(c->J, c->cap_batch);
    float *feat = c->ws + L.feat, *xout = c->ws + L.xout, *pc = c->ws + L.pc;
    { StageTimer t(c, "gat", stream); rc = basic_gat_forward(c, pose2d, B, xout, feat, stream); }
    if (rc) return rc;
    c->set_tap(TAP_FEAT, feat, (int64_t)B * c->J * kC);
    rc = basic_build_pc(c, pose2d, xout, feat, B, pc, pose3d, stream);
    if (rc) return rc;
    StageTimer t(c, "mdr+upsample", stream);
    return basic_mdr_forward(c, pc, B, verts, stream);
}
extern "C" int gator_forward_f32(gator_ctx* c, const float* pose2d, int32_t B, float* verts, float* pose3d, void* stream) { return finish_fwd(c, forward_f32_impl(c, pose2d, B, verts, pose3d, stream), "gator_forward_f32"); }

static void prof_clear(gator_ctx* c, bool destroy = false) {
    for (auto& r : c->prof) { c->ev_pool.push_back(r.start); c->ev_pool.push_back(r.stop); }
    c->prof.clear();
    if (destroy) {
        for (void* e : c->ev_pool) (void)hipEventDestroy((hipEvent_t)e);
        c->ev_pool.clear();
    }
}

extern "C" int gator_profile_enable(gator_ctx* c, int32_t on) {
    if (!c) return fail(GATOR_EINVAL, "gator_profile_enable: null ctx");
    prof_clear(c, false);
    if (on && c->ev_pool.size() < 1024) {    // pre-create so that the timed loop only records
        for (int i = 0; i < 1024; ++i) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) == hipSuccess) c->ev_pool.push_back(e);
        }
    }
    c->prof_stride = on > 0 ? on : 0;      // on = n: bracket the stages of every n-th forward (sampling keeps the hook's cost low)
    c->prof_calls = 0;
    c->profiling = false;
    return GATOR_OK;
}

extern "C" int gator_profile_read(gator_ctx* c, char* names, int64_t cap, float* total_ms, int32_t* calls, int32_t max_entries,
                                  int32_t* n_entries) {
    if (!c || !names || !total_ms || !calls || !n_entries || cap <= 0) return fail(GATOR_EINVAL, "gator_profile_read: null argument");
    std::vector<std::string> order;
    std::map<std::string, std::pair<double, int>> acc;
    for (auto& r : c->prof) {
        GATOR_HIP_CHECK(hipEventSynchronize((hipEvent_t)r.stop));
        float ms = 0.f;
        GATOR_HIP_CHECK(hipEventElapsedTime(&ms, (hipEvent_t)r.start, (hipEvent_t)r.stop));
        if (!acc.count(r.name)) order.push_back(r.name);
        acc[r.name].first += ms;
        acc[r.name].second += 1;
    }
    prof_clear(c, false);
    std::string joined;
    int n = 0;
    for (auto& k : order) {
        if (n >= max_entries) break;
        if (n) joined += "\n";
        joined += k;
        total_ms[n] = (float)acc[k].first;
        calls[n] = acc[k].second;
        ++n;
    }
    if ((int64_t)joined.size() + 1 > cap) return fail(GATOR_ESHAPE, "gator_profile_read: names buffer too small");
    memcpy(names, joined.c_str(), joined.size() + 1);
    *n_entries = n;
    return GATOR_OK;
}

extern "C" int gator_set_joint_regressor(gator_ctx* c, const int32_t* coo_row, const int32_t* coo_col, const float* coo_val, int32_t nnz,
                                         int32_t n_joint) {
    if (!c || !coo_row || !coo_col || !coo_val || nnz <= 0 || n_joint <= 0) return fail(GATOR_EINVAL, "gator_set_joint_regressor: bad arguments");
    if (c->impl != GATOR_IMPL_FUSED || !(c->parts & GATOR_PART_MDR)) return fail(GATOR_EUNSUPPORTED, "gator_set_joint_regressor: fused ctx with the MDR weights only");
    GATOR_HIP_CHECK(hipSetDevice(c->device));
    return fused_set_joint_regressor(c, coo_row, coo_col, coo_val, nnz, n_joint);
}

static int forward_joints_f32_impl(gator_ctx* c, const float* pose2d, int32_t B, float* joints, float* pose3d, float* verts, void* stream) {
    int rc = check_fwd(c, pose2d, joints, B, "gator_forward_joints_f32");
    if (rc) return rc;
    if (!pose3d) return fail(GATOR_EINVAL, "gator_forward_joints_f32: pose3d is null");
    if (c->parts != (GATOR_PART_GAT | GATOR_PART_MDR) || c->impl != GATOR_IMPL_FUSED)
        return fail(GATOR_EUNSUPPORTED, "gator_forward_joints_f32: needs a fused ctx with both GAT and MDR weights");
    c->last_batch = B;
    return fused_forward_joints(c, pose2d, B, joints, pose3d, verts, stream);
}
extern "C" int gator_forward_joints_f32(gator_ctx* c, const float* pose2d, int32_t B, float* joints, float* pose3d, float* verts, void* stream) { return finish_fwd(c, forward_joints_f32_impl(c, pose2d, B, joints, pose3d, verts, stream), "gator_forward_joints_f32"); }

static int forward_bf16_impl(gator_ctx* c, const float* pose2d, int32_t B, float* verts, float* pose3d, void* stream) {
    int rc = check_fwd(c, pose2d, verts, B, "gator_forward_bf16");
    if (rc) return rc;
    if (!pose3d) return fail(GATOR_EINVAL, "gator_forward_bf16: pose3d is null");
    if (c->parts != (GATOR_PART_GAT | GATOR_PART_MDR) || c->impl != GATOR_IMPL_FUSED)
        return fail(GATOR_EUNSUPPORTED, "gator_forward_bf16: needs a fused ctx with both GAT and MDR weights");
    c->last_batch = B;
    c->clear_taps();
    return fused_forward(c, pose2d, B, verts, pose3d, stream, true);
}
extern "C" int gator_forward_bf16(gator_ctx* c, const float* pose2d, int32_t B, float* verts, float* pose3d, void* stream) { return finish_fwd(c, forward_bf16_impl(c, pose2d, B, verts, pose3d, stream), "gator_forward_bf16"); }

static int upsample_bf16_impl(gator_ctx* c, const float* vert431, int32_t B, float* verts, void* stream) {
    int rc = check_fwd(c, vert431, verts, B, "gator_upsample_bf16");
    if (rc) return rc;
    if (!(c->parts & GATOR_PART_MDR) || c->impl != GATOR_IMPL_FUSED)
        return fail(GATOR_EUNSUPPORTED, "gator_upsample_bf16: needs a fused ctx with the MDR weights");
    return fused_upsample_bf16(c, vert431, B, verts, stream);
}
extern "C" int gator_upsample_bf16(gator_ctx* c, const float* vert431, int32_t B, float* verts, void* stream) { return finish_fwd(c, upsample_bf16_impl(c, vert431, B, verts, stream), "gator_upsample_bf16"); }

extern "C" int gator_enable_block_taps(gator_ctx* c, int32_t on) {
    if (!c) return fail(GATOR_EINVAL, "gator_enable_block_taps: null ctx");
    if (c->impl != GATOR_IMPL_FUSED) return fail(GATOR_EUNSUPPORTED, "gator_enable_block_taps: fused ctx only");
    c->block_taps = on != 0;
    return GATOR_OK;
}

extern "C" int gator_encoder_for_batch(gator_ctx* c, int32_t B) {
    if (!c || B <= 0) return fail(GATOR_EINVAL, "gator_encoder_for_batch: null ctx or batch <= 0");
    if (c->impl == GATOR_IMPL_BASIC || !c->fused) return GATOR_ENCODER_SAMPLE;
    const int n = fused_tiled_samples(c, B, /* unpinned: what AUTO would do, whatever gator_set_encoder has pinned */ true);
    return n == 0 ? GATOR_ENCODER_SAMPLE : GATOR_ENCODER_TILED;      // a batch the policy splits between both counts as tiled (its full rounds are)
}

extern "C" int gator_set_graph_replay(gator_ctx* c, int32_t on) {
    if (!c) return fail(GATOR_EINVAL, "gator_set_graph_replay: null ctx");
    if (c->impl != GATOR_IMPL_FUSED || !c->fused) return fail(GATOR_EUNSUPPORTED, "gator_set_graph_replay: fused ctx only");
    return fused_set_graph_replay(c, on);
}

extern "C" int gator_set_encoder(gator_ctx* c, int32_t mode) {
    if (!c) return fail(GATOR_EINVAL, "gator_set_encoder: null ctx");
    if (c->impl != GATOR_IMPL_FUSED || !c->fused) return fail(GATOR_EUNSUPPORTED, "gator_set_encoder: fused ctx only");
    if (mode < GATOR_ENCODER_AUTO || mode > GATOR_ENCODER_TILED) return fail(GATOR_EINVAL, "gator_set_encoder: mode must be -1, 0 or 1");
    return fused_set_encoder(c, mode);
}

extern "C" int gator_get_tap(gator_ctx* c, const char* name, float* dst, int64_t capacity, int64_t* count, void* stream) {
    if (!c || !name || !dst) return fail(GATOR_EINVAL, "gator_get_tap: null argument");
    const float* src = nullptr;
    int64_t n = 0;
    if (!strcmp(name, "hop_path_bias")) { src = c->hop_bias; n = (int64_t)kH * c->J * c->J; }
    else {
        static const struct { const char* name; int id; } kNames[] = {{"feat", TAP_FEAT}, {"mdr_lbf2", TAP_MDR_LBF2}, {"vert431", TAP_VERT431}};
        int blk = -1;
        if (!strncmp(name, "gat_block", 9) && name[9] >= '0' && name[9] < '0' + kDepth && !name[10]) blk = name[9] - '0';
        for (auto& k : kNames)
            if (!strcmp(name, k.name)) { src = c->taps[k.id].p; n = c->taps[k.id].n; }
        if (blk >= 0 && c->taps[TAP_GAT_BLOCKS].p) {     // [depth][B][J][128]
            n = c->taps[TAP_GAT_BLOCKS].n / kDepth;
            src = c->taps[TAP_GAT_BLOCKS].p + (int64_t)blk * n;
        }
        if (!src && (!strcmp(name, "mdr_lbf2") || blk >= 0))
            return fail(GATOR_EMISSING, "gator_get_tap: '%s' is recorded only after gator_enable_block_taps(ctx, 1) (fused ctx: its stores cost time)", name);
        if (!src) return fail(GATOR_EMISSING, "gator_get_tap: no tap named '%s' from the last forward", name);
    }
    if (count) *count = n;
    if (n > capacity) return fail(GATOR_ESHAPE, "gator_get_tap: '%s' needs %lld floats, capacity %lld", name, (long long)n, (long long)capacity);
    GATOR_HIP_CHECK(hipMemcpyAsync(dst, src, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return GATOR_OK;
}

namespace {
// verts [B,6890,3] -> joints [B,nj,3]; one block per (sample, joint): gather the joint's non-zeros.
__global__ void k_regress(const float* __restrict__ verts, const int32_t* __restrict__ row, const int32_t* __restrict__ col,
                          const float* __restrict__ val, int nnz, int nj, float* __restrict__ joints) {
    const int b = blockIdx.x / nj, j = blockIdx.x % nj, t = threadIdx.x;
    __shared__ double red[3][64];
    double a[3] = {0, 0, 0};
    for (int e = t; e < nnz; e += 64)
        if (row[e] == j)
            for (int c = 0; c < 3; ++c) a[c] += (double)val[e] * (double)verts[((int64_t)b * kNV + col[e]) * 3 + c];
    for (int c = 0; c < 3; ++c) red[c][t] = a[c];
    __syncthreads();
    if (t < 3) {
        double s = 0;
        for (int e = 0; e < 64; ++e) s += red[t][e];
        joints[((int64_t)b * nj + j) * 3 + t] = (float)s;
    }
}
}  // namespace

extern "C" int gator_regress_joints_f32(const float* verts, int32_t B, const int32_t* coo_row, const int32_t* coo_col,
                                        const float* coo_val, int32_t nnz, int32_t n_joint, float* joints, void* stream) {
    if (!verts || !coo_row || !coo_col || !coo_val || !joints || B <= 0 || nnz <= 0 || n_joint <= 0)
        return fail(GATOR_EINVAL, "gator_regress_joints_f32: bad arguments");
    k_regress<<<B * n_joint, 64, 0, (hipStream_t)stream>>>(verts, coo_row, coo_col, coo_val, nnz, n_joint, joints);
    GATOR_HIP_CHECK(hipGetLastError());
    return GATOR_OK;
}
