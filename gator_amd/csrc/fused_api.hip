// Fused path: create-time packing, workspace, and stage dispatch.  No fallback to anything but HIP kernels.
#include <hip/hip_runtime.h>

#include <vector>

#include "fused_state.h"

namespace gator {
namespace {

int fused_ensure_ws(gator_ctx* c, int B) {
    FusedState* f = c->fused;
    if (f->ws && B <= f->cap) return GATOR_OK;
    if (f->ws) {
        GATOR_HIP_CHECK(hipDeviceSynchronize());
        GATOR_HIP_CHECK(hipFree(f->ws));
        f->ws = nullptr;
    }
    const int cap = B, MT = (cap + 31) / 32, J = c->J;
    const size_t tiles = (size_t)cap * kVT * 2 * kTile;
    size_t n = 0;
    auto take = [&](size_t k) { size_t o = n; n += (k + 63) & ~(size_t)63; return o; };
    const size_t o_vcp = take((size_t)MT * 3 * kCB * kTile), o_vc = take((size_t)cap * kV * 3), o_vf = take(2 * tiles),
                 o_q = take(2 * tiles), o_k = take(2 * tiles), o_v = take(2 * tiles), o_jkv = take((size_t)cap * 12 * kTile),
                 o_hf = take((size_t)cap * kV * 32), o_lbf = take((size_t)cap * kV * kE), o_feat = take((size_t)cap * J * kC),
                 o_xout = take((size_t)cap * J * 3), o_pc = take((size_t)cap * J * 133);
    GATOR_HIP_CHECK(hipMalloc(&f->ws, n * sizeof(float)));
    GATOR_HIP_CHECK(hipMemset(f->ws, 0, n * sizeof(float)));
    f->ws_floats = n;
    f->cap = cap;
    f->vcp = f->ws + o_vcp; f->vc = f->ws + o_vc; f->vf = f->ws + o_vf; f->q = f->ws + o_q; f->k = f->ws + o_k;
    f->v = f->ws + o_v; f->jkv = f->ws + o_jkv; f->hf = f->ws + o_hf; f->lbf = f->ws + o_lbf; f->feat = f->ws + o_feat;
    f->xout = f->ws + o_xout; f->pc = f->ws + o_pc;
    return GATOR_OK;
}

std::vector<float> d2h(const float* p, size_t n) {
    std::vector<float> v(n);
    (void)hipMemcpy(v.data(), p, n * sizeof(float), hipMemcpyDeviceToHost);
    return v;
}

}  // namespace

int fused_create(gator_ctx* c, void* stream) {
    if (c->impl == GATOR_IMPL_BASIC) return GATOR_OK;
    FusedState* f = new FusedState();
    c->fused = f;
    const Weights& w = c->w;
    if (!(c->parts & GATOR_PART_MDR)) return GATOR_OK;
    const size_t n_up = (size_t)3 * kOB * kCB * kTile, n_layer = (size_t)56 * kTile;
    const size_t total = n_up + 3 * n_layer + 2 * kTile + 64 + (size_t)kVT * 2 * kTile + 3 * 64 + 1024;
    GATOR_HIP_CHECK(hipMalloc(&f->wbuf, total * sizeof(float)));
    f->wbuf_floats = total;
    float* p = f->wbuf;
    auto take = [&](size_t k) { float* r = p; p += k; return r; };
    // upsample_conv.weight [6890][431][3] -> one packed [216][14] tile grid per tap
    float* up = take(n_up);
    for (int tap = 0; tap < 3; ++tap) {
        int rc = fused_pack_linear(w.up_w + tap, (int64_t)kV * 3, 3, kNV, kV, up + (size_t)tap * kOB * kCB * kTile, stream);
        if (rc) return rc;
    }
    f->up_w = up;
    for (int li = 0; li < 3; ++li) {
        const MdrLayerW& r = w.lay[li];
        MdrLayerP& q = f->lay[li];
        struct { const float* src; int N, K; const float** dst; } items[] = {
            {r.wq, 64, 64, &q.wq}, {r.proj_w, 64, 64, &q.proj}, {r.fc1_w, 256, 64, &q.fc1}, {r.fc2_w, 64, 256, &q.fc2},
            {r.sa_w[0], 64, 64, &q.sa[0]}, {r.sa_w[1], 64, 64, &q.sa[1]}, {r.sa_w[2], 64, 64, &q.sa[2]}, {r.sa_w[3], 64, 64, &q.sa[3]}};
        for (auto& it : items) {
            float* dst = take((size_t)nblk32(it.N) * nblk32(it.K) * kTile);
            int rc = fused_pack_linear(it.src, it.K, 1, it.N, it.K, dst, stream);
            if (rc) return rc;
            *it.dst = dst;
        }
    }
    // head: one 32-row linear. rows 0..19 motion_linear[0..19] (mat_A), 24..26 bias_linear, 27 scale_linear (alpha head),
    // 28..30 motion_linear[20..22] (mat_C)   -- MDR.py:156-162
    {
        std::vector<float> hw(32 * 64, 0.f), hb(32, 0.f);
        const std::vector<float> mw = d2h(w.motion_w, 23 * 64), mb = d2h(w.motion_b, 23), bw = d2h(w.biasl_w, 3 * 64), bb = d2h(w.biasl_b, 3);
        for (int r = 0; r < 20; ++r) { std::copy(mw.begin() + r * 64, mw.begin() + (r + 1) * 64, hw.begin() + r * 64); hb[r] = mb[r]; }
        for (int r = 0; r < 3; ++r) {
            std::copy(bw.begin() + r * 64, bw.begin() + (r + 1) * 64, hw.begin() + (24 + r) * 64); hb[24 + r] = bb[r];
            std::copy(mw.begin() + (20 + r) * 64, mw.begin() + (21 + r) * 64, hw.begin() + (28 + r) * 64); hb[28 + r] = mb[20 + r];
        }
        if (c->alpha) {
            const std::vector<float> sw = d2h(w.scale_w, 64), sb = d2h(w.scale_b, 1);
            std::copy(sw.begin(), sw.end(), hw.begin() + 27 * 64);
            hb[27] = sb[0];
        }
        float* tmp = nullptr;
        GATOR_HIP_CHECK(hipMalloc(&tmp, hw.size() * sizeof(float)));
        GATOR_HIP_CHECK(hipMemcpy(tmp, hw.data(), hw.size() * sizeof(float), hipMemcpyHostToDevice));
        float* dst = take(2 * kTile);
        int rc = fused_pack_linear(tmp, 64, 1, 32, 64, dst, stream);
        GATOR_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
        GATOR_HIP_CHECK(hipFree(tmp));
        if (rc) return rc;
        f->head_w = dst;
        float* hbd = take(64);
        GATOR_HIP_CHECK(hipMemcpy(hbd, hb.data(), 32 * sizeof(float), hipMemcpyHostToDevice));
        f->head_b = hbd;
    }
    // tokenise constants: get_verts_feature on [v431 | pose3d_nn] + pos_v (MDR.py:126-137); v431/bias/pos part folded per token
    {
        const std::vector<float> vw = d2h(w.vfeat_w, 64 * 6), vb = d2h(w.vfeat_b, 64), pv = d2h(w.pos_v, (size_t)(kV + 1) * 64), v4 = d2h(w.v431, kV * 3);
        std::vector<float> base((size_t)kVT * 2 * kTile, 0.f), w3(3 * 64);
        for (int t = 0; t < kVT; ++t)
            for (int nb = 0; nb < 2; ++nb)
                for (int g = 0; g < 4; ++g)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 4; ++j) {
                            const int token = 32 * t + (lane & 31), ch = 32 * nb + 8 * g + 4 * (lane >> 5) + j;
                            if (token >= kV) continue;
                            double s = (double)vb[ch];
                            for (int i = 0; i < 3; ++i) s += (double)vw[ch * 6 + i] * (double)v4[token * 3 + i];
                            base[(((size_t)(t * 2 + nb) * 4 + g) * 64 + lane) * 4 + j] = (float)(s + (double)pv[(size_t)(token + 1) * 64 + ch]);
                        }
        for (int i = 0; i < 3; ++i)
            for (int ch = 0; ch < 64; ++ch) w3[i * 64 + ch] = vw[ch * 6 + 3 + i];
        float* bd = take(base.size());
        GATOR_HIP_CHECK(hipMemcpy(bd, base.data(), base.size() * sizeof(float), hipMemcpyHostToDevice));
        f->tok_base = bd;
        float* wd = take(3 * 64);
        GATOR_HIP_CHECK(hipMemcpy(wd, w3.data(), w3.size() * sizeof(float), hipMemcpyHostToDevice));
        f->tok_w3 = wd;
    }
    GATOR_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return GATOR_OK;
}

void fused_destroy(gator_ctx* c) {
    if (!c->fused) return;
    if (c->fused->ws) (void)hipFree(c->fused->ws);
    if (c->fused->wbuf) (void)hipFree(c->fused->wbuf);
    delete c->fused;
    c->fused = nullptr;
}

int fused_gat_forward(gator_ctx* c, const float* pose2d, int B, float* x_out, float* feat, void* stream) {
    StageTimer tm(c, "gat", stream);
    return basic_gat_forward(c, pose2d, B, x_out, feat, stream);     // TODO(round 1): GAT megakernel
}

int fused_upsample(gator_ctx* c, const float* vert431, int B, float* verts, void* stream) {
    int rc = fused_ensure_ws(c, B);
    if (rc) return rc;
    FusedState* f = c->fused;
    rc = launch_pack_vc(vert431, B, f->vcp, stream);
    if (rc) return rc;
    StageTimer tm(c, "upsample", stream);
    return launch_upsample(f, c, B, verts, stream);
}

int fused_mdr_forward(gator_ctx* c, const float* pc, int B, float* verts, void* stream) {
    int rc = fused_ensure_ws(c, B);
    if (rc) return rc;
    FusedState* f = c->fused;
    rc = launch_mdr(c, f, pc, B, stream);
    if (rc) return rc;
    return fused_upsample(c, f->vc, B, verts, stream);
}

int fused_forward(gator_ctx* c, const float* pose2d, int B, float* verts, float* pose3d, void* stream) {
    int rc = fused_ensure_ws(c, B);
    if (rc) return rc;
    FusedState* f = c->fused;
    rc = fused_gat_forward(c, pose2d, B, f->xout, f->feat, stream);
    if (rc) return rc;
    c->taps["feat"] = {f->feat, (int64_t)B * c->J * kC};
    rc = basic_build_pc(c, pose2d, f->xout, f->feat, B, f->pc, pose3d, stream);
    if (rc) return rc;
    return fused_mdr_forward(c, f->pc, B, verts, stream);
}

}  // namespace gator
