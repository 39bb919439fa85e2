// Internal declarations shared by the translation units of libgator_hip.so (not part of the C ABI).
#pragma once
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "gator_hip.h"

namespace gator {

int fail(int code, const char* fmt, ...);   // records the thread-local error message, returns `code`

constexpr int kC = 128;       // GAT embed dim        (lib/core/base.py:57)
constexpr int kH = 8;         // GAT heads            (lib/models/GAT.py:46)
constexpr int kDepth = 6;     // GAT blocks           (lib/core/base.py:57)
constexpr int kE = 64;        // MDR width            (lib/models/MDR.py:74)
constexpr int kV = 431;       // coarse vertex tokens (lib/models/MDR.py:80-81)
constexpr int kNV = 6890;     // SMPL vertices
constexpr int kHid = 256;     // MDR Mlp hidden       (mlp_ratio 4, MDR.py:49,61)
constexpr int kMaxJ = 32;

struct TensorRef {
    const void* data = nullptr;     // device copy inside the ctx arena
    int dtype = 0, ndim = 0;
    int64_t shape[4] = {0, 0, 0, 0};
    int64_t numel = 0;
    const float* f() const { return static_cast<const float*>(data); }
};

struct GatBlockW {
    const float *n1w, *n1b, *qkv_w, *qkv_b, *proj_w, *proj_b, *gcn_W, *gcn_M, *gcn_adj2, *gcn_bias;
    const float *xl0_w, *xl0_b, *xl1_w, *xl1_b, *xlb_w, *xlb_b, *n2w, *n2b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
};
struct MdrLayerW {
    const float *n1w, *n1b, *wq, *wk, *wv, *proj_w, *proj_b, *n2w, *n2b, *fc1_w, *fc1_b, *fc2_w, *fc2_b, *a2, *b2;
    const float *sa_w[4], *sa_b[4];
};
struct Weights {
    // GAT
    const float *graph_adj, *gl0_W, *gl0_b, *gn_w, *gn_b, *gl3_W, *gl3_b, *pos_id, *pos_num;
    const float *hp_W, *hp_emb, *hp_ew, *hp_eb, *norm_w, *norm_b, *lifter_w, *lifter_b;
    GatBlockW blk[kDepth];
    // MDR
    const float *v431, *v6890, *pos_j, *pos_v, *jfeat_w, *jfeat_b, *vfeat_w, *vfeat_b;
    MdrLayerW lay[3];
    const float *motion_w, *motion_b, *biasl_w, *biasl_b, *bn_w, *bn_b, *bn_mean, *bn_var, *scale_w, *scale_b;
    const float *bconv_w, *bconv_b, *up_w, *up_b;
    // constants
    const int64_t* sp;        // [J,J]
    const float* edge_input;  // [J,J,D]
    const int32_t* vj;        // [431]
};

// debug taps (gator_get_tap): slot ids and names
enum TapId { TAP_FEAT = 0, TAP_MDR_LBF2, TAP_VERT431, TAP_GAT_BLOCKS, TAP_COUNT };
struct Tap { const float* p = nullptr; int64_t n = 0; };

// Sticky device status word of a ctx (host-mapped memory written by kernels, read by the host at the next API call / gator_device_status)
enum DeviceStatus { DEV_OK = 0, DEV_PERSIST_INCOMPLETE = 1, DEV_NONFINITE = 2 };

struct FusedState;   // packed weights + workspace of the fused path (fused_*.hip)
struct ProfRec { const char* name; void* start; void* stop; };

}  // namespace gator

struct gator_ctx {
    int J = 0, alpha = 0, impl = 0, device = 0, D = 0, parts = 3, subbatch_streams = 0, arithmetic = 0;
    std::string prefix_gat, prefix_mdr;
    std::map<std::string, gator::TensorRef> t;
    char* arena = nullptr;
    size_t arena_bytes = 0;
    gator::Weights w{};
    // folded, input-independent constants (device, fp32)
    float* hop_bias = nullptr;     // [8,J,J]   HopPathEncoding.forward, modules.py:98-107
    float* adj_diag = nullptr;     // [6,J]     diag of sym(A+adj2), modules.py:247-251
    float* adj_off = nullptr;      // [6,J,J]   off-diagonal part
    float* mask1 = nullptr;        // [J,J]     sp<=1, modules.py:163-170
    float* mask2 = nullptr;        // [J,J]     sp==2
    float* pos_embed = nullptr;    // [J,128]   pos_id_embed[1..J] + pos_num_embed[deg], GAT.py:141-144
    // bring-up path workspace
    float* ws = nullptr;
    size_t ws_floats = 0;
    int cap_batch = 0;
    int last_batch = 0;
    // debug taps of the last forward: fixed slots (no allocation / map insertion on the forward path)
    gator::Tap taps[gator::TAP_COUNT] = {};
    bool block_taps = false;       // gator_enable_block_taps: k_gat also stores the residual stream after every GATBlock
    void clear_taps() { for (auto& t : taps) t = gator::Tap{}; }
    void set_tap(int id, const float* p, int64_t n) { taps[id].p = p; taps[id].n = n; }
    gator::FusedState* fused = nullptr;
    unsigned* status_host = nullptr;   // sticky device status (gator::DeviceStatus): pinned host word the kernels write ...
    unsigned* status_dev = nullptr;    // ... through this device pointer
    int status_reason = 0;             // DeviceStatus of the last report (gator_status_reason)
    unsigned deferred_status = 0;      // an EARLIER call's status, taken by the running entry point and reported when it returns (api.hip: finish_fwd)
    // measurement hook (gator_profile_*): (stage name, start event, stop event) per launch
    bool profiling = false;       // StageTimer records only when set; forwards toggle it from prof_stride
    int prof_stride = 0, prof_calls = 0;   // record every prof_stride-th forward (0 = off)
    std::vector<gator::ProfRec> prof;
    std::vector<void*> ev_pool;
};

namespace gator {
// basic_kernels.hip
int basic_fold_constants(gator_ctx* c, void* stream);
int basic_gat_forward(gator_ctx* c, const float* pose2d, int B, float* x_out, float* feat, void* stream);
int basic_mdr_forward(gator_ctx* c, const float* pc, int B, float* verts, void* stream);
int basic_upsample(gator_ctx* c, const float* vert431, int B, float* verts, void* stream);
int basic_build_pc(gator_ctx* c, const float* pose2d, const float* x_out, const float* feat, int B, float* pc, float* pose3d, void* stream);
int ensure_workspace(gator_ctx* c, int B);
struct BasicLayout { size_t feat, xout, pc, mdr, col, tmp, total; };
BasicLayout basic_layout(int J, int B);
// fused path (fused_api.hip)
int fused_create(gator_ctx* c, void* stream);
void fused_destroy(gator_ctx* c);
int fused_gat_forward(gator_ctx* c, const float* pose2d, int B, float* x_out, float* feat, void* stream);
int fused_mdr_forward(gator_ctx* c, const float* pc, int B, float* verts, void* stream);
int fused_upsample(gator_ctx* c, const float* vert431, int B, float* verts, void* stream);
int fused_forward(gator_ctx* c, const float* pose2d, int B, float* verts, float* pose3d, void* stream, bool bf16 = false);
int fused_upsample_bf16(gator_ctx* c, const float* vert431, int B, float* verts, void* stream);
int fused_set_joint_regressor(gator_ctx* c, const int32_t* row, const int32_t* col, const float* val, int nnz, int nj);
int fused_forward_joints(gator_ctx* c, const float* pose2d, int B, float* joints, float* pose3d, float* verts, void* stream);
int fused_set_encoder(gator_ctx* c, int mode);
int fused_c3_state(const gator_ctx* c, float* bound);
void fused_disable_persist(gator_ctx* c);
int fused_set_graph_replay(gator_ctx* c, int on);          // returns the number of graph launches so far (>= 0)
// samples of a batch of B that the sample-tiled encoder takes under the policy in force (unpinned: under the ctx's own AUTO policy)
int fused_tiled_samples(const gator_ctx* c, int B, bool unpinned = false);      // from now on the four MDR stages run as four launches on this ctx
}  // namespace gator

namespace gator {
// RAII stage bracket: records hipEvents around a stage when ctx->profiling (defined in api.hip)
struct StageTimer {
    gator_ctx* c; void* stream; int idx;
    StageTimer(gator_ctx* c, const char* name, void* stream);
    ~StageTimer();
};
}  // namespace gator

#define GATOR_HIP_CHECK(expr)                                                                         \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) return gator::fail(GATOR_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)
