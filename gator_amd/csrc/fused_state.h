// State of the fused path: packed (MFMA-operand-order) weights and the per-batch workspace.
#pragma once
#include "internal.h"
#include <vector>

namespace gator {

constexpr int kVT = 14;                 // 32-token tiles per sample (431 -> 448)
constexpr int kOB = 216;                // 32-vertex output blocks of the upsample GEMM (6890 -> 6912)
constexpr int kCB = 14;                 // 32-wide k blocks over the 431 coarse vertices
constexpr int kTile = 32 * 32;          // floats in one packed 32x32 tile ([4 g][64 lanes][4])
constexpr int kMdrCtrChunks = 64;       // persistent MDR launches one forward may be cut into (launch_mdr: chunks of 256 .. 511 samples)
// words of k_mdr_persist's counters for a forward of B samples: per launch a 32-word header (tickets, error flag) + 4 counts per sample
__host__ __device__ inline size_t mdr_ctr_words(int B) { return (size_t)32 * kMdrCtrChunks + (size_t)4 * B; }

struct MdrLayerP {                      // packed weights of one LBF layer (device pointers into FusedState::wbuf)
    const float *wq, *wk, *wv, *proj, *fc1, *fc2, *sa[4];
};

struct GatBlockPk {                     // packed tiles of one GATBlock
    const float *qkv, *proj, *w0, *w1, *lin0, *lin1, *back, *fc1, *fc2;
    const float *mc, *mdT, *aoffT, *f1b; // M (channel on lane), diag(A).M (TOKEN on lane), offdiag(A)^T B-operand tile, hop-2 bias term
};

// Per-sub-batch workspace.  FusedState derives from it so kernels launchers read `f->vf` etc.; fused_forward swaps the
// base part between the two sets when it runs two half-batches on two streams.
struct FusedWs {
    // workspace (per cap batch)
    float* ws = nullptr;
    size_t ws_floats = 0;
    int cap = 0;
    float *vcp = nullptr;               // [MT][3][kCB][4][64][4]  packed vert431 (A operand of the fp32-MFMA upsample GEMM)
    void* vcp3 = nullptr;               // split-precision A operand of the vertex GEMM: bf16 [plane 3][MT][3][28][64][8] (hi/mid/lo, upsample_x3.hip)
                                        // or fp16 [MT/4][28][4][3][plane 2][64][8] (hi/lo, upsample_x2.hip); sized for the larger
    float *vc = nullptr;                // [B][431][3]
    float *vf = nullptr, *q = nullptr, *k = nullptr, *v = nullptr;   // [B][14][2][kTile] each
    float *jkv = nullptr;               // [B][3 layers][2 (k,v)][2 heads][kTile]
    float *hf = nullptr;                // [B][431][32] head features
    float *lbf = nullptr;               // [B][431][64] tap: verts tokens after LBF3 (reference layout)
    float *feat = nullptr, *xout = nullptr, *pc = nullptr;
    float* hpart = nullptr;             // [cap][14][64] DOUBLES: the head conv's per-tile partial sums (mdr_fused.hip: head_conv_partial)
    float *lpart = nullptr;             // [MT][J][2][kTile] lifter partial tiles (gat_tail.hip)
    bool mdr_ctr_clean = false;         // the joint-token kernel queued before launch_mdr has zeroed mdr_ctr for it
    unsigned* mdr_ctr = nullptr;        // k_mdr_persist: the counter blocks of a forward's launches, mdr_ctr_words(cap) words (mdr_fused.hip: MdrChunkPlan)
    void* vcp16 = nullptr;              // bf16 packed vert431 for the bf16 vertex GEMM (cap-sized)
    int vcp16_cap = 0;
};

struct FusedState : FusedWs {
    float* wbuf = nullptr;              // all packed weights
    float* gbuf = nullptr;              // packed GAT weights + tables
    GatBlockPk gblk[kDepth];
    const float *g_biasT = nullptr, *g_m1T = nullptr, *g_m2T = nullptr, *g_gl3 = nullptr, *g_posT = nullptr, *g_vecs = nullptr;
    size_t wbuf_floats = 0;
    // upsample: Wp[tap][ob][cb][4][64][4]
    const float* up_w = nullptr;
    void* up_w3 = nullptr;              // bf16 [plane 3][tap][ob][28][64][8]  hi/mid/lo split of upsample_conv.weight
    int gat_tiled = -1;                 // encoder policy in force: 0 never tiled, 1 always, -1: by batch size (fused_tiled_samples); gator_set_encoder changes it
    int gat_tiled_env = -1;             // ... as gator_create found it (GATOR_GAT_TILED, default -1): what GATOR_ENCODER_AUTO means for this ctx
    int gat_tiled_min_batch = 1024;
    int n_cu = 256;                     // compute units of the ctx's device
    bool gat_split_tail = true;         // full forward: lifter + joint tokens as batched launches (GATOR_GAT_TAIL=0: inside k_gat)
    bool gat_x3 = true;                 // GAT linears on split-precision bf16 MFMA (GATOR_GAT_X3=0: fp32-input MFMA)
    float* gxbuf = nullptr;             // X3 tiles of the GAT block weights, tile-for-tile image of gbuf from gblk[0].qkv on
    float* gxbuf_h3 = nullptr;          // the same grids as three fp16 planes of 2^gat_tiled_wshift * w (k_gat_tiled's four-product form)
    bool gat_tiled_h4 = true;           // GATOR_GAT_TILED_H4=0: the sample-tiled encoder on the exact six products
    int gat_tiled_wshift = 0;
    float* g8stream = nullptr;          // the same tiles as four per-wave streams in consumption order (gat_roles.hip)
    bool gat8 = true;                   // one-sample-per-workgroup encoder: the two-role kernel k_gat8 (GATOR_GAT8=0: k_gat)
    bool gat8_tail = true;              // k_gat8 runs the lifter and the MDR joint tokens as its epilogue (round 6; GATOR_GAT8_TAIL=0: the two launches of gat_tail.hip)
    bool gat8_h4 = true;                // ... with its token-wise products on four partial products (x3_common.h; GATOR_GAT8_H4=0: the exact six)
    float* g8stream_b = nullptr;        // ... and its byte-lo image (H3B tiles, 5 KiB: gat_roles.hip), what k_gat8<true, LR, false, true> streams
    bool gat8_lobyte = false;           // set when every weight's lo plane survives the byte round trip (always, for finite weights; GATOR_GAT8_LOBYTE=0: off)
    int gat8_wshift = 0;                // its weight stream holds three fp16 planes of 2^gat8_wshift * w
    float* wxbuf = nullptr;             // X3 tiles of the MDR layer + head weights, tile-for-tile image of wbuf from lay[0].wq on
    int mdr_persist = -1;               // the four MDR stages as ONE persistent launch (k_mdr_persist): -1 by batch size (launch_mdr), GATOR_MDR_PERSIST=0 never, =1 always
    int mdr_persist_chunk = 0;          // GATOR_MDR_PERSIST_CHUNK: most samples per persistent launch (0: 384)
    bool mdr_head_partials = true;      // GATOR_MDR_HEAD_PARTIALS=0 (read at create): the whole head in k_mdr_head instead of the tiles' conv partial sums + k_mdr_head_finish (A/B)
    int mdr_persist_grid = 0;           // GATOR_MDR_PERSIST_GRID: workgroups of the persistent launch (0: two per CU)
    float* jf128_h3 = nullptr;          // get_joint_feature columns 5..132 as H3 tiles [2][4] of 2^jf128_wshift * w (k_gat8's fused tail)
    int jf128_wshift = 0;
    int mdr_wshift = 0;                 // GATOR_MDR_X3=2: wxbuf holds three fp16 planes of 2^mdr_wshift * w
    int mdr_x3 = 2;                     // GATOR_MDR_X3: 0 fp32-input MFMA; 1 exact bf16 x 3 split everywhere; 2 (default) that + the 431x431 attention on two fp16 planes
    float c3_logit_bound = 0.f;         // bound on |q . k| / sqrt(d_k) of the MDR self-attention in the exp2 domain, from the weights (fused_create)
    bool c3_guarded = false;            // ... exceeded 2^10: c3_mdr switched off for this ctx (GATOR_C3_GUARD=0: never)
    bool c3_mdr = true;                 // gator_forward_bf16 (BASELINE config 3): the MDR layers on one fp16 activation plane (GATOR_C3_MDR=0: fp32 form)
    bool c3_encoder = true;             // ... the encoder's token-wise products on one fp16 activation plane as well (GATOR_C3_ENCODER=0: the fp32 configuration's)
    bool c3_up_w1 = true;               // ... the vertex regressor's weights on ONE fp16 plane, coarse vertices on two (GATOR_C3_UPSAMPLE_W1=0: weights on two)
    bool c3_up_bf16 = false;            // ... and the vertex regressor on one bf16 plane (GATOR_C3_UPSAMPLE_BF16=1; default: its two fp16 planes)
    bool x3 = true;                     // split-precision vertex regressor (GATOR_UPSAMPLE_X3=0: fp32-input MFMA kernel)
    bool up_x2 = true;                  // ... on two fp16 planes (default; GATOR_UPSAMPLE_X3=1: the exact three bf16 planes)
    void* up_w2 = nullptr;              // fp16 [ob/2][28][2][tap 3][plane 2][64][8]  scaled hi/lo split of upsample_conv.weight
    float up_w2_unscale = 1.f;          // 2^-(weight shift + activation shift), applied to the finished sums
    void* up_w16 = nullptr;             // bf16 [tap][ob][28][64][8] (packed on the first bf16 call, which waits for the pack)
    // joint regressor fused into the vertex GEMM's epilogue (gator_set_joint_regressor / gator_forward_joints_f32)
    void *jr_blk = nullptr, *jr_ent = nullptr;    // int2 [kOB] (first, count) ; int2 [nnz] (vertex, slot)
    float* jr_w = nullptr;                        // [nnz] weights in entry order
    int* jr_rowptr = nullptr;                     // [nj + 1] CSR row pointers over the slots (sorted by joint, then vertex)
    float* jr_P = nullptr;                        // [cap][nnz][3] partial products
    int jr_nnz = 0, jr_nj = 0, jr_cap = 0;
    float* blk_tap = nullptr;           // debug: residual stream after every GATBlock [depth][B][J][128] (gator_enable_block_taps)
    int blk_tap_cap = 0;
    // MDR
    MdrLayerP lay[3];
    const float* head_w = nullptr;      // [1 nb][2 kb] combined motion/bias/scale linear
    const float* head_b = nullptr;      // [32]
    const float* tok_base = nullptr;    // [14][2][4][64][4]  v431 part of get_verts_feature + bias + pos_v  (T-layout tiles)
    const float* tok_w3 = nullptr;
    // hipGraph replay of the full forward (gator_set_graph_replay / GATOR_GRAPH=1).  A forward is identified by (batch, the three
    // caller pointers, precision, encoder pin, persistent-launch state, workspace): the first time a key is seen the forward runs
    // directly (lazy allocations happen there), the second time it is captured on a private stream, from then on one hipGraphLaunch
    // on the caller's stream replaces the six launches.  Keys are kept LRU (kGraphSlots); anything unexpected switches the feature off.
    struct GraphSlot {
        int B = 0; const void *in = nullptr, *verts = nullptr, *pose3d = nullptr; bool bf16 = false;
        int tiled = 0, persist = 0; const void* ws = nullptr;
        void *graph = nullptr, *exec = nullptr;
        unsigned long long used = 0;
    };
    static constexpr int kGraphSlots = 8;
    bool graph_replay = false;
    void* cap_stream = nullptr;
    std::vector<GraphSlot> graphs;
    unsigned long long graph_clock = 0, graph_launches = 0;
    // sub-batch pipelining (two half-batches on two streams: one half's kernel tails are filled by the other's work)
    FusedWs sets[2];
    void* aux_stream = nullptr;
    void *ev_fork = nullptr, *ev_join = nullptr;
    const float* jfeat_p = nullptr;     // get_joint_feature.weight packed [2 nb][5 kb]
    const float* jfeat5 = nullptr;      // its columns 0..4 (pose2d, pose3d/1000) as [5][64]
    const float* jfeat128_p = nullptr;  // its columns 5..132 (feat) packed [2 nb][4 kb]
    const float* posj_T = nullptr;      // [2] T-layout tiles of pos_j_id_embed[1..J]      // [3][64]            pose3d part of get_verts_feature (columns 3..5), row-major [i][ch]
};

// fused_pack.hip
int fused_pack_linear(const float* W, int64_t wsn, int64_t wsk, int N, int K, float* dst, void* stream);   // -> [NB][KB] tiles
inline int nblk32(int n) { return (n + 31) / 32; }
// upsample_fused.hip
int launch_pack_vc(const float* vc, int B, float* vcp, void* stream);
int launch_upsample(const FusedState* f, const gator_ctx* c, int B, float* verts, void* stream);
// gat_fused.hip
int gat_prepare_device();
int gat_ensure_blk_tap(gator_ctx* c, FusedState* f, int B);
int launch_gat(gator_ctx* c, FusedState* f, const float* pose2d, int B, float* x_out, float* feat, void* stream, bool joint_epilogue = false,
               int B_total = 0, int tap_row0 = 0, bool half16 = false, float* tail_jkv = nullptr);      // half16: the one-plane form of the two-role kernel (config 3); other forms ignore it
// gat_roles.hip
int gat8_prepare_device();
int gat8_build_stream(FusedState* f, void* stream);
int launch_gat8(gator_ctx* c, FusedState* f, const float* pose2d, int B, float* feat, void* stream, int B_total = 0, int tap_row0 = 0, bool half16 = false,
                float* tail_x_out = nullptr, float* tail_jkv = nullptr, int ctr_B = 0);
bool gat8_tail_supported(const gator_ctx* c, const FusedState* f, bool half16);
// gat_tiled.hip
int gat_tiled_prepare_device();
int gat_tiled_samples_per_wg(int J);
int launch_gat_tiled(gator_ctx* c, FusedState* f, const float* pose2d, int B, float* feat, void* stream, int B_total = 0, bool half16 = false);
// gat_tail.hip
size_t gat_tail_part_floats(int B, int J);
int launch_gat_tail(gator_ctx* c, FusedState* f, const float* pose2d, const float* feat, int B, float* x_out, void* stream, bool joint, bool zero_ctr = true);
// upsample_bf16.hip
size_t upsample_bf16_weight_elems();
size_t upsample_bf16_vcp_elems(int B);
int pack_upsample_bf16(const float* up_w, void* dst, void* stream);
int launch_upsample_bf16(const FusedState* f, const gator_ctx* c, const float* vc, int B, float* verts, void* stream);
// upsample_x3.hip
size_t upsample_x3_weight_elems();
size_t upsample_x3_vcp_elems(int B);
int pack_upsample_x3(const float* up_w, void* dst, void* stream);
int launch_pack_vc_x3(const float* vc, int B, int cap, void* vcp3, void* stream);
int launch_upsample_x3(const FusedState* f, const gator_ctx* c, int B, float* verts, void* stream, bool with_joints = false);
int launch_jreg_reduce(const FusedState* f, int B, float* joints, void* stream);
// upsample_x2.hip
size_t upsample_x2_weight_elems();
size_t upsample_x2_vcp_elems(int B);
int upsample_x2_prepare_device();
int pack_upsample_x2(const float* up_w, void* dst, float* unscale, void* stream);
int launch_pack_vc_x2(const float* vc, int B, void* vcp2, void* stream);
int launch_upsample_x2(const FusedState* f, const gator_ctx* c, int B, float* verts, void* stream, bool with_joints = false, bool w1 = false);
// the vertex regressor the ctx was created with (fp32-input MFMA | bf16 x 3 | fp16 x 2)
int launch_upsample_any(const FusedState* f, const gator_ctx* c, int B, float* verts, void* stream, bool with_joints = false, bool w1 = false);
// mdr_fused.hip
int launch_mdr(gator_ctx* c, FusedState* f, const float* pc, int B, void* stream, const float* x_out = nullptr, const float* pose2d = nullptr, bool half16 = false);

}  // namespace gator
