/*
 * gator_hip.h -- C ABI of libgator_hip.so: the MI355X (gfx950) implementation of the GATOR inference
 * forward pass (GAT graph-aware transformer encoder -> MDR regression head -> 6890 SMPL vertices).
 *
 * The reference (kasvii/GATOR) has no FFI of its own: its boundary for this path is the Python
 * nn.Module API of lib/models (GATOR.py:16-27, GAT.py:133-156, MDR.py:124-174).  These entry points are
 * what a binding for that boundary calls; the modules under gator_amd/models/ are such a binding (ctypes) and
 * INTEGRATION.md shows the stub a reference maintainer would add.  Plain pointers and sizes only.
 *
 * Conventions
 *   - every function returns 0 on success, a negative gator_status otherwise; gator_last_error() then
 *     returns a thread-local human-readable message (reference convention: Python exceptions,
 *     lib/funcs_utils.py:121-127 -> the Python wrapper raises RuntimeError).
 *   - "device pointer" = hipMalloc'd memory on the ctx's device (a torch tensor's data_ptr()).
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); all work is enqueued on it
 *     and nothing synchronises the host.  A ctx is not thread-safe: one ctx per device per thread.
 */
#ifndef GATOR_HIP_H
#define GATOR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gator_ctx gator_ctx;

typedef enum { GATOR_OK = 0, GATOR_EINVAL = -1, GATOR_EMISSING = -2, GATOR_ESHAPE = -3, GATOR_EHIP = -4,
               GATOR_ENOMEM = -5, GATOR_EUNSUPPORTED = -6,
               GATOR_EDEVICE = -7,    /* a kernel of an EARLIER call flagged its result as invalid (gator_device_status) */
               GATOR_EDEVICE_DEFERRED = -8   /* the same report, riding on a forward that WAS queued normally: this call's outputs are being
                                                written and are valid unless the next call reports again (gator_status_reason says why) */
} gator_status;

/* ABI 2 (round 6): gator_config starts with struct_size; forwards return GATOR_EDEVICE_DEFERRED instead of GATOR_EDEVICE when they carry
 * an earlier call's report; gator_abi_version / gator_status_reason added.  A binding checks gator_abi_version() == GATOR_ABI_VERSION at load. */
#define GATOR_ABI_VERSION 2

typedef enum { GATOR_F32 = 0, GATOR_I64 = 1, GATOR_I32 = 2 } gator_dtype;

/* One named tensor.  Names are the reference checkpoint's state_dict keys (SURVEY.md Appendix B:
 * "pose_lifter.blocks.0.attn.qkv.weight", "pose2mesh.upsample_conv.weight", ...) -- the checkpoint layout IS
 * the weight contract (lib/core/base.py:70, demo/run.py:98) -- plus the derived constants the reference keeps as
 * plain module attributes:
 *     "const.shortest_path" i64 [J,J]   (data/base_data/shortest_path_*.npy, lib/models/GAT.py:89-93)
 *     "const.edge_input"    f32 [J,J,D] (gen_edg_input, lib/models/backbones/modules.py:13-29)
 *     "const.vj_relation"   i32 [431]   (build_verts_joints_relation, lib/graph_utils.py:71-89)
 * `data` may be a device or a host pointer (is_host says which); the ctx copies what it needs. */
typedef struct {
    const char* name;
    const void* data;
    int32_t dtype;      /* gator_dtype */
    int32_t ndim;       /* <= 4 */
    int64_t shape[4];
    int32_t is_host;
    int32_t reserved;
} gator_tensor;

/* GATOR_IMPL_FUSED computes every large product on the 16-bit MFMA from SPLIT fp32 operands (csrc/x3_common.h): weights always
 * exactly (three fp16 or bf16 planes), and by default activations, attention operands and the vertex regressor's operands as TWO
 * fp16 planes of 16 x value (22 significant bits; four resp. three partial products per fp32 product instead of six).
 * Operand range of that default: every value that feeds a token-wise linear and every coarse vertex must stay below 4 094 in
 * magnitude (16 x value must fit an fp16 plane); beyond it the plane overflows, the vertices of that forward come out NaN and the
 * NEXT call on the ctx (or gator_device_status) returns GATOR_EDEVICE -- loud, never silently wrong.  Trained checkpoints are
 * orders of magnitude inside (LayerNorm outputs, GELU hiddens, metres).  gator_config.arithmetic = GATOR_ARITH_EXACT_SPLIT (or the environment
 * variables it stands for, read by gator_create) selects the forms without a rounded operand and without that limit (GATOR_MDR_X3=1 GATOR_UPSAMPLE_X3=1 GATOR_GAT8_H4=0 GATOR_GAT_TILED_H4=0: exact
 * three-way bf16 split, six products) or the fp32-input MFMA form of a stage (GATOR_GAT_X3 / GATOR_MDR_X3 / GATOR_UPSAMPLE_X3 = 0);
 * all forms are the same accuracy class and under test (tests/test_gpu_x3.py; statistics: profiles/r04_error_budget.md). */
typedef enum { GATOR_IMPL_FUSED = 0,   /* MFMA / register-resident fused kernels (default) */
               GATOR_IMPL_BASIC = 1    /* bring-up kernels: one simple HIP kernel per reference op; used as an
                                          on-device cross-check of the fused path */
} gator_impl;

typedef enum { GATOR_ARITH_DEFAULT = 0, GATOR_ARITH_EXACT_SPLIT = 1 } gator_arith;

typedef enum { GATOR_PART_GAT = 1,    /* pose_lifter.*  (models.GAT.get_model used stand-alone, lib/core/base.py:59) */
               GATOR_PART_MDR = 2     /* pose2mesh.*    (models.MDR.get_model, lib/models/MDR.py:172-174) */
} gator_parts;

typedef struct {
    int32_t struct_size; /* = sizeof(gator_config) of the header the caller was built with.  Fields the caller's header does not have yet
                            read as 0 (their defaults); a value that is not a multiple of 4 in [8, 1024] is refused (GATOR_EINVAL) -- which is
                            what a caller built against the ABI-1 header (first field num_joint = 17 / 19) gets instead of a mis-read struct */
    int32_t num_joint;   /* 17 (Human3.6M) or 19 (COCO + pelvis + neck); lib/models/GAT.py:79-93 */
    int32_t alpha;       /* cfg.MODEL.alpha: LayerNorm(3)+scale head instead of BatchNorm1d(431); MDR.py:115-119,162 */
    int32_t impl;        /* gator_impl */
    int32_t max_batch;   /* workspace is sized for this many samples up front (0 = grow on demand) */
    int32_t parts;       /* gator_parts bitmask: which sub-modules' weights are present (0 = both) */
    int32_t subbatch_streams; /* 2: run batches >= 128 as two half-batches on two streams (identical results, better tails) */
    int32_t arithmetic;  /* gator_arith: 0 = default (two-plane activations, operand range |value| < 4 094, see above);
                            1 = GATOR_ARITH_EXACT_SPLIT: every product on the exact three-way bf16 split (six partial products), no
                            rounded operand and no operand-range limit -- what a ctx that reported GATOR_EDEVICE for an out-of-range
                            operand is re-created with.  Same accuracy class, ~1.5 x the time (tests/test_gpu_x3.py) */
} gator_config;

/* Replaces: models.GATOR.get_model(...) + load_state_dict + .cuda()  (lib/models/GATOR.py:24-27,
 * lib/core/base.py:57,70,197).  Copies/packs the weights, folds the input-independent constants
 * (hop/path attention bias modules.py:98-107, MGCN adjacency :247-249, hop masks :163-170, BatchNorm affine). */
int gator_create(const gator_tensor* tensors, int32_t n_tensors, const gator_config* cfg, gator_ctx** out);
int gator_destroy(gator_ctx* ctx);

/* Device-side validity of the forwards issued so far.  Kernels never synchronise the host, so a failure that only the device can
 * see -- non-finite / out-of-range coarse vertices (operand range above), or a persistent MDR launch that did not finish every
 * sample (an XCD without workgroups under a CU mask; the ctx then switches to the four-launch form for good) -- is recorded in a sticky,
 * host-visible status word and reported by the NEXT entry point called on the ctx, once, as GATOR_EDEVICE (the affected vertices are NaN).  gator_device_status reports it on demand; sync != 0 waits for the device first
 * (the reference raises at the point of use, lib/core/base.py:210-237; this is the asynchronous equivalent). */
int gator_device_status(gator_ctx* ctx, int32_t sync);
/* Why the last GATOR_EDEVICE / GATOR_EDEVICE_DEFERRED of this ctx was returned: 1 = a persistent MDR launch did not finish (the ctx has switched to
 * the four-launch form), 2 = non-finite / out-of-range coarse vertices (input poses not finite, or the default arithmetic's operand range:
 * re-create the ctx with GATOR_ARITH_EXACT_SPLIT -- gator_amd/models/_base.py does exactly that by itself), 0 = none yet. */
int gator_status_reason(gator_ctx* ctx);

/* Replaces: GATOR.forward (lib/models/GATOR.py:16-22).
 *   pose2d [B,J,2] f32 device, contiguous  ->  verts [B,6890,3] f32 (metres), pose3d [B,J,3] f32 (mm).
 * fp32 in, fp32 out, fp32 accumulation; within 1e-3 mm of the fp64 evaluation of the reference (tests/).  The products themselves run
 * on the 16-bit MFMA with split operands: weights are always carried exactly (three planes); by default activations, the attention
 * operands and the vertex regressor's operands are rounded to 22 bits (two fp16 planes), which costs less than the fp32 rounding noise
 * the reference forward has itself.  Environment switches read at gator_create select the forms without any rounded operand
 * (GATOR_MDR_X3=1 GATOR_UPSAMPLE_X3=1 GATOR_GAT8_H4=0 GATOR_GAT_TILED_H4=0) or the fp32-input MFMA (=0). */
int gator_forward_f32(gator_ctx* ctx, const float* pose2d, int32_t batch, float* verts, float* pose3d, void* stream);

/* BASELINE config 3 ("full GATOR forward bf16"): the same forward in 16-bit operand mode (round 5).  Activations travel as ONE fp16 plane
 * into every token-wise linear of the encoder (lib/models/GAT.py:33-43, backbones/modules.py:121-196) and of the three MDR layers
 * (lib/models/MDR.py:140-153, vanilla_transformer_encoder.py:82-94), whose attention cores take Q, K, V and the probabilities as one plane
 * too; weights stay on two planes (22 bits) - except upsample_conv's (MDR.py:122,167-168), which go on one while the coarse vertices stay on
 * two.  fp32 accumulate / softmax / norms / GELU / residual stream / output; the head features, the encoder's J x J operators, the lifter
 * and the tokenisers keep gator_forward_f32's operands.  Which operand may be one plane was decided with the fp64 oracle under operand
 * rounding (tools/emulate_16bit.py): a single 16-bit plane for the WEIGHTS of the linears, or bf16 anywhere, is what costs millimetres.
 * Against the fp64 evaluation of the reference over 2048 samples x every coordinate: 0.76 mm max, 0.10 mm rms, |delta MPJPE| < 1e-3 mm
 * (tests/test_gpu_bf16.py; bar 1 mm / 0.2 mm / 0.05 mm); 1.6 x the fp32 forward at B = 2048.  Not for weights that drive the attention
 * logits to hundreds (near one-hot softmaxes): there one plane moves a score by 2^-12 of its magnitude.  Needs the default operand forms
 * (GATOR_MDR_X3=2, GATOR_UPSAMPLE_X3=2, GATOR_GAT8_H4=1, GATOR_GAT_TILED_H4=1).  Read at gator_create: GATOR_C3_ENCODER=0 (encoder as in
 * gator_forward_f32), GATOR_C3_UPSAMPLE_W1=0 (regressor weights on two planes), GATOR_C3_MDR=0 (MDR layers as in gator_forward_f32),
 * GATOR_C3_UPSAMPLE_BF16=1 (round 4's form of the regressor: both operands one bf16 plane, 8 mm max / 1 mm rms).
 * gator_upsample_bf16: the stage entry point of that bf16 vertex regressor. */
int gator_forward_bf16(gator_ctx* ctx, const float* pose2d, int32_t batch, float* verts, float* pose3d, void* stream);
/* Guard of that mode (round 6).  One fp16 plane moves an attention score by up to 2^-11 |q| |k|; gator_create bounds |q . k| / sqrt(d_k) of the three
 * 431 x 431 self-attentions from the weights alone (sigma_max(Wq_h^T Wk_h) x the custom LayerNorm's bound on |x|^2, exp2 domain).  Above 2^10 the MDR
 * layers and the encoder of gator_forward_bf16 keep gator_forward_f32's two planes for this ctx (the vertex regressor stays in the mode); GATOR_C3_GUARD=0
 * switches the guard off.  Returns 1 if the MDR layers run on one plane under gator_forward_bf16, 0 if not (guard, GATOR_C3_MDR=0 or a ctx without the
 * default operand forms), negative on error; *logit_bound (may be NULL) receives the bound. */
int gator_c3_state(gator_ctx* ctx, float* logit_bound);
int gator_upsample_bf16(gator_ctx* ctx, const float* vert431, int32_t batch, float* verts, void* stream);

/* Stage entry points (parity tests; same semantics as the reference sub-modules):
 *   GAT.forward   lib/models/GAT.py:133-152 : pose2d [B,J,2] -> x_out [B,3J] (mm), feat [B,J,128]
 *   MDR.forward   lib/models/MDR.py:124-170 : pose_combine [B,J,133] -> verts [B,6890,3]
 *   upsample_conv + template add  MDR.py:167-168 : vert431 [B,431,3] -> verts [B,6890,3]              */
int gator_gat_forward_f32(gator_ctx* ctx, const float* pose2d, int32_t batch, float* x_out, float* feat, void* stream);
int gator_mdr_forward_f32(gator_ctx* ctx, const float* pose_combine, int32_t batch, float* verts, void* stream);
int gator_upsample_f32(gator_ctx* ctx, const float* vert431, int32_t batch, float* verts, void* stream);

/* Debug taps of the LAST forward on this ctx, converted to the reference's layout:
 *   "hop_path_bias" [8,J,J]   "feat" [B,J,128]   "mdr_lbf2" [B,431,64]   "vert431" [B,431,3]
 * dst is a device pointer with room for `capacity` floats; *count receives the element count. */
int gator_get_tap(gator_ctx* ctx, const char* name, float* dst, int64_t capacity, int64_t* count, void* stream);
/* Additional taps "gat_block0" .. "gat_block5" [B,J,128]: the residual stream after each GATBlock (lib/models/GAT.py:33-43,
 * :145-147).  Off by default (the stores cost time); fused ctx only.  On a fused ctx the same switch also governs "mdr_lbf2" (110 KB of
 * stores per sample that nothing else reads): without it gator_get_tap("mdr_lbf2") returns GATOR_EMISSING.  Taps never outlive the next call on the ctx; the "feat" tap
 * of the stand-alone gator_gat_forward_f32 aliases the caller's `feat` buffer. */
int gator_enable_block_taps(gator_ctx* ctx, int32_t on);

/* Encoder policy of a fused ctx.  The six GAT blocks (lib/models/GAT.py:145-147) have two kernels: one workgroup per sample
 * (bit-identical results whatever the batch size or the position in the batch) and a sample-tiled one for large batches
 * (7 / 6 samples per workgroup; bit-identical within itself, equal to the first to fp32 rounding).  mode GATOR_ENCODER_AUTO (the
 * default) picks per call from the batch size, so above 1 024 samples a sample's last bits depend on how the batch was cut;
 * GATOR_ENCODER_SAMPLE / GATOR_ENCODER_TILED pin one kernel for every call on the ctx - what a sharded run uses so that the
 * all-gathered result is independent of the number of ranks (gathered == single-GPU, bit for bit). */
#define GATOR_ENCODER_AUTO (-1)
#define GATOR_ENCODER_SAMPLE 0
#define GATOR_ENCODER_TILED 1
int gator_set_encoder(gator_ctx* ctx, int32_t mode);
/* Which kernel GATOR_ENCODER_AUTO would use for a call of `batch` samples on this ctx (its environment switches, the device's CU
 * count): GATOR_ENCODER_SAMPLE or GATOR_ENCODER_TILED (a batch the policy splits between both counts as TILED); negative on error.
 * What a sharded run pins for every call so that the rule lives in one place (gator_amd/parallel.py). */
int gator_encoder_for_batch(gator_ctx* ctx, int32_t batch);

/* hipGraph replay of repeated forwards on a fused ctx (off by default; GATOR_GRAPH=1 switches it on at gator_create).  A forward is
 * identified by (batch, the three pointers, precision, encoder pin): the second time the same call is seen it is captured on a private
 * stream, from then on gator_forward_f32 / _bf16 is ONE hipGraphLaunch on the caller's stream (same kernels, same results bit for
 * bit; up to 8 calls are remembered, least recently used first out).  Needs stable pointers (pre-allocated outputs); a call that
 * is profiled, tapped or split into sub-batches runs directly.  on = 1 / 0 switches, on < 0 only queries; returns the number of
 * graph launches so far (>= 0) or a negative error. */
int gator_set_graph_replay(gator_ctx* ctx, int32_t on);

/* Measurement hook (bench.py `roofline`): gator_profile_enable(ctx, n) with n >= 1 brackets every stage launch of every
 * n-th forward by a hipEvent pair recorded on the launch stream (n = 0 switches it off).  gator_profile_read synchronises those events and returns, per stage name
 * ('\n'-separated in `names`), the summed duration in ms and the number of launches since the last enable/read. */
int gator_profile_enable(gator_ctx* ctx, int32_t on);
int gator_profile_read(gator_ctx* ctx, char* names, int64_t names_capacity, float* total_ms, int32_t* calls,
                       int32_t max_entries, int32_t* n_entries);

/* "Next" row 8(f)-1: J_regressor @ verts (lib/core/base.py:221, demo/run.py:142) as a sparse product.
 *   coo_{row,col,val}: nnz entries of a [n_joint,6890] regressor (device); joints [B,n_joint,3]. */
int gator_regress_joints_f32(const float* verts, int32_t batch, const int32_t* coo_row, const int32_t* coo_col,
                             const float* coo_val, int32_t nnz, int32_t n_joint, float* joints, void* stream);

/* The same regression FUSED into the forward: register a sparse regressor once (COO; host or device pointers), then
 * gator_forward_joints_f32 = GATOR.forward + J_regressor @ mesh in one go -- the vertex GEMM's epilogue forms the regressor's
 * partial products while the vertices are still in registers, a tiny kernel sums them in a fixed order (no atomics).
 *   joints [B,n_joint,3] f32 (metres, like the mesh; the reference scales by 1000 before regressing, lib/core/base.py:219-221),
 *   pose3d [B,J,3] (mm); verts [B,6890,3] or NULL: with NULL no vertex is ever written -- evaluation (lib/core/base.py:219-237,
 *   which copies every mesh to the host twice) then needs 12*n_joint bytes per sample instead of 82 680. */
int gator_set_joint_regressor(gator_ctx* ctx, const int32_t* coo_row, const int32_t* coo_col, const float* coo_val, int32_t nnz,
                              int32_t n_joint);
int gator_forward_joints_f32(gator_ctx* ctx, const float* pose2d, int32_t batch, float* joints, float* pose3d, float* verts, void* stream);

/* "Next" row 8(f)-3: the input contract in front of the path (demo/run.py:103-121,127-134, data/PW3D/dataset.py:168-183,241-250):
 *   joints [batch, num_joint_in, comps] raw 2D joints in pixels (comps >= 2: x, y[, score]); add_pelvis_neck != 0 appends
 *   pelvis = (joint 11 + joint 12)/2 and neck = (joint 5 + joint 6)/2 (COCO order); pose2d [batch, num_joint_out, 2] =
 *   per-sample per-axis (xy - mean) / std over the joints (population std) -- what the bbox/affine/normalise chain reduces
 *   to for rot = 0, flip = 0. */
int gator_preprocess_pose2d_f32(const float* joints, int32_t batch, int32_t num_joint_in, int32_t comps,
                                int32_t add_pelvis_neck, float* pose2d, void* stream);

/* The same contract WITHOUT the reduction: the reference's whole chain per sample (data/PW3D/dataset.py:236-250) -- tight bbox
 * (lib/coord_utils.py:21-39), process_bbox (:42-66), get_affine_transform with in-plane rotation rot_deg[b] (lib/aug_utils.py:140-173),
 * affine per joint, horizontal flip with left/right pairs when flip[b] != 0 (:31-38), /[res_w,res_h], standardisation.
 * valid[b] = 0 (and pose2d[b] = 0) where process_bbox returns None (a box under one pixel wide or high; the datasets drop such
 * samples).  rot_deg / flip / valid may be NULL (no rotation / no flip / not reported); res = (288, 384) in every reference config. */
int gator_preprocess_chain_f32(const float* joints, int32_t batch, int32_t num_joint_in, int32_t comps, int32_t add_pelvis_neck,
                               const float* rot_deg, const int32_t* flip, const int32_t* flip_pairs, int32_t n_pairs,
                               int32_t res_w, int32_t res_h, float* pose2d, int32_t* valid, void* stream);

/* "Next" row 8(f)-2: per-sample similarity (Procrustes) alignment of a onto b, [batch, n_points, 3] each
 * (rigid_transform_3D / rigid_align, lib/coord_utils.py:127-149: 3x3 SVD with the reflection fix, scale, translation);
 * the kernel under PA-MPJPE (data/PW3D/dataset.py:337-375). */
int gator_rigid_align_f32(const float* a, const float* b, int32_t batch, int32_t n_points, float* aligned, void* stream);

/* Multi-GPU: samples are independent, the batch is sharded contiguously over one process per GPU, and the path's one
 * collective is the all-gather of the predicted vertices [B/N,6890,3] (+ pose3d [B/N,J,3]) over xGMI.  gator_amd/parallel.py issues
 * it through torch.distributed ("nccl" = RCCL); these entry points do the same on RCCL directly for hosts without torch:
 * rank 0 calls gator_comm_unique_id and hands the 128 bytes to every rank (any host-side channel), every rank calls
 * gator_comm_create on its device, then gator_allgather_verts per step (rank-major outputs: rank r's rows at r*batch_local).
 * librccl is resolved at run time (the copy already loaded in the process, e.g. PyTorch's, else the system one), never linked. */
#define GATOR_COMM_ID_BYTES 128
typedef struct gator_comm gator_comm;
int gator_comm_unique_id(uint8_t* id /* [GATOR_COMM_ID_BYTES] */);
int gator_comm_create(const uint8_t* id, int32_t rank, int32_t world_size, gator_comm** out);
int gator_comm_destroy(gator_comm* comm);
int gator_allgather_verts(gator_comm* comm, const float* verts_local, const float* pose3d_local /* or NULL */, int32_t batch_local,
                          int32_t num_joint, float* verts_all, float* pose3d_all /* or NULL */, void* stream);

/* Measurement helper (tools/contention_model.py): the memory traffic a rank's share of an all-gather causes on ITS device --
 * read `bytes` at src, write them `copies` times to dst (dst holds copies x bytes) -- from `n_workgroups` workgroups of 256 threads on `stream`,
 * i.e. a stand-in for RCCL's send / receive kernels at a given channel count, to be run beside the forward on a one-GPU box.  No reference counterpart. */
int gator_emulate_gather_traffic(const void* src, void* dst, int64_t bytes, int32_t copies, int32_t n_workgroups, void* stream);

/* Evaluation errors of a batch in ONE launch (data/PW3D/dataset.py:273-286 compute_both_err, :337-375 PA alignment):
 *   pred_joints [B,n_joint,3] (scaled by pred_scale, e.g. 1000 for metres -> mm, lib/core/base.py:219), target_joints [B,n_joint,3];
 *   eval_joints: n_eval joint indices (device) or NULL for all; root: the joint both sets are aligned to.
 *   errors [B,2]: per sample (MPJPE, PA-MPJPE) = mean joint distance after root alignment, resp. after the similarity alignment of
 *   the evaluation joints; the caller averages over samples (or all-reduces the sums: gator_amd.parallel, mode='eval'). */
int gator_joint_errors_f32(const float* pred_joints, const float* target_joints, int32_t batch, int32_t n_joint,
                           const int32_t* eval_joints, int32_t n_eval, int32_t root, float pred_scale, float* errors, void* stream);

/* Host-side graph constants (no GPU needed).  Replace the absent Cython algos.pyx
 * (lib/models/backbones/setup.py:1-6) and lib/models/backbones/modules.py:6-29, lib/graph_utils.py:71-89. */
int gator_floyd_warshall(const float* adj, int32_t n, int64_t* dist, int64_t* path);
int gator_gen_edge_input(const int64_t* path, const float* edge_len, int32_t n, int32_t max_dist, float* out);
int gator_verts_joints_relation(const float* joints, int32_t n_joint, const float* verts, int32_t n_vert, int32_t* rel);

const char* gator_last_error(void);
const char* gator_version(void);
int gator_abi_version(void);      /* GATOR_ABI_VERSION of the library that is loaded */

#ifdef __cplusplus
}
#endif
#endif /* GATOR_HIP_H */
