/* gator_train.h -- C ABI of the training row: the fp32 primitives the training step of the GATOR path is
 * composed from on the device.  The reference trains with torch autograd over aten kernels (lib/core/base.py:122-183: forward,
 * five losses of lib/core/loss.py:10-118, loss.backward(), Adam); here every arithmetic operation of that step -- forward in
 * training mode, backward, loss, optimiser -- is one of the HIP kernels below, sequenced by gator_amd/train/ (autograd only
 * orders the calls).  All tensors are float32 device pointers described by a 4-D shape and ELEMENT strides (a stride of 0
 * broadcasts), so transposes, head splits and broadcast operands need no copies.
 *
 * Every function returns 0 on success, else a hipError_t-mapped code with the message in gator_last_error() (gator_hip.h). */
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* gator_stream;   /* hipStream_t */

/* elementwise, broadcasting: out = a (op) b.  op: 0 add, 1 sub, 2 mul, 3 div.   replaces aten add/sub/mul/div */
int gator_t_binary(int op, const float* a, const int64_t* stride_a, const float* b, const int64_t* stride_b, float* out,
                   const int64_t* stride_out, const int64_t* shape4, gator_stream stream);

/* out = a + b (+ c) (+ d), dense tensors of n elements: the gradient of a value with several consumers in one launch */
int gator_t_add_n(const float* a, const float* b, const float* c, const float* d, float* out, int64_t n, gator_stream stream);

/* elementwise: out = f(x; p0, p1).  op: 0 p0*x+p1, 1 gelu (erf form, torch F.gelu), 2 d gelu/dx, 3 exp, 4 rsqrt, 5 sqrt, 6 1/x,
 * 7 |x|, 8 sign, 9 p0**x, 10 x*x, 11 (x > p0 ? 1 : 0) */
int gator_t_unary(int op, const float* x, const int64_t* stride_x, float* out, const int64_t* stride_out, const int64_t* shape4,
                  float p0, float p1, gator_stream stream);

/* out[kept dims] (+)= sum over the dims with reduce4[d] != 0; out is contiguous over the kept dims in order.  Fixed summation
 * order, double accumulators.  ws: device scratch of gator_t_reduce_ws_bytes() bytes (may be NULL when that is 0). */
int64_t gator_t_reduce_ws_bytes(const int64_t* shape4, const int32_t* reduce4);
int gator_t_reduce_sum(const float* x, const int64_t* stride_x, const int64_t* shape4, const int32_t* reduce4, float* out,
                       int accumulate, void* ws, gator_stream stream);

/* C[b1,b2] = alpha * A[b1,b2] (M x K) . B[b1,b2] (K x N) (+ bias[n]) (+ C if accumulate), fp32-input MFMA (exact fp32 products).
 * strides in elements: A (m, k), B (k, n), C (m, n), and per batch level for each operand (0 broadcasts).
 * ksplit > 1 (needs nb1 == nb2 == 1): K is cut into ksplit slices summed in slice order through ws (ksplit*(M*N+M) floats).
 * a_rowsum != NULL (unbatched only): also a_rowsum[m] = alpha * sum_k A[m][k] from the tiles already in LDS - the bias gradient
 * rides on its weight-gradient GEMM (A = dY^T) instead of a reduction pass of its own. */
int gator_t_gemm(const float* A, const float* B, float* C, int M, int N, int K, const int64_t* stride_a2, const int64_t* stride_b2,
                 const int64_t* stride_c2, int nb1, int nb2, const int64_t* batch_a2, const int64_t* batch_b2,
                 const int64_t* batch_c2, const float* bias, float alpha, int accumulate, int ksplit, float* ws,
                 float* a_rowsum, gator_stream stream);

/* Grouped form: ONE launch for a list of independent unbatched products C_i = alpha_i * A_i . B_i (+ a_rowsum_i), each split over K as
 * its ksplit says (+ one launch that sums the slices).  The weight gradients of a backward pass are such a list: nothing reads them
 * before the optimiser, so they leave the dependent chain of activation gradients and fill the chip together.
 * gator_t_gemm_grouped_prepare fills the bookkeeping fields and returns the workspace size in floats (-1: bad problem);
 * table_host must stay unchanged until the upload has run (for a captured stream: as long as the graph is replayed). */
typedef struct gator_gemm_problem {
    const float *A, *B;
    float *C, *a_rowsum;
    int32_t M, N, K, ksplit;
    int64_t stride_a[2], stride_b[2], stride_c[2];   /* (m,k) (k,n) (m,n), elements */
    float alpha;
    int32_t accumulate;
    int32_t wg_begin, fin_begin;                     /* filled by _prepare */
    int64_t ws_off;                                  /* filled by _prepare */
    int32_t total_wgs, total_fin;                    /* filled by _prepare in entry 0 */
    const float* bias;                               /* optional [N], added to every row */
} gator_gemm_problem;
int64_t gator_t_gemm_grouped_prepare(gator_gemm_problem* problems, int n);
int gator_t_gemm_grouped(const gator_gemm_problem* table_host, int n, void* table_dev, float* ws, gator_stream stream);


/* Multi-head self-attention core with dropout on the probabilities (vanilla_transformer_encoder.py:36-46): o = dropout(softmax(scale q k^T)) v
 * for q, o of shape [B, T, H*D] and k, v [B, Tk, H*D] (head h in columns D*h ..; D = 32; Tk = T: the 431-token self-attention, Tk = J: the
 * cross-attention of vertex queries on joint keys, MDR.py:34-46), without materialising the [B,H,T,Tk] tensors; lse [B,H,T] is kept
 * for the backward, which recomputes the probabilities.  Masks: the Philox stream of gator_t_dropout at the flat index of a
 * contiguous [B,H,T,Tk] tensor (offset = 0 or rate = 0: no dropout).  dsum: scratch [B,H,T].  qsplit > 1 (few keys, many queries): dk / dv
 * are [qsplit][B,Tk,H*D] partial sums over interleaved query tiles, to be added up by the caller. */
int gator_t_attn_fwd(const float* q, const float* k, const float* v, float* o, float* lse, int B, int H, int T, int Tk, int D, float scale, float rate,
                     uint64_t seed, uint64_t offset, const uint64_t* step_counter, gator_stream stream);
int gator_t_attn_bwd(const float* q, const float* k, const float* v, const float* o, const float* lse, const float* d_o, float* dq, float* dk,
                     float* dv, float* dsum, int B, int H, int T, int Tk, int D, float scale, float rate, uint64_t seed, uint64_t offset,
                     const uint64_t* step_counter, int qsplit, gator_stream stream);

/* The GAT encoder's attention over J <= 32 joint tokens (modules.py:121-138): o = dropout(softmax(scale q k^T + bias)) v with q, k, v taken
 * from the in-projection's output qkv [B,J,3,H,D] (D = 16), bias [H,J,J]; o [B,J,H*D] head-major; P [B,H,J,J] kept for the backward, which
 * writes dqkv (layout of qkv) and dS [B,H,J,J] (its sum over the batch is the gradient of the bias).  One wave per (sample, head). */
int gator_t_attn_small_fwd(const float* qkv, const float* bias, float* o, float* P, int B, int H, int J, int D, float scale, float rate,
                           uint64_t seed, uint64_t offset, const uint64_t* step_counter, gator_stream stream);
int gator_t_attn_small_bwd(const float* qkv, const float* bias, const float* P, const float* d_o, float* dqkv, float* dS, int B, int H, int J, int D,
                           float scale, float rate, uint64_t seed, uint64_t offset, const uint64_t* step_counter, gator_stream stream);

/* MGCN (modules.py:243-255) after its two products h0 = x W[0], h1 = x W[1]: out = diag(adj) (M . h0) + offdiag(adj) @ (M . h1) + bias with the
 * symmetrised adjacency adj [J,J]; the backward writes d h0, d h1, the per-sample partials pm [B,J,C] (their sum over B is d M) and
 * dadj [B,J,J] (sum over B: d adj); d bias = column sums of d_out. */
int gator_t_mgcn_fwd(const float* h0, const float* h1, const float* adj, const float* M, const float* bias, float* out, int B, int J, int C,
                     gator_stream stream);
int gator_t_mgcn_bwd(const float* h0, const float* h1, const float* adj, const float* M, const float* d_out, float* dh0, float* dh1, float* pm,
                     float* dadj, int B, int J, int C, gator_stream stream);

/* nn.BatchNorm1d(C) in training mode on x [B,C,L] (the MDR head's BatchNorm1d(431) over [B,431,3], MDR.py:119,159): batch statistics over
 * (B, L), biased variance in the normalisation; run_mean / run_var (or NULL) updated with `momentum` and the unbiased variance, as torch does.
 * mean / rinv [C] are kept for the backward, which writes dx and the whole dw, db. */
int gator_t_batchnorm_fwd(const float* x, const float* w, const float* b, float* y, float* mean, float* rinv, float* run_mean, float* run_var, int B,
                          int C, int L, float eps, float momentum, gator_stream stream);
int gator_t_batchnorm_bwd(const float* dy, const float* x, const float* w, const float* mean, const float* rinv, float* dx, float* dw, float* db, int B,
                          int C, int L, gator_stream stream);

/* sizeof(gator_gemm_problem) (which = 0): lets a binding check its mirror of the struct */
int64_t gator_t_struct_size(int which);

/* rows of n contiguous floats.  mode 0: nn.LayerNorm (biased variance, eps inside the root); mode 1: the MDR LayerNorm
 * (lib/models/vanilla_transformer_encoder.py:31-34: unbiased std, eps added to the std).  w, b may be NULL (no affine).
 * forward saves mean[rows] and rinv[rows] (1/sqrt(var+eps) resp. 1/(std+eps)); backward writes dx and, if dy_xhat != NULL,
 * dy * xhat (whose column sums are the weight gradient). */
int gator_t_layernorm_fwd(const float* x, int64_t rows, int n, const float* w, const float* b, float eps, int mode, float* y,
                          float* mean, float* rinv, gator_stream stream);
int gator_t_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rinv, const float* w, int64_t rows,
                          int n, float eps, int mode, float* dx, float* dy_xhat, const float* add /* dx += add (the residual branch), or NULL */,
                          gator_stream stream);

/* softmax over rows of n contiguous floats, and its backward dx = p * (dp - sum(dp * p)) */
int gator_t_softmax_fwd(const float* x, int64_t rows, int n, float* p, gator_stream stream);
int gator_t_softmax_bwd(const float* p, const float* dp, int64_t rows, int n, float* dx, gator_stream stream);

/* dropout: keep[i] = philox4x32-7(seed, offset'; i) >= rate * 2^32; out = x * keep / (1 - rate); mask (uint8) is stored for
 * gator_t_mask_scale (the backward: out = x * mask * scale).  x == NULL writes the scaled mask itself (DropPath's per-sample factor).
 * offset' = offset + 2^32 * step_counter[0] when step_counter (a DEVICE uint64) is given: a step captured in a hipGraph draws new
 * masks on every replay.  gator_t_step_advance adds one to the counter (launch it once per step, inside the graph). */
int gator_t_dropout(const float* x, int64_t n, float rate, uint64_t seed, uint64_t offset, const uint64_t* step_counter, float* out,
                    uint8_t* mask, gator_stream stream);
int gator_t_step_advance(uint64_t* step_counter, gator_stream stream);
int gator_t_mask_scale(const float* x, const uint8_t* mask, int64_t n, float scale, float* out, gator_stream stream);

/* out = res + path[b] * dropout(act(x)) in one launch: act = GELU when gelu != 0; dropout as gator_t_dropout (rate / offset, same masks);
 * DropPath as gator_t_dropout(x = NULL) over the B = n / per_sample samples (path_rate / path_offset; factor kept in path_factor [B]);
 * res may be NULL; a rate of 0 or an offset of 0 switches that stage off.  _bwd: dx = g * path[b] * mask / (1 - rate) (* gelu'(x)). */
int gator_t_drop_fused(const float* x, const float* res, int64_t n, int64_t per_sample, int gelu, float rate, uint64_t seed, uint64_t offset,
                       float path_rate, uint64_t path_offset, const uint64_t* step_counter, float* out, uint8_t* mask, float* path_factor,
                       gator_stream stream);
int gator_t_drop_fused_bwd(const float* g, const float* x, const uint8_t* mask, const float* path_factor, int64_t n, int64_t per_sample, int gelu,
                           float rate, float* out, gator_stream stream);

/* torch.optim.Adam (lib/funcs_utils.py:91-95: lr only, betas 0.9/0.999, eps 1e-8, no weight decay) on one flat buffer */
int gator_t_adam(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr, double beta1,
                 double beta2, double eps, int step, const uint64_t* step_counter /* device; overrides `step` when given */,
                 gator_stream stream);

/* The mesh losses of lib/core/loss.py on device, value and gradient in one pass each.
 * coord (CoordLoss, loss.py:10-25):   mean |pred*valid - target*valid|; valid broadcast by strides (shape4 / element strides).
 * normal (NormalVectorLoss :59-86) and edge (EdgeLengthLoss :89-112) over faces [F,3] int32 of a [B,V,3] mesh.
 * inc_ptr [V+1] / inc_idx [3F]: for every vertex the list of (3*face + corner) entries it appears in (ascending), so the
 * gradient is GATHERED per vertex in a fixed order (no atomics, bit-reproducible).
 * Each writes loss_out[0] = weight * loss and, if grad != NULL, ACCUMULATES weight * d loss / d pred into grad (layout of pred).
 * ws: device scratch, gator_t_loss_ws_bytes(B, F) bytes. */
int64_t gator_t_loss_ws_bytes(int64_t B, int64_t F);   /* also covers the coord loss of B*F*9 elements or fewer: pass F >= numel/(9B) */
int gator_t_coord_loss(const float* pred, const float* target, const float* valid, const int64_t* stride_valid,
                       const int64_t* shape4, float weight, float* loss_out, float* grad, void* ws, gator_stream stream);
int gator_t_normal_loss(const float* pred, const float* target, const int32_t* faces, const int32_t* inc_ptr,
                        const int32_t* inc_idx, int64_t B, int64_t V, int64_t F, float weight, float* loss_out, float* grad,
                        void* ws, gator_stream stream);
int gator_t_edge_loss(const float* pred, const float* target, const int32_t* faces, const int32_t* inc_ptr,
                      const int32_t* inc_idx, int64_t B, int64_t V, int64_t F, float weight, float* loss_out, float* grad,
                      void* ws, gator_stream stream);

#ifdef __cplusplus
}
#endif
