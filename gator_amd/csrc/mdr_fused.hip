// MDR head (lib/models/MDR.py:124-170) as register-resident MFMA kernels (fp32 values; products on split 16-bit operands, see XA below).
//
// One wave owns one 32-token tile of one sample's 431 coarse-vertex tokens (14 tiles/sample) and keeps its 64-channel
// token state in registers in the MFMA accumulator layout (fused_common.h).  Everything that is row-wise in the
// reference -- LayerNorms, the cross-attention over the J joint tokens (MDR.py:34-46), the Mlp 64->256->64 (timm Mlp,
// MDR.py:61,68), the Annotated-Transformer LayerNorm (vanilla_transformer_encoder.py:31-34), the q/k/v in-projections and
// the out-projection + residual of the 431x431 self-attention (vanilla_transformer_encoder.py:82-94) -- chains through
// MFMAs without touching LDS.  The only cross-token dependency is the self-attention's K/V of the whole sample, so the
// three LBF layers become four stages - four launches (k_mdr_layer<MODE>) or, where the batch leaves a fractional generation of
// workgroups, ONE persistent launch that hands the same stages out as tickets (k_mdr_persist, further down):
//     L0: tokenise -> tokenwise(0)              (writes vf, Q, K, V of layer 0, all in operand-packed tiles)
//     L1: attention(0)+out-proj+res -> tokenwise(1)
//     L2: attention(1)+out-proj+res -> tokenwise(2)
//     L3: attention(2)+out-proj+res -> head features (motion_linear | bias_linear | scale_linear, MDR.py:156-162)
// Attention is flash-style: S^T = K Q^T per 32-key tile with the key on the accumulator row, online softmax in registers
// (one lane<->lane^32 max exchange per tile), and the probability registers are fed straight back as the B operand of
// O^T += V^T P^T.  K and V tiles are stored by the producer in exactly the operand order the consumer loads (1 KiB
// coalesced float4 wave loads from L2), so no LDS staging or barrier is needed.
#include "fused_common.h"
#include "fused_state.h"
#include "x3_common.h"

#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <type_traits>

namespace gator {
namespace {

constexpr float kLog2e = 1.4426950408889634f;
#ifndef MDR_X1_PIPE
#define MDR_X1_PIPE 1      // the one-plane form's attention and Mlp loops software-pipelined inside the wave (0: the plain loops, A/B)
#endif

// Diagnostic library only (GATOR_MDR_CUT, time-only experiments: results are meaningless): bit 0 = every weight load of a matrix reads its
// tile 0, bit 1 = every K / V load of the 431-key attention reads key tile 0 -- the loads stay, their L2 -> L1 traffic goes (L1 hits).
#ifdef GATOR_DIAG
__device__ int g_mdr_wmask = -1, g_mdr_kvmask = -1;
__device__ unsigned long long* g_persist_ends = nullptr;      // GATOR_MDR_ENDS=1: [workgroup][start, last ticket taken, end] wall-clock stamps (100 MHz) of k_mdr_persist
#define MDR_WIDX(i) ((i) & g_mdr_wmask)
#define MDR_KVIDX(i) ((i) & g_mdr_kvmask)
#else
#define MDR_WIDX(i) (i)
#define MDR_KVIDX(i) (i)
#endif

struct LayerW {   // packed tiles (MdrLayerP) + reference-layout vectors of one LBF layer
    const float *wq, *proj, *fc1, *fc2, *sa0, *sa1, *sa2, *sa3;
    const float *n1w, *n1b, *proj_b, *n2w, *n2b, *fc1_b, *fc2_b, *a2, *b2, *sa0_b, *sa1_b, *sa2_b, *sa3_b;
};

struct MdrArgs {
    int B, J, layer;
    const float *vf_in, *q_in, *k_in, *v_in;
    float *vf_out, *q_out, *k_out, *v_out;
    const float* jkv;        // [B][3][2][2][kTile]
    const float* pc;         // [B][J][133]  (stand-alone MDR entry) or nullptr when x_out is given
    const float* xout;       // [B][3J] pose3d in mm (full forward: pose_combine is never materialised)
    const int32_t* vj;       // [431]
    const float *tok_base, *tok_w3;
    const float *head_w, *head_b;
    float *hf, *lbf;
    LayerW prev, cur;        // prev: layer whose attention/out-proj runs first; cur: layer whose tokenwise part runs
    float lin_s, lin_inv;    // GATOR_MDR_X3=2: every token-wise linear returns lin_s x its value (x3_common.h: 4-product linears); lin_inv = 1 / lin_s
    // the head's Conv1d(431 -> 20, k3, p1) (MDR.py:121,163) as per-tile partial sums made by the tile that has the tokens' bias features in registers (round 6)
    double* hpart;           // [B][14][64] (60 used: row m, position l at m * 3 + l); nullptr: not computed (the A/B form with k_mdr_head)
    const float *bconv_w, *hbn_w, *hbn_b, *hbn_mean, *hbn_var;
    int halpha;
#ifdef GATOR_DIAG
    unsigned long long* stamps;   // diagnostic build only (libgator_hip_diag.so, GATOR_MDR_STAMPS=1)
#endif
};

// ---- row-wise helpers over the 64 channels (2 blocks) of a token (lane pair l, l^32) -------------------------------
__device__ __forceinline__ float row_sum64(const f32x16& a, const f32x16& b) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += a[r] + b[r];
    return s + xhalf(s);
}

// nn.LayerNorm(64), eps inside the sqrt
__device__ __forceinline__ void layernorm64(const f32x16 (&x)[2], const float* __restrict__ w, const float* __restrict__ b,
                                            int h, f32x16 (&y)[2]) {
    const float mean = row_sum64(x[0], x[1]) * (1.0f / 64.0f);
    f32x16 d0 = x[0] - mean, d1 = x[1] - mean;
    const float var = row_sum64(d0 * d0, d1 * d1) * (1.0f / 64.0f);
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    y[0] = d0 * rstd * load_chanvec_S(w, 0, h) + load_chanvec_S(b, 0, h);
    y[1] = d1 * rstd * load_chanvec_S(w, 32, h) + load_chanvec_S(b, 32, h);
}

// sum of squares of the 64 channel deviations of a token: four FMA chains per block instead of 32 products + 32 adds (and each term
// rounded once instead of twice)
__device__ __forceinline__ float row_sumsq64(const f32x16& a, const f32x16& b) {
    float p[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 16; ++r) p[r & 3] = __builtin_fmaf(a[r], a[r], p[r & 3]);
#pragma unroll
    for (int r = 0; r < 16; ++r) p[r & 3] = __builtin_fmaf(b[r], b[r], p[r & 3]);
    const float s = (p[0] + p[1]) + (p[2] + p[3]);
    return s + xhalf(s);
}
// nn.LayerNorm(64) with the affine part as one FMA per value: (d * rstd) * w + b.  w and b come from the workgroup's LDS table; when
// the result only feeds 4-product linears the table holds 16 w and 16 b (mdr_stage_vectors), so the result IS the operand scale.
__device__ __forceinline__ void layernorm64_L(const f32x16 (&x)[2], const float* w, const float* b, int h, f32x16 (&y)[2]) {
    const float mean = row_sum64(x[0], x[1]) * (1.0f / 64.0f);
    f32x16 d0 = x[0] - mean, d1 = x[1] - mean;
    const float var = row_sumsq64(d0, d1) * (1.0f / 64.0f);
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    y[0] = __builtin_elementwise_fma(d0 * rstd, chanvec_lds(w, 0, h), chanvec_lds(b, 0, h));
    y[1] = __builtin_elementwise_fma(d1 * rstd, chanvec_lds(w, 32, h), chanvec_lds(b, 32, h));
}
__device__ __forceinline__ void custom_ln64_L(f32x16 (&x)[2], const float* a2, const float* b2, int h) {
    const float mean = row_sum64(x[0], x[1]) * (1.0f / 64.0f);
    f32x16 d0 = x[0] - mean, d1 = x[1] - mean;
    const float std = sqrtf(row_sumsq64(d0, d1) * (1.0f / 63.0f));
    const float inv = 1.0f / (std + 1e-6f);
    x[0] = __builtin_elementwise_fma(chanvec_lds(a2, 0, h) * d0, f32x16(inv), chanvec_lds(b2, 0, h));
    x[1] = __builtin_elementwise_fma(chanvec_lds(a2, 32, h) * d1, f32x16(inv), chanvec_lds(b2, 32, h));
}

// Annotated-Transformer LayerNorm: a_2 * (x - mean) / (std_unbiased + 1e-6) + b_2
__device__ __forceinline__ void custom_ln64(f32x16 (&x)[2], const float* __restrict__ a2, const float* __restrict__ b2, int h) {
    const float mean = row_sum64(x[0], x[1]) * (1.0f / 64.0f);
    f32x16 d0 = x[0] - mean, d1 = x[1] - mean;
    const float std = sqrtf(row_sum64(d0 * d0, d1 * d1) * (1.0f / 63.0f));
    const float inv = 1.0f / (std + 1e-6f);
    x[0] = load_chanvec_S(a2, 0, h) * d0 * inv + load_chanvec_S(b2, 0, h);
    x[1] = load_chanvec_S(a2, 32, h) * d1 * inv + load_chanvec_S(b2, 32, h);
}

// y(T-layout, 64 ch) = W[64][64] x (+bias)
__device__ __forceinline__ void linear64_T(const float* __restrict__ Wp, const float* __restrict__ bias, const f32x16 (&x)[2],
                                           int lane, f32x16 (&y)[2]) {
    const int h = lane >> 5;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {   // one 32-term chain per k-block (interleaved), then one add
        f32x16 a0 = bias ? load_chanvec_S(bias, 32 * nb, h) : zero16(), a1 = zero16();
        mma2_T(load_wtile(Wp, nb * 2 + 0, lane), x[0], a0, load_wtile(Wp, nb * 2 + 1, lane), x[1], a1);
        y[nb] = a0 + a1;
    }
}

// Two waves share a SIMD.  A wave whose next instruction is an MFMA that cannot issue yet (matrix pipe busy) still wins the issue
// arbitration against a younger partner and starves the partner's VALU work (tools/microbench/helper_valu.hip: a VALU wave beside
// a wave of back-to-back MFMAs makes NO progress at equal priority, full speed at priority 1 - and the MFMAs still issue every 32
// cycles).  So a wave raises its priority while it runs VALU sections (softmax, splits, LayerNorm, GELU) and drops it for its
// MFMA bursts: whoever has vector work gets the issue slots, the matrix pipe is fed from the gaps.
#define MDR_PRIO_VALU() __builtin_amdgcn_s_setprio(1)
#define MDR_PRIO_MFMA() __builtin_amdgcn_s_setprio(0)
#define MDR_PIN()                            \
    do {                                     \
        asm volatile("" ::: "memory");       \
        __builtin_amdgcn_sched_barrier(0);   \
    } while (0)

// one 32-key tile of the flash attention: scores, online softmax, P.V into accumulator OACC
#define ATTN_TILE(KT, KB, VB, OACC, OACB)                                                                       \
    {                                                                                                       \
        f32x16 S = dot16(KB, qv, zero16());   /* S^T[key][query], two interleaved 8-step chains */            \
        float bm = -1e30f;                                                                                  \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                    \
            float sc = S[r] * c;                                                                            \
            if ((KT) == kVT - 1 && kap(r) + 4 * h >= kV - 32 * (kVT - 1)) sc = -1e30f; /* keys 431..447 do not exist */ \
            S[r] = sc;                                                                                      \
            bm = fmaxf(bm, sc);                                                                             \
        }                                                                                                   \
        bm = fmaxf(bm, xhalf(bm));                                                                          \
        if (!__all(bm <= m + 8.0f)) { /* lazy rescale (wave-uniform): P stays <= 2^8, exact in fp32 */      \
            const float mn = fmaxf(m, bm);                                                                  \
            const float al = __builtin_amdgcn_exp2f(m - mn);                                                \
            O = O * al;                                                                                     \
            O2 = O2 * al;                                                                                   \
            O3 = O3 * al;                                                                                   \
            O4 = O4 * al;                                                                                   \
            l *= al;                                                                                        \
            m = mn;                                                                                         \
        }                                                                                                   \
        float ps = 0.f;                                                                                     \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                    \
            const float pe = __builtin_amdgcn_exp2f(S[r] - m);                                              \
            S[r] = pe;                                                                                      \
            ps += pe;                                                                                       \
        }                                                                                                   \
        l += ps;                                                                                            \
        /* O^T[d][query] += V^T[d][key] P^T[key][query] */                                                  \
        _Pragma("unroll") for (int r = 0; r < 16; r += 2) {                                                 \
            OACC = GATOR_MFMA(VB[r], S[r], OACC);                                                           \
            OACB = GATOR_MFMA(VB[r + 1], S[r + 1], OACB);                                                   \
        }                                                                                                   \
    }

// ---- flash attention of one 32-query tile against the 431 keys of its sample, one head --------------------------------
// Even key tiles accumulate into chains O/O2 (even/odd key of the tile), odd tiles into O3/O4: four fp32 chains of ~110 products,
// and no two consecutive MFMAs write the same accumulator.  (Explicit K/V double buffering was measured:
// no gain -- the co-resident wave already covers the tile loads -- and it costs 32 VGPRs.)
__device__ __forceinline__ f32x16 self_attention_head(const float* __restrict__ qt, const float* __restrict__ kbase,
                                                      const float* __restrict__ vbase, int lane) {
    const int h = lane >> 5;
    const f32x16 qv = load_block(qt, lane);
    f32x16 O = zero16(), O2 = zero16(), O3 = zero16(), O4 = zero16();
    float m = -1e30f, l = 0.f;
    const float c = kLog2e * 0.17677669529663688110f;      // log2(e) / sqrt(d_k): scores kept in the exp2 domain
#pragma unroll 1
    for (int kt = 0; kt < kVT - 2; kt += 4) {               // 12 tiles in 3 trips of 4 (chain kt&3), then the last two
        {
            const f32x16 kb = load_block(kbase + (size_t)kt * 2 * kTile, lane), vb = load_block(vbase + (size_t)kt * 2 * kTile, lane);
            ATTN_TILE(kt, kb, vb, O, O2)
        }
        {
            const f32x16 kb = load_block(kbase + (size_t)(kt + 1) * 2 * kTile, lane), vb = load_block(vbase + (size_t)(kt + 1) * 2 * kTile, lane);
            ATTN_TILE(kt + 1, kb, vb, O3, O4)
        }
        {
            const f32x16 kb = load_block(kbase + (size_t)(kt + 2) * 2 * kTile, lane), vb = load_block(vbase + (size_t)(kt + 2) * 2 * kTile, lane);
            ATTN_TILE(kt + 2, kb, vb, O, O2)
        }
        {
            const f32x16 kb = load_block(kbase + (size_t)(kt + 3) * 2 * kTile, lane), vb = load_block(vbase + (size_t)(kt + 3) * 2 * kTile, lane);
            ATTN_TILE(kt + 3, kb, vb, O3, O4)
        }
    }
    {
        const f32x16 kb = load_block(kbase + (size_t)(kVT - 2) * 2 * kTile, lane), vb = load_block(vbase + (size_t)(kVT - 2) * 2 * kTile, lane);
        ATTN_TILE(kVT - 2, kb, vb, O, O2)
    }
    {
        const f32x16 kb = load_block(kbase + (size_t)(kVT - 1) * 2 * kTile, lane), vb = load_block(vbase + (size_t)(kVT - 1) * 2 * kTile, lane);
        ATTN_TILE(kVT - 1, kb, vb, O3, O4)
    }
    l += xhalf(l);
    return ((O + O2) + (O3 + O4)) * (1.0f / l);
}

// ---- the same on split-precision operands (x3_common.h): Q, K, V arrive as X3 tiles, S^T = K Q^T and O^T += V^T P^T are 12
// bf16 MFMAs each (768 cycles per key tile against 2048), the probabilities are split in registers.  The bf16 MFMA sums
// 16 products internally and its dependent chain issues back to back, so two accumulators (even / odd key tiles) suffice.
#define ATTN_TILE_X3(KT, KB, VB, OACC)                                                                      \
    {                                                                                                       \
        f32x16 S = x3_mma(KB, qx, zero16());  /* S^T[key][query] */                                         \
        float bm = -1e30f;                                                                                  \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {   /* Q arrives pre-scaled by log2(e)/sqrt(d_k) */ \
            float sc = S[r];                                                                                \
            if ((KT) == kVT - 1 && kap(r) + 4 * h >= kV - 32 * (kVT - 1)) sc = -1e30f;                      \
            S[r] = sc;                                                                                      \
            bm = fmaxf(bm, sc);                                                                             \
        }                                                                                                   \
        bm = fmaxf(bm, xhalf(bm));                                                                          \
        if (!__all(bm <= m + 8.0f)) {                                                                       \
            const float mn = fmaxf(m, bm);                                                                  \
            const float al = __builtin_amdgcn_exp2f(m - mn);                                                \
            O = O * al;                                                                                     \
            O2 = O2 * al;                                                                                   \
            l *= al;                                                                                        \
            m = mn;                                                                                         \
        }                                                                                                   \
        float ps = 0.f;                                                                                     \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                    \
            const float pe = __builtin_amdgcn_exp2f(S[r] - m);                                              \
            S[r] = pe;                                                                                      \
            ps += pe;                                                                                       \
        }                                                                                                   \
        l += ps;                                                                                            \
        OACC = x3_mma(VB, x3_split(S), OACC);   /* O^T[d][query] += V^T[d][key] P^T[key][query] */          \
    }
__device__ __forceinline__ f32x16 self_attention_head_x3(const float* __restrict__ qt, const float* __restrict__ kbase,
                                                         const float* __restrict__ vbase, int lane) {
    const int h = lane >> 5;
    const X3 qx = x3_load(qt, lane);
    f32x16 O = zero16(), O2 = zero16();
    float m = -1e30f, l = 0.f;
    X3 kb = x3_load(kbase, lane), vb = x3_load(vbase, lane);
#pragma unroll 1
    for (int kt = 0; kt < kVT; kt += 2) {                   // tiles kt (-> O) and kt + 1 (-> O2); the next tile's K/V in flight
        X3 kn = x3_load(kbase + (size_t)(kt + 1) * 2 * kTileX3, lane), vn = x3_load(vbase + (size_t)(kt + 1) * 2 * kTileX3, lane);
        ATTN_TILE_X3(kt, kb, vb, O)
        const int k2 = kt + 2 < kVT ? kt + 2 : kt;
        kb = x3_load(kbase + (size_t)k2 * 2 * kTileX3, lane);
        vb = x3_load(vbase + (size_t)k2 * 2 * kTileX3, lane);
        ATTN_TILE_X3(kt + 1, kn, vn, O2)
    }
    l += xhalf(l);
    return (O + O2) * (1.0f / l);
}

// ---- the same on two-plane fp16 operands (x3_common.h, "X2"): 6 MFMAs per product instead of 12, 3 VALU ops per split value
// instead of 5.5, 4 KiB tiles instead of 6.  Q and K arrive scaled by 16 (Q also by log2(e)/sqrt(d_k)), so the accumulator holds
// 256 x the score: the running maximum is kept in that domain and the 2^-8 rides on the FMA that forms the exponent.  V arrives
// scaled by 16 and the probabilities carry an extra 2^6 (so that their low plane stays a normal fp16 number); both cancel in
// O / (16 l).
constexpr float kX2QK = 16.0f, kX2V = 16.0f;
#define ATTN_PV(VB, PX) { O2 = x2_mma_small(VB, PX, O2); O = x2_mma_main(VB, PX, O); }
#define ATTN_TILE_X2(KT, KB, VB)                                                                            \
    {                                                                                                       \
        MDR_PRIO_MFMA();                                                                                    \
        f32x16 S = x2_mma(KB, qx, zero16());  /* 256 x S^T[key][query] */                                   \
        MDR_PRIO_VALU();                                                                                    \
        float bm = -1e30f;                                                                                  \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                    \
            float sc = S[r];                                                                                \
            if ((KT) == kVT - 1 && kap(r) + 4 * h >= kV - 32 * (kVT - 1)) sc = -1e30f;                      \
            S[r] = sc;                                                                                      \
            bm = fmaxf(bm, sc);                                                                             \
        }                                                                                                   \
        bm = fmaxf(bm, xhalf(bm));                                                                          \
        if (!__all(bm <= m + 2048.0f)) {      /* lazy rescale: P stays <= 2^8 (x 2^6 below) */               \
            const float mn = fmaxf(m, bm);                                                                  \
            const float al = __builtin_amdgcn_exp2f((m - mn) * 0.00390625f);                                \
            O = O * al;                                                                                     \
            O2 = O2 * al;                                                                                   \
            l *= al;                                                                                        \
            m = mn;                                                                                         \
        }                                                                                                   \
        const float off = 6.0f - m * 0.00390625f;                                                           \
        float ps = 0.f;                                                                                     \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                    \
            const float pe = __builtin_amdgcn_exp2f(fmaf(S[r], 0.00390625f, off));                          \
            S[r] = pe;                                                                                      \
            ps += pe;                                                                                       \
        }                                                                                                   \
        l += ps;                                                                                            \
        const X2 px_ = x2_split(S);                                                                         \
        MDR_PRIO_MFMA();                                                                                    \
        /* O^T[d][query] += V^T[d][key] P^T[key][query]: the hi*hi products of every key tile into O, the cross products (2^-11 of    \
           it) into O2 -- O is rounded 28 times at full magnitude over the 14 tiles instead of 42 times for each of two equal halves.   \
           (Measured and not taken: a third accumulator for the odd tiles' hi*hi products spills; the same split for the MLP's fc2      \
           accumulators fits in exactly 256 registers, makes the launch 8 % slower and moves the error by nothing.) */                   \
        ATTN_PV(VB, px_)                                                                                    \
    }
template <bool kActScale16>
__device__ __forceinline__ f32x16 self_attention_head_x2(const float* __restrict__ qt, const float* __restrict__ kbase,
                                                         const float* __restrict__ vbase, int lane) {
    const int h = lane >> 5;
    const X2 qx = x2_load(qt, lane);
    f32x16 O = zero16(), O2 = zero16();
    float m = -1e30f, l = 0.f;
    X2 kb = x2_load(kbase, lane), vb = x2_load(vbase, lane);
    // tiles kt and kt + 1 per trip, the next tile's K/V in flight.  The last pair is peeled so that the mask of the 17 keys that
    // do not exist (431 = 13 x 32 + 15) is compile-time there and absent from the loop (it cost 5 selects per tile as a runtime test).
    // (Round 4, measured and dropped: tiles after the first WITHOUT the row maximum -- probabilities against the running reference,
    // only their row sums inspected (a lane's 16 probabilities are below their sum, so "sum <= 2^14" proves the fp16 range), the
    // whole tile redone the long way where that fails.  14 of ~120 VALU instructions fewer per tile, same results to rounding, the
    // large-logit test green -- and the launch 12 us SLOWER.)
#pragma unroll 1
    for (int kt = 0; kt < kVT - 2; kt += 2) {
        X2 kn = x2_load(kbase + (size_t)MDR_KVIDX(kt + 1) * 2 * kTile, lane), vn = x2_load(vbase + (size_t)MDR_KVIDX(kt + 1) * 2 * kTile, lane);
        ATTN_TILE_X2(0, kb, vb)
        kb = x2_load(kbase + (size_t)MDR_KVIDX(kt + 2) * 2 * kTile, lane);
        vb = x2_load(vbase + (size_t)MDR_KVIDX(kt + 2) * 2 * kTile, lane);
        ATTN_TILE_X2(0, kn, vn)
    }
    {
        X2 kn = x2_load(kbase + (size_t)MDR_KVIDX(kVT - 1) * 2 * kTile, lane), vn = x2_load(vbase + (size_t)MDR_KVIDX(kVT - 1) * 2 * kTile, lane);
        ATTN_TILE_X2(kVT - 2, kb, vb)
        ATTN_TILE_X2(kVT - 1, kn, vn)
    }
    l += xhalf(l);
    return (O + O2) * (((kActScale16 ? 16.0f : 1.0f) / kX2V) / l);      // kActScale16: 16 x the head's output, the operand scale of the out-projection
}


// ---- the same loop, software-pipelined by one key tile inside the wave (round 5; MDR_ATTN_PIPE=1, tools/microbench/attn_pipe.hip) ------
//     block A:  lo plane of P[kt-1] (deferred), O += V[kt-1] P[kt-1] (hi.hi first: it needs no lo plane)  ||  row maximum of S[kt]
//     rescale (rare, wave-uniform)
//     block B:  S[kt+1] = K[kt+1] Q  ||  exp2, row sum, hi plane of P[kt]
// so that every MFMA is followed IN THE WAVE'S OWN STREAM by independent vector work.  Same arithmetic, same order of every
// accumulation: bitwise the results of self_attention_head_x2.  Alone the loop takes 16 % fewer cycles (776 -> 649 per key tile
// and SIMD) -- and the chip gives most of it back as clock on real data (profiles/r05_microbench_attn_pipe.txt).
#ifndef MDR_ATTN_PIPE
#define MDR_ATTN_PIPE 0
#endif
template <bool kActScale16>
__device__ __forceinline__ f32x16 self_attention_head_x2_pipe(const float* __restrict__ qt, const float* __restrict__ kbase,
                                                              const float* __restrict__ vbase, int lane) {
    const int h = lane >> 5;
    const X2 qx = x2_load(qt, lane);
    f32x16 O = zero16(), O2 = zero16();
    float m = -1e30f, l = 0.f;
    X2 kb0 = x2_load(kbase, lane), kb1 = x2_load(kbase + (size_t)MDR_KVIDX(1) * 2 * kTile, lane);
    X2 vb0 = x2_load(vbase, lane), vb1 = x2_load(vbase + (size_t)MDR_KVIDX(1) * 2 * kTile, lane);
    f32x16 S0 = x2_mma(kb0, qx, zero16()), S1 = zero16();          // S[0]; "P[-1]" = 0 in fp32 and in its hi plane
    kb0 = x2_load(kbase + (size_t)MDR_KVIDX(2) * 2 * kTile, lane);
    f16x8 ph[2] = {f16x8(0), f16x8(0)};
    auto step = [&](auto last_, const int kt, f32x16& Sc, f32x16& Sp, X2& Vp, X2& Kn) {
        constexpr bool LAST = decltype(last_)::value;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (LAST) {                                      // keys 431..447 do not exist
#pragma unroll
            for (int r = 0; r < 16; ++r) if (kap(r) + 4 * h >= kV - 32 * (kVT - 1)) Sc[r] = -1e30f;
        }
        f16x8 plo[2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) plo[s][j] = (_Float16)(Sp[8 * s + j] - (float)ph[s][j]);
        O = GATOR_MFMA_F16(Vp.p[0][0], ph[0], O);
        O = GATOR_MFMA_F16(Vp.p[0][1], ph[1], O);
        O2 = GATOR_MFMA_F16(Vp.p[1][0], ph[0], O2);
        O2 = GATOR_MFMA_F16(Vp.p[0][0], plo[0], O2);
        O2 = GATOR_MFMA_F16(Vp.p[1][1], ph[1], O2);
        O2 = GATOR_MFMA_F16(Vp.p[0][1], plo[1], O2);
        float bm = -1e30f;
#pragma unroll
        for (int r = 0; r < 16; ++r) bm = fmaxf(bm, Sc[r]);
        bm = fmaxf(bm, xhalf(bm));
        const bool calm = __all(bm <= m + 2048.0f);
        __builtin_amdgcn_sched_barrier(0);
        Vp = x2_load(vbase + (size_t)MDR_KVIDX(kt + 1 < kVT ? kt + 1 : kVT - 1) * 2 * kTile, lane);
        __builtin_amdgcn_sched_barrier(0);
        if (!calm) {
            const float mn = fmaxf(m, bm);
            const float al = __builtin_amdgcn_exp2f((m - mn) * 0.00390625f);
            O = O * al;
            O2 = O2 * al;
            l *= al;
            m = mn;
        }
        __builtin_amdgcn_sched_barrier(0);
        Sp = x2_mma(Kn, qx, zero16());
        const float off = 6.0f - m * 0.00390625f;
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pe = __builtin_amdgcn_exp2f(fmaf(Sc[r], 0.00390625f, off));
            Sc[r] = pe;
            ps += pe;
        }
        l += ps;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) ph[s][j] = (_Float16)Sc[8 * s + j];
        __builtin_amdgcn_sched_barrier(0);
        Kn = x2_load(kbase + (size_t)MDR_KVIDX(kt + 3 < kVT ? kt + 3 : kVT - 1) * 2 * kTile, lane);
        __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll 1
    for (int kt = 0; kt < kVT - 2; kt += 2) {
        step(std::false_type(), kt, S0, S1, vb1, kb1);
        step(std::false_type(), kt + 1, S1, S0, vb0, kb0);
    }
    step(std::false_type(), kVT - 2, S0, S1, vb1, kb1);
    step(std::true_type(), kVT - 1, S1, S0, vb0, kb0);
    {   // P[13] V[13]: fp32 probabilities in S1, V[13] in vb1
        f16x8 plo[2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) plo[s][j] = (_Float16)(S1[8 * s + j] - (float)ph[s][j]);
        O = GATOR_MFMA_F16(vb1.p[0][0], ph[0], O);
        O = GATOR_MFMA_F16(vb1.p[0][1], ph[1], O);
        O2 = GATOR_MFMA_F16(vb1.p[1][0], ph[0], O2);
        O2 = GATOR_MFMA_F16(vb1.p[0][0], plo[0], O2);
        O2 = GATOR_MFMA_F16(vb1.p[1][1], ph[1], O2);
        O2 = GATOR_MFMA_F16(vb1.p[0][1], plo[1], O2);
    }
    l += xhalf(l);
    return (O + O2) * (((kActScale16 ? 16.0f : 1.0f) / kX2V) / l);
}

// ---- ONE fp16 plane ("X1", BASELINE config 3: the MDR layers in 16-bit operand mode, XA == 3) --------------------------------------------
// Activations, Q, K, V and the probabilities are ONE fp16 plane of 16 x value (64 x for P): no split, 2 KiB tiles, 2 MFMAs per 32-deep
// product in the attention cores (against 6) and 4 per token-wise product (weights on their two leading fp16 planes, 22 bits: against
// 8); accumulation, softmax, norms, GELU and the residual stream stay fp32.  What that costs in accuracy is the activation rounding
// (2^-12 relative per operand element): tools/emulate_16bit.py, profiles/r05_emulate_16bit.txt (sub-millimetre vertices).
constexpr int kTileX1 = kTile / 2;
// One key tile.  The VALU work per tile is what bounds this form (4 MFMAs against ~50 vector instructions), so the softmax is cut to
// exp2 + row sum + one conversion per pair:
//   * Q arrives scaled by log2(e) / sqrt(d_k) and K unscaled, so the MFMA delivers the score in the exp2 domain, and the running
//     reference rides on the accumulator's initial value (Ci = 6 - m in every register: the 2^6 keeps P's fp16 image normal), so the
//     exponent is S itself -- no scale / offset instruction per value;
//   * no row maximum: the probabilities are non-negative, so "the lane's row sum < 2^15" proves every one of them is inside fp16's range;
//     where that fails (first tile, a tile whose scores jump by 2^9, anything non-finite) the tile is redone the long way: raw scores,
//     maximum, rescale of O and l, new reference.  Wave-uniform and rare.
#define ATTN_TILE_X1(KT, KB, VB)                                                                            \
    {                                                                                                       \
        f32x16 S = x1_mma(KB, qx, Ci);        /* S^T[key][query] - m + 6, exp2 domain */                    \
        if ((KT) == kVT - 1) {                                                                              \
            _Pragma("unroll") for (int r = 0; r < 16; ++r)                                                  \
                if (kap(r) + 4 * h >= kV - 32 * (kVT - 1)) S[r] = -1e30f;   /* keys 431..447 do not exist */ \
        }                                                                                                   \
        float ps = 0.f;                                                                                     \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                    \
            S[r] = __builtin_amdgcn_exp2f(S[r]);                                                            \
            ps += S[r];                                                                                     \
        }                                                                                                   \
        if (!__all(ps < 32768.0f)) {                                                                        \
            f32x16 R = x1_mma(KB, qx, zero16());                                                            \
            float bm = -1e30f;                                                                              \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                \
                if ((KT) == kVT - 1 && kap(r) + 4 * h >= kV - 32 * (kVT - 1)) R[r] = -1e30f;                \
                bm = fmaxf(bm, R[r]);                                                                       \
            }                                                                                               \
            bm = fmaxf(bm, xhalf(bm));                                                                      \
            const float mn = fmaxf(m, bm);                                                                  \
            const float al = __builtin_amdgcn_exp2f(m - mn);                                                \
            O = O * al;                                                                                     \
            l *= al;                                                                                        \
            m = mn;                                                                                         \
            Ci = f32x16(6.0f - m);                                                                          \
            ps = 0.f;                                                                                       \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                \
                S[r] = __builtin_amdgcn_exp2f(R[r] + Ci[r]);                                                \
                ps += S[r];                                                                                 \
            }                                                                                               \
        }                                                                                                   \
        l += ps;                                                                                            \
        O = x1_mma(VB, x1_cvt(S), O);                                                                       \
    }
template <bool kActScale16>
__device__ __forceinline__ f32x16 self_attention_head_x1(const float* __restrict__ qt, const float* __restrict__ kbase,
                                                         const float* __restrict__ vbase, int lane) {
    const int h = lane >> 5;
    const X1 qx = x1_load(qt, lane);
    f32x16 O = zero16();
    float m = -1e30f, l = 0.f;
#if !MDR_X1_PIPE
    f32x16 Ci = f32x16(1e30f);                 // "6 - m" of an empty history: the first tile overflows its row sums and takes the long way
#endif
#if MDR_X1_PIPE
    // Software-pipelined by one key tile (unrolled by two: fixed register names): the RAW scores of tile kt + 1 are issued before the
    // exponentials of tile kt, so the MFMAs run under the vector work of the same wave.  (Here the reference is added per value -- one
    // v_add more than the accumulator-initial-value form below, which needs a 16-register tile per head on top of the two score tiles
    // and spills; this form is bound by latency, not by its vector instruction count.)
    X1 kA = x1_load(kbase, lane), vA = x1_load(vbase, lane);
    X1 kB = x1_load(kbase + (size_t)MDR_KVIDX(1) * 2 * kTileX1, lane), vB = x1_load(vbase + (size_t)MDR_KVIDX(1) * 2 * kTileX1, lane);
    f32x16 SA = x1_mma(kA, qx, zero16()), SB;
    float ci = 1e30f;                           // 6 - m; an empty history overflows the first tile's row sums: it takes the long way
    auto step = [&](auto last_, const int kt, f32x16& Sc, f32x16& Sn, X1& Kc, const X1& Kn, X1& Vc) {
        constexpr int KT = decltype(last_)::value;      // kVT - 1 for the last tile (compile-time mask), else 0
        if (KT != kVT - 1) Sn = x1_mma(Kn, qx, zero16());     // raw scores of tile kt + 1: independent of everything below
        if (KT == kVT - 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) if (kap(r) + 4 * h >= kV - 32 * (kVT - 1)) Sc[r] = -1e30f;     // keys 431..447 do not exist
        }
        float ps = 0.f;
        if (__all(ci < 1e29f)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float pe = __builtin_amdgcn_exp2f(Sc[r] + ci); ps += pe; Sc[r] = pe; }
        } else ps = 1e30f;
        if (!__all(ps < 32768.0f)) {            // (the raw scores are gone where the fast way ran: back from K)
            f32x16 R = x1_mma(Kc, qx, zero16());
            float bm = -1e30f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (KT == kVT - 1 && kap(r) + 4 * h >= kV - 32 * (kVT - 1)) R[r] = -1e30f;
                bm = fmaxf(bm, R[r]);
            }
            bm = fmaxf(bm, xhalf(bm));
            const float mn = fmaxf(m, bm);
            const float al = __builtin_amdgcn_exp2f(m - mn);
            O = O * al;
            l *= al;
            m = mn;
            ci = 6.0f - m;
            ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { Sc[r] = __builtin_amdgcn_exp2f(R[r] + ci); ps += Sc[r]; }
        }
        l += ps;
        O = x1_mma(Vc, x1_cvt(Sc), O);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 2 < kVT) {                             // tile kt + 2 into the buffers tile kt has just left
            Kc = x1_load(kbase + (size_t)MDR_KVIDX(kt + 2) * 2 * kTileX1, lane);
            Vc = x1_load(vbase + (size_t)MDR_KVIDX(kt + 2) * 2 * kTileX1, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll 1
    for (int kt = 0; kt < kVT - 2; kt += 2) {
        step(std::integral_constant<int, 0>(), kt, SA, SB, kA, kB, vA);
        step(std::integral_constant<int, 0>(), kt + 1, SB, SA, kB, kA, vB);
    }
    step(std::integral_constant<int, 0>(), kVT - 2, SA, SB, kA, kB, vA);
    step(std::integral_constant<int, kVT - 1>(), kVT - 1, SB, SA, kB, kA, vB);
#else
    X1 kb = x1_load(kbase, lane), vb = x1_load(vbase, lane);
#pragma unroll 1
    for (int kt = 0; kt < kVT - 2; kt += 2) {
        X1 kn = x1_load(kbase + (size_t)MDR_KVIDX(kt + 1) * 2 * kTileX1, lane), vn = x1_load(vbase + (size_t)MDR_KVIDX(kt + 1) * 2 * kTileX1, lane);
        ATTN_TILE_X1(0, kb, vb)
        kb = x1_load(kbase + (size_t)MDR_KVIDX(kt + 2) * 2 * kTileX1, lane);
        vb = x1_load(vbase + (size_t)MDR_KVIDX(kt + 2) * 2 * kTileX1, lane);
        ATTN_TILE_X1(0, kn, vn)
    }
    {
        X1 kn = x1_load(kbase + (size_t)MDR_KVIDX(kVT - 1) * 2 * kTileX1, lane), vn = x1_load(vbase + (size_t)MDR_KVIDX(kVT - 1) * 2 * kTileX1, lane);
        ATTN_TILE_X1(kVT - 2, kb, vb)
        ATTN_TILE_X1(kVT - 1, kn, vn)
    }
#endif
    l += xhalf(l);
    return O * (((kActScale16 ? 16.0f : 1.0f) / kX2V) / l);
}

// ---- cross-attention over the J joint tokens (keys/values precomputed per sample by k_mdr_joint) -------------------------
__device__ __forceinline__ f32x16 cross_attention_head(const float* __restrict__ kj, const float* __restrict__ vjp,
                                                       const f32x16& qh, int J, int lane) {
    const int h = lane >> 5;
    const f32x16 kb = load_block(kj, lane);
    f32x16 S = dot16(kb, qh, zero16());                                    // S^T[joint][token]
    const float c = kLog2e * 0.17677669529663688110f;                      // head_dim ** -0.5 (MDR.py:25), exp2 domain
    float mx = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float s = (kap(r) + 4 * h < J) ? S[r] * c : -1e30f;
        S[r] = s;
        mx = fmaxf(mx, s);
    }
    mx = fmaxf(mx, xhalf(mx));
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(S[r] - mx);
        S[r] = p;
        sum += p;
    }
    sum += xhalf(sum);
    const float inv = 1.0f / sum;
    const f32x16 vb = load_block(vjp, lane);
    return dot16(vb, S * inv, zero16());
}

// The same on two fp16 planes (GATOR_MDR_X3=2): the joint K/V tiles arrive as X2 operand tiles scaled by 16 (k_gat_joint /
// k_mdr_joint), q and the probabilities are split in registers (x 16, x 64) - 24 fp16 MFMAs of 32 cycles per tile instead of 64
// fp32-input MFMAs of 64 cycles.  Like the 431x431 attention this rounds its operands to 22 bits; a softmax average over 17 joints.
__device__ __forceinline__ f32x16 cross_attention_head_x2(const float* __restrict__ kj, const float* __restrict__ vjp,
                                                          const f32x16& qh, float qscale /* 16 / (the factor qh carries) */, int J, int lane) {
    const int h = lane >> 5;
    const X2 kx = x2_load(kj, lane);
    f32x16 S = x2_mma(kx, x2_split(qh * qscale), zero16());                // 256 x S^T[joint][token]
    const X2 vx = x2_load(vjp, lane);                                      // in flight during the softmax
    const float c = kLog2e * 0.17677669529663688110f * (1.0f / 256.0f);    // head_dim ** -0.5 (MDR.py:25), exp2 domain
    float mx = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float s = (kap(r) + 4 * h < J) ? S[r] * c : -1e30f;
        S[r] = s;
        mx = fmaxf(mx, s);
    }
    mx = fmaxf(mx, xhalf(mx));
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(S[r] - mx);
        S[r] = p;
        sum += p;
    }
    sum += xhalf(sum);
    const float inv = 64.0f / sum;                                         // probabilities x 64: low plane stays fp16-normal
    return x2_mma(vx, x2_split(S * inv), zero16()) * (1.0f / 64.0f);      // 16 x the head's output: the operand scale of the projection that follows
}


// The same on one fp16 plane (XA == 3): the hi planes of the joint K / V tiles, q and the probabilities rounded once
__device__ __forceinline__ f32x16 cross_attention_head_x1(const float* __restrict__ kj, const float* __restrict__ vjp,
                                                          const f32x16& qh, float qscale, int J, int lane) {
    const int h = lane >> 5;
    const X1 kx = x1_load(kj, lane);
    f32x16 S = x1_mma(kx, x1_cvt(qh * qscale), zero16());                  // 256 x S^T[joint][token]
    const X1 vx = x1_load(vjp, lane);
    const float c = kLog2e * 0.17677669529663688110f * (1.0f / 256.0f);
    float mx = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float s = (kap(r) + 4 * h < J) ? S[r] * c : -1e30f;
        S[r] = s;
        mx = fmaxf(mx, s);
    }
    mx = fmaxf(mx, xhalf(mx));
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(S[r] - mx);
        S[r] = p;
        sum += p;
    }
    sum += xhalf(sum);
    const float inv = 64.0f / sum;
    return x1_mma(vx, x1_cvt(S * inv), zero16()) * (1.0f / 64.0f);        // 16 x the head's output
}

// ---- weight stream of the tokenwise part: two buffers of one tile pair each (2 x 32 VGPRs).  The pair for the NEXT
// product is requested right after the current product's MFMAs are queued, so its L2 latency hides behind them and
// behind the co-resident wave.  MDR_PIN keeps the order (memory ops and scheduler).
// Q/K/V tile store in the attention's operand form: 0 fp32 block, 1 exact three-plane bf16 (X3), 2 two-plane fp16 (X2)
template <int X> __device__ __forceinline__ void st_op(float* p, int lane, const f32x16& v) {
    if constexpr (X == 1) x3_store(p, lane, x3_split(v));
    else if constexpr (X == 2) x2_store(p, lane, x2_split(v));
    else if constexpr (X == 3) x1_store(p, lane, x1_cvt(v));
    else store_block(p, lane, v);
}
struct W2 { WTile t[2]; };
__device__ __forceinline__ W2 ldw2(const float* __restrict__ Wp, int i0, int i1, int lane) {
    W2 w;
    w.t[0] = load_wtile(Wp, i0, lane);
    w.t[1] = load_wtile(Wp, i1, lane);
    return w;
}
// 64-term contraction as two independent 32-term chains (one per k-block)
__device__ __forceinline__ f32x16 lin2_T(const W2& w, const f32x16 (&x)[2], f32x16 init) {
    f32x16 a1 = zero16();
    mma2_T(w.t[0], x[0], init, w.t[1], x[1], a1);
    return init + a1;
}
__device__ __forceinline__ f32x16 lin2_C(const W2& w, const f32x16 (&x)[2]) {
    f32x16 a0 = zero16(), a1 = zero16();
    mma2_C(w.t[0], x[0], a0, w.t[1], x[1], a1);
    return a0 + a1;
}

// X: self-attention on split-precision operands (q/k/v tiles are X3 tiles, 1.5x the size, same tile indices)
// split-precision forms: the weight pair is two X3 tiles (48 VGPRs), activations are split once per linear input, one chain
struct W2X { X3 t[2]; };
__device__ __forceinline__ W2X ldw2x(const float* __restrict__ Wx, int i0, int i1, int lane) {
    W2X w;
    w.t[0] = x3_load(Wx + (size_t)i0 * kTileX3, lane);
    w.t[1] = x3_load(Wx + (size_t)i1 * kTileX3, lane);
    return w;
}
__device__ __forceinline__ f32x16 lin2_T(const W2X& w, const X3 (&x)[2], f32x16 init) { return x3_mma(w.t[1], x[1], x3_mma(w.t[0], x[0], init)); }
__device__ __forceinline__ f32x16 lin2_C(const W2X& w, const X3 (&x)[2]) { return x3_mma(x[1], w.t[1], x3_mma(x[0], w.t[0], zero16())); }
// XA == 2: weights as two H3 tiles (three exact fp16 planes, 48 VGPRs), activations as X2 (two fp16 planes of 16 x value)
constexpr float kActScale = 16.0f;
struct W2H { H3 t[2]; };
__device__ __forceinline__ W2H ldw2h(const float* __restrict__ Wx, int i0, int i1, int lane) {
    W2H w;
    w.t[0] = h3_load(Wx + (size_t)i0 * kTileX3, lane);
    w.t[1] = h3_load(Wx + (size_t)i1 * kTileX3, lane);
    return w;
}
// a 64-deep product: the twelve cross products of both tiles first (accumulator still at bias magnitude), the four hi*hi products last
__device__ __forceinline__ f32x16 lin2_T(const W2H& w, const X2 (&x)[2], f32x16 init) {
    return h3_mma_wa_main(w.t[1], x[1], h3_mma_wa_main(w.t[0], x[0], h3_mma_wa_small(w.t[1], x[1], h3_mma_wa_small(w.t[0], x[0], init))));
}
__device__ __forceinline__ f32x16 lin2_C(const W2H& w, const X2 (&x)[2]) {
    return h3_mma_aw_main(x[1], w.t[1], h3_mma_aw_main(x[0], w.t[0], h3_mma_aw_small(x[1], w.t[1], h3_mma_aw_small(x[0], w.t[0], zero16()))));
}

// XA == 3: weights as the two leading fp16 planes (hi, mid: 22 bits) of two H3 tiles (32 VGPRs), activations as X1
struct W2G { G2 t[2]; };
__device__ __forceinline__ W2G ldw2g(const float* __restrict__ Wx, int i0, int i1, int lane) {
    W2G w;
    w.t[0] = g2_load(Wx + (size_t)i0 * kTileX3, lane);
    w.t[1] = g2_load(Wx + (size_t)i1 * kTileX3, lane);
    return w;
}
__device__ __forceinline__ f32x16 lin2_T(const W2G& w, const X1 (&x)[2], f32x16 init) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s) init = GATOR_MFMA_F16(w.t[t].mid[s], x[t].p[s], init);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s) init = GATOR_MFMA_F16(w.t[t].hi[s], x[t].p[s], init);
    return init;
}
__device__ __forceinline__ f32x16 lin2_C(const W2G& w, const X1 (&x)[2]) {
    f32x16 acc = zero16();
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s) acc = GATOR_MFMA_F16(x[t].p[s], w.t[t].mid[s], acc);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s) acc = GATOR_MFMA_F16(x[t].p[s], w.t[t].hi[s], acc);
    return acc;
}

template <int XA> struct TokOp;
template <> struct TokOp<0> { typedef W2 W; typedef f32x16 A; };
template <> struct TokOp<1> { typedef W2X W; typedef X3 A; };
template <> struct TokOp<2> { typedef W2H W; typedef X2 A; };
template <> struct TokOp<3> { typedef W2G W; typedef X1 A; };
template <int XA> __device__ __forceinline__ typename TokOp<XA>::W ldw(const float* __restrict__ Wp, int i0, int i1, int lane) {
    i0 = MDR_WIDX(i0); i1 = MDR_WIDX(i1);
    if constexpr (XA == 3) return ldw2g(Wp, i0, i1, lane); else if constexpr (XA == 2) return ldw2h(Wp, i0, i1, lane); else if constexpr (XA == 1) return ldw2x(Wp, i0, i1, lane); else return ldw2(Wp, i0, i1, lane);
}
// operand form of an accumulator tile holding `pre` x its value (pre = 1, or 1 / lin_s when it is a 4-product linear's raw output)
template <int XA> __device__ __forceinline__ typename TokOp<XA>::A mk(const f32x16& v, float pre = 1.0f) {
    if constexpr (XA == 3) return x1_cvt(v * (kActScale * pre)); else if constexpr (XA == 2) return x2_split(v * (kActScale * pre)); else if constexpr (XA == 1) return x3_split(v); else return v;
}

// ---- GELU by table (XA == 3 only).  The exact GELU is a quarter of the one-plane tile's vector work (one degree-8 polynomial + exp2 per
// value, in packed fp32 that the SIMD issues at half rate); its result is rounded to ONE fp16 plane right afterwards, so Phi(x) is read
// from an LDS table instead: 3 072 (value, forward difference) pairs on [-6, 6) in steps of 1 / 256, linear interpolation -- error of Phi
// below 4.6e-7 (h^2 / 8 max|Phi''|), against 2.4e-4 |x| for the fp16 rounding that follows; |x| >= 6 clamps to Phi = 0 / 1 (exact to 1e-9).
// Seven vector instructions and one ds_read_b64 per value instead of ~14 issue slots.  The fp32 configuration keeps the polynomial.
#ifndef MDR_X1_GELU_TABLE
#define MDR_X1_GELU_TABLE 1
#endif
constexpr bool kX1GeluTable = MDR_X1_GELU_TABLE != 0;
constexpr int kGeluTab = 3072;
__device__ __forceinline__ void gelu_table_fill(float* GT) {      // cooperative (256 threads); the caller puts a barrier behind it
    for (int e = threadIdx.x; e < kGeluTab; e += 256) {
        const float x0 = (float)(e - kGeluTab / 2) * (1.0f / 256.0f), x1 = x0 + (1.0f / 256.0f);
        const float p0 = 0.5f * (1.0f + erf_fast(x0 * 0.70710678118654752440f)), p1 = 0.5f * (1.0f + erf_fast(x1 * 0.70710678118654752440f));
        GT[2 * e] = p0;
        GT[2 * e + 1] = e + 1 < kGeluTab ? p1 - p0 : 0.f;
    }
}
// v holds S x value (k = 1 / S): returns S x GELU(value)
__device__ __forceinline__ void gelu_tile_table(f32x16& v, float k, const float* GT) {
    const float ks = k * 256.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float t = __builtin_fmaf(v[r], ks, (float)(kGeluTab / 2));
        t = __builtin_amdgcn_fmed3f(t, 0.0f, (float)kGeluTab - 0.0005f);
        const float fl = __builtin_floorf(t);
        const f32x2 e = *reinterpret_cast<const f32x2*>(GT + 2 * (int)fl);
        v[r] = v[r] * __builtin_fmaf(t - fl, e[1], e[0]);
    }
}

// MODE 0: tokenise + tokenwise(0) ; 1: attention + tokenwise ; 2: attention + head features
// XA   0: everything on the fp32-input MFMA ; 1: split precision (exact bf16 x 3, six partial products) everywhere ; 2 (the default):
//         token-wise linears on four partial products (weights exact on three fp16 planes, activations on two: x3_common.h), the
//         431x431 self-attention and the J-joint cross-attention on two fp16 planes
// per-channel vectors of a stage (biases, norm weights) staged once per workgroup in LDS
enum { VO_SA3B = 0, VO_N1W = 64, VO_N1B = 128, VO_PROJB = 192, VO_N2W = 256, VO_N2B = 320, VO_FC2B = 384, VO_A2 = 448, VO_B2 = 512,
       VO_SA0B = 576, VO_SA1B = 640, VO_HEADB = 704, VO_FC1B = 768, VO_TOKW3 = 1024, VO_TOTAL = 1216 };
// ---- the head's Conv1d(431 -> 20, kernel 3, padding 1 over the xyz axis; MDR.py:121,163) as per-tile partial sums ------------------------------
// Until round 6 k_mdr_head did the whole conv per sample: a 15 us launch of its own whose conv loop and cross-lane sums were most of its chain.  The tile
// that computes a token's head features has the conv's input -- the token's three bias features -- in registers, so it leaves the tile's 60 partial sums
// (20 rows x 3 positions over its 32 tokens) behind; the finish (head_finish: sum of the 14 partials in tile order, softmax-mix per token) is light.
// Everything in double: each product exact, fixed association -- the result does not depend on who computes it.
// The finish stays a launch of its own (11 us against k_mdr_head's 15): inside the persistent launch -- as a fifth ticket stage, or run by the workgroup that
// publishes a sample's last tile -- it measured +8 .. +51 us, because the last ~18 samples of an XCD finish together at the launch's end and their heads then
// stand behind the last tile instead of beside each other (DESIGN 4c'', profiles/r06_fused_head_stage.txt, docs/history/r06_inlaunch_head.patch).
__device__ __forceinline__ double dpp_mov_f64t(double v, int which) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    int lo = (int)(unsigned)u, hi = (int)(unsigned)(u >> 32);
    if (which == 0) { lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xf, 0xf, false); }             // quad_perm [1,0,3,2]
    else if (which == 1) { lo = __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xf, 0xf, false); }        // quad_perm [2,3,0,1]
    else if (which == 2) { lo = __builtin_amdgcn_update_dpp(lo, lo, 0x141, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x141, 0xf, 0xf, false); }      // row_half_mirror
    else { lo = __builtin_amdgcn_update_dpp(lo, lo, 0x140, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x140, 0xf, 0xf, false); }                      // row_mirror
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
// Sum of 32 values per lane over the 32 lanes of its half (lanes 0..31 | 32..63) as a reduce-scatter: at step s a lane hands the half of its list that its partner
// keeps (row mirror, half-row mirror, quad reverse, quad swap, v_permlane16_swap) and adds what it receives to the half it keeps -- 16 + 8 + 4 + 2 + 1 exchanges
// instead of 32 x 5, and lane p ends with the total of entry slot32(p).  A fixed tree: the result does not depend on who runs it.
// the partner's value at step s.  The partner must hold the SAME entries as the lane, i.e. differ from it only in bits that have not been decided yet: the row
// mirror (15 - i: flips bits 0..3, decided: bit 3), the half-row mirror (7 - i: bits 0..2, decided: bit 2), the quad reverse (3 - i: bits 0, 1, decided: bit 1), the
// quad swap (bit 0), v_permlane16_swap (bit 4)
__device__ __forceinline__ double lane_xchg_f64(double v, int step) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    int lo = (int)(unsigned)u, hi = (int)(unsigned)(u >> 32);
    if (step == 0) { lo = __builtin_amdgcn_update_dpp(lo, lo, 0x140, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x140, 0xf, 0xf, false); }           // row_mirror
    else if (step == 1) { lo = __builtin_amdgcn_update_dpp(lo, lo, 0x141, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x141, 0xf, 0xf, false); }      // row_half_mirror
    else if (step == 2) { lo = __builtin_amdgcn_update_dpp(lo, lo, 0x1B, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x1B, 0xf, 0xf, false); }        // quad_perm [3,2,1,0]
    else if (step == 3) { lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xf, 0xf, false); }        // quad_perm [1,0,3,2]
    else {
        const auto rl = __builtin_amdgcn_permlane16_swap((unsigned)lo, (unsigned)lo, false, false), rh = __builtin_amdgcn_permlane16_swap((unsigned)hi, (unsigned)hi, false, false);
        const bool up = (threadIdx.x & 16) != 0;      // [0]: the value of the lane with bit 4 clear, [1]: with bit 4 set
        lo = (int)(up ? rl[0] : rl[1]); hi = (int)(up ? rh[0] : rh[1]);
    }
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
// the lane keeps the upper half of its list iff the step's deciding bit of its own index is set; its partner (the other value of that bit) keeps the other half
template <int N, int STEP, int BIT>
__device__ __forceinline__ void rs_step(double (&v)[32], int lane) {
    const bool up = ((lane >> BIT) & 1) != 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double send = up ? v[i] : v[N + i], keep = up ? v[N + i] : v[i];
        v[i] = keep + lane_xchg_f64(send, STEP);
    }
}
// -> the lane's total and (slot) which of the 32 entries it is the total of
__device__ __forceinline__ double half_reduce_scatter32(double (&v)[32], int lane, int& slot) {
    rs_step<16, 0, 3>(v, lane);
    rs_step<8, 1, 2>(v, lane);
    rs_step<4, 2, 1>(v, lane);
    rs_step<2, 3, 0>(v, lane);
    rs_step<1, 4, 4>(v, lane);
    slot = 16 * ((lane >> 3) & 1) + 8 * ((lane >> 2) & 1) + 4 * ((lane >> 1) & 1) + 2 * (lane & 1) + ((lane >> 4) & 1);
    return v[0];
}
// lane (token, h) of a T-layout tile: `x` = the token's bias features after bias_norm + GELU (MDR.py:159-160; zeros for a token that does not exist).  Half h makes
// rows m = 10 h .. 10 h + 9: position l of row m collects tap k of the input at l + k - 1.  `dst`: the tile's 64 doubles, entry (m, l) at m * 3 + l.
__device__ __forceinline__ void head_conv_partial(const float* __restrict__ bconv_w, int token, const float (&x)[3], int lane, double* __restrict__ dst) {
    const int h = lane >> 5;
    const float* wp = bconv_w + (size_t)(10 * h) * (kV * 3) + 3 * (token < kV ? token : 0);
    double v[32];
#pragma unroll
    for (int mm = 0; mm < 10; ++mm) {      // per token in fp32 (three products each: the shipped head's lanes ran 21-term fp32 chains), across tokens and tiles in double
        const float w0 = wp[mm * (kV * 3)], w1 = wp[mm * (kV * 3) + 1], w2 = wp[mm * (kV * 3) + 2];
        v[mm * 3 + 0] = (double)fmaf(w2, x[1], w1 * x[0]);
        v[mm * 3 + 1] = (double)fmaf(w2, x[2], fmaf(w1, x[1], w0 * x[0]));
        v[mm * 3 + 2] = (double)fmaf(w1, x[2], w0 * x[1]);
    }
    v[30] = 0.0; v[31] = 0.0;
    int slot;
    const double tot = half_reduce_scatter32(v, lane, slot);
    if (slot < 30) dst[30 * h + slot] = tot;
}
// bias_norm (BatchNorm1d(431) over the vertex axis in eval mode, or LayerNorm(3) in the alpha variant) + GELU of a token's three bias features (MDR.py:159-160)
__device__ __forceinline__ void head_bias_act(bool alpha, const float* bn_w, const float* bn_b, const float* bn_mean, const float* bn_var, int v, float (&x)[3]) {
    if (alpha) {      // LayerNorm(3)
        const float m = (x[0] + x[1] + x[2]) / 3.0f;
        const float qq = ((x[0] - m) * (x[0] - m) + (x[1] - m) * (x[1] - m) + (x[2] - m) * (x[2] - m)) / 3.0f;
        const float rs = 1.0f / sqrtf(qq + 1e-5f);
#pragma unroll
        for (int c = 0; c < 3; ++c) x[c] = (x[c] - m) * rs * bn_w[c] + bn_b[c];
    } else {          // BatchNorm1d(431) eval: channel = vertex
        const float rs = 1.0f / sqrtf(bn_var[v] + 1e-5f);
#pragma unroll
        for (int c = 0; c < 3; ++c) x[c] = (x[c] - bn_mean[v]) * rs * bn_w[v] + bn_b[v];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) x[c] = gelu_f(x[c]);
}

constexpr int kParkF4 = 8 * 256;       // f32x4 slots of the residual-stream parking area (split-precision forms only)

// cooperative (256 threads); the caller puts a barrier behind it
template <int MODE, int XA>
__device__ __forceinline__ void mdr_stage_vectors(const MdrArgs& a, float* VT) {
    for (int e = threadIdx.x; e < VO_TOTAL / 4; e += 256) {
        const int off = 4 * e;
        const float* src = nullptr;
        if (off < VO_N1W) { if (MODE > 0) src = a.prev.sa3_b + off; }
        else if (off >= VO_HEADB && off < VO_HEADB + 32) { if (MODE == 2) src = a.head_b + (off - VO_HEADB); }
        else if (off >= VO_TOKW3) { if (MODE == 0) src = a.tok_w3 + (off - VO_TOKW3); }
        else if (MODE < 2 && off < VO_HEADB) {
            const float* tab[10] = {a.cur.n1w, a.cur.n1b, a.cur.proj_b, a.cur.n2w, a.cur.n2b, a.cur.fc2_b, a.cur.a2, a.cur.b2, a.cur.sa0_b, a.cur.sa1_b};
            src = tab[(off - VO_N1W) >> 6] + (off & 63);
        } else if (MODE < 2 && off >= VO_FC1B && off < VO_TOKW3) src = a.cur.fc1_b + (off - VO_FC1B);
        if (src) {
            f32x4 v = *reinterpret_cast<const f32x4*>(src);
            // 4-product linears return lin_s x their value: the biases that start or join their accumulators carry the factor too
            if (XA >= 2 && (off < VO_N1W || (off >= VO_PROJB && off < VO_N2W) || (off >= VO_FC2B && off < VO_A2) || (off >= VO_SA0B && off < VO_TOKW3)))
                v = v * a.lin_s;
            // norm1 / norm2 only feed 4-product linears: their affine part carries the operand scale, so LayerNorm's FMA delivers 16 x value
            if (XA >= 2 && ((off >= VO_N1W && off < VO_PROJB) || (off >= VO_N2W && off < VO_FC2B))) v = v * kActScale;
            reinterpret_cast<f32x4*>(VT)[e] = v;
        }
    }
}

// one wave, one 32-token tile `id` = sample * 14 + tile of the sample
template <int MODE, int XA>
__device__ __forceinline__ void mdr_tile(const MdrArgs& a, const int id, const float* VT, f32x4* park, const float* GT = nullptr, const MdrArgs* a_mem = nullptr) {      // a_mem: the same arguments IN MEMORY (kernel-argument segment)
    constexpr bool X = XA != 0;
    constexpr int TQ = XA == 1 ? kTileX3 : (XA == 3 ? kTileX1 : kTile);
    auto park_vf = [&](const f32x16 (&v)[2]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            f32x4 t4;
#pragma unroll
            for (int j = 0; j < 4; ++j) t4[j] = v[i >> 2][4 * (i & 3) + j];
            park[i * 256 + threadIdx.x] = t4;
        }
    };
    auto unpark_vf = [&](f32x16 (&v)[2]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x4 t4 = park[i * 256 + threadIdx.x];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[i >> 2][4 * (i & 3) + j] = t4[j];
        }
    };
    // `lane` is opaque to the optimiser: inside k_mdr_persist's ticket loops every address that depends only on the lane (the
    // weight tiles) would otherwise be loop-invariant, hoisted in front of the loop and kept alive across it (470 B of scratch)
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
    const int h = lane >> 5;
    if (id >= a.B * kVT) return;
#ifdef GATOR_DIAG
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = a.stamps ? clock64() : 0;
    const unsigned long long rt0 = a.stamps ? wall_clock64() : 0, ck0 = st_last;
#define MDR_STAMP(i)                                      \
    __builtin_amdgcn_sched_barrier(0);                    \
    if (a.stamps) {                                       \
        const unsigned long long now_ = clock64();        \
        st_acc[i] += now_ - st_last;                      \
        st_last = now_;                                   \
    }
#else
#define MDR_STAMP(i)
#endif
    constexpr bool H = XA >= 2;                 // 4-product (XA 2) / 2-product (XA 3) linears: their raw outputs carry the factor a.lin_s
    const float inv = H ? a.lin_inv : 1.0f;
    const int b = id / kVT, t = id % kVT;
    const size_t tile = ((size_t)b * kVT + t) * 2;          // index of this wave's first block in vf/q/k/v
    const int token = 32 * t + (lane & 31);
    const LayerW& w = a.cur;
    f32x16 vf[2];
    typename TokOp<XA>::W A, B;
    typedef typename TokOp<XA>::A Act;
    if (MODE == 0) {
        A = ldw<XA>(w.wq, 0, 1, lane);
        B = ldw<XA>(w.wq, 2, 3, lane);
        // verts tokens = Linear(6->64)([v431, pose3d[vj]/1000]) + pos_v   (MDR.py:126-137); the v431/bias/pos part is folded
        const int tk = token < kV ? token : kV - 1;
        float x0, x1, x2;
        if (a.xout) {             // pose3d / 1000 (GATOR.py:19), same fp32 division as the reference
            const float* p3 = a.xout + ((size_t)b * a.J + a.vj[tk]) * 3;
            x0 = p3[0] / 1000.f; x1 = p3[1] / 1000.f; x2 = p3[2] / 1000.f;
        } else {
            const float* p3 = a.pc + ((size_t)b * a.J + a.vj[tk]) * 133 + 2;
            x0 = p3[0]; x1 = p3[1]; x2 = p3[2];
        }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            f32x16 v = load_block(a.tok_base + ((size_t)t * 2 + nb) * kTile, lane);
            v += chanvec_lds(VT, VO_TOKW3 + 32 * nb, h) * x0;
            v += chanvec_lds(VT, VO_TOKW3 + 64 + 32 * nb, h) * x1;
            v += chanvec_lds(VT, VO_TOKW3 + 128 + 32 * nb, h) * x2;
            vf[nb] = v;
        }
    } else {
        f32x16 att[2];
        if constexpr (XA == 3) {
            att[0] = self_attention_head_x1<true>(a.q_in + (tile + 0) * TQ, a.k_in + ((size_t)b * kVT * 2 + 0) * TQ,
                                    a.v_in + ((size_t)b * kVT * 2 + 0) * TQ, lane);
            att[1] = self_attention_head_x1<true>(a.q_in + (tile + 1) * TQ, a.k_in + ((size_t)b * kVT * 2 + 1) * TQ,
                                    a.v_in + ((size_t)b * kVT * 2 + 1) * TQ, lane);
        } else if constexpr (XA == 2) {
#if MDR_ATTN_PIPE
            att[0] = self_attention_head_x2_pipe<true>(a.q_in + (tile + 0) * TQ, a.k_in + ((size_t)b * kVT * 2 + 0) * TQ,
                                    a.v_in + ((size_t)b * kVT * 2 + 0) * TQ, lane);
            att[1] = self_attention_head_x2_pipe<true>(a.q_in + (tile + 1) * TQ, a.k_in + ((size_t)b * kVT * 2 + 1) * TQ,
                                    a.v_in + ((size_t)b * kVT * 2 + 1) * TQ, lane);
#else
            att[0] = self_attention_head_x2<true>(a.q_in + (tile + 0) * TQ, a.k_in + ((size_t)b * kVT * 2 + 0) * TQ,
                                    a.v_in + ((size_t)b * kVT * 2 + 0) * TQ, lane);
            att[1] = self_attention_head_x2<true>(a.q_in + (tile + 1) * TQ, a.k_in + ((size_t)b * kVT * 2 + 1) * TQ,
                                    a.v_in + ((size_t)b * kVT * 2 + 1) * TQ, lane);
#endif
        } else if constexpr (XA == 1) {
            att[0] = self_attention_head_x3(a.q_in + (tile + 0) * TQ, a.k_in + ((size_t)b * kVT * 2 + 0) * TQ,
                                            a.v_in + ((size_t)b * kVT * 2 + 0) * TQ, lane);
            att[1] = self_attention_head_x3(a.q_in + (tile + 1) * TQ, a.k_in + ((size_t)b * kVT * 2 + 1) * TQ,
                                            a.v_in + ((size_t)b * kVT * 2 + 1) * TQ, lane);
        } else {
            att[0] = self_attention_head(a.q_in + (tile + 0) * kTile, a.k_in + ((size_t)b * kVT * 2 + 0) * kTile,
                                         a.v_in + ((size_t)b * kVT * 2 + 0) * kTile, lane);
            att[1] = self_attention_head(a.q_in + (tile + 1) * kTile, a.k_in + ((size_t)b * kVT * 2 + 1) * kTile,
                                         a.v_in + ((size_t)b * kVT * 2 + 1) * kTile, lane);
        }
        A = ldw<XA>(a.prev.sa3, 0, 1, lane);
        B = ldw<XA>(a.prev.sa3, 2, 3, lane);
        vf[0] = load_block(a.vf_in + (tile + 0) * kTile, lane);
        vf[1] = load_block(a.vf_in + (tile + 1) * kTile, lane);
        MDR_PIN();
        MDR_STAMP(0)
        // linears[-1] + residual (vanilla_transformer_encoder.py:94, MDR.py:143)
        Act attx[2];
        if constexpr (XA == 3) { attx[0] = x1_cvt(att[0]); attx[1] = x1_cvt(att[1]); }
        else if constexpr (XA == 2) { attx[0] = x2_split(att[0]); attx[1] = x2_split(att[1]); }      // the heads come out at 16 x value already
        else { attx[0] = mk<XA>(att[0]); attx[1] = mk<XA>(att[1]); }
        const f32x16 y0 = lin2_T(A, attx, chanvec_lds(VT, VO_SA3B, h));
        if (MODE == 1) A = ldw<XA>(w.wq, 0, 1, lane); else if (XA != 3) A = ldw<XA>(a.head_w, 0, 1, lane);
        MDR_PIN();
        const f32x16 y1 = lin2_T(B, attx, chanvec_lds(VT, VO_SA3B + 32, h));
        if (MODE == 1) B = ldw<XA>(w.wq, 2, 3, lane);
        MDR_PIN();
        if constexpr (H) { vf[0] = fma16(y0, inv, vf[0]); vf[1] = fma16(y1, inv, vf[1]); }
        else { vf[0] += y0; vf[1] += y1; }
    }
    if (MODE == 2) {
        if (a.lbf != nullptr && token < kV) {      // the "mdr_lbf2" tap (110 KB per sample): only when taps are recorded
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v4[j] = vf[nb][4 * g + j];
                    *reinterpret_cast<f32x4*>(a.lbf + ((size_t)b * kV + token) * kE + 32 * nb + 8 * g + 4 * h) = v4;
                }
        }
        f32x16 acc;
        if constexpr (XA == 3) {
            // the head features (mat_A | bias_linear | scale_linear | mat_C, MDR.py:156-162) keep the fp32 configuration's operands even in
            // 16-bit mode: mat_C goes straight into the coarse vertices, and this is 8 MFMAs of a tile's ~350 (profiles/r05_emulate_16bit.txt, C3d)
            const X2 vfx2[2] = {x2_split(vf[0] * kActScale), x2_split(vf[1] * kActScale)};
            const W2H Ah = ldw2h(a.head_w, MDR_WIDX(0), MDR_WIDX(1), lane);
            acc = lin2_T(Ah, vfx2, chanvec_lds(VT, VO_HEADB, h));
        } else {
            const Act vfx[2] = {mk<XA>(vf[0]), mk<XA>(vf[1])};
            acc = lin2_T(A, vfx, chanvec_lds(VT, VO_HEADB, h));
        }
        if constexpr (H) acc = acc * inv;
        if (token < kV) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v4;
#pragma unroll
                for (int j = 0; j < 4; ++j) v4[j] = acc[4 * g + j];
                *reinterpret_cast<f32x4*>(a.hf + ((size_t)b * kV + token) * 32 + 8 * g + 4 * h) = v4;
            }
        }
        {   // the head conv's partial sums over this tile's tokens (their bias features are channels 24..26 = registers 12..14 of the lower half's lanes)
            const MdrArgs* ah = a_mem;        // (taking the address of a by-value kernel argument would copy all of it to scratch)
            asm volatile("" : "+s"(ah));      // its fields are fetched here, behind the tile body: nothing of them lives across it
            if (ah != nullptr && ah->hpart) {
                float x[3] = {acc[12], acc[13], acc[14]};
                if (h == 0 && token < kV) head_bias_act(ah->halpha != 0, ah->hbn_w, ah->hbn_b, ah->hbn_mean, ah->hbn_var, token, x);
                else { x[0] = x[1] = x[2] = 0.f; }
#pragma unroll
                for (int c = 0; c < 3; ++c) {      // the upper half works on the same tokens: rows 10..19
                    const float other = xhalf(x[c]);
                    x[c] = h ? other : x[c];
                }
                head_conv_partial(ah->bconv_w, token, x, lane, ah->hpart + ((size_t)b * kVT + t) * 64);
            }
        }
        return;
    }
    MDR_STAMP(1)
    // ---- CrossAttentionBlock (MDR.py:64-69) ----
    {
        f32x16 q[2], o[2];
        Act fz[2];
        {
            f32x16 fzf[2];
            layernorm64_L(vf, VT + VO_N1W, VT + VO_N1B, h, fzf);
            if constexpr (XA == 3) { fz[0] = x1_cvt(fzf[0]); fz[1] = x1_cvt(fzf[1]); }
            else if constexpr (XA == 2) { fz[0] = x2_split(fzf[0]); fz[1] = x2_split(fzf[1]); }      // already 16 x value (staged 16 w, 16 b)
            else { fz[0] = mk<XA>(fzf[0]); fz[1] = mk<XA>(fzf[1]); }
        }
        const float* jb = a.jkv + (((size_t)b * 3 + a.layer) * 4) * kTile;       // [k/v][head] tiles
        q[0] = lin2_T(A, fz, zero16());
        A = ldw<XA>(w.proj, 0, 1, lane);
        MDR_PIN();
        q[1] = lin2_T(B, fz, zero16());
        B = ldw<XA>(w.proj, 2, 3, lane);
        MDR_PIN();
#pragma unroll
        for (int hd = 0; hd < 2; ++hd) {
            if constexpr (XA == 3) o[hd] = cross_attention_head_x1(jb + hd * kTile, jb + (2 + hd) * kTile, q[hd], 16.0f * inv, a.J, lane);
            else if constexpr (XA == 2) o[hd] = cross_attention_head_x2(jb + hd * kTile, jb + (2 + hd) * kTile, q[hd], 16.0f * inv, a.J, lane);
            else o[hd] = cross_attention_head(jb + hd * kTile, jb + (2 + hd) * kTile, q[hd], a.J, lane);
        }
        Act ox[2];
        if constexpr (XA == 3) { ox[0] = x1_cvt(o[0]); ox[1] = x1_cvt(o[1]); }
        else if constexpr (XA == 2) { ox[0] = x2_split(o[0]); ox[1] = x2_split(o[1]); }      // cross_attention_head_x2 returns 16 x value
        else { ox[0] = mk<XA>(o[0]); ox[1] = mk<XA>(o[1]); }
        const f32x16 y0 = lin2_T(A, ox, chanvec_lds(VT, VO_PROJB, h));
        A = ldw<XA>(w.fc1, 0, 1, lane);                                            // MLP chunk 0: fc1 rows 0..31
        MDR_PIN();
        const f32x16 y1 = lin2_T(B, ox, chanvec_lds(VT, VO_PROJB + 32, h));
        B = ldw<XA>(w.fc2, 0, 8, lane);                                            //              fc2 columns 0..31, both row blocks
        MDR_PIN();
        if constexpr (H) { vf[0] = fma16(y0, inv, vf[0]); vf[1] = fma16(y1, inv, vf[1]); }
        else { vf[0] += y0; vf[1] += y1; }
    }
    MDR_STAMP(2)
    {
        f32x16 acc2[2][2];
        Act y2[2];
        {
            f32x16 y2f[2];
            layernorm64_L(vf, VT + VO_N2W, VT + VO_N2B, h, y2f);
            if constexpr (XA == 3) { y2[0] = x1_cvt(y2f[0]); y2[1] = x1_cvt(y2f[1]); }
            else if constexpr (XA == 2) { y2[0] = x2_split(y2f[0]); y2[1] = x2_split(y2f[1]); }
            else { y2[0] = mk<XA>(y2f[0]); y2[1] = mk<XA>(y2f[1]); }
        }
        if constexpr (X) {      // the residual stream waits in LDS while the MLP needs the registers
            park_vf(vf);
        }
        acc2[0][0] = chanvec_lds(VT, VO_FC2B, h);
        acc2[1][0] = chanvec_lds(VT, VO_FC2B + 32, h);
        acc2[0][1] = zero16();
        acc2[1][1] = zero16();
        // XA 3 is a latency-bound kernel (waves parked or issue-stalled 64 % of their time, profiles/r05_pmc_config3_B256.txt): its loop is
        // software-pipelined by one chunk inside the wave -- fc1 of chunk c + 1 is issued BEFORE the bias + GELU + conversion of chunk c, so
        // the eight dependent MFMAs run under that vector work instead of in front of it, and fc2's two chains are interleaved.
        if constexpr (XA == 3 && MDR_X1_PIPE) {
            f32x16 hn = lin2_T(A, y2, zero16());                                        // fc1, chunk 0
            A = ldw<XA>(w.fc1, 2, 3, lane);
            MDR_PIN();
#pragma unroll 1
            for (int c = 0; c < 8; ++c) {
                f32x16 hdn = hn;
                if (c < 7) hn = lin2_T(A, y2, zero16());                               // fc1, chunk c + 1: independent of everything below
                hdn += chanvec_lds(VT, VO_FC1B + 32 * c, h);
                if constexpr (kX1GeluTable) gelu_tile_table(hdn, inv, GT); else gelu_tile_scaled(hdn, inv);
                const X1 hx = mk<XA>(hdn, inv);
                __builtin_amdgcn_sched_barrier(0);
                if (c < 6) A = ldw<XA>(w.fc1, 2 * (c + 2), 2 * (c + 2) + 1, lane); else if (c == 6) A = ldw<XA>(w.sa0, 0, 1, lane);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 2; ++s) {                                          // fc2: two independent chains, small planes first
                    acc2[0][0] = GATOR_MFMA_F16(B.t[0].mid[s], hx.p[s], acc2[0][0]);
                    acc2[1][0] = GATOR_MFMA_F16(B.t[1].mid[s], hx.p[s], acc2[1][0]);
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    acc2[0][0] = GATOR_MFMA_F16(B.t[0].hi[s], hx.p[s], acc2[0][0]);
                    acc2[1][0] = GATOR_MFMA_F16(B.t[1].hi[s], hx.p[s], acc2[1][0]);
                }
                if (c < 7) B = ldw<XA>(w.fc2, c + 1, 8 + c + 1, lane); else B = ldw<XA>(w.sa0, 2, 3, lane);
                MDR_PIN();
            }
        } else
#pragma unroll 1
        for (int c = 0; c < 8; ++c) {           // 256 hidden units in 8 chunks of 32: fc1 -> GELU -> fc2 partial
            f32x16 hdn;
            if constexpr (X) {      // bias after the products: its scalar loads fly during the MFMAs instead of in front of them
                hdn = lin2_T(A, y2, zero16());
                if (c < 7) A = ldw<XA>(w.fc1, 2 * (c + 1), 2 * (c + 1) + 1, lane); else A = ldw<XA>(w.sa0, 0, 1, lane);
                MDR_PIN();
                hdn += chanvec_lds(VT, VO_FC1B + 32 * c, h);
            } else {
                hdn = lin2_T(A, y2, chanvec_lds(VT, VO_FC1B + 32 * c, h));
                if (c < 7) A = ldw<XA>(w.fc1, 2 * (c + 1), 2 * (c + 1) + 1, lane); else A = ldw<XA>(w.sa0, 0, 1, lane);
                MDR_PIN();
            }
            if constexpr (XA == 3 && kX1GeluTable) gelu_tile_table(hdn, inv, GT); else if constexpr (H) gelu_tile_scaled(hdn, inv); else gelu_tile(hdn);
            if constexpr (XA == 3) {
                const X1 hx = mk<XA>(hdn, inv);
                acc2[0][0] = g2_mma_wa(B.t[0], hx, acc2[0][0]);
                acc2[1][0] = g2_mma_wa(B.t[1], hx, acc2[1][0]);
            } else if constexpr (H) {
                const X2 hx = mk<XA>(hdn, inv);
                acc2[0][0] = h3_mma_wa(B.t[0], hx, acc2[0][0]);
                acc2[1][0] = h3_mma_wa(B.t[1], hx, acc2[1][0]);
            } else if constexpr (X) {
                const X3 hx = x3_split(hdn);
                acc2[0][0] = x3_mma(B.t[0], hx, acc2[0][0]);
                acc2[1][0] = x3_mma(B.t[1], hx, acc2[1][0]);
            } else {
                mma2_T(B.t[0], hdn, acc2[0][c & 1], B.t[1], hdn, acc2[1][c & 1]);   // even / odd chunks: 2 chains of 128 products each
            }
            if (c < 7) B = ldw<XA>(w.fc2, c + 1, 8 + c + 1, lane); else B = ldw<XA>(w.sa0, 2, 3, lane);
            MDR_PIN();
        }
        if constexpr (H) {
            unpark_vf(vf);
            vf[0] = fma16(acc2[0][0], inv, vf[0]);
            vf[1] = fma16(acc2[1][0], inv, vf[1]);
        } else if constexpr (X) {
            unpark_vf(vf);
            vf[0] += acc2[0][0];
            vf[1] += acc2[1][0];
        } else {
            vf[0] += acc2[0][0] + acc2[0][1];
            vf[1] += acc2[1][0] + acc2[1][1];
        }
    }
    MDR_STAMP(3)
    custom_ln64_L(vf, VT + VO_A2, VT + VO_B2, h);                                           // MDR.py:142 self.norm
    store_block(a.vf_out + (tile + 0) * kTile, lane, vf[0]);
    store_block(a.vf_out + (tile + 1) * kTile, lane, vf[1]);
    // ---- in-projections of the self-attention (vanilla_transformer_encoder.py:87-89) in the consumer's operand order ----
    {
        const Act vfx[2] = {mk<XA>(vf[0]), mk<XA>(vf[1])};
        f32x16 y0 = lin2_T(A, vfx, chanvec_lds(VT, VO_SA0B, h));
        A = ldw<XA>(w.sa1, 0, 1, lane);
        MDR_PIN();
        f32x16 y1 = lin2_T(B, vfx, chanvec_lds(VT, VO_SA0B + 32, h));
        B = ldw<XA>(w.sa1, 2, 3, lane);
        MDR_PIN();
        if constexpr (X) {      // the consumer's softmax works in the exp2 domain: fold log2(e) / sqrt(d_k) into Q once, here
            const float qs = kLog2e * 0.17677669529663688110f * (XA == 2 ? kX2QK : 1.0f) * inv;      // (XA 3: Q . K is the exp2-domain score itself)
            y0 = y0 * qs;
            y1 = y1 * qs;
        }
        st_op<XA>(a.q_out + (tile + 0) * TQ, lane, y0);
        st_op<XA>(a.q_out + (tile + 1) * TQ, lane, y1);
        y0 = lin2_T(A, vfx, chanvec_lds(VT, VO_SA1B, h));
        A = ldw<XA>(w.sa2, 0, 1, lane);
        MDR_PIN();
        y1 = lin2_T(B, vfx, chanvec_lds(VT, VO_SA1B + 32, h));
        B = ldw<XA>(w.sa2, 2, 3, lane);
        const float bv0 = w.sa2_b[lane & 31], bv1 = w.sa2_b[32 + (lane & 31)];
        MDR_PIN();
        if (token >= kV) { y0 = zero16(); y1 = zero16(); }                   // pad keys: finite (they are masked anyway)
        if constexpr (XA == 2) { y0 = y0 * (kX2QK * inv); y1 = y1 * (kX2QK * inv); }
        if constexpr (XA == 3) { y0 = y0 * inv; y1 = y1 * inv; }
        st_op<XA>(a.k_out + (tile + 0) * TQ, lane, y0);
        st_op<XA>(a.k_out + (tile + 1) * TQ, lane, y1);
        y0 = lin2_C(A, vfx);                                                  // V in C-layout: channel on the lane
        y1 = lin2_C(B, vfx);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool ok = 32 * t + kap(r) + 4 * h < kV;
            y0[r] = ok ? (H ? __builtin_fmaf(y0[r], inv, bv0) : y0[r] + bv0) : 0.f;
            y1[r] = ok ? (H ? __builtin_fmaf(y1[r], inv, bv1) : y1[r] + bv1) : 0.f;
        }
        if constexpr (XA >= 2) { y0 = y0 * kX2V; y1 = y1 * kX2V; }
        st_op<XA>(a.v_out + (tile + 0) * TQ, lane, y0);
        st_op<XA>(a.v_out + (tile + 1) * TQ, lane, y1);
    }
    MDR_STAMP(4)
#ifdef GATOR_DIAG
    if (a.stamps && id == 0 && lane == 0)
        for (int i = 0; i < 8; ++i) a.stamps[i] = st_acc[i];
    if (a.stamps && (id % 64) == 0 && lane == 0) {      // timeline sample: [start, end] in 100 MHz ticks, cycles, XCC id
        unsigned long long* q = a.stamps + 8 + 4 * (id / 64);
        q[0] = rt0;
        q[1] = wall_clock64();
        q[2] = clock64() - ck0;
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        q[3] = ((unsigned long long)(xcc & 0xf) << 32) | hwid;
    }
#endif
}

template <int MODE, int XA>
__global__ __launch_bounds__(256, 2) void k_mdr_layer(const MdrArgs a, int nwg) {
    __shared__ f32x4 park[XA != 0 ? kParkF4 : 1];
    __shared__ __attribute__((aligned(16))) float VT[VO_TOTAL];
    __shared__ __attribute__((aligned(16))) float GT[XA == 3 && MODE < 2 && kX1GeluTable ? 2 * kGeluTab : 2];
    if constexpr (XA == 3 && MODE < 2 && kX1GeluTable) gelu_table_fill(GT);
    mdr_stage_vectors<MODE, XA>(a, VT);
    __syncthreads();
    mdr_tile<MODE, XA>(a, xcd_remap(blockIdx.x, nwg) * 4 + (threadIdx.x >> 6), VT, park, GT,
                       (const MdrArgs*)(const __attribute__((address_space(4))) MdrArgs*)__builtin_amdgcn_kernarg_segment_ptr());      // `a` is the first kernel argument
}

// ---- all four stages in ONE persistent launch -------------------------------------------------------------------------------
// Why: a tile costs one SIMD ~85k cycles whether or not a second wave shares the SIMD, so a launch takes
// ceil(tiles / 1024 SIMDs) tile times -- at B = 256 that is 3 584 tiles = 3.5 per SIMD, billed as 4, in each of the four launches
// (measured: B = 219 / 256 / 292 -> 127 / 160 / 164 us for the middle launch).  Over all four stages there are exactly 14 tiles per
// SIMD, so one launch that hands out (stage, tile) units from a queue and lets a unit wait only for ITS OWN sample's previous
// stage -- the path's only cross-tile dependency is the self-attention's K/V of the sample -- has no fractional generation left.
//   * one queue per XCD: sample b lives on XCD b % 8 for all four stages, so the Q/K/V/residual tile sets of a sample are written
//     and read through the same L2 (workgroups are dealt round-robin to the XCDs; if they were not, only locality would suffer);
//   * a workgroup takes a ticket (4 consecutive tiles of its XCD's list, stage-major), stages the stage's channel vectors, each
//     wave waits until `done[stage - 1][sample]` says all 14 tiles of its sample are finished (agent-scope acquire), runs the
//     unchanged tile body and bumps `done[stage][sample]` behind an agent-scope release;
//   * tickets are handed out in dependency order and a workgroup holds a ticket only while it runs, so every wait is for a unit
//     that some running workgroup already owns: no deadlock whatever the residency.  A poll budget (~6 s) turns a would-be hang (a bug)
//     into a flag in ctr[kCtrError] and garbage output instead of a dead GPU.
// What if an XCD gets no workgroup (a CU mask, reserved CUs, a partitioned device)?  Its queue is never served and nobody waits for it.
// That must not be silent: EVERY stage counts its finished tiles per sample (the last one too), k_mdr_head refuses a sample whose 14
// head-feature tiles were not all written -- NaN vertices plus the ctx's sticky status word -- and the host side answers the report
// (GATOR_EDEVICE at the next call, api.hip) by switching that ctx to the four-launch form for good.  (Serving foreign queues instead
// was built and measured in round 4: a workgroup claimed a queue by compare-and-swap, its own first, then any unowned one, so that an
// orphan queue was adopted as a whole by ONE other XCD.  Correct in every placement tried -- grids of 1, 3, 5, 13 workgroups -- but with
// the queue loop around the three ticket loops hipcc lays the tile bodies out differently and the launch takes 423 us instead of 385
// even when the loop runs once; a second, cold copy of the body spills 2 KB per lane.  Not worth 10 % of the dominant kernel.)
// Counter block of ONE persistent launch: [0..7] tickets per XCD, [8] error flag, [kCtrDone + stage * B + b] finished tiles of (stage,
// sample), stage 0..3, with B and b the launch's own batch (a large forward runs as several launches over chunks of its samples,
// launch_mdr: their blocks follow each other in FusedWs::mdr_ctr, see mdr_ctr_block).  The kernel's argument block stays exactly
// {stage arguments, ctr}: the tile body needs every scalar register there is, and each further scalar that must survive it
// (measured with a base pointer, an offset and a stride more) is spilled into VGPR lanes and reloaded in its loops: +30 - 40 us.
constexpr int kCtrError = 8, kCtrDone = 32;
// chunk plan of a forward of B samples in nch launches: the first B % nch chunks have one sample more.  -> (chunk, first sample, size)
// of sample b, and the word offset of a chunk's counter block
struct MdrChunkPlan {
    int nch, base, rem;
    __host__ __device__ void locate(int b, int& ch, int& b0, int& n) const {
        const int split = rem * (base + 1);
        if (b < split) { ch = b / (base + 1); n = base + 1; b0 = ch * n; }
        else { ch = rem + (b - split) / base; n = base; b0 = split + (ch - rem) * base; }
    }
    __host__ __device__ size_t block(int ch) const {
        const int big = ch < rem ? ch : rem;
        return (size_t)ch * kCtrDone + 4 * ((size_t)big * (base + 1) + (size_t)(ch - big) * base);
    }
};

struct HeadArgs {
    const float *hf, *bn_w, *bn_b, *bn_mean, *bn_var, *bconv_w, *bconv_b;
    float *vc, *vcp;
    __bf16* vcp3;           // non-null: write the hi/mid/lo bf16 planes of the split-precision vertex GEMM instead of vcp
    size_t vcp3_plane;
    _Float16* vcp2;         // non-null: write the scaled hi/lo fp16 planes of the two-plane vertex GEMM (upsample_x2.hip) instead
    const unsigned* persist_ctr;   // non-null: the counter blocks of the forward's persistent launches.  The sample's launch must not have tripped its
    MdrChunkPlan plan;             // hang guard and must have counted all 14 last-stage tiles of the sample; else its vertices are NaN (loud, not silent)
    unsigned* status;              // the ctx's sticky device status word (host-mapped; internal.h: DeviceStatus), read by the next API call
    int alpha;
};

// where a coarse vertex coordinate goes: the reference layout (tap / stage API) and the packed A operand of whichever vertex GEMM the ctx runs
__device__ __forceinline__ void head_store(const HeadArgs& a, int b, int v, int c, float val) {
    const int mt = b >> 5, sl = b & 31;
    a.vc[((size_t)b * kV + v) * 3 + c] = val;
    if (a.vcp2) {       // two fp16 planes of 2^4 * val, in k_upsample_x2's operand order [mt/4][v/16][mt%4][l'][plane][lane][v%8]
        const float sv = val * 16.0f;
        const _Float16 hi = (_Float16)sv;
        const _Float16 lo = (_Float16)(sv - (float)hi);
        const size_t pair = ((((size_t)(mt >> 2) * 28 + (v >> 4)) * 4 + (mt & 3)) * 3 + c) * 2;
        const size_t e = (size_t)(((v >> 3) & 1) * 32 + sl) * 8 + (v & 7);
        a.vcp2[pair * 512 + e] = hi; a.vcp2[(pair + 1) * 512 + e] = lo;
    } else if (a.vcp3) {       // exact three-way bf16 split, in k_upsample_x3's operand order [plane][mt][l'][v/16][lane][v%8]
        const __bf16 hi = (__bf16)val;
        const float r1 = val - (float)hi;
        const __bf16 mid = (__bf16)r1;
        const __bf16 lo = (__bf16)(r1 - (float)mid);
        const size_t e = ((((size_t)mt * 3 + c) * 28 + (v >> 4)) * 64 + ((v >> 3) & 1) * 32 + sl) * 8 + (v & 7);
        a.vcp3[e] = hi; a.vcp3[a.vcp3_plane + e] = mid; a.vcp3[2 * a.vcp3_plane + e] = lo;
    } else {
        const int cb = v >> 5, g = (v & 31) >> 3, hh = (v & 7) >> 2, j = v & 3;
        a.vcp[(((((size_t)mt * 3 + c) * kCB + cb) * 4 + g) * 64 + hh * 32 + sl) * 4 + j] = val;
    }
}

// ---- the head behind the tiles' conv partials (round 6): bias_conv1d's result = bias + the 14 partials in tile order; then per token the softmax-mix
// of MDR.py:161-166.  One sample; `bc`: 60 floats of LDS; called by every thread of a workgroup of NT threads (k_mdr_head_finish).
// Light by construction: no conv, no cross-lane sums.
template <int NT>
__device__ __forceinline__ void head_finish(const HeadArgs& a, const double* __restrict__ hpart_b, int b, bool poisoned, float (*bc)[3]) {
    const int t = threadIdx.x, lane = t & 63;
    const float* hf = a.hf + (size_t)b * kV * 32;
    constexpr int NR = (kV + NT - 1) / NT;
    // every global read up front: this thread's tokens' head features, and (threads 0..59) the partials of output (row, position) t
    f32x4 row[NR][5], tail[NR], cc[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int v = t + NT * i;
        const float* r = hf + (v < kV ? v : 0) * 32;
#pragma unroll
        for (int g = 0; g < 5; ++g) row[i][g] = *reinterpret_cast<const f32x4*>(r + 4 * g);
        tail[i] = *reinterpret_cast<const f32x4*>(r + 24);
        cc[i] = *reinterpret_cast<const f32x4*>(r + 28);
    }
    if (t < 60) {
        double pv[kVT];
#pragma unroll
        for (int k = 0; k < kVT; ++k) pv[k] = hpart_b[k * 64 + t];
        double s = pv[0];
#pragma unroll
        for (int k = 1; k < kVT; ++k) s += pv[k];
        bc[t / 3][t % 3] = (float)(s + (double)a.bconv_b[t / 3]);
    }
    __syncthreads();
    const float limit = a.vcp2 ? 4094.0f : 3.0e38f;      // (two-plane vertex regressor: 16 x value in an fp16 plane)
    bool bad = false;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int v = t + NT * i;
        if (v < kV) {
            float av[20];
#pragma unroll
            for (int g = 0; g < 5; ++g) { av[4 * g] = row[i][g][0]; av[4 * g + 1] = row[i][g][1]; av[4 * g + 2] = row[i][g][2]; av[4 * g + 3] = row[i][g][3]; }
            float mx = -1e30f, p[20], l = 0.f;
            for (int m = 0; m < 20; ++m) mx = fmaxf(mx, av[m]);
            for (int m = 0; m < 20; ++m) {
                p[m] = __builtin_amdgcn_exp2f((av[m] - mx) * kLog2e);
                l += p[m];
            }
            const float il = 1.0f / l;
            // alpha = 1.1 ** scale_linear(x)  (MDR.py:162): powf via double exp keeps it exact to fp32 rounding; once per token
            const float sc = a.alpha ? (float)exp((double)tail[i][3] * 0.09531017980432493) : 1.0f;
            for (int c = 0; c < 3; ++c) {
                float o = 0.f;
                for (int m = 0; m < 20; ++m) o += (p[m] * il) * bc[m][c];
                float val = sc * o + cc[i][c];
                if (poisoned) val = __builtin_nanf("");
                bad = bad || !(fabsf(val) < limit);
                head_store(a, b, v, c, val);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (a.status && __any(bad) && lane == 0)      // sticky, host-visible: the next API call on the ctx (or gator_device_status) reports it
        __hip_atomic_store(a.status, poisoned ? (unsigned)DEV_PERSIST_INCOMPLETE : (unsigned)DEV_NONFINITE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct MdrPersistArgs {
    MdrArgs st[4];            // their B and every per-sample pointer are this launch's CHUNK of the batch
    unsigned* ctr;            // this launch's counter block
};
// (Round 5, measured and dropped: the one-plane form XA = 3 built for 168 registers and launched with THREE workgroups per CU -- a SIMD
// issues the vector instructions of three waves faster than of two, tools/microbench/valu_rate.hip -- spills 164 B per lane even with
// one weight pair alive at a time, and runs in the same time: 3.25 ms per forward of 2 048 samples either way.)
template <int XA>
__global__ __launch_bounds__(256, 2) void k_mdr_persist(const MdrPersistArgs p) {
    __shared__ f32x4 park[XA != 0 ? kParkF4 : 1];
    __shared__ __attribute__((aligned(16))) float VT[VO_TOTAL];
    __shared__ __attribute__((aligned(16))) float GT[XA == 3 && kX1GeluTable ? 2 * kGeluTab : 2];
    __shared__ int s_unit;
    if constexpr (XA == 3 && kX1GeluTable) gelu_table_fill(GT);      // published by the first ticket's barriers
    const int B = p.st[0].B, lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned xcc;                                                // the XCD this workgroup REALLY runs on picks its queue
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const int xcd = (int)(xcc & 7u);
    const int nb = xcd < B ? (B - xcd + 7) >> 3 : 0;            // samples of this XCD: xcd, xcd + 8, ...
    const int ntile = nb * kVT, units = (ntile + 3) >> 2;       // per stage
    if (units == 0) return;
    auto ticket = [&]() {
        __syncthreads();                                        // everyone is done with s_unit (and, at a stage change, with VT)
        if (threadIdx.x == 0) s_unit = (int)atomicAdd(p.ctr + xcd, 1u);
        __syncthreads();
        return __builtin_amdgcn_readfirstlane(s_unit);          // a scalar: stage, tile and sample ids stay out of the VGPRs
    };
    int staged = -1;                                            // the stage whose channel vectors are in VT
    // one ticket: (stage the channel vectors,) wait for the sample's previous stage, run the tile, publish it
    auto run = [&](auto mode, int stage, int unit) {
        constexpr int MODE = decltype(mode)::value;
        // the stage's arguments (p.st[stage]) through a pointer into the kernel-argument segment that the optimiser cannot see
        // through: otherwise every argument load of the body is loop-invariant, hoisted in front of the ticket loop and held in
        // SGPRs across it (106 SGPRs, spills into VGPRs, scratch)
        typedef const __attribute__((address_space(4))) MdrArgs* KArgPtr;
        typedef const __attribute__((address_space(4))) char* KBytePtr;
        unsigned aoff = (unsigned)offsetof(MdrPersistArgs, st) + (unsigned)__builtin_amdgcn_readfirstlane(stage) * (unsigned)sizeof(MdrArgs);
        asm volatile("" : "+s"(aoff));
        KArgPtr ap = (KArgPtr)((KBytePtr)__builtin_amdgcn_kernarg_segment_ptr() + aoff);
        const MdrArgs& a = *(const MdrArgs*)ap;
        const int lt = 4 * (unit - stage * units) + wave;
        const bool live = lt < ntile;                           // (wave-uniform) a ticket's last waves may have nothing left
        const int smp = xcd + 8 * (lt / kVT), id = smp * kVT + lt % kVT;
        // the completion count of the sample's previous stage is requested FIRST: its L2 round trip (1 - 2 us under load, once per
        // tile) hides behind the staging below instead of standing in front of the tile
        const unsigned* d = p.ctr + kCtrDone + (size_t)(MODE > 0 ? stage - 1 : 0) * B + smp;
        unsigned seen = kVT;
        if (MODE > 0 && live && lane == 0) seen = __hip_atomic_load(d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (staged != stage) {                                  // tickets come in stage order: at most four times per workgroup
            mdr_stage_vectors<MODE, XA>(a, VT);
            staged = stage;
            __syncthreads();
        }
        if (!live) return;
        if (MODE > 0) {
            if (lane == 0) {
                int budget = 1 << 24;                           // ~6 s of polling
                while (seen < (unsigned)kVT && --budget > 0) {
                    __builtin_amdgcn_s_sleep(8);
                    seen = __hip_atomic_load(d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (budget <= 0) atomicExch(p.ctr + kCtrError, 1u + stage);
            }
            // Acquire among CUs that share an L2, with NO cache invalidate: every tile of the three tile sets is written exactly once
            // per launch (launch_mdr) and read only behind its completion count, and the L1 starts a launch empty, so neither
            // the L1 nor the L2 can hold an older copy of what is read from here on.  (The agent-scope fence pair instead --
            // `buffer_wbl2 sc1` / `buffer_inv sc1` at each of ~10k tile starts -- measured +240 us per forward; `buffer_inv sc1`
            // alone +30 us.)  The compiler barrier keeps the tile's loads behind the poll.
            asm volatile("" ::: "memory");
        }
        mdr_tile<MODE, XA>(a, id, VT, park, GT, &a);
        // Release to the same L2: the L1 is write-through, so once the stores are acknowledged (vmcnt 0) every CU of the XCD
        // sees them; then the count goes up (an atomic executed in that L2).  The last stage counts too (for k_mdr_head, see above).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(p.ctr + kCtrDone + (size_t)stage * B + smp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // Tickets come in stage order, so a workgroup's stages only ever go up: three plain loops, one tile body each (one loop with a
    // switch keeps all three bodies' state alive at once: 256 VGPRs + 760 B of scratch).
#ifdef GATOR_DIAG
    const unsigned long long pe_t0 = wall_clock64();
    unsigned long long pe_last = pe_t0;
#define PERSIST_END_MARK() pe_last = wall_clock64()
#else
#define PERSIST_END_MARK()
#endif
    int unit = ticket();
    for (; unit < units; unit = ticket()) run(std::integral_constant<int, 0>(), 0, unit);
    for (; unit < 3 * units; unit = ticket()) {
        const int stage = unit >= 2 * units ? 2 : 1;
        run(std::integral_constant<int, 1>(), stage, unit);
    }
    for (; unit < 4 * units; unit = ticket()) { PERSIST_END_MARK(); run(std::integral_constant<int, 2>(), 3, unit); }
#ifdef GATOR_DIAG
    if (g_persist_ends && threadIdx.x == 0) {
        g_persist_ends[3 * blockIdx.x] = pe_t0;
        g_persist_ends[3 * blockIdx.x + 1] = pe_last;
        g_persist_ends[3 * blockIdx.x + 2] = wall_clock64();
    }
#endif
}

// Joint tokens: jf = Linear(133->64)(pose_combine) + pos_j (MDR.py:130-134); per layer k = wk(LN1(jf)), v = wv(LN1(jf))
// (MDR.py:37-38 with norm1 applied to the concatenated tokens, :65).  jf does not change across the three layers.
// One workgroup (2 waves) per sample; wave w owns channel block w (= head w).  Output in MFMA operand order:
//   K tile [hd][g][lane=(joint,h)][j] = k[joint][32hd+8g+4h+j]  (T-layout block hd)
//   V tile [hd][g][lane=(d,h)][j]     = v[joint=8g+4h+j][32hd+d] (C-layout block hd)
struct JointArgs {
    const float *pc, *jw_p, *jb, *posj_T;       // jw_p: packed [2 nb][5 kb]; posj_T: [2] T-layout tiles of pos_j[1..J]
    const float *n1w[3], *n1b[3], *wk_p[3], *wv_p[3];
    float* jkv;
    int J;
    unsigned* mdr_ctr;      // non-null: zero k_mdr_persist's tickets and completion counts (B = gridDim.x)
    int x2;                 // K/V tiles as two fp16 planes of 16 x value (cross_attention_head_x2) instead of fp32 blocks
};
__global__ __launch_bounds__(128) void k_mdr_joint(const JointArgs a) {
    __shared__ __attribute__((aligned(16))) float PCt[5 * kTile];
    __shared__ __attribute__((aligned(16))) float JF[2 * kTile];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, h = lane >> 5, J = a.J;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    if (a.mdr_ctr) {
        // every launch's counter block (tickets, error flag, completion counts): the whole region, dealt over the workgroups
        for (size_t i = (size_t)b * 128 + t; i < mdr_ctr_words((int)gridDim.x); i += (size_t)gridDim.x * 128) a.mdr_ctr[i] = 0u;
    }
    for (int e = t; e < 5 * kTile; e += 128) {
        const int j4 = e & 3, ln = (e >> 2) & 63, g = (e >> 8) & 3, kb = e >> 10;
        const int tok = ln & 31, k = 32 * kb + 8 * g + 4 * (ln >> 5) + j4;
        PCt[e] = (tok < J && k < 133) ? a.pc[((size_t)b * J + tok) * 133 + k] : 0.f;
    }
    __syncthreads();
    {
        f32x16 a0 = load_chanvec_S(a.jb, 32 * wave, h) + load_block(a.posj_T + wave * kTile, lane), a1 = zero16();
#pragma unroll
        for (int kb = 0; kb < 5; ++kb) {
            if (kb & 1) a1 = mma_T(load_wtile(a.jw_p, wave * 5 + kb, lane), load_block(PCt + kb * kTile, lane), a1);
            else a0 = mma_T(load_wtile(a.jw_p, wave * 5 + kb, lane), load_block(PCt + kb * kTile, lane), a0);
        }
        store_block(JF + wave * kTile, lane, a0 + a1);
    }
    __syncthreads();
    f32x16 jf[2];
    jf[0] = load_block(JF, lane);
    jf[1] = load_block(JF + kTile, lane);
    const bool tok_ok = (lane & 31) < J;
#pragma unroll 1
    for (int li = 0; li < 3; ++li) {
        f32x16 fz[2];
        layernorm64(jf, a.n1w[li], a.n1b[li], h, fz);
        float* out = a.jkv + (((size_t)b * 3 + li) * 4) * kTile;
        f32x16 kt, k1 = zero16();
        kt = zero16();
        mma2_T(load_wtile(a.wk_p[li], wave * 2 + 0, lane), fz[0], kt, load_wtile(a.wk_p[li], wave * 2 + 1, lane), fz[1], k1);
        kt += k1;
        if (!tok_ok) kt = zero16();             // joints >= J: zero rows (masked in the softmax anyway)
        if (a.x2) x2_store(out + wave * kTile, lane, x2_split(kt * 16.0f)); else store_block(out + wave * kTile, lane, kt);
        f32x16 vt = zero16(), v1 = zero16();
        mma2_C(load_wtile(a.wv_p[li], wave * 2 + 0, lane), fz[0], vt, load_wtile(a.wv_p[li], wave * 2 + 1, lane), fz[1], v1);
        vt += v1;
#pragma unroll
        for (int r = 0; r < 16; ++r) vt[r] = (kap(r) + 4 * h < J) ? vt[r] : 0.f;
        if (a.x2) x2_store(out + (2 + wave) * kTile, lane, x2_split(vt * 16.0f)); else store_block(out + (2 + wave) * kTile, lane, vt);
    }
}

// sum of a double over the 64 lanes on DPP row operations (quad swaps, half-row and row mirrors: every lane ends with its row's total) and four
// v_readlane per half -- 12 cross-lane moves in registers instead of the 12 ds_bpermute round trips of a __shfl_xor butterfly; fixed association
__device__ __forceinline__ double dpp_mov_f64(double v, int which) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    int lo = (int)(unsigned)u, hi = (int)(unsigned)(u >> 32);
    if (which == 0) { lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xf, 0xf, false); }             // quad_perm [1,0,3,2]
    else if (which == 1) { lo = __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xf, 0xf, false); }        // quad_perm [2,3,0,1]
    else if (which == 2) { lo = __builtin_amdgcn_update_dpp(lo, lo, 0x141, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x141, 0xf, 0xf, false); }      // row_half_mirror
    else { lo = __builtin_amdgcn_update_dpp(lo, lo, 0x140, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x140, 0xf, 0xf, false); }                      // row_mirror
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double wave_sum64_f64(double s) {
    s += dpp_mov_f64(s, 0); s += dpp_mov_f64(s, 1); s += dpp_mov_f64(s, 2); s += dpp_mov_f64(s, 3);
    const unsigned long long u = __builtin_bit_cast(unsigned long long, s);
    const int lo = (int)(unsigned)u, hi = (int)(unsigned)(u >> 32);
    double t[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const unsigned l2 = (unsigned)__builtin_amdgcn_readlane(lo, 16 * r), h2 = (unsigned)__builtin_amdgcn_readlane(hi, 16 * r);
        t[r] = __builtin_bit_cast(double, ((unsigned long long)h2 << 32) | l2);
    }
    return (t[0] + t[1]) + (t[2] + t[3]);
}

// MDR head (MDR.py:156-166) from the per-token head features hf[b][v][32]:
//   ch 0..19 = mat_A, 24..26 = bias_linear out, 27 = scale_linear out, 28..30 = mat_C   (our own packing order)
// Writes vert431 both in the reference layout (tap / stage API) and as the packed A operand of the vertex GEMM.
// One workgroup per sample, three short phases with a barrier between them.  The kernel is a LATENCY chain, not a throughput one
// (18.7 us at B = 64, 24 us at B = 256, round-3 sweep): with HOIST every global read it will ever need -- this lane's 63 conv
// weights, its token's 32 head features -- is issued before the first phase, and the phases run on registers and LDS only
// (16 / 13 us).  That costs 196 VGPRs, one workgroup per CU: batches of more than two workgroups per CU take the rolled form
// (same arithmetic in the same order, 2 workgroups per CU), which is the faster one there.
#ifndef MDR_HEAD_CUT
#define MDR_HEAD_CUT 0      // timing experiments only (tools/build_variant.py ... -DMDR_HEAD_CUT=n): 1 no conv FMAs, 2 no wave reductions, 4 no softmax-mix, 8 no conv-weight loads
#endif
template <int NT, bool HOIST>
__global__ __launch_bounds__(NT, HOIST ? 2 : 4) void k_mdr_head(const HeadArgs a) {
    static_assert(NT >= kV, "one token per thread");
    __shared__ float bn[kV][5];      // [0 | x y z | 0]: the conv's zero padding of the xyz axis as stored zeros (unconditional reads in the loop)
    __shared__ float bc[20][3];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float* hf = a.hf + (size_t)b * kV * 32;
    const int v = t;
    const bool tok = v < kV;
    constexpr int NW = NT / 64, NIT = (kV * 3 + 63) / 64;
    // ---- all global reads up front
    f32x4 row[5], tail = {0.f, 0.f, 0.f, 0.f}, cc = tail;
    auto load_rows = [&]() {
        const float* r = hf + v * 32;
#pragma unroll
        for (int g = 0; g < 5; ++g) row[g] = *reinterpret_cast<const f32x4*>(r + 4 * g);
        cc = *reinterpret_cast<const f32x4*>(r + 28);
    };
    if (tok) {
        tail = *reinterpret_cast<const f32x4*>(hf + v * 32 + 24);
        if (HOIST) load_rows();
    }
    // Conv1d(431->20,k3,p1) weight of row m = wave + 8 q at e = lane + 64 it
    auto conv_w = [&](int q, int it) {
        const int e = lane + 64 * it, m = wave + NW * q;
        if (MDR_HEAD_CUT & 8) return 0.5f;
        return e < kV * 3 ? a.bconv_w[(m < 20 ? m : 0) * (kV * 3) + e] : 0.f;
    };
    float wreg[3][HOIST ? NIT : 1];
    if (HOIST) {
#pragma unroll
        for (int it = 0; it < NIT; ++it)
#pragma unroll
            for (int q = 0; q < 3; ++q) wreg[q][HOIST ? it : 0] = conv_w(q, it);
    }
    if (tok) {
        float x[3] = {tail[0], tail[1], tail[2]};
        if (a.alpha) {      // LayerNorm(3)
            const float m = (x[0] + x[1] + x[2]) / 3.0f;
            const float qq = ((x[0] - m) * (x[0] - m) + (x[1] - m) * (x[1] - m) + (x[2] - m) * (x[2] - m)) / 3.0f;
            const float rs = 1.0f / sqrtf(qq + 1e-5f);
            for (int c = 0; c < 3; ++c) x[c] = (x[c] - m) * rs * a.bn_w[c] + a.bn_b[c];
        } else {            // BatchNorm1d(431) eval: channel = vertex
            const float rs = 1.0f / sqrtf(a.bn_var[v] + 1e-5f);
            for (int c = 0; c < 3; ++c) x[c] = (x[c] - a.bn_mean[v]) * rs * a.bn_w[v] + a.bn_b[v];
        }
        bn[v][0] = 0.f; bn[v][4] = 0.f;
        for (int c = 0; c < 3; ++c) bn[v][1 + c] = gelu_f(x[c]);
    }
    __syncthreads();
    {   // Conv1d(431->20,k3,p1) over the xyz axis.  Wave w owns output rows m = w, w+8, w+16 and walks the whole (c,k) axis:
        // 9 accumulators and 9 wave reductions per wave.
        float acc[3][3];
#pragma unroll
        for (int q = 0; q < 3; ++q) acc[q][0] = acc[q][1] = acc[q][2] = 0.f;
        // Round 6 (the conv loop was 6.5 of the launch's 17 us, the nine double-precision butterflies through LDS 2.6: MDR_HEAD_CUT): the walk
        // e = lane + 64 it advances (channel c, tap k) by (21, +1) instead of dividing; the padding is stored zeros, not conditions; the
        // products are fused multiply-adds; the wave sums run on DPP row operations (still in double: bc feeds every coarse vertex).
        int cch = lane / 3, ktap = lane - 3 * cch;
#pragma unroll(HOIST ? NIT : 1)
        for (int it = 0; it < NIT; ++it) {
            const int e = lane + 64 * it;
            if (!(MDR_HEAD_CUT & 1) && e < kV * 3) {
                // tap k of channel c meets input position l + k - 1 (zero padding outside 0..2 = the stored zeros)
                const float in0 = bn[cch][ktap], in1 = bn[cch][ktap + 1], in2 = bn[cch][ktap + 2];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const float w = HOIST ? wreg[q][HOIST ? it : 0] : conv_w(q, it);
                    acc[q][0] = fmaf(w, in0, acc[q][0]);
                    acc[q][1] = fmaf(w, in1, acc[q][1]);
                    acc[q][2] = fmaf(w, in2, acc[q][2]);
                }
            }
            cch += ktap == 2 ? 22 : 21;          // e + 64 = 3 (c + 21) + (k + 1)
            ktap = ktap == 2 ? 0 : ktap + 1;
        }
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                const double s = (MDR_HEAD_CUT & 2) ? (double)acc[q][l] : wave_sum64_f64((double)acc[q][l]);
                const int m = wave + NW * q;
                if (lane == 0 && m < 20) bc[m][l] = (float)(s + (double)a.bconv_b[m]);
            }
    }
    __syncthreads();
    const int mt = b >> 5, sl = b & 31;
    bool poisoned = false;
    if (a.persist_ctr) {
        int ch, b0, n;
        a.plan.locate(b, ch, b0, n);
        const unsigned* blk = a.persist_ctr + a.plan.block(ch);
        poisoned = blk[kCtrError] != 0u || blk[kCtrDone + (size_t)3 * n + (b - b0)] != (unsigned)kVT;
    }
    // |vert431| must stay below 4 094 m for the two-plane vertex regressor (16 x value in an fp16 plane); any non-finite value -- e.g.
    // an activation beyond +-4 094 that overflowed an fp16 operand plane somewhere upstream -- ends up here as NaN too
    const float limit = a.vcp2 ? 4094.0f : 3.0e38f;
    bool bad = false;
    if (tok) {
        if (!HOIST) load_rows();
        float av[20];
#pragma unroll
        for (int g = 0; g < 5; ++g) { av[4 * g] = row[g][0]; av[4 * g + 1] = row[g][1]; av[4 * g + 2] = row[g][2]; av[4 * g + 3] = row[g][3]; }
        float mx = -1e30f, p[20], l = 0.f;
        for (int m = 0; m < 20; ++m) mx = fmaxf(mx, av[m]);
        for (int m = 0; m < 20; ++m) {
            p[m] = (MDR_HEAD_CUT & 4) ? av[m] : __builtin_amdgcn_exp2f((av[m] - mx) * kLog2e);
            l += p[m];
        }
        const float il = 1.0f / l;
        // alpha = 1.1 ** scale_linear(x)  (MDR.py:162): powf via double exp keeps it exact to fp32 rounding; once per token
        const float sc = a.alpha ? (float)exp((double)tail[3] * 0.09531017980432493) : 1.0f;
        const int cb = v >> 5, g = (v & 31) >> 3, hh = (v & 7) >> 2, j = v & 3;
        for (int c = 0; c < 3; ++c) {
            float o = 0.f;
            for (int m = 0; m < 20; ++m) o += (p[m] * il) * bc[m][c];
            float val = sc * o + cc[c];
            if (poisoned) val = __builtin_nanf("");
            bad = bad || !(fabsf(val) < limit);
            a.vc[((size_t)b * kV + v) * 3 + c] = val;
            if (a.vcp2) {       // two fp16 planes of 2^4 * val, in k_upsample_x2's operand order [mt/4][v/16][mt%4][l'][plane][lane][v%8]
                const float sv = val * 16.0f;
                const _Float16 hi = (_Float16)sv;
                const _Float16 lo = (_Float16)(sv - (float)hi);
                const size_t pair = ((((size_t)(mt >> 2) * 28 + (v >> 4)) * 4 + (mt & 3)) * 3 + c) * 2;
                const size_t e = (size_t)(((v >> 3) & 1) * 32 + sl) * 8 + (v & 7);
                a.vcp2[pair * 512 + e] = hi; a.vcp2[(pair + 1) * 512 + e] = lo;
            } else if (a.vcp3) {       // exact three-way bf16 split, in k_upsample_x3's operand order [plane][mt][l'][v/16][lane][v%8]
                const __bf16 hi = (__bf16)val;
                const float r1 = val - (float)hi;
                const __bf16 mid = (__bf16)r1;
                const __bf16 lo = (__bf16)(r1 - (float)mid);
                const size_t e = ((((size_t)mt * 3 + c) * 28 + (v >> 4)) * 64 + ((v >> 3) & 1) * 32 + sl) * 8 + (v & 7);
                a.vcp3[e] = hi; a.vcp3[a.vcp3_plane + e] = mid; a.vcp3[2 * a.vcp3_plane + e] = lo;
            } else {
                a.vcp[(((((size_t)mt * 3 + c) * kCB + cb) * 4 + g) * 64 + hh * 32 + sl) * 4 + j] = val;
            }
        }
    }
    if (a.status && __any(bad) && lane == 0)      // sticky, host-visible: the next API call on the ctx (or gator_device_status) reports it
        __hip_atomic_store(a.status, poisoned ? (unsigned)DEV_PERSIST_INCOMPLETE : (unsigned)DEV_NONFINITE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(256) void k_mdr_head_finish(const HeadArgs a, const double* __restrict__ hpart) {
    __shared__ float bc[20][3];
    const int b = blockIdx.x;
    bool poisoned = false;
    if (a.persist_ctr) {      // the sample's launch must not have tripped its hang guard and must have counted all 14 last-stage tiles
        int ch, b0, n;
        a.plan.locate(b, ch, b0, n);
        const unsigned* blk = a.persist_ctr + a.plan.block(ch);
        poisoned = blk[kCtrError] != 0u || blk[kCtrDone + (size_t)3 * n + (b - b0)] != (unsigned)kVT;
    }
    head_finish<256>(a, hpart + (size_t)b * kVT * 64, b, poisoned, bc);
}

LayerW make_layer(const FusedState* f, const gator_ctx* c, int li) {
    const MdrLayerP& p = f->lay[li];
    const MdrLayerW& r = c->w.lay[li];
    LayerW w;
    // X3 images mirror the fp32 tile grids tile for tile (fused_create)
    auto sel = [&](const float* t) { return f->mdr_x3 ? f->wxbuf + (size_t)(t - f->lay[0].wq) / kTile * kTileX3 : t; };
    w.wq = sel(p.wq); w.proj = sel(p.proj); w.fc1 = sel(p.fc1); w.fc2 = sel(p.fc2);
    w.sa0 = sel(p.sa[0]); w.sa1 = sel(p.sa[1]); w.sa2 = sel(p.sa[2]); w.sa3 = sel(p.sa[3]);
    w.n1w = r.n1w; w.n1b = r.n1b; w.proj_b = r.proj_b; w.n2w = r.n2w; w.n2b = r.n2b; w.fc1_b = r.fc1_b; w.fc2_b = r.fc2_b;
    w.a2 = r.a2; w.b2 = r.b2; w.sa0_b = r.sa_b[0]; w.sa1_b = r.sa_b[1]; w.sa2_b = r.sa_b[2]; w.sa3_b = r.sa_b[3];
    return w;
}

}  // namespace

// pc [B,J,133] (reference layout) -> f->vc [B,431,3] (vert431) ; taps: f->lbf
int launch_mdr(gator_ctx* c, FusedState* f, const float* pc, int B, void* stream, const float* x_out, const float* pose2d, bool half16) {
    (void)pose2d;
    // half16 (BASELINE config 3): the layers on ONE fp16 activation plane (XA = 3); needs the default weight / joint-tile forms (GATOR_MDR_X3=2)
    if (half16 && f->mdr_x3 != 2) return fail(GATOR_EUNSUPPORTED, "16-bit MDR layers need GATOR_MDR_X3=2 (the default)");
    const int xa = half16 ? 3 : f->mdr_x3;
    hipStream_t st = (hipStream_t)stream;
    const Weights& w = c->w;
    JointArgs ja;
    ja.pc = pc; ja.jw_p = f->jfeat_p; ja.jb = w.jfeat_b; ja.posj_T = f->posj_T; ja.jkv = f->jkv; ja.J = c->J;
    for (int i = 0; i < 3; ++i) { ja.n1w[i] = w.lay[i].n1w; ja.n1b[i] = w.lay[i].n1b; ja.wk_p[i] = f->lay[i].wk; ja.wv_p[i] = f->lay[i].wv; }
    ja.mdr_ctr = nullptr;
    ja.x2 = f->mdr_x3 == 2;
    if (pc) {
        if (f->mdr_persist != 0) { ja.mdr_ctr = f->mdr_ctr; f->mdr_ctr_clean = true; }
        StageTimer tm(c, "mdr_joint", stream);
        k_mdr_joint<<<B, 128, 0, st>>>(ja);
    }    // else: done by k_gat's epilogue / k_gat_joint
    const size_t per = (size_t)f->cap * kVT * 2 * kTile;      // one [B][14][2] tile set
    const size_t perq = (size_t)f->cap * kVT * 2 * (f->mdr_x3 == 1 ? kTileX3 : kTile);      // q/k/v tile sets: X3 tiles are 1.5x, fp32 and X2 tiles 4 KiB (X1 tiles 2 KiB: half of a set is used)
    float* set[3][4] = {{f->vf, f->q, f->k, f->v}, {f->vf + per, f->q + perq, f->k + perq, f->v + perq},
                        {f->vf + 2 * per, f->q + 2 * perq, f->k + 2 * perq, f->v + 2 * perq}};
    MdrArgs a{};
    a.B = B; a.J = c->J; a.jkv = f->jkv; a.pc = pc; a.xout = pc ? nullptr : x_out; a.vj = w.vj; a.tok_base = f->tok_base; a.tok_w3 = f->tok_w3;
    a.head_w = f->mdr_x3 ? f->wxbuf + (size_t)(f->head_w - f->lay[0].wq) / kTile * kTileX3 : f->head_w; a.head_b = f->head_b; a.hf = f->hf; a.lbf = c->block_taps ? f->lbf : nullptr;      // the "mdr_lbf2" tap costs 110 KB of stores per sample: recorded with the block taps only
    a.hpart = f->mdr_head_partials ? reinterpret_cast<double*>(f->hpart) : nullptr;      // default on; GATOR_MDR_HEAD_PARTIALS=0 at create: the whole head in k_mdr_head (A/B)
    a.bconv_w = w.bconv_w; a.hbn_w = w.bn_w; a.hbn_b = w.bn_b; a.hbn_mean = w.bn_mean; a.hbn_var = w.bn_var; a.halpha = c->alpha;
    a.lin_s = f->mdr_x3 == 2 ? std::ldexp(kActScale, f->mdr_wshift) : 1.0f;      // 4-product linears: 16 x activations, 2^wshift x weights
    a.lin_inv = 1.0f / a.lin_s;
    const int nwg = (B * kVT + 3) / 4;
#ifdef GATOR_DIAG
    {
        static const int cut = getenv("GATOR_MDR_CUT") ? atoi(getenv("GATOR_MDR_CUT")) : 0;
        static const int wm = (cut & 1) ? 0 : -1, kvm = (cut & 2) ? 0 : -1;
        GATOR_HIP_CHECK(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_mdr_wmask), &wm, sizeof(int), 0, hipMemcpyHostToDevice, st));
        GATOR_HIP_CHECK(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_mdr_kvmask), &kvm, sizeof(int), 0, hipMemcpyHostToDevice, st));
    }
    static const bool want_stamps = getenv("GATOR_MDR_STAMPS") != nullptr;
    static const size_t solo = getenv("GATOR_MDR_SOLO") ? 60 * 1024 : 0;      // 1 workgroup per CU (1 wave/SIMD)
    unsigned long long* d_st = nullptr;
    if (want_stamps) { GATOR_HIP_CHECK(hipMalloc(&d_st, 512 * sizeof(unsigned long long))); GATOR_HIP_CHECK(hipMemset(d_st, 0, 512 * sizeof(unsigned long long))); }
#else
    constexpr size_t solo = 0;
#endif
    MdrPersistArgs pa{};
    MdrChunkPlan plan{1, B, 0};
    // Persistent launch(es) or four per-stage launches?  Same tile body, bitwise the same results.  The persistent form has no
    // fractional generation per stage (R = workgroups per CU: 3.5 at B = 256 is billed as 4 by each of the four launches) and, cut
    // into chunks, keeps a sample's tiles cache-resident between stages; below R = 3 (B < ~220) its per-sample dependency chain
    // costs more than it saves.
    const bool ctr_clean = f->mdr_ctr_clean;      // zeroed for THIS call by the joint-token kernel queued just before (either entry point)
    f->mdr_ctr_clean = false;
    const double R = (double)nwg / f->n_cu;
    const bool auto_persist = R >= 3.0;      // (round 3 also asked for a wasted fractional generation >= 4 %; with chunked launches the persistent form wins from R = 3 on: sweep in DESIGN.md)
    bool persist = f->mdr_persist < 0 ? auto_persist : f->mdr_persist > 0;
    // the queues are per XCD and a workgroup serves the queue of the XCD it runs on: that drains every queue only when the device is
    // the whole 8-XCD part (a partitioned device shows fewer CUs; its workgroups would all sit on one XCD).  A placement that leaves
    // an XCD empty anyway is caught by k_mdr_head (completion counts) and answered by api.hip (four launches from then on).
    if (f->n_cu != 256 && f->mdr_persist < 0) persist = false;
#ifdef GATOR_DIAG
    if (want_stamps) persist = false;       // the stamps describe the per-stage launches
#endif
    for (int li = 0; li <= 3; ++li) {
#ifdef GATOR_DIAG
        a.stamps = (li == 1) ? d_st : nullptr;
#endif
        // four launches: two sets in turn.  One persistent launch: stage li writes set li and nothing else ever does, so no CU can
        // hold a stale L1 copy of a tile it reads (a line is only read after its one and only write) -- no cache invalidate in the
        // kernel at all (`buffer_inv sc1` per tile start cost 30 us per forward, and `sc0` does not touch the L1 in this mode)
        float** in = persist ? set[(li + 2) % 3] : set[(li + 1) & 1];
        float** out = persist ? set[li % 3] : set[li & 1];
        a.layer = li;
        a.vf_in = in[0]; a.q_in = in[1]; a.k_in = in[2]; a.v_in = in[3];
        a.vf_out = out[0]; a.q_out = out[1]; a.k_out = out[2]; a.v_out = out[3];
        if (li > 0) a.prev = make_layer(f, c, li - 1);
        if (li < 3) a.cur = make_layer(f, c, li);
        if (persist) { pa.st[li] = a; continue; }
        StageTimer tm(c, li == 0 ? "mdr_layer0" : (li < 3 ? "mdr_layer" : "mdr_attn_head"), stream);
        if (xa == 3) {
            if (li == 0) k_mdr_layer<0, 3><<<nwg, 256, solo, st>>>(a, nwg);
            else if (li < 3) k_mdr_layer<1, 3><<<nwg, 256, solo, st>>>(a, nwg);
            else k_mdr_layer<2, 3><<<nwg, 256, 0, st>>>(a, nwg);
        } else if (f->mdr_x3 == 2) {
            if (li == 0) k_mdr_layer<0, 2><<<nwg, 256, solo, st>>>(a, nwg);
            else if (li < 3) k_mdr_layer<1, 2><<<nwg, 256, solo, st>>>(a, nwg);
            else k_mdr_layer<2, 2><<<nwg, 256, 0, st>>>(a, nwg);
        } else if (f->mdr_x3 == 1) {
            if (li == 0) k_mdr_layer<0, 1><<<nwg, 256, solo, st>>>(a, nwg);
            else if (li < 3) k_mdr_layer<1, 1><<<nwg, 256, solo, st>>>(a, nwg);
            else k_mdr_layer<2, 1><<<nwg, 256, 0, st>>>(a, nwg);
        } else {
            if (li == 0) k_mdr_layer<0, 0><<<nwg, 256, 0, st>>>(a, nwg);
            else if (li < 3) k_mdr_layer<1, 0><<<nwg, 256, 0, st>>>(a, nwg);
            else k_mdr_layer<2, 0><<<nwg, 256, 0, st>>>(a, nwg);
        }
    }
#ifdef GATOR_DIAG
    static const bool want_ends = getenv("GATOR_MDR_ENDS") != nullptr;
    unsigned long long* d_ends = nullptr;
    if (want_ends && persist) {
        GATOR_HIP_CHECK(hipMalloc(&d_ends, 3 * 1024 * sizeof(unsigned long long)));
        GATOR_HIP_CHECK(hipMemset(d_ends, 0, 3 * 1024 * sizeof(unsigned long long)));
        GATOR_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_persist_ends), &d_ends, sizeof(d_ends)));
    }
#endif
    if (persist) {      // the four stages as persistent launches (k_mdr_persist): tickets and per-sample completion counts start from zero
        if (!ctr_clean) GATOR_HIP_CHECK(hipMemsetAsync(f->mdr_ctr, 0, mdr_ctr_words(B) * sizeof(unsigned), st));
        StageTimer tm(c, "mdr_layers", stream);
        int grid = 2 * f->n_cu;             // two workgroups per CU is what the registers allow; any grid drains the queues
        if (f->mdr_persist_grid > 0) grid = f->mdr_persist_grid;      // GATOR_MDR_PERSIST_GRID (tests: a grid that leaves XCDs without a workgroup)
        // A large batch runs as several launches over chunks of 256 .. 511 samples.  The tickets are stage-major, so a sample's Q/K/V/residual
        // tiles (448 KB) are read one stage after they were written: at B = 256 / 384 the 115 / 172 MB in between stay in the Infinity
        // Cache, at B = 512 and above they do not and the launch costs 1.545 - 1.565 us per sample instead of 1.495 (measured, round 4).
        // Chunks are independent (a tile depends on its own sample only) and each writes its own tiles exactly once, so the hand-off
        // rules of the kernel hold per launch; results are bitwise those of one launch.
        // Chunk size, measured at the end of round 4 (MDR stage in ms, one box, two repetitions; GATOR_MDR_PERSIST_CHUNK): B = 512: one
        // launch 0.79, 2 x 256 0.76; B = 1024: 3 x 342 1.525, 4 x 256 1.49; B = 2048: 6 x 342 2.99 - 3.18, 8 x 256 2.97 - 3.00; B = 4096:
        // 12 x 342 6.43, 16 x 256 6.34 -- so floor(B / 256) launches.  (Every chunk using the first chunk's tile region again -- legal: a
        // chunk-local sample keeps its XCD and a launch starts with an empty L1 -- changes nothing: the tiles are written before they are read.
        // Odd chunks on a second, low-priority stream so that a chunk's workgroups move into the slots the previous chunk's tail frees:
        // nothing either, B = 512 .. 2048, two repetitions.)
        // (the one-plane form's Q / K / V tiles are half the size: chunks of 342 - 512 samples measure 6 % faster than 256 at B = 2 048 -- 3.25
        // against 3.45 ms per forward, two repetitions, one box -- so it runs ceil(B / 384) launches)
        int nch = f->mdr_persist_chunk > 0 ? (B + f->mdr_persist_chunk - 1) / f->mdr_persist_chunk : (xa == 3 ? (B + 383) / 384 : B / 256);
        if (nch < 1) nch = 1;
        if (nch > kMdrCtrChunks) nch = kMdrCtrChunks;
        const size_t tq = f->mdr_x3 == 1 ? kTileX3 : (xa == 3 ? kTileX1 : kTile);
        plan = MdrChunkPlan{nch, B / nch, B % nch};
        for (int ch = 0, b0 = 0; ch < nch; ++ch) {
            const int n = B / nch + (ch < B % nch ? 1 : 0);
            MdrPersistArgs pc_ = pa;
            for (int li = 0; li <= 3; ++li) {
                MdrArgs& s = pc_.st[li];
                const size_t ov = (size_t)b0 * kVT * 2 * kTile, oq = (size_t)b0 * kVT * 2 * tq;
                s.B = n;
                s.vf_in += ov; s.q_in += oq; s.k_in += oq; s.v_in += oq;
                s.vf_out += ov; s.q_out += oq; s.k_out += oq; s.v_out += oq;
                s.jkv += (size_t)b0 * 12 * kTile;
                if (s.pc) s.pc += (size_t)b0 * c->J * 133;
                if (s.xout) s.xout += (size_t)b0 * c->J * 3;
                s.hf += (size_t)b0 * kV * 32;
                if (s.hpart) s.hpart += (size_t)b0 * kVT * 64;
                if (s.lbf) s.lbf += (size_t)b0 * kV * kE;
            }
            pc_.ctr = f->mdr_ctr + plan.block(ch);
            if (xa == 3) k_mdr_persist<3><<<grid, 256, 0, st>>>(pc_);
            else if (f->mdr_x3 == 2) k_mdr_persist<2><<<grid, 256, 0, st>>>(pc_);
            else if (f->mdr_x3 == 1) k_mdr_persist<1><<<grid, 256, 0, st>>>(pc_);
            else k_mdr_persist<0><<<grid, 256, 0, st>>>(pc_);
            b0 += n;
        }
    }
#ifdef GATOR_DIAG
    if (d_ends) {      // synchronous read-back (diagnostic build only): when did each workgroup of the LAST persistent launch start / take its last ticket / end?
        std::vector<unsigned long long> he(3 * 1024);
        GATOR_HIP_CHECK(hipMemcpy(he.data(), d_ends, he.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned long long* none = nullptr;
        GATOR_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_persist_ends), &none, sizeof(none)));
        GATOR_HIP_CHECK(hipFree(d_ends));
        unsigned long long t0 = ~0ull, t1 = 0;
        int n = 0;
        for (int i = 0; i < 1024; ++i) if (he[3 * i + 2]) { t0 = std::min(t0, he[3 * i]); t1 = std::max(t1, he[3 * i + 2]); ++n; }
        if (n) {
            double idle = 0, lastlen = 0, first_end = 1e30, start_spread = 0;
            std::vector<double> ends;
            for (int i = 0; i < 1024; ++i) if (he[3 * i + 2]) {
                idle += (double)(t1 - he[3 * i + 2]) / 100.0;
                lastlen += (double)(he[3 * i + 2] - he[3 * i + 1]) / 100.0;
                first_end = std::min(first_end, (double)(he[3 * i + 2] - t0) / 100.0);
                start_spread = std::max(start_spread, (double)(he[3 * i] - t0) / 100.0);
                ends.push_back((double)(he[3 * i + 2] - t0) / 100.0);
            }
            std::sort(ends.begin(), ends.end());
            fprintf(stderr, "[k_mdr_persist ends, B=%d, %d workgroups] span %.1f us; workgroup starts spread over %.1f us; ends: first %.1f, p10 %.1f, median %.1f, p90 %.1f, last %.1f us; "
                            "mean idle before the last workgroup ends %.1f us; mean length of a workgroup's last stage-3 ticket %.1f us\n",
                    B, n, (double)(t1 - t0) / 100.0, start_spread, first_end, ends[n / 10], ends[n / 2], ends[n * 9 / 10], ends[n - 1], idle / n, lastlen / n);
        }
    }
    if (d_st) {
        unsigned long long hst[512];
        GATOR_HIP_CHECK(hipMemcpy(hst, d_st, sizeof(hst), hipMemcpyDeviceToHost));
        const int ns = (B * kVT + 63) / 64;
        unsigned long long t0 = ~0ull;
        for (int i = 0; i < ns && i < 120; ++i) if (hst[8 + 4 * i] && hst[8 + 4 * i] < t0) t0 = hst[8 + 4 * i];
        fprintf(stderr, "[k_mdr_layer<1> timeline, every 64th tile: tile:(start_us end_us kcycles xcc cu)]");
        for (int i = 0; i < ns && i < 120; ++i)
            fprintf(stderr, " %d:(%.0f,%.0f,%llu,x%llu,cu%llu.se%llu)", i * 64, (hst[8 + 4 * i] - t0) / 100.0, (hst[9 + 4 * i] - t0) / 100.0,
                    hst[10 + 4 * i] / 1000, hst[11 + 4 * i] >> 32, (hst[11 + 4 * i] >> 8) & 0xf, (hst[11 + 4 * i] >> 13) & 0x7);
        fprintf(stderr, "\n");
        GATOR_HIP_CHECK(hipFree(d_st));
        fprintf(stderr, "[k_mdr_layer<1> stamps, tile 0] attention(2 heads)=%llu outproj+res=%llu cross-attn block=%llu mlp=%llu customLN+qkv=%llu\n",
                hst[0], hst[1], hst[2], hst[3], hst[4]);
    }
#endif
    HeadArgs ha;
    ha.hf = f->hf; ha.bn_w = w.bn_w; ha.bn_b = w.bn_b; ha.bn_mean = w.bn_mean; ha.bn_var = w.bn_var;
    ha.bconv_w = w.bconv_w; ha.bconv_b = w.bconv_b; ha.vc = f->vc; ha.vcp = f->vcp;
    ha.persist_ctr = persist ? f->mdr_ctr : nullptr;
    ha.plan = plan;
    ha.status = c->status_dev;
    ha.vcp2 = f->x3 && f->up_x2 ? (_Float16*)f->vcp3 : nullptr;
    ha.vcp3 = f->x3 && !f->up_x2 ? (__bf16*)f->vcp3 : nullptr; ha.vcp3_plane = upsample_x3_vcp_elems(f->cap) / 3;     // plane stride fixed by the workspace capacity
    ha.alpha = c->alpha;
    {
        StageTimer tm(c, "mdr_head", stream);
        if (a.hpart) k_mdr_head_finish<<<B, 256, 0, st>>>(ha, a.hpart);      // the conv came out of the tiles as partial sums: what is left is light
        else if (B <= 2 * f->n_cu) k_mdr_head<512, true><<<B, 512, 0, st>>>(ha);
        else k_mdr_head<512, false><<<B, 512, 0, st>>>(ha);
    }
    GATOR_HIP_CHECK(hipGetLastError());
    c->set_tap(TAP_MDR_LBF2, c->block_taps ? f->lbf : nullptr, c->block_taps ? (int64_t)B * kV * kE : 0);
    c->set_tap(TAP_VERT431, f->vc, (int64_t)B * kV * 3);
    return GATOR_OK;
}

}  // namespace gator
