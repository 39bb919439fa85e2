// GAT graph-aware transformer encoder (lib/models/GAT.py:133-152, GATBlock :33-43), six blocks in ONE launch, one workgroup per
// sample -- the form for one sample per CU (batches up to one round of the chip), built around what bounds that case.
//
// What bounds it (tools/microbench/wstream.hip, profiles/r03_microbench_weight_stream.txt).  A sample walks 1 560 operand tiles of
// split-precision weights (9.4 MB) that nobody else on its CU can share.  Streamed alone they take 85 us (54 B/clk through the
// CU's vector L1); with their 12 MFMAs each issued by the same four waves 101 us, provided those waves do nothing else.  k_gat
// (gat_fused.hip) has each of its four waves alternate between that stream, the J x J attention, LayerNorms, GELUs and operand
// splits, so the stream stops whenever the wave does anything else: 226 us.
//
// Two roles, eight waves (two per SIMD):
//   waves 0-3  "product waves": ONLY the weight stream (five tiles in flight in registers, one contiguous per-wave stream in
//              consumption order, packed at create), the bf16 MFMAs of the token-wise linears with the activation operand read
//              from LDS, and the raw fp32 accumulator parked in LDS.  Wave w owns channel block w of every 128-wide layer.
//   waves 4-7  "helper waves": everything else.  Helper w picks up product wave w's raw tile one step later and does bias,
//              residual, LayerNorm, the J x J attention of heads 2w / 2w+1 (fp32-input MFMA, scores and probabilities in
//              registers), the MGCN adjacency product, the hop-1 / hop-2 aggregations, GELU and the exact hi/mid/lo split into
//              the next operand tile.
// (Round 4, measured and not kept: s_setprio 1 / 2 / 3 on the helpers only inside their long steps -- scores + softmax, P.V, the four
// GELU steps -- moves k_gat8 by less than 1 us; round 3 had the static form making the sum worse.  The GELU as plain instead of
// packed FMAs, the file without packed fp32 or without SLP vectorisation: no change either.  With the diagnostic library:
// GATOR_GAT8_DBG=16 (helpers keep only the barriers and the L2 warm-up) 128 us against 167 for the whole kernel on one box -- the
// helpers cost 39 us, and their steps are latency (LDS round trips, barrier skew), not instruction count: 40 % fewer VALU
// instructions in the GELU steps bought 2.5 us.)
// One workgroup barrier per step, 22 steps per block; in 16 of them a product wave streams one 4-tile unit (a 32-channel output
// block over K = 128) while the helper finishes the previous unit, six are helper-only (SB, the hop aggregations, residual +
// LayerNorm twice).  Both roles live in disjoint branches of one kernel so that neither pays the other's registers; the barriers
// are numbered GAT8_BAR(n) in both and tests/test_host_cpu.py checks that the two sequences are identical.
//
// LDS (149 KiB): A = 4 operand tiles (Y = LN1(x) | Y2 = LN2(x) | hidden blocks 4w+3), B = 12 operand tiles (AT | SB | FB, later
// hidden blocks 4w+j, j < 3), R0 / R1 = 2 x 4 raw tiles (product wave w writes R[step & 1][w], helper w reads it in the next
// step), X = 4 fp32 tiles: the embedding's output (first LayerNorm), then the four partial hop-2 linears; ST = per-wave LayerNorm
// statistics.  k_gat8<false, 16>: arithmetic and operand formats are k_gat's (x3_common.h: exact three-way bf16 split, six partial
// products, fp32 accumulation); it agrees with k_gat to fp32 rounding (the MLP's four partial sums group the hidden blocks
// differently, the LayerNorm variance is combined from per-wave statistics), not bit for bit.  k_gat8<true, LR> (default): the
// token-wise products on four partial products, the J x J attention and the hop aggregations on fp16 planes; LR = the
// registers of a rows-over-tokens tile that hold tokens which exist (10 for J <= 18, 12 for J <= 20): the helpers skip the others, and
// the MLP's hidden tiles go through a C-layout GELU (hid_store) -- bitwise the results of the all-rows form.
#include "fused_common.h"
#include "fused_state.h"
#include "x3_common.h"

#include <algorithm>
#include <cmath>
#include <type_traits>
#include <cstdlib>
#include <vector>

namespace gator {
namespace {

constexpr float kLog2e8 = 1.4426950408889634f;
#ifndef GAT8_NT
#define GAT8_NT 5
#endif
constexpr int kNT = GAT8_NT;                            // weight tiles in flight per product wave.  6 (one dummy tile per block, +28 VGPRs) was measured:
                                                        // no change (188 - 192 us either way on one box), so the stream is not short of bytes in flight
constexpr int kUseTiles = 65;                           // q k v h0 h1 proj lin0 (4 each) + lin1 (1) + back (4) + fc1 (16) + fc2 (16)
constexpr int kPadTiles = (kNT - kUseTiles % kNT) % kNT;       // dummy tiles behind a block so that every block starts at slot 0
constexpr int kBlkTiles = kUseTiles + kPadTiles;
// tile offsets of the units inside a block (their slot = offset mod kNT, compile-time)
enum { OFF_Q = 0, OFF_K = 4, OFF_V = 8, OFF_H0 = 12, OFF_H1 = 16, OFF_PROJ = 20, OFF_LIN0 = 24, OFF_LIN1 = 28, OFF_BACK = 29, OFF_FC1 = 33, OFF_FC2 = 49 };
constexpr int kWaveTiles = kBlkTiles * kDepth;          // per product wave
constexpr int kStreamFloats = (4 * kWaveTiles + kNT) * kTileX3;
constexpr int kTileH3B = 2 * 512 + 256;                // floats of a byte-lo tile: hi and mid planes as in H3 (1 KiB per k-step each), lo as one byte per weight
constexpr int kStreamFloatsB = (4 * kWaveTiles + kNT) * kTileH3B;

// offsets into a block's vector table (fused_api.hip packs them in this order, 2048 floats per block)
enum { V_N1W = 0, V_N1B = 128, V_QKVB = 256, V_PROJB = 640, V_GCNB = 768, V_LIN0B = 896, V_BACKB = 1024, V_N2W = 1152,
       V_N2B = 1280, V_FC1B = 1408, V_FC2B = 1920 };

struct Gat8Blk { const float *back32, *mc, *mdT, *aoffT, *f1b, *vecs; };
// Epilogue of k_gat8<..., TAIL = true> (round 6): what gat_tail.hip's two launches do for the batch, done by the workgroup that owns
// the sample and has its feat on chip.
struct Gat8Tail {
    const float *lifter_w, *lifter_b;       // lifter.weight [3J][128J] as the reference stores it, bias [3J]  (GAT.py:151-152)
    float* x_out;                           // [B][3J] = pose3d [B][J][3] (mm)
    float* jkv;                             // nullptr: lifter only (stand-alone GAT entry point)
    unsigned* mdr_ctr;                      // non-null: zero k_mdr_persist's tickets / completion counts of the forward (as k_gat_joint does)
    int ctr_B;                              //           ... of a forward of ctr_B samples (mdr_ctr_words)
    const float *jf5, *jf_h3, *jf_b, *posj_T;      // get_joint_feature: columns 0..4 as [5][64], columns 5..132 as H3 tiles [2][4], bias; pos_j tiles
    const float *j_n1w[3], *j_n1b[3], *j_wk_h3[3], *j_wv_h3[3];      // per LBF layer: norm1, wk / wv as H3 tiles [2][2] (the MDR layers' own weight image)
    const float *jf_p, *j_wk_p[3], *j_wv_p[3];     // the same weights as packed fp32 tiles (GAT8_TAIL_F32: fp32-input MFMA forms of the two linears)
    float jf_inv, kv_inv;                          // 1 / (16 x 2^weight shift) of the two H3 weight sets (x3_common.h: four-product linears)
#ifdef GATOR_DIAG
    unsigned long long* tstamps;                   // [8 waves][8] s_memtime stamps of workgroup B/2's epilogue (GATOR_GAT_STAMPS)
#endif
    int warm_n;                                    // L2 warm-up of the epilogue's weights: workgroups per XCD that share it (0: off)
};

struct Gat8Args {
    int B, J;
    const float* pose2d;
    const float *gl0_W, *gl0_b, *gn_w, *gn_b, *gl3_p, *gl3_b, *posT;
    const float *biasT, *m1T, *m2T;
    const float *norm_w, *norm_b;
    const float* wstream;            // [4 product waves][390 tiles][kTileX3] (+ kNT tiles of slack behind the last wave)
    Gat8Blk blk[kDepth];
    float* feat;
    int tapB;
    float* blk_tap;
    float lin_inv;                   // H4 form: 1 / (16 x 2^weight shift), the factor every raw product tile carries
    Gat8Tail tl;                     // k_gat8<..., TAIL = true>: the lifter and the MDR joint tokens as the kernel's epilogue
    int pf_n, pf_loads;              // L2 warm-up of the next block's weights: workgroups per XCD that share it, 8 KiB touches per helper wave (0: off)
#ifdef GATOR_DIAG
    unsigned long long* stamps;      // [2 roles][kDepth][23 steps][work, wait]
    int dbg;                         // GATOR_GAT8_DBG: 1 = helpers only keep the barriers (what do the product waves cost alone?)
#endif
};

// LDS map (floats)
constexpr int kA = 0;                                   // 4 operand tiles
constexpr int kBq = kA + 4 * kTileX3;                   // 12 operand tiles
constexpr int kR = kBq + 12 * kTileX3;                  // R0 (4 raw tiles) | R1 (4 raw tiles)
constexpr int kXo = kR + 8 * kTile;                     // 4 fp32 tiles
constexpr int kDummy = kXo + 4 * kTile;                 // 4 x 1 KiB landing window of the L2 warm-up (never read)
constexpr int kStat = kDummy + 1024;                     // [4 helper waves][32 tokens][mean, M2] of the wave's 32 channels (LayerNorm statistics)
constexpr int kGat8LdsFloats = kStat + 256;             // 149 KiB
#ifdef GATOR_DIAG
constexpr int kDiagStamps = 2 * kDepth * 23 * 2 + kDepth * 8;      // u64 cycle stamps, kept in LDS while the kernel runs (a global
constexpr int kDiagLdsFloats = 2 * kDiagStamps;                    // store waits ~0.5k cycles behind the product waves' stream and
#else                                                               // would be measured as work of the step it sits in)
constexpr int kDiagLdsFloats = 0;
#endif

#ifdef GATOR_DIAG
// diagnostic library only (python -m gator_amd.build --diag; GATOR_GAT_STAMPS=1): per role, block and step the cycles spent working
// (from leaving the previous barrier to arriving at this one) and waiting in the barrier, of workgroup 0's waves 0 and 4; written to
// LDS, copied out by GAT8_STAMPS_OUT when the role is done
#define GAT8_BAR(n)                                                                                      \
    do {                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        const unsigned long long t_arr_ = __builtin_amdgcn_s_memtime();                                  \
        __syncthreads();                                                                                 \
        const unsigned long long t_go_ = __builtin_amdgcn_s_memtime();                                   \
        if (st_out) { st_out[((size_t)bi_ * 23 + (n)) * 2] = t_arr_ - st_last; st_out[((size_t)bi_ * 23 + (n)) * 2 + 1] = t_go_ - t_arr_; } \
        st_last = t_go_;                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                               \
    } while (0)
#define GAT8_STAMPS_OUT(n0, n1)                                                                          \
    do { if (st_out) for (int i_ = (n0); i_ < (n1); ++i_) a.stamps[i_] = st_lds[i_]; } while (0)
#define GAT8_SUB(k)                                                                                     \
    do {                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        if (st_out) st_out[(size_t)kDepth * 23 * 2 + bi_ * 8 + (k)] = __builtin_amdgcn_s_memtime() - st_last;              \
        __builtin_amdgcn_sched_barrier(0);                                                               \
    } while (0)
#else
#define GAT8_BAR(n) __syncthreads()
#define GAT8_SUB(k)
#define GAT8_STAMPS_OUT(n0, n1)
#endif

// A helper wave is alone with its own dependency chains (its SIMD partner issues MFMAs, nobody fills its latency slots), so the
// reductions are trees of independent partial sums and the GELU walks its polynomial level by level over all eight register pairs.
__device__ __forceinline__ float rsum128(const f32x16 (&x)[4]) {
    float p[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        p[q] = 0.f;
#pragma unroll
        for (int r = 4 * q; r < 4 * q + 4; ++r) p[q] += (x[0][r] + x[1][r]) + (x[2][r] + x[3][r]);
    }
    const float s = (p[0] + p[1]) + (p[2] + p[3]);
    return s + xhalf(s);
}

// gelu_f2 (fused_common.h) over a whole register tile, the eight pairs advanced together: same operations per element, so the same
// bits; only the instruction order differs (one pair after the other is a chain of 12 dependent packed operations, 8 times)
__device__ __forceinline__ void gelu_tile8(f32x16& v) {     // (operates on true-scale values in both forms)
    f32x2 x[8], t[8], r[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        x[p][0] = v[2 * p]; x[p][1] = v[2 * p + 1];
        const f32x2 a = x[p] * 0.70710678118654752440f;
        t[p][0] = fminf(fabsf(a[0]), 4.3f);
        t[p][1] = fminf(fabsf(a[1]), 4.3f);
    }
    const float c[8] = {4.369443071e-04f, -1.460381877e-03f, -8.251648338e-04f, 2.830188636e-02f, -1.485066472e-01f, -9.184098145e-01f,
                        -1.627909326e+00f, -9.999999783e-01f};
#pragma unroll
    for (int p = 0; p < 8; ++p) r[p] = pk_fma(f32x2(-4.435285315e-05f), t[p], f32x2(c[0]));
#pragma unroll
    for (int k = 1; k < 8; ++k) {
#pragma unroll
        for (int p = 0; p < 8; ++p) r[p] = pk_fma(r[p], t[p], f32x2(c[k]));
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        r[p][0] = __builtin_amdgcn_exp2f(r[p][0]);
        r[p][1] = __builtin_amdgcn_exp2f(r[p][1]);
    }
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const f32x2 up = 1.0f - r[p];
        f32x2 phi;
        phi[0] = x[p][0] < 0.f ? r[p][0] : up[0];
        phi[1] = x[p][1] < 0.f ? r[p][1] : up[1];
        const f32x2 y = x[p] * phi;
        v[2 * p] = y[0]; v[2 * p + 1] = y[1];
    }
}

// The MLP's hidden tiles in the four-product form.  fc1 leaves its accumulator in C layout (channel on the lane, token rows in the
// registers), so the GELU and the split run over the LR registers that hold tokens instead of all 16 -- in T layout 15 of a tile's
// 32 token lanes are padding at J = 17 and every instruction pays for them.  fc2 wants the hidden activations as an operand with k =
// channel, i.e. the transpose of what a C-layout lane holds, and the operand goes through LDS anyway: lane (c, h) writes its value of
// token t as ONE half into chunk (plane, s = c >> 4, lane' = t + 32 ((c >> 2) & 1)), element 4 ((c >> 3) & 1) + (c & 3).  The chunk
// index is XOR-ed with s + 2 (lane' >> 5) so that the 64 lanes of one such store hit 32 different banks (unswizzled: 8); the fc2
// units read the tile with the same XOR (x2_load_swz).  Token rows that do not exist are not written: their halves keep whatever
// finite operand the tile held before, which only ever reaches the same padding token's outputs.
__device__ __forceinline__ X2 x2_load_swz(const float* __restrict__ tile, int lane) {
    const f16x8* q = reinterpret_cast<const f16x8*>(tile);
    X2 o;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int s = 0; s < 2; ++s) o.p[pl][s] = q[(pl * 2 + s) * 64 + (lane ^ (s + 2 * (lane >> 5)))];
    return o;
}
struct HidAddr { int base, off[4]; };                  // byte offsets of a C-layout lane into a hidden operand tile
__device__ __forceinline__ HidAddr hid_addr(int lane) {
    const int c = lane & 31, hh = lane >> 5, s = c >> 4, kh = (c >> 2) & 1, jj = 4 * ((c >> 3) & 1) + (c & 3), swz = s + 2 * kh;
    HidAddr a;
    a.base = s * 1024 + (4 * hh + 32 * kh) * 16 + jj * 2;
#pragma unroll
    for (int q = 0; q < 4; ++q) a.off[q] = (q ^ swz) * 16;
    return a;
}
// two planes of 16 x v[r] for the token rows r < LR, scattered into the operand tile
template <int LR>
__device__ __forceinline__ void hid_store(float* tile, const HidAddr& ha, const f32x16& v) {
    char* lb = reinterpret_cast<char*>(tile) + ha.base;
#pragma unroll
    for (int r = 0; r < LR; ++r) {
        const float x = v[r] * 16.0f;
        const _Float16 hi = (_Float16)x;
        const _Float16 lo = (_Float16)(x - (float)hi);
        char* pr = lb + ha.off[r & 3] + (r >> 2) * 128;
        *reinterpret_cast<_Float16*>(pr) = hi;
        *reinterpret_cast<_Float16*>(pr + 2048) = lo;
    }
}
// gelu_tile8 over the first NP register pairs only
template <int NP>
__device__ __forceinline__ void gelu_pairs(f32x16& v) {
    f32x2 x[NP], t[NP], r[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        x[p][0] = v[2 * p]; x[p][1] = v[2 * p + 1];
        const f32x2 a = x[p] * 0.70710678118654752440f;
        t[p][0] = fminf(fabsf(a[0]), 4.3f);
        t[p][1] = fminf(fabsf(a[1]), 4.3f);
    }
    const float c[8] = {4.369443071e-04f, -1.460381877e-03f, -8.251648338e-04f, 2.830188636e-02f, -1.485066472e-01f, -9.184098145e-01f,
                        -1.627909326e+00f, -9.999999783e-01f};
#pragma unroll
    for (int p = 0; p < NP; ++p) r[p] = pk_fma(f32x2(-4.435285315e-05f), t[p], f32x2(c[0]));
#pragma unroll
    for (int k = 1; k < 8; ++k) {
#pragma unroll
        for (int p = 0; p < NP; ++p) r[p] = pk_fma(r[p], t[p], f32x2(c[k]));
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        r[p][0] = __builtin_amdgcn_exp2f(r[p][0]);
        r[p][1] = __builtin_amdgcn_exp2f(r[p][1]);
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const f32x2 up = 1.0f - r[p];
        f32x2 phi;
        phi[0] = x[p][0] < 0.f ? r[p][0] : up[0];
        phi[1] = x[p][1] < 0.f ? r[p][1] : up[1];
        const f32x2 y = x[p] * phi;
        v[2 * p] = y[0]; v[2 * p + 1] = y[1];
    }
}
// the first ceil(LR / 4) register groups of a block
template <int LR>
__device__ __forceinline__ f32x16 load_block_rows(const float* __restrict__ p, int lane) {
    const f32x4* q = reinterpret_cast<const f32x4*>(p) + lane;
    f32x16 v = zero16();
#pragma unroll
    for (int g = 0; g < (LR + 3) / 4; ++g) {
        const f32x4 t = q[g * 64];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[4 * g + j] = t[j];
    }
    return v;
}

// nn.LayerNorm(128) of the residual stream X (4 fp32 tiles in LDS): statistics over all four tiles, result for this wave's own
// channel block only (xw = its registers, the same values as X[w]); same operation order as k_gat's layernorm128
__device__ __forceinline__ f32x16 ln_own(const float* X, const f32x16& xw, const f32x16& wv, const f32x16& bv, int lane) {
    f32x16 x[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) x[kb] = load_block(X + kb * kTile, lane);
    const float mean = rsum128(x) * (1.0f / 128.0f);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) { x[kb] = x[kb] - mean; x[kb] = x[kb] * x[kb]; }
    const float rstd = 1.0f / sqrtf(rsum128(x) * (1.0f / 128.0f) + 1e-5f);
    return (xw - mean) * rstd * wv + bv;
}

// LayerNorm(128) inside the block loop without a second pass over the residual stream: the wave that updates its 32 channels of a
// token leaves their (mean, sum of squared deviations) in LDS, and the four pairs of a token are combined exactly (Chan et al.):
// M2 = sum M2_w + 32 sum (mean_w - mean)^2.  ln_own re-read all four tiles and reduced twice in a step the product waves wait through.
__device__ __forceinline__ void tile_stats(const f32x16& xw, float* st_w, int lane) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) s += (xw[4 * q] + xw[4 * q + 1]) + (xw[4 * q + 2] + xw[4 * q + 3]);
    s += xhalf(s);
    const float m = s * (1.0f / 32.0f);
    float p[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        p[q] = 0.f;
#pragma unroll
        for (int r = 4 * q; r < 4 * q + 4; ++r) { const float d = xw[r] - m; p[q] += d * d; }
    }
    float m2 = (p[0] + p[1]) + (p[2] + p[3]);
    m2 += xhalf(m2);
    if (lane < 32) { st_w[2 * lane] = m; st_w[2 * lane + 1] = m2; }
}
__device__ __forceinline__ f32x16 ln_stats(const float* ST, const f32x16& xw, const f32x16& wv, const f32x16& bv, int lane) {
    const int tok = lane & 31;
    float m[4], q[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { m[i] = ST[i * 64 + 2 * tok]; q[i] = ST[i * 64 + 2 * tok + 1]; }
    const float mean = ((m[0] + m[1]) + (m[2] + m[3])) * 0.25f;
    float dev = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float d = m[i] - mean; dev += d * d; }
    const float m2 = ((q[0] + q[1]) + (q[2] + q[3])) + 32.0f * dev;
    const float rstd = 1.0f / sqrtf(m2 * (1.0f / 128.0f) + 1e-5f);
    return (xw - mean) * rstd * wv + bv;
}

// ---- token rows that do not exist.  A tile whose REGISTERS run over tokens (C layout: v[r] <-> token kap(r) + 4h; S^T and P^T with
// the key on the row) has live data only in registers r < LR: tokens 0 .. J-1 with J = 17 / 19 sit in r <= 8 / 10.  The rows behind
// them hold products against zero operand rows or masked scores (probability exactly 0): every instruction spent on them is wasted,
// and the helper waves are the long side of most steps.  LR (even: 10 for J <= 18, 12 for J <= 20; 16 = every row, the six-product form) is a template
// parameter of the kernel; skipping dead rows changes no bit of a live value (sums lose terms that are exactly +0).
template <int LR>
__device__ __forceinline__ X2 x2_split_rows(const f32x16& v) {     // x2_split with zero halves for the dead rows
    X2 o;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (8 * s + j < LR) {
                const float x = v[8 * s + j];
                const _Float16 hh = (_Float16)x;
                o.p[0][s][j] = hh;
                o.p[1][s][j] = (_Float16)(x - (float)hh);
            } else {
                o.p[0][s][j] = (_Float16)0.f;
                o.p[1][s][j] = (_Float16)0.f;
            }
        }
    return o;
}
// (a + b) + (c + d) over the live registers of group q, the association of the full tree (a dead term is an exact +0)
template <int LR, class F>
__device__ __forceinline__ float tree4(int q, F f) {
    const int n = LR - 4 * q;          // live registers in this group
    if (n <= 0) return 0.f;
    if (n == 1) return f(4 * q);
    if (n == 2) return f(4 * q) + f(4 * q + 1);
    if (n == 3) return (f(4 * q) + f(4 * q + 1)) + f(4 * q + 2);
    return (f(4 * q) + f(4 * q + 1)) + (f(4 * q + 2) + f(4 * q + 3));
}
// dot16 (fused_common.h) over the live k rows only: the same two chains (even / odd registers)
template <int LR>
__device__ __forceinline__ f32x16 dot16_rows(const f32x16& A, const f32x16& B, f32x16 init) {
    f32x16 e = init, o = zero16();
#pragma unroll
    for (int r = 0; r < LR; r += 2) {
        e = GATOR_MFMA(A[r], B[r], e);
        o = GATOR_MFMA(A[r + 1], B[r + 1], o);
    }
    return e + o;
}

// ---- product wave: one unit = the K = 128 contraction of one 32-channel output block (4 weight tiles) --------------------------
// CL = false: weights as A operand -> T-layout accumulator (token on the lane); CL = true: activations as A -> C-layout.
// Slot indices are compile-time (S0 = tile offset of the unit inside its block, mod kNT); every slot is refilled with the tile
// kNT positions further down the wave's stream right after the MFMAs that read it are queued.
// LDS-DMA through inline asm (hipcc drains the builtin form with vmcnt(0) before the next LDS access, see gat_tiled.hip): used
// only to pull lines into this XCD's L2 - the bytes land in a window nobody reads, no registers are held
__device__ __forceinline__ void glds4(const float* gsrc, const float* lds_dst) {      // one dword per lane, per-lane source address
    unsigned keep;
    const unsigned dst = (unsigned)(unsigned long long)lds_dst;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

// One tile product with the refill of its weight slot woven in: a plane's load (tile kNT positions further down the stream) is
// issued right behind the LAST MFMA that reads that plane, so it issues in the shadow of the next MFMA.  Issued as a burst behind
// the twelve MFMAs (k_gat's form) the six 1 KiB loads hold the in-order wave at the memory pipeline's issue rate while the matrix
// pipe idles: measured here as 3.2k cycles per 4-tile unit against 1.5k of MFMA time.  sched_barrier pins the order.
template <bool CL>
__device__ __forceinline__ void tile_mma_refill(X3& w, const X3& b, f32x16& acc, f32x16& /* six-product form: one accumulator */, const float* __restrict__ wp, int lane) {
    const bf16x8* q = reinterpret_cast<const bf16x8*>(wp) + lane;
#define GAT8_MM(wpl, bpl, s) acc = CL ? GATOR_MFMA_BF16(b.p[bpl][s], w.p[wpl][s], acc) : GATOR_MFMA_BF16(w.p[wpl][s], b.p[bpl][s], acc)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        GAT8_MM(2, 0, s);                                  // lo*hi
        __builtin_amdgcn_sched_barrier(0);
        w.p[2][s] = q[(2 * 2 + s) * 64];
        __builtin_amdgcn_sched_barrier(0);
        GAT8_MM(0, 2, s);                                  // hi*lo
        GAT8_MM(1, 1, s);                                  // mid*mid
        GAT8_MM(1, 0, s);                                  // mid*hi
        __builtin_amdgcn_sched_barrier(0);
        w.p[1][s] = q[(1 * 2 + s) * 64];
        __builtin_amdgcn_sched_barrier(0);
        GAT8_MM(0, 1, s);                                  // hi*mid
        GAT8_MM(0, 0, s);                                  // hi*hi
        __builtin_amdgcn_sched_barrier(0);
        w.p[0][s] = q[(0 * 2 + s) * 64];
        __builtin_amdgcn_sched_barrier(0);
    }
#undef GAT8_MM
}
// The four-product form (x3_common.h: weights exact on three fp16 planes, activations on two): w_hi a_lo | w_lo a_hi, w_mid a_hi,
// w_hi a_hi -- 8 MFMAs per tile; the refill of a plane still follows the last MFMA that reads it.  The three cross products go to
// their OWN accumulator `acs` (2^-11 of the result: its roundings do not matter), so the main one is rounded twice per tile instead of
// eight times -- every MFMA rounds its accumulator at the accumulator's magnitude, and in-product accumulation is where the path's
// fp32 noise comes from (x3_common.h, x2_mma).  unit4 adds the two once, in the product wave's idle vector slots.
template <bool CL>
__device__ __forceinline__ void tile_mma_refill(H3& w, const X2& b, f32x16& acc, f32x16& acs, const float* __restrict__ wp, int lane) {
    const f16x8* q = reinterpret_cast<const f16x8*>(wp) + lane;
#define GAT8_MM(wpl, bpl, s, AC) AC = CL ? GATOR_MFMA_F16(b.p[bpl][s], w.p[wpl][s], AC) : GATOR_MFMA_F16(w.p[wpl][s], b.p[bpl][s], AC)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        GAT8_MM(0, 1, s, acs);                             // w hi  * a lo
        GAT8_MM(2, 0, s, acs);                             // w lo  * a hi
        __builtin_amdgcn_sched_barrier(0);
        w.p[2][s] = q[(2 * 2 + s) * 64];
        __builtin_amdgcn_sched_barrier(0);
        GAT8_MM(1, 0, s, acs);                             // w mid * a hi
        __builtin_amdgcn_sched_barrier(0);
        w.p[1][s] = q[(1 * 2 + s) * 64];
        __builtin_amdgcn_sched_barrier(0);
        GAT8_MM(0, 0, s, acc);                             // w hi  * a hi
        __builtin_amdgcn_sched_barrier(0);
        w.p[0][s] = q[(0 * 2 + s) * 64];
        __builtin_amdgcn_sched_barrier(0);
    }
#undef GAT8_MM
}
// The one-plane form (BASELINE config 3, the 16-bit operand mode: DESIGN.md 4e): activations on their hi plane only, weights on their hi and mid
// planes -- w_mid a_hi | w_hi a_hi, four MFMAs per tile, and the lo plane of the weight stream (a third of its bytes) is never loaded.
// Same tile geometry, same stream; a type of its own so that the overloads below pick the form.
struct H3P2 { f16x8 p[2][2]; };      // [plane hi/mid][k-step]: 16 VGPRs
template <bool CL>
__device__ __forceinline__ void tile_mma_refill(H3P2& w, const X2& b, f32x16& acc, f32x16& acs, const float* __restrict__ wp, int lane) {
    const f16x8* q = reinterpret_cast<const f16x8*>(wp) + lane;
#define GAT8_MM(wpl, s, AC) AC = CL ? GATOR_MFMA_F16(b.p[0][s], w.p[wpl][s], AC) : GATOR_MFMA_F16(w.p[wpl][s], b.p[0][s], AC)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        GAT8_MM(1, s, acs);                                // w mid * a hi
        __builtin_amdgcn_sched_barrier(0);
        w.p[1][s] = q[(1 * 2 + s) * 64];
        __builtin_amdgcn_sched_barrier(0);
        GAT8_MM(0, s, acc);                                // w hi  * a hi
        __builtin_amdgcn_sched_barrier(0);
        w.p[0][s] = q[(0 * 2 + s) * 64];
        __builtin_amdgcn_sched_barrier(0);
    }
#undef GAT8_MM
}
// The byte-lo form of the four-product stream (round 5; default, GATOR_GAT8_LOBYTE=0 keeps H3): the same three fp16 planes, but the lo
// plane travels as ONE BYTE per weight.  lo = fp16(w - hi - mid) is zero or a single power of two between 2^-24 and 2^-10 (the
// residual of two 11-bit roundings of a 24-bit significand is at most one unit of w's last place), so the top byte of
// fp16(2^24 lo) -- sign, five exponent bits, two mantissa bits -- holds it exactly; the product wave rebuilds the fp16 value with
// two v_perm_b32 and two v_pk_mul_f16 (x 2^-24, exact: fp16 denormals are on) per four weights in the shadow of the MFMA before,
// and multiplies the SAME bits in the SAME order: results are bit for bit those of the H3 stream (tests/test_gpu_digest.py), a
// sixth of the stream's bytes never crosses the L2 -> CU path that bounds this kernel.  gat8_build_stream checks
// on the device, with this very expansion, that every weight's lo survives the round trip, and keeps the H3 stream if one does not.
struct H3B { f16x8 p[2][2]; uint4 lo; };     // [plane hi/mid][k-step] + 16 lo bytes (k-step 0: x y, k-step 1: z w): 20 VGPRs
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f16x8 lo_expand(unsigned d0, unsigned d1) {
    const f16x2_t sc = {(_Float16)5.9604644775390625e-08f, (_Float16)5.9604644775390625e-08f};      // 2^-24
    union { unsigned u[4]; f16x2_t h[4]; f16x8 v; } o;
    o.u[0] = __builtin_amdgcn_perm(d0, d0, 0x010c000cu);      // (byte 0, byte 1) -> the high bytes of two halves
    o.u[1] = __builtin_amdgcn_perm(d0, d0, 0x030c020cu);
    o.u[2] = __builtin_amdgcn_perm(d1, d1, 0x010c000cu);
    o.u[3] = __builtin_amdgcn_perm(d1, d1, 0x030c020cu);
#pragma unroll
    for (int i = 0; i < 4; ++i) o.h[i] = o.h[i] * sc;
    return o.v;
}
template <bool CL>
__device__ __forceinline__ void tile_mma_refill(H3B& w, const X2& b, f32x16& acc, f32x16& acs, const float* __restrict__ wp, int lane) {
    const f16x8* q = reinterpret_cast<const f16x8*>(wp) + lane;
#define GAT8_MM(wv, bpl, s, AC) AC = CL ? GATOR_MFMA_F16(b.p[bpl][s], (wv), AC) : GATOR_MFMA_F16((wv), b.p[bpl][s], AC)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        GAT8_MM(w.p[0][s], 1, s, acs);                     // w hi  * a lo
        const f16x8 lo = lo_expand(s == 0 ? w.lo.x : w.lo.z, s == 0 ? w.lo.y : w.lo.w);
        GAT8_MM(lo, 0, s, acs);                            // w lo  * a hi
        __builtin_amdgcn_sched_barrier(0);
        if (s == 1) w.lo = reinterpret_cast<const uint4*>(wp + 1024)[lane];
        __builtin_amdgcn_sched_barrier(0);
        GAT8_MM(w.p[1][s], 0, s, acs);                     // w mid * a hi
        __builtin_amdgcn_sched_barrier(0);
        w.p[1][s] = q[(1 * 2 + s) * 64];
        __builtin_amdgcn_sched_barrier(0);
        GAT8_MM(w.p[0][s], 0, s, acc);                     // w hi  * a hi
        __builtin_amdgcn_sched_barrier(0);
        w.p[0][s] = q[(0 * 2 + s) * 64];
        __builtin_amdgcn_sched_barrier(0);
    }
#undef GAT8_MM
}
// floats from one tile of a wave's stream to the next
template <class WT> struct StreamTile { static constexpr int floats = kTileX3; };
template <> struct StreamTile<H3B> { static constexpr int floats = kTileH3B; };
// one k-step of the two-plane product (x3_common.h: x2_mma does both): lo*hi | hi*lo | hi*hi
__device__ __forceinline__ f32x16 x2_mma_step(const X2& A, const X2& B, int s, f32x16 acc) {
    acc = GATOR_MFMA_F16(A.p[1][s], B.p[0][s], acc);
    acc = GATOR_MFMA_F16(A.p[0][s], B.p[1][s], acc);
    return GATOR_MFMA_F16(A.p[0][s], B.p[0][s], acc);
}
// operand / weight tile access of the two forms
__device__ __forceinline__ void ld_tile(X3& o, const float* p, int lane) { o = x3_load(p, lane); }
__device__ __forceinline__ void ld_tile(X2& o, const float* p, int lane) { o = x2_load(p, lane); }
__device__ __forceinline__ void ld_tile(H3& o, const float* p, int lane) { o = h3_load(p, lane); }
__device__ __forceinline__ void ld_tile(H3P2& o, const float* p, int lane) {
    const f16x8* q = reinterpret_cast<const f16x8*>(p) + lane;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) o.p[pl][s2] = q[(pl * 2 + s2) * 64];
}
__device__ __forceinline__ void ld_tile(H3B& o, const float* p, int lane) {
    const f16x8* q = reinterpret_cast<const f16x8*>(p) + lane;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) o.p[pl][s2] = q[(pl * 2 + s2) * 64];
    o.lo = reinterpret_cast<const uint4*>(p + 1024)[lane];
}
template <bool SWZ> __device__ __forceinline__ void ld_opnd(X3& o, const float* p, int lane) { o = x3_load(p, lane); }
template <bool SWZ> __device__ __forceinline__ void ld_opnd(X2& o, const float* p, int lane) { o = SWZ ? x2_load_swz(p, lane) : x2_load(p, lane); }

// ---- product wave: one unit = the K = 128 contraction of one 32-channel output block (4 weight tiles) --------------------------
// CL = false: weights as A operand -> T-layout accumulator (token on the lane); CL = true: activations as A -> C-layout.
// Slot indices are compile-time (S0 = tile offset of the unit inside its block, mod kNT).
// `b` = the unit's first operand tile, already requested: where that tile was complete two barriers ago (Y for k / v / h0 / h1, Y2
// for the later fc1 units, the hidden blocks for fc2) the caller reads it BEFORE the barrier that opens the step, so the matrix
// pipe does not idle through an LDS round trip after every barrier.
template <int S0, bool CL, bool SWZ = false, class WT, class OT>
__device__ __forceinline__ void unit4(WT (&W)[kNT], const float* __restrict__& wp, OT b, const float* o1, const float* o2, const float* o3,
                                      float* raw, int lane) {
    const float* ops[4] = {o1, o1, o2, o3};
    f32x16 acc = zero16(), acs = zero16();
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        OT bn = b;
        if (kb < 3) ld_opnd<SWZ>(bn, ops[kb + 1], lane);
        tile_mma_refill<CL>(W[(S0 + kb) % kNT], b, acc, acs, wp, lane);
        wp += StreamTile<WT>::floats;
        b = bn;
    }
    if constexpr (!std::is_same<WT, X3>::value) acc = acc + acs;
    store_block(raw, lane, acc);
}
// one tile: partial hop-2 linear (linears[1], 128 -> 16) over k block `w` of SB; C-layout
template <int S0, class OT, class WT>
__device__ __forceinline__ void unit1(WT (&W)[kNT], const float* __restrict__& wp, const float* o0, float* raw, int lane) {
    f32x16 acc = zero16(), acs = zero16();
    OT b;
    ld_tile(b, o0, lane);
    tile_mma_refill<true>(W[S0 % kNT], b, acc, acs, wp, lane);
    wp += StreamTile<WT>::floats;
    if constexpr (!std::is_same<WT, X3>::value) acc = acc + acs;
    store_block(raw, lane, acc);
}

// consume the kPadTiles dummy tiles behind a block: their slots get the tiles kNT positions further down (the next block's)
template <int S0, class WT>
__device__ __forceinline__ void skip_pad(WT (&W)[kNT], const float* __restrict__& wp, int lane) {
#pragma unroll
    for (int i = 0; i < kPadTiles; ++i) {
        ld_tile(W[(S0 + i) % kNT], wp, lane);
        wp += StreamTile<WT>::floats;
    }
}


// ---- epilogue (TAIL): lifter Linear(128J -> 3J) (GAT.py:151-152), MDR joint tokens and the three layers' cross-attention K / V
// operand tiles (MDR.py:130-134,37-38,65) -- formerly k_gat_lifter + k_gat_joint (gat_tail.hip: 5.7 + 16.4 us and two launch
// boundaries at B = 256, with ~2 MFLOP per sample between them; those launches stay for the sample-tiled encoder).  This workgroup owns
// the sample and ends with its feat on chip:
//   lifter       a 3J x 128J matrix-vector product: 444 KB (J = 17) of fp32 weights per sample that nobody on the CU can share, i.e.
//                a stream like the rest of this kernel.  All eight waves walk rows o = wave + 8 r as coalesced 1 KiB loads (lane l owns
//                k = 256 i + 4 l), exact fp32 FMAs against feat held in registers (read once from the T-layout tiles in X), one DPP
//                reduction per row.  (The product waves idle from barrier 20 of the last block on, but rows fetched ahead there would live beside the five
//                weight-stream tiles the block loop carries: 140 - 430 B per lane of scratch.  They stage the small operands instead.)  Fixed order: results do not depend on the batch.
//   joint tokens jf = Linear(133 -> 64)(cat(pose2d, pose3d / 1000, feat)) + pos_j: the 128 feat columns do not need pose3d, so product waves
//                0 / 1 (whose share of the rows is the smaller one) contract them behind their last rows; the five other columns are FMAs
//                once pose3d is known.  Then LayerNorm per layer and the 12 (layer, K | V, channel block) jobs, two on each of the other
//                six waves, their weight tiles requested behind the last lifter rows (same operand tiles out as k_gat_joint: two fp16
//                planes of 16 x value).  These token-wise linears run as every other one of the
//                default arithmetic does -- weights exact on three fp16 planes, activations on two, four partial products -- instead of
//                k_gat_joint's fp32-input MFMAs: measured with the MFMAs cut out (GAT8_TAIL_CUT), the 32-MFMA fp32 chains of the K / V jobs
//                alone were 6.5 us of the epilogue's 15.
// LDS: the operand tiles A | Bq are dead after barrier 20 of the last block.
constexpr int kTP = kA;                                 // the joint tokens without their pose3d columns (2 channel blocks, token on the lane)
constexpr int kTPosj = kTP + 2 * kTile;                 // pos_j tiles (2)
constexpr int kTVJ = kTPosj + 2 * kTile;                // per-channel vectors
constexpr int kTXO = kTVJ + 768;                        // pose3d of the sample (3J <= 64 floats)
enum { TVJ_JFB = 0, TVJ_JF5 = 64, TVJ_N1W = 384, TVJ_N1B = 576, TVJ_TOTAL = 768 };      // jf bias | jf columns 0..4 [5][64] | norm1 w / b [3 layers][64]
static_assert(kTXO + 64 <= kR, "the epilogue's LDS lives in the dead operand tiles");

#ifndef GAT8_TAIL_F32
#define GAT8_TAIL_F32 0              // 1: joint-token linear on the fp32-input MFMA (exact products), 2: the K / V jobs too  (A/B of the epilogue's arithmetic)
#endif
#ifndef GAT8_TAIL_CUT
#define GAT8_TAIL_CUT 0              // timing experiments only (tools/build_variant.py ... -DGAT8_TAIL_CUT=n): 1 no lifter loads, 2 no joint-token MFMAs, 4 no K / V MFMAs, 8 lifter only, 16 no L2 warm-up of the lifter weight
#endif
__device__ __forceinline__ float dpp_add(float s, int ctrl_tag) {
    const int u = __builtin_bit_cast(int, s);
    int m;
    if (ctrl_tag == 0) m = __builtin_amdgcn_update_dpp(u, u, 0xB1, 0xf, 0xf, false);            // quad_perm [1,0,3,2]
    else if (ctrl_tag == 1) m = __builtin_amdgcn_update_dpp(u, u, 0x4E, 0xf, 0xf, false);       // quad_perm [2,3,0,1]
    else if (ctrl_tag == 2) m = __builtin_amdgcn_update_dpp(u, u, 0x141, 0xf, 0xf, false);      // row_half_mirror
    else m = __builtin_amdgcn_update_dpp(u, u, 0x140, 0xf, 0xf, false);                         // row_mirror
    return s + __builtin_bit_cast(float, m);
}
// sum over the 64 lanes, the same value (and the same association) in every lane
__device__ __forceinline__ float wave_sum64(float s) {
    s = dpp_add(s, 0); s = dpp_add(s, 1); s = dpp_add(s, 2); s = dpp_add(s, 3);      // every lane: the total of its row of 16
    const int u = __builtin_bit_cast(int, s);
    const float t0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(u, 0)), t1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(u, 16)),
                t2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(u, 32)), t3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(u, 48));
    return (t0 + t1) + (t2 + t3);
}
// one row of the lifter weight: NI - 1 full 1 KiB chunks and a half chunk (K = 128 J = 256 (NI - 1) + 128 for J = 17, 19)
template <int NI>
struct LiftRow { f32x4 w[NI]; };
template <int NI>
__device__ __forceinline__ void lift_load(LiftRow<NI>& r, const float* __restrict__ wrow, int lane, bool live) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NI; ++i) r.w[i] = (!(GAT8_TAIL_CUT & 1) && live && (i < NI - 1 || lane < 32)) ? *reinterpret_cast<const f32x4*>(wrow + 256 * i + 4 * lane) : z;
}
template <int NI>
__device__ __forceinline__ float lift_dot(const LiftRow<NI>& r, const f32x4 (&f)[NI]) {
    f32x2 acc = {0.f, 0.f};                               // two chains of packed FMAs (components 0, 2 | 1, 3), fixed order
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        acc = pk_fma(f32x2{r.w[i][0], r.w[i][1]}, f32x2{f[i][0], f[i][1]}, acc);
        acc = pk_fma(f32x2{r.w[i][2], r.w[i][3]}, f32x2{f[i][2], f[i][3]}, acc);
    }
    return wave_sum64(acc[0] + acc[1]);
}
constexpr int kLiftPre = 2;          // rows per batch; two batches in registers (one being multiplied, one in flight)

#ifdef GATOR_DIAG
#define GAT8_TSTAMP(k) do { __builtin_amdgcn_sched_barrier(0); if (tl.tstamps && b == 32 && lane == 0) tl.tstamps[v8 * 8 + (k)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define GAT8_TSTAMP(k)
#endif
// HELPER: the calling wave is helper w (holds channel block w of feat in `y`); else product wave w.
template <int LR, bool HELPER>
__device__ __forceinline__ void gat8_tail(const Gat8Tail& tl, const float* pose2d, float* lds, int b, int J, int w, int lane) {
    constexpr int NI = LR == 10 ? 9 : 10;
    const int h = lane >> 5, tok = lane & 31, tkj = tok < J ? tok : 0;
    const int v8 = HELPER ? w : 4 + w;                        // row phase of this wave (phases 0 .. 2 own one row more: the helpers', who have nothing else here)
    const int K = kC * J, R = 3 * J;
    const float* X = lds + kXo;                               // feat as four T-layout tiles, zero for the tokens that do not exist (stored before barrier 22)
    float* P = lds + kTP;
    const float* VJ = lds + kTVJ;
    float* XO = lds + kTXO;
    const bool joint = tl.jkv != nullptr && !(GAT8_TAIL_CUT & 8);
    // who does what besides its lifter rows: product waves 0 / 1 the feat columns of the joint-token linear (channel block w, all of K = 128);
    // the other six waves two K / V jobs each
    const bool jf_wave = !HELPER && w < 2 && joint;
    // K / V jobs (layer li, k | v, channel block nb) = 2 pair + nb, pair = 2 li + kv: helper w both blocks of pair w, product wave w block
    // w & 1 of pair 4 + (w >> 1) -- three jobs on every SIMD, and a wave's jobs share their operand (one layer's LayerNorm)
    const int pair = HELPER ? w : 4 + (w >> 1), li = pair >> 1, kv = pair & 1;
    const int job0 = HELPER ? 2 * w : 8 + w, njob = HELPER ? 2 : 1;
    GAT8_TSTAMP(0);
    const float bias_r = (lane < 8 && v8 + 8 * lane < R) ? tl.lifter_b[v8 + 8 * lane] : 0.f;      // lane r: the bias of row v8 + 8 r
    const float* wbase = tl.lifter_w + (size_t)v8 * K;
    // rows r = 0 .. 7 of this wave in batches of kLiftPre, the next batch in flight while the current one is multiplied
    constexpr int NB = 8 / kLiftPre;
    LiftRow<NI> cur[kLiftPre], nxt[kLiftPre];
#pragma unroll
    for (int q = 0; q < kLiftPre; ++q) lift_load(cur[q], wbase + (size_t)(8 * q) * K, lane, v8 + 8 * q < R);
    GAT8_TSTAMP(1);
    f32x4 f[NI];                                              // feat[k], k = 256 i + 4 lane: token 2 i + (lane >> 5), channels 4 (lane & 31) ..
    {
        const int c = 4 * (lane & 31);
        const float* fx = X + (c >> 5) * kTile + (((c & 31) >> 3) * 64 + 32 * ((c >> 2) & 1)) * 4;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int j = 2 * i + (lane >> 5);
            f[i] = *reinterpret_cast<const f32x4*>(fx + (j < J ? j : 0) * 4);
        }
    }
    auto job_tile = [&](int job, int i) {
        const int li = job >> 2, kv = (job >> 1) & 1, nb = job & 1;
        return h3_load((kv ? tl.j_wv_h3[li] : tl.j_wk_h3[li]) + (size_t)(nb * 2 + i) * kTileX3, lane);
    };
    // the weights of what follows the lifter are requested behind the last batch of rows, so that they ride in the same stream
    H3 wt[4];                                                 // jf wave: its channel block's four k tiles; K / V wave: [job][k block]
    float p2x = 0.f, p2y = 0.f;
    float res = 0.f;
#pragma unroll
    for (int bt = 0; bt < NB; ++bt) {
        if (bt + 1 < NB) {
#pragma unroll
            for (int q = 0; q < kLiftPre; ++q) {
                const int r = (bt + 1) * kLiftPre + q;
                lift_load(nxt[q], wbase + (size_t)(8 * r) * K, lane, v8 + 8 * r < R);
            }
        }
        auto tail_tiles = [&](int i0) {       // two of the four weight tiles of what follows (requested while the last rows are multiplied)
            if (jf_wave) {
#pragma unroll
                for (int kb = i0; kb < i0 + 2; ++kb) wt[kb] = h3_load(tl.jf_h3 + (size_t)(w * 4 + kb) * kTileX3, lane);
            } else if (i0 < 2 * njob) {
#pragma unroll
                for (int i = i0; i < i0 + 2; ++i) wt[i] = job_tile(job0 + (i >> 1), i & 1);
            }
        };
        if (bt + 1 == NB && joint) {
            tail_tiles(0);
            p2x = pose2d[((size_t)b * J + tkj) * 2];
            p2y = pose2d[((size_t)b * J + tkj) * 2 + 1];
        }
#pragma unroll
        for (int q = 0; q < kLiftPre; ++q) {
            const int r = bt * kLiftPre + q;
            const float tot = lift_dot(cur[q], f);
            res = lane == r ? tot : res;
            if (bt + 1 == NB && q == 0 && joint) { __builtin_amdgcn_sched_barrier(0); tail_tiles(2); }      // (this row's registers are free now)
        }
        if (bt + 1 < NB) {
#pragma unroll
            for (int q = 0; q < kLiftPre; ++q) cur[q] = nxt[q];
        }
    }
    GAT8_TSTAMP(2);
    // x_out[o] = bias[o] + sum_k W[o][k] feat[k]
    if (lane < 8 && v8 + 8 * lane < R) {
        const float xo = res + bias_r;
        tl.x_out[(size_t)b * R + v8 + 8 * lane] = xo;
        XO[v8 + 8 * lane] = xo;
    }
    if (!joint) return;
    if (jf_wave) {
        // jf without its pose3d columns: bias + pos_j + feat columns (GATOR.py:19, MDR.py:130-134), channel block w, token on the lane
        f32x16 acc = zero16(), acs = zero16();
        const f32x16 init = chanvec_lds(VJ, TVJ_JFB + 32 * w, h) + load_block(lds + kTPosj + w * kTile, lane);
        if constexpr ((GAT8_TAIL_F32 & 1) != 0) {
            mma2_T(load_wtile(tl.jf_p, w * 4 + 0, lane), load_block(X, lane), acc, load_wtile(tl.jf_p, w * 4 + 1, lane), load_block(X + kTile, lane), acs);
            mma2_T(load_wtile(tl.jf_p, w * 4 + 2, lane), load_block(X + 2 * kTile, lane), acc, load_wtile(tl.jf_p, w * 4 + 3, lane), load_block(X + 3 * kTile, lane), acs);
            store_block(P + w * kTile, lane, (acc + acs) + init);
        } else {
            if (!(GAT8_TAIL_CUT & 2)) {
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    const X2 fx2 = x2_split(load_block(X + kb * kTile, lane) * 16.0f);
                    acs = h3_mma_wa_small(wt[kb], fx2, acs);
                    acc = h3_mma_wa_main(wt[kb], fx2, acc);
                }
            }
            store_block(P + w * kTile, lane, fma16(acc + acs, tl.jf_inv, init));
        }
        wt[0] = job_tile(job0, 0);                             // (its own K / V job's tiles, in flight across the barrier)
        wt[1] = job_tile(job0, 1);
    }
    GAT8_TSTAMP(3);
    __syncthreads();
    GAT8_TSTAMP(4);
    // jf = that + columns 0..4 (pose2d, pose3d / 1000)
    f32x16 jf[2];
    {
        const float pin[5] = {p2x, p2y, XO[tkj * 3] / 1000.f, XO[tkj * 3 + 1] / 1000.f, XO[tkj * 3 + 2] / 1000.f};
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            f32x16 acc = load_block(P + nb * kTile, lane);
#pragma unroll
            for (int i = 0; i < 5; ++i) acc += chanvec_lds(VJ, TVJ_JF5 + i * 64 + 32 * nb, h) * pin[i];
            jf[nb] = acc;
        }
    }
    auto rs = [&](const f32x16& p, const f32x16& q) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += p[r] + q[r];
        return s + xhalf(s);
    };
    const float mean = rs(jf[0], jf[1]) * (1.0f / 64.0f);
    const f32x16 d0 = jf[0] - mean, d1 = jf[1] - mean;
    const float rstd = 1.0f / sqrtf(rs(d0 * d0, d1 * d1) * (1.0f / 64.0f) + 1e-5f);
    GAT8_TSTAMP(5);
    // per LBF layer: k = wk(LN1(jf)), v = wv(LN1(jf)) as MFMA operand tiles (two fp16 planes of 16 x value: mdr_fused.hip, cross_attention_head_x2)
    const f32x16 fz0 = d0 * rstd * chanvec_lds(VJ, TVJ_N1W + 64 * li, h) + chanvec_lds(VJ, TVJ_N1B + 64 * li, h);
    const f32x16 fz1 = d1 * rstd * chanvec_lds(VJ, TVJ_N1W + 64 * li + 32, h) + chanvec_lds(VJ, TVJ_N1B + 64 * li + 32, h);
    const X2 z0 = x2_split(fz0 * 16.0f), z1 = x2_split(fz1 * 16.0f);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        if (q >= njob) break;
        const int nb = (job0 + q) & 1;
        float* out = tl.jkv + (((size_t)b * 3 + li) * 4 + kv * 2 + nb) * kTile;
        const H3 &k0 = wt[2 * q], &k1 = wt[2 * q + 1];
        f32x16 r0 = zero16(), r1 = zero16();                   // r1: the cross products of both k blocks, r0: the hi x hi ones (x3_common.h on the order)
        if constexpr ((GAT8_TAIL_F32 & 2) != 0) {
            const float* wp32 = kv ? tl.j_wv_p[li] : tl.j_wk_p[li];
            const WTile t0 = load_wtile(wp32, nb * 2, lane), t1 = load_wtile(wp32, nb * 2 + 1, lane);
            if (kv == 0) {
                mma2_T(t0, fz0, r0, t1, fz1, r1);
                r0 += r1;
                if (tok >= J) r0 = zero16();
            } else {
                mma2_C(t0, fz0, r0, t1, fz1, r1);
                r0 += r1;
#pragma unroll
                for (int r = 0; r < 16; ++r) r0[r] = (kap(r) + 4 * h < J) ? r0[r] : 0.f;
            }
            x2_store(out, lane, x2_split(r0 * 16.0f));
            continue;
        }
        if (kv == 0) {
            if (!(GAT8_TAIL_CUT & 4)) {
                r1 = h3_mma_wa_small(k1, z1, h3_mma_wa_small(k0, z0, r1));
                r0 = h3_mma_wa_main(k1, z1, h3_mma_wa_main(k0, z0, r0));
            }
            r0 += r1;
            if (tok >= J) r0 = zero16();                       // joints >= J: zero rows (masked in the softmax anyway)
        } else {
            if (!(GAT8_TAIL_CUT & 4)) {
                r1 = h3_mma_aw_small(z1, k1, h3_mma_aw_small(z0, k0, r1));
                r0 = h3_mma_aw_main(z1, k1, h3_mma_aw_main(z0, k0, r0));
            }
            r0 += r1;
#pragma unroll
            for (int r = 0; r < 16; ++r) r0[r] = (kap(r) + 4 * h < J) ? r0[r] : 0.f;
        }
        x2_store(out, lane, x2_split(r0 * (16.0f * tl.kv_inv)));
        GAT8_TSTAMP(6 + q);
    }
}

// H4: the token-wise products on four partial products (weights H3, operands X2; raw tiles carry 1 / a.lin_inv) instead of six
template <bool H4, int LR, bool H2 = false, bool LB = false, bool TAIL = false>
__global__ __launch_bounds__(512, 2) void k_gat8(const Gat8Args a) {
    static_assert(H4 || !TAIL, "the fused tail writes two-plane K / V tiles: four-product forms only");
    static_assert(H4 || !H2, "the one-plane form is a variant of the four-product form");
    static_assert((H4 && !H2) || !LB, "the byte-lo stream is a form of the four-product stream");
    typedef typename std::conditional<LB, H3B, typename std::conditional<H2, H3P2, typename std::conditional<H4, H3, X3>::type>::type>::type WT;
    constexpr int kWF = StreamTile<WT>::floats;                 // floats per tile of the weight stream
    typedef typename std::conditional<H4, X2, X3>::type OT;
    const float inv = H4 ? a.lin_inv : 1.0f;
    // operand tile of a true-scale register tile; pick-up of a raw product tile with what is added to it
    // Token lanes that do not exist (15 of a tile's 32 at J = 17) store ZERO operands: no live value changes (a token's row never
    // meets another token's except through the masked J x J operators), but the MFMAs stop toggling on 47 % of their columns and
    // the chip holds a higher clock -- k_gat8 163.7 -> 159.7 us on one box, bit-identical results (DESIGN.md 4c on why data matters).
    const float op16 = ((threadIdx.x & 31) < a.J) ? 16.0f : 0.0f;
    auto st_opnd = [&](float* dst, int lane_, const f32x16& v) {
        if constexpr (H4) x2_store(dst, lane_, x2_split(v * op16)); else x3_store(dst, lane_, x3_split(v));
    };
    auto pick = [&](const float* raw, int lane_, const f32x16& add) {
        if constexpr (H4) return fma16(load_block(raw, lane_), inv, add); else return load_block(raw, lane_) + add;
    };
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* A = lds + kA;
    float* Bq = lds + kBq;
    float* X = lds + kXo;
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, h = lane >> 5, J = a.J;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const bool is_prod = wave < 4;
    const int w = wave & 3;                              // channel block of this wave (either role).  Waves i and i + 4 share a SIMD: every
                                                         // SIMD hosts one product and one helper wave (both product waves of a pair on one
                                                         // SIMD, helpers on the other two: 180 us instead of 162, measured in round 4)
    float* R0w = lds + kR + w * kTile;
    float* R1w = lds + kR + (4 + w) * kTile;
    if constexpr (TAIL) {      // the persistent MDR launch's counter blocks (tickets, error flag, completion counts), dealt over the workgroups:
        if (a.tl.mdr_ctr)      // the previous forward's launches are complete (stream order), this forward's start after this kernel
            for (size_t i = (size_t)b * 512 + t; i < mdr_ctr_words(a.tl.ctr_B); i += (size_t)a.B * 512) a.tl.mdr_ctr[i] = 0u;
    }

    // ---------------- embedding: GraphLinear(2->64) . GroupNorm(4,64) . GELU . GraphLinear(64->128) + pos (GAT.py:135-144)
    {
        float* hbuf = Bq;                 // [64][32]
        float* gt = Bq + 2 * kTile;       // 2 T-layout tiles
        float* stat = Bq + 4 * kTile;     // [4][2]
        const float* p = a.pose2d + (size_t)b * J * 2;
        for (int e = t; e < 64 * 32; e += 512) {
            const int c = e >> 5, j = e & 31;
            hbuf[e] = j < J ? a.gl0_W[c * 2] * p[j * 2] + a.gl0_W[c * 2 + 1] * p[j * 2 + 1] + a.gl0_b[c] : 0.f;
        }
        __syncthreads();
        if (!is_prod) {   // helper w -> GroupNorm group w (16 channels x J tokens), two-pass
            float s = 0.f;
            for (int e = lane; e < 16 * 32; e += 64) s += ((e & 31) < J) ? hbuf[w * 512 + e] : 0.f;
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            const float mean = s / (16.0f * J);
            float q = 0.f;
            for (int e = lane; e < 16 * 32; e += 64) {
                const float d = hbuf[w * 512 + e] - mean;
                q += ((e & 31) < J) ? d * d : 0.f;
            }
            for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
            if (lane == 0) { stat[w * 2] = mean; stat[w * 2 + 1] = 1.0f / sqrtf(q / (16.0f * J) + 1e-5f); }
        }
        __syncthreads();
        // GroupNorm affine + GELU as two T-layout operand tiles gt[kb][g][lane][j] <-> token lane&31, channel 32kb+8g+4h+j
        for (int e = t; e < 2 * kTile; e += 512) {
            const int j4 = e & 3, ln = (e >> 2) & 63, g = (e >> 8) & 3, kb = e >> 10;
            const int tok = ln & 31, c = 32 * kb + 8 * g + 4 * (ln >> 5) + j4;
            gt[e] = tok < J ? gelu_f((hbuf[c * 32 + tok] - stat[(c >> 4) * 2]) * stat[(c >> 4) * 2 + 1] * a.gn_w[c] + a.gn_b[c]) : 0.f;
        }
        __syncthreads();
        if (!is_prod) {   // GraphLinear(64->128) on the fp32-input MFMA: helper w -> channel block w; + folded position tiles
            f32x16 acc = load_chanvec_T(a.gl3_b, 32 * w, h) + load_block(a.posT + (size_t)w * kTile, lane), ac1 = zero16();
            mma2_T(load_wtile(a.gl3_p, w * 2 + 0, lane), load_block(gt, lane), acc, load_wtile(a.gl3_p, w * 2 + 1, lane),
                   load_block(gt + kTile, lane), ac1);
            store_block(X + w * kTile, lane, acc + ac1);
        }
        __syncthreads();
    }

    // From here on the two roles run in DISJOINT branches and meet only at workgroup barriers: s_barrier counts arrivals, it does not
    // care which branch a wave arrives from, so this is well defined exactly as long as both branches execute the SAME NUMBER of
    // barriers in the same order -- barrier 0, then 1 .. 22 per block, each written with the numbered macro.  tests/test_host_cpu.py parses this file and fails if the
    // two sequences differ (a barrier only one role executes hangs the GPU); keep that test in the CPU suite.
    if (is_prod) {
        // =========================================== product waves ===========================================================
        WT W[kNT];                                                          // the head of the weight stream (held back until here: five tiles
        const float* __restrict__ wp = a.wstream + (size_t)w * kWaveTiles * kWF;         // live across the embedding would spill)
#pragma unroll
        for (int s = 0; s < kNT; ++s) { ld_tile(W[s], wp, lane); wp += kWF; }
        asm volatile("" ::: "memory");
#ifdef GATOR_DIAG
        unsigned long long* st_lds = reinterpret_cast<unsigned long long*>(lds + kGat8LdsFloats);
        unsigned long long* st_out = (a.stamps && b == 0 && w == 0 && lane == 0) ? st_lds : nullptr;
        if (st_out) for (int i_ = 0; i_ < kDepth * 23 * 2; ++i_) st_lds[i_] = 0;
        unsigned long long st_last = __builtin_amdgcn_s_memtime();
        int bi_ = 0;
#endif
        GAT8_BAR(0);                                                        // helpers: Y = LN1(x) of block 0
#pragma unroll 1
        for (int bi = 0; bi < kDepth; ++bi) {
#ifdef GATOR_DIAG
            bi_ = bi;
#endif
            const float *Y0 = A, *Y1 = A + kTileX3, *Y2 = A + 2 * kTileX3, *Y3 = A + 3 * kTileX3;
            OT pre;
            ld_tile(pre, Y0, lane);
            unit4<OFF_Q % kNT, false>(W, wp, pre, Y1, Y2, Y3, R0w, lane);             // q  (T)
            ld_tile(pre, Y0, lane);
            GAT8_BAR(1);
            unit4<OFF_K % kNT, false>(W, wp, pre, Y1, Y2, Y3, R1w, lane);             // k  (T)
            ld_tile(pre, Y0, lane);
            GAT8_BAR(2);
            unit4<OFF_V % kNT, true>(W, wp, pre, Y1, Y2, Y3, R0w, lane);              // v  (C)
            ld_tile(pre, Y0, lane);
            GAT8_BAR(3);
            unit4<OFF_H0 % kNT, false>(W, wp, pre, Y1, Y2, Y3, R1w, lane);             // h0 = y W[0]  (T: its MGCN term is token-wise)
            ld_tile(pre, Y0, lane);
            GAT8_BAR(4);
            unit4<OFF_H1 % kNT, true>(W, wp, pre, Y1, Y2, Y3, R0w, lane);              // h1 = y W[1]  (C)
            GAT8_BAR(5);
            ld_tile(pre, Bq, lane);
            unit4<OFF_PROJ % kNT, false>(W, wp, pre, Bq + kTileX3, Bq + 2 * kTileX3, Bq + 3 * kTileX3, R1w, lane);                   // proj(AT)
            GAT8_BAR(6);
            GAT8_BAR(7);                                                    // helpers: SB = proj + attention bias + MGCN
            ld_tile(pre, Bq + 4 * kTileX3, lane);
            unit4<OFF_LIN0 % kNT, true>(W, wp, pre, Bq + 5 * kTileX3, Bq + 6 * kTileX3, Bq + 7 * kTileX3, R0w, lane);                // linears[0](SB)  (C)
            unit1<OFF_LIN1 % kNT, OT>(W, wp, Bq + (4 + w) * kTileX3, X + w * kTile, lane);                                              // linears[1], k block w
            GAT8_BAR(8);
            GAT8_BAR(9);                                                    // helpers: hop aggregations -> FB
            ld_tile(pre, Bq + 8 * kTileX3, lane);
            unit4<OFF_BACK % kNT, false>(W, wp, pre, Bq + 9 * kTileX3, Bq + 10 * kTileX3, Bq + 11 * kTileX3, R1w, lane);             // linearback(FB), k < 128
            GAT8_BAR(10);
            GAT8_BAR(11);                                                   // helpers: residual
            GAT8_BAR(12);                                                   // helpers: Y2 = LN2(x)
            ld_tile(pre, Y0, lane);
            unit4<(OFF_FC1 + 0) % kNT, H4>(W, wp, pre, Y1, Y2, Y3, R0w, lane);             // fc1, hidden block 4w
            ld_tile(pre, Y0, lane);
            GAT8_BAR(13);
            unit4<(OFF_FC1 + 4) % kNT, H4>(W, wp, pre, Y1, Y2, Y3, R1w, lane);             //      4w + 1
            ld_tile(pre, Y0, lane);
            GAT8_BAR(14);
            unit4<(OFF_FC1 + 8) % kNT, H4>(W, wp, pre, Y1, Y2, Y3, R0w, lane);             //      4w + 2
            ld_tile(pre, Y0, lane);
            GAT8_BAR(15);
            unit4<(OFF_FC1 + 12) % kNT, H4>(W, wp, pre, Y1, Y2, Y3, R1w, lane);             //      4w + 3
            ld_opnd<H4>(pre, Bq, lane);                                    // (hidden blocks 4w' were complete at barrier 14)
            GAT8_BAR(16);
            // fc2: unit u contracts over hidden blocks {4w' + u}: tiles 3w' + u of B for u < 3, tile w' of A for u = 3
            unit4<(OFF_FC2 + 0) % kNT, false, H4>(W, wp, pre, Bq + 3 * kTileX3, Bq + 6 * kTileX3, Bq + 9 * kTileX3, R0w, lane);
            ld_opnd<H4>(pre, Bq + 1 * kTileX3, lane);
            GAT8_BAR(17);
            unit4<(OFF_FC2 + 4) % kNT, false, H4>(W, wp, pre, Bq + 4 * kTileX3, Bq + 7 * kTileX3, Bq + 10 * kTileX3, R1w, lane);
            ld_opnd<H4>(pre, Bq + 2 * kTileX3, lane);
            GAT8_BAR(18);
            unit4<(OFF_FC2 + 8) % kNT, false, H4>(W, wp, pre, Bq + 5 * kTileX3, Bq + 8 * kTileX3, Bq + 11 * kTileX3, R0w, lane);
            ld_opnd<H4>(pre, Y0, lane);                                    // (hidden blocks 4w' + 3, complete at barrier 17)
            GAT8_BAR(19);
            unit4<(OFF_FC2 + 12) % kNT, false, H4>(W, wp, pre, Y1, Y2, Y3, R1w, lane);
            skip_pad<(OFF_FC2 + 16) % kNT>(W, wp, lane);                          // the block's dummy tiles: refill their slots, no product
            GAT8_BAR(20);
            if constexpr (TAIL) {
                if (bi + 1 == kDepth) {      // idle from here on: stage the epilogue's vectors and pos_j tiles (the operand tiles are dead), fetch the first lifter rows
                    if (a.tl.jkv) {
                        const Gat8Tail& tl = a.tl;
                        if (t < TVJ_TOTAL / 4) {
                            const int off = 4 * t;
                            const float* src = off < TVJ_JF5 ? tl.jf_b + off : off < TVJ_N1W ? tl.jf5 + (off - TVJ_JF5)
                                               : off < TVJ_N1B ? tl.j_n1w[(off - TVJ_N1W) >> 6] + (off & 63) : tl.j_n1b[(off - TVJ_N1B) >> 6] + (off & 63);
                            reinterpret_cast<f32x4*>(lds + kTVJ)[t] = *reinterpret_cast<const f32x4*>(src);
                        }
                        for (int e = t; e < 2 * kTile / 4; e += 256)
                            reinterpret_cast<f32x4*>(lds + kTPosj)[e] = reinterpret_cast<const f32x4*>(tl.posj_T)[e];
                    }
                }
            }
            GAT8_BAR(21);                                                   // helpers: residual
            GAT8_BAR(22);                                                   // helpers: Y = LN1(x) of the next block | final norm
        }
        GAT8_STAMPS_OUT(0, kDepth * 23 * 2);
        if constexpr (TAIL) gat8_tail<LR, false>(a.tl, a.pose2d, lds, b, J, w, lane);
        return;
    }

    // =============================================== helper waves ===============================================================
    const int tok = lane & 31;
    const HidAddr hid = hid_addr(lane);
#ifdef GATOR_DIAG
    if (a.dbg & 1) {
        for (int n = 0; n < 1 + 22 * kDepth; ++n) __syncthreads();
        return;
    }
    if (a.dbg & 16) {      // barriers + the L2 warm-up of the weight streams only: what the product waves cost with warm weights
        __syncthreads();
        for (int bi = 0; bi < kDepth; ++bi)
            for (int n = 1; n <= 22; ++n) {
                if ((n == 2 || n == 19) && bi + 1 < kDepth && a.pf_loads > 0) {
                    const char* wsrc = reinterpret_cast<const char*>(a.wstream + ((size_t)w * kWaveTiles + (size_t)(bi + 1) * kBlkTiles) * kWF);
                    const int share = a.pf_loads * 8192, first = ((b >> 3) % a.pf_n) * share, lim = kBlkTiles * kWF * 4 - 128;
                    const int i0 = n == 2 ? 0 : (a.pf_loads + 1) / 2, i1 = n == 2 ? (a.pf_loads + 1) / 2 : a.pf_loads;
                    for (int i = i0; i < i1; ++i)
                        glds4(reinterpret_cast<const float*>(wsrc + min(first + i * 8192 + lane * 128, lim)), lds + kDummy + w * 256);
                }
                __syncthreads();
            }
        return;
    }
    unsigned long long* st_lds = reinterpret_cast<unsigned long long*>(lds + kGat8LdsFloats);
    unsigned long long* st_out = (a.stamps && b == 0 && w == 0 && lane == 0) ? st_lds + (size_t)kDepth * 23 * 2 : nullptr;
    if (st_out) for (int i_ = 0; i_ < kDepth * 23 * 2 + kDepth * 8; ++i_) st_out[i_] = 0;
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
    int bi_ = 0;
#endif
    f32x16 xw = load_block(X + w * kTile, lane);          // this wave's channel block of the residual stream
    {
        const float* vec = a.blk[0].vecs;
        const f32x16 y = ln_own(X, xw, load_chanvec_T(vec, V_N1W + 32 * w, h), load_chanvec_T(vec, V_N1B + 32 * w, h), lane);
        st_opnd(A + w * kTileX3, lane, y);
    }
    GAT8_BAR(0);
#pragma unroll 1
    for (int bi = 0; bi < kDepth; ++bi) {
#ifdef GATOR_DIAG
        bi_ = bi;
#endif
        const Gat8Blk& kb = a.blk[bi];
        const float* vec = kb.vecs;
        // L2 warm-up.  Between two launches of this kernel the rest of the forward moves ~1 GB, so the weights start in HBM; all
        // workgroups of an XCD walk them in step, so every tile would be an HBM-latency miss for all of them at once (175 -> 195 us
        // without it).  Each workgroup pulls its 1/n-th of the NEXT block's four streams into its XCD's L2 one block ahead, and this
        // helper's share of that block's tables.  One dword per 128-byte line: a wave instruction touches 64 lines = 8 KiB and costs the
        // issuing wave ~350 cycles, so the touches sit in the steps where the helper has slack against its product wave (2 and 18).
        const bool warm = bi + 1 < kDepth && a.pf_loads > 0;
        const float* dummy = lds + kDummy + w * 256;
        const Gat8Blk& nx = a.blk[bi + 1 < kDepth ? bi + 1 : bi];
        auto warm_weights = [&](int i0, int i1) {
            const char* wsrc = reinterpret_cast<const char*>(a.wstream + ((size_t)w * kWaveTiles + (size_t)(bi + 1) * kBlkTiles) * kWF);
            const int share = a.pf_loads * 8192, first = ((b >> 3) % a.pf_n) * share, lim = kBlkTiles * kWF * 4 - 128;
            for (int i = i0; i < i1; ++i)
                glds4(reinterpret_cast<const float*>(wsrc + min(first + i * 8192 + lane * 128, lim)), dummy);
        };
        // ---- step 1: constants of the attention (nothing to pick up yet)
        const f32x16 bq = load_chanvec_T(vec, V_QKVB + 32 * w, h), bk = load_chanvec_T(vec, V_QKVB + 128 + 32 * w, h);
        const float vb = vec[V_QKVB + 256 + 32 * w + tok];
        const f32x16 ba = load_block(a.biasT + (size_t)(2 * w) * kTile, lane), bb = load_block(a.biasT + (size_t)(2 * w + 1) * kTile, lane);
        GAT8_BAR(1);
        // ---- step 2: q
        const f32x16 q = pick(R0w, lane, bq);
        X2 qx;                                                 // H4: q on two planes already here (the helper has slack in this step)
        if constexpr (H4) qx = x2_split(q * op16);            // (op16: zero for the token lanes that do not exist)
        if (warm) warm_weights(0, (a.pf_loads + 1) / 2);
        if constexpr (TAIL) {
            // The epilogue's weights are HBM-cold like the streams (the lifter's 444 KB at J = 17, the joint-token linear's and the three
            // layers' wk / wv H3 tiles): every workgroup of the XCD would miss on them together.  One block ahead -- the last block has no
            // next block to warm -- this helper pulls its share of each region into the XCD's L2, one dword per 128-byte line.
            if (!(GAT8_TAIL_CUT & 16) && bi + 1 == kDepth && a.tl.warm_n > 0) {
                const int idx = ((b >> 3) % a.tl.warm_n) * 4 + w, parts = a.tl.warm_n * 4;
                auto touch = [&](const float* base, int bytes) {
                    const int share = ((bytes / parts + 127) / 128) * 128;
                    for (int off = lane * 128; off < share; off += 8192)
                        glds4(reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + min(idx * share + off, bytes - 128)), dummy);
                };
                touch(a.tl.lifter_w, 3 * J * kC * J * 4);
                if (a.tl.jkv) {
                    touch(a.tl.jf_h3, 8 * kTileX3 * 4);
#pragma unroll 1
                    for (int li = 0; li < 3; ++li) touch(a.tl.j_wk_h3[li], 8 * kTileX3 * 4);      // wk | wv: eight contiguous tiles of the layer's grid
                }
            }
        }
        GAT8_BAR(2);
        // ---- step 3: k; scores and softmax of heads 2w, 2w+1 (modules.py:121-133): S^T[key][query], query on the lane
        f32x16 sa = zero16(), sb = zero16();
        {
            const f32x16 k = pick(R1w, lane, bk);
            float sscale = 0.25f;
            if constexpr (H4) {      // q, k on two fp16 planes of 16 x value: a head's 16 channels are exactly one k-step (registers 0..7 | 8..15)
                const X2 kx = x2_split(k * op16);
                sa = GATOR_MFMA_F16(kx.p[1][0], qx.p[0][0], sa);
                sb = GATOR_MFMA_F16(kx.p[1][1], qx.p[0][1], sb);
                sa = GATOR_MFMA_F16(kx.p[0][0], qx.p[1][0], sa);
                sb = GATOR_MFMA_F16(kx.p[0][1], qx.p[1][1], sb);
                sa = GATOR_MFMA_F16(kx.p[0][0], qx.p[0][0], sa);
                sb = GATOR_MFMA_F16(kx.p[0][1], qx.p[0][1], sb);
                sscale = 0.25f / 256.0f;
            } else {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    sa = GATOR_MFMA(k[r], q[r], sa);                                // head 2w:   channels 0..15 of the block
                    sb = GATOR_MFMA(k[r + 8], q[r + 8], sb);                        // head 2w+1: channels 16..31
                }
            }
            // rows = keys: only registers r < LR can hold a key that exists (the others' probability is exactly 0)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (r < LR) {
                    const bool ok = kap(r) + 4 * h < J;
                    sa[r] = ok ? (sa[r] * sscale + ba[r]) * kLog2e8 : -1e30f;        // q k^T * head_dim**-0.5 + hop/path bias
                    sb[r] = ok ? (sb[r] * sscale + bb[r]) * kLog2e8 : -1e30f;
                } else {
                    sa[r] = 0.f;
                    sb[r] = 0.f;
                }
            }
            float ma = sa[0], mb = sb[0], la, lb;
            {   // row maxima; sums as trees (four independent partials each)
#pragma unroll
                for (int r = 1; r < LR; ++r) { ma = fmaxf(ma, sa[r]); mb = fmaxf(mb, sb[r]); }
                ma = fmaxf(ma, xhalf(ma));
                mb = fmaxf(mb, xhalf(mb));
#pragma unroll
                for (int r = 0; r < LR; ++r) {
                    sa[r] = __builtin_amdgcn_exp2f(sa[r] - ma);
                    sb[r] = __builtin_amdgcn_exp2f(sb[r] - mb);
                }
                float pa[4], pb[4];
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    pa[q4] = tree4<LR>(q4, [&](int r) { return sa[r]; });
                    pb[q4] = tree4<LR>(q4, [&](int r) { return sb[r]; });
                }
                la = LR > 12 ? (pa[0] + pa[1]) + (pa[2] + pa[3]) : LR > 8 ? (pa[0] + pa[1]) + pa[2] : pa[0] + pa[1];
                lb = LR > 12 ? (pb[0] + pb[1]) + (pb[2] + pb[3]) : LR > 8 ? (pb[0] + pb[1]) + pb[2] : pb[0] + pb[1];
            }
            la += xhalf(la);
            lb += xhalf(lb);
            const float ia = 1.0f / la, ib = 1.0f / lb;
#pragma unroll
            for (int r = 0; r < LR; ++r) { sa[r] = sa[r] * ia; sb[r] = sb[r] * ib; }
        }
        GAT8_BAR(3);
        // ---- steps 4, 5: v; P.V (both heads, rows 0..15 <- head 2w, rows 16..31 <- head 2w+1); pick up h0 in between
        f32x16 O = zero16(), Ob = zero16();
        const bool lo = tok < 16;
        {
            f32x16 v = load_block(R0w, lane);
            if constexpr (H4) v = v * inv;
            X2 va, vbx, pa, pb;      // H4: V of head 2w (channel lanes 0..15) / head 2w+1 (16..31) and the two probability tiles on two planes
            if constexpr (H4) {
                const X2 vx = x2_split_rows<LR>((v + vb) * 16.0f);      // rows = keys
                const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) { va.p[pl][ks] = lo ? vx.p[pl][ks] : z; vbx.p[pl][ks] = lo ? z : vx.p[pl][ks]; }
                pa = x2_split_rows<LR>(sa * (4.0f * op16));          // x 64, or zero for a query lane that does not exist
                pb = x2_split_rows<LR>(sb * (4.0f * op16));
                O = x2_mma_step(va, pa, 0, O);
                Ob = x2_mma_step(vbx, pb, 0, Ob);
            } else {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float vv = v[r] + vb;
                    O = GATOR_MFMA(lo ? vv : 0.f, sa[r], O);
                    Ob = GATOR_MFMA(lo ? 0.f : vv, sb[r], Ob);
                }
            }
            GAT8_BAR(4);
            const f32x16 mdt = load_block(kb.mdT + (size_t)w * kTile, lane);        // (constants are requested about one step before their use)
            f32x16 h0 = load_block(R1w, lane);
            if constexpr (H4) h0 = h0 * inv;
            if constexpr (H4) {
                O = x2_mma_step(va, pa, 1, O);
                Ob = x2_mma_step(vbx, pb, 1, Ob);
                x2_store(Bq + w * kTileX3, lane, x2_split((O + Ob) * (op16 * (1.0f / 1024.0f))));      // AT[w]: O carries 16 x 64
            } else {
#pragma unroll
                for (int r = 8; r < 16; ++r) {
                    const float vv = v[r] + vb;
                    O = GATOR_MFMA(lo ? vv : 0.f, sa[r], O);
                    Ob = GATOR_MFMA(lo ? 0.f : vv, sb[r], Ob);
                }
                st_opnd(Bq + w * kTileX3, lane, O + Ob);                            // AT[w]
            }
            O = h0 * mdt;                                                          // diag(A)[t] * M[t][n] * h0[t][n]  (O re-used)
        }
        const f32x16 mct = load_block(kb.mc + (size_t)w * kTile, lane), aoff = load_block(kb.aoffT, lane);
        const f32x16 bg = load_chanvec_T(vec, V_GCNB + 32 * w, h);
        GAT8_BAR(5);
        // ---- step 6: h1; MGCN (modules.py:243-255): sum_j (M.h1)[j][n] Aoff[t][j] as one MFMA product + the token-wise term
        f32x16 g_out;
        {
            f32x16 h1 = load_block(R0w, lane);
            if constexpr (H4) h1 = h1 * inv;
            h1 = h1 * mct;
            g_out = dot16_rows<LR>(h1, aoff, bg) + O;          // k = token j: Aoff[t][j] is zero for the rows that do not exist
        }
        const f32x16 bp = load_chanvec_T(vec, V_PROJB + 32 * w, h);
        GAT8_BAR(6);
        // ---- step 7: SB = proj(attention) + bias + MGCN
        {
            const f32x16 acc = pick(R1w, lane, bp);
            st_opnd(Bq + (4 + w) * kTileX3, lane, acc + g_out);
        }
        GAT8_BAR(7);
        // constants of the X_Feat steps
        const f32x16 m1 = load_block(a.m1T, lane), m2 = load_block(a.m2T, lane);
        const float b0 = vec[V_LIN0B + 32 * w + tok];
        f32x16 f1 = load_block(kb.f1b, lane);                                      // rowsum(m2)[t] * linears[1].bias[n]
        GAT8_BAR(8);
        // ---- step 9: X_Feat (modules.py:158-177): hop<=1 aggregation of linears[0], hop==2 aggregation of linears[1]
        {
            const f32x16 u0 = pick(R0w, lane, f32x16(b0));
            f32x16 u1 = (load_block(X, lane) + load_block(X + kTile, lane)) + (load_block(X + 2 * kTile, lane) + load_block(X + 3 * kTile, lane));
            if constexpr (H4) {
                // the hop masks are 0 / 1: one exact fp16 plane; the aggregated tiles on two planes like every other operand of this
                // form -> 2 x 4 fp16 MFMAs (256 cycles) instead of 32 fp32-input ones (2 048) in a step the product waves wait through
                const X2 u0x = x2_split_rows<LR>(u0 * 16.0f), u1x = x2_split_rows<LR>(u1 * (16.0f * inv));      // rows = the aggregated token
                f32x16 f0 = zero16(), f1a = zero16();
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    f16x8 m1h, m2h;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const bool live = 8 * s + j < LR;
                        m1h[j] = live ? (_Float16)m1[8 * s + j] : (_Float16)0.f;
                        m2h[j] = live ? (_Float16)m2[8 * s + j] : (_Float16)0.f;
                    }
                    f0 = GATOR_MFMA_F16(u0x.p[1][s], m1h, f0);
                    f1a = GATOR_MFMA_F16(u1x.p[1][s], m2h, f1a);
                    f0 = GATOR_MFMA_F16(u0x.p[0][s], m1h, f0);
                    f1a = GATOR_MFMA_F16(u1x.p[0][s], m2h, f1a);
                }
                x2_store(Bq + (8 + w) * kTileX3, lane, x2_split(f0));              // FB[w]: f0 already carries the operand scale 16
                f1 = fma16(f1a, 1.0f / 16.0f, f1);
            } else {
                f32x16 f0 = zero16(), f1a = zero16();
                dot16x2(u0, m1, f0, u1, m2, f1a);
                st_opnd(Bq + (8 + w) * kTileX3, lane, f0);                        // FB[w]
                f1 += f1a;
            }
        }
        const f32x16 bback = load_chanvec_T(vec, V_BACKB + 32 * w, h);
        f32x4 wb40, wb41;                                                          // linearback's k block 4 (fp32 tile): only k < 16 is live
        {
            const f32x4* p = reinterpret_cast<const f32x4*>(kb.back32) + ((size_t)(w * 5 + 4) * 4) * 64 + lane;
            wb40 = p[0];
            wb41 = p[64];
        }
        GAT8_BAR(9);
        // ---- step 10: linearback's k tail (channels 128..143 <- the hop-2 features), exact fp32 products
        f32x16 tl = bback, tl2 = zero16();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            tl = GATOR_MFMA(wb40[j], f1[j], tl);
            tl2 = GATOR_MFMA(wb41[j], f1[4 + j], tl2);
        }
        const f32x16 n2w = load_chanvec_T(vec, V_N2W + 32 * w, h), n2b = load_chanvec_T(vec, V_N2B + 32 * w, h);
        GAT8_BAR(10);
        // ---- step 11: residual
        xw += pick(R1w, lane, tl) + tl2;
        tile_stats(xw, lds + kStat + w * 64, lane);
        GAT8_BAR(11);
        // ---- step 12: Y2 = LN2(x)
        st_opnd(A + w * kTileX3, lane, ln_stats(lds + kStat, xw, n2w, n2b, lane));
        GAT8_BAR(12);
        // ---- steps 13-17: MLP hidden blocks 4w + j: bias, GELU, split (modules.py:188-196).  H4: the raw tile is in C layout (channel
        // on the lane: ONE bias value per lane; token rows r < LR in the registers), see hid_store; else T layout, all 16 registers.
        struct FcBias { float s; f32x16 v; };
        auto fc1_bias = [&](int j) {
            FcBias o;
            if constexpr (H4) {
                o.v = zero16();
#ifdef GATOR_DIAG
                if (a.dbg & 8) { o.s = 0.f; return o; }
#endif
                o.s = vec[V_FC1B + 32 * (4 * w + j) + (lane & 31)];
            }
            else { o.s = 0.f; o.v = load_chanvec_T(vec, V_FC1B + 32 * (4 * w + j), h); }
            return o;
        };
        auto hidden = [&](const float* raw, float* dst, const FcBias& fb) {
            if constexpr (H4) {
                f32x16 hd = fma16(load_block_rows<LR>(raw, lane), inv, f32x16(fb.s));
                GAT8_SUB(0);
#ifdef GATOR_DIAG
                if (!(a.dbg & 2))
#endif
                gelu_pairs<LR / 2>(hd);
                GAT8_SUB(1);
#ifdef GATOR_DIAG
                if (!(a.dbg & 4))
#endif
                hid_store<LR>(dst, hid, hd);
                GAT8_SUB(3);
            } else {
                f32x16 hd = pick(raw, lane, fb.v);
                gelu_tile8(hd);
                st_opnd(dst, lane, hd);
            }
        };
        FcBias fb0 = fc1_bias(0), fb1 = fc1_bias(1);
        GAT8_BAR(13);
        hidden(R0w, Bq + (3 * w + 0) * kTileX3, fb0);
        fb0 = fc1_bias(2);
        GAT8_BAR(14);
        hidden(R1w, Bq + (3 * w + 1) * kTileX3, fb1);
        fb1 = fc1_bias(3);
        GAT8_BAR(15);
        hidden(R0w, Bq + (3 * w + 2) * kTileX3, fb0);
        GAT8_BAR(16);
        hidden(R1w, A + w * kTileX3, fb1);                                             // (Y2 is dead: fc1 finished before barrier 16)
        const f32x16 bfc2 = load_chanvec_T(vec, V_FC2B + 32 * w, h);
        GAT8_BAR(17);
        // ---- steps 18-21: the four partial sums of fc2, residual
        f32x16 c01 = pick(R0w, lane, bfc2);
        GAT8_BAR(18);
        c01 = pick(R1w, lane, c01);
        if (warm) warm_weights((a.pf_loads + 1) / 2, a.pf_loads);
        if (warm) {                                  // this helper's share of the next block's tables and vectors, one lane per line
            const int l32 = lane & 31;
            glds4(h == 0 ? nx.mdT + (size_t)w * kTile + l32 * 32 : nx.mc + (size_t)w * kTile + l32 * 32, dummy);
            if (w == 0) glds4(nx.vecs + lane * 32, dummy);
            if (w == 1) glds4(h == 0 ? nx.aoffT + l32 * 32 : nx.f1b + l32 * 32, dummy);
            glds4(nx.back32 + (size_t)(w * 5 + 4) * kTile + (lane & 15) * 32, dummy);
        }
        GAT8_BAR(19);
        f32x16 c23 = load_block(R0w, lane);
        if constexpr (H4) c23 = c23 * inv;
        // LayerNorm weights of what follows: the next block's norm1, or the encoder's final norm
        const bool last = bi + 1 == kDepth;
        const float* nvec = last ? a.norm_w : a.blk[bi + 1 < kDepth ? bi + 1 : bi].vecs + V_N1W;
        const float* nvecb = last ? a.norm_b : a.blk[bi + 1 < kDepth ? bi + 1 : bi].vecs + V_N1B;
        const f32x16 nw = load_chanvec_T(nvec, 32 * w, h), nb = load_chanvec_T(nvecb, 32 * w, h);
        GAT8_BAR(20);
        c23 = pick(R1w, lane, c23);
        xw += c01 + c23;
        tile_stats(xw, lds + kStat + w * 64, lane);
        if (a.blk_tap && tok < J) {          // debug tap (off in timed runs): this wave's channel block of the block output
            float* dst = a.blk_tap + (((size_t)bi * a.tapB + b) * J + tok) * kC + 32 * w + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v4;
#pragma unroll
                for (int j = 0; j < 4; ++j) v4[j] = xw[4 * g + j];
                *reinterpret_cast<f32x4*>(dst + 8 * g) = v4;
            }
        }
        GAT8_BAR(21);
        // ---- step 22: Y = LN1(x) for the next block; after the last block LN -> GELU -> feat (GAT.py:148-150)
        {
            f32x16 y = ln_stats(lds + kStat, xw, nw, nb, lane);
            if (!last) {
                st_opnd(A + w * kTileX3, lane, y);
            } else {
                gelu_tile8(y);
                if (tok < J) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 v4;
#pragma unroll
                        for (int j = 0; j < 4; ++j) v4[j] = y[4 * g + j];
                        *reinterpret_cast<f32x4*>(a.feat + ((size_t)b * J + tok) * kC + 32 * w + 8 * g + 4 * h) = v4;
                    }
                }
                if constexpr (TAIL) {        // feat stays on chip for the epilogue: channel block w as a T-layout tile (X is free since step 9);
                    if (tok >= J) y = zero16();      // zero for the token lanes that do not exist (they are operand columns of the joint-token linear)
                    store_block(X + w * kTile, lane, y);
                }
            }
        }
        GAT8_BAR(22);
    }
    GAT8_STAMPS_OUT(kDepth * 23 * 2, 2 * kDepth * 23 * 2 + kDepth * 8);
    if constexpr (TAIL) gat8_tail<LR, true>(a.tl, a.pose2d, lds, b, J, w, lane);
}

// one workgroup per destination tile: dst tile i <- src tile idx[i]  (TILE floats: 6 KiB X3 tiles or 4 KiB fp32 tiles)
template <int TILE>
__global__ void k_gather_tiles(const float* __restrict__ src, const int* __restrict__ idx, float* __restrict__ dst) {
    const f32x4* s = reinterpret_cast<const f32x4*>(src + (size_t)idx[blockIdx.x] * TILE);
    f32x4* d = reinterpret_cast<f32x4*>(dst + (size_t)blockIdx.x * TILE);
    for (int e = threadIdx.x; e < TILE / 4; e += blockDim.x) d[e] = s[e];
}
// H3 stream -> byte-lo stream, one wave per tile: hi and mid planes copied, lo packed as the top byte of fp16(2^24 lo); *bad counts
// the weights whose lo does not come back bit for bit from the product wave's own expansion (lo_expand)
__global__ void k_h3_to_h3b(const float* __restrict__ src, float* __restrict__ dst, unsigned* __restrict__ bad) {
    const int lane = threadIdx.x;
    const f16x8* q = reinterpret_cast<const f16x8*>(src + (size_t)blockIdx.x * kTileX3) + lane;
    f16x8* d = reinterpret_cast<f16x8*>(dst + (size_t)blockIdx.x * kTileH3B) + lane;
#pragma unroll
    for (int i = 0; i < 4; ++i) d[i * 64] = q[i * 64];
    unsigned pk[4] = {0, 0, 0, 0};
    unsigned nbad = 0;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        const f16x8 lo = q[(4 + s2) * 64];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const _Float16 up = (_Float16)((float)lo[j] * 16777216.0f);
            pk[s2 * 2 + (j >> 2)] |= (unsigned)(__builtin_bit_cast(unsigned short, up) >> 8) << (8 * (j & 3));
        }
        const f16x8 back = lo_expand(pk[s2 * 2], pk[s2 * 2 + 1]);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            nbad += __builtin_bit_cast(unsigned short, (_Float16)back[j]) != __builtin_bit_cast(unsigned short, (_Float16)lo[j]);
    }
    reinterpret_cast<uint4*>(dst + (size_t)blockIdx.x * kTileH3B + 1024)[lane] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    if (nbad) atomicAdd(bad, nbad);
}

}  // namespace

constexpr size_t kGat8Lds = (size_t)(kGat8LdsFloats + kDiagLdsFloats) * sizeof(float);

int gat8_prepare_device() {
    GATOR_HIP_CHECK(hipFuncSetAttribute((const void*)k_gat8<false, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGat8Lds));
    GATOR_HIP_CHECK(hipFuncSetAttribute((const void*)k_gat8<true, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGat8Lds));
    GATOR_HIP_CHECK(hipFuncSetAttribute((const void*)k_gat8<true, 12>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGat8Lds));
    GATOR_HIP_CHECK(hipFuncSetAttribute((const void*)k_gat8<true, 10, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGat8Lds));
    GATOR_HIP_CHECK(hipFuncSetAttribute((const void*)k_gat8<true, 12, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGat8Lds));
    GATOR_HIP_CHECK(hipFuncSetAttribute((const void*)k_gat8<true, 10, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGat8Lds));
    GATOR_HIP_CHECK(hipFuncSetAttribute((const void*)k_gat8<true, 12, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGat8Lds));
    GATOR_HIP_CHECK(hipFuncSetAttribute((const void*)k_gat8<true, 10, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGat8Lds));
    GATOR_HIP_CHECK(hipFuncSetAttribute((const void*)k_gat8<true, 12, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGat8Lds));
    GATOR_HIP_CHECK(hipFuncSetAttribute((const void*)k_gat8<true, 10, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGat8Lds));
    GATOR_HIP_CHECK(hipFuncSetAttribute((const void*)k_gat8<true, 12, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGat8Lds));
    return GATOR_OK;
}

// The per-wave weight streams: the X3 tile grids of fused_create_gat (gxbuf, tile-for-tile image of gbuf from gblk[0].qkv on)
// re-ordered into the order product wave w consumes them, block after block.
int gat8_build_stream(FusedState* f, void* stream) {
    if (!f->gat_x3 || !f->gxbuf) return GATOR_OK;
    std::vector<int> idx((size_t)4 * kWaveTiles + kNT, 0);
    auto tile_of = [&](const float* grid) { return (int)((grid - f->gblk[0].qkv) / kTile); };
    for (int w = 0; w < 4; ++w) {
        int* o = idx.data() + (size_t)w * kWaveTiles;
        for (int bi = 0; bi < kDepth; ++bi) {
            const GatBlockPk& p = f->gblk[bi];
            const int qkv = tile_of(p.qkv), w0 = tile_of(p.w0), w1 = tile_of(p.w1), proj = tile_of(p.proj), lin0 = tile_of(p.lin0), lin1 = tile_of(p.lin1),
                      back = tile_of(p.back), fc1 = tile_of(p.fc1), fc2 = tile_of(p.fc2);
            for (int kb = 0; kb < 4; ++kb) *o++ = qkv + (w * 4 + kb);                  // q
            for (int kb = 0; kb < 4; ++kb) *o++ = qkv + ((4 + w) * 4 + kb);            // k
            for (int kb = 0; kb < 4; ++kb) *o++ = qkv + ((8 + w) * 4 + kb);            // v
            for (int kb = 0; kb < 4; ++kb) *o++ = w0 + (w * 4 + kb);                   // MGCN W[0]
            for (int kb = 0; kb < 4; ++kb) *o++ = w1 + (w * 4 + kb);                   // MGCN W[1]
            for (int kb = 0; kb < 4; ++kb) *o++ = proj + (w * 4 + kb);                 // attention proj
            for (int kb = 0; kb < 4; ++kb) *o++ = lin0 + (w * 4 + kb);                 // X_Feat linears[0]
            *o++ = lin1 + w;                                                           // X_Feat linears[1], k block w
            for (int kb = 0; kb < 4; ++kb) *o++ = back + (w * 5 + kb);                 // linearback, k < 128
            for (int j = 0; j < 4; ++j)
                for (int kb = 0; kb < 4; ++kb) *o++ = fc1 + ((4 * w + j) * 4 + kb);    // fc1, hidden block 4w + j
            for (int u = 0; u < 4; ++u)
                for (int wq = 0; wq < 4; ++wq) *o++ = fc2 + (w * 16 + 4 * wq + u);     // fc2, hidden blocks {4w' + u}
            for (int i = 0; i < kPadTiles; ++i) *o++ = qkv;                             // dummy: loaded into a slot, never multiplied
        }
        if (o - idx.data() != (ptrdiff_t)(w + 1) * kWaveTiles) return fail(GATOR_EINVAL, "gat8_build_stream: tile count");
    }
    struct DevFree { void* p = nullptr; ~DevFree() { if (p) (void)hipFree(p); } } t_idx, t_h3;      // temporaries: freed on every return path
    GATOR_HIP_CHECK(hipMalloc(&t_idx.p, idx.size() * sizeof(int)));
    int* d_idx = (int*)t_idx.p;
    GATOR_HIP_CHECK(hipMemcpyAsync(d_idx, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice, (hipStream_t)stream));
    GATOR_HIP_CHECK(hipMalloc(&f->g8stream, (size_t)kStreamFloats * sizeof(float)));
    if (f->gat8_h4) {      // the four-product form streams three fp16 planes of 2^shift * w: the fp32 tiles are put in stream order
        float left = 0.f;  // first, so that the shift comes from exactly the weights the kernel multiplies (not the tables in between)
        GATOR_HIP_CHECK(hipMalloc(&t_h3.p, idx.size() * kTile * sizeof(float)));
        float* h3 = (float*)t_h3.p;
        k_gather_tiles<kTile><<<(unsigned)idx.size(), 128, 0, (hipStream_t)stream>>>(f->gblk[0].qkv, d_idx, h3);
        int rc = fused_repack_h3(h3, f->g8stream, (int64_t)idx.size(), &f->gat8_wshift, &left, stream);
        if (rc == GATOR_OK && left > 1e-7f) rc = fail(GATOR_EUNSUPPORTED, "GAT weights span more than fp16 x 3 planes hold exactly: use GATOR_GAT8_H4=0");
        if (rc) return rc;
        static const bool lobyte = [] { const char* e = getenv("GATOR_GAT8_LOBYTE"); return !(e && atoi(e) == 0); }();     // default on; =0 for A/B
        if (lobyte) {      // the byte-lo image of the same stream (H3B): used only if every lo value survives the round trip
            DevFree t_bad;
            GATOR_HIP_CHECK(hipMalloc(&t_bad.p, sizeof(unsigned)));
            GATOR_HIP_CHECK(hipMemsetAsync(t_bad.p, 0, sizeof(unsigned), (hipStream_t)stream));
            GATOR_HIP_CHECK(hipMalloc(&f->g8stream_b, (size_t)kStreamFloatsB * sizeof(float)));
            k_h3_to_h3b<<<(unsigned)idx.size(), 64, 0, (hipStream_t)stream>>>(f->g8stream, f->g8stream_b, (unsigned*)t_bad.p);
            GATOR_HIP_CHECK(hipGetLastError());
            unsigned nbad = 0;
            GATOR_HIP_CHECK(hipMemcpyAsync(&nbad, t_bad.p, sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)stream));
            GATOR_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
            f->gat8_lobyte = nbad == 0;
            if (!f->gat8_lobyte) { (void)hipFree(f->g8stream_b); f->g8stream_b = nullptr; }
        }
    } else {
        k_gather_tiles<kTileX3><<<(unsigned)idx.size(), 128, 0, (hipStream_t)stream>>>(f->gxbuf, d_idx, f->g8stream);
    }
    GATOR_HIP_CHECK(hipGetLastError());
    GATOR_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return GATOR_OK;
}

// feat only (the lifter and the MDR joint tokens are the batched launches of gat_tail.hip)
// tail_x_out != nullptr: the lifter (-> tail_x_out [B][3J]) and, with tail_jkv, the MDR joint tokens / K / V tiles run as the kernel's epilogue
// (gat8_tail_supported); ctr_B > 0: the kernel also zeroes the persistent MDR launch's counters of a forward of ctr_B samples
bool gat8_tail_supported(const gator_ctx* c, const FusedState* f, bool half16) {
    if (!half16 && !f->gat8_lobyte) return false;        // (the three-plane stream's kernel has no registers left for the epilogue: GATOR_GAT8_LOBYTE=0 keeps the two launches)
    return f->gat8 && f->gat8_h4 && f->g8stream != nullptr && f->mdr_x3 == 2 && f->wxbuf != nullptr && f->jf128_h3 != nullptr && (c->J == 17 || c->J == 19);
}

int launch_gat8(gator_ctx* c, FusedState* f, const float* pose2d, int B, float* feat, void* stream, int B_total, int tap_row0, bool half16,
                float* tail_x_out, float* tail_jkv, int ctr_B) {
    if (tail_x_out && !gat8_tail_supported(c, f, half16)) return fail(GATOR_EUNSUPPORTED, "k_gat8: the fused tail needs the four-product forms and 17 or 19 joints");
    if (half16 && !f->gat8_h4) return fail(GATOR_EUNSUPPORTED, "the 16-bit encoder needs the four-product weight stream (GATOR_GAT8_H4=1, the default)");
    Gat8Args a;
    const Weights& w = c->w;
    a.B = B; a.J = c->J; a.pose2d = pose2d;
    a.gl0_W = w.gl0_W; a.gl0_b = w.gl0_b; a.gn_w = w.gn_w; a.gn_b = w.gn_b; a.gl3_p = f->g_gl3; a.gl3_b = w.gl3_b; a.posT = f->g_posT;
    a.biasT = f->g_biasT; a.m1T = f->g_m1T; a.m2T = f->g_m2T;
    a.norm_w = w.norm_w; a.norm_b = w.norm_b;
    a.wstream = f->g8stream;
    for (int i = 0; i < kDepth; ++i) {
        const GatBlockPk& p = f->gblk[i];
        a.blk[i] = Gat8Blk{p.back, p.mc, p.mdT, p.aoffT, p.f1b, f->g_vecs + (size_t)i * 2048};
    }
    a.feat = feat;
    a.tl = Gat8Tail{};
    const bool tail = tail_x_out != nullptr;
    if (tail) {
        Gat8Tail& tl = a.tl;
        tl.lifter_w = w.lifter_w; tl.lifter_b = w.lifter_b; tl.x_out = tail_x_out; tl.jkv = tail_jkv;
        if (ctr_B > 0 && f->mdr_persist != 0) { tl.mdr_ctr = f->mdr_ctr; tl.ctr_B = ctr_B; f->mdr_ctr_clean = true; }
        tl.jf_p = f->jfeat128_p;
        for (int i = 0; i < 3; ++i) { tl.j_wk_p[i] = f->lay[i].wk; tl.j_wv_p[i] = f->lay[i].wv; }
        tl.jf5 = f->jfeat5; tl.jf_h3 = f->jf128_h3; tl.jf_b = w.jfeat_b; tl.posj_T = f->posj_T;
        auto h3_of = [&](const float* t) { return f->wxbuf + (size_t)(t - f->lay[0].wq) / kTile * kTileX3; };      // the MDR layers' H3 image mirrors wbuf tile for tile
        for (int i = 0; i < 3; ++i) { tl.j_n1w[i] = w.lay[i].n1w; tl.j_n1b[i] = w.lay[i].n1b; tl.j_wk_h3[i] = h3_of(f->lay[i].wk); tl.j_wv_h3[i] = h3_of(f->lay[i].wv); }
        tl.jf_inv = std::ldexp(1.0f, -(4 + f->jf128_wshift));
        tl.kv_inv = std::ldexp(1.0f, -(4 + f->mdr_wshift));
    }
    a.blk_tap = nullptr;
    a.tapB = B;
    if (c->block_taps) {
        const int Bt = B_total > 0 ? B_total : B;
        int rc = gat_ensure_blk_tap(c, f, Bt);
        if (rc) return rc;
        a.blk_tap = f->blk_tap + (size_t)tap_row0 * c->J * kC;
        a.tapB = Bt;
    }
    static const bool l2warm = [] { const char* e = getenv("GATOR_GAT_L2WARM"); return !(e && atoi(e) == 0); }();     // default on; =0 for A/B
    a.pf_n = std::min(32, (B + 7) / 8);                  // workgroups b and b + 8 share an XCD (round-robin dispatch; speed only)
    const bool lob = f->gat8_h4 && f->gat8_lobyte && !half16;                         // the byte-lo stream (the one-plane form reads hi and mid of the H3 stream)
    if (lob) a.wstream = f->g8stream_b;
    a.pf_loads = l2warm ? (kBlkTiles * (lob ? kTileH3B : kTileX3) * 4 / a.pf_n + 8191) / 8192 : 0;       // 8 KiB (64 lines) per instruction
    if (a.pf_loads > 6) a.pf_loads = 0;      // fewer than ~8 workgroups per XCD (B < 64): a share is so large that touching it costs more than it hides
    a.tl.warm_n = (tail && l2warm && a.pf_n >= 8) ? a.pf_n : 0;      // (fewer than 8 workgroups per XCD: a share is too large to be worth touching)
#ifdef GATOR_DIAG
    a.stamps = nullptr;
    a.dbg = getenv("GATOR_GAT8_DBG") ? atoi(getenv("GATOR_GAT8_DBG")) : 0;
    static const bool want_stamps = getenv("GATOR_GAT_STAMPS") != nullptr;
    constexpr int kSt = 2 * kDepth * 23 * 2 + kDepth * 8;
    if (want_stamps) {
        GATOR_HIP_CHECK(hipMalloc(&a.stamps, kSt * sizeof(unsigned long long)));
        GATOR_HIP_CHECK(hipMemset(a.stamps, 0, kSt * sizeof(unsigned long long)));
    }
#endif
#ifdef GATOR_DIAG
    a.tl.tstamps = nullptr;
    if (want_stamps && tail) {
        GATOR_HIP_CHECK(hipMalloc(&a.tl.tstamps, 64 * sizeof(unsigned long long)));
        GATOR_HIP_CHECK(hipMemset(a.tl.tstamps, 0, 64 * sizeof(unsigned long long)));
    }
#endif
    a.lin_inv = std::ldexp(1.0f, -(4 + f->gat8_wshift));
    // token rows that exist sit in registers r < LR of a row-over-token tile: token t <-> r = (t & 3) + 4 (t >> 3), so J <= 18 / 20 -> 10 / 12
    if (tail && half16 && c->J == 17) k_gat8<true, 10, true, false, true><<<B, 512, kGat8Lds, (hipStream_t)stream>>>(a);
    else if (tail && half16) k_gat8<true, 12, true, false, true><<<B, 512, kGat8Lds, (hipStream_t)stream>>>(a);
    else if (tail && lob && c->J == 17) k_gat8<true, 10, false, true, true><<<B, 512, kGat8Lds, (hipStream_t)stream>>>(a);
    else if (tail && lob) k_gat8<true, 12, false, true, true><<<B, 512, kGat8Lds, (hipStream_t)stream>>>(a);
    else if (tail) return fail(GATOR_EUNSUPPORTED, "k_gat8: no fused tail on the three-plane weight stream");
    else if (!f->gat8_h4) k_gat8<false, 16><<<B, 512, kGat8Lds, (hipStream_t)stream>>>(a);
    else if (half16 && c->J <= 18) k_gat8<true, 10, true><<<B, 512, kGat8Lds, (hipStream_t)stream>>>(a);
    else if (half16 && c->J <= 20) k_gat8<true, 12, true><<<B, 512, kGat8Lds, (hipStream_t)stream>>>(a);
    else if (lob && c->J <= 18) k_gat8<true, 10, false, true><<<B, 512, kGat8Lds, (hipStream_t)stream>>>(a);
    else if (lob && c->J <= 20) k_gat8<true, 12, false, true><<<B, 512, kGat8Lds, (hipStream_t)stream>>>(a);
    else if (c->J <= 18) k_gat8<true, 10><<<B, 512, kGat8Lds, (hipStream_t)stream>>>(a);
    else if (c->J <= 20) k_gat8<true, 12><<<B, 512, kGat8Lds, (hipStream_t)stream>>>(a);
    else return fail(GATOR_EUNSUPPORTED, "k_gat8: more than 20 joints (gator_create admits 17 and 19)");
    GATOR_HIP_CHECK(hipGetLastError());
#ifdef GATOR_DIAG
    if (a.tl.tstamps) {
        unsigned long long ts[64];
        GATOR_HIP_CHECK(hipMemcpy(ts, a.tl.tstamps, sizeof(ts), hipMemcpyDeviceToHost));
        GATOR_HIP_CHECK(hipFree(a.tl.tstamps));
        fprintf(stderr, "[k_gat8 tail stamps, one workgroup, B=%d] s_memtime cycles since wave 0 entered: entry | loads issued + joint-token MFMAs | lifter done | at barrier | past barrier | jf + LN done | job 1 | job 2\n", B);
        for (int v = 0; v < 8; ++v) {
            fprintf(stderr, "  wave %d:", v);
            for (int k = 0; k < 8; ++k) fprintf(stderr, " %6lld", ts[v * 8 + k] ? (long long)(ts[v * 8 + k] - ts[0]) : -1LL);
            fprintf(stderr, "\n");
        }
    }
    if (a.stamps) {     // diagnostic build: synchronous read-back; blocks 1..5 averaged (block 0 carries the cold start)
        std::vector<unsigned long long> hs(kSt);
        GATOR_HIP_CHECK(hipMemcpy(hs.data(), a.stamps, kSt * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        GATOR_HIP_CHECK(hipFree(a.stamps));
        double tot[2] = {0, 0};
        fprintf(stderr, "[k_gat8 stamps, wg0, B=%d] cycles per step, mean of blocks 1-5: step | product work wait | helper work wait\n", B);
        for (int n = 0; n < 23; ++n) {
            double v[4] = {0, 0, 0, 0};
            for (int role = 0; role < 2; ++role)
                for (int bi = 1; bi < kDepth; ++bi)
                    for (int k = 0; k < 2; ++k) v[role * 2 + k] += (double)hs[(((size_t)role * kDepth + bi) * 23 + n) * 2 + k] / (kDepth - 1);
            fprintf(stderr, "  %2d | %7.0f %7.0f | %7.0f %7.0f\n", n, v[0], v[1], v[2], v[3]);
            tot[0] += v[0] + v[1]; tot[1] += v[2] + v[3];
        }
        fprintf(stderr, "  per block: product %.0f, helper %.0f cycles\n", tot[0], tot[1]);
        fprintf(stderr, "  helper sub-stamps of step 15 (block 1), cycles since the barrier:");
        for (int k = 0; k < 8; ++k) fprintf(stderr, " %llu", hs[(size_t)2 * kDepth * 23 * 2 + 1 * 8 + k]);
        fprintf(stderr, "\n");
    }
#endif
    return GATOR_OK;
}

}  // namespace gator
