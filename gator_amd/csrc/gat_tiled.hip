// Sample-tiled GAT encoder for large batches (lib/models/GAT.py:133-150, GATBlock :33-43): several samples per workgroup, their
// joint tokens packed DENSELY into 32-token MFMA tiles, every weight tile fetched once per workgroup.
//
// k_gat (gat_fused.hip) gives one sample a whole workgroup: J = 17 tokens fill 17 of a tile's 32 columns (47 % of every MFMA is
// padding) and the 9.5 MB of split-precision weights are re-streamed from L2 for every sample.  That is the right shape while
// there are no more samples than CUs; beyond that this kernel takes over:
//   * a workgroup owns S consecutive samples (S*J <= 128 token slots; J=17: S=7, 119/128 rows used; J=19: S=6), one WAVE per
//     32-token tile.  Token-wise work -- LayerNorms, all nine linears of a block, GELU -- never leaves the wave's registers: the
//     128-channel residual stream of the tile is 4 accumulator-layout blocks (fused_common.h), an accumulator tile is the next
//     product's operand after an in-lane split (x3_common.h);
//   * weights come through LDS: the four waves copy the next group of 4-5 operand tiles (24-30 KB) with LDS-DMA
//     (global_load_lds, no registers) while the current group feeds the MFMAs -- one workgroup barrier per group, and a tile that
//     k_gat reads once per sample is read once per S samples;
//   * the only cross-token operators -- the J x J attention (modules.py:121-138), the MGCN adjacency product (:243-255) and the
//     hop-1 / hop-2 aggregations of X_Feat (:158-177) -- act inside a sample, whose tokens may straddle two tiles.  They run on
//     the VALU in exact fp32 on rows exchanged through LDS ([token][32 channels] images): lane (token, half h) works on head
//     2*nb+h resp. on its own 16 channels, reading the J rows of its sample.  Padded to 32x32 MFMA tiles per sample these
//     products would cost 3x as many cycles (17/32 squared), and the VALU is otherwise idle while the matrix pipe works.
// The kernel ends with `feat`; the lifter and the MDR joint tokens follow as batched launches (gat_tail.hip).
#include "fused_common.h"
#include "fused_state.h"
#include "x3_common.h"

#include <cmath>
#include <type_traits>

namespace gator {
namespace {

constexpr int kTW = 4;                 // waves = 32-token tiles per workgroup
constexpr int kTT = 32 * kTW;          // token slots per workgroup
constexpr int kEXS = 36;               // row stride (floats) of an exchange image: 16-byte aligned rows, conflict-free b128 access
constexpr int kTabS = 20;              // row stride of the J x J tables (J <= 19)
constexpr int kGrpTiles = 4;           // tiles per weight group (ring slot = 4 x 6 KiB = 24 KiB, 6 pieces of 1 KiB per wave)
constexpr int kRing = 3;               // ring slots: group k+2 is copied while group k is in use
constexpr int kGroupsPerBlock = 66;
constexpr float kLog2eT = 1.4426950408889634f;

enum { TV_N1W = 0, TV_N1B = 128, TV_QKVB = 256, TV_PROJB = 640, TV_GCNB = 768, TV_LIN0B = 896, TV_BACKB = 1024, TV_N2W = 1152,
       TV_N2B = 1280, TV_FC1B = 1408, TV_FC2B = 1920, TV_TOTAL = 2048 };     // order of FusedState::g_vecs (gat_fused.hip)

struct TiledBlk {
    const float *qkv, *proj, *w0, *w1, *lin0, *lin1, *back, *fc1, *fc2;       // X3 tile grids [NB][KB]
    const float *vecs, *M, *lin1_b;                                           // per-block vectors (TV_*), gcn.M [J][128], linears[1].bias [16]
};
struct TiledArgs {
    int B, S, Btap;                 // Btap: batch stride of the block-tap buffer (the whole batch of the call)
    const float* pose2d;
    const float *gl0_W, *gl0_b, *gn_w, *gn_b, *gl3_p, *gl3_b, *pos;           // pos = pos_id_embed[1..J] + pos_num_embed[deg], [J][128]
    const float *hop_bias, *adj_diag, *adj_off, *m1, *m2;                     // [8][J][J], [6][J], [6][J][J], [J][J], [J][J]
    const float *norm_w, *norm_b;
    TiledBlk blk[kDepth];
    float *feat, *blk_tap;
    float lin_s, lin_inv;           // H4 form: the token-wise products return lin_s x their value (x3_common.h: 4-product linears)
};

// LDS carve-up (floats)
constexpr int kRingSlot = kGrpTiles * kTileX3;
constexpr int oRING = 0, oEXK = oRING + kRing * kRingSlot, oEXV = oEXK + kTT * kEXS, oVEC = oEXV + kTT * kEXS,
              oBIAS = oVEC + TV_TOTAL, oAOFF = oBIAS + 8 * 19 * kTabS, oM1 = oAOFF + kDepth * 19 * kTabS, oM2 = oM1 + 19 * kTabS,
              oADIAG = oM2 + 19 * kTabS, oL1B = oADIAG + kDepth * 32, oMT = oL1B + kDepth * 16, kTiledLdsFloats = oMT + 19 * kC;
static_assert(kTiledLdsFloats * 4 <= 160 * 1024, "LDS budget");

__device__ __forceinline__ f32x16 chanvec_L(const float* V, int off, int h) {      // v[r] = V[off + kap(r) + 4h]
    f32x16 v;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(V + off + 8 * g + 4 * h);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[4 * g + j] = t[j];
    }
    return v;
}
// the 16 channels of block `cb` that lane half h holds, read from row-major [..][128] at row pointer `row` (global or LDS)
__device__ __forceinline__ f32x16 row_block(const float* row, int cb, int h) { return chanvec_L(row, 32 * cb, h); }

__device__ __forceinline__ float sum64(const f32x16 (&x)[4]) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += (x[0][r] + x[1][r]) + (x[2][r] + x[3][r]);
    return s + xhalf(s);
}
// nn.LayerNorm(128) of the wave's tokens, result split into operand planes
template <bool GELU>
__device__ __forceinline__ void ln128(const f32x16 (&x)[4], const float* V, int ow, int ob, int h, f32x16 (&y)[4]) {
    const float mean = sum64(x) * (1.0f / 128.0f);
    f32x16 d[4], sq[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) { d[kb] = x[kb] - mean; sq[kb] = d[kb] * d[kb]; }
    const float rstd = 1.0f / sqrtf(sum64(sq) * (1.0f / 128.0f) + 1e-5f);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        y[kb] = d[kb] * rstd * chanvec_L(V, ow + 32 * kb, h) + chanvec_L(V, ob + 32 * kb, h);
        if (GELU) gelu_tile(y[kb]);
    }
}

// write a T-layout block (token on the lane) as row `t` of an exchange image / read the same 16 columns of another row back
__device__ __forceinline__ void ex_write(float* EX, int t, int h, const f32x16& v) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 q;
#pragma unroll
        for (int j = 0; j < 4; ++j) q[j] = v[4 * g + j];
        *reinterpret_cast<f32x4*>(EX + t * kEXS + 8 * g + 4 * h) = q;
    }
}
// out[r] = sum_j' coef[j'] * EX[base + j'][my 16 columns]   (the lane's own channels of the J tokens of its sample).  The coefficients are
// read from their LDS table row as they are used: held in a register array (J floats, two such arrays live across the whole X_Feat loop) they
// were what pushed the kernel over its 512 registers (round 6: 160 / 188 B of scratch per lane -> 0; same values, same order, same bits)
template <int J>
__device__ __forceinline__ f32x16 aggregate(const float* EX, int base, int h, const float* coef) {
    f32x16 acc = zero16();
    // a (mostly) ROLLED loop: fully unrolled, hipcc hoists all 4 J row reads (68 - 76 b128 loads = 270 - 300 registers in flight) to the top of the sum -- that,
    // not the kernel's state, was its scratch (round 6, found by cutting the kernel apart: without this sum 0 B, with it 160 / 188 B per lane)
#pragma unroll 4
    for (int jp = 0; jp < J; ++jp) {
        const float* row = EX + (base + jp) * kEXS + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(row + 8 * g);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[4 * g + j] = fmaf(coef[jp], q[j], acc[4 * g + j]);
        }
    }
    return acc;
}
template <int J>
__device__ __forceinline__ void table_row(const float* T, int j, float (&out)[J]) {     // row j of a [.][kTabS] table
#pragma unroll
    for (int q = 0; q < (J + 3) / 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(T + j * kTabS + 4 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (4 * q + i < J) out[4 * q + i] = v[i];
    }
}

// T-layout block <-> "head layout": lane (token, h) gets the 16 channels of head h of the block in natural order.
// v_permlane32_swap exchanges lanes 32..63 of its first operand with lanes 0..31 of its second.
__device__ __forceinline__ void swap32(float& a, float& b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}
// channel (within the head) of element i of the swapped pair list: a_r -> kap8(r), b_r -> kap8(r) + 4
__device__ __forceinline__ void to_head(const f32x16& v, float (&q)[16]) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        float a = v[r], b = v[r + 8];
        swap32(a, b);                       // h=0: a = own reg r, b = partner's reg r ; h=1: a = partner's reg r+8, b = own reg r+8
        const int c = (r & 3) + 8 * (r >> 2);
        q[c] = a;
        q[c + 4] = b;
    }
}
__device__ __forceinline__ f32x16 from_head(const float (&q)[16]) {
    f32x16 v;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int c = (r & 3) + 8 * (r >> 2);
        float a = q[c], b = q[c + 4];
        swap32(a, b);
        v[r] = a;
        v[r + 8] = b;
    }
    return v;
}

// LDS-DMA of one 1 KiB piece (64 lanes x 16 B): global -> LDS without registers.  Issued through inline asm ON PURPOSE: hipcc
// counts the builtin form in its vmcnt bookkeeping and, because the copy writes LDS, drains it (s_waitcnt vmcnt(0)) before the
// next LDS access or the next copy -- measured: 0.48 of the kernel's 1.2 ms at B=1024 was that wait.  The asm form is invisible
// to that bookkeeping; begin_group() waits for it explicitly before the barrier that publishes the slot.
__device__ __forceinline__ void glds16(const float* gsrc, const float* lds_dst) {
    unsigned keep;
    const unsigned dst = (unsigned)(unsigned long long)lds_dst;             // low 32 bits of a generic LDS address = the LDS byte offset
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

// N weight tiles of a ring slot, one product each: tile i+1 is read from LDS while tile i feeds its 12 MFMAs, and never earlier
// (unfenced, hipcc hoists all of a group's LDS reads to its top: 5 x 24 registers on top of the 300 the state needs -> scratch).
__device__ __forceinline__ void ld_wtile(X3& o, const float* p, int lane) { o = x3_load(p, lane); }
__device__ __forceinline__ void ld_wtile(H3& o, const float* p, int lane) { o = h3_load(p, lane); }
__device__ __forceinline__ void ld_wtile(G2& o, const float* p, int lane) { o = g2_load(p, lane); }      // the hi and mid planes of an H3 tile
template <int N, class WT, class F>
__device__ __forceinline__ void for_tiles(const float* slot, int lane, F&& f) {
    WT cur;
    ld_wtile(cur, slot, lane);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        WT nxt = cur;
        if (i + 1 < N) ld_wtile(nxt, slot + (i + 1) * kTileX3, lane);
        f(i, cur);
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
    }
}

// tiles of weight group g (0..65) of a block, in consumption order (see the kernel body)
__device__ __forceinline__ int group_tiles(const TiledBlk& w, int g, const float* (&t)[kGrpTiles]) {
    auto T = [](const float* base, int idx) { return base + (size_t)idx * kTileX3; };
    if (g < 24) {                             // per channel block nb: {q,k,v,W1}[kb] x 4, W0[kb = 0..3], proj[.][nb]
        const int nb = g / 6, r = g % 6;
        if (r < 4) {
            t[0] = T(w.qkv, nb * 4 + r); t[1] = T(w.qkv, (4 + nb) * 4 + r); t[2] = T(w.qkv, (8 + nb) * 4 + r); t[3] = T(w.w1, nb * 4 + r);
        } else if (r == 4) {
            for (int i = 0; i < 4; ++i) t[i] = T(w.w0, nb * 4 + i);
        } else {
            for (int i = 0; i < 4; ++i) t[i] = T(w.proj, i * 4 + nb);
        }
        return 4;
    }
    if (g < 34) {
        const int q = g - 24;                 // GL(0), GL1, GB(0), GL(1), GB(1), GL(2), GB(2), GL(3), GB(3), GB(4)
        if (q == 1) { for (int i = 0; i < 4; ++i) t[i] = T(w.lin1, i); return 4; }
        if (q == 9) { for (int i = 0; i < 4; ++i) t[i] = T(w.back, i * 5 + 4); return 4; }
        const int nb = q == 0 ? 0 : (q - 1) / 2;
        const bool lin = q == 0 || (q >= 3 && (q & 1));
        if (lin) { for (int i = 0; i < 4; ++i) t[i] = T(w.lin0, nb * 4 + i); }
        else { for (int i = 0; i < 4; ++i) t[i] = T(w.back, i * 5 + nb); }
        return 4;
    }
    const int c = (g - 34) >> 1;
    if (((g - 34) & 1) == 0) { for (int i = 0; i < 4; ++i) t[i] = T(w.fc1, c * 4 + i); }
    else { for (int i = 0; i < 4; ++i) t[i] = T(w.fc2, i * 16 + c); }
    return 4;
}

// H4: token-wise products on four partial products (weights H3 in the ring, operands X2 in registers): every accumulator of such a
// product - and every exchange image, bias and aggregate made from one - carries the factor a.lin_s; the residual stream, the
// LayerNorms and the GELU's result do not.
// H2 (with H4; BASELINE config 3, the 16-bit operand mode: DESIGN.md 4e): activations as ONE fp16 plane, weights as the hi and mid planes
// of the same ring tiles (the lo plane is not even copied): two MFMAs per k-step instead of four; the J x J operators stay fp32 on the VALU.
template <int J, bool H4, bool H2 = false>
__global__ __launch_bounds__(256, 1) void k_gat_tiled(const TiledArgs a) {
    static_assert(H4 || !H2, "the one-plane form shares the four-product form's scales");
    typedef typename std::conditional<H2, G2, typename std::conditional<H4, H3, X3>::type>::type WT;
    typedef typename std::conditional<H2, X1, typename std::conditional<H4, X2, X3>::type>::type OT;
    constexpr int NP = H2 ? 4 : 6;                          // 1 KiB pieces of a ring tile that are copied
    const float inv = H4 ? a.lin_inv : 1.0f;
    auto sp = [&](const f32x16& v, float pre) -> OT {       // operand form of a register tile that holds 1 / pre x its value
        if constexpr (H2) return x1_cvt(v * (16.0f * pre)); else if constexpr (H4) return x2_split(v * (16.0f * pre)); else return x3_split(v);
    };
    auto mm = [&](const WT& wt, const OT& x, const f32x16& acc) -> f32x16 {
        if constexpr (H2) return g2_mma_wa(wt, x, acc); else if constexpr (H4) return h3_mma_wa(wt, x, acc); else return x3_mma(wt, x, acc);
    };
    // the block's vector table: biases that start an accumulator carry lin_s
    auto stage_vecs = [&](const float* vecs, float* Vd, int tid_) {
        for (int e = tid_; e < TV_TOTAL / 4; e += 256) {
            f32x4 v = reinterpret_cast<const f32x4*>(vecs)[e];
            const int off = 4 * e;
            if (H4 && ((off >= TV_QKVB && off < TV_N2W) || off >= TV_FC1B)) v = v * a.lin_s;
            reinterpret_cast<f32x4*>(Vd)[e] = v;
        }
    };
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *RING = lds + oRING, *EXK = lds + oEXK, *EXV = lds + oEXV, *V = lds + oVEC;
    float *TBIAS = lds + oBIAS, *TAOFF = lds + oAOFF, *TM1 = lds + oM1, *TM2 = lds + oM2, *TADIAG = lds + oADIAG, *TL1B = lds + oL1B, *MT = lds + oMT;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s0 = blockIdx.x * a.S, ns = min(a.S, a.B - s0);               // samples of this workgroup
    const int t = 32 * wave + (lane & 31);                                   // token slot
    const bool valid = t < ns * J;
    const int tc = valid ? t : 0;                                            // clamped slot for addressing
    const int j = tc % J, base = tc - j;                                     // joint, first slot of the sample
    const size_t gtok = (size_t)s0 * J + tc;                                 // global token row (samples are contiguous)

    // ---- weight ring: the four waves copy group k+1 while group k is in use; one barrier per group ----------------------
    int gk = 0;                                                              // next group to begin (global order, bi * 62 + g)
    // Weight stream: LDS-DMA two groups ahead.  Every wave copies 6 of a group's 24 pieces (1 KiB each).  All workgroups walk the
    // same 9.5 MB in step, so each group is an L2 miss for everybody at once (Infinity-Cache latency, 1-2 us under load): with
    // one group of lead the copy was 0.5 of 1.2 ms at B=1024; with two it is hidden (issuing the copies in front of the group or
    // one tile's worth after each tile's MFMAs measures the same).  Register staging (plain loads, ds_write_b128 a group later)
    // needs a second 24-register set for the same lead; with one set it measured 1.08 ms.
    auto issue = [&](int k) {
        if (k >= kDepth * kGroupsPerBlock) return;
        const float* tl[kGrpTiles];
        group_tiles(a.blk[k / kGroupsPerBlock], k % kGroupsPerBlock, tl);
        float* slot = RING + (k % kRing) * kRingSlot;
#pragma unroll
        for (int i = 0; i < kGrpTiles; ++i)         // tile i: wave w takes pieces p = (w - 2i) mod 4 and p + 4 (if < 6): 6 per wave
#pragma unroll
            for (int p0 = 0; p0 < 2; ++p0) {
                const int p = ((wave - 2 * i) & 3) + 4 * p0;
                if (p < NP) glds16(tl[i] + p * 256 + lane * 4, slot + (i * 6 + p) * 256);
            }
    };
    auto begin_group = [&]() -> const float* {      // -> ring slot holding the group that begins now
        // vmcnt retires in order: all but this wave's 6 youngest copies (group gk+1) done  =>  its pieces of group gk have landed
        if (gk + 1 < kDepth * kGroupsPerBlock) { if constexpr (NP == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                            // ... and so have everybody's; slot (gk+2) % 3 (group gk-1) is free
        issue(gk + 2);
        const float* slot = RING + (gk % kRing) * kRingSlot;
        ++gk;
        return slot;
    };
    issue(0);
    issue(1);
    // ---- constant J x J tables -> LDS -------------------------------------------------------------------------------------
    for (int e = tid; e < 8 * J * J; e += 256) TBIAS[(e / J) * kTabS + e % J] = a.hop_bias[e];
    for (int e = tid; e < kDepth * J * J; e += 256) TAOFF[(e / J) * kTabS + e % J] = a.adj_off[e];
    for (int e = tid; e < J * J; e += 256) { TM1[(e / J) * kTabS + e % J] = a.m1[e]; TM2[(e / J) * kTabS + e % J] = a.m2[e]; }
    for (int e = tid; e < kDepth * J; e += 256) TADIAG[(e / J) * 32 + e % J] = a.adj_diag[e];
    for (int e = tid; e < kDepth * 16; e += 256) TL1B[e] = a.blk[e >> 4].lin1_b[e & 15] * (H4 ? a.lin_s : 1.0f);
    for (int e = tid; e < J * kC / 4; e += 256) reinterpret_cast<f32x4*>(MT)[e] = reinterpret_cast<const f32x4*>(a.blk[0].M)[e];
    stage_vecs(a.blk[0].vecs, V, tid);

    // ---- embedding: GraphLinear(2->64) . GroupNorm(4,64) . GELU . GraphLinear(64->128) + pos (GAT.py:135-144) -------------
    f32x16 x[4];
    {
        const float px = a.pose2d[gtok * 2], py = a.pose2d[gtok * 2 + 1];
        f32x16 hb[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = 32 * kb + kap(r) + 4 * h;
                hb[kb][r] = a.gl0_W[c * 2] * px + a.gl0_W[c * 2 + 1] * py + a.gl0_b[c];
            }
        }
        // GroupNorm over (16 channels x J tokens) per sample and group: group = 2 kb + (r >> 3); two-pass, sums exchanged through LDS
        float* ST = EXK;                                                     // [slot][4]
        float gs[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) s += hb[g4 >> 1][8 * (g4 & 1) + r];
            gs[g4] = s + xhalf(s);
        }
        if (h == 0) *reinterpret_cast<f32x4*>(ST + t * 4) = f32x4{gs[0], gs[1], gs[2], gs[3]};
        __syncthreads();
        float mean[4] = {0.f, 0.f, 0.f, 0.f};
        for (int jp = 0; jp < J; ++jp) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(ST + (base + jp) * 4);
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) mean[g4] += v[g4];
        }
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) mean[g4] = mean[g4] / (16.0f * J);
        __syncthreads();
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) { const float d = hb[g4 >> 1][8 * (g4 & 1) + r] - mean[g4]; s += d * d; }
            gs[g4] = s + xhalf(s);
        }
        if (h == 0) *reinterpret_cast<f32x4*>(ST + t * 4) = f32x4{gs[0], gs[1], gs[2], gs[3]};
        __syncthreads();
        float rstd[4] = {0.f, 0.f, 0.f, 0.f};
        for (int jp = 0; jp < J; ++jp) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(ST + (base + jp) * 4);
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) rstd[g4] += v[g4];
        }
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) rstd[g4] = 1.0f / sqrtf(rstd[g4] / (16.0f * J) + 1e-5f);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = 32 * kb + kap(r) + 4 * h, g4 = 2 * kb + (r >> 3);
                hb[kb][r] = gelu_f((hb[kb][r] - mean[g4]) * rstd[g4] * a.gn_w[c] + a.gn_b[c]);
            }
        // GraphLinear(64->128) on the fp32-input MFMA + folded position embeddings of joint j
        const float* prow = a.pos + (size_t)j * kC;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            f32x16 acc = chanvec_L(a.gl3_b, 32 * nb, h) + row_block(prow, nb, h), ac1 = zero16();
            mma2_T(load_wtile(a.gl3_p, nb * 2 + 0, lane), hb[0], acc, load_wtile(a.gl3_p, nb * 2 + 1, lane), hb[1], ac1);
            x[nb] = acc + ac1;
        }
    }

    for (int bi = 0; bi < kDepth; ++bi) {
        float adiag;
        // ================= phase 1: x_hat = LN1(x); per channel block nb: q,k,v (attention), h0,h1 (MGCN), then proj =================
        f32x16 pacc[4];                                                      // s = proj(attn) + proj.bias + MGCN + gcn.bias
        {
            OT ys[4];
            __syncthreads();                                                 // V of this block (written behind the previous block's last barrier) is visible
            {
                f32x16 yf[4];
                ln128<false>(x, V, TV_N1W, TV_N1B, h, yf);
#pragma unroll
                for (int i = 0; i < 4; ++i) ys[i] = sp(yf[i], 1.0f);
            }
            adiag = TADIAG[bi * 32 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i) pacc[i] = chanvec_L(V, TV_PROJB + 32 * i, h) + chanvec_L(V, TV_GCNB + 32 * i, h);
#pragma unroll 1
            for (int nb = 0; nb < 4; ++nb) {
                f32x16 q = chanvec_L(V, TV_QKVB + 32 * nb, h), k = chanvec_L(V, TV_QKVB + 128 + 32 * nb, h),
                       v = chanvec_L(V, TV_QKVB + 256 + 32 * nb, h), h1 = zero16();
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    const float* slot = begin_group();
                    for_tiles<4, WT>(slot, lane, [&](int i, const WT& wt) {
                        if (i == 0) q = mm(wt, ys[kb], q);
                        if (i == 1) k = mm(wt, ys[kb], k);
                        if (i == 2) v = mm(wt, ys[kb], v);
                        if (i == 3) h1 = mm(wt, ys[kb], h1);
                    });
                }
                // exchange images of this channel block: K and V rows of every token slot
                ex_write(EXK, t, h, k);
                ex_write(EXV, t, h, v);
                f32x16 h0 = zero16();
                {
                    const float* slot = begin_group();                       // W0 tiles; the barrier publishes K and V
                    for_tiles<4, WT>(slot, lane, [&](int kb, const WT& wt) { h0 = mm(wt, ys[kb], h0); });
                }
                // ---- attention of head 2 nb + h for this lane's query token (modules.py:121-138) ----
                f32x16 att;
                {
                    float qh[16], sc[J], o[16];
                    to_head(q, qh);
                    const float* brow = TBIAS + ((2 * nb + h) * J + j) * kTabS;       // row j of the head's hop / path bias table (LDS)
                    float mx = -1e30f;
#pragma unroll
                    for (int jp = 0; jp < J; ++jp) {
                        const float* row = EXK + (base + jp) * kEXS + 16 * h;
                        float s = 0.f;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4 kv = *reinterpret_cast<const f32x4*>(row + 4 * g);
#pragma unroll
                            for (int i = 0; i < 4; ++i) s = fmaf(qh[4 * g + i], kv[i], s);
                        }
                        s = (s * (0.25f * inv * inv) + brow[jp]) * kLog2eT;  // q k^T * head_dim**-0.5 + hop/path bias (q, k carry lin_s)
                        sc[jp] = s;
                        mx = fmaxf(mx, s);
                    }
                    float l = 0.f;
#pragma unroll
                    for (int jp = 0; jp < J; ++jp) { sc[jp] = __builtin_amdgcn_exp2f(sc[jp] - mx); l += sc[jp]; }
                    const float il = 1.0f / l;
#pragma unroll
                    for (int i = 0; i < 16; ++i) o[i] = 0.f;
#pragma unroll
                    for (int jp = 0; jp < J; ++jp) {
                        const float* row = EXV + (base + jp) * kEXS + 16 * h;
                        const float p = sc[jp] * il;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4 vv = *reinterpret_cast<const f32x4*>(row + 4 * g);
#pragma unroll
                            for (int i = 0; i < 4; ++i) o[4 * g + i] = fmaf(p, vv[i], o[4 * g + i]);
                        }
                    }
                    att = from_head(o);
                }
                const f32x16 mblk = row_block(MT + j * kC, nb, h);             // gcn.M[j][my channels], staged per block
                const float* slot = begin_group();                           // proj tiles; every wave is done with the K image
                ex_write(EXK, t, h, mblk * h1);                              // M . h1 takes its place
                // ---- proj: contribution of channel block nb of the attention output to all four output blocks ----
                {
                    const OT ax = sp(att, inv);                                // (att = P . V' carries lin_s like V')
                    for_tiles<4, WT>(slot, lane, [&](int i, const WT& wt) { pacc[i] = mm(wt, ax, pacc[i]); });
                }
                __syncthreads();                                             // publishes the M . h1 image
                // ---- MGCN (modules.py:243-255): diag(A) (M . h0) + offdiag(A) (M . h1) on this lane's 16 channels ----
                {
                    const f32x16 g = h0 * mblk * adiag + aggregate<J>(EXK, base, h, TAOFF + (bi * J + j) * kTabS);
                    if (nb == 0) pacc[0] += g;
                    if (nb == 1) pacc[1] += g;
                    if (nb == 2) pacc[2] += g;
                    if (nb == 3) pacc[3] += g;
                }
            }
        }
        // ================= phase 2: X_Feat (modules.py:158-177) + residual =================
        {
            const float *c1 = TM1 + j * kTabS, *c2 = TM2 + j * kTabS;      // this token's rows of the hop masks (LDS)
            OT ss[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) ss[i] = sp(pacc[i], inv);
            f32x16 xb[4];                                                    // linearback output; added to the residual once
#pragma unroll
            for (int i = 0; i < 4; ++i) xb[i] = chanvec_L(V, TV_BACKB + 32 * i, h);
#pragma unroll 1
            for (int nb = 0; nb < 4; ++nb) {
                {   // linears[0], output block nb
                    const float* slot = begin_group();
                    f32x16 u0 = chanvec_L(V, TV_LIN0B + 32 * nb, h);
                    for_tiles<4, WT>(slot, lane, [&](int kb, const WT& wt) { u0 = mm(wt, ss[kb], u0); });
                    ex_write(EXK, t, h, u0);
                }
                if (nb == 0) {   // linears[1] (128 -> 16): rows >= 16 of its tiles are zero
                    const float* slot = begin_group();
                    f32x16 u1;
#pragma unroll
                    for (int r = 0; r < 16; ++r) { const int c = kap(r) + 4 * h; u1[r] = c < 16 ? TL1B[bi * 16 + (c & 15)] : 0.f; }
                    for_tiles<4, WT>(slot, lane, [&](int kb, const WT& wt) { u1 = mm(wt, ss[kb], u1); });
                    ex_write(EXV, t, h, u1);
                }
                {   // hop<=1 aggregation of block nb, then its contribution to linearback
                    const float* slot = begin_group();                       // the barrier publishes u0 (and u1)
                    const OT fx = sp(aggregate<J>(EXK, base, h, c1), inv);
                    __builtin_amdgcn_sched_barrier(0);
                    for_tiles<4, WT>(slot, lane, [&](int i, const WT& wt) { xb[i] = mm(wt, fx, xb[i]); });
                }
            }
            {   // hop==2 aggregation of the 16-wide branch: k-block 4 of linearback
                const float* slot = begin_group();
                const OT fx = sp(aggregate<J>(EXV, base, h, c2), inv);
                __builtin_amdgcn_sched_barrier(0);
                for_tiles<4, WT>(slot, lane, [&](int i, const WT& wt) { xb[i] = mm(wt, fx, xb[i]); });
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) { if constexpr (H4) x[i] = fma16(xb[i], inv, x[i]); else x[i] += xb[i]; }
        }
        // ================= phase 3: MLP (modules.py:188-196) + residual =================
        {
            OT y2[4];
            {
                f32x16 yf[4];
                ln128<false>(x, V, TV_N2W, TV_N2B, h, yf);
#pragma unroll
                for (int i = 0; i < 4; ++i) y2[i] = sp(yf[i], 1.0f);
            }
            f32x16 xm[4];                                                    // fc2 output; added to the residual once
#pragma unroll
            for (int i = 0; i < 4; ++i) xm[i] = chanvec_L(V, TV_FC2B + 32 * i, h);
#pragma unroll 1
            for (int c = 0; c < 16; ++c) {
                const float* s1 = begin_group();
                f32x16 hd = chanvec_L(V, TV_FC1B + 32 * c, h);
                for_tiles<4, WT>(s1, lane, [&](int kb, const WT& wt) { hd = mm(wt, y2[kb], hd); });
                if constexpr (H4) gelu_tile_scaled(hd, inv); else gelu_tile(hd);
                const OT hx = sp(hd, inv);
                const float* s2 = begin_group();
                for_tiles<4, WT>(s2, lane, [&](int i, const WT& wt) { xm[i] = mm(wt, hx, xm[i]); });
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) { if constexpr (H4) x[i] = fma16(xm[i], inv, x[i]); else x[i] += xm[i]; }
        }
        if (a.blk_tap && valid) {
            float* dst = a.blk_tap + ((size_t)bi * a.Btap * J + gtok) * kC + 4 * h;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v4;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) v4[jj] = x[kb][4 * g + jj];
                    *reinterpret_cast<f32x4*>(dst + 32 * kb + 8 * g) = v4;
                }
        }
        // next block's vectors: every read of V of this block is done once all waves pass this barrier
        if (bi + 1 < kDepth) {
            __syncthreads();
            stage_vecs(a.blk[bi + 1].vecs, V, tid);
            for (int e = tid; e < J * kC / 4; e += 256) reinterpret_cast<f32x4*>(MT)[e] = reinterpret_cast<const f32x4*>(a.blk[bi + 1].M)[e];
        }
    }
    // ---- tail: feat = GELU(LN(x))  (GAT.py:148-150) ----
    {
        f32x16 d[4], sq[4];
        const float mean = sum64(x) * (1.0f / 128.0f);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) { d[kb] = x[kb] - mean; sq[kb] = d[kb] * d[kb]; }
        const float rstd = 1.0f / sqrtf(sum64(sq) * (1.0f / 128.0f) + 1e-5f);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            f32x16 y = d[kb] * rstd * chanvec_L(a.norm_w, 32 * kb, h) + chanvec_L(a.norm_b, 32 * kb, h);
            gelu_tile(y);
            if (valid) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v4;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) v4[jj] = y[4 * g + jj];
                    *reinterpret_cast<f32x4*>(a.feat + gtok * kC + 32 * kb + 8 * g + 4 * h) = v4;
                }
            }
        }
    }
}

}  // namespace

int gat_tiled_samples_per_wg(int J) { return kTT / J; }

int gat_tiled_prepare_device() {
    const int ldsb = (int)(kTiledLdsFloats * sizeof(float));
    GATOR_HIP_CHECK(hipFuncSetAttribute((const void*)k_gat_tiled<17, false>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
    GATOR_HIP_CHECK(hipFuncSetAttribute((const void*)k_gat_tiled<19, false>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
    GATOR_HIP_CHECK(hipFuncSetAttribute((const void*)k_gat_tiled<17, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
    GATOR_HIP_CHECK(hipFuncSetAttribute((const void*)k_gat_tiled<19, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
    GATOR_HIP_CHECK(hipFuncSetAttribute((const void*)k_gat_tiled<17, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
    GATOR_HIP_CHECK(hipFuncSetAttribute((const void*)k_gat_tiled<19, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
    return GATOR_OK;
}

// pose2d [B,J,2] -> feat [B,J,128]; requires the split-precision weight images (FusedState::gxbuf)
int launch_gat_tiled(gator_ctx* c, FusedState* f, const float* pose2d, int B, float* feat, void* stream, int B_total, bool half16) {
    if (half16 && !f->gat_tiled_h4) return fail(GATOR_EUNSUPPORTED, "the 16-bit encoder needs the four-product weight image (GATOR_GAT_TILED_H4=1, the default)");
    if (!f->gat_x3) return fail(GATOR_EUNSUPPORTED, "the sample-tiled GAT kernel needs the split-precision weights (GATOR_GAT_X3=1)");
    const Weights& w = c->w;
    TiledArgs a;
    a.B = B; a.Btap = B_total > 0 ? B_total : B; a.S = gat_tiled_samples_per_wg(c->J); a.pose2d = pose2d;
    a.gl0_W = w.gl0_W; a.gl0_b = w.gl0_b; a.gn_w = w.gn_w; a.gn_b = w.gn_b; a.gl3_p = f->g_gl3; a.gl3_b = w.gl3_b; a.pos = c->pos_embed;
    a.hop_bias = c->hop_bias; a.adj_diag = c->adj_diag; a.adj_off = c->adj_off; a.m1 = c->mask1; a.m2 = c->mask2;
    a.norm_w = w.norm_w; a.norm_b = w.norm_b;
    for (int i = 0; i < kDepth; ++i) {
        const GatBlockPk& p = f->gblk[i];
        TiledBlk& q = a.blk[i];
        const float* image = f->gat_tiled_h4 ? f->gxbuf_h3 : f->gxbuf;
        auto sel = [&](const float* t) { return image + (size_t)(t - f->gblk[0].qkv) / kTile * kTileX3; };
        q.qkv = sel(p.qkv); q.proj = sel(p.proj); q.w0 = sel(p.w0); q.w1 = sel(p.w1); q.lin0 = sel(p.lin0); q.lin1 = sel(p.lin1);
        q.back = sel(p.back); q.fc1 = sel(p.fc1); q.fc2 = sel(p.fc2);
        q.vecs = f->g_vecs + (size_t)i * 2048;
        q.M = w.blk[i].gcn_M;
        q.lin1_b = w.blk[i].xl1_b;
    }
    a.feat = feat;
    a.blk_tap = nullptr;
    if (c->block_taps) {
        int rc = gat_ensure_blk_tap(c, f, a.Btap);
        if (rc) return rc;
        a.blk_tap = f->blk_tap;
    }
    const int nwg = (B + a.S - 1) / a.S;
    const size_t ldsb = kTiledLdsFloats * sizeof(float);
    a.lin_s = f->gat_tiled_h4 ? std::ldexp(16.0f, f->gat_tiled_wshift) : 1.0f;
    a.lin_inv = 1.0f / a.lin_s;
    if (half16) {
        if (c->J == 17) k_gat_tiled<17, true, true><<<nwg, 256, ldsb, (hipStream_t)stream>>>(a);
        else k_gat_tiled<19, true, true><<<nwg, 256, ldsb, (hipStream_t)stream>>>(a);
    } else if (f->gat_tiled_h4) {
        if (c->J == 17) k_gat_tiled<17, true><<<nwg, 256, ldsb, (hipStream_t)stream>>>(a);
        else k_gat_tiled<19, true><<<nwg, 256, ldsb, (hipStream_t)stream>>>(a);
    } else {
        if (c->J == 17) k_gat_tiled<17, false><<<nwg, 256, ldsb, (hipStream_t)stream>>>(a);
        else k_gat_tiled<19, false><<<nwg, 256, ldsb, (hipStream_t)stream>>>(a);
    }
    GATOR_HIP_CHECK(hipGetLastError());
    return GATOR_OK;
}

}  // namespace gator
