// GAT graph-aware transformer encoder (lib/models/GAT.py:133-152, GATBlock :33-43) as ONE kernel launch.
//
// One workgroup = one sample (J <= 32 joint tokens = one 32-token MFMA tile), 4 waves.  Every linear is split over the
// waves by output-channel block (wave w owns channels 32w..32w+31 of each 128-wide layer, 3 of the 12 qkv blocks, 4 of the
// 16 MLP-hidden blocks) and runs on fp32-input MFMA 32x32x2 with the weights streamed from L2 in operand-packed 1 KiB
// wave loads.  Activations live in LDS (80 KB) as 4 KB tiles in exactly the register layout of fused_common.h, so a wave
// re-reads another wave's output as a ready-made MFMA operand with conflict-free ds_read_b128.  The graph operators are
// LDS/register-resident adjacency x feature products: the hop-1 / hop-2 masks of X_Feat (modules.py:158-177) and the
// symmetrised dense MGCN adjacency (modules.py:243-255) are 32x32 constant B-operand tiles; the per-sample J x J
// attention (modules.py:121-138) keeps scores, softmax and P.V in registers, two heads per wave.
#include "fused_common.h"
#include "fused_state.h"
#include "x3_common.h"

#include <cstdlib>

namespace gator {
namespace {

constexpr float kLog2eG = 1.4426950408889634f;

struct GatBlockP {   // per-GATBlock packed tiles + reference-layout vectors
    const float *qkv, *proj, *w0, *w1, *lin0, *lin1, *back, *fc1, *fc2;      // packed [NB][KB] tile grids (fp32 tiles, or X3 tiles)
    const float* back32;                                                       // linearback always also as fp32 tiles (its 16-wide k tail)
    const float *mc, *mdT, *aoffT, *f1b;                                       // packed tables: M (channel on lane), diag(A).M (token on lane), offdiag(A)^T, hop-2 bias
    const float* vecs;                                                         // V_TOTAL floats, order above
};

struct GatArgs {
    int B, J;
    const float* pose2d;
    const float *gl0_W, *gl0_b, *gn_w, *gn_b, *gl3_p, *gl3_b, *posT;           // embed: gl3_p packed [4][2]; posT = 4 folded T-layout tiles
    const float *biasT, *m1T, *m2T;                                            // [8] tiles, 1 tile, 1 tile
    const float *norm_w, *norm_b, *lifter_w, *lifter_b;                        // lifter_w: the reference weight [3J][128J], row-major
    GatBlockP blk[kDepth];
    float *x_out, *feat;
    // optional epilogue (full forward): the MDR joint tokens and their per-layer cross-attention K/V (MDR.py:130-134,37-38,65),
    // so that no separate launch has to re-read pose_combine.  jkv == nullptr: skipped (stand-alone GAT entry point).
    float* jkv;
    const float *jf5, *jf_p, *jf_b, *posj_T;       // get_joint_feature: columns 0..4 as [5][64], columns 5..132 packed [2][4], bias
    const float *j_n1w[3], *j_n1b[3], *j_wk_p[3], *j_wv_p[3];
    int pf_share, pf_n, pf_block;   // L2 warm-up of the next block's weights: bytes per workgroup, workgroups per XCD, bytes per block (0: off)
    int tapB;                       // batch stride of blk_tap (the whole batch of the call)
    float* blk_tap;                 // gator_enable_block_taps: residual stream after every GATBlock, [depth][B][J][128] (else nullptr)
#ifdef GATOR_DIAG
    unsigned long long* stamps;     // diagnostic build only (libgator_hip_diag.so): per-phase cycle sums of workgroup 0, wave 0
#endif
};

// Per-block small vectors (LayerNorm weights, biases) are staged in LDS one block ahead: 2048 floats in this order.
// A global load issued at its point of use would queue behind the weight prefetch in flight (vmcnt is in-order).
enum { V_N1W = 0, V_N1B = 128, V_QKVB = 256, V_PROJB = 640, V_GCNB = 768, V_LIN0B = 896, V_BACKB = 1024, V_N2W = 1152,
       V_N2B = 1280, V_FC1B = 1408, V_FC2B = 1920, V_TOTAL = 2048 };
// per-channel vector in T-layout from LDS: v[r] = V[off + kap(r) + 4h]  (two distinct addresses per ds_read_b128: broadcast)
__device__ __forceinline__ f32x16 load_chanvec_L(const float* V, int off, int h) {
    f32x16 v;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(V + off + 8 * g + 4 * h);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[4 * g + j] = t[j];
    }
    return v;
}

__device__ __forceinline__ float row_sum32x2(const f32x16& a, const f32x16& b) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += a[r] + b[r];
    return s + xhalf(s);
}
__device__ __forceinline__ float row_sum128(const f32x16 (&x)[4]) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += (x[0][r] + x[1][r]) + (x[2][r] + x[3][r]);
    return s + xhalf(s);
}

// nn.LayerNorm(128); reads the residual stream X (4 tiles in LDS), result in registers
template <bool GELU>
__device__ __forceinline__ void layernorm128(const float* X, const float* w, const float* b, int lane, f32x16 (&y)[4]) {
    const int h = lane >> 5;
    f32x16 x[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) x[kb] = load_block(X + kb * kTile, lane);
    const float mean = row_sum128(x) * (1.0f / 128.0f);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) x[kb] = x[kb] - mean;
    f32x16 sq[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) sq[kb] = x[kb] * x[kb];
    const float rstd = 1.0f / sqrtf(row_sum128(sq) * (1.0f / 128.0f) + 1e-5f);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        y[kb] = x[kb] * rstd * load_chanvec_L(w, 32 * kb, h) + load_chanvec_L(b, 32 * kb, h);
        if (GELU) {
            gelu_tile(y[kb]);
        }
    }
}

// Fence for the hand-built pipeline: memory ops may not cross (keeps a prefetch from sinking to its first use) and -- on the
// fp32-input MFMA path -- the machine scheduler may not move anything across (measured: phase 1 of a block 51k -> 36k cycles).
// On the split-precision path the MFMA share is small enough that letting the scheduler interleave one chunk's VALU work
// (GELU, operand splits) with the next chunk's MFMAs wins instead (k_gat 0.288 -> 0.274 ms), so only the memory fence stays.
#define GATOR_PIN()                                                     \
    do {                                                                \
        asm volatile("" ::: "memory");                                  \
        if (!X3K) __builtin_amdgcn_sched_barrier(0);                     \
    } while (0)

// one weight tile as B operand (C-layout output), two interleaved 8-step chains
__device__ __forceinline__ f32x16 dot16_C(const WTile& w, const f32x16& x) {
    f32x16 e = zero16(), o = zero16();
#pragma unroll
    for (int g = 0; g < 4; g += 2)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            e = GATOR_MFMA(x[4 * g + j], w.g[g][j], e);
            o = GATOR_MFMA(x[4 * (g + 1) + j], w.g[g + 1][j], o);
        }
    return e + o;
}

// ---- operand abstraction: X = false -> fp32-input MFMA on packed fp32 tiles; X = true -> split-precision bf16 MFMA on X3 tiles
template <bool X> struct Op;
template <> struct Op<false> { typedef f32x16 T; typedef WTile W; static constexpr int kT = kTile; };
template <> struct Op<true> { typedef X3 T; typedef X3 W; static constexpr int kT = kTileX3; };

template <bool X> __device__ __forceinline__ typename Op<X>::T mkop(const f32x16& v);
template <> __device__ __forceinline__ f32x16 mkop<false>(const f32x16& v) { return v; }
template <> __device__ __forceinline__ X3 mkop<true>(const f32x16& v) { return x3_split(v); }
template <bool X> __device__ __forceinline__ typename Op<X>::T ldop(const float* p, int lane);
template <> __device__ __forceinline__ f32x16 ldop<false>(const float* p, int lane) { return load_block(p, lane); }
template <> __device__ __forceinline__ X3 ldop<true>(const float* p, int lane) { return x3_load(p, lane); }
template <bool X> __device__ __forceinline__ void stop(float* p, int lane, const f32x16& v) {
    if constexpr (X) x3_store(p, lane, x3_split(v)); else store_block(p, lane, v);
}
template <bool X> __device__ __forceinline__ typename Op<X>::W ldw1(const float* __restrict__ Wp, int tile, int lane);
template <> __device__ __forceinline__ WTile ldw1<false>(const float* __restrict__ Wp, int tile, int lane) { return load_wtile(Wp, tile, lane); }
template <> __device__ __forceinline__ X3 ldw1<true>(const float* __restrict__ Wp, int tile, int lane) { return x3_load(Wp + (size_t)tile * kTileX3, lane); }
__device__ __forceinline__ f32x16 dotC(const WTile& w, const f32x16& x) { return dot16_C(w, x); }
__device__ __forceinline__ f32x16 dotC(const X3& w, const X3& x) { return x3_mma(x, w, zero16()); }

// one 4-tile weight group (K = 128 of one 32-wide output block): 64 VGPRs fp32, 96 VGPRs X3
template <bool X> struct Grp { typename Op<X>::W t[4]; };
template <bool X> __device__ __forceinline__ Grp<X> ldg4(const float* __restrict__ Wp, int tile0, int lane) {
    Grp<X> g;
#pragma unroll
    for (int i = 0; i < 4; ++i) g.t[i] = ldw1<X>(Wp, tile0 + i, lane);
    return g;
}
// Rolling weight buffer.  The wave holds ONE 4-tile group; each tile slot is refilled with the matching tile of the NEXT group
// in the fixed order q,k,v,W0,W1,proj,lin0,back,fc1[4],fc2[4] right after the MFMAs that read it are queued, so a load is
// always one group-time ahead of its use (the same distance as two alternating buffers, at half the registers).
// fp32: two independent 64-term chains; X3: one chain (the bf16 MFMA sums 16 products internally, dependent issue is back to back).
template <bool CL>
__device__ __forceinline__ f32x16 lin4r(Grp<false>& g, const f32x16 (&xs)[4], f32x16 init, const float* __restrict__ nW, int nt, int lane) {
    f32x16 a0 = init, a1 = zero16();
    if (CL) mma2_C(g.t[0], xs[0], a0, g.t[1], xs[1], a1); else mma2_T(g.t[0], xs[0], a0, g.t[1], xs[1], a1);
    g.t[0] = load_wtile(nW, nt + 0, lane);
    g.t[1] = load_wtile(nW, nt + 1, lane);
    if (CL) mma2_C(g.t[2], xs[2], a0, g.t[3], xs[3], a1); else mma2_T(g.t[2], xs[2], a0, g.t[3], xs[3], a1);
    g.t[2] = load_wtile(nW, nt + 2, lane);
    g.t[3] = load_wtile(nW, nt + 3, lane);
    return a0 + a1;
}
template <bool CL>
__device__ __forceinline__ f32x16 lin4r(Grp<true>& g, const X3 (&xs)[4], f32x16 init, const float* __restrict__ nW, int nt, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        init = CL ? x3_mma(xs[i], g.t[i], init) : x3_mma(g.t[i], xs[i], init);
        g.t[i] = x3_load(nW + (size_t)(nt + i) * kTileX3, lane);
    }
    return init;
}
// same with the 4 operand tiles read from LDS (consecutive tiles at T)
__device__ __forceinline__ f32x16 lin4l(Grp<false>& g, const float* T, int lane, f32x16 init, const float* __restrict__ nW, int nt) {
    f32x16 a0 = init, a1 = zero16();
    mma2_T(g.t[0], load_block(T + 0 * kTile, lane), a0, g.t[1], load_block(T + 1 * kTile, lane), a1);
    g.t[0] = load_wtile(nW, nt + 0, lane);
    g.t[1] = load_wtile(nW, nt + 1, lane);
    mma2_T(g.t[2], load_block(T + 2 * kTile, lane), a0, g.t[3], load_block(T + 3 * kTile, lane), a1);
    g.t[2] = load_wtile(nW, nt + 2, lane);
    g.t[3] = load_wtile(nW, nt + 3, lane);
    return a0 + a1;
}
__device__ __forceinline__ f32x16 lin4l(Grp<true>& g, const float* T, int lane, f32x16 init, const float* __restrict__ nW, int nt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        init = x3_mma(g.t[i], x3_load(T + i * kTileX3, lane), init);
        g.t[i] = x3_load(nW + (size_t)(nt + i) * kTileX3, lane);
    }
    return init;
}

// TAIL = false: the kernel ends with `feat`; the lifter and the MDR joint tokens run as batched launches (gat_tail.hip)
constexpr int kGatLdsFloatsX3 = 4 * kTile + 16 * kTileX3 + 2048;

// LDS-DMA through inline asm (hipcc drains the builtin form with vmcnt(0) before the next LDS access, see gat_tiled.hip)
__device__ __forceinline__ void glds16(const float* gsrc, const float* lds_dst) {
    unsigned keep;
    const unsigned dst = (unsigned)(unsigned long long)lds_dst;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

template <bool X3K, bool TAIL>
__global__ __launch_bounds__(256, 1) void k_gat(const GatArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* X = lds;                       // residual stream, 4 tiles (T-layout)
    constexpr int TA = Op<X3K>::kT;       // floats per operand tile in LDS (fp32 block, or X3 planes)
    float* R = lds + 4 * kTile;           // phase-local scratch: AT | SB | FB (4 operand tiles each) | F1P (4 fp32 tiles), aliased by HB (16)
    float *AT = R, *SB = R + 4 * TA, *FB = R + 8 * TA, *F1P = R + 12 * TA, *HB = R;
    float* V = R + 16 * TA;               // per-block vectors (V_TOTAL floats)
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, h = lane >> 5, J = a.J;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);      // provably wave-uniform (scalar loads, scalar addressing)
#ifdef GATOR_DIAG
    unsigned long long st_acc[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last = a.stamps ? clock64() : 0;
#define GAT_STAMP(i)                                      \
    __builtin_amdgcn_sched_barrier(0);                    \
    if (a.stamps) {                                       \
        const unsigned long long now_ = clock64();        \
        st_acc[i] += now_ - st_last;                      \
        st_last = now_;                                   \
    }
#else
// Production build: no stamps, but the phase boundaries stay scheduling fences -- measured: without them hipcc interleaves the
// phases' loads and MFMAs differently and k_gat runs 265 -> 364 us at B=256.
#define GAT_STAMP(i) __builtin_amdgcn_sched_barrier(0);
#endif

    // ---------------- embedding: GraphLinear(2->64) . GroupNorm(4,64) . GELU . GraphLinear(64->128) + pos (GAT.py:135-144)
    {
        float* hbuf = R;                  // [64][32]
        float* gt = R + 2 * kTile;        // 2 T-layout tiles
        float* stat = R + 4 * kTile;      // [4][2]
        const float* p = a.pose2d + (size_t)b * J * 2;
        for (int e = t; e < 64 * 32; e += 256) {
            const int c = e >> 5, j = e & 31;
            hbuf[e] = j < J ? a.gl0_W[c * 2] * p[j * 2] + a.gl0_W[c * 2 + 1] * p[j * 2 + 1] + a.gl0_b[c] : 0.f;
        }
        __syncthreads();
        {   // wave g -> GroupNorm group g (16 channels x J tokens), two-pass
            float s = 0.f;
            for (int e = lane; e < 16 * 32; e += 64) s += ((e & 31) < J) ? hbuf[wave * 512 + e] : 0.f;
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            const float mean = s / (16.0f * J);
            float q = 0.f;
            for (int e = lane; e < 16 * 32; e += 64) {
                const float d = hbuf[wave * 512 + e] - mean;
                q += ((e & 31) < J) ? d * d : 0.f;
            }
            for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
            if (lane == 0) { stat[wave * 2] = mean; stat[wave * 2 + 1] = 1.0f / sqrtf(q / (16.0f * J) + 1e-5f); }
        }
        __syncthreads();
        // GroupNorm affine + GELU, written as two T-layout operand tiles gt[kb][g][lane][j] <-> token lane&31, channel 32kb+8g+4h+j
        for (int e = t; e < 2 * kTile; e += 256) {
            const int j4 = e & 3, ln = (e >> 2) & 63, g = (e >> 8) & 3, kb = e >> 10;
            const int tok = ln & 31, c = 32 * kb + 8 * g + 4 * (ln >> 5) + j4;
            gt[e] = tok < J ? gelu_f((hbuf[c * 32 + tok] - stat[(c >> 4) * 2]) * stat[(c >> 4) * 2 + 1] * a.gn_w[c] + a.gn_b[c]) : 0.f;
        }
        __syncthreads();
        for (int e = t; e < V_TOTAL / 4; e += 256)
            reinterpret_cast<f32x4*>(V)[e] = reinterpret_cast<const f32x4*>(a.blk[0].vecs)[e];
        // GraphLinear(64->128) on the MFMA: wave w -> channel block w;  + pos_id_embed + pos_num_embed (folded T-layout tiles)
        {
            f32x16 acc = load_chanvec_S(a.gl3_b, 32 * wave, h) + load_block(a.posT + (size_t)wave * kTile, lane), ac1 = zero16();
            mma2_T(load_wtile(a.gl3_p, wave * 2 + 0, lane), load_block(gt, lane), acc, load_wtile(a.gl3_p, wave * 2 + 1, lane),
                   load_block(gt + kTile, lane), ac1);
            store_block(X + wave * kTile, lane, acc + ac1);
        }
        __syncthreads();
    }
    GAT_STAMP(0)
    f32x16 xw = load_block(X + wave * kTile, lane);       // this wave's block of the residual stream

    // Software pipeline of the weight stream.  One wave per SIMD means nobody hides this wave's L2 latency, so the next
    // 4-tile weight group is always in flight while the current one feeds the MFMAs (rolling buffer G, see lin4r).
    // GATOR_PIN keeps the compiler from sinking a prefetch back down to its first use.
    Grp<X3K> G = ldg4<X3K>(a.blk[0].qkv, wave * 4, lane);
    GATOR_PIN();
    // L2 warm-up.  All workgroups of an XCD walk the same 1.6 MB of weights per block in step, so every 4-tile group is an L2 miss
    // for all of them at once and arrives with Infinity-Cache latency; with 96 KB in flight per CU that caps the stream near
    // 45 GB/s per CU.  Each of the XCD's workgroups therefore pulls 1/n-th of the NEXT block's weights into the XCD's L2 one block
    // ahead (LDS-DMA into a dummy 4 KB window: no registers; one burst at the top of a block, so the in-order vmcnt delays only
    // the first group wait behind it).  Measured on one box: k_gat 235.8 -> 228.9 us (-3 %): the stream is NOT what holds the
    // phases at twice their MFMA time - that is the wave's own MFMA -> VALU -> LDS-exchange dependency chain.
    auto l2_warm = [&](const float* first) {
        if (!X3K || a.pf_share == 0) return;
        const int rank = (b >> 3) % a.pf_n;
        const char* src = reinterpret_cast<const char*>(first) + (size_t)rank * a.pf_share;
        const int left = a.pf_block - rank * a.pf_share;
        float* dummy = lds + kGatLdsFloatsX3;
        for (int off = 0; off < a.pf_share; off += 4096)
            if (off + t * 16 < left) glds16(reinterpret_cast<const float*>(src + off + t * 16), dummy + wave * 256);   // (M0 = the wave's 1 KB window)
    };
    l2_warm(a.blk[0].qkv);

    for (int bi = 0; bi < kDepth; ++bi) {
        const GatBlockP& w = a.blk[bi];
        const GatBlockP& wn = a.blk[bi + 1 < kDepth ? bi + 1 : bi];     // next block (prefetch target; harmless re-load at the end)
        if (bi + 1 < kDepth) l2_warm(wn.qkv);
        f32x16 g_out;
        f32x4 vn0, vn1;
        {
            // next block's vectors: requested now (behind nothing that is needed soon), written to LDS after the last read of V
            vn0 = reinterpret_cast<const f32x4*>(wn.vecs)[t];
            vn1 = reinterpret_cast<const f32x4*>(wn.vecs)[256 + t];
            GATOR_PIN();
            const f32x16 bq = load_chanvec_L(V, V_QKVB + 32 * wave, h), bk = load_chanvec_L(V, V_QKVB + 128 + 32 * wave, h);
            const float vb = V[V_QKVB + 256 + 32 * wave + (lane & 31)];
            typename Op<X3K>::T y[4];
            {
                f32x16 yf[4];
                layernorm128<false>(X, V + V_N1W, V + V_N1B, lane, yf);
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) y[kb] = mkop<X3K>(yf[kb]);
            }
            GAT_STAMP(10)
            // ---- Attention (modules.py:121-138): wave owns heads 2*wave, 2*wave+1 ----
            const f32x16 q = lin4r<false>(G, y, bq, w.qkv, (4 + wave) * 4, lane);
            GATOR_PIN();
            GAT_STAMP(11)
            const f32x16 k = lin4r<false>(G, y, bk, w.qkv, (8 + wave) * 4, lane);
            const f32x16 ba = load_block(a.biasT + (size_t)(2 * wave) * kTile, lane);
            const f32x16 bb = load_block(a.biasT + (size_t)(2 * wave + 1) * kTile, lane);
            GATOR_PIN();
            GAT_STAMP(12)
            const f32x16 v = lin4r<true>(G, y, zero16(), w.w0, wave * 4, lane);
            GATOR_PIN();
            GAT_STAMP(13)
            {
                f32x16 sa = zero16(), sb = zero16();
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    sa = GATOR_MFMA(k[r], q[r], sa);                                    // head 2w:   channels 0..15 of the block
                    sb = GATOR_MFMA(k[r + 8], q[r + 8], sb);                            // head 2w+1: channels 16..31
                }
                float ma = -1e30f, mb = -1e30f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool ok = kap(r) + 4 * h < J;
                    sa[r] = ok ? (sa[r] * 0.25f + ba[r]) * kLog2eG : -1e30f;             // q k^T * head_dim**-0.5 + bias
                    sb[r] = ok ? (sb[r] * 0.25f + bb[r]) * kLog2eG : -1e30f;
                    ma = fmaxf(ma, sa[r]);
                    mb = fmaxf(mb, sb[r]);
                }
                ma = fmaxf(ma, xhalf(ma));
                mb = fmaxf(mb, xhalf(mb));
                float la = 0.f, lb = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    sa[r] = __builtin_amdgcn_exp2f(sa[r] - ma);
                    sb[r] = __builtin_amdgcn_exp2f(sb[r] - mb);
                    la += sa[r];
                    lb += sb[r];
                }
                la += xhalf(la);
                lb += xhalf(lb);
                const float ia = 1.0f / la, ib = 1.0f / lb;
                const bool lo = (lane & 31) < 16;
                f32x16 O = zero16(), Ob = zero16();
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float vv = v[r] + vb;
                    O = GATOR_MFMA(lo ? vv : 0.f, sa[r] * ia, O);                          // rows (channels) 0..15  <- head 2w
                    Ob = GATOR_MFMA(lo ? 0.f : vv, sb[r] * ib, Ob);                        // rows 16..31            <- head 2w+1
                }
                stop<X3K>(AT + wave * TA, lane, O + Ob);
            }
            GAT_STAMP(14)
            // ---- MGCN (modules.py:243-255): h_k = y @ W[k]; out = diag(A)(M.h0) + offdiag(A)(M.h1) + bias ----
            const f32x16 mdt = load_block(w.mdT + (size_t)wave * kTile, lane), mct = load_block(w.mc + (size_t)wave * kTile, lane);
            const f32x16 aoff = load_block(w.aoffT, lane);
            const f32x16 bg = load_chanvec_L(V, V_GCNB + 32 * wave, h), bp = load_chanvec_L(V, V_PROJB + 32 * wave, h);
            GATOR_PIN();
            f32x16 h0 = lin4r<false>(G, y, zero16(), w.w1, wave * 4, lane);        // token on the lane: its term is token-wise
            GATOR_PIN();
            GAT_STAMP(15)
            f32x16 h1 = lin4r<true>(G, y, zero16(), w.proj, wave * 4, lane);
            GATOR_PIN();
            GAT_STAMP(16)
            h0 = h0 * mdt;                                                                  // diag(A)[t] * M[t][n] * h0[t][n]
            h1 = h1 * mct;                                                                  // M[t][n] * h1[t][n]
            // sum_j (M.h1)[j][n] * Aoff[t][j] as one MFMA product (two interleaved 8-step chains); the diagonal term needs no
            // product: h0 was formed with the token on the lane (round 1 moved it there through an identity MFMA product, 16 of 32)
            g_out = dot16(h1, aoff, bg) + h0;
            GAT_STAMP(17)
            __syncthreads();
            GAT_STAMP(2)
            // proj + (attention + MGCN) sum  -> SB
            const f32x16 acc = lin4l(G, AT, lane, bp, w.lin0, wave * 4);
            GATOR_PIN();
            stop<X3K>(SB + wave * TA, lane, acc + g_out);
        }
        // small operands of the X_Feat phase, requested before the barrier
        const typename Op<X3K>::W wl1 = ldw1<X3K>(w.lin1, wave, lane);
        const WTile wb4 = load_wtile(w.back32, wave * 5 + 4, lane);
        const f32x16 m1 = load_block(a.m1T, lane), m2 = load_block(a.m2T, lane), f1bias = load_block(w.f1b, lane);
        const float b0 = V[V_LIN0B + 32 * wave + (lane & 31)];
        const f32x16 bback = load_chanvec_L(V, V_BACKB + 32 * wave, h);
        GATOR_PIN();
        __syncthreads();
        GAT_STAMP(3)
        {   // ---- X_Feat (modules.py:158-177) ----
            typename Op<X3K>::T s[4];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) s[kb] = ldop<X3K>(SB + kb * TA, lane);
            const f32x16 u0 = lin4r<true>(G, s, zero16(), w.back, wave * 5, lane);
            GATOR_PIN();
            // linears[1] (128->16): this wave contributes k-block `wave`; partial hop-2 aggregation, summed by the reader
            const f32x16 u1 = dotC(wl1, ldop<X3K>(SB + wave * TA, lane));                    // (s[wave] would index registers dynamically)
            f32x16 f0 = zero16(), f1 = zero16();
            dot16x2(u0 + b0, m1, f0, u1, m2, f1);                                           // hop<=1 and hop==2 aggregations
            stop<X3K>(FB + wave * TA, lane, f0);
            store_block(F1P + wave * kTile, lane, f1);
        }
        __syncthreads();
        GAT_STAMP(4)
        {   // linearback(144->128) + residual
            f32x16 acc = lin4l(G, FB, lane, bback, w.fc1, (4 * wave + 0) * 4);
            GATOR_PIN();
            f32x16 f1 = f1bias;                                                             // rowsum(m2)[t] * linears[1].bias[n]
#pragma unroll
            for (int q = 0; q < 4; ++q) f1 += load_block(F1P + q * kTile, lane);
            f32x16 acb = zero16();
#pragma unroll
            for (int j = 0; j < 4; ++j) {                                                   // channels 128..143 only (r < 8)
                acc = GATOR_MFMA(wb4.g[0][j], f1[j], acc);
                acb = GATOR_MFMA(wb4.g[1][j], f1[4 + j], acb);
            }
            xw += acc + acb;
            store_block(X + wave * kTile, lane, xw);
        }
        __syncthreads();
        GAT_STAMP(5)
        {   // ---- MLP (modules.py:188-196): fc1 + GELU -> HB (16 tiles), fc2 + residual ----
            typename Op<X3K>::T y2[4];
            {
                f32x16 yf[4];
                layernorm128<false>(X, V + V_N2W, V + V_N2B, lane, yf);
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) y2[kb] = mkop<X3K>(yf[kb]);
            }
            f32x16 hd = lin4r<false>(G, y2, load_chanvec_L(V, V_FC1B + 32 * (4 * wave + 0), h), w.fc1, (4 * wave + 1) * 4, lane);
            GATOR_PIN();
            gelu_tile(hd);
            stop<X3K>(HB + (4 * wave + 0) * TA, lane, hd);
            hd = lin4r<false>(G, y2, load_chanvec_L(V, V_FC1B + 32 * (4 * wave + 1), h), w.fc1, (4 * wave + 2) * 4, lane);
            GATOR_PIN();
            gelu_tile(hd);
            stop<X3K>(HB + (4 * wave + 1) * TA, lane, hd);
            hd = lin4r<false>(G, y2, load_chanvec_L(V, V_FC1B + 32 * (4 * wave + 2), h), w.fc1, (4 * wave + 3) * 4, lane);
            GATOR_PIN();
            gelu_tile(hd);
            stop<X3K>(HB + (4 * wave + 2) * TA, lane, hd);
            hd = lin4r<false>(G, y2, load_chanvec_L(V, V_FC1B + 32 * (4 * wave + 3), h), w.fc2, wave * 16 + 0, lane);
            GATOR_PIN();
            gelu_tile(hd);
            stop<X3K>(HB + (4 * wave + 3) * TA, lane, hd);
        }
        const f32x16 bfc2 = load_chanvec_L(V, V_FC2B + 32 * wave, h);
        GAT_STAMP(6)
        __syncthreads();
        GAT_STAMP(7)
        reinterpret_cast<f32x4*>(V)[t] = vn0;          // every read of this block's vectors happened before the barrier above
        reinterpret_cast<f32x4*>(V)[256 + t] = vn1;
        {   // fc2: four independent 128-product chains (one per group of 4 hidden blocks)
            const f32x16 c0 = lin4l(G, HB + 0 * TA, lane, bfc2, w.fc2, wave * 16 + 4);
            GATOR_PIN();
            const f32x16 c1 = lin4l(G, HB + 4 * TA, lane, zero16(), w.fc2, wave * 16 + 8);
            GATOR_PIN();
            const f32x16 c2 = lin4l(G, HB + 8 * TA, lane, zero16(), w.fc2, wave * 16 + 12);
            GATOR_PIN();
            const f32x16 c3 = lin4l(G, HB + 12 * TA, lane, zero16(), wn.qkv, wave * 4);               // next block's q group
            GATOR_PIN();
            xw += (c0 + c1) + (c2 + c3);
            store_block(X + wave * kTile, lane, xw);     // every wave finished reading X (LN2) before the HB barrier
            if (a.blk_tap && (lane & 31) < J) {          // debug tap (off in timed runs): this wave's channel block of the block output
                float* dst = a.blk_tap + (((size_t)bi * a.tapB + b) * J + (lane & 31)) * kC + 32 * wave + 4 * h;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v4[j] = xw[4 * g + j];
                    *reinterpret_cast<f32x4*>(dst + 8 * g) = v4;
                }
            }
        }
        __syncthreads();
        GAT_STAMP(8)
    }
    // ---------------- tail: LN -> GELU -> feat ; lifter Linear(128J -> 3J)  (GAT.py:148-152)
    const int tok = lane & 31;
    {
        f32x16 x[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) x[kb] = load_block(X + kb * kTile, lane);
        const float mean = row_sum128(x) * (1.0f / 128.0f);
        f32x16 sq[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) { x[kb] = x[kb] - mean; sq[kb] = x[kb] * x[kb]; }
        const float rstd = 1.0f / sqrtf(row_sum128(sq) * (1.0f / 128.0f) + 1e-5f);
        f32x16 mine = (xw - mean) * rstd * load_chanvec_S(a.norm_w, 32 * wave, h) + load_chanvec_S(a.norm_b, 32 * wave, h);
        gelu_tile(mine);      // each wave finishes only its own channel block
        store_block(R + wave * kTile, lane, mine);
        if (tok < J) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v4;
#pragma unroll
                for (int j = 0; j < 4; ++j) v4[j] = mine[4 * g + j];
                *reinterpret_cast<f32x4*>(a.feat + ((size_t)b * J + tok) * kC + 32 * wave + 8 * g + 4 * h) = v4;
            }
        }
    }
    if constexpr (!TAIL) return;
    float* XO = R + 6 * kTile;            // x_out of this sample (3J floats), for the joint-token epilogue
    GAT_STAMP(18)
    // lifter (GAT.py:151-152): x_out[o] = <feat flattened [J*128], W[o]> + b[o].  The weight rows are read as the reference
    // stores them ([3J][128J] row-major; one float2 per lane = 512 B per wave load, no padding to 32 tokens); each lane
    // gathers the 2J feat values that face its weight elements from LDS once and keeps them for all of the wave's outputs.
    constexpr int kJmax = 19;
    const float* Wl = a.lifter_w + (size_t)lane * 2;
    const size_t wrow = (size_t)kC * J;
    __syncthreads();
    f32x16 ft[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) ft[kb] = load_block(R + kb * kTile, lane);      // T-layout blocks for the joint-token epilogue
    f32x2 fv[kJmax];
#pragma unroll
    for (int k = 0; k < kJmax; ++k) {
        const int i = (k * 64 + lane) * 2, tk = i >> 7, ch = i & 127, cb = ch & 31;        // flat index -> (token, channel)
        fv[k] = *reinterpret_cast<const f32x2*>(R + (ch >> 5) * kTile + (((cb >> 3) * 64 + tk + 32 * ((cb >> 2) & 1)) * 4 + (cb & 3)));
        if (k >= J) fv[k] = f32x2(0.f);
    }
    // Operands of the joint-token epilogue are requested here: one wave per SIMD means nothing else hides their latency,
    // and vmcnt retires in order, so they are complete for free once the lifter loop's own loads have been waited for.
    const int tkj = tok < J ? tok : 0;
    WTile jw[4], kw[2];
    f32x16 posj;
    float p2x = 0.f, p2y = 0.f;
    auto job_tile = [&](int job, int i) {
        const int li = job >> 2, kv = (job >> 1) & 1, nb = job & 1;
        return load_wtile(kv ? a.j_wv_p[li] : a.j_wk_p[li], nb * 2 + i, lane);
    };
    if (a.jkv) {
        if (wave < 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) jw[i] = load_wtile(a.jf_p, wave * 4 + i, lane);
            posj = load_block(a.posj_T + (size_t)wave * kTile, lane);
            p2x = a.pose2d[((size_t)b * J + tkj) * 2];
            p2y = a.pose2d[((size_t)b * J + tkj) * 2 + 1];
        }
        kw[0] = job_tile(wave * 3, 0);
        kw[1] = job_tile(wave * 3, 1);
    }
    GATOR_PIN();
    // four outputs of the wave per trip (o = wave + 4i): all 4 x J row loads are in flight together, so the L2 latency is paid
    // once per trip instead of once per output (one wave per SIMD: nobody else hides it)
    for (int o0 = wave; o0 < 3 * J; o0 += 16) {
        f32x2 wr[4][kJmax];
        float bo[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int o = o0 + 4 * u < 3 * J ? o0 + 4 * u : o0;
            bo[u] = a.lifter_b[o];
#pragma unroll
            for (int k = 0; k < kJmax; ++k)      // unconditional (a load under a runtime condition is branched around and waited for singly)
                wr[u][k] = *reinterpret_cast<const f32x2*>(Wl + (size_t)o * wrow + (k < J ? k : 0) * 128);
        }
        float sv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int k = 0; k < kJmax; ++k) {    // fv[k] = 0 for k >= J
                s0 += fv[k][0] * wr[u][k][0];
                s1 += fv[k][1] * wr[u][k][1];
            }
            sv[u] = s0 + s1;
        }
        for (int d = 32; d > 0; d >>= 1) {
#pragma unroll
            for (int u = 0; u < 4; ++u) sv[u] += __shfl_xor(sv[u], d);
        }
        if (lane == 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (o0 + 4 * u < 3 * J) {
                    const float xo = sv[u] + bo[u];
                    a.x_out[(size_t)b * 3 * J + o0 + 4 * u] = xo;
                    XO[o0 + 4 * u] = xo;
                }
        }
    }
    GAT_STAMP(19)
    if (a.jkv) {
        // ---------------- MDR joint tokens: jf = Linear(133->64)(cat(pose2d, pose3d/1000, feat)) + pos_j ; per LBF layer
        // k = wk(LN1(jf)), v = wv(LN1(jf)) written as MFMA operand tiles (same layout as k_mdr_joint)
        __syncthreads();                                            // XO complete
        float* JF = R + 4 * kTile;                                  // 2 tiles
        if (wave < 2) {
            f32x16 acc = load_chanvec_S(a.jf_b, 32 * wave, h) + posj, ac1 = zero16();
            const int tk = tkj;
            const float pin[5] = {p2x, p2y, XO[tk * 3] / 1000.f, XO[tk * 3 + 1] / 1000.f, XO[tk * 3 + 2] / 1000.f};   // GATOR.py:19
#pragma unroll
            for (int i = 0; i < 5; ++i) acc += load_chanvec_S(a.jf5 + i * 64, 32 * wave, h) * pin[i];
            mma2_T(jw[0], ft[0], acc, jw[1], ft[1], ac1);
            mma2_T(jw[2], ft[2], acc, jw[3], ft[3], ac1);
            store_block(JF + wave * kTile, lane, acc + ac1);
        }
        __syncthreads();
        f32x16 jf[2];
        jf[0] = load_block(JF, lane);
        jf[1] = load_block(JF + kTile, lane);
        const float mean = (row_sum32x2(jf[0], jf[1])) * (1.0f / 64.0f);
        const f32x16 d0 = jf[0] - mean, d1 = jf[1] - mean;
        const float rstd = 1.0f / sqrtf(row_sum32x2(d0 * d0, d1 * d1) * (1.0f / 64.0f) + 1e-5f);
        // 12 jobs (layer, k|v, channel block) over the 4 waves
#pragma unroll 1
        for (int job = wave * 3; job < wave * 3 + 3; ++job) {
            const int li = job >> 2, kv = (job >> 1) & 1, nb = job & 1;
            const int jn = job + 1 < wave * 3 + 3 ? job + 1 : job;       // next job's weight tiles in flight during this one
            const WTile nw0 = job_tile(jn, 0), nw1 = job_tile(jn, 1);
            GATOR_PIN();
            f32x16 fz[2];
            fz[0] = d0 * rstd * load_chanvec_S(a.j_n1w[li], 0, h) + load_chanvec_S(a.j_n1b[li], 0, h);
            fz[1] = d1 * rstd * load_chanvec_S(a.j_n1w[li], 32, h) + load_chanvec_S(a.j_n1b[li], 32, h);
            float* out = a.jkv + (((size_t)b * 3 + li) * 4 + kv * 2 + nb) * kTile;
            f32x16 r0 = zero16(), r1 = zero16();
            if (kv == 0) {
                mma2_T(kw[0], fz[0], r0, kw[1], fz[1], r1);
                r0 += r1;
                if (tok >= J) r0 = zero16();                       // joints >= J: zero rows (masked in the softmax anyway)
            } else {
                mma2_C(kw[0], fz[0], r0, kw[1], fz[1], r1);
                r0 += r1;
#pragma unroll
                for (int r = 0; r < 16; ++r) r0[r] = (kap(r) + 4 * h < J) ? r0[r] : 0.f;
            }
            store_block(out, lane, r0);
            kw[0] = nw0;
            kw[1] = nw1;
        }
    }
    GAT_STAMP(9)
#ifdef GATOR_DIAG
    if (a.stamps && b == 0 && t == 0)
        for (int i = 0; i < 20; ++i) a.stamps[i] = st_acc[i];
#endif
}

}  // namespace

constexpr size_t kGatLds = (20 * kTile + 2048) * sizeof(float);                   // 88 KB
constexpr size_t kGatLdsX3 = (kGatLdsFloatsX3 + 1024) * sizeof(float);            // 120 KB + the 4 KB landing window of the L2 warm-up

// Dynamic-LDS opt-in of the kernels, per DEVICE: called from fused_create_gat with the ctx's device current (a function
// attribute set on one device does not carry to another, and a process may hold contexts on several).
int gat_prepare_device() {
    GATOR_HIP_CHECK(hipFuncSetAttribute((const void*)k_gat<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGatLds));
    GATOR_HIP_CHECK(hipFuncSetAttribute((const void*)k_gat<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGatLdsX3));
    GATOR_HIP_CHECK(hipFuncSetAttribute((const void*)k_gat<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGatLds));
    GATOR_HIP_CHECK(hipFuncSetAttribute((const void*)k_gat<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGatLdsX3));
    return GATOR_OK;
}

// debug buffer of the per-block taps, [depth][B][J][128], allocated on first use only
int gat_ensure_blk_tap(gator_ctx* c, FusedState* f, int B) {
    if (B > f->blk_tap_cap) {
        if (f->blk_tap) { GATOR_HIP_CHECK(hipDeviceSynchronize()); GATOR_HIP_CHECK(hipFree(f->blk_tap)); f->blk_tap = nullptr; }
        GATOR_HIP_CHECK(hipMalloc(&f->blk_tap, (size_t)kDepth * B * c->J * kC * sizeof(float)));
        f->blk_tap_cap = B;
    }
    c->set_tap(TAP_GAT_BLOCKS, f->blk_tap, (int64_t)kDepth * B * c->J * kC);
    return GATOR_OK;
}

int launch_gat(gator_ctx* c, FusedState* f, const float* pose2d, int B, float* x_out, float* feat, void* stream, bool joint_epilogue,
               int B_total, int tap_row0, bool half16, float* tail_jkv) {
    GatArgs a;
    const Weights& w = c->w;
    a.B = B; a.J = c->J; a.pose2d = pose2d;
    a.gl0_W = w.gl0_W; a.gl0_b = w.gl0_b; a.gn_w = w.gn_w; a.gn_b = w.gn_b; a.gl3_p = f->g_gl3; a.gl3_b = w.gl3_b; a.posT = f->g_posT;
    a.biasT = f->g_biasT; a.m1T = f->g_m1T; a.m2T = f->g_m2T;
    a.norm_w = w.norm_w; a.norm_b = w.norm_b; a.lifter_w = w.lifter_w; a.lifter_b = w.lifter_b;
    for (int i = 0; i < kDepth; ++i) {
        const GatBlockW& r = w.blk[i];
        const GatBlockPk& p = f->gblk[i];
        GatBlockP& q = a.blk[i];
        // X3 tile grids mirror the fp32 ones tile for tile (fused_create_gat): same tile index, 1.5x the tile size
        auto sel = [&](const float* t) { return f->gat_x3 ? f->gxbuf + (size_t)(t - f->gblk[0].qkv) / kTile * kTileX3 : t; };
        q.qkv = sel(p.qkv); q.proj = sel(p.proj); q.w0 = sel(p.w0); q.w1 = sel(p.w1); q.lin0 = sel(p.lin0); q.lin1 = sel(p.lin1);
        q.back = sel(p.back); q.fc1 = sel(p.fc1); q.fc2 = sel(p.fc2);
        q.back32 = p.back;
        q.mc = p.mc; q.mdT = p.mdT; q.aoffT = p.aoffT; q.f1b = p.f1b;
        q.vecs = f->g_vecs + (size_t)i * 2048;
        (void)r;
    }
    a.x_out = x_out; a.feat = feat;
    a.pf_share = a.pf_n = a.pf_block = 0;
    static const bool l2warm = [] { const char* e = getenv("GATOR_GAT_L2WARM"); return !(e && atoi(e) == 0); }();     // default on; =0 for A/B
    if (f->gat_x3 && l2warm) {
        a.pf_block = (int)((a.blk[1].qkv - a.blk[0].qkv) * sizeof(float));
        a.pf_n = std::min(32, (B + 7) / 8);
        a.pf_share = ((a.pf_block + a.pf_n - 1) / a.pf_n + 4095) / 4096 * 4096;
    }
    a.jkv = nullptr;
    if (joint_epilogue) {
        a.jkv = f->jkv; a.jf5 = f->jfeat5; a.jf_p = f->jfeat128_p; a.jf_b = w.jfeat_b; a.posj_T = f->posj_T;
        for (int i = 0; i < 3; ++i) { a.j_n1w[i] = w.lay[i].n1w; a.j_n1b[i] = w.lay[i].n1b; a.j_wk_p[i] = f->lay[i].wk; a.j_wv_p[i] = f->lay[i].wv; }
    }
    a.blk_tap = nullptr;
    a.tapB = B;
    if (c->block_taps) {
        const int Bt = B_total > 0 ? B_total : B;
        int rc = gat_ensure_blk_tap(c, f, Bt);
        if (rc) return rc;
        a.blk_tap = f->blk_tap + (size_t)tap_row0 * c->J * kC;      // this launch's samples start at row tap_row0 of the batch
        a.tapB = Bt;
    }
#ifdef GATOR_DIAG
    a.stamps = nullptr;
    static const bool want_stamps = getenv("GATOR_GAT_STAMPS") != nullptr;
    unsigned long long* d_st = nullptr;
    if (want_stamps) {
        GATOR_HIP_CHECK(hipMalloc(&d_st, 20 * sizeof(unsigned long long)));
        a.stamps = d_st;
    }
#endif
    // (a launch that covers only part of a batch -- the remainder behind the sample-tiled kernel -- always leaves the tail to the
    // batched launches, which then run once over the whole batch)
    const bool split_tail = joint_epilogue && (f->gat_split_tail || (B_total > 0 && B_total != B));
    if (split_tail && f->gat_x3 && f->gat8 && f->g8stream) {                             // the two-role form (gat_roles.hip)
#ifdef GATOR_DIAG
        if (d_st) GATOR_HIP_CHECK(hipFree(d_st));                                         // (it prints its own stamps)
#endif
        // tail_jkv: k_gat8 runs the lifter + joint tokens of ITS samples as its epilogue (and zeroes the MDR counters of the whole forward)
        return launch_gat8(c, f, pose2d, B, feat, stream, B_total, tap_row0, half16, tail_jkv ? x_out : nullptr, tail_jkv, tail_jkv ? (B_total > 0 ? B_total : B) : 0);
    }
    if (split_tail) {
        if (f->gat_x3) k_gat<true, false><<<B, 256, kGatLdsX3, (hipStream_t)stream>>>(a);
        else k_gat<false, false><<<B, 256, kGatLds, (hipStream_t)stream>>>(a);
    } else {
        if (f->gat_x3) k_gat<true, true><<<B, 256, kGatLdsX3, (hipStream_t)stream>>>(a);
        else k_gat<false, true><<<B, 256, kGatLds, (hipStream_t)stream>>>(a);
    }
    GATOR_HIP_CHECK(hipGetLastError());

#ifdef GATOR_DIAG
    if (d_st) {     // diagnostic build: synchronous read-back
        unsigned long long hst[20];
        GATOR_HIP_CHECK(hipMemcpy(hst, d_st, sizeof(hst), hipMemcpyDeviceToHost));
        GATOR_HIP_CHECK(hipFree(d_st));
        static const char* nm[10] = {"embed", "ln1+qkv+attn+mgcn", "barrier1", "proj+prefetch+barrier2", "xfeat+barrier3",
                                     "back+barrier4", "ln2+fc1+gelu", "barrier5", "fc2+barrier6", "tail"};
        unsigned long long tot = 0;
        for (int i = 0; i < 10; ++i) tot += hst[i];
        tot += hst[18] + hst[19];
        hst[1] += hst[10] + hst[11] + hst[12] + hst[13] + hst[14] + hst[15] + hst[16] + hst[17];
        fprintf(stderr, "[k_gat stamps, wg0 wave0, B=%d] total %llu cycles:", B, tot);
        for (int i = 0; i < 10; ++i) fprintf(stderr, " %s=%llu", nm[i], hst[i]);
        fprintf(stderr, " | phase1 detail: ln1=%llu q=%llu k=%llu v=%llu attn=%llu h0=%llu h1=%llu mgcn=%llu", hst[10], hst[11], hst[12], hst[13], hst[14], hst[15], hst[16], hst[17]);
        fprintf(stderr, " | tail detail: ln+gelu+feat=%llu lifter=%llu joint tokens=%llu", hst[18], hst[19], hst[9]);
        fprintf(stderr, "\n");
    }
#endif
    return GATOR_OK;
}

}  // namespace gator
