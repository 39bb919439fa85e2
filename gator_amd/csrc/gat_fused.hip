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

namespace gator {
namespace {

constexpr float kLog2eG = 1.4426950408889634f;

struct GatBlockP {   // per-GATBlock packed tiles + reference-layout vectors
    const float *qkv, *proj, *w0, *w1, *lin0, *lin1, *back, *fc1, *fc2;      // packed [NB][KB] tile grids
    const float *mc, *md, *aoffT, *f1b;                                        // packed tables (see GatTables)
    const float *n1w, *n1b, *qkv_b, *proj_b, *gcn_b, *lin0_b, *back_b, *n2w, *n2b, *fc1_b, *fc2_b;
};

struct GatArgs {
    int B, J;
    const float* pose2d;
    const float *gl0_W, *gl0_b, *gn_w, *gn_b, *gl3_W, *gl3_b, *pos;            // embed (reference layout; pos = folded table [J][128])
    const float *biasT, *m1T, *m2T;                                            // [8] tiles, 1 tile, 1 tile
    const float *norm_w, *norm_b, *lifter_p, *lifter_b;                        // lifter_p: [3J][4 kb] tiles
    GatBlockP blk[kDepth];
    float *x_out, *feat;
};

__device__ __forceinline__ float row_sum128(const f32x16 (&x)[4]) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += (x[0][r] + x[1][r]) + (x[2][r] + x[3][r]);
    return s + xhalf(s);
}

// nn.LayerNorm(128); reads the residual stream X (4 tiles in LDS), result in registers
template <bool GELU>
__device__ __forceinline__ void layernorm128(const float* X, const float* __restrict__ w, const float* __restrict__ b, int lane,
                                             f32x16 (&y)[4]) {
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
        y[kb] = x[kb] * rstd * load_chanvec_T(w, 32 * kb, h) + load_chanvec_T(b, 32 * kb, h);
        if (GELU) {
#pragma unroll
            for (int r = 0; r < 16; ++r) y[kb][r] = gelu_f(y[kb][r]);
        }
    }
}

// sum over 4 k-blocks held in registers (xs) with two independent 64-term chains (fp32 accuracy), T- or C-layout output
template <bool CL>
__device__ __forceinline__ f32x16 lin4(const float* __restrict__ Wp, int tile0, const f32x16 (&xs)[4], int lane, f32x16 init) {
    f32x16 a0 = init, a1 = zero16();
    if (CL) {
        a0 = mma_C(load_wtile(Wp, tile0 + 0, lane), xs[0], a0);
        a1 = mma_C(load_wtile(Wp, tile0 + 1, lane), xs[1], a1);
        a0 = mma_C(load_wtile(Wp, tile0 + 2, lane), xs[2], a0);
        a1 = mma_C(load_wtile(Wp, tile0 + 3, lane), xs[3], a1);
    } else {
        a0 = mma_T(load_wtile(Wp, tile0 + 0, lane), xs[0], a0);
        a1 = mma_T(load_wtile(Wp, tile0 + 1, lane), xs[1], a1);
        a0 = mma_T(load_wtile(Wp, tile0 + 2, lane), xs[2], a0);
        a1 = mma_T(load_wtile(Wp, tile0 + 3, lane), xs[3], a1);
    }
    return a0 + a1;
}
// same with the operand tiles read from LDS
__device__ __forceinline__ f32x16 lin4_lds(const float* __restrict__ Wp, int tile0, const float* T, int lane, f32x16 init) {
    f32x16 a0 = init, a1 = zero16();
    a0 = mma_T(load_wtile(Wp, tile0 + 0, lane), load_block(T + 0 * kTile, lane), a0);
    a1 = mma_T(load_wtile(Wp, tile0 + 1, lane), load_block(T + 1 * kTile, lane), a1);
    a0 = mma_T(load_wtile(Wp, tile0 + 2, lane), load_block(T + 2 * kTile, lane), a0);
    a1 = mma_T(load_wtile(Wp, tile0 + 3, lane), load_block(T + 3 * kTile, lane), a1);
    return a0 + a1;
}

__global__ __launch_bounds__(256, 1) void k_gat(const GatArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* X = lds;                       // residual stream, 4 tiles (T-layout)
    float* R = lds + 4 * kTile;           // 16 tiles of phase-local scratch
    float *AT = R, *SB = R + 4 * kTile, *FB = R + 8 * kTile, *F1P = R + 12 * kTile, *HB = R;
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6, h = lane >> 5, J = a.J;

    // ---------------- embedding: GraphLinear(2->64) . GroupNorm(4,64) . GELU . GraphLinear(64->128) + pos (GAT.py:135-144)
    {
        float* hbuf = R;                  // [64][32]
        float* gbuf = R + 2 * kTile;      // [64][32]
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
        for (int e = t; e < 64 * 32; e += 256) {
            const int c = e >> 5;
            gbuf[e] = gelu_f((hbuf[e] - stat[(c >> 4) * 2]) * stat[(c >> 4) * 2 + 1] * a.gn_w[c] + a.gn_b[c]);
        }
        __syncthreads();
        // x[token][n] -> X tiles: X[kb][g][lane][j] <-> token = lane&31, ch = 32kb + 8g + 4(lane>>5) + j
        for (int e = t; e < 4 * kTile; e += 256) {
            const int j4 = e & 3, ln = (e >> 2) & 63, g = (e >> 8) & 3, kb = e >> 10;
            const int tok = ln & 31, n = 32 * kb + 8 * g + 4 * (ln >> 5) + j4;
            float v = 0.f;
            if (tok < J) {
                v = a.gl3_b[n];
                for (int c = 0; c < 64; ++c) v += a.gl3_W[n * 64 + c] * gbuf[c * 32 + tok];
                v += a.pos[tok * kC + n];
            }
            X[e] = v;
        }
        __syncthreads();
    }

    f32x16 xw = load_block(X + wave * kTile, lane);       // this wave's block of the residual stream
    f32x16 ident;                                          // identity as a B operand: I[t_out = lane&31][j = kap(r)+4h]
#pragma unroll
    for (int r = 0; r < 16; ++r) ident[r] = (kap(r) + 4 * h == (lane & 31)) ? 1.f : 0.f;

    for (int bi = 0; bi < kDepth; ++bi) {
        const GatBlockP& w = a.blk[bi];
        f32x16 g_out;
        {
            f32x16 y[4];
            layernorm128<false>(X, w.n1w, w.n1b, lane, y);
            // ---- Attention (modules.py:121-138): wave owns heads 2*wave, 2*wave+1 ----
            {
                const f32x16 q = lin4<false>(w.qkv, wave * 4, y, lane, load_chanvec_T(w.qkv_b, 32 * wave, h));
                const f32x16 k = lin4<false>(w.qkv, (4 + wave) * 4, y, lane, load_chanvec_T(w.qkv_b, 128 + 32 * wave, h));
                const f32x16 v = lin4<true>(w.qkv, (8 + wave) * 4, y, lane, zero16());
                const float vb = w.qkv_b[256 + 32 * wave + (lane & 31)];
                f32x16 sa = zero16(), sb = zero16();
#pragma unroll
                for (int r = 0; r < 8; ++r) sa = GATOR_MFMA(k[r], q[r], sa);            // head 2w:   channels 0..15 of the block
#pragma unroll
                for (int r = 8; r < 16; ++r) sb = GATOR_MFMA(k[r], q[r], sb);           // head 2w+1: channels 16..31
                const f32x16 ba = load_block(a.biasT + (size_t)(2 * wave) * kTile, lane);
                const f32x16 bb = load_block(a.biasT + (size_t)(2 * wave + 1) * kTile, lane);
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
                f32x16 O = zero16();
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float vv = v[r] + vb;
                    O = GATOR_MFMA(lo ? vv : 0.f, sa[r] * ia, O);                          // rows (channels) 0..15  <- head 2w
                    O = GATOR_MFMA(lo ? 0.f : vv, sb[r] * ib, O);                          // rows 16..31            <- head 2w+1
                }
                store_block(AT + wave * kTile, lane, O);
            }
            // ---- MGCN (modules.py:243-255): h_k = y @ W[k]; out = diag(A)(M.h0) + offdiag(A)(M.h1) + bias ----
            {
                f32x16 h0 = lin4<true>(w.w0, wave * 4, y, lane, zero16());
                f32x16 h1 = lin4<true>(w.w1, wave * 4, y, lane, zero16());
                h0 = h0 * load_block(w.md + (size_t)wave * kTile, lane);                    // diag(A)[t] * M[t][n] * h0[t][n]
                h1 = h1 * load_block(w.mc + (size_t)wave * kTile, lane);                    // M[t][n] * h1[t][n]
                const f32x16 aoff = load_block(w.aoffT, lane);
                g_out = load_chanvec_T(w.gcn_b, 32 * wave, h);
#pragma unroll
                for (int r = 0; r < 16; ++r) g_out = GATOR_MFMA(h1[r], aoff[r], g_out);  // sum_j (M.h1)[j][n] * Aoff[t][j]
#pragma unroll
                for (int r = 0; r < 16; ++r) g_out = GATOR_MFMA(h0[r], ident[r], g_out); // C-layout -> T-layout of the diagonal term
            }
        }
        __syncthreads();
        {   // proj + (attention + MGCN) sum  -> SB
            const f32x16 acc = lin4_lds(w.proj, wave * 4, AT, lane, load_chanvec_T(w.proj_b, 32 * wave, h));
            store_block(SB + wave * kTile, lane, acc + g_out);
        }
        __syncthreads();
        {   // ---- X_Feat (modules.py:158-177) ----
            f32x16 s[4];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) s[kb] = load_block(SB + kb * kTile, lane);
            const f32x16 u0 = lin4<true>(w.lin0, wave * 4, s, lane, zero16());
            const float b0 = w.lin0_b[32 * wave + (lane & 31)];
            const f32x16 m1 = load_block(a.m1T, lane), m2 = load_block(a.m2T, lane);
            f32x16 f0 = zero16();
#pragma unroll
            for (int r = 0; r < 16; ++r) f0 = GATOR_MFMA(u0[r] + b0, m1[r], f0);           // hop<=1 aggregation
            store_block(FB + wave * kTile, lane, f0);
            // linears[1] (128->16): this wave contributes k-block `wave`; partial hop-2 aggregation, summed by the reader
            f32x16 u1 = mma_C(load_wtile(w.lin1, wave, lane), s[wave], zero16());
            f32x16 f1 = zero16();
#pragma unroll
            for (int r = 0; r < 16; ++r) f1 = GATOR_MFMA(u1[r], m2[r], f1);
            store_block(F1P + wave * kTile, lane, f1);
        }
        __syncthreads();
        {   // linearback(144->128) + residual
            f32x16 acc = lin4_lds(w.back, wave * 5, FB, lane, load_chanvec_T(w.back_b, 32 * wave, h));
            f32x16 f1 = load_block(w.f1b, lane);                                            // rowsum(m2)[t] * linears[1].bias[n]
#pragma unroll
            for (int q = 0; q < 4; ++q) f1 += load_block(F1P + q * kTile, lane);
            const WTile wt = load_wtile(w.back, wave * 5 + 4, lane);
#pragma unroll
            for (int g = 0; g < 2; ++g)                                                     // channels 128..143 only (r < 8)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = GATOR_MFMA(wt.g[g][j], f1[4 * g + j], acc);
            xw += acc;
            store_block(X + wave * kTile, lane, xw);
        }
        __syncthreads();
        {   // ---- MLP (modules.py:188-196): fc1 + GELU -> HB (16 tiles), fc2 + residual ----
            f32x16 y2[4];
            layernorm128<false>(X, w.n2w, w.n2b, lane, y2);
#pragma unroll 1
            for (int q = 0; q < 4; ++q) {
                const int nb = 4 * wave + q;
                f32x16 hd = lin4<false>(w.fc1, nb * 4, y2, lane, load_chanvec_T(w.fc1_b, 32 * nb, h));
#pragma unroll
                for (int r = 0; r < 16; ++r) hd[r] = gelu_f(hd[r]);
                store_block(HB + nb * kTile, lane, hd);
            }
        }
        __syncthreads();
        {
            f32x16 acc0 = load_chanvec_T(w.fc2_b, 32 * wave, h), acc1 = zero16(), acc2 = zero16(), acc3 = zero16();   // 4 chains x 128
#pragma unroll 1
            for (int kb = 0; kb < 16; kb += 4) {
                acc0 = mma_T(load_wtile(w.fc2, wave * 16 + kb, lane), load_block(HB + kb * kTile, lane), acc0);
                acc1 = mma_T(load_wtile(w.fc2, wave * 16 + kb + 1, lane), load_block(HB + (kb + 1) * kTile, lane), acc1);
                acc2 = mma_T(load_wtile(w.fc2, wave * 16 + kb + 2, lane), load_block(HB + (kb + 2) * kTile, lane), acc2);
                acc3 = mma_T(load_wtile(w.fc2, wave * 16 + kb + 3, lane), load_block(HB + (kb + 3) * kTile, lane), acc3);
            }
            xw += (acc0 + acc1) + (acc2 + acc3);
            store_block(X + wave * kTile, lane, xw);     // every wave finished reading X (LN2) before the HB barrier
        }
        __syncthreads();
    }
    // ---------------- tail: LN -> GELU -> feat ; lifter Linear(128J -> 3J)  (GAT.py:148-152)
    f32x16 ft[4];
    layernorm128<true>(X, a.norm_w, a.norm_b, lane, ft);
    const int tok = lane & 31;
    if (tok < J) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v4;
#pragma unroll
            for (int j = 0; j < 4; ++j) v4[j] = ft[wave][4 * g + j];
            *reinterpret_cast<f32x4*>(a.feat + ((size_t)b * J + tok) * kC + 32 * wave + 8 * g + 4 * h) = v4;
        }
    }
    for (int o = wave; o < 3 * J; o += 4) {
        float s = 0.f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const f32x16 wl = load_block(a.lifter_p + ((size_t)o * 4 + kb) * kTile, lane);
            float p = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) p += ft[kb][r] * wl[r];
            s += p;
        }
        for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d);
        if (lane == 0) a.x_out[(size_t)b * 3 * J + o] = s + a.lifter_b[o];
    }
}

}  // namespace

int launch_gat(gator_ctx* c, FusedState* f, const float* pose2d, int B, float* x_out, float* feat, void* stream) {
    GatArgs a;
    const Weights& w = c->w;
    a.B = B; a.J = c->J; a.pose2d = pose2d;
    a.gl0_W = w.gl0_W; a.gl0_b = w.gl0_b; a.gn_w = w.gn_w; a.gn_b = w.gn_b; a.gl3_W = w.gl3_W; a.gl3_b = w.gl3_b; a.pos = c->pos_embed;
    a.biasT = f->g_biasT; a.m1T = f->g_m1T; a.m2T = f->g_m2T;
    a.norm_w = w.norm_w; a.norm_b = w.norm_b; a.lifter_p = f->g_lifter; a.lifter_b = w.lifter_b;
    for (int i = 0; i < kDepth; ++i) {
        const GatBlockW& r = w.blk[i];
        const GatBlockPk& p = f->gblk[i];
        GatBlockP& q = a.blk[i];
        q.qkv = p.qkv; q.proj = p.proj; q.w0 = p.w0; q.w1 = p.w1; q.lin0 = p.lin0; q.lin1 = p.lin1; q.back = p.back; q.fc1 = p.fc1; q.fc2 = p.fc2;
        q.mc = p.mc; q.md = p.md; q.aoffT = p.aoffT; q.f1b = p.f1b;
        q.n1w = r.n1w; q.n1b = r.n1b; q.qkv_b = r.qkv_b; q.proj_b = r.proj_b; q.gcn_b = r.gcn_bias; q.lin0_b = r.xl0_b; q.back_b = r.xlb_b;
        q.n2w = r.n2w; q.n2b = r.n2b; q.fc1_b = r.fc1_b; q.fc2_b = r.fc2_b;
    }
    a.x_out = x_out; a.feat = feat;
    constexpr size_t kLds = 20 * kTile * sizeof(float);     // 80 KB: two workgroups per CU
    static bool attr = false;
    if (!attr) {
        GATOR_HIP_CHECK(hipFuncSetAttribute((const void*)k_gat, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds));
        attr = true;
    }
    k_gat<<<B, 256, kLds, (hipStream_t)stream>>>(a);
    GATOR_HIP_CHECK(hipGetLastError());
    return GATOR_OK;
}

}  // namespace gator
