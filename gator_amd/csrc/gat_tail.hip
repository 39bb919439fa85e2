// Tail of the GAT encoder as two batched launches: the lifter Linear(128J -> 3J) (lib/models/GAT.py:151-152) and the MDR joint
// tokens with their per-layer cross-attention K/V (lib/models/MDR.py:130-134,37-38,65).
//
// Inside k_gat these steps run once per sample on a workgroup that owns its CU alone (one wave per SIMD): the lifter re-reads the
// whole 444 KB weight for every sample and every L2 round trip is exposed -- 70k of the kernel's 400k cycles at B=256.  Here
//   k_gat_lifter  treats the lifter as the GEMM it is over 32 samples at a time: workgroup (sample tile mt, joint j) contracts
//                 joint j's 128 channels for both 32-row output tiles on the fp32-input MFMA (exact fp32 products, fixed order)
//                 and writes a partial tile; the weight is read once per 32 samples;
//   k_gat_joint   one workgroup per sample: sums the J partials in a fixed order (+ bias) -> pose3d, then builds the joint
//                 tokens and the three layers' K/V operand tiles (the body of k_gat's former epilogue, 12 jobs over 4 waves).
// Every output depends only on its own sample (an MFMA column never mixes samples), so results stay independent of the batch.
#include "fused_common.h"
#include "fused_state.h"
#include "x3_common.h"

namespace gator {
namespace {

struct LifterArgs {
    const float *feat, *w;      // feat [B][J][128]; lifter.weight [3J][128J] as the reference stores it
    float* part;                // [MT][J][2][kTile] partial out^T tiles (row = output o within the tile, column = sample)
    int B, J;
};

__global__ __launch_bounds__(128) void k_gat_lifter(const LifterArgs a) {
    const int lane = threadIdx.x & 63, nt = threadIdx.x >> 6, h = lane >> 5, J = a.J;
    const int mt = blockIdx.x / J, j = blockIdx.x % J;
    const int o = 32 * nt + (lane & 31), smp = 32 * mt + (lane & 31);
    const bool ok_o = o < 3 * J, ok_s = smp < a.B;
    const size_t krow = (size_t)kC * J;
    // lane (row, half h) feeds k = 128 j + 8 q + 4 h + i to MFMA step i of chunk q -- the same k on the A and the B side
    const f32x4* wp = reinterpret_cast<const f32x4*>(a.w + (size_t)(ok_o ? o : 0) * krow + (size_t)j * kC + 4 * h);
    const f32x4* fp = reinterpret_cast<const f32x4*>(a.feat + ((size_t)(ok_s ? smp : 0) * J + j) * kC + 4 * h);
    f32x4 wv[16], fv[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) { wv[q] = wp[2 * q]; fv[q] = fp[2 * q]; }      // everything in flight at once
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    f32x16 acc = zero16();
#pragma unroll
    for (int q = 0; q < 16; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = GATOR_MFMA(ok_o ? wv[q][i] : 0.f, ok_s ? fv[q][i] : 0.f, acc);
    store_block(a.part + (((size_t)mt * J + j) * 2 + nt) * kTile, lane, acc);
}

struct JointTailArgs {
    const float *pose2d, *feat, *part, *lifter_b;
    float *x_out, *jkv;           // jkv == nullptr: lifter only (stand-alone GAT entry point)
    unsigned* mdr_ctr;            // non-null: zero k_mdr_persist's tickets and completion counts for the launch that follows (mdr_fused.hip)
    const float *jf5, *jf_p, *jf_b, *posj_T;       // get_joint_feature: columns 0..4 as [5][64], columns 5..132 packed [2][4], bias
    const float *j_n1w[3], *j_n1b[3], *j_wk_p[3], *j_wv_p[3];
    int B, J;
    int x2;                       // K/V tiles as two fp16 planes of 16 x value (mdr_fused.hip: cross_attention_head_x2)
};

__device__ __forceinline__ float row_sum32x2(const f32x16& a, const f32x16& b) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += a[r] + b[r];
    return s + xhalf(s);
}

__global__ __launch_bounds__(256) void k_gat_joint(const JointTailArgs a) {
    __shared__ float XO[64];
    __shared__ __attribute__((aligned(16))) float JF[2 * kTile];
    // every per-channel vector the joint-token part needs, fetched by one vector load per thread at the start (in flight during the
    // lifter sums; published by the barrier that follows them).  Through the scalar cache, as before, each of the 18 vectors was an
    // exposed round trip in the middle of the chain: 6.6 of the launch's 20 us.
    enum { VJ_JFB = 0, VJ_JF5 = 64, VJ_N1W = 384, VJ_N1B = 576, VJ_TOTAL = 768 };      // n1w / n1b: [3 layers][64]
    __shared__ __attribute__((aligned(16))) float VJ[VJ_TOTAL];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, h = lane >> 5, J = a.J, tok = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    if (a.mdr_ctr) {
        // every persistent launch's counter block (tickets, error flag, completion counts): the whole region, dealt over the workgroups
        for (size_t i = (size_t)b * 256 + t; i < mdr_ctr_words(a.B); i += (size_t)a.B * 256) a.mdr_ctr[i] = 0u;
    }
    // operands of the joint-token part are requested first: their latency hides behind the partial sums below
    const int tkj = tok < J ? tok : 0;
    WTile jw[4], kw[2];
    f32x16 posj, ft[4];
    float p2x = 0.f, p2y = 0.f;
    auto job_tile = [&](int job, int i) {
        const int li = job >> 2, kv = (job >> 1) & 1, nb = job & 1;
        return load_wtile(kv ? a.j_wv_p[li] : a.j_wk_p[li], nb * 2 + i, lane);
    };
    if (a.jkv) {
        if (t < VJ_TOTAL / 4) {
            const int off = 4 * t;
            const float* src = off < VJ_JF5 ? a.jf_b + off : off < VJ_N1W ? a.jf5 + (off - VJ_JF5)
                               : off < VJ_N1B ? a.j_n1w[(off - VJ_N1W) >> 6] + (off & 63) : a.j_n1b[(off - VJ_N1B) >> 6] + (off & 63);
            reinterpret_cast<f32x4*>(VJ)[t] = *reinterpret_cast<const f32x4*>(src);
        }
        if (wave < 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) jw[i] = load_wtile(a.jf_p, wave * 4 + i, lane);
            posj = load_block(a.posj_T + (size_t)wave * kTile, lane);
            p2x = a.pose2d[((size_t)b * J + tkj) * 2];
            p2y = a.pose2d[((size_t)b * J + tkj) * 2 + 1];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)         // feat as T-layout blocks: v[4g+j] = feat[tok][32kb + 8g + 4h + j]
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v4 = *reinterpret_cast<const f32x4*>(a.feat + ((size_t)b * J + tkj) * kC + 32 * kb + 8 * g + 4 * h);
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) ft[kb][4 * g + jj] = tok < J ? v4[jj] : 0.f;
                }
        }
        kw[0] = job_tile(wave * 3, 0);
        kw[1] = job_tile(wave * 3, 1);
    }
    // lifter: x_out[o] = bias[o] + sum_j partial_j[o][sample], j ascending (GAT.py:151-152)
    if (t < 3 * J) {
        const int mt = b >> 5, s = b & 31, nt = t >> 5, ro = t & 31;
        const float* p = a.part + ((size_t)mt * J * 2 + nt) * kTile + (((ro >> 3) * 64 + ((ro >> 2) & 1) * 32 + s) * 4 + (ro & 3));
        float pj[19];                       // all partials in flight at once (a rolled loop pays one L2 round trip per joint)
#pragma unroll
        for (int j = 0; j < 19; ++j) pj[j] = p[(size_t)(j < J ? j : 0) * 2 * kTile];
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 19; ++j) acc += j < J ? pj[j] : 0.f;
        acc += a.lifter_b[t];
        a.x_out[(size_t)b * 3 * J + t] = acc;
        XO[t] = acc;
    }
    if (!a.jkv) return;
    __syncthreads();
    // jf = Linear(133->64)(cat(pose2d, pose3d/1000, feat)) + pos_j   (GATOR.py:19, MDR.py:130-134)
    if (wave < 2) {
        f32x16 acc = chanvec_lds(VJ, VJ_JFB + 32 * wave, h) + posj, ac1 = zero16();
        const float pin[5] = {p2x, p2y, XO[tkj * 3] / 1000.f, XO[tkj * 3 + 1] / 1000.f, XO[tkj * 3 + 2] / 1000.f};
#pragma unroll
        for (int i = 0; i < 5; ++i) acc += chanvec_lds(VJ, VJ_JF5 + i * 64 + 32 * wave, h) * pin[i];
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
    // per LBF layer: k = wk(LN1(jf)), v = wv(LN1(jf)) as MFMA operand tiles; 12 jobs (layer, k|v, channel block) over the 4 waves
#pragma unroll 1
    for (int job = wave * 3; job < wave * 3 + 3; ++job) {
        const int li = job >> 2, kv = (job >> 1) & 1, nb = job & 1;
        const int jn = job + 1 < wave * 3 + 3 ? job + 1 : job;       // next job's weight tiles in flight during this one
        const WTile nw0 = job_tile(jn, 0), nw1 = job_tile(jn, 1);
        asm volatile("" ::: "memory");
        f32x16 fz[2];
        fz[0] = d0 * rstd * chanvec_lds(VJ, VJ_N1W + 64 * li, h) + chanvec_lds(VJ, VJ_N1B + 64 * li, h);
        fz[1] = d1 * rstd * chanvec_lds(VJ, VJ_N1W + 64 * li + 32, h) + chanvec_lds(VJ, VJ_N1B + 64 * li + 32, h);
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
        if (a.x2) x2_store(out, lane, x2_split(r0 * 16.0f)); else store_block(out, lane, r0);
        kw[0] = nw0;
        kw[1] = nw1;
    }
}

}  // namespace

size_t gat_tail_part_floats(int B, int J) { return (size_t)((B + 31) / 32) * J * 2 * kTile; }

int launch_gat_tail(gator_ctx* c, FusedState* f, const float* pose2d, const float* feat, int B, float* x_out, void* stream, bool joint, bool zero_ctr) {
    const Weights& w = c->w;
    const int J = c->J, MT = (B + 31) / 32;
    hipStream_t st = (hipStream_t)stream;
    LifterArgs la{feat, w.lifter_w, f->lpart, B, J};
    k_gat_lifter<<<MT * J, 128, 0, st>>>(la);
    JointTailArgs a{};
    a.pose2d = pose2d; a.feat = feat; a.part = f->lpart; a.lifter_b = w.lifter_b; a.x_out = x_out; a.B = B; a.J = J;
    a.jkv = nullptr;
    a.mdr_ctr = nullptr;
    a.x2 = f->mdr_x3 == 2;
    if (joint) {
        if (f->mdr_persist != 0 && zero_ctr) { a.mdr_ctr = f->mdr_ctr; f->mdr_ctr_clean = true; }      // (!zero_ctr: k_gat8's fused tail zeroes them for the whole forward)
        a.jkv = f->jkv; a.jf5 = f->jfeat5; a.jf_p = f->jfeat128_p; a.jf_b = w.jfeat_b; a.posj_T = f->posj_T;
        for (int i = 0; i < 3; ++i) { a.j_n1w[i] = w.lay[i].n1w; a.j_n1b[i] = w.lay[i].n1b; a.j_wk_p[i] = f->lay[i].wk; a.j_wv_p[i] = f->lay[i].wv; }
    }
    k_gat_joint<<<B, 256, 0, st>>>(a);
    GATOR_HIP_CHECK(hipGetLastError());
    return GATOR_OK;
}

}  // namespace gator
