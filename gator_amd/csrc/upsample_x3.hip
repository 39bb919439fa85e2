// fp32-accurate vertex regressor (upsample_conv + template, MDR.py:167-168) on the bf16 MFMA by split precision.
//
// Every fp32 operand is split EXACTLY into three bf16 planes, x = hi + mid + lo (8+8+8 significand bits; each remainder is
// exactly representable, so nothing is lost).  A product a*w is then the sum of nine exact partial products; the six largest
// (hi*hi | hi*mid, mid*hi | mid*mid, hi*lo, lo*hi) are computed on v_mfma_f32_32x32x16_bf16 -- 6 x 32 cycles per 16-deep
// k-step against 8 x 64 cycles for the fp32-input MFMA (2.7x) -- and the dropped terms are below 2^-23 of the product.
// hi*hi goes to one accumulator, the five small-magnitude terms to another, so they are not rounded away against the big sum.
#include "fused_common.h"
#include "fused_state.h"

namespace gator {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int kS16 = 28;
#define GATOR_MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r = x - (float)h;
    m = (__bf16)r;
    l = (__bf16)(r - (float)m);
}

// dst[plane][tap][ob][s][lane][j] <- split3( w[32ob + (lane&31)][16s + 8(lane>>5) + j][tap] )
__global__ void k_pack_up_x3(const float* __restrict__ w, __bf16* __restrict__ dst, int64_t plane_elems) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= plane_elems) return;
    const int j = e & 7, lane = (e >> 3) & 63;
    int64_t r = e >> 9;
    const int s = r % kS16; r /= kS16;
    const int ob = r % kOB;
    const int tap = (int)(r / kOB);
    const int o = 32 * ob + (lane & 31), c = 16 * s + 8 * (lane >> 5) + j;
    const float x = (o < kNV && c < kV) ? w[((int64_t)o * kV + c) * 3 + tap] : 0.f;
    __bf16 h, m, l;
    split3(x, h, m, l);
    dst[e] = h; dst[plane_elems + e] = m; dst[2 * plane_elems + e] = l;
}
// vcp[plane][mt][l'][s][lane][j]
__global__ void k_pack_vc_x3(const float* __restrict__ vc, int B, __bf16* __restrict__ vcp, int64_t n, int64_t plane_elems) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int j = e & 7, lane = (e >> 3) & 63;
    int64_t r = e >> 9;
    const int s = r % kS16; r /= kS16;
    const int lp = r % 3;
    const int mt = (int)(r / 3);
    const int smp = 32 * mt + (lane & 31), c = 16 * s + 8 * (lane >> 5) + j;
    const float x = (smp < B && c < kV) ? vc[((int64_t)smp * kV + c) * 3 + lp] : 0.f;
    __bf16 h, m, l;
    split3(x, h, m, l);
    vcp[e] = h; vcp[plane_elems + e] = m; vcp[2 * plane_elems + e] = l;
}

struct __attribute__((packed)) F3 { float x, y, z; };

// One workgroup = one 32-vertex output tile x up to 8 sample tiles (one per compute wave) + ONE LOADER WAVE.
// The 9 weight fragments of a k-step (3 taps x 3 planes, 9 KiB; the 54 MB weight stream comes from HBM) are fetched by
// the loader wave four k-steps ahead into a register ring and handed to the compute waves through a double-buffered LDS
// stage: its vmcnt queue is its own, so the compute waves' (L2-resident) activation prefetch never queues behind an
// HBM miss (vmcnt retires in order).  Each compute wave register-prefetches its 9 activation fragments one step ahead.
// Joint-regression epilogue (lib/core/base.py:221, demo/run.py:142: joints = J_regressor @ mesh, a 107-nnz matrix): a wave that
// has just formed vertex v of its samples also writes w_e * v for every regressor entry e = (joint, v, w_e) of its 32-vertex
// block into P[sample][e][xyz]; k_jreg_reduce sums each joint's entries in a fixed order.  No atomics, no second pass over the
// 82 kB/mesh of vertices -- and with `out` == nullptr the vertices are never written at all (evaluation needs the joints only).
struct JregEpi {
    const int2* blk;            // [kOB] (first entry, entry count) of every 32-vertex block, entries sorted by vertex
    const int2* ent;            // (vertex, slot in P) per entry
    const float* w;             // weight per entry
    float* P;                   // [B][nnz][3]
    int nnz;
};
constexpr int kX3Waves = 8, kX3Ring = 4;
__global__ __launch_bounds__(64 * (kX3Waves + 1), 1) void k_upsample_x3(const __bf16* __restrict__ vcp, const __bf16* __restrict__ wp,
                                                                       const float* __restrict__ bias, const float* __restrict__ tpl,
                                                                       float* __restrict__ out, int B, int MT, int nwg, int64_t a_plane,
                                                                       int64_t w_plane, const JregEpi jr) {
    __shared__ bf16x8 wl[2][9][64];
    __shared__ f32x4 tot[kX3Waves][12][64];       // running totals of the hi*hi chains, see the flush below (96 KiB)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wg = xcd_remap(blockIdx.x, nwg);
    const int mgroups = (MT + kX3Waves - 1) / kX3Waves;
    const int ob = wg / mgroups;
    // wave-uniform base (scalar registers) + one per-lane offset: nine 64-bit per-lane pointers would cost the loader 18 VGPRs
    const bf16x8* wu = reinterpret_cast<const bf16x8*>(wp) + ((size_t)ob * kS16) * 64;
    const size_t w_tap = (size_t)kOB * kS16 * 64, wpl = (size_t)w_plane / 8;
#define W_AT(e, step) wu[(size_t)((e) % 3) * wpl + (size_t)((e) / 3) * w_tap + (size_t)(step) * 64 + lane]
    static_assert(kS16 % kX3Ring == 0, "k-steps must be a multiple of the ring depth");
    if (wave == kX3Waves) {
        bf16x8 ring[kX3Ring][9];
#pragma unroll
        for (int j = 0; j < kX3Ring; ++j)
#pragma unroll
            for (int e = 0; e < 9; ++e) ring[j][e] = W_AT(e, j);
#pragma unroll
        for (int e = 0; e < 9; ++e) {
            wl[0][e][lane] = ring[0][e];
            ring[0][e] = W_AT(e, kX3Ring);
        }
        __syncthreads();
#pragma unroll 1
        for (int s0 = 0; s0 < kS16; s0 += kX3Ring) {
#pragma unroll
            for (int j = 0; j < kX3Ring; ++j) {
                const int s = s0 + j, slot = (j + 1) % kX3Ring;
                const int nxt = s + 1 + kX3Ring < kS16 ? s + 1 + kX3Ring : kS16 - 1;
#pragma unroll
                for (int e = 0; e < 9; ++e) {
                    wl[(s + 1) & 1][e][lane] = ring[slot][e];                   // W(s+1), issued three steps ago
                    ring[slot][e] = W_AT(e, nxt);
                }
                __syncthreads();
            }
        }
        return;
    }
    const int mt_raw = (wg % mgroups) * kX3Waves + wave;
    const bool live = mt_raw < MT;
    const int mt = live ? mt_raw : MT - 1;             // idle waves shadow the last tile (they must keep the barriers)
    const bf16x8* ab = reinterpret_cast<const bf16x8*>(vcp) + ((size_t)__builtin_amdgcn_readfirstlane(mt) * 3 * kS16) * 64 + lane;
    const size_t a_lp = (size_t)kS16 * 64, ap = (size_t)a_plane / 8;
    f32x16 big[3], sm[3];
#pragma unroll
    for (int l = 0; l < 3; ++l) { big[l] = zero16(); sm[l] = zero16(); }
    // Activation fragments x[lp][plane] are reloaded ROLLING: the MFMAs run grouped by input position lp (for a fixed output l
    // that is still tap order k = 0,1,2, so every accumulator sees the same sequence of products as a tap-major loop), and as
    // soon as the group of lp has been queued its three fragments are re-requested for the next k-step -- the loads fly behind
    // the remaining groups and the barrier, at no extra registers (a second prefetch set would cost 36 VGPRs and spill: nine
    // waves per workgroup leave 168).
    bf16x8 x[3][3];
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
        for (int p = 0; p < 3; ++p) x[l][p] = ab[p * ap + l * a_lp];
    __syncthreads();
#pragma unroll 1
    for (int s = 0; s < kS16; ++s) {
        const int cur = s & 1;
        const size_t on = (size_t)(s + 1 < kS16 ? s + 1 : s) * 64;
#pragma unroll
        for (int lp = 0; lp < 3; ++lp) {
#pragma unroll
            for (int k = 2; k >= 0; --k) {             // l = lp + 1 - k ascending <=> k descending; accumulators are independent
                const int l = lp + 1 - k;
                if (l < 0 || l > 2) continue;
                const bf16x8 w0 = wl[cur][3 * k + 0][lane], w1 = wl[cur][3 * k + 1][lane], w2 = wl[cur][3 * k + 2][lane];
                big[l] = GATOR_MFMA_BF16(x[lp][0], w0, big[l]);      // hi*hi
                sm[l] = GATOR_MFMA_BF16(x[lp][0], w1, sm[l]);        // hi*mid
                sm[l] = GATOR_MFMA_BF16(x[lp][1], w0, sm[l]);        // mid*hi
                sm[l] = GATOR_MFMA_BF16(x[lp][1], w1, sm[l]);        // mid*mid
                sm[l] = GATOR_MFMA_BF16(x[lp][0], w2, sm[l]);        // hi*lo
                sm[l] = GATOR_MFMA_BF16(x[lp][2], w0, sm[l]);        // lo*hi
            }
            asm volatile("" ::: "memory");             // keep the reload below behind this group's MFMAs (and its LDS reads above it)
#pragma unroll
            for (int p = 0; p < 3; ++p) x[lp][p] = ab[p * ap + lp * a_lp + on];
        }
        // Every 7 k-steps the hi*hi chain is flushed into an LDS-resident total and restarted from zero: the chain's
        // partial sums stay small (their roundings scale with their magnitude) and only 4 additions happen at full
        // magnitude -- the two-level summation of the fp32 kernel, with LDS instead of 48 more accumulator registers.
        if (s % 7 == 6 && s != kS16 - 1) {
#pragma unroll
            for (int l = 0; l < 3; ++l)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v4[j] = big[l][4 * g + j];
                    if (s != 6) v4 += tot[wave][l * 4 + g][lane];
                    tot[wave][l * 4 + g][lane] = v4;
                }
#pragma unroll
            for (int l = 0; l < 3; ++l) big[l] = zero16();
        }
        __syncthreads();
    }
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v4 = tot[wave][l * 4 + g][lane];
#pragma unroll
            for (int j = 0; j < 4; ++j) big[l][4 * g + j] += v4[j];
        }
    const int ov = 32 * ob + (lane & 31), h = lane >> 5;
    if (ov >= kNV || !live) return;
    const float bo = bias[ov];
    const float t0 = tpl[ov * 3], t1 = tpl[ov * 3 + 1], t2 = tpl[ov * 3 + 2];
    const int2 jb = jr.P ? jr.blk[ob] : int2{0, 0};           // regressor entries of this vertex block (most blocks have none)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int smp = 32 * mt + kap(r) + 4 * h;
        if (smp < B) {
            F3 v;
            v.x = ((big[0][r] + sm[0][r]) + bo) + t0;
            v.y = ((big[1][r] + sm[1][r]) + bo) + t1;
            v.z = ((big[2][r] + sm[2][r]) + bo) + t2;
            if (out) *reinterpret_cast<F3*>(out + ((int64_t)smp * kNV + ov) * 3) = v;
            for (int e = jb.x; e < jb.x + jb.y; ++e) {
                const int2 en = jr.ent[e];
                if (en.x == ov) {
                    const float we = jr.w[e];
                    F3 q;
                    q.x = we * v.x; q.y = we * v.y; q.z = we * v.z;
                    *reinterpret_cast<F3*>(jr.P + ((int64_t)smp * jr.nnz + en.y) * 3) = q;
                }
            }
        }
    }
}

// joints[b][j][c] = sum of joint j's entries of P[b], ascending vertex order, accumulated in fp64
__global__ void k_jreg_reduce(const float* __restrict__ P, const int* __restrict__ row_ptr, int nnz, int nj, int B, float* __restrict__ joints) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * nj * 3) return;
    const int c = (int)(i % 3), j = (int)((i / 3) % nj);
    const int64_t b = i / (3 * nj);
    double acc = 0.0;
    for (int e = row_ptr[j]; e < row_ptr[j + 1]; ++e) acc += (double)P[(b * nnz + e) * 3 + c];
    joints[i] = (float)acc;
}

}  // namespace

size_t upsample_x3_weight_elems() { return (size_t)3 * 3 * kOB * kS16 * 512; }                 // bf16 elements, 3 planes
size_t upsample_x3_vcp_elems(int B) { return (size_t)3 * ((B + 31) / 32) * 3 * kS16 * 512; }      // bf16 elements, 3 planes

int pack_upsample_x3(const float* up_w, void* dst, void* stream) {
    const int64_t w_plane = (int64_t)upsample_x3_weight_elems() / 3;
    k_pack_up_x3<<<(unsigned)((w_plane + 255) / 256), 256, 0, (hipStream_t)stream>>>(up_w, (__bf16*)dst, w_plane);
    GATOR_HIP_CHECK(hipGetLastError());
    return GATOR_OK;
}

// planes are `cap`-strided (the workspace capacity), so padding written once stays valid for every smaller batch
int launch_pack_vc_x3(const float* vc, int B, int cap, void* vcp3, void* stream) {
    const int64_t a_plane = (int64_t)upsample_x3_vcp_elems(cap) / 3, n = (int64_t)upsample_x3_vcp_elems(B) / 3;
    k_pack_vc_x3<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(vc, B, (__bf16*)vcp3, n, a_plane);
    GATOR_HIP_CHECK(hipGetLastError());
    return GATOR_OK;
}

int launch_jreg_reduce(const FusedState* f, int B, float* joints, void* stream) {
    const int64_t n = (int64_t)B * f->jr_nj * 3;
    k_jreg_reduce<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(f->jr_P, f->jr_rowptr, f->jr_nnz, f->jr_nj, B, joints);
    GATOR_HIP_CHECK(hipGetLastError());
    return GATOR_OK;
}

// verts == nullptr: vertices are not stored (joint regression only); with_joints: also fill f->jr_P for launch_jreg_reduce
int launch_upsample_x3(const FusedState* f, const gator_ctx* c, int B, float* verts, void* stream, bool with_joints) {
    const int MT = (B + 31) / 32;
    JregEpi jr{};
    if (with_joints) { jr.blk = (const int2*)f->jr_blk; jr.ent = (const int2*)f->jr_ent; jr.w = f->jr_w; jr.P = f->jr_P; jr.nnz = f->jr_nnz; }
    const int64_t w_plane = (int64_t)upsample_x3_weight_elems() / 3, a_plane = (int64_t)upsample_x3_vcp_elems(f->cap) / 3;
    const int nwg = kOB * ((MT + kX3Waves - 1) / kX3Waves);
    k_upsample_x3<<<nwg, 64 * (kX3Waves + 1), 0, (hipStream_t)stream>>>((const __bf16*)f->vcp3, (const __bf16*)f->up_w3, c->w.up_b,
                                                                       c->w.v6890, verts, B, MT, nwg, a_plane, w_plane, jr);
    GATOR_HIP_CHECK(hipGetLastError());
    return GATOR_OK;
}

}  // namespace gator
