// Vertex regressor (upsample_conv + template, MDR.py:167-168) on the fp16 MFMA with every fp32 operand carried as TWO fp16 planes,
// x = hi + lo (hi = fp16(x), lo = fp16(x - hi): 22 significant bits).  Three partial products per k-step (hi*hi | hi*lo, lo*hi)
// instead of the six of the exact three-way bf16 split (upsample_x3.hip) -- half the MFMA work, and the MFMA pipe is what bounds
// that kernel (tools/microbench/upsample_lab.hip: 42 us of its 62 are MFMA issue at the clock the chip holds under that load).
//
// What it costs: the operands are ROUNDED to 22 bits (2^-23 relative) and lo*lo (2^-22) is dropped.  This is the LAST linear of
// the path, a 1293-term sum per coordinate with nothing downstream to amplify it: in the fp64 oracle with two real fp16 planes and
// everything else exact the vertices move by 2.2e-5 mm max / 4e-6 mm rms (fp32 arithmetic on the same product: 1.8e-4 / 2e-5), see
// DESIGN.md 4c.  Operands are pre-scaled by powers of two so the low planes stay fp16-normal: activations x 2^4 (|vert431| must
// stay below 4094 m -- beyond that the high plane overflows to inf where the reference would still return a number), weights by
// the largest power of two that keeps max|w| below 2^14, chosen when the weights are packed; the epilogue undoes both exactly.
// GATOR_UPSAMPLE_X3=1 selects the exact three-plane kernel instead, =0 the fp32-input MFMA kernel.
//
// Shape: one workgroup = 64 vertices x 128 samples, eight waves (2 vertex blocks x 4 sample tiles), no loader wave.  EVERY operand
// goes global -> LDS by LDS-DMA issued from all eight waves (36 KiB per 16-deep k-step: 4 x 6 activation + 2 x 6 weight fragments),
// four one-k-step stages in LDS.  The loop has no compiler-visible VMEM, so the only vmcnt traffic is the DMA and every wait is
// written by hand.  The barrier at the top of step s publishes stage s + 1 -- one step early -- so the fragments step s + 1 needs
// first are read from LDS behind the last MFMA of step s and no step starts with an LDS round trip.
#include <cmath>
#include <cstring>

#include "fused_common.h"
#include "fused_state.h"

namespace gator {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int kS16 = 28;                            // 16-deep k-steps over the 431 (-> 448) coarse vertices
constexpr int kNBuf = 4, kStageFrags = 36;          // 1 KiB fragments per stage: [mt 4][lp 3][plane 2] then [ob 2][tap 3][plane 2]
constexpr int kActShift = 4;                        // activations x 2^4
#define GATOR_MFMA_F16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ void split2(float x, _Float16& h, _Float16& l) {
    h = (_Float16)x;
    l = (_Float16)(x - (float)h);
}

// wp[ob pair][s][ob 2][tap 3][plane 2][lane][8] <- split2( scale * w[32 ob + (lane&31)][16 s + 8 (lane>>5) + j][tap] )
__global__ void k_pack_up_x2(const float* __restrict__ w, _Float16* __restrict__ dst, int64_t n_frag_pairs, float scale) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one element of one (hi, lo) fragment pair
    if (e >= n_frag_pairs * 512) return;
    const int j = e & 7, lane = (e >> 3) & 63;
    int64_t r = e >> 9;
    const int tap = r % 3; r /= 3;
    const int o2 = r & 1; r >>= 1;
    const int s = r % kS16;
    const int op = (int)(r / kS16);
    const int o = 32 * (2 * op + o2) + (lane & 31), c = 16 * s + 8 * (lane >> 5) + j;
    const float x = (o < kNV && c < kV) ? w[((int64_t)o * kV + c) * 3 + tap] * scale : 0.f;
    _Float16 h, l;
    split2(x, h, l);
    const int64_t pair = (((int64_t)op * kS16 + s) * 2 + o2) * 3 + tap;
    dst[(pair * 2 + 0) * 512 + lane * 8 + j] = h;
    dst[(pair * 2 + 1) * 512 + lane * 8 + j] = l;
}
// ap[mt / 4][s][mt % 4][lp 3][plane 2][lane][8] <- split2( 2^4 * vc[32 mt + (lane&31)][16 s + 8 (lane>>5) + j][lp] )
__global__ void k_pack_vc_x2(const float* __restrict__ vc, int B, _Float16* __restrict__ ap, int64_t n_frag_pairs) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_frag_pairs * 512) return;
    const int j = e & 7, lane = (e >> 3) & 63;
    int64_t r = e >> 9;
    const int lp = r % 3; r /= 3;
    const int mi = r & 3; r >>= 2;
    const int s = r % kS16;
    const int mg = (int)(r / kS16);
    const int smp = 32 * (4 * mg + mi) + (lane & 31), c = 16 * s + 8 * (lane >> 5) + j;
    const float x = (smp < B && c < kV) ? vc[((int64_t)smp * kV + c) * 3 + lp] * (float)(1 << kActShift) : 0.f;
    _Float16 h, l;
    split2(x, h, l);
    const int64_t pair = (((int64_t)mg * kS16 + s) * 4 + mi) * 3 + lp;
    ap[(pair * 2 + 0) * 512 + lane * 8 + j] = h;
    ap[(pair * 2 + 1) * 512 + lane * 8 + j] = l;
}

__global__ void k_absmax(const float* __restrict__ w, int64_t n, unsigned* __restrict__ out) {
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float a = fabsf(w[i]);
        m = a > m || a != a ? a : m;              // a NaN weight wins: the scale falls back to 1
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float q = __shfl_xor(m, o);
        m = q > m || q != q ? q : m;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));      // non-negative floats (and NaN above inf) order as unsigned
}

// one 1 KiB fragment global -> LDS without registers: wave-uniform base in SGPRs, one per-lane byte offset.  Inline asm on
// purpose: hipcc counts the builtin form in its vmcnt bookkeeping and drains it before the next LDS access (gat_tiled.hip).
__device__ __forceinline__ void dma_frag(const void* base, unsigned lane_off, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_off), "s"(base), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct __attribute__((packed)) F3 { float x, y, z; };

// Joint-regression epilogue: see upsample_x3.hip (same tables, same order of operations on the finished vertex).
struct JregEpi2 {
    const int2* blk;
    const int2* ent;
    const float* w;
    float* P;
    int nnz;
};

// WLO = false (BASELINE config 3, the 16-bit operand mode): the weights' lo plane is not used -- weights on ONE fp16 plane, coarse vertices on two:
// two MFMAs per product instead of three (0.35 mm max / 0.06 mm rms by the emulation: profiles/r05_emulate_16bit.txt)
template <bool WLO = true>
__global__ __launch_bounds__(512, 1) void k_upsample_x2(const _Float16* __restrict__ ap, const _Float16* __restrict__ wp,
                                                        const float* __restrict__ bias, const float* __restrict__ tpl,
                                                        float* __restrict__ out, int B, int MT, int MG, int nwg, float unscale,
                                                        const JregEpi2 jr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    f16x8 (*st)[kStageFrags][64] = reinterpret_cast<f16x8 (*)[kStageFrags][64]>(lds_raw);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wg = xcd_remap(blockIdx.x, nwg);
    const int obp = wg / MG, mg = wg % MG;           // sample groups of one vertex-block pair are neighbours: they share its weights in L2
    const int mi = wave & 3, oi = wave >> 2;
    const int ob = 2 * obp + oi, mt = 4 * mg + mi;
    const bool live = mt < MT;                       // a dead wave keeps its copies and barriers and skips the MFMAs
    const unsigned lane_off = lane * 16;
    const unsigned lds0 = (unsigned)(unsigned long long)lds_raw;
    const char* a_src = reinterpret_cast<const char*>(ap) + (size_t)mg * kS16 * 24 * 1024;
    const char* w_src = reinterpret_cast<const char*>(wp) + (size_t)obp * kS16 * 12 * 1024;
    // fragments of a stage this wave copies: f = wave + 8 i (i < 5, f < 36) -- source, per-stage stride and LDS offset of each
    const char* src[5];
    unsigned stride[5], dsto[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int f = wave + 8 * i;
        src[i] = f < 24 ? a_src + f * 1024 : w_src + (f - 24) * 1024;
        stride[i] = f < 24 ? 24 * 1024 : 12 * 1024;
        dsto[i] = lds0 + f * 1024;
    }
    auto issue = [&](int s) {
        const unsigned dst = (unsigned)(s & (kNBuf - 1)) * kStageFrags * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_frag(src[i] + (size_t)s * stride[i], lane_off, dst + dsto[i]);
        if (wave < 4) dma_frag(src[4] + (size_t)s * stride[4], lane_off, dst + dsto[4]);
    };
    // this wave's copies of a stage must have landed before the barrier that publishes it; `younger` stages stay in flight
    auto wait_landed = [&](int younger) {
        if (younger >= 1) { if (wave < 4) wait_vm<5>(); else wait_vm<4>(); }
        else wait_vm<0>();
    };
    f32x16 big[3], sm[3], tot[3];
#pragma unroll
    for (int l = 0; l < 3; ++l) { big[l] = zero16(); sm[l] = zero16(); tot[l] = zero16(); }
    f16x8 a[3][2], w[3][2];
    auto load_first = [&](int s) {                   // position 0, taps 0 and 1: what a step multiplies first
        const f16x8 (*cur)[64] = st[s & (kNBuf - 1)];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            a[0][p] = cur[mi * 6 + p][lane];
            w[0][p] = cur[24 + oi * 6 + p][lane];
            w[1][p] = cur[24 + oi * 6 + 2 + p][lane];
        }
    };
    auto load_rest = [&](int s) {
        const f16x8 (*cur)[64] = st[s & (kNBuf - 1)];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            a[1][p] = cur[mi * 6 + 2 + p][lane];
            w[2][p] = cur[24 + oi * 6 + 4 + p][lane];
            a[2][p] = cur[mi * 6 + 4 + p][lane];
        }
    };
    issue(0); issue(1); issue(2);
    if (wave < 4) wait_vm<10>(); else wait_vm<8>();  // stage 0 landed, 1 and 2 in flight
    __builtin_amdgcn_s_barrier();
    load_first(0);
#pragma unroll 1
    for (int s = 0; s < kS16; ++s) {
        if (s + 1 < kS16) wait_landed(kS16 - 2 - s);
        __builtin_amdgcn_s_barrier();                // publishes stage s + 1; every wave is done with the slot of stage s - 1
        if (s + 3 < kS16) issue(s + 3);
        load_rest(s);
        if (live) {
            // out[l] += x[lp] * w[k] with l = lp + 1 - k (MDR.py:167, padding 1): grouped by input position, taps descending, so
            // every accumulator sees tap order 0, 1, 2 within a k-step -- the order of upsample_x3.hip
#pragma unroll
            for (int lp = 0; lp < 3; ++lp)
#pragma unroll
                for (int k = 2; k >= 0; --k) {
                    const int l = lp + 1 - k;
                    if (l < 0 || l > 2) continue;
                    big[l] = GATOR_MFMA_F16(a[lp][0], w[k][0], big[l]);      // hi*hi
                    if constexpr (WLO) sm[l] = GATOR_MFMA_F16(a[lp][0], w[k][1], sm[l]);        // hi*lo
                    sm[l] = GATOR_MFMA_F16(a[lp][1], w[k][0], sm[l]);        // lo*hi
                }
            // every 7 k-steps the hi*hi chain moves into a running total and restarts from zero: its partial sums stay small, only
            // 4 additions happen at full magnitude (the two-level summation of the fp32 and bf16 x 3 kernels)
            if (s % 7 == 6) {
#pragma unroll
                for (int l = 0; l < 3; ++l) { tot[l] += big[l]; big[l] = zero16(); }
            }
        }
        if (s + 1 < kS16) load_first(s + 1);
    }
    const int ov = 32 * ob + (lane & 31), h = lane >> 5;
    if (ov >= kNV || !live) return;
    const float bo = bias[ov];
    const float t0 = tpl[ov * 3], t1 = tpl[ov * 3 + 1], t2 = tpl[ov * 3 + 2];
    const int2 jb = jr.P ? jr.blk[ob] : int2{0, 0};
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int smp = 32 * mt + kap(r) + 4 * h;
        if (smp < B) {
            F3 v;
            v.x = ((tot[0][r] + sm[0][r]) * unscale + bo) + t0;
            v.y = ((tot[1][r] + sm[1][r]) * unscale + bo) + t1;
            v.z = ((tot[2][r] + sm[2][r]) * unscale + bo) + t2;
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

constexpr size_t kX2Lds = (size_t)kNBuf * kStageFrags * 1024;      // 144 KiB

}  // namespace

size_t upsample_x2_weight_elems() { return (size_t)(kOB / 2) * kS16 * 12 * 512; }                    // fp16 elements
size_t upsample_x2_vcp_elems(int B) { return (size_t)((B + 127) / 128) * kS16 * 24 * 512; }           // fp16 elements

int upsample_x2_prepare_device() {
    GATOR_HIP_CHECK(hipFuncSetAttribute((const void*)k_upsample_x2<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kX2Lds));
    GATOR_HIP_CHECK(hipFuncSetAttribute((const void*)k_upsample_x2<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kX2Lds));
    return GATOR_OK;
}

// Packs upsample_conv.weight [6890][431][3] and returns the factor that undoes both operand scalings in *unscale
int pack_upsample_x2(const float* up_w, void* dst, float* unscale, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    unsigned* d_max = nullptr;
    GATOR_HIP_CHECK(hipMalloc(&d_max, sizeof(unsigned)));
    GATOR_HIP_CHECK(hipMemsetAsync(d_max, 0, sizeof(unsigned), st));
    const int64_t n = (int64_t)kNV * kV * 3;
    k_absmax<<<1024, 256, 0, st>>>(up_w, n, d_max);
    unsigned bits = 0;
    GATOR_HIP_CHECK(hipMemcpyAsync(&bits, d_max, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    GATOR_HIP_CHECK(hipStreamSynchronize(st));
    (void)hipFree(d_max);
    float wmax;
    static_assert(sizeof(wmax) == sizeof(bits), "float bits");
    memcpy(&wmax, &bits, sizeof(wmax));
    int shift = 0;
    if (wmax > 0.f && std::isfinite(wmax)) {
        int e;
        (void)std::frexp(wmax, &e);                  // wmax = m * 2^e, m in [0.5, 1)  ->  wmax * 2^(14 - e) in [2^13, 2^14)
        shift = 14 - e;
        if (shift > 24) shift = 24;                  // all-tiny weights: any scale keeps them normal enough
        if (shift < -16) shift = -16;
    }
    const int64_t pairs = (int64_t)upsample_x2_weight_elems() / 1024;
    k_pack_up_x2<<<(unsigned)((pairs * 512 + 255) / 256), 256, 0, st>>>(up_w, (_Float16*)dst, pairs, std::ldexp(1.0f, shift));
    GATOR_HIP_CHECK(hipGetLastError());
    *unscale = std::ldexp(1.0f, -(shift + kActShift));
    return GATOR_OK;
}

int launch_pack_vc_x2(const float* vc, int B, void* vcp2, void* stream) {
    const int64_t pairs = (int64_t)upsample_x2_vcp_elems(B) / 1024;
    k_pack_vc_x2<<<(unsigned)((pairs * 512 + 255) / 256), 256, 0, (hipStream_t)stream>>>(vc, B, (_Float16*)vcp2, pairs);
    GATOR_HIP_CHECK(hipGetLastError());
    return GATOR_OK;
}

// verts == nullptr: vertices are not stored (joint regression only); with_joints: also fill f->jr_P for launch_jreg_reduce
int launch_upsample_x2(const FusedState* f, const gator_ctx* c, int B, float* verts, void* stream, bool with_joints, bool w1) {
    const int MT = (B + 31) / 32, MG = (MT + 3) / 4;
    JregEpi2 jr{};
    if (with_joints) { jr.blk = (const int2*)f->jr_blk; jr.ent = (const int2*)f->jr_ent; jr.w = f->jr_w; jr.P = f->jr_P; jr.nnz = f->jr_nnz; }
    const int nwg = (kOB / 2) * MG;
    if (w1) k_upsample_x2<false><<<nwg, 512, kX2Lds, (hipStream_t)stream>>>((const _Float16*)f->vcp3, (const _Float16*)f->up_w2, c->w.up_b, c->w.v6890,
                                                                           verts, B, MT, MG, nwg, f->up_w2_unscale, jr);
    else k_upsample_x2<true><<<nwg, 512, kX2Lds, (hipStream_t)stream>>>((const _Float16*)f->vcp3, (const _Float16*)f->up_w2, c->w.up_b, c->w.v6890,
                                                                       verts, B, MT, MG, nwg, f->up_w2_unscale, jr);
    GATOR_HIP_CHECK(hipGetLastError());
    return GATOR_OK;
}

}  // namespace gator
