// Device building blocks of the fused path (gfx950, wave64, fp32-input MFMA 32x32x2).
//
// Register-resident activation tiles.  A wave owns a tile of 32 tokens; one "block" = 32 tokens x 32 channels held as
// an f32x16 per lane in exactly the MFMA 32x32 accumulator layout:
//     T-layout (token on lane):   lane l -> token l&31, half h = l>>5;  v[r] <-> channel kap(r) + 4h
//     C-layout (channel on lane): lane l -> channel l&31;               v[r] <-> token   kap(r) + 4h
//     kap(r) = (r&3) + 8*(r>>2)                                   (C/D map of v_mfma_f32_32x32x2_f32)
// Because a 32x32x2 MFMA takes ONE f32 per lane for A (A[i=l&31][k=l>>5]) and for B (B[k=l>>5][j=l&31]), register r of
// such a block is directly the operand of the MFMA step that contracts over index kap(r)+4h -- as A (rows = lane index)
// or as B (cols = lane index).  So a chain of linears, LayerNorms, softmaxes and attention products runs with no LDS
// round trip and no cross-lane traffic except one lane<->lane^32 exchange per row reduction.
//
// Packed linear weight (nn.Linear W[N][K]):  Wp[nb][kb][g][lane][j] = W[32nb + (lane&31)][32kb + 8g + 4(lane>>5) + j]
// i.e. one coalesced 1 KiB float4 wave-load feeds 4 MFMA steps, as A operand (T-layout output) or B operand (C-layout).
#pragma once
#include <hip/hip_runtime.h>

namespace gator {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define GATOR_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ int kap(int r) { return (r & 3) + 8 * (r >> 2); }

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}

// lane <-> lane ^ 32 through gfx950's v_permlane32_swap (one VALU instruction; __shfl_xor is a ds_bpermute: an LDS round trip the
// wave waits out, and these exchanges sit on the critical path of every row reduction - 28 of them per attention head).
// The swap returns, in every lane l, the values held by lane l & 31 and by lane 32 + (l & 31).
struct HalfPair { float lo, hi; };
__device__ __forceinline__ HalfPair xhalves(float v) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return HalfPair{__uint_as_float(r[0]), __uint_as_float(r[1])};
}
__device__ __forceinline__ float xhalf(float v) {
    const HalfPair q = xhalves(v);
    return (threadIdx.x & 32) ? q.lo : q.hi;
}

// one packed (nb,kb) weight tile = 4 float4 per lane
struct WTile { f32x4 g[4]; };

__device__ __forceinline__ WTile load_wtile(const float* __restrict__ Wp, int tile, int lane) {
    const f32x4* p = reinterpret_cast<const f32x4*>(Wp) + ((size_t)tile * 4) * 64 + lane;
    WTile t;
#pragma unroll
    for (int g = 0; g < 4; ++g) t.g[g] = p[g * 64];
    return t;
}

// acc(T-layout)[n][token] += sum_k W[n][k] x[token][k] over one 32-wide k block:  A = weights, B = activations
__device__ __forceinline__ f32x16 mma_T(const WTile& w, const f32x16& x, f32x16 acc) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = GATOR_MFMA(w.g[g][j], x[4 * g + j], acc);
    return acc;
}
// acc(C-layout)[token][n] : A = activations, B = weights
__device__ __forceinline__ f32x16 mma_C(const WTile& w, const f32x16& x, f32x16 acc) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = GATOR_MFMA(x[4 * g + j], w.g[g][j], acc);
    return acc;
}

// Two independent chains, MFMAs interleaved so that consecutive instructions never write the same accumulator: a 32x32x2
// MFMA that reads the previous one's result does NOT issue back to back (measured ~21 extra cycles per dependent pair,
// i.e. 75 % of the pipe on a fully dependent chain when no second wave fills the gaps).
__device__ __forceinline__ void mma2_T(const WTile& w0, const f32x16& x0, f32x16& a0, const WTile& w1, const f32x16& x1, f32x16& a1) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a0 = GATOR_MFMA(w0.g[g][j], x0[4 * g + j], a0);
            a1 = GATOR_MFMA(w1.g[g][j], x1[4 * g + j], a1);
        }
}
__device__ __forceinline__ void mma2_C(const WTile& w0, const f32x16& x0, f32x16& a0, const WTile& w1, const f32x16& x1, f32x16& a1) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a0 = GATOR_MFMA(x0[4 * g + j], w0.g[g][j], a0);
            a1 = GATOR_MFMA(x1[4 * g + j], w1.g[g][j], a1);
        }
}
// sum_r A[r] x B[r] over the 16 registers of two operand blocks, as two interleaved 8-term chains: init + even + odd
__device__ __forceinline__ f32x16 dot16(const f32x16& A, const f32x16& B, f32x16 init) {
    f32x16 e = init, o = zero16();
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        e = GATOR_MFMA(A[r], B[r], e);
        o = GATOR_MFMA(A[r + 1], B[r + 1], o);
    }
    return e + o;
}
// two independent 16-register products interleaved: a0 += A0.B0 ; a1 += A1.B1
__device__ __forceinline__ void dot16x2(const f32x16& A0, const f32x16& B0, f32x16& a0, const f32x16& A1, const f32x16& B1, f32x16& a1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        a0 = GATOR_MFMA(A0[r], B0[r], a0);
        a1 = GATOR_MFMA(A1[r], B1[r], a1);
    }
}

// per-channel vector (bias / norm weight) in T-layout: v[r] = vec[base + kap(r) + 4h]
__device__ __forceinline__ f32x16 load_chanvec_T(const float* __restrict__ vec, int base, int h) {
    f32x16 v;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(vec + base + 8 * g + 4 * h);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[4 * g + j] = t[j];
    }
    return v;
}

// a block stored as 4 float4 per lane ([g][lane][4]); same form for T- and C-layout blocks
__device__ __forceinline__ f32x16 load_block(const float* __restrict__ p, int lane) {
    const f32x4* q = reinterpret_cast<const f32x4*>(p) + lane;
    f32x16 v;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 t = q[g * 64];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[4 * g + j] = t[j];
    }
    return v;
}
__device__ __forceinline__ void store_block(float* __restrict__ p, int lane, const f32x16& v) {
    f32x4* q = reinterpret_cast<f32x4*>(p) + lane;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 t;
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = v[4 * g + j];
        q[g * 64] = t;
    }
}

// erf to < 1 ulp (N. Juffa's two-range minimax form), both ranges evaluated and selected branch-free: ~23 VALU ops against
// ~60 for the branchy library erff, which costs both sides whenever a wave's lanes straddle |x| = 1.
__device__ __forceinline__ float erf_fast(float a) {
    const float t = fabsf(a), s = a * a;
    float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
    const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
    r = fmaf(r, s, u);
    r = fmaf(r, t, -1.06777877e-1f);
    r = fmaf(r, t, -6.34846687e-1f);
    r = fmaf(r, t, -1.28717512e-1f);
    r = fmaf(r, t, -t);
    const float big = copysignf(1.0f - __builtin_amdgcn_exp2f(r * 1.4426950408889634f), a);
    float q = -5.96761703e-4f;
    q = fmaf(q, s, 4.99119423e-3f);
    q = fmaf(q, s, -2.67681349e-2f);
    q = fmaf(q, s, 1.12819925e-1f);
    q = fmaf(q, s, -3.76125336e-1f);
    q = fmaf(q, s, 1.28379166e-1f);
    const float small = fmaf(q, a, a);
    return t > 0.927734375f ? big : small;
}
// exact (erf) GELU, as nn.GELU() / F.gelu default
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752440f)); }

// GELU for the MFMA kernels, two values at once with packed fp32 math (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two IEEE
// operations per lane per issue slot).  x * Phi(x) with Phi from ONE polynomial and one exp2:
//     0.5 erfc(t) = exp2(R(t)),  t = min(|x| / sqrt 2, 4.3),  R = degree-8 minimax fit of log2(erfc(t)) - 1 weighted by erfc(t)
//     Phi(x) = x < 0 ? 0.5 erfc(t) : 1 - 0.5 erfc(t)
// No cancellation on either side (the negative tail keeps its relative accuracy).  Error of Phi 7.5e-9 from the fit plus
// fp32 rounding; against exact GELU max |err| / |x| = 1.1e-7 over 2.4M points (torch's fp32 F.gelu: 3.6e-7), at ~12 issue
// slots per element instead of ~18 for the two-range erf form above (gelu_f), which stays for scalar uses.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 gelu_f2(f32x2 x) {
    const f32x2 a = x * 0.70710678118654752440f;
    f32x2 t;
    t[0] = fminf(fabsf(a[0]), 4.3f);
    t[1] = fminf(fabsf(a[1]), 4.3f);
    f32x2 r = pk_fma(f32x2(-4.435285315e-05f), t, f32x2(4.369443071e-04f));
    r = pk_fma(r, t, f32x2(-1.460381877e-03f));
    r = pk_fma(r, t, f32x2(-8.251648338e-04f));
    r = pk_fma(r, t, f32x2(2.830188636e-02f));
    r = pk_fma(r, t, f32x2(-1.485066472e-01f));
    r = pk_fma(r, t, f32x2(-9.184098145e-01f));
    r = pk_fma(r, t, f32x2(-1.627909326e+00f));
    r = pk_fma(r, t, f32x2(-9.999999783e-01f));
    f32x2 e;
    e[0] = __builtin_amdgcn_exp2f(r[0]);
    e[1] = __builtin_amdgcn_exp2f(r[1]);
    const f32x2 up = 1.0f - e;
    f32x2 phi;
    phi[0] = x[0] < 0.f ? e[0] : up[0];
    phi[1] = x[1] < 0.f ? e[1] : up[1];
    return x * phi;
}
// y * k + x as packed FMAs (y: a 4-product linear's raw output, k = 1 / lin_s: a power of two, so this is exactly x + value)
__device__ __forceinline__ f32x16 fma16(const f32x16& y, float k, const f32x16& x) { return __builtin_elementwise_fma(y, f32x16(k), x); }

// The same for a tile that holds S x its value (k = 1 / S, a power of two): returns S x GELU(value).  k only rescales the constant
// of the first multiplication, so every rounding is the one gelu_f2 makes on the unscaled value.
__device__ __forceinline__ f32x2 gelu_f2_scaled(f32x2 x, float k) {
    const f32x2 a = x * (0.70710678118654752440f * k);
    f32x2 t;
    t[0] = fminf(fabsf(a[0]), 4.3f);
    t[1] = fminf(fabsf(a[1]), 4.3f);
    f32x2 r = pk_fma(f32x2(-4.435285315e-05f), t, f32x2(4.369443071e-04f));
    r = pk_fma(r, t, f32x2(-1.460381877e-03f));
    r = pk_fma(r, t, f32x2(-8.251648338e-04f));
    r = pk_fma(r, t, f32x2(2.830188636e-02f));
    r = pk_fma(r, t, f32x2(-1.485066472e-01f));
    r = pk_fma(r, t, f32x2(-9.184098145e-01f));
    r = pk_fma(r, t, f32x2(-1.627909326e+00f));
    r = pk_fma(r, t, f32x2(-9.999999783e-01f));
    f32x2 e;
    e[0] = __builtin_amdgcn_exp2f(r[0]);
    e[1] = __builtin_amdgcn_exp2f(r[1]);
    const f32x2 up = 1.0f - e;
    f32x2 phi;
    phi[0] = x[0] < 0.f ? e[0] : up[0];
    phi[1] = x[1] < 0.f ? e[1] : up[1];
    return x * phi;
}
__device__ __forceinline__ void gelu_tile_scaled(f32x16& v, float k) {
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        f32x2 p;
        p[0] = v[r]; p[1] = v[r + 1];
        p = gelu_f2_scaled(p, k);
        v[r] = p[0]; v[r + 1] = p[1];
    }
}
// GELU over a whole register tile
__device__ __forceinline__ void gelu_tile(f32x16& v) {
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        f32x2 p;
        p[0] = v[r]; p[1] = v[r + 1];
        p = gelu_f2(p);
        v[r] = p[0]; v[r + 1] = p[1];
    }
}

// Per-channel vector in T-layout through the SCALAR cache: the value depends only on (register, lane half), so the 8
// floats of each g are fetched with wave-uniform s_load (constant address space -> SMEM, counted on lgkmcnt) and selected
// by half.  Unlike a vector load this never queues behind the in-order vmcnt of a weight prefetch in flight.
// `vec + base` must be wave-uniform (kernel-argument pointer, compile-time or readfirstlane'd base).
typedef const float __attribute__((address_space(4))) gator_cfloat;
__device__ __forceinline__ f32x16 load_chanvec_S(const float* vec, int base, int h) {
    const gator_cfloat* cv = (const gator_cfloat*)(unsigned long long)(vec + base);
    f32x16 v;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float lo = cv[8 * g + j], hi = cv[8 * g + 4 + j];
            v[4 * g + j] = h ? hi : lo;
        }
    return v;
}

// The same in two steps, so that the scalar loads can be ISSUED ahead of a block of MFMAs and CONSUMED after it: an s_load
// waited for at its point of use costs its whole latency (1-2k cycles under load when the wave is alone on its SIMD).
struct SVec { float lo[16], hi[16]; };
__device__ __forceinline__ SVec chanvec_issue(const float* vec, int base) {
    const gator_cfloat* cv = (const gator_cfloat*)(unsigned long long)(vec + base);
    SVec s;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s.lo[4 * g + j] = cv[8 * g + j];
            s.hi[4 * g + j] = cv[8 * g + 4 + j];
        }
    return s;
}
__device__ __forceinline__ f32x16 chanvec_select(const SVec& s, int h) {
    f32x16 v;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = h ? s.hi[r] : s.lo[r];
    return v;
}

// per-channel vector from the workgroup's LDS table (k_mdr_layer): 4 ds_read_b128 with two distinct addresses each, against 32
// scalar loads + 16 v_mov + 16 v_cndmask for the scalar-cache form -- 30 such vectors per tile were 9 % of the kernel's VALU work
__device__ __forceinline__ f32x16 chanvec_lds(const float* V, int off, int h) {
    f32x16 v;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(V + off + 8 * g + 4 * h);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[4 * g + j] = t[j];
    }
    return v;
}

// XCD-aware bijective remap of a 1-D grid: blocks that share an XCD (equal blockIdx % 8) get CONTIGUOUS logical ids,
// so the workgroups of one sample hit one L2.  Speed only -- never correctness (cdna_hip_programming.md T1).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

}  // namespace gator
