// Does the 431 x 431 self-attention of the MDR layers (mdr_fused.hip: self_attention_head_x2, copied here) get faster per SIMD with MORE
// than two waves on it?  The shipped kernel is held at two waves per SIMD by the token-wise part of the tile body (239 VGPRs); the
// attention loop alone needs ~150.  Here: the loop alone, both heads of a 32-query tile per trip, K / V of ONE sample (L2-resident
// for the whole chip, so the memory side is the best case for every variant), launched with 1, 2, 3 and 4 waves per SIMD
// (__launch_bounds__ caps the registers; a variant that spills says so), the same total number of tiles each.
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I gator_amd/csrc -I include tools/microbench/attn_occupancy.hip -o tools/microbench/attn_occupancy.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include "x3_common.h"
using namespace gator;

constexpr int kV = 431, kVT = 14, kTile = 1024;

// accumulators in AGPRs (inline asm; the compiler cannot re-form these): does the matrix pipe then overlap the partner's VALU work?
__device__ __forceinline__ void mfma_acc(f32x16& c, const f16x8& a, const f16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_acc0(f32x16& c, const f16x8& a, const f16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ f32x16 x2_mma_agpr0(const X2& A, const X2& B) {
    f32x16 c;
    mfma_acc0(c, A.p[1][0], B.p[0][0]);
    mfma_acc(c, A.p[0][0], B.p[1][0]);
    mfma_acc(c, A.p[1][1], B.p[0][1]);
    mfma_acc(c, A.p[0][1], B.p[1][1]);
    mfma_acc(c, A.p[0][0], B.p[0][0]);
    mfma_acc(c, A.p[0][1], B.p[0][1]);
    return c;
}

// the 16x16x32 shape: one 32x32 (k = 32) product = 4 output quadrants x 1 MFMA of 16 cycles instead of 2 x 32 cycles; timing only
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mma16_pair(f32x16& acc, const f16x8& a0, const f16x8& b0, const f16x8& a1, const f16x8& b1) {
    f32x4v q[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { q[i][0] = acc[4 * i]; q[i][1] = acc[4 * i + 1]; q[i][2] = acc[4 * i + 2]; q[i][3] = acc[4 * i + 3]; }
    q[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, q[0], 0, 0, 0);
    q[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, q[1], 0, 0, 0);
    q[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, q[2], 0, 0, 0);
    q[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, q[3], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc[4 * i] = q[i][0]; acc[4 * i + 1] = q[i][1]; acc[4 * i + 2] = q[i][2]; acc[4 * i + 3] = q[i][3]; }
}
__device__ __forceinline__ f32x16 x2_mma16(const X2& A, const X2& B, f32x16 acc) {      // three products, each both k-steps at once
    mma16_pair(acc, A.p[1][0], B.p[0][0], A.p[1][1], B.p[0][1]);
    mma16_pair(acc, A.p[0][0], B.p[1][0], A.p[0][1], B.p[1][1]);
    mma16_pair(acc, A.p[0][0], B.p[0][0], A.p[0][1], B.p[0][1]);
    return acc;
}
#define ATTN_PV(VB, PX) { O2 = x2_mma_small(VB, PX, O2); O = x2_mma_main(VB, PX, O); }
#define ATTN_TILE_X2(KT, KB, VB) ATTN_TILE_CUT(KT, KB, VB, 0)
#define ATTN_TILE_CUT(KT, KB, VB, CUT_)                                                                            \
    {                                                                                                       \
        __builtin_amdgcn_s_setprio(0);                                                                      \
        f32x16 S;                                                                                           \
        if constexpr ((CUT & 32) != 0) S = x2_mma16(KB, qx, zero16()); else if constexpr ((CUT & 16) != 0) S = x2_mma_agpr0(KB, qx); else if constexpr ((CUT & 4) == 0) S = x2_mma(KB, qx, zero16()); else { _Pragma("unroll") for (int r = 0; r < 16; ++r) S[r] = (float)KB.p[0][r >> 3][r & 7] + (float)qx.p[0][r >> 3][r & 7]; } \
        __builtin_amdgcn_s_setprio(1);                                                                      \
        float bm = -1e30f;                                                                                  \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                    \
            float sc = S[r];                                                                                \
            if ((KT) == kVT - 1 && kap(r) + 4 * h >= kV - 32 * (kVT - 1)) sc = -1e30f;                      \
            S[r] = sc;                                                                                      \
            if constexpr ((CUT & 64) == 0) bm = fmaxf(bm, sc);                                              \
        }                                                                                                   \
        if constexpr ((CUT & 64) != 0) bm = m; else bm = fmaxf(bm, xhalf(bm));                              \
        if (!__all(bm <= m + 2048.0f)) {                                                                    \
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
            const float pe = (CUT & 1) ? fmaf(S[r], 0.00390625f, off) : __builtin_amdgcn_exp2f(fmaf(S[r], 0.00390625f, off)); \
            S[r] = pe;                                                                                      \
            ps += pe;                                                                                       \
        }                                                                                                   \
        l += ps;                                                                                            \
        X2 px_;                                                                                             \
        if constexpr ((CUT & 2) == 0) px_ = x2_split(S); else { _Pragma("unroll") for (int r = 0; r < 16; ++r) { px_.p[0][r >> 3][r & 7] = (_Float16)S[r]; } px_.p[1][0] = px_.p[0][1]; px_.p[1][1] = px_.p[0][0]; } \
        __builtin_amdgcn_s_setprio(0);                                                                      \
        if constexpr ((CUT & 32) != 0) { O = x2_mma16(VB, px_, O); } else if constexpr ((CUT & 16) != 0) { mfma_acc(O2, VB.p[1][0], px_.p[0][0]); mfma_acc(O2, VB.p[0][0], px_.p[1][0]); mfma_acc(O2, VB.p[1][1], px_.p[0][1]); mfma_acc(O2, VB.p[0][1], px_.p[1][1]); mfma_acc(O, VB.p[0][0], px_.p[0][0]); mfma_acc(O, VB.p[0][1], px_.p[0][1]); } else if constexpr ((CUT & 4) == 0) { ATTN_PV(VB, px_) } else { _Pragma("unroll") for (int r = 0; r < 16; ++r) O[r] += (float)px_.p[0][r >> 3][r & 7] * (float)VB.p[0][r >> 3][r & 7]; } \
    }
// DB = true: the shipped form (next tile's K / V in flight in a second register set); false: one set, refilled right behind its last use
template <bool DB, int CUT = 0>
__device__ __forceinline__ f32x16 attn_head(const float* __restrict__ qt, const float* __restrict__ kbase, const float* __restrict__ vbase, int lane) {
    const int h = lane >> 5;
    const X2 qx = x2_load(qt, lane);
    f32x16 O = zero16(), O2 = zero16();
    float m = (CUT & 64) ? 0.f : -1e30f, l = 0.f;
    X2 kb = x2_load(kbase, lane), vb = x2_load(vbase, lane);
    if constexpr (DB) {
#pragma unroll 1
        for (int kt = 0; kt < kVT - 2; kt += 2) {
            X2 kn = x2_load(kbase + (size_t)(kt + 1) * 2 * kTile, lane), vn = x2_load(vbase + (size_t)(kt + 1) * 2 * kTile, lane);
            ATTN_TILE_X2(0, kb, vb)
            kb = x2_load(kbase + (size_t)(kt + 2) * 2 * kTile, lane);
            vb = x2_load(vbase + (size_t)(kt + 2) * 2 * kTile, lane);
            ATTN_TILE_X2(0, kn, vn)
        }
        {
            X2 kn = x2_load(kbase + (size_t)(kVT - 1) * 2 * kTile, lane), vn = x2_load(vbase + (size_t)(kVT - 1) * 2 * kTile, lane);
            ATTN_TILE_X2(kVT - 2, kb, vb)
            ATTN_TILE_X2(kVT - 1, kn, vn)
        }
    } else {
#pragma unroll 1
        for (int kt = 0; kt < kVT - 1; ++kt) {
            ATTN_TILE_X2(0, kb, vb)
            kb = x2_load(kbase + (size_t)(kt + 1) * 2 * kTile, lane);
            vb = x2_load(vbase + (size_t)(kt + 1) * 2 * kTile, lane);
        }
        ATTN_TILE_X2(kVT - 1, kb, vb)
    }
    l += xhalf(l);
    return (O + O2) * (1.0f / (16.0f * l));
}

#define PIN() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
// One wave = one stream of 32-query tiles (both heads each); tiles per wave = tiles_total / waves
template <int WPS, bool DB, int CUT = 0>
__global__ __launch_bounds__(256, WPS) void k_attn(const float* __restrict__ q, const float* __restrict__ kv, float* __restrict__ out, int tiles_per_wave) {
    const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    f32x16 acc = zero16();
    for (int t = 0; t < tiles_per_wave; ++t) {
        const float* qt = q + (size_t)((wave + t) % kVT) * 2 * kTile;
        PIN();
        acc += attn_head<DB, CUT>(qt, kv, kv + (size_t)kVT * 2 * kTile, lane);
        PIN();
        acc += attn_head<DB, CUT>(qt + kTile, kv + kTile, kv + (size_t)kVT * 2 * kTile + kTile, lane);
        PIN();
    }
    store_block(out + (size_t)wave * kTile, lane, acc);
}


// ---- ping-pong: an 8-wave workgroup, waves i and i + 4 on one SIMD, held in ANTI-PHASE by a workgroup barrier after every half step:
// while one of the pair runs its 12 MFMAs (P.V of tile t, S of tile t + 1) the other runs the softmax + split of its own tile.
// Left to themselves the two waves fall into lockstep (both in the matrix phase, then both in the vector phase) and the SIMD's
// time is the plain sum of its VALU and MFMA cycles.
#define PP_BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
template <int NBAR>
__device__ __forceinline__ f32x16 attn_head_pp(const float* __restrict__ qt, const float* __restrict__ kbase, const float* __restrict__ vbase, int lane, bool odd) {
    const int h = lane >> 5;
    const X2 qx = x2_load(qt, lane);
    f32x16 O = zero16(), O2 = zero16();
    float m = -1e30f, l = 0.f;
    X2 kb = x2_load(kbase, lane), vb = x2_load(vbase, lane);
    f32x16 S = x2_mma(kb, qx, zero16());                  // S(0)
    if (odd) PP_BAR();                                    // the odd half starts half a step late
#pragma unroll 1
    for (int kt = 0; kt < kVT; ++kt) {
        // ---- vector half step: softmax + split of tile kt
        __builtin_amdgcn_s_setprio(1);
        float bm = -1e30f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float sc = S[r];
            if (kt == kVT - 1 && kap(r) + 4 * h >= kV - 32 * (kVT - 1)) sc = -1e30f;
            S[r] = sc;
            bm = fmaxf(bm, sc);
        }
        bm = fmaxf(bm, xhalf(bm));
        if (!__all(bm <= m + 2048.0f)) {
            const float mn = fmaxf(m, bm);
            const float al = __builtin_amdgcn_exp2f((m - mn) * 0.00390625f);
            O = O * al; O2 = O2 * al; l *= al; m = mn;
        }
        const float off = 6.0f - m * 0.00390625f;
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pe = __builtin_amdgcn_exp2f(fmaf(S[r], 0.00390625f, off));
            S[r] = pe;
            ps += pe;
        }
        l += ps;
        const X2 px = x2_split(S);
        const X2 vcur = vb;
        const int kn = kt + 1 < kVT ? kt + 1 : kt;
        kb = x2_load(kbase + (size_t)kn * 2 * kTile, lane);
        vb = x2_load(vbase + (size_t)kn * 2 * kTile, lane);
        __builtin_amdgcn_s_setprio(0);
        if (NBAR >= 1) PP_BAR();
        // ---- matrix half step: P.V of tile kt, S of tile kt + 1
        O2 = x2_mma_small(vcur, px, O2);
        O = x2_mma_main(vcur, px, O);
        S = x2_mma(kb, qx, zero16());
        if (NBAR >= 2) PP_BAR();
    }
    if (!odd) PP_BAR();
    l += xhalf(l);
    return (O + O2) * (1.0f / (16.0f * l));
}
template <int NBAR>
__global__ __launch_bounds__(512, 2) void k_attn_pp(const float* __restrict__ q, const float* __restrict__ kv, float* __restrict__ out, int tiles_per_wave) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wave = blockIdx.x * 8 + wv;
    const bool odd = NBAR > 0 && wv >= 4;
    f32x16 acc = zero16();
    for (int t = 0; t < tiles_per_wave; ++t) {
        const float* qt = q + (size_t)((wave + t) % kVT) * 2 * kTile;
        PIN();
        acc += attn_head_pp<NBAR>(qt, kv, kv + (size_t)kVT * 2 * kTile, lane, odd);
        PIN();
        acc += attn_head_pp<NBAR>(qt + kTile, kv + kTile, kv + (size_t)kVT * 2 * kTile + kTile, lane, odd);
        PIN();
    }
    store_block(out + (size_t)wave * kTile, lane, acc);
}
template <int NBAR>
static void run_pp(const float* q, const float* kv, float* out, int n_cu, const char* name) {
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, (const void*)k_attn_pp<NBAR>);
    const int tpw = 24;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int it = 0; it < 6; ++it) {
        (void)hipEventRecord(e0, 0);
        k_attn_pp<NBAR><<<n_cu, 512>>>(q, kv, out, tpw);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (it > 0) best = std::min(best, ms);
    }
    printf("%-44s %3d VGPR %4zu B scratch | %d tiles per wave | %8.1f us | %6.0f ns per tile per SIMD\n", name, fa.numRegs, (size_t)fa.localSizeBytes, tpw,
           best * 1e3, best * 1e6 / (48.0));
}

template <int WPS, bool DB, int CUT = 0>
static void run(const float* q, const float* kv, float* out, int n_cu, const char* name) {
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, (const void*)k_attn<WPS, DB, CUT>);
    const int wgs = n_cu * WPS, waves = wgs * 4;
    const int total = n_cu * 4 * 12 * 4;                 // 12 x 4 tiles per SIMD for every variant
    const int tpw = total / waves;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int it = 0; it < 6; ++it) {
        (void)hipEventRecord(e0, 0);
        k_attn<WPS, DB, CUT><<<wgs, 256>>>(q, kv, out, tpw);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (it > 0) best = std::min(best, ms);
    }
    printf("%-44s %3d VGPR %4zu B scratch | %d tiles per wave | %8.1f us | %6.0f ns per tile per SIMD\n", name, fa.numRegs, (size_t)fa.localSizeBytes, tpw,
           best * 1e3, best * 1e6 / (48.0));
}

int main() {
    int dev = 0; hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, dev);
    const int n_cu = p.multiProcessorCount;
    const size_t nq = (size_t)kVT * 2 * kTile, nkv = (size_t)2 * kVT * 2 * kTile;
    std::vector<_Float16> h((nq + nkv) * 2);
    unsigned s = 12345;
    for (auto& x : h) { s = s * 1664525u + 1013904223u; x = (_Float16)(((int)(s >> 20) % 2001 - 1000) * 1e-3f); }
    float *q, *kv, *out;
    (void)hipMalloc(&q, nq * 4); (void)hipMalloc(&kv, nkv * 4); (void)hipMalloc(&out, (size_t)n_cu * 16 * kTile * 4);
    (void)hipMemcpy(q, h.data(), nq * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(kv, h.data() + nq * 2, nkv * 4, hipMemcpyHostToDevice);
    printf("attention loop alone, both heads of a 32-query tile, %d CUs; the same 48 tiles per SIMD in every variant\n", n_cu);
    run<1, true>(q, kv, out, n_cu, "1 wave / SIMD, K/V double-buffered");
    run<2, true>(q, kv, out, n_cu, "2 waves / SIMD, K/V double-buffered (shipped)");
    run<3, true>(q, kv, out, n_cu, "3 waves / SIMD, K/V double-buffered");
    run<3, false>(q, kv, out, n_cu, "3 waves / SIMD, one K/V set");
    run<4, false>(q, kv, out, n_cu, "4 waves / SIMD, one K/V set");
    run<2, false>(q, kv, out, n_cu, "2 waves / SIMD, one K/V set");
    printf("parts of the loop switched off (2 waves / SIMD, one K/V set; results meaningless, time only):\n");
    run<2, false, 1>(q, kv, out, n_cu, "no exp2 (16 of ~120 VALU)");
    run<2, false, 2>(q, kv, out, n_cu, "no two-plane split of P (one cvt per value)");
    run<2, false, 3>(q, kv, out, n_cu, "neither");
    run<2, false, 4>(q, kv, out, n_cu, "no MFMA (12 of 12), all VALU");
    run<1, false, 4>(q, kv, out, n_cu, "no MFMA, 1 wave / SIMD");
    run<1, false, 0>(q, kv, out, n_cu, "everything, 1 wave / SIMD, one K/V set");
    run<2, false, 16>(q, kv, out, n_cu, "S, O, O2 in AGPRs (inline-asm MFMAs), 2 waves");
    run<1, false, 16>(q, kv, out, n_cu, "S, O, O2 in AGPRs, 1 wave / SIMD");
    run<3, false, 16>(q, kv, out, n_cu, "S, O, O2 in AGPRs, 3 waves / SIMD");
    run<2, false, 32>(q, kv, out, n_cu, "the same matrix cycles on the 16x16x32 shape, 2 waves");
    run<2, false, 0>(q, kv, out, n_cu, "2 waves / SIMD, one K/V set (again)");
    printf("round 5: the tile maximum skipped (a norm bound would clear most tiles: 16 v_max + a lane exchange of ~120 VALU), time only:\n");
    run<2, true, 0>(q, kv, out, n_cu, "shipped loop (2 waves, double-buffered)");
    run<2, true, 64>(q, kv, out, n_cu, "shipped loop without the tile maximum");
    run<2, true, 0>(q, kv, out, n_cu, "shipped loop (again)");
    run<2, true, 64>(q, kv, out, n_cu, "shipped loop without the tile maximum (again)");
    printf("8-wave workgroups (2 waves / SIMD), software-pipelined loop (S of the next tile issued behind P.V):\n");
    run_pp<0>(q, kv, out, n_cu, "no barrier (free-running pair)");
    run_pp<2>(q, kv, out, n_cu, "ping-pong: barrier after each half step");
    run_pp<1>(q, kv, out, n_cu, "one barrier per key tile (same phase)");
    return 0;
}
