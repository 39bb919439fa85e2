// Round 6 (review item 6): the 431 x 431 self-attention loop of the MDR layers (fp32 configuration: Q, K, V, P on two fp16 planes) on the
// 16x16x32 MFMA shape, against the shipped loop on 32x32x16 -- one wall-clock verdict.
//
// The hardware guide ranks the shape separately: bare loops of v_mfma_f32_16x16x32_f16 hold ~1.15 x the FLOP/s of 32x32x16 on random data at
// equal matrix cycles, and DESIGN 4c' found the MDR launch bound by the energy of its MFMAs at the clock the chip holds.  Rounds 4 / 5 priced
// the shape with timing-only stand-ins.  Here the WHOLE loop is real on both shapes: same arithmetic (three partial products of two-plane
// operands, exp2-domain online softmax with lazy rescale, probabilities split in registers), same bytes per key tile (128 B per lane), same
// MFMA cycles per key tile (24 x 16 = 12 x 32 = 384), checked against a double-precision reference on the host.
//
// What makes the shape usable without lane traffic: operand layouts are ours.  A 16x16x32 B operand wants, in lane (n = l & 15, g = l >> 4),
// eight k-slots of column n.  The scores come out of S^T = K Q^T as 16 x 16 blocks whose lane (n, g) holds keys 4 g .. 4 g + 3 of query n, so
// the k-slot order of the P.V product is DEFINED as "slots 0..3 = keys 4 g + i of the first 16-key block, slots 4..7 = of the second": the
// probabilities become the B operand from the lane's own registers, and V^T is stored in the same slot order by whoever writes it (here: the
// host).  Costs of the shape inside the loop: a lane carries two queries (16 q + n), so the running maximum, its reference and the row sum
// exist twice, and the row maximum crosses four lane groups (v_permlane16_swap + v_permlane32_swap) instead of two.
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I gator_amd/csrc -I include tools/microbench/attn_1632.hip -o tools/microbench/attn_1632.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
#include "x3_common.h"
using namespace gator;

constexpr int kV = 431, kVT = 14, kTile = 1024;
typedef float f32x4_t __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

// ---- the shipped loop (mdr_fused.hip: self_attention_head_x2) ----------------------------------------------------------------------
#define ATTN_PV(VB, PX) { O2 = x2_mma_small(VB, PX, O2); O = x2_mma_main(VB, PX, O); }
#define ATTN_TILE_X2(KT, KB, VB)                                                                            \
    {                                                                                                       \
        __builtin_amdgcn_s_setprio(0);                                                                      \
        f32x16 S = x2_mma(KB, qx, zero16());                                                                \
        __builtin_amdgcn_s_setprio(1);                                                                      \
        float bm = -1e30f;                                                                                  \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                    \
            float sc = S[r];                                                                                \
            if ((KT) == kVT - 1 && kap(r) + 4 * h >= kV - 32 * (kVT - 1)) sc = -1e30f;                      \
            S[r] = sc;                                                                                      \
            bm = fmaxf(bm, sc);                                                                             \
        }                                                                                                   \
        bm = fmaxf(bm, xhalf(bm));                                                                          \
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
            const float pe = __builtin_amdgcn_exp2f(fmaf(S[r], 0.00390625f, off));                          \
            S[r] = pe;                                                                                      \
            ps += pe;                                                                                       \
        }                                                                                                   \
        l += ps;                                                                                            \
        const X2 px_ = x2_split(S);                                                                         \
        __builtin_amdgcn_s_setprio(0);                                                                      \
        ATTN_PV(VB, px_)                                                                                    \
    }
__device__ __forceinline__ f32x16 attn_head_ref(const float* __restrict__ qt, const float* __restrict__ kbase, const float* __restrict__ vbase, int lane) {
    const int h = lane >> 5;
    const X2 qx = x2_load(qt, lane);
    f32x16 O = zero16(), O2 = zero16();
    float m = -1e30f, l = 0.f;
    X2 kb = x2_load(kbase, lane), vb = x2_load(vbase, lane);
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
    l += xhalf(l);
    return (O + O2) * (1.0f / (16.0f * l));
}

// ---- the same loop on 16x16x32 -----------------------------------------------------------------------------------------------------
// One operand tile = 4 KiB as before: [plane 2][half 2][lane 64][8 halves].  K: half = key block (A operand: lane (m, g) = key 16 kh + m,
// channels 8 g ..); Q: half = query block (B operand: lane (n, g) = query 16 qh + n, channels 8 g ..); V: half = channel block (A operand:
// lane (m, g) = channel 16 dh + m, k-slots = keys 4 g + i | 16 + 4 g + i).
struct Y2 { f16x8 p[2][2]; };      // [plane][half]
__device__ __forceinline__ Y2 y2_load(const float* __restrict__ tile, int lane) {
    const f16x8* q = reinterpret_cast<const f16x8*>(tile) + lane;
    Y2 o;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) o.p[pl][hf] = q[(pl * 2 + hf) * 64];
    return o;
}
// max / sum over the four lane groups that share a query: v_permlane16_swap and v_permlane32_swap return both partners' values in every lane
__device__ __forceinline__ float max_groups(float v) {
    const unsigned u = __float_as_uint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float w = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const unsigned x = __float_as_uint(w);
    const auto b = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float sum_groups(float v) {
    const unsigned u = __float_as_uint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float w = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const unsigned x = __float_as_uint(w);
    const auto b = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
struct O16 { f32x4_t b[2][2]; };      // [channel block][query block]
template <int KT>
__device__ __forceinline__ void tile_1632(const Y2& kb, const Y2& vb, const Y2& qx, O16& O, O16& O2, float (&m)[2], float (&l)[2], int g) {
    const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_s_setprio(0);
    f32x4_t S[2][2];      // [key block][query block]: lane (n, g) holds keys 16 kh + 4 g + i of query 16 qh + n
    // product-major: the three partial products of a block are a dependent chain, and a 4-pass MFMA that reads the previous one's result stalls;
    // the four blocks are independent
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int qh = 0; qh < 2; ++qh) S[kh][qh] = MFMA16(kb.p[1][kh], qx.p[0][qh], z);                   // lo * hi
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int qh = 0; qh < 2; ++qh) S[kh][qh] = MFMA16(kb.p[0][kh], qx.p[1][qh], S[kh][qh]);           // hi * lo
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int qh = 0; qh < 2; ++qh) S[kh][qh] = MFMA16(kb.p[0][kh], qx.p[0][qh], S[kh][qh]);           // hi * hi
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    float bm[2] = {-1e30f, -1e30f};
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int qh = 0; qh < 2; ++qh)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float sc = S[kh][qh][i];
                if (KT == kVT - 1 && 16 * kh + 4 * g + i >= kV - 32 * (kVT - 1)) sc = -1e30f;      // keys 431..447 do not exist
                S[kh][qh][i] = sc;
                bm[qh] = fmaxf(bm[qh], sc);
            }
    bm[0] = max_groups(bm[0]);
    bm[1] = max_groups(bm[1]);
    if (!__all(bm[0] <= m[0] + 2048.0f && bm[1] <= m[1] + 2048.0f)) {
#pragma unroll
        for (int qh = 0; qh < 2; ++qh) {
            const float mn = fmaxf(m[qh], bm[qh]);
            const float al = __builtin_amdgcn_exp2f((m[qh] - mn) * 0.00390625f);
#pragma unroll
            for (int dh = 0; dh < 2; ++dh) { O.b[dh][qh] = O.b[dh][qh] * al; O2.b[dh][qh] = O2.b[dh][qh] * al; }
            l[qh] *= al;
            m[qh] = mn;
        }
    }
    f16x8 ph[2], plo[2];
#pragma unroll
    for (int qh = 0; qh < 2; ++qh) {
        const float off = 6.0f - m[qh] * 0.00390625f;
        float ps = 0.f;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float pe = __builtin_amdgcn_exp2f(fmaf(S[kh][qh][i], 0.00390625f, off));
                ps += pe;
                const _Float16 hi = (_Float16)pe;
                ph[qh][4 * kh + i] = hi;                              // k-slot 4 kh + i of this lane group: the lane's own registers
                plo[qh][4 * kh + i] = (_Float16)(pe - (float)hi);
            }
        l[qh] += ps;
    }
    __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int qh = 0; qh < 2; ++qh) O2.b[dh][qh] = MFMA16(vb.p[1][dh], ph[qh], O2.b[dh][qh]);      // lo * hi
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int qh = 0; qh < 2; ++qh) O2.b[dh][qh] = MFMA16(vb.p[0][dh], plo[qh], O2.b[dh][qh]);     // hi * lo
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int qh = 0; qh < 2; ++qh) O.b[dh][qh] = MFMA16(vb.p[0][dh], ph[qh], O.b[dh][qh]);      // hi * hi
}
__device__ __forceinline__ f32x16 attn_head_1632(const float* __restrict__ qt, const float* __restrict__ kbase, const float* __restrict__ vbase, int lane) {
    const int g = lane >> 4;
    const Y2 qx = y2_load(qt, lane);
    const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
    O16 O, O2;
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int qh = 0; qh < 2; ++qh) { O.b[dh][qh] = z; O2.b[dh][qh] = z; }
    float m[2] = {-1e30f, -1e30f}, l[2] = {0.f, 0.f};
    Y2 kb = y2_load(kbase, lane), vb = y2_load(vbase, lane);
#pragma unroll 1
    for (int kt = 0; kt < kVT - 2; kt += 2) {
        Y2 kn = y2_load(kbase + (size_t)(kt + 1) * 2 * kTile, lane), vn = y2_load(vbase + (size_t)(kt + 1) * 2 * kTile, lane);
        tile_1632<0>(kb, vb, qx, O, O2, m, l, g);
        kb = y2_load(kbase + (size_t)(kt + 2) * 2 * kTile, lane);
        vb = y2_load(vbase + (size_t)(kt + 2) * 2 * kTile, lane);
        tile_1632<0>(kn, vn, qx, O, O2, m, l, g);
    }
    {
        Y2 kn = y2_load(kbase + (size_t)(kVT - 1) * 2 * kTile, lane), vn = y2_load(vbase + (size_t)(kVT - 1) * 2 * kTile, lane);
        tile_1632<kVT - 2>(kb, vb, qx, O, O2, m, l, g);
        tile_1632<kVT - 1>(kn, vn, qx, O, O2, m, l, g);
    }
    f32x16 out;      // [dh][qh][i]: channel 16 dh + 4 g + i of query 16 qh + (lane & 15)
#pragma unroll
    for (int qh = 0; qh < 2; ++qh) {
        const float inv = 1.0f / (16.0f * sum_groups(l[qh]));
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int i = 0; i < 4; ++i) out[(dh * 2 + qh) * 4 + i] = (O.b[dh][qh][i] + O2.b[dh][qh][i]) * inv;
    }
    return out;
}

#define PIN() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
template <int WPS, int MODE>
__global__ __launch_bounds__(256, WPS) void k_attn(const float* __restrict__ q, const float* __restrict__ kv, float* __restrict__ out, int tiles_per_wave,
                                                   unsigned long long* __restrict__ stamps) {
    const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    f32x16 acc = zero16();
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int t = 0; t < tiles_per_wave; ++t) {
        const float* qt = q + (size_t)((wave + t) % kVT) * 2 * kTile;
        PIN();
        if (MODE == 0) acc += attn_head_ref(qt, kv, kv + (size_t)kVT * 2 * kTile, lane);
        else acc += attn_head_1632(qt, kv, kv + (size_t)kVT * 2 * kTile, lane);
        PIN();
        if (MODE == 0) acc += attn_head_ref(qt + kTile, kv + kTile, kv + (size_t)kVT * 2 * kTile + kTile, lane);
        else acc += attn_head_1632(qt + kTile, kv + kTile, kv + (size_t)kVT * 2 * kTile + kTile, lane);
        PIN();
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    store_block(out + (size_t)wave * kTile, lane, acc);
    if (lane == 0) { stamps[2 * wave] = c1 - c0; stamps[2 * wave + 1] = r1 - r0; }
}
// one head of one query tile, for the check against the host
template <int MODE>
__global__ __launch_bounds__(64) void k_one(const float* __restrict__ q, const float* __restrict__ kv, float* __restrict__ out) {
    const int lane = threadIdx.x;
    const f32x16 o = MODE == 0 ? attn_head_ref(q, kv, kv + (size_t)kVT * 2 * kTile, lane) : attn_head_1632(q, kv, kv + (size_t)kVT * 2 * kTile, lane);
    store_block(out, lane, o);
}

static unsigned long long* g_stamps = nullptr;
template <int WPS, int MODE>
static double run(const float* q, const float* kv, float* out, int n_cu, const char* name) {
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, (const void*)k_attn<WPS, MODE>);
    const int wgs = n_cu * WPS, waves = wgs * 4;
    const int total = n_cu * 4 * 12 * 4;                 // 48 tiles per SIMD for every variant
    const int tpw = total / waves;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int it = 0; it < 6; ++it) {
        (void)hipEventRecord(e0, 0);
        k_attn<WPS, MODE><<<wgs, 256>>>(q, kv, out, tpw, g_stamps);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (it > 0) best = std::min(best, ms);
    }
    std::vector<unsigned long long> st((size_t)2 * waves);
    (void)hipMemcpy(st.data(), g_stamps, st.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, tick = 0;
    for (int w = 0; w < waves; ++w) { cyc += (double)st[2 * w]; tick += (double)st[2 * w + 1]; }
    const double per_step = cyc / waves / (tpw * 2.0 * kVT);
    printf("%-44s %3d VGPR %4zu B scr | %2d tiles/wave | %7.1f us | %6.0f wave-cyc/step = %5.0f SIMD-cyc/step | %.2f GHz\n", name, fa.numRegs,
           (size_t)fa.localSizeBytes, tpw, best * 1e3, per_step, per_step / WPS, cyc / (tick * 10.0));
    fflush(stdout);
    return best * 1e3;
}

// logical operands of the check: q [32][32], k, v [448][32] as hi + lo of the halves the device sees
static float h2f(_Float16 a, _Float16 b) { return (float)a + (float)b; }

int main(int argc, char** argv) {
    int dev = 0; hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, dev);
    const int n_cu = p.multiProcessorCount;
    const size_t nq = (size_t)kVT * 2 * kTile, nkv = (size_t)2 * kVT * 2 * kTile;
    std::vector<_Float16> h((nq + nkv) * 2);
    unsigned s = 12345;
    // hi planes random in [-1, 1]; lo planes 2^-11 of that (what a split leaves): the device buffers are [tile][plane][half][lane][8]
    for (size_t i = 0; i < h.size(); ++i) {
        s = s * 1664525u + 1013904223u;
        const float v = ((int)(s >> 20) % 2001 - 1000) * 1e-3f;
        const bool lo_plane = ((i / 1024) & 1) != 0;      // halves [0, 1024) of every 2048: plane 0, the rest plane 1
        h[i] = (_Float16)(lo_plane ? v * 4.8828125e-4f : v);
    }
    const bool zeros = argc > 1 && !strcmp(argv[1], "zeros");
    if (zeros) std::fill(h.begin(), h.end(), (_Float16)0.0f);
    float *q, *kv, *out;
    (void)hipMalloc(&q, nq * 4); (void)hipMalloc(&kv, nkv * 4); (void)hipMalloc(&out, (size_t)n_cu * 16 * kTile * 4);
    (void)hipMalloc(&g_stamps, (size_t)n_cu * 16 * 2 * 8);
    (void)hipMemcpy(q, h.data(), nq * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(kv, h.data() + nq * 2, nkv * 4, hipMemcpyHostToDevice);
    if (zeros) printf("ALL-ZERO OPERANDS\n");
    if (!zeros) {
        // ---- check of the 16x16x32 loop: head 0 of query tile 0 against softmax(q k^T / 256) v in double, from the SAME halves under this loop's layout
        const _Float16* hq = h.data();                       // tile 0 of q: [plane][qh][lane][8]
        const _Float16* hk = h.data() + nq * 2;              // K tiles of head 0: tile kt at kt * 2 * 2048 halves
        const _Float16* hv = hk + (size_t)kVT * 2 * 2048;
        auto at = [](const _Float16* t, int pl, int hf, int lane, int j) { return t[((pl * 2 + hf) * 64 + lane) * 8 + j]; };
        std::vector<double> Q(32 * 32), K(448 * 32), Vv(448 * 32);
        for (int qh = 0; qh < 2; ++qh) for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 8; ++j)
            Q[(16 * qh + (lane & 15)) * 32 + 8 * (lane >> 4) + j] = h2f(at(hq, 0, qh, lane, j), at(hq, 1, qh, lane, j));
        for (int kt = 0; kt < kVT; ++kt) {
            const _Float16* tk = hk + (size_t)kt * 2 * 2048;
            const _Float16* tv = hv + (size_t)kt * 2 * 2048;
            for (int hf = 0; hf < 2; ++hf) for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 8; ++j) {
                K[(32 * kt + 16 * hf + (lane & 15)) * 32 + 8 * (lane >> 4) + j] = h2f(at(tk, 0, hf, lane, j), at(tk, 1, hf, lane, j));
                const int key = 32 * kt + (j < 4 ? 4 * (lane >> 4) + j : 16 + 4 * (lane >> 4) + (j - 4));      // the k-slot order of the P.V product
                Vv[key * 32 + 16 * hf + (lane & 15)] = h2f(at(tv, 0, hf, lane, j), at(tv, 1, hf, lane, j));
            }
        }
        std::vector<double> want(32 * 32);
        for (int qi = 0; qi < 32; ++qi) {
            std::vector<double> sc(kV);
            double mx = -1e300;
            for (int key = 0; key < kV; ++key) {
                double d = 0;
                for (int c = 0; c < 32; ++c) d += Q[qi * 32 + c] * K[key * 32 + c];
                sc[key] = d / 256.0;      // exp2 domain, as the loop has it
                mx = std::max(mx, sc[key]);
            }
            double den = 0;
            for (int key = 0; key < kV; ++key) { sc[key] = std::exp2(sc[key] - mx); den += sc[key]; }
            for (int c = 0; c < 32; ++c) {
                double o = 0;
                for (int key = 0; key < kV; ++key) o += sc[key] * Vv[key * 32 + c];
                want[qi * 32 + c] = o / den / 16.0;
            }
        }
        k_one<1><<<1, 64>>>(q, kv, out);
        std::vector<float> got(kTile);
        (void)hipMemcpy(got.data(), out, kTile * 4, hipMemcpyDeviceToHost);
        double err = 0, sc_ = 0;
        for (int lane = 0; lane < 64; ++lane)
            for (int r = 0; r < 16; ++r) {      // store_block: out[(r >> 2) * 256 + lane * 4 + (r & 3)] = acc[r]; acc[(dh * 2 + qh) * 4 + i]
                const int dh = r >> 3, qh = (r >> 2) & 1, i = r & 3;
                const double w = want[(16 * qh + (lane & 15)) * 32 + 16 * dh + 4 * (lane >> 4) + i];
                err = std::max(err, std::fabs((double)got[(r >> 2) * 256 + lane * 4 + (r & 3)] - w));
                sc_ = std::max(sc_, std::fabs(w));
            }
        printf("check of the 16x16x32 loop (head 0, query tile 0) against the host in double: max |err| %.3e on a scale of %.3e\n", err, sc_);
    }
    printf("attention loop alone, both heads of a 32-query tile, %d CUs; 48 tiles per SIMD in every variant; two runs of each, interleaved\n", n_cu);
    double t[2][2][2];
    for (int rep = 0; rep < 2; ++rep) {
        t[rep][0][0] = run<2, 0>(q, kv, out, n_cu, "shipped loop 32x32x16, 2 waves / SIMD");
        t[rep][0][1] = run<2, 1>(q, kv, out, n_cu, "16x16x32 loop, 2 waves / SIMD");
        t[rep][1][0] = run<1, 0>(q, kv, out, n_cu, "shipped loop 32x32x16, 1 wave / SIMD");
        t[rep][1][1] = run<1, 1>(q, kv, out, n_cu, "16x16x32 loop, 1 wave / SIMD");
    }
    for (int o = 0; o < 2; ++o)
        printf("%d wave(s) / SIMD: 16x16x32 / shipped = %.4f, %.4f (wall time)\n", 2 - o, t[0][o][1] / t[0][o][0], t[1][o][1] / t[1][o][0]);
    return 0;
}
