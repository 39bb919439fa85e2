// Round 6 (review item 2, the gate it asked for): the one-plane ("X1", BASELINE config 3) 431 x 431 self-attention loop of the MDR layers
// fed from the L2 -- the shipped form: one wave per (sample, query tile), K / V tiles global -> registers two tiles ahead, four waves of a
// workgroup sharing them through the L1 -- against the workgroup-per-sample form the review proposed: a sample's K and V of both heads
// (14 tiles x 2 heads x 2 x 2 KiB = 112 KiB of fp16) resident in ONE CU's LDS, eight (or seven) waves, each running two query tiles.
//
// Same loop body on both sides (mdr_fused.hip ATTN_TILE_X1: scores in the exp2 domain on the accumulator's initial value, exp2 + row sum,
// one conversion, the long way when a row sum overflows), same tile layout (lane l reads 16 B at l and at 64 + l of a 2 KiB tile: coalesced
// from global memory, conflict-free from LDS), random fp16 operands, wall clock by HIP events and cycles by s_memtime.
//   lds      : K / V staged global -> LDS by the workgroup for every unit of work (what a stage costs if K / V had to come from memory)
//   lds-res  : staged once, the attention of the sample repeated `reps` times (K / V written to LDS by the stage that produced them:
//              the form the review describes; the staging is amortised away)
//   l2       : the shipped form, the attention of a unit repeated `reps` times as well
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I gator_amd/csrc -I include tools/microbench/attn_x1_lds.hip -o tools/microbench/attn_x1_lds.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
#include "x3_common.h"
using namespace gator;

constexpr int kV = 431, kVT = 14, kTile = 1024, kTileX1 = kTile / 2;      // a 2 KiB tile = 512 floats' worth of bytes

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool LDS>
__device__ __forceinline__ X1 tile_load(const float* tile, int lane) {
    if constexpr (LDS) {
        // the same 16 B per lane at l and 64 + l, from LDS: two ds_read_b128
        typedef __attribute__((address_space(3))) const f16x8* LP;
        LP q = (LP)(uintptr_t)(uint32_t)(uintptr_t)tile + lane;
        X1 o;
        o.p[0] = q[0];
        o.p[1] = q[64];
        return o;
    } else {
        return x1_load(tile, lane);
    }
}

#define ATTN_TILE_X1(KT, KB, VB)                                                                            \
    {                                                                                                       \
        f32x16 S = x1_mma(KB, qx, Ci);                                                                      \
        if ((KT) == kVT - 1) {                                                                              \
            _Pragma("unroll") for (int r = 0; r < 16; ++r)                                                  \
                if (kap(r) + 4 * h >= kV - 32 * (kVT - 1)) S[r] = -1e30f;                                   \
        }                                                                                                   \
        float ps = 0.f;                                                                                     \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                    \
            S[r] = __builtin_amdgcn_exp2f(S[r]);                                                            \
            ps += S[r];                                                                                     \
        }                                                                                                   \
        if (!__all(ps < 32768.0f)) {                                                                        \
            f32x16 R = x1_mma(KB, qx, zero16());                                                            \
            float bm = -1e30f;                                                                              \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                \
                if ((KT) == kVT - 1 && kap(r) + 4 * h >= kV - 32 * (kVT - 1)) R[r] = -1e30f;                \
                bm = fmaxf(bm, R[r]);                                                                       \
            }                                                                                               \
            bm = fmaxf(bm, xhalf(bm));                                                                      \
            const float mn = fmaxf(m, bm);                                                                  \
            const float al = __builtin_amdgcn_exp2f(m - mn);                                                \
            O = O * al;                                                                                     \
            l *= al;                                                                                        \
            m = mn;                                                                                         \
            Ci = f32x16(6.0f - m);                                                                          \
            ps = 0.f;                                                                                       \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                \
                S[r] = __builtin_amdgcn_exp2f(R[r] + Ci[r]);                                                \
                ps += S[r];                                                                                 \
            }                                                                                               \
        }                                                                                                   \
        l += ps;                                                                                            \
        O = x1_mma(VB, x1_cvt(S), O);                                                                       \
    }

// one head of one query tile; K / V tile kt of this head at kbase + kt * stride (floats)
template <bool LDS>
__device__ __forceinline__ f32x16 attn_head(const float* __restrict__ qt, const float* kbase, const float* vbase, int stride, int lane) {
    const int h = lane >> 5;
    const X1 qx = x1_load(qt, lane);
    f32x16 O = zero16();
    float m = -1e30f, l = 0.f;
    f32x16 Ci = f32x16(1e30f);
    X1 kb = tile_load<LDS>(kbase, lane), vb = tile_load<LDS>(vbase, lane);
#pragma unroll 1
    for (int kt = 0; kt < kVT - 2; kt += 2) {
        X1 kn = tile_load<LDS>(kbase + (size_t)(kt + 1) * stride, lane), vn = tile_load<LDS>(vbase + (size_t)(kt + 1) * stride, lane);
        ATTN_TILE_X1(0, kb, vb)
        kb = tile_load<LDS>(kbase + (size_t)(kt + 2) * stride, lane);
        vb = tile_load<LDS>(vbase + (size_t)(kt + 2) * stride, lane);
        ATTN_TILE_X1(0, kn, vn)
    }
    {
        X1 kn = tile_load<LDS>(kbase + (size_t)(kVT - 1) * stride, lane), vn = tile_load<LDS>(vbase + (size_t)(kVT - 1) * stride, lane);
        ATTN_TILE_X1(kVT - 2, kb, vb)
        ATTN_TILE_X1(kVT - 1, kn, vn)
    }
    l += xhalf(l);
    return O * (1.0f / l);
}

// ---- the shipped form: one wave per (sample, query tile), four waves per workgroup, two workgroups per CU, K / V from the L2 ------------
// Tiles of a sample: [kVT][2 heads][kTileX1 floats] for each of Q, K, V, O.
__global__ __launch_bounds__(256, 2) void k_l2(const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ Vv, float* __restrict__ Oo,
                                               int units, int reps, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long t0 = __builtin_readcyclecounter();
    // persistent, as in the library (there: tickets): the grid is two workgroups per CU and a workgroup walks the units with the grid's stride, so that
    // no workgroup dispatch stands between two units (a grid of one workgroup per four units measured 850 us for what takes 560 us of wave cycles)
    for (int u0 = blockIdx.x * 4; u0 < units; u0 += gridDim.x * 4) {
        const int unit = u0 + wave;
        if (unit >= units) break;
        const int b = unit / kVT, tile = unit % kVT;
        for (int r = 0; r < reps; ++r)
#pragma unroll 1
            for (int hd = 0; hd < 2; ++hd) {
                const f32x16 o = attn_head<false>(Q + ((size_t)(b * kVT + tile) * 2 + hd) * kTileX1, K + ((size_t)b * kVT * 2 + hd) * kTileX1,
                                                  Vv + ((size_t)b * kVT * 2 + hd) * kTileX1, 2 * kTileX1, lane);
                x1_store(Oo + ((size_t)(b * kVT + tile) * 2 + hd) * kTileX1, lane, x1_cvt(o));
            }
    }
    if (cyc && lane == 0) atomicAdd(cyc, __builtin_readcyclecounter() - t0);
}

// ---- the workgroup-per-sample form: K / V of both heads in LDS, NW waves, wave w runs query tiles w, w + NW ----------------------------
template <int NW>
__global__ __launch_bounds__(NW * 64, 1) void k_lds(const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ Vv, float* __restrict__ Oo,
                                                    int samples, int reps, int restage, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) float lds[];          // [K | V][kVT][2][kTileX1]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int kPer = kVT * 2 * kTileX1;                             // floats of K (or V) of one sample: 14 336 = 56 KiB
    float* Ks = lds;
    float* Vs = lds + kPer;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int b = blockIdx.x; b < samples; b += gridDim.x) {
        for (int r = 0; r < reps; ++r) {
            if (r == 0 || restage) {
                __syncthreads();                                        // everyone is done with the previous contents
                const f32x4* gk = reinterpret_cast<const f32x4*>(K + (size_t)b * kPer);
                const f32x4* gv = reinterpret_cast<const f32x4*>(Vv + (size_t)b * kPer);
                f32x4* sk = reinterpret_cast<f32x4*>(Ks);
                f32x4* sv = reinterpret_cast<f32x4*>(Vs);
                for (int i = threadIdx.x; i < kPer / 4; i += NW * 64) { sk[i] = gk[i]; sv[i] = gv[i]; }
                __syncthreads();
            }
#pragma unroll 1
            for (int tile = wave; tile < kVT; tile += NW)
#pragma unroll 1
                for (int hd = 0; hd < 2; ++hd) {
                    const f32x16 o = attn_head<true>(Q + ((size_t)(b * kVT + tile) * 2 + hd) * kTileX1, Ks + (size_t)hd * kTileX1, Vs + (size_t)hd * kTileX1,
                                                     2 * kTileX1, lane);
                    x1_store(Oo + ((size_t)(b * kVT + tile) * 2 + hd) * kTileX1, lane, x1_cvt(o));
                }
        }
    }
    if (cyc && lane == 0) atomicAdd(cyc, __builtin_readcyclecounter() - t0);
}


// ---- inside the ticket design: the four waves of a workgroup run four consecutive query tiles of ONE sample (as the library's tickets mostly do) and share every
// K / V tile through a two-slot LDS ring: each tile leaves the L2 once per workgroup (one 16 B load per thread) instead of once per wave; one workgroup barrier per
// key tile keeps the four waves in step.
template <int KT_LAST>
__device__ __forceinline__ void ring_tile(const X1& kb, const X1& vb, const X1& qx, f32x16& O, f32x16& Ci, float& m, float& l, int h) {
    ATTN_TILE_X1(KT_LAST, kb, vb)
}
__global__ __launch_bounds__(256, 2) void k_ring(const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ Vv, float* __restrict__ Oo,
                                                 int samples, int reps, unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) float ring[2][2][kTileX1];      // [slot][K | V]: 8 KiB
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, h = lane >> 5;
    const unsigned long long t0 = __builtin_readcyclecounter();
    const int ngroups = samples * 4;
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int b = grp >> 2, tile = (grp & 3) * 4 + wave;
        const bool valid = tile < kVT;
        for (int r = 0; r < reps; ++r)
#pragma unroll 1
            for (int hd = 0; hd < 2; ++hd) {
                const f32x4* src = reinterpret_cast<const f32x4*>((t < 128 ? K : Vv) + ((size_t)b * kVT * 2 + hd) * kTileX1) + (t & 127);     // + kt * (2 * kTileX1 / 4)
                f32x4* dst0 = reinterpret_cast<f32x4*>(&ring[0][t >> 7][0]) + (t & 127);
                f32x4* dst1 = reinterpret_cast<f32x4*>(&ring[1][t >> 7][0]) + (t & 127);
                const X1 qx = x1_load(Q + ((size_t)(b * kVT + (valid ? tile : 0)) * 2 + hd) * kTileX1, lane);
                f32x16 O = zero16();
                float m = -1e30f, l = 0.f;
                f32x16 Ci = f32x16(1e30f);
                __syncthreads();                         // the ring is free
                *dst0 = src[0];
                __syncthreads();
#pragma unroll 1
                for (int kt = 0; kt < kVT - 1; ++kt) {
                    const f32x4 nxt = src[(size_t)(kt + 1) * (2 * kTileX1 / 4)];           // the next tile's 16 B of this thread: in flight over this tile's work
                    const float* slot = &ring[kt & 1][0][0];
                    const X1 kb = tile_load<true>(slot, lane), vb = tile_load<true>(slot + kTileX1, lane);
                    if (valid) ring_tile<0>(kb, vb, qx, O, Ci, m, l, h);
                    *((kt & 1) ? dst0 : dst1) = nxt;
                    __syncthreads();
                }
                {
                    const float* slot = &ring[(kVT - 1) & 1][0][0];
                    const X1 kb = tile_load<true>(slot, lane), vb = tile_load<true>(slot + kTileX1, lane);
                    if (valid) ring_tile<kVT - 1>(kb, vb, qx, O, Ci, m, l, h);
                }
                if (valid) {
                    l += xhalf(l);
                    x1_store(Oo + ((size_t)(b * kVT + tile) * 2 + hd) * kTileX1, lane, x1_cvt(O * (1.0f / l)));
                }
            }
    }
    if (cyc && lane == 0) atomicAdd(cyc, __builtin_readcyclecounter() - t0);
}

static uint16_t f2h(float f) {
    _Float16 h = (_Float16)f;
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}

int main(int argc, char** argv) {
    const int samples = argc > 1 ? atoi(argv[1]) : 2048;
    const int reps = argc > 2 ? atoi(argv[2]) : 3;
    const size_t per = (size_t)kVT * 2 * kTileX1;                       // floats per sample and tensor
    const size_t n = per * samples;
    std::vector<uint16_t> hq(2 * n), hk(2 * n), hv(2 * n);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (size_t i = 0; i < 2 * n; ++i) { hq[i] = f2h(rnd() * 2.0f); hk[i] = f2h(rnd() * 2.0f); hv[i] = f2h(rnd() * 16.0f); }
    float *Q, *K, *V, *O1, *O2;
    unsigned long long* cyc;
    CHECK(hipMalloc(&Q, n * 4)); CHECK(hipMalloc(&K, n * 4)); CHECK(hipMalloc(&V, n * 4)); CHECK(hipMalloc(&O1, n * 4)); CHECK(hipMalloc(&O2, n * 4));
    CHECK(hipMalloc(&cyc, 8));
    CHECK(hipMemcpy(Q, hq.data(), n * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(K, hk.data(), n * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(V, hv.data(), n * 4, hipMemcpyHostToDevice));
    const int units = samples * kVT;
    const size_t lds_bytes = 2 * per * 4;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds<7>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch, int waves_total) {
        float best = 1e30f, sum = 0.f;
        unsigned long long c = 0;
        for (int it = 0; it < 7; ++it) {
            CHECK(hipMemset(cyc, 0, 8));
            CHECK(hipEventRecord(e0));
            launch();
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipGetLastError());
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (it >= 2) { best = std::min(best, ms); sum += ms; }
            CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
        }
        const double att = (double)samples * reps;                     // (sample, layer) attentions
        printf("%-34s  %8.1f us best  %8.1f us mean  | %6.3f us per sample-attention  | %7.0f cycles per wave  (%d waves)\n", name, best * 1e3, sum / 5 * 1e3,
               best * 1e3 / att, (double)c / waves_total, waves_total);
    };
    printf("one-plane self-attention of %d samples x %d repetitions, %d CUs; random fp16 operands\n", samples, reps, ncu);
    run("l2   (shipped: wave per tile)", [&] { hipLaunchKernelGGL(k_l2, dim3(2 * ncu), dim3(256), 0, 0, Q, K, V, O1, units, reps, cyc); }, 2 * ncu * 4);
    run("l2   one workgroup per 4 tiles", [&] { hipLaunchKernelGGL(k_l2, dim3((units + 3) / 4), dim3(256), 0, 0, Q, K, V, O1, units, reps, cyc); }, units);
    run("lds  8 waves, staged per attention", [&] { hipLaunchKernelGGL(k_lds<8>, dim3(ncu), dim3(512), lds_bytes, 0, Q, K, V, O2, samples, reps, 1, cyc); }, ncu * 8);
    run("lds  8 waves, K / V resident", [&] { hipLaunchKernelGGL(k_lds<8>, dim3(ncu), dim3(512), lds_bytes, 0, Q, K, V, O2, samples, reps, 0, cyc); }, ncu * 8);
    run("lds  7 waves, staged per attention", [&] { hipLaunchKernelGGL(k_lds<7>, dim3(ncu), dim3(448), lds_bytes, 0, Q, K, V, O2, samples, reps, 1, cyc); }, ncu * 7);
    run("lds  7 waves, K / V resident", [&] { hipLaunchKernelGGL(k_lds<7>, dim3(ncu), dim3(448), lds_bytes, 0, Q, K, V, O2, samples, reps, 0, cyc); }, ncu * 7);
    run("ring 4 waves share K / V tiles in LDS", [&] { hipLaunchKernelGGL(k_ring, dim3(2 * ncu), dim3(256), 0, 0, Q, K, V, O2, samples, reps, cyc); }, 2 * ncu * 4);
    // the two forms compute the same thing
    std::vector<uint16_t> a(2 * n), b2(2 * n);
    CHECK(hipMemcpy(a.data(), O1, n * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(b2.data(), O2, n * 4, hipMemcpyDeviceToHost));
    size_t diff = 0;
    for (size_t i = 0; i < 2 * n; ++i) diff += a[i] != b2[i];
    printf("outputs of the two forms differ in %zu of %zu fp16 values\n", diff, 2 * n);
    return 0;
}
