// What does the per-sample weight stream of the GAT encoder cost by itself?  One workgroup per CU (256 workgroups), every wave
// walks its own share of the SAME tile sequence (all workgroups in step, as k_gat does): 1 584 operand tiles per sample
// (264 tile products x 6 blocks), 6 KiB each as three bf16 planes (9.5 MB) or 4 KiB each as fp32 (6.3 MB).
//   reg<TD, NT, WAVES, MF>: tiles loaded to registers by global_load_dwordx4, NT tiles in flight per wave, MF MFMAs per tile
//   dma<TD, R, WAVES, MF>:  tiles copied by LDS-DMA into a wave-private ring of R slots, read back with ds_read_b128
// Prints us per launch, GB/s per CU, bytes per clock per CU (at 2.1 GHz nominal).
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench/wstream.hip -o tools/microbench/wstream.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "x3_common.h"
using gator::X3; using gator::x3_split; using gator::x3_store; using gator::pk_fma;
#if 0
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#endif
using gator::f32x16; using gator::f32x4; using gator::f32x2; using gator::bf16x8;

constexpr int kTiles = 1584;

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int TD, int NT, int WAVES, int MF>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void k_reg(const f32x4* __restrict__ W, float* out) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int per = kTiles / WAVES;
    f32x4 buf[NT][TD];
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    bf16x8 bop;
    for (int j = 0; j < 8; ++j) bop[j] = (__bf16)(1.0f + j);
    auto ld = [&](int i, f32x4(&dst)[TD]) {
        const f32x4* p = W + ((size_t)(i * WAVES + wave) * TD) * 64 + lane;
#pragma unroll
        for (int k = 0; k < TD; ++k) dst[k] = p[k * 64];
    };
#pragma unroll
    for (int s = 0; s < NT; ++s) ld(s, buf[s]);
    asm volatile("" ::: "memory");
    for (int i0 = 0; i0 < per; i0 += NT) {
#pragma unroll
        for (int s = 0; s < NT; ++s) {
            const int i = i0 + s;
            // consume tile i (slot s)
            if (MF) {
#pragma unroll
                for (int m = 0; m < MF; ++m) {
                    bf16x8 a = __builtin_bit_cast(bf16x8, buf[s][m % TD]);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bop, acc, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int k = 0; k < TD; ++k) asm volatile("" ::"v"(buf[s][k]));
            }
            const int nx = i + NT < per ? i + NT : i;
            ld(nx, buf[s]);
            asm volatile("" ::: "memory");
        }
    }
    float sum = 0.f;
    for (int r = 0; r < 16; ++r) sum += acc[r];
    if (sum == 123.456f) out[blockIdx.x] = sum;
}

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_dst) : "memory");
}

template <int TD, int R, int WAVES, int MF>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void k_dma(const f32x4* __restrict__ W, float* out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int per = kTiles / WAVES;
    constexpr int kSlot = TD * 256;      // floats per slot
    float* ring = lds + wave * R * kSlot;
    const unsigned ring_b = (unsigned)(unsigned long long)ring;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    bf16x8 bop;
    for (int j = 0; j < 8; ++j) bop[j] = (__bf16)(1.0f + j);
    auto issue = [&](int i, int slot) {
        const f32x4* p = W + ((size_t)(i * WAVES + wave) * TD) * 64 + lane;
#pragma unroll
        for (int k = 0; k < TD; ++k) glds16(p + k * 64, ring_b + (slot * kSlot + k * 256) * 4);
    };
#pragma unroll
    for (int s = 0; s < R; ++s) issue(s, s);
    for (int i0 = 0; i0 < per; i0 += R) {
#pragma unroll
        for (int s = 0; s < R; ++s) {
            const int i = i0 + s;
            wait_vm<(R - 1) * TD>();
            f32x4 t[TD];
#pragma unroll
            for (int k = 0; k < TD; ++k) t[k] = reinterpret_cast<const f32x4*>(ring + s * kSlot + k * 256)[lane];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int nx = i + R < per ? i + R : i;
            issue(nx, s);
            if (MF) {
#pragma unroll
                for (int m = 0; m < MF; ++m) {
                    bf16x8 a = __builtin_bit_cast(bf16x8, t[m % TD]);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bop, acc, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int k = 0; k < TD; ++k) asm volatile("" ::"v"(t[k]));
            }
        }
    }
    wait_vm<0>();
    float sum = 0.f;
    for (int r = 0; r < 16; ++r) sum += acc[r];
    if (sum == 123.456f) out[blockIdx.x] = sum;
}

static float* g_out;
static f32x4* g_w;
static int g_nwg = 256;

// the same launch after 1 GiB of unrelated stores (the weights are then neither in an L2 nor in the Infinity Cache): median of 7
static char* g_flush = nullptr;
template <typename F> double time_cold_us(F launch) {
    if (!g_flush) hipMalloc(&g_flush, (size_t)1 << 30);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    std::vector<float> ts;
    for (int rep = 0; rep < 7; ++rep) {
        hipMemsetAsync(g_flush, rep, (size_t)1 << 30, 0);
        hipEventRecord(a);
        launch();
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        ts.push_back(ms * 1000.f);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}
template <typename F> double time_us(F launch) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) launch();
    hipDeviceSynchronize();
    std::vector<float> ts;
    for (int rep = 0; rep < 7; ++rep) {
        hipEventRecord(a);
        for (int i = 0; i < 5; ++i) launch();
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        ts.push_back(ms / 5 * 1000.f);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}


__device__ __forceinline__ void gelu_tile8(f32x16& v) {
    f32x2 x[8], t[8], r[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        x[p][0] = v[2 * p]; x[p][1] = v[2 * p + 1];
        const f32x2 a = x[p] * 0.70710678118654752440f;
        t[p][0] = fminf(fabsf(a[0]), 4.3f);
        t[p][1] = fminf(fabsf(a[1]), 4.3f);
    }
    const float c[8] = {4.369443071e-04f, -1.460381877e-03f, -8.251648338e-04f, 2.830188636e-02f, -1.485066472e-01f, -9.184098145e-01f,
                        -1.627909326e+00f, -9.999999783e-01f};
#pragma unroll
    for (int p = 0; p < 8; ++p) r[p] = pk_fma(f32x2(-4.435285315e-05f), t[p], f32x2(c[0]));
#pragma unroll
    for (int k = 1; k < 8; ++k) {
#pragma unroll
        for (int p = 0; p < 8; ++p) r[p] = pk_fma(r[p], t[p], f32x2(c[k]));
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int p = 0; p < 8; ++p) { r[p][0] = __builtin_amdgcn_exp2f(r[p][0]); r[p][1] = __builtin_amdgcn_exp2f(r[p][1]); }
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const f32x2 up = 1.0f - r[p];
        f32x2 phi;
        phi[0] = x[p][0] < 0.f ? r[p][0] : up[0];
        phi[1] = x[p][1] < 0.f ? r[p][1] : up[1];
        const f32x2 y = x[p] * phi;
        v[2 * p] = y[0]; v[2 * p + 1] = y[1];
    }
}

// Two roles in one workgroup of 8 waves: waves 0-3 stream weights (NT tiles in flight in registers) and issue the MFMAs of one
// 4-tile unit per step with the activation operand re-read from LDS, then park the raw accumulator in LDS; waves 4-7 read the
// previous step's raw tile, run NV VALU instructions on it and write a 6 KiB operand tile back.  One workgroup barrier per step.
template <int NT, int NV, int WM, int PRIO = 0>
__global__ __launch_bounds__(512, 2) void k_roles(const f32x4* __restrict__ W, float* out, unsigned long long* cyc = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* OP = lds;                     // 4 operand tiles of 6 KiB
    float* RAW = lds + 4 * 1536;         // 2 x 4 raw tiles of 4 KiB
    constexpr int per = kTiles / 4, units = per / 4;
    if (wave < 4) {
        f32x4 buf[NT][6];
        auto ld = [&](int i, f32x4(&dst)[6]) {
            const f32x4* p = W + ((size_t)(WM ? wave * (per + 8) + i : i * 4 + wave) * 6) * 64 + lane;      // WM: one contiguous stream per wave
#pragma unroll
            for (int k = 0; k < 6; ++k) dst[k] = p[k * 64];
        };
#pragma unroll
        for (int s = 0; s < NT; ++s) ld(s, buf[s]);
        asm volatile("" ::: "memory");
        int i = 0;
        unsigned long long p_go = __builtin_readcyclecounter();
        for (int u = 0; u < units; u += NT) {          // NT units per trip so that slot indices stay compile-time
#pragma unroll
            for (int uu = 0; uu < NT; ++uu) {
                f32x16 acc;
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    constexpr int dummy = 0; (void)dummy;
                    const int s = (uu * 4 + kb) % NT;
                    f32x4 op[6];
#pragma unroll
                    for (int k = 0; k < 6; ++k) op[k] = reinterpret_cast<const f32x4*>(OP + kb * 1536 + k * 256)[lane];
#pragma unroll
                    for (int m = 0; m < 12; ++m)
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, buf[s][m % 6]), __builtin_bit_cast(bf16x8, op[m / 2]), acc, 0, 0, 0);
                    const int nx = i + NT < per ? i + NT : i;
                    ld(nx, buf[s]);
                    asm volatile("" ::: "memory");
                    ++i;
                }
                float* r = RAW + (((u + uu) & 1) * 4 + wave) * 1024;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 t; for (int j = 0; j < 4; ++j) t[j] = acc[4 * g + j];
                    reinterpret_cast<f32x4*>(r)[g * 64 + lane] = t;
                }
                __syncthreads();
                if (cyc && lane == 0 && blockIdx.x == 0 && wave == 0) { const unsigned long long now = __builtin_readcyclecounter(); cyc[256 + u + uu] = now - p_go; p_go = now; }
            }
        }
    } else {
        const int hw = wave - 4;
        float keep = 0.f;
        if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
        for (int u = 0; u < (units + NT - 1) / NT * NT; ++u) {
            const unsigned long long t_go = __builtin_readcyclecounter();
            if (u > 0) {
                const float* r = RAW + (((u - 1) & 1) * 4 + hw) * 1024;
                f32x4 t[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) t[g] = reinterpret_cast<const f32x4*>(r)[g * 64 + lane];
                float v[16];
#pragma unroll
                for (int g = 0; g < 4; ++g) for (int j = 0; j < 4; ++j) v[4 * g + j] = t[g][j];
                if (NV >= 0) {
#pragma unroll
                    for (int n = 0; n < NV; ++n) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[n & 15]) : "v"(keep));
                    f32x4 o[6];
#pragma unroll
                    for (int k = 0; k < 6; ++k) for (int j = 0; j < 4; ++j) o[k][j] = v[(4 * k + j) & 15] * 1e-30f;
#pragma unroll
                    for (int k = 0; k < 6; ++k) reinterpret_cast<f32x4*>(OP + hw * 1536 + k * 256)[lane] = o[k];
                } else {            // the real thing: GELU + exact three-way split + operand tile store (gat_roles.hip, steps 14-17)
                    f32x16 hd;
#pragma unroll
                    for (int q = 0; q < 16; ++q) hd[q] = v[q] * 1e-3f + 0.25f;
                    if (NV == -1 || NV == -2) gelu_tile8(hd);
                    if (NV == -1 || NV == -3) x3_store(OP + hw * 1536, lane, x3_split(hd));
                    else keep += hd[3];
                }
                if (cyc && lane == 0 && blockIdx.x == 0 && hw == 0) { const unsigned long long now = __builtin_readcyclecounter(); cyc[u] = now - t_go; }
            }
            __syncthreads();
        }
        if (keep == 1.f) out[0] = keep;
    }
}
template <int NT, int NV, int WM = 0, int PRIO = 0> void run_roles() {
    const size_t ldsb = (4 * 1536 + 8 * 1024) * 4;
    hipFuncSetAttribute((const void*)k_roles<NT, NV, WM, PRIO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    const double us = time_us([&] { k_roles<NT, NV, WM, PRIO><<<g_nwg, 512, ldsb>>>(g_w, g_out); });
    const double bytes = (double)kTiles * 6 * 1024;
    static unsigned long long* d_cyc = nullptr;
    if (!d_cyc) hipMalloc(&d_cyc, 512 * 8);
    hipMemset(d_cyc, 0, 512 * 8);
    k_roles<NT, NV, WM, PRIO><<<g_nwg, 512, ldsb>>>(g_w, g_out, d_cyc);
    hipDeviceSynchronize();
    std::vector<unsigned long long> hc(512);
    hipMemcpy(hc.data(), d_cyc, 512 * 8, hipMemcpyDeviceToHost);
    std::sort(hc.begin() + 10, hc.begin() + 90); std::sort(hc.begin() + 266, hc.begin() + 346);
    printf("   [in-kernel, wg 0: helper work per step %llu cycles, product step (incl. barrier) %llu cycles]\n", hc[50], hc[306]);
    const double cold = time_cold_us([&] { k_roles<NT, NV, WM, PRIO><<<g_nwg, 512, ldsb>>>(g_w, g_out); });
    printf("helper prio %d: roles 4 stream+MFMA waves (%d tiles in flight each, %s) + 4 helper waves (%3d VALU per step), barrier per 4-tile unit : %7.1f us  %6.1f GB/s/CU | weights cold: %7.1f us\n",
           PRIO, NT, WM ? "stream per wave" : "tiles interleaved", NV, us, bytes / us * 1e-3, cold);
    fflush(stdout);
}

template <int TD, int NT, int WAVES, int MF> void run_reg() {
    const double us = time_us([&] { k_reg<TD, NT, WAVES, MF><<<g_nwg, WAVES * 64>>>(g_w, g_out); });
    const double bytes = (double)kTiles * TD * 1024;
    printf("reg  tile %d KiB  waves %d  in-flight %2d tiles/wave (%3d KiB/CU)  mfma/tile %2d : %7.1f us  %6.1f GB/s/CU  %5.1f B/clk/CU @2.1GHz\n",
           TD, WAVES, NT, NT * TD * WAVES, MF, us, bytes / us * 1e-3, bytes / (us * 2100.0));
    fflush(stdout);
}
template <int TD, int R, int WAVES, int MF> void run_dma() {
    const size_t ldsb = (size_t)WAVES * R * TD * 1024;
    hipFuncSetAttribute((const void*)k_dma<TD, R, WAVES, MF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    const double us = time_us([&] { k_dma<TD, R, WAVES, MF><<<g_nwg, WAVES * 64, ldsb>>>(g_w, g_out); });
    const double bytes = (double)kTiles * TD * 1024;
    printf("dma  tile %d KiB  waves %d  ring %2d slots/wave     (%3d KiB/CU)  mfma/tile %2d : %7.1f us  %6.1f GB/s/CU  %5.1f B/clk/CU @2.1GHz\n",
           TD, WAVES, R, R * TD * WAVES, MF, us, bytes / us * 1e-3, bytes / (us * 2100.0));
    fflush(stdout);
}

int main(int argc, char** argv) {
    if (argc > 1) g_nwg = atoi(argv[1]);
    hipMalloc(&g_out, 4096 * 4);
    const size_t bytes = (size_t)kTiles * 6 * 1024;
    hipMalloc(&g_w, bytes + (2u << 20));      // (the two-role kernel rounds its unit count up: reads past the last tile)
    std::vector<unsigned short> h(bytes / 2);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (unsigned short)(i * 2654435761u >> 22);     // random-ish bf16 near 1
    hipMemcpy(g_w, h.data(), bytes, hipMemcpyHostToDevice);
    printf("workgroups: %d\n", g_nwg);
    printf("-- bf16 x3 tiles (6 KiB), no compute\n");
    run_reg<6, 2, 4, 0>(); run_reg<6, 4, 4, 0>(); run_reg<6, 6, 4, 0>(); run_reg<6, 8, 4, 0>();
    run_reg<6, 2, 8, 0>(); run_reg<6, 3, 8, 0>(); run_reg<6, 4, 8, 0>();
    run_dma<6, 3, 4, 0>(); run_dma<6, 6, 4, 0>(); run_dma<6, 2, 8, 0>(); run_dma<6, 3, 8, 0>();
    printf("-- bf16 x3 tiles (6 KiB), 12 MFMAs per tile\n");
    run_reg<6, 4, 4, 12>(); run_reg<6, 6, 4, 12>(); run_reg<6, 3, 8, 12>(); run_reg<6, 4, 8, 12>();
    run_dma<6, 6, 4, 12>(); run_dma<6, 3, 8, 12>();
    printf("-- fp32 tiles (4 KiB), no compute / 12 MFMAs per tile\n");
    run_reg<4, 4, 4, 0>(); run_reg<4, 8, 4, 0>(); run_reg<4, 4, 8, 0>(); run_reg<4, 6, 8, 0>();
    run_reg<4, 6, 4, 12>(); run_reg<4, 4, 8, 12>(); run_dma<4, 8, 4, 12>(); run_dma<4, 4, 8, 12>();
    printf("-- two roles\n");
    run_roles<3, 0>(); run_roles<3, 300>(); run_roles<5, 0>(); run_roles<5, 300>(); run_roles<5, 600>(); run_roles<6, 300>();
    run_roles<5, 0, 1>(); run_roles<5, 300, 1>();
    printf("-- helper = GELU + split + store (-1), GELU only (-2), split + store only (-3)\n");
    run_roles<5, -1, 1>(); run_roles<5, -3, 1>();
    run_roles<5, -1, 1, 1>(); run_roles<5, -1, 1, 3>(); run_roles<5, 300, 1, 3>(); run_roles<5, -3, 1, 3>();
    printf("-- MFMA only reference: 1584 x 12 MFMAs over 4 / 8 waves = %.1f us at 2.1 GHz if issue-bound\n", 1584.0 * 12 * 32 / 4 / 2100.0);
    return 0;
}
