// How fast does a helper wave's VALU work (GELU + exact 3-way bf16 split of one register tile, gat_roles.hip) run when its SIMD
// partner issues bf16 MFMAs back to back?  512-thread workgroups: waves 0-3 loop over dependent MFMAs (MF = 1) or idle (MF = 0),
// waves 4-7 loop over KIND: 0 = gelu_tile8 + x3_split, 1 = x3_split only, 2 = gelu only, 3 = plain fma chain of the same count.
// Prints helper cycles per iteration.
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I gator_amd/csrc -I include tools/microbench/helper_valu.hip -o tools/microbench/helper_valu.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include "x3_common.h"
using namespace gator;

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

template <int KIND, int MF, int PRIO>
__global__ __launch_bounds__(512, 2) void k(float* out, unsigned long long* cyc, int reps, float seed) {
    __shared__ __attribute__((aligned(16))) float lds[8 * 1536];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave < 4) {
        f32x16 acc;
        for (int r = 0; r < 16; ++r) acc[r] = seed * r;
        bf16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(seed * j + lane); b[j] = (__bf16)(seed + j); }
        const unsigned long long m0 = __builtin_readcyclecounter();
        if (MF)
            for (int r = 0; r < reps * 2; ++r) {
#pragma unroll
                for (int m = 0; m < 16; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
            }
        if (lane == 0) cyc[1024 + blockIdx.x * 4 + wave] = (__builtin_readcyclecounter() - m0) / (reps * 2 * 16);
        float s = 0.f;
        for (int r = 0; r < 16; ++r) s += acc[r];
        out[blockIdx.x * 512 + threadIdx.x] = s;
        return;
    }
    f32x16 v;
    for (int r = 0; r < 16; ++r) v[r] = seed * (r + lane) - 0.3f;
    if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        if (KIND == 0 || KIND == 2) gelu_tile8(v);
        if (KIND == 0 || KIND == 1) {
            const X3 sp = x3_split(v);
            x3_store(lds + (wave - 4) * 1536, lane, sp);
            for (int q = 0; q < 16; ++q) v[q] += 0.125f;
        }
        if (KIND == 3) {
#pragma unroll
            for (int i = 0; i < 280; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i & 15]) : "v"(seed));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += v[r];
    out[blockIdx.x * 512 + threadIdx.x] = s + lds[lane];
    if (lane == 0) cyc[blockIdx.x * 4 + wave - 4] = (t1 - t0) / reps;
}
static double g_mfma = 0;
template <int KIND, int MF, int PRIO = 0> double run(float* out, unsigned long long* cyc) {
    const int nwg = 256, reps = 200;
    k<KIND, MF, PRIO><<<nwg, 512>>>(out, cyc, reps, 0.001f);
    k<KIND, MF, PRIO><<<nwg, 512>>>(out, cyc, reps, 0.001f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(2048);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.begin() + 1024);
    std::sort(h.begin() + 1024, h.end());
    g_mfma = (double)h[1024 + 512];
    return (double)h[512];
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 2048 * 8);
    printf("helper cycles per iteration (the MFMA partner runs 32 MFMAs per helper iteration = 1024 cycles if undisturbed) | partner cycles per MFMA\n");
    double a, b, c, d;
    a = run<0, 0>(out, cyc); printf("gelu + split + store, partner idle        : %8.0f\n", a);
    b = run<0, 1>(out, cyc); printf("gelu + split + store, partner MFMA prio 0 : %8.0f | %5.1f\n", b, g_mfma);
    c = run<0, 1, 1>(out, cyc); printf("gelu + split + store, partner MFMA prio 1 : %8.0f | %5.1f\n", c, g_mfma);
    d = run<0, 1, 3>(out, cyc); printf("gelu + split + store, partner MFMA prio 3 : %8.0f | %5.1f\n", d, g_mfma);
    a = run<3, 0>(out, cyc); printf("280 v_fma_f32, partner idle               : %8.0f\n", a);
    b = run<3, 1>(out, cyc); printf("280 v_fma_f32, partner MFMA prio 0        : %8.0f | %5.1f\n", b, g_mfma);
    c = run<3, 1, 3>(out, cyc); printf("280 v_fma_f32, partner MFMA prio 3        : %8.0f | %5.1f\n", c, g_mfma);
    c = run<2, 1, 3>(out, cyc); printf("gelu only, partner MFMA prio 3            : %8.0f | %5.1f\n", c, g_mfma);
    c = run<1, 1, 3>(out, cyc); printf("split + store only, partner MFMA prio 3   : %8.0f | %5.1f\n", c, g_mfma);
    return 0;
}
