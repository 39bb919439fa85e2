// Round 5: what does one vector instruction cost a SIMD on gfx950, by instruction class and by the number of waves that share the SIMD?
// (The MDR / GAT kernels are bound by vector issue; every "floor" in DESIGN.md is a count of VALU instructions times a price.)
// Each wave runs a loop of 256 instructions of one class on 16 independent registers (ILP 16) or on one register (dependent chain);
// prints SIMD cycles per instruction = wave cycles / instructions / waves per SIMD, from s_memtime.
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench/valu_rate.hip -o tools/microbench/valu_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int KIND, int MF>      // MF: 1 = every 8th instruction slot an independent-accumulator f16 MFMA is issued by the SAME wave (one per 8 VALU)
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int reps, float seed) {
    float v[16];
    for (int r = 0; r < 16; ++r) v[r] = seed * (r + (threadIdx.x & 63)) + 0.5f;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = seed * r;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(seed * j + (threadIdx.x & 63)); b[j] = (_Float16)(seed + j); }
    const float c = seed + 1.0f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < reps; ++it) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(c));
#define FMAD(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[0]) : "v"(c));
#define EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
#define CVTPK(i) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
#define CVTF(i) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(v[i]));
#define CVTSD(i) asm volatile("v_cvt_f32_f16_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(v[i]));
#define FMAMK(i) asm volatile("v_fmamk_f32 %0, %0, 0x3b800000, %1" : "+v"(v[i]) : "v"(c));
#define MAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(c));
#define ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
#define SUBN(i) asm volatile("v_add_f32_e64 %0, %0, -%1" : "+v"(v[i]) : "v"(c));
#define MOV(i) asm volatile("v_mov_b32 %0, %1" : "=v"(v[i]) : "v"(c));
#define PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double*)&v[(i) & 14]) : "v"(*(const double*)&v[14]));
#define FMAMIX(i) asm volatile("v_fma_mix_f32 %0, %0, %1, %0 op_sel_hi:[0,1,0]" : "+v"(v[i]) : "v"(c));
            if (KIND == 0) { REP16(FMA) }
            if (KIND == 1) { REP16(FMAD) }
            if (KIND == 2) { REP16(EXP) }
            if (KIND == 3) { REP16(CVTPK) }
            if (KIND == 4) { REP16(CVTF) }
            if (KIND == 5) { REP16(CVTSD) }
            if (KIND == 6) { REP16(FMAMK) }
            if (KIND == 7) { REP16(MAX3) }
            if (KIND == 8) { REP16(ADD) }
            if (KIND == 9) { REP16(SUBN) }
            if (KIND == 10) { REP16(MOV) }
            if (KIND == 11) { REP16(FMAMIX) }
            if (MF && (g & 1) == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));     // one MFMA per 32 VALU
            if (MF == 2 && (g & 1) == 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));   // MF 2: one per 16
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += v[r] + acc[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

static const char* kNames[] = {"v_fma_f32 (16 independent)", "v_fma_f32 (dependent chain)", "v_exp_f32", "v_cvt_pk_f16_f32", "v_cvt_f32_f16", "v_cvt_f32_f16 sdwa WORD_1",
                               "v_fmamk_f32 (literal)", "v_max3_f32", "v_add_f32", "v_add_f32_e64 (neg)", "v_mov_b32", "v_fma_mix_f32"};

template <int KIND, int MF>
static void run(float* out, unsigned long long* cyc, int n_cu) {
    printf("%-30s %s |", kNames[KIND], MF == 0 ? "no MFMA      " : (MF == 1 ? "+1 MFMA / 32 " : "+1 MFMA / 16 "));
    for (int wps : {1, 2, 3, 4, 8}) {
        const int reps = 64, nwg = n_cu * wps;
        k<KIND, MF><<<nwg, 256>>>(out, cyc, reps, 0.001f);
        k<KIND, MF><<<nwg, 256>>>(out, cyc, reps, 0.001f);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h((size_t)nwg * 4);
        (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double wave_cyc = (double)h[h.size() / 2];
        printf("  %d w/SIMD: %5.2f", wps, wave_cyc / (reps * 256.0) / wps);
    }
    printf("   SIMD cycles per VALU instruction\n");
    fflush(stdout);
}

int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int n_cu = p.multiProcessorCount;
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, (size_t)n_cu * 8 * 256 * 4); (void)hipMalloc(&cyc, (size_t)n_cu * 8 * 4 * 8);
    printf("SIMD cycles per vector instruction (wave cycles / instructions / waves per SIMD); with MFMA: the wave's own stream carries one f16 32x32x16 MFMA per 32 or 16 VALU\n");
    run<0, 0>(out, cyc, n_cu); run<0, 1>(out, cyc, n_cu); run<0, 2>(out, cyc, n_cu);
    run<1, 0>(out, cyc, n_cu);
    run<2, 0>(out, cyc, n_cu); run<2, 2>(out, cyc, n_cu);
    run<3, 0>(out, cyc, n_cu); run<3, 2>(out, cyc, n_cu);
    run<4, 0>(out, cyc, n_cu);
    run<5, 0>(out, cyc, n_cu);
    run<6, 0>(out, cyc, n_cu);
    run<7, 0>(out, cyc, n_cu);
    run<8, 0>(out, cyc, n_cu);
    run<9, 0>(out, cyc, n_cu);
    run<10, 0>(out, cyc, n_cu);
    run<11, 0>(out, cyc, n_cu); run<11, 2>(out, cyc, n_cu);
    return 0;
}
