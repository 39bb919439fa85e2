// Which VALU instruction classes co-execute with the SIMD partner's MFMAs?  512-thread workgroups; waves 0-3 issue dependent bf16
// MFMAs back to back, waves 4-7 (s_setprio PRIO) loop over 256 independent instructions of ONE class.  Read with
//   rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES -- ./coexec_classes.bin
// (one dispatch per class, in the order printed); the binary itself prints the helper's cycles per instruction.
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench/coexec_classes.hip -o tools/microbench/coexec_classes.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int PRIO>
__global__ __launch_bounds__(512, 2) void k(float* out, unsigned long long* cyc, int reps, float seed) {
    __shared__ __attribute__((aligned(16))) float lds[4 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave < 4) {
        f32x16 acc;
        for (int r = 0; r < 16; ++r) acc[r] = seed * r;
        bf16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(seed * j + lane); b[j] = (__bf16)(seed + j); }
        for (int r = 0; r < reps * 3; ++r) {
#pragma unroll
            for (int m = 0; m < 16; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
        float s = 0.f;
        for (int r = 0; r < 16; ++r) s += acc[r];
        out[blockIdx.x * 512 + threadIdx.x] = s;
        return;
    }
    if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
    float v[16]; f32x2 p[8]; unsigned u[16];
    for (int i = 0; i < 16; ++i) { v[i] = seed + i + lane; u[i] = i * 77 + lane; }
    for (int i = 0; i < 8; ++i) { p[i][0] = seed + i; p[i][1] = seed - i; }
    const unsigned ldsa = (unsigned)(unsigned long long)(lds + (wave - 4) * 1024 + lane * 4);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int i = 0; i < 256; ++i) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i & 15]) : "v"(seed));
            if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i & 7]) : "v"(p[(i + 1) & 7]));
            if (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i & 7]) : "v"(p[(i + 1) & 7]));
            if (KIND == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i & 15]));
            if (KIND == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[i & 15]) : "v"(seed));
            if (KIND == 5) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i & 15]) : "v"(u[(i + 3) & 15]));
            if (KIND == 6) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[i & 15]) : "v"(seed));
            if (KIND == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i & 15]) : "v"(seed));
            if (KIND == 8) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[i & 15]) : "v"(seed), "v"(v[(i + 5) & 15]));
            if (KIND == 9) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(u[i & 15]));
            if (KIND == 10) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[i & 15]) : "v"(seed));
            if (KIND == 11) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(v[i & 15]));
            if (KIND == 12) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i & 15]) : "v"(seed));
            if (KIND == 13) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i & 15]) : "v"(u[(i + 3) & 15]), "v"(u[(i + 7) & 15]));
            if (KIND == 14 && (i & 7) == 0) asm volatile("ds_write_b128 %0, %1" :: "v"(ldsa), "v"(*(reinterpret_cast<float __attribute__((ext_vector_type(4)))*>(&v[0]))) : "memory");
            if (KIND == 15 && (i & 7) == 0) asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(*(reinterpret_cast<float __attribute__((ext_vector_type(4)))*>(&v[0]))) : "v"(ldsa) : "memory");
            if (KIND == 16) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i & 15]) : "v"(u[(i + 3) & 15]));
            if (KIND == 17) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i & 15]) : "v"(seed));
            if (KIND == 18) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[i & 15]) : "v"(seed), "v"(v[(i + 5) & 15]));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += v[i] + (float)u[i];
    for (int i = 0; i < 8; ++i) s += p[i][0] + p[i][1];
    out[blockIdx.x * 512 + threadIdx.x] = s + lds[lane];
    if (lane == 0) cyc[blockIdx.x * 4 + wave - 4] = (t1 - t0) * 100 / (reps * 256);
}
static float* g_out; static unsigned long long* g_cyc;
template <int KIND, int PRIO> void run(const char* name) {
    const int nwg = 256, reps = 40;
    k<KIND, PRIO><<<nwg, 512>>>(g_out, g_cyc, reps, 0.001f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nwg * 4);
    hipMemcpy(h.data(), g_cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-22s prio %d : %6.2f helper cycles per instruction\n", name, PRIO, h[h.size() / 2] / 100.0);
}
#define BOTH(K, NAME) run<K, 0>(NAME); run<K, 1>(NAME);
int main() {
    hipMalloc(&g_out, 256 * 512 * 4); hipMalloc(&g_cyc, 256 * 4 * 8);
    BOTH(0, "v_fma_f32") BOTH(1, "v_pk_fma_f32") BOTH(2, "v_pk_add_f32") BOTH(3, "v_exp_f32") BOTH(4, "v_cvt_pk_bf16_f32") BOTH(5, "v_and_b32")
    BOTH(6, "v_sub_f32") BOTH(7, "v_cndmask_b32") BOTH(8, "v_max3_f32") BOTH(9, "v_lshlrev_b32") BOTH(10, "v_cvt_pk_f16_f32") BOTH(11, "v_cvt_f32_f16")
    BOTH(12, "v_mul_f32") BOTH(13, "v_perm_b32") BOTH(14, "ds_write_b128 (1/8)") BOTH(15, "ds_read_b128 (1/8)") BOTH(16, "v_mov_b32") BOTH(17, "v_add_f32") BOTH(18, "v_fmac_f32")
    return 0;
}
