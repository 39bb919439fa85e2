// Does a DEPENDENT chain of 32x32x16 MFMAs (same accumulator) issue every 32 cycles like independent ones?  NA accumulators used
// round-robin, 1 or 2 waves per SIMD, f16 and bf16 operands; cycles per MFMA as one wave sees them.
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench/mfma_chain.hip -o tools/microbench/mfma_chain.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NA, int WAVES, int BF, int USE>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void k(float* out, unsigned long long* cyc, int reps, float seed) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = seed * (i + r);
    f16x8 a, b; bf16x8 ab, bb;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(seed * j + lane); b[j] = (_Float16)(seed + j); ab[j] = (__bf16)(seed * j + lane); bb[j] = (__bf16)(seed + j); }
    float sink = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int m = 0; m < 24; ++m) {
            if (BF) acc[m % NA] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[m % NA], 0, 0, 0);
            else acc[m % NA] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m % NA], 0, 0, 0);
            if (USE && (m % 6) == 5) {               // a VALU consumer of the chain's result every 6 MFMAs (as the softmax is)
                float s = acc[m % NA][0];
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(sink) : "v"(s));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = sink;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * WAVES * 64 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * WAVES + wave] = (t1 - t0) * 10 / (reps * 24);
}
static float* g_out; static unsigned long long* g_cyc;
template <int NA, int WAVES, int BF, int USE> void run() {
    const int nwg = 256, reps = 200;
    k<NA, WAVES, BF, USE><<<nwg, WAVES * 64>>>(g_out, g_cyc, reps, 0.001f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nwg * WAVES);
    hipMemcpy(h.data(), g_cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%s  accumulators %d  waves/SIMD %d  consumer every 6: %d : %5.1f cycles per MFMA per wave (%5.1f per SIMD)\n", BF ? "bf16" : "f16 ", NA, WAVES / 4, USE,
           h[h.size() / 2] / 10.0, h[h.size() / 2] / 10.0 / (WAVES / 4));
}
int main() {
    hipMalloc(&g_out, 256 * 512 * 4); hipMalloc(&g_cyc, 256 * 8 * 8);
    run<1, 4, 0, 0>(); run<2, 4, 0, 0>(); run<4, 4, 0, 0>(); run<1, 8, 0, 0>(); run<2, 8, 0, 0>(); run<4, 8, 0, 0>();
    run<1, 4, 1, 0>(); run<4, 4, 1, 0>(); run<1, 8, 1, 0>();
    run<1, 4, 0, 1>(); run<2, 4, 0, 1>(); run<1, 8, 0, 1>(); run<2, 8, 0, 1>(); run<4, 8, 0, 1>();
    return 0;
}
