// Does VALU work hide in the shadow of bf16 MFMAs issued by the SAME wave?  One wave per SIMD, 64 MFMAs per trip on 4 independent
// accumulators, NV VALU instructions of one kind after each MFMA (independent register chains), cycles per MFMA gap.
// Build: hipcc -O3 --offload-arch=gfx950 -I gator_amd/csrc -I include tools/microbench/coissue.hip -o tools/microbench/coissue.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int NV>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, int reps, float seed) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = seed * (i + r);
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(seed * j); b[j] = (__bf16)(seed + j); }
    float v[8]; f32x2 p[8];
    for (int i = 0; i < 8; ++i) { v[i] = seed + i + lane; p[i][0] = seed + i; p[i][1] = seed - i; }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int m = 0; m < 64; ++m) {
            acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i & 7]) : "v"(seed));
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i & 7]) : "v"(p[(i + 1) & 7]));
                if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i & 7]));
                if (KIND == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[i & 7]) : "v"(seed));
                if (KIND == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i & 7]) : "v"(seed));
                if (KIND == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i & 7]) : "v"(p[(i + 1) & 7]));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += v[i] + p[i][0] + p[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int KIND, int NV> double run(float* out, unsigned long long* cyc) {
    const int nwg = 256, reps = 50;
    k<KIND, NV><<<nwg, 256>>>(out, cyc, reps, 0.001f);
    k<KIND, NV><<<nwg, 256>>>(out, cyc, reps, 0.001f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nwg * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    return (double)h[h.size() / 2] / reps / 64;
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 4 * 8);
    printf("cycles per MFMA gap (bf16 32x32x16, 4 independent accumulators, one wave per SIMD)\n");
    printf("NV VALU per gap :      0      2      4      6      8\n");
    printf("v_fma_f32       : %6.1f %6.1f %6.1f %6.1f %6.1f\n", run<0,0>(out,cyc), run<0,2>(out,cyc), run<0,4>(out,cyc), run<0,6>(out,cyc), run<0,8>(out,cyc));
    printf("v_pk_fma_f32    : %6.1f %6.1f %6.1f %6.1f %6.1f\n", run<1,0>(out,cyc), run<1,2>(out,cyc), run<1,4>(out,cyc), run<1,6>(out,cyc), run<1,8>(out,cyc));
    printf("v_exp_f32       : %6.1f %6.1f %6.1f %6.1f %6.1f\n", run<2,0>(out,cyc), run<2,2>(out,cyc), run<2,4>(out,cyc), run<2,6>(out,cyc), run<2,8>(out,cyc));
    printf("v_cvt_pk_bf16   : %6.1f %6.1f %6.1f %6.1f %6.1f\n", run<3,0>(out,cyc), run<3,2>(out,cyc), run<3,4>(out,cyc), run<3,6>(out,cyc), run<3,8>(out,cyc));
    printf("v_cndmask_b32   : %6.1f %6.1f %6.1f %6.1f %6.1f\n", run<4,0>(out,cyc), run<4,2>(out,cyc), run<4,4>(out,cyc), run<4,6>(out,cyc), run<4,8>(out,cyc));
    printf("v_pk_mul_f32    : %6.1f %6.1f %6.1f %6.1f %6.1f\n", run<5,0>(out,cyc), run<5,2>(out,cyc), run<5,4>(out,cyc), run<5,6>(out,cyc), run<5,8>(out,cyc));
    return 0;
}
