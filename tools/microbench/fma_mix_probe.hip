// x - float(fp16 half of a packed pair) in ONE instruction (v_fma_mix_f32 reads an f16 half as a source): check against the plain form.
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/microbench/fma_mix_probe.hip -o tools/microbench/fma_mix_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float sub_lo(float x, h2 p) {
    float r;
    asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p), "v"(x));
    return r;
}
__device__ __forceinline__ float sub_hi(float x, h2 p) {
    float r;
    asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p), "v"(x));
    return r;
}
__global__ void k(const float* in, float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const float x0 = in[2 * i], x1 = in[2 * i + 1];
    h2 p;
    p[0] = (_Float16)x0; p[1] = (_Float16)x1;
    out[4 * i + 0] = sub_lo(x0, p);
    out[4 * i + 1] = sub_hi(x1, p);
    out[4 * i + 2] = x0 - (float)p[0];
    out[4 * i + 3] = x1 - (float)p[1];
}
int main() {
    const int n = 1 << 20;
    float* h = (float*)malloc(n * 4);
    srand(5);
    for (int i = 0; i < n; ++i) { unsigned u = ((unsigned)rand() << 16) ^ (unsigned)rand(); u = (u & 0x807fffffu) | ((unsigned)(100 + rand() % 40) << 23); memcpy(&h[i], &u, 4); }
    float *din, *dout, *ho = (float*)malloc(2 * n * 4);
    hipMalloc(&din, n * 4); hipMalloc(&dout, 2 * n * 4);
    hipMemcpy(din, h, n * 4, hipMemcpyHostToDevice);
    k<<<n / 2 / 256, 256>>>(din, dout, n);
    hipMemcpy(ho, dout, 2 * n * 4, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int i = 0; i < n / 2; ++i) { if (memcmp(&ho[4 * i], &ho[4 * i + 2], 4)) ++bad; if (memcmp(&ho[4 * i + 1], &ho[4 * i + 3], 4)) ++bad; }
    printf("%d values (magnitudes 2^-27 .. 2^12), mismatches between v_fma_mix_f32 and cvt + sub: %ld (%s)\n", n, bad, hipGetErrorString(hipGetLastError()));
    return 0;
}
