// What does v_permlane32_swap_b32 return?  (gfx950: swaps vdst lanes 32..63 with src lanes 0..31)
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench/permlane_probe.hip -o tools/microbench/permlane_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
    const unsigned v = threadIdx.x;
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    o[threadIdx.x] = r[0];
    o[64 + threadIdx.x] = r[1];
    o[128 + threadIdx.x] = __shfl_xor(v, 32);
}
int main() {
    unsigned *d, h[192];
    hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("lane: r0 r1 shfl_xor32\n");
    for (int l : {0, 1, 31, 32, 33, 63}) printf("%2d: %2u %2u %2u\n", l, h[l], h[64 + l], h[128 + l]);
    return 0;
}
