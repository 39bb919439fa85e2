// Create-time packing of reference-layout weights into MFMA operand order (see fused_common.h).
#include "fused_common.h"
#include "fused_state.h"

namespace gator {
namespace {
// dst[(nb*KB + kb)][g][lane][j] = W[n*wsn + k*wsk],  n = 32nb + (lane&31),  k = 32kb + 8g + 4(lane>>5) + j
__global__ void k_pack_linear(const float* __restrict__ W, int64_t wsn, int64_t wsk, int N, int K, int KB,
                              float* __restrict__ dst, int64_t total) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int j = e & 3, lane = (e >> 2) & 63, g = (e >> 8) & 3;
    const int64_t tile = e >> 10;
    const int kb = (int)(tile % KB), nb = (int)(tile / KB);
    const int n = 32 * nb + (lane & 31), k = 32 * kb + 8 * g + 4 * (lane >> 5) + j;
    dst[e] = (n < N && k < K) ? W[(int64_t)n * wsn + (int64_t)k * wsk] : 0.f;
}
}  // namespace

int fused_pack_linear(const float* W, int64_t wsn, int64_t wsk, int N, int K, float* dst, void* stream) {
    const int NB = nblk32(N), KB = nblk32(K);
    const int64_t total = (int64_t)NB * KB * kTile;
    k_pack_linear<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(W, wsn, wsk, N, K, KB, dst, total);
    GATOR_HIP_CHECK(hipGetLastError());
    return GATOR_OK;
}
}  // namespace gator
