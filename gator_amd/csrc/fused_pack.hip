// Create-time packing of reference-layout weights into MFMA operand order (see fused_common.h).
#include "fused_common.h"
#include "fused_state.h"
#include "x3_common.h"

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
// X3 tile [plane][s][lane][jj] <- split3( fp32 tile [g = 2s + (jj>>2)][lane][j = jj&3] )
__global__ void k_repack_x3(const float* __restrict__ src, __bf16* __restrict__ dst, int64_t total) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per (tile, s, lane, jj)
    if (e >= total) return;
    const int jj = e & 7, lane = (e >> 3) & 63, s = (e >> 9) & 1;
    const int64_t tile = e >> 10;
    const float x = src[tile * kTile + ((2 * s + (jj >> 2)) * 64 + lane) * 4 + (jj & 3)];
    const __bf16 h = (__bf16)x;
    const float r = x - (float)h;
    const __bf16 m = (__bf16)r;
    __bf16* d = dst + tile * (2 * kTileX3) + (s * 64 + lane) * 8 + jj;
    d[0] = h;
    d[1024] = m;
    d[2048] = (__bf16)(r - (float)m);
}
}  // namespace

int fused_repack_x3(const float* src_tiles, float* dst_tiles, int64_t ntiles, void* stream) {
    const int64_t total = ntiles * kTile;
    k_repack_x3<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(src_tiles, (__bf16*)dst_tiles, total);
    GATOR_HIP_CHECK(hipGetLastError());
    return GATOR_OK;
}

int fused_pack_linear(const float* W, int64_t wsn, int64_t wsk, int N, int K, float* dst, void* stream) {
    const int NB = nblk32(N), KB = nblk32(K);
    const int64_t total = (int64_t)NB * KB * kTile;
    k_pack_linear<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(W, wsn, wsk, N, K, KB, dst, total);
    GATOR_HIP_CHECK(hipGetLastError());
    return GATOR_OK;
}
}  // namespace gator
