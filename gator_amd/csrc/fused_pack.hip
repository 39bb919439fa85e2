// Create-time packing of reference-layout weights into MFMA operand order (see fused_common.h).
#include "fused_common.h"
#include "fused_state.h"
#include "x3_common.h"

#include <cmath>
#include <cstring>

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
// H3 tile [plane][s][lane][jj] <- three fp16 planes of scale * (fp32 tile element); res: max residual (as float bits)
__global__ void k_repack_h3(const float* __restrict__ src, _Float16* __restrict__ dst, int64_t total, float scale, unsigned* __restrict__ res) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int jj = e & 7, lane = (e >> 3) & 63, s = (e >> 9) & 1;
    const int64_t tile = e >> 10;
    const float x = src[tile * kTile + ((2 * s + (jj >> 2)) * 64 + lane) * 4 + (jj & 3)] * scale;
    const _Float16 h = (_Float16)x;
    const float r = x - (float)h;
    const _Float16 m = (_Float16)r;
    const float r2 = r - (float)m;
    const _Float16 l = (_Float16)r2;
    _Float16* d = dst + tile * (2 * kTileX3) + (s * 64 + lane) * 8 + jj;
    d[0] = h;
    d[1024] = m;
    d[2048] = l;
    const float left = fabsf(r2 - (float)l);
    if (left > 0.f) atomicMax(res, __float_as_uint(left));
}
// max |w| over the tiles with (tile % period) < live (period 0: all of them)
__global__ void k_absmax_tiles(const float* __restrict__ w, int64_t n, unsigned* __restrict__ out, int64_t period, int64_t live) {
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if (period == 0 || (i / kTile) % period < live) m = fmaxf(m, fabsf(w[i]));
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}
}  // namespace

namespace {
struct DevFree {        // temporaries of the pack routines: freed on every return path
    void* p = nullptr;
    ~DevFree() { if (p) (void)hipFree(p); }
};
}  // namespace

int fused_repack_h3(const float* src_tiles, float* dst_tiles, int64_t ntiles, int* shift, float* residual, void* stream, int64_t period, int64_t live) {
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = ntiles * kTile;
    DevFree tmp;
    GATOR_HIP_CHECK(hipMalloc(&tmp.p, 2 * sizeof(unsigned)));
    unsigned* d = (unsigned*)tmp.p;
    GATOR_HIP_CHECK(hipMemsetAsync(d, 0, 2 * sizeof(unsigned), st));
    k_absmax_tiles<<<512, 256, 0, st>>>(src_tiles, total, d, period, live);
    unsigned bits[2] = {0, 0};
    GATOR_HIP_CHECK(hipMemcpyAsync(bits, d, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    GATOR_HIP_CHECK(hipStreamSynchronize(st));
    float wmax;
    memcpy(&wmax, &bits[0], sizeof(wmax));
    int sh = 0;
    if (wmax > 0.f && std::isfinite(wmax)) {
        int e;
        (void)std::frexp(wmax, &e);              // wmax 2^(14 - e) in [2^13, 2^14)
        sh = 14 - e;
        if (sh > 24) sh = 24;
        if (sh < -16) sh = -16;
    }
    k_repack_h3<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(src_tiles, (_Float16*)dst_tiles, total, std::ldexp(1.0f, sh), d + 1);
    GATOR_HIP_CHECK(hipGetLastError());
    GATOR_HIP_CHECK(hipMemcpyAsync(bits, d, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, st));
    GATOR_HIP_CHECK(hipStreamSynchronize(st));
    float left;
    memcpy(&left, &bits[1], sizeof(left));
    *shift = sh;
    *residual = wmax > 0.f ? left / std::ldexp(wmax, sh) : 0.f;
    return GATOR_OK;
}

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
