// train_ops.hip -- the training row (include/gator_train.h): stride-aware fp32 primitives from which
// gator_amd/train/ composes the reference's training step (lib/core/base.py:122-183: forward in train mode, the losses of
// lib/core/loss.py, backward, Adam).  First correct form of the row: general kernels, exact fp32 products (the fp32-input
// MFMA for every contraction), fixed summation orders (no atomics anywhere, so a step is bit-reproducible).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>

#include "gator_train.h"
#include "internal.h"

#ifndef GATOR_PHILOX_ROUNDS
// Philox4x32 rounds of every dropout mask.  7 is the smallest count that Salmon et al. (SC'11, Random123) report as passing BigCrush
// ("Crush-resistant"); 10 is their default with a safety margin.  Measured in the fused attention kernels: at 10 rounds the integer
// multiplies of the generator (quarter rate) make the forward launch 1.7x its dropout-free time at B=1024, at 4 rounds they hide
// completely behind the MFMAs; 7 keeps the statistical guarantee at 0.7x the arithmetic.
#define GATOR_PHILOX_ROUNDS 7
#endif

namespace gator {
namespace {

constexpr int kThreads = 256;

struct Idx4 {
    int64_t n[4];
};
struct Str4 {
    int64_t s[4];
};

__device__ __forceinline__ int64_t offset4(int64_t i, const Idx4& n, const Str4& s) {
    const int64_t i3 = i % n.n[3];
    i /= n.n[3];
    const int64_t i2 = i % n.n[2];
    i /= n.n[2];
    const int64_t i1 = i % n.n[1];
    const int64_t i0 = i / n.n[1];
    return i0 * s.s[0] + i1 * s.s[1] + i2 * s.s[2] + i3 * s.s[3];
}

// 32-bit form of offset4 for tensors below 2^31 elements (every tensor of a training step): 64-bit divisions cost ~10x the 32-bit ones
__device__ __forceinline__ void coords4(uint32_t i, const Idx4& n, uint32_t (&c)[4]) {
    const uint32_t n3 = (uint32_t)n.n[3], n2 = (uint32_t)n.n[2], n1 = (uint32_t)n.n[1];
    c[3] = i % n3;
    i /= n3;
    c[2] = i % n2;
    i /= n2;
    c[1] = i % n1;
    c[0] = i / n1;
}
__device__ __forceinline__ int64_t dot4(const uint32_t (&c)[4], const Str4& s) {
    return (int64_t)c[0] * s.s[0] + (int64_t)c[1] * s.s[1] + (int64_t)c[2] * s.s[2] + (int64_t)c[3] * s.s[3];
}

__device__ __forceinline__ float binary_apply(int op, float x, float y) {
    switch (op) {
        case 0: return x + y;
        case 1: return x - y;
        case 2: return x * y;
        default: return x / y;
    }
}

// both operands and the result dense and of one shape: a plain vectorised stream
__global__ __launch_bounds__(kThreads) void k_t_binary_flat(int op, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, int64_t total) {
    const int64_t nv = total >> 2;
    for (int64_t i = blockIdx.x * (int64_t)kThreads + threadIdx.x; i < nv; i += (int64_t)gridDim.x * kThreads) {
        const float4 x = reinterpret_cast<const float4*>(a)[i], y = reinterpret_cast<const float4*>(b)[i];
        float4 r;
        r.x = binary_apply(op, x.x, y.x); r.y = binary_apply(op, x.y, y.y); r.z = binary_apply(op, x.z, y.z); r.w = binary_apply(op, x.w, y.w);
        reinterpret_cast<float4*>(o)[i] = r;
    }
    if (blockIdx.x == 0 && threadIdx.x < (total & 3)) {
        const int64_t i = (nv << 2) + threadIdx.x;
        o[i] = binary_apply(op, a[i], b[i]);
    }
}

// out = a + b (+ c) (+ d) on dense tensors of one shape, summed left to right: the gradient of a value with several consumers
__global__ __launch_bounds__(kThreads) void k_t_add_n(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                                      const float* __restrict__ d, float* __restrict__ o, int64_t total) {
    for (int64_t i = blockIdx.x * (int64_t)kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
        float v = a[i] + b[i];
        if (c) v += c[i];
        if (d) v += d[i];
        o[i] = v;
    }
}

__global__ __launch_bounds__(kThreads) void k_t_binary32(int op, const float* __restrict__ a, Str4 sa, const float* __restrict__ b, Str4 sb,
                                                         float* __restrict__ o, Str4 so, Idx4 n, uint32_t total) {
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < total; i += gridDim.x * kThreads) {
        uint32_t c[4];
        coords4(i, n, c);
        o[dot4(c, so)] = binary_apply(op, a[dot4(c, sa)], b[dot4(c, sb)]);
    }
}

__global__ __launch_bounds__(kThreads) void k_t_binary(int op, const float* __restrict__ a, Str4 sa, const float* __restrict__ b, Str4 sb,
                                                       float* __restrict__ o, Str4 so, Idx4 n, int64_t total) {
    for (int64_t i = blockIdx.x * (int64_t)kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
        const float x = a[offset4(i, n, sa)], y = b[offset4(i, n, sb)];
        float r;
        switch (op) {
            case 0: r = x + y; break;
            case 1: r = x - y; break;
            case 2: r = x * y; break;
            default: r = x / y; break;
        }
        o[offset4(i, n, so)] = r;
    }
}

__device__ __forceinline__ float unary_apply(int op, float x, float p0, float p1) {
    switch (op) {
        case 0: return p0 * x + p1;
        case 1: return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));                          // F.gelu (erf form)
        case 2: return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
        case 3: return expf(x);
        case 4: return 1.0f / sqrtf(x);
        case 5: return sqrtf(x);
        case 6: return 1.0f / x;
        case 7: return fabsf(x);
        case 8: return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f);
        case 9: return powf(p0, x);
        case 10: return x * x;
        default: return x > p0 ? 1.f : 0.f;
    }
}

__global__ __launch_bounds__(kThreads) void k_t_unary_flat(int op, const float* __restrict__ x, float* __restrict__ o, int64_t total, float p0, float p1) {
    const int64_t nv = total >> 2;
    for (int64_t i = blockIdx.x * (int64_t)kThreads + threadIdx.x; i < nv; i += (int64_t)gridDim.x * kThreads) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        float4 r;
        r.x = unary_apply(op, v.x, p0, p1); r.y = unary_apply(op, v.y, p0, p1); r.z = unary_apply(op, v.z, p0, p1); r.w = unary_apply(op, v.w, p0, p1);
        reinterpret_cast<float4*>(o)[i] = r;
    }
    if (blockIdx.x == 0 && threadIdx.x < (total & 3)) {
        const int64_t i = (nv << 2) + threadIdx.x;
        o[i] = unary_apply(op, x[i], p0, p1);
    }
}

__global__ __launch_bounds__(kThreads) void k_t_unary32(int op, const float* __restrict__ x, Str4 sx, float* __restrict__ o, Str4 so, Idx4 n,
                                                        uint32_t total, float p0, float p1) {
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < total; i += gridDim.x * kThreads) {
        uint32_t c[4];
        coords4(i, n, c);
        o[dot4(c, so)] = unary_apply(op, x[dot4(c, sx)], p0, p1);
    }
}

__global__ __launch_bounds__(kThreads) void k_t_unary(int op, const float* __restrict__ x, Str4 sx, float* __restrict__ o, Str4 so, Idx4 n,
                                                      int64_t total, float p0, float p1) {
    for (int64_t i = blockIdx.x * (int64_t)kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads)
        o[offset4(i, n, so)] = unary_apply(op, x[offset4(i, n, sx)], p0, p1);
}

// ------------------------------------------------------------------------------------------------------------ reductions
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ double block_sum(double v, double* sh) {       // all threads get the total; sh: 4 doubles of LDS
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// general form: one workgroup per (output element, slice of the reduction space); the reduction index runs over the reduced
// dims in order (last reduced dim fastest)
__global__ __launch_bounds__(kThreads) void k_t_reduce(const float* __restrict__ x, Str4 sx, Idx4 keep, Idx4 red, int64_t R, int nsplit,
                                                       double* __restrict__ part, float* __restrict__ out, int accumulate) {
    __shared__ double sh[4];
    const int64_t o = blockIdx.x;
    const int sp = blockIdx.y;
    Idx4 kn = keep;
    int64_t base;
    {
        int64_t i = o;
        const int64_t i3 = i % kn.n[3];
        i /= kn.n[3];
        const int64_t i2 = i % kn.n[2];
        i /= kn.n[2];
        const int64_t i1 = i % kn.n[1];
        const int64_t i0 = i / kn.n[1];
        base = i0 * sx.s[0] + i1 * sx.s[1] + i2 * sx.s[2] + i3 * sx.s[3];
    }
    const int64_t per = (R + nsplit - 1) / nsplit, r0 = sp * per, r1 = r0 + per < R ? r0 + per : R;
    double acc = 0.0;
    for (int64_t r = r0 + threadIdx.x; r < r1; r += kThreads) acc += (double)x[base + offset4(r, red, sx)];
    acc = block_sum(acc, sh);
    if (threadIdx.x == 0) {
        if (nsplit == 1)
            out[o] = accumulate ? out[o] + (float)acc : (float)acc;
        else
            part[o * nsplit + sp] = acc;
    }
}

// column form: x is [R, N] with row stride ld, N contiguous outputs; thread column-coalesced
__global__ __launch_bounds__(kThreads) void k_t_reduce_cols(const float* __restrict__ x, int64_t R, int64_t N, int64_t ld, int nsplit,
                                                            double* __restrict__ part, float* __restrict__ out, int accumulate) {
    __shared__ double sh[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t col = blockIdx.x * 64 + tx;
    const int sp = blockIdx.y;
    const int64_t per = (R + nsplit - 1) / nsplit, r0 = sp * per, r1 = r0 + per < R ? r0 + per : R;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;              // four independent loads in flight per thread, fixed order
    if (col < N) {
        int64_t r = r0 + ty;
        for (; r + 12 < r1; r += 16) {
            const float v0 = x[r * ld + col], v1 = x[(r + 4) * ld + col], v2 = x[(r + 8) * ld + col], v3 = x[(r + 12) * ld + col];
            a0 += (double)v0;
            a1 += (double)v1;
            a2 += (double)v2;
            a3 += (double)v3;
        }
        for (; r < r1; r += 4) a0 += (double)x[r * ld + col];
    }
    sh[ty][tx] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (ty == 0 && col < N) {
        const double t = sh[0][tx] + sh[1][tx] + sh[2][tx] + sh[3][tx];
        if (nsplit == 1)
            out[col] = accumulate ? out[col] + (float)t : (float)t;
        else
            part[col * nsplit + sp] = t;
    }
}

// one wave per output: lanes stride over the slices, fixed-order butterfly
__global__ __launch_bounds__(kThreads) void k_t_reduce_finish(const double* __restrict__ part, int nsplit, int64_t n_out, float* __restrict__ out,
                                                              int accumulate, float scale) {
    const int64_t o = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (o >= n_out) return;
    double s = 0.0;
    for (int i = lane; i < nsplit; i += 64) s += part[o * nsplit + i];
    s = wave_sum(s);
    if (lane == 0) {
        const float r = (float)(s * (double)scale);
        out[o] = accumulate ? out[o] + r : r;
    }
}

struct ReducePlan {
    Idx4 keep, red;
    Str4 s;
    int64_t n_out = 1, R = 1, ld = 0;
    int nsplit = 1;
    bool cols = false;
};

ReducePlan plan_reduce(const int64_t* shape, const int64_t* stride, const int32_t* red) {
    ReducePlan p;
    for (int d = 0; d < 4; ++d) {
        p.keep.n[d] = red[d] ? 1 : shape[d];
        p.red.n[d] = red[d] ? shape[d] : 1;
        p.s.s[d] = stride ? stride[d] : 0;
        p.n_out *= p.keep.n[d];
        p.R *= p.red.n[d];
    }
    // column form: leading dims reduced, trailing dims kept and contiguous, reduced dims collapse to one row stride
    int first_keep = 0;
    while (first_keep < 4 && (red[first_keep] || shape[first_keep] == 1)) ++first_keep;
    bool tail_keep = true;
    for (int d = first_keep; d < 4; ++d) tail_keep = tail_keep && (!red[d] || shape[d] == 1);
    if (stride && tail_keep && first_keep > 0 && first_keep < 4 && p.R > 1) {
        int64_t expect = 1;
        bool contig = true;
        for (int d = 3; d >= first_keep; --d) {
            if (shape[d] != 1 && stride[d] != expect) contig = false;
            expect *= shape[d];
        }
        int64_t ld = 0;
        bool lead_ok = true;                        // reduced dims must form one uniform row stride
        int64_t run = 1;
        for (int d = first_keep - 1; d >= 0; --d) {
            if (shape[d] == 1) continue;
            if (ld == 0) ld = stride[d];
            if (stride[d] != ld * run) lead_ok = false;
            run *= shape[d];
        }
        if (contig && lead_ok && ld > 0) {
            p.cols = true;
            p.ld = ld;
        }
    }
    const int64_t units = p.cols ? (p.n_out + 63) / 64 : p.n_out;
    const int64_t min_rows = p.cols ? 32 : 512;                  // rows (column form) / elements (general form) per slice
    int ns = 1;
    while (units * ns < 2048 && p.R / (ns * 2) >= min_rows && ns < 1024) ns *= 2;
    p.nsplit = ns;
    return p;
}

// ------------------------------------------------------------------------------------------------------------ GEMM
typedef float f32x16t __attribute__((ext_vector_type(16)));

struct GemmArgs {
    const float *A, *B, *bias;
    float* C;
    int M, N, K, nb2, ksplit;
    int64_t am, ak, bk, bn, cm, cn;
    int64_t a1, a2, b1, b2, c1, c2;
    float alpha;
    int accumulate;
    int a_kfast, b_nfast;
    float* rowsum;      // optional: out[m] = alpha * sum_k A[m][k] (a bias gradient riding on its weight-gradient GEMM)
};

// 64x64 output tile per workgroup, K in steps of 32 through LDS ([k][m] / [k][n] images: the MFMA operand fragments are
// conflict-free row reads); wave w owns the 32x32 quadrant (w>>1, w&1); v_mfma_f32_32x32x2_f32 keeps fp32 products exact.
constexpr int kGemmBK = 32;

__device__ __forceinline__ void gemm_tile(const GemmArgs& g, float (&As)[2][kGemmBK][65], float (&Bs)[2][kGemmBK][65], int tile, int batch, int zslice) {
    constexpr int BK = kGemmBK;
    const int tiles_n = (g.N + 63) / 64;
    const int tm = tile / tiles_n, tn = tile % tiles_n;
    const int bi1 = batch / g.nb2, bi2 = batch % g.nb2;
    const float* A = g.A + bi1 * g.a1 + bi2 * g.a2;
    const float* B = g.B + bi1 * g.b1 + bi2 * g.b2;
    const int m0 = tm * 64, n0 = tn * 64;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
    const int kper = ((g.K + g.ksplit - 1) / g.ksplit + BK - 1) / BK * BK;
    const int kbeg = zslice * kper, kend = kbeg + kper < g.K ? kbeg + kper : g.K;
    f32x16t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float ra[8], rb[8], rs[8];
    const bool do_rowsum = g.rowsum != nullptr && tn == 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) rs[i] = 0.f;
    // lanes run along whichever index is contiguous in memory: (k fast) k = t % 32, m = t / 32 + 8 i; (m fast) m = t % 64, k = t / 64 + 4 i.
    // The element offsets are formed once (row part, 64-bit) and advanced by a per-step scalar; only the k bound is tested per step.
    int ak_[8], bk_[8], am_[8], bn_[8];
    int64_t aoff[8], boff[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (g.a_kfast) { ak_[i] = t & 31; am_[i] = (t >> 5) + 8 * i; } else { am_[i] = t & 63; ak_[i] = (t >> 6) + 4 * i; }
        if (g.b_nfast) { bn_[i] = t & 63; bk_[i] = (t >> 6) + 4 * i; } else { bk_[i] = t & 31; bn_[i] = (t >> 5) + 8 * i; }
        const bool okm = (m0 + am_[i]) < g.M, okn = (n0 + bn_[i]) < g.N;
        aoff[i] = okm ? (int64_t)(m0 + am_[i]) * g.am + (int64_t)ak_[i] * g.ak : -1;
        boff[i] = okn ? (int64_t)(n0 + bn_[i]) * g.bn + (int64_t)bk_[i] * g.bk : -1;
    }
    auto fetch = [&](int k0) {
        const int64_t ka = (int64_t)k0 * g.ak, kb = (int64_t)k0 * g.bk;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            ra[i] = (aoff[i] >= 0 && k0 + ak_[i] < kend) ? A[aoff[i] + ka] : 0.f;
            rb[i] = (boff[i] >= 0 && k0 + bk_[i] < kend) ? B[boff[i] + kb] : 0.f;
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            As[buf][ak_[i]][am_[i]] = ra[i];
            rs[i] += ra[i];
            Bs[buf][bk_[i]][bn_[i]] = rb[i];
        }
    };
    int buf = 0;
    if (kbeg < kend) {
        fetch(kbeg);
        stash(0);
    }
    __syncthreads();
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        const bool more = k0 + BK < kend;
        if (more) fetch(k0 + BK);                       // next slice's loads fly over this slice's MFMAs
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float a = As[buf][kk + (lane >> 5)][wm * 32 + (lane & 31)];
            const float b = Bs[buf][kk + (lane >> 5)][wn * 32 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        if (more) stash(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    if (do_rowsum) {                                    // row sums of the A tile: per-thread partials -> LDS -> one fixed-order sum per row
        float* S = &As[0][0][0];
        if (g.a_kfast) {
#pragma unroll
            for (int i = 0; i < 8; ++i) S[((t >> 5) + 8 * i) * 33 + (t & 31)] = rs[i];
        } else {
            S[(t & 63) * 33 + (t >> 6)] = ((rs[0] + rs[1]) + (rs[2] + rs[3])) + ((rs[4] + rs[5]) + (rs[6] + rs[7]));
        }
        __syncthreads();
        if (t < 64 && m0 + t < g.M) {
            const int slots = g.a_kfast ? 32 : 4;
            float v = 0.f;
            for (int j = 0; j < slots; ++j) v += S[t * 33 + j];
            if (g.ksplit > 1)
                g.rowsum[(int64_t)zslice * g.M + m0 + t] = v;
            else
                g.rowsum[m0 + t] = g.alpha * v;
        }
    }
    float* C = g.C + bi1 * g.c1 + bi2 * g.c2 + (g.ksplit > 1 ? (int64_t)zslice * g.M * g.N : 0);
    const int col = n0 + wn * 32 + (lane & 31);
    if (col >= g.N) return;
    const float bias = (g.bias && g.ksplit == 1) ? g.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = m0 + wm * 32 + (i >> 2) * 8 + (lane >> 5) * 4 + (i & 3);
        if (row < g.M) {
            float* c = C + (int64_t)row * g.cm + (int64_t)col * g.cn;
            float v = g.alpha * acc[i] + bias;
            if (g.accumulate) v += *c;
            *c = v;
        }
    }
}

__global__ __launch_bounds__(kThreads) void k_t_gemm(GemmArgs g) {
    __shared__ float As[2][kGemmBK][65], Bs[2][kGemmBK][65];
    gemm_tile(g, As, Bs, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Grouped form: ONE launch runs a list of independent unbatched products (the weight gradients of a whole backward pass, which
// nothing consumes before the optimiser).  Workgroup -> (problem, tile, K slice) through the table's running workgroup count.
struct GroupedProblem {
    const float *A, *B;
    float *C, *rowsum;                 // final destinations
    int M, N, K, ksplit;
    int64_t am, ak, bk, bn, cm, cn;
    float alpha;
    int accumulate;
    int wg_begin, fin_begin;           // first workgroup of this problem in the product launch / in the finish launch
    int64_t ws_off;                    // floats into the shared split-K workspace
    int total_wgs, total_fin;          // (entry 0: launch sizes; keeps the layout of gator_gemm_problem)
    const float* bias;                 // optional [N], added to every row (forward linears run as a group)
};

__device__ __forceinline__ int find_problem(const GroupedProblem* __restrict__ tab, int n, int wg, bool fin) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((fin ? tab[mid].fin_begin : tab[mid].wg_begin) <= wg) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__global__ __launch_bounds__(kThreads) void k_t_gemm_grouped(const GroupedProblem* __restrict__ tab, int n, float* __restrict__ ws) {
    __shared__ float As[2][kGemmBK][65], Bs[2][kGemmBK][65];
    const int pi = find_problem(tab, n, blockIdx.x, false);
    const GroupedProblem p = tab[pi];
    const int local = blockIdx.x - p.wg_begin, tiles = ((p.M + 63) / 64) * ((p.N + 63) / 64);
    GemmArgs g;
    g.A = p.A; g.B = p.B; g.bias = p.bias;
    g.M = p.M; g.N = p.N; g.K = p.K; g.nb2 = 1; g.ksplit = p.ksplit;
    g.am = p.am; g.ak = p.ak; g.bk = p.bk; g.bn = p.bn;
    g.a1 = g.a2 = g.b1 = g.b2 = g.c1 = g.c2 = 0;
    if (p.ksplit > 1) {
        g.C = ws + p.ws_off; g.cm = p.N; g.cn = 1; g.alpha = 1.f; g.accumulate = 0;
        g.rowsum = p.rowsum ? ws + p.ws_off + (int64_t)p.ksplit * p.M * p.N : nullptr;
    } else {
        g.C = p.C; g.cm = p.cm; g.cn = p.cn; g.alpha = p.alpha; g.accumulate = p.accumulate;
        g.rowsum = p.rowsum;
    }
    g.a_kfast = (g.ak == 1 || g.am != 1) ? 1 : 0;
    g.b_nfast = (g.bn == 1 || g.bk != 1) ? 1 : 0;
    gemm_tile(g, As, Bs, local % tiles, 0, local / tiles);
}

// 64 consecutive outputs per workgroup; the four waves take the slices k = w, w+4, ... with four loads in flight each, then combine
__device__ __forceinline__ void splitk_finish_block(float (&sh)[4][64], int64_t block, const float* __restrict__ ws, int ksplit, int M, int N,
                                                    float* __restrict__ C, int64_t cm, int64_t cn, const float* __restrict__ bias, float alpha,
                                                    int accumulate, const float* __restrict__ ws_rowsum, float* __restrict__ rowsum) {
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t MN = (int64_t)M * N;
    const int64_t blocks_c = (MN + 63) / 64;
    const bool is_rs = block >= blocks_c;                         // trailing workgroups finish the row sums
    const int64_t i = (is_rs ? block - blocks_c : block) * 64 + tx;
    const int64_t total = is_rs ? M : MN;
    const float* src = is_rs ? ws_rowsum : ws;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (i < total) {
        int k = ty;
        for (; k + 12 < ksplit; k += 16) {
            const float v0 = src[(int64_t)k * total + i], v1 = src[(int64_t)(k + 4) * total + i], v2 = src[(int64_t)(k + 8) * total + i],
                        v3 = src[(int64_t)(k + 12) * total + i];
            a0 += v0;
            a1 += v1;
            a2 += v2;
            a3 += v3;
        }
        for (; k < ksplit; k += 4) a0 += src[(int64_t)k * total + i];
    }
    sh[ty][tx] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (ty != 0 || i >= total) return;
    const float s = (sh[0][tx] + sh[1][tx]) + (sh[2][tx] + sh[3][tx]);
    if (is_rs) {
        rowsum[i] = alpha * s;
        return;
    }
    const int m = (int)(i / N), n = (int)(i % N);
    float v = alpha * s + (bias ? bias[n] : 0.f);
    float* c = C + (int64_t)m * cm + (int64_t)n * cn;
    if (accumulate) v += *c;
    *c = v;
}

__global__ __launch_bounds__(kThreads) void k_t_splitk_finish(const float* __restrict__ ws, int ksplit, int M, int N, float* __restrict__ C,
                                                              int64_t cm, int64_t cn, const float* __restrict__ bias, float alpha,
                                                              int accumulate, const float* __restrict__ ws_rowsum, float* __restrict__ rowsum) {
    __shared__ float sh[4][64];
    splitk_finish_block(sh, blockIdx.x, ws, ksplit, M, N, C, cm, cn, bias, alpha, accumulate, ws_rowsum, rowsum);
}

__global__ __launch_bounds__(kThreads) void k_t_splitk_finish_grouped(const GroupedProblem* __restrict__ tab, int n, const float* __restrict__ ws) {
    __shared__ float sh[4][64];
    const int pi = find_problem(tab, n, blockIdx.x, true);
    const GroupedProblem p = tab[pi];
    if (p.ksplit <= 1) return;                                   // (problems without split-K own no finish workgroups; defensive)
    const float* w = ws + p.ws_off;
    splitk_finish_block(sh, blockIdx.x - p.fin_begin, w, p.ksplit, p.M, p.N, p.C, p.cm, p.cn, p.bias, p.alpha, p.accumulate,
                        p.rowsum ? w + (int64_t)p.ksplit * p.M * p.N : nullptr, p.rowsum);
}

// ------------------------------------------------------------------------------------------------------------ row kernels
// one wave per row of n contiguous floats
__global__ __launch_bounds__(kThreads) void k_t_ln_fwd(const float* __restrict__ x, int64_t rows, int n, const float* __restrict__ w,
                                                       const float* __restrict__ b, float eps, int mode, float* __restrict__ y,
                                                       float* __restrict__ mean, float* __restrict__ rinv) {
    const int64_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + row * n;
    double s = 0.0;
    for (int i = lane; i < n; i += 64) s += (double)xr[i];
    const float mu = (float)(wave_sum(s) / n);
    double q = 0.0;
    for (int i = lane; i < n; i += 64) {
        const float d = xr[i] - mu;
        q += (double)d * d;
    }
    q = wave_sum(q);
    const float ri = mode == 0 ? 1.0f / sqrtf((float)(q / n) + eps) : 1.0f / (sqrtf((float)(q / (n - 1))) + eps);
    for (int i = lane; i < n; i += 64) {
        float v = (xr[i] - mu) * ri;
        if (w) v *= w[i];
        if (b) v += b[i];
        y[row * n + i] = v;
    }
    if (lane == 0) {
        mean[row] = mu;
        rinv[row] = ri;
    }
}

__global__ __launch_bounds__(kThreads) void k_t_ln_bwd(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean,
                                                       const float* __restrict__ rinv, const float* __restrict__ w, int64_t rows, int n,
                                                       float eps, int mode, float* __restrict__ dx, float* __restrict__ dyx,
                                                       const float* __restrict__ add) {
    const int64_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float *xr = x + row * n, *dr = dy + row * n;
    const float mu = mean[row], ri = rinv[row];
    double s1 = 0.0, s2 = 0.0;                          // sum dxhat, sum dxhat * xhat
    for (int i = lane; i < n; i += 64) {
        const float xh = (xr[i] - mu) * ri, g = dr[i] * (w ? w[i] : 1.f);
        s1 += (double)g;
        s2 += (double)g * xh;
    }
    const float m1 = (float)(wave_sum(s1) / n), t2 = (float)wave_sum(s2);
    if (mode == 0) {
        const float m2 = t2 / n;
        for (int i = lane; i < n; i += 64) {
            const float xh = (xr[i] - mu) * ri, g = dr[i] * (w ? w[i] : 1.f);
            dx[row * n + i] = ri * (g - m1 - xh * m2) + (add ? add[row * n + i] : 0.f);
            if (dyx) dyx[row * n + i] = dr[i] * xh;
        }
    } else {
        // y = w xc / s + b, s = sigma + eps, sigma = sqrt(sum xc^2 / (n-1)):  d sigma = -sum(g xc) / s^2 = -ri * t2
        const float sden = 1.0f / ri, sigma = sden - eps;
        const float dsig = -ri * t2;
        const float c2 = sigma > 0.f ? dsig / ((n - 1) * sigma) : 0.f;
        for (int i = lane; i < n; i += 64) {
            const float xc = xr[i] - mu, g = dr[i] * (w ? w[i] : 1.f);
            dx[row * n + i] = ri * (g - m1) + c2 * xc + (add ? add[row * n + i] : 0.f);
            if (dyx) dyx[row * n + i] = dr[i] * xc * ri;
        }
    }
}

__global__ __launch_bounds__(kThreads) void k_t_softmax_fwd(const float* __restrict__ x, int64_t rows, int n, float* __restrict__ p) {
    const int64_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + row * n;
    float mx = -INFINITY;
    for (int i = lane; i < n; i += 64) mx = fmaxf(mx, xr[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float s = 0.f;
    for (int i = lane; i < n; i += 64) s += expf(xr[i] - mx);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float inv = 1.0f / s;
    for (int i = lane; i < n; i += 64) p[row * n + i] = expf(xr[i] - mx) * inv;
}

__global__ __launch_bounds__(kThreads) void k_t_softmax_bwd(const float* __restrict__ p, const float* __restrict__ dp, int64_t rows, int n,
                                                            float* __restrict__ dx) {
    const int64_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float *pr = p + row * n, *dr = dp + row * n;
    double s = 0.0;
    for (int i = lane; i < n; i += 64) s += (double)pr[i] * dr[i];
    const float t = (float)wave_sum(s);
    for (int i = lane; i < n; i += 64) dx[row * n + i] = pr[i] * (dr[i] - t);
}

// ------------------------------------------------------------------------------------------------------------ dropout
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t (&k)[2]) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0], n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
    c[0] = n0; c[1] = (uint32_t)p1; c[2] = n2; c[3] = (uint32_t)p0;
    k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
}

__device__ __forceinline__ bool philox_keep_raw(uint64_t seed, uint64_t offset, int64_t idx, uint32_t thresh) {
    const int64_t q = idx >> 2;
    uint32_t c[4] = {(uint32_t)q, (uint32_t)(q >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)};
    uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
    for (int r = 0; r < GATOR_PHILOX_ROUNDS; ++r) philox_round(c, k);
    return c[idx & 3] >= thresh;
}

__global__ __launch_bounds__(kThreads) void k_t_dropout(const float* __restrict__ x, int64_t n, uint32_t thresh, float scale, uint64_t seed,
                                                        uint64_t offset, const uint64_t* __restrict__ step_counter, float* __restrict__ out,
                                                        uint8_t* __restrict__ mask) {
    const int64_t q = blockIdx.x * (int64_t)kThreads + threadIdx.x;          // four elements per thread: one philox block
    if (q * 4 >= n) return;
    if (step_counter) offset += step_counter[0] << 32;
    uint32_t c[4] = {(uint32_t)q, (uint32_t)(q >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)};
    uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
    for (int r = 0; r < GATOR_PHILOX_ROUNDS; ++r) philox_round(c, k);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t i = q * 4 + j;
        if (i < n) {
            const uint8_t keep = c[j] >= thresh ? 1 : 0;
            mask[i] = keep;
            out[i] = keep ? (x ? x[i] : 1.f) * scale : 0.f;
        }
    }
}

// out = res + path[b] * dropout(act(x)): act = identity or GELU, dropout / DropPath / residual optional.  The stored byte mask carries the
// element's keep decision; `fac` [B] the per-sample DropPath factor (0 or 1/(1-p)).  One launch for what the composed form does in up to 5.
__global__ __launch_bounds__(kThreads) void k_t_drop_fused(const float* __restrict__ x, const float* __restrict__ res, int64_t n, int64_t per_sample,
                                                           int gelu, uint32_t thresh, float scale, int drop_on, uint64_t seed, uint64_t offset,
                                                           uint32_t pthresh, float pscale, int path_on, uint64_t poffset,
                                                           const uint64_t* __restrict__ step_counter, float* __restrict__ out, uint8_t* __restrict__ mask,
                                                           float* __restrict__ fac) {
    const int64_t q = blockIdx.x * (int64_t)kThreads + threadIdx.x;
    if (q * 4 >= n) return;
    const uint64_t hi = step_counter ? (step_counter[0] << 32) : 0ull;
    uint32_t c[4] = {(uint32_t)q, (uint32_t)(q >> 32), (uint32_t)(offset + hi), (uint32_t)((offset + hi) >> 32)};
    if (drop_on) {
        uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
        for (int r = 0; r < GATOR_PHILOX_ROUNDS; ++r) philox_round(c, k);
    }
    int64_t sb = -1;
    float f = 1.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t i = q * 4 + j;
        if (i >= n) break;
        float v = x[i];
        if (gelu) v = unary_apply(1, v, 0.f, 0.f);
        if (drop_on) {
            const uint8_t keep = c[j] >= thresh ? 1 : 0;
            mask[i] = keep;
            v = keep ? v * scale : 0.f;
        }
        if (path_on) {
            const int64_t b = i / per_sample;
            if (b != sb) {                              // (one generator call per thread unless its quad straddles two samples)
                sb = b;
                f = philox_keep_raw(seed, poffset + hi, b, pthresh) ? pscale : 0.f;
            }
            if (i == b * per_sample) fac[b] = f;
            v *= f;
        }
        out[i] = res ? res[i] + v : v;
    }
}

// its backward: dx = g * path[b] * mask * scale (* gelu'(x))
__global__ __launch_bounds__(kThreads) void k_t_drop_fused_bwd(const float* __restrict__ g, const float* __restrict__ x, const uint8_t* __restrict__ mask,
                                                               const float* __restrict__ fac, int64_t n, int64_t per_sample, int gelu, float scale,
                                                               float* __restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        float v = g[i];
        if (fac) v *= fac[i / per_sample];
        if (mask) v = mask[i] ? v * scale : 0.f;
        if (gelu) v *= unary_apply(2, x[i], 0.f, 0.f);
        out[i] = v;
    }
}

__global__ __launch_bounds__(kThreads) void k_t_mask_scale(const float* __restrict__ x, const uint8_t* __restrict__ mask, int64_t n, float scale,
                                                           float* __restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads)
        out[i] = mask[i] ? x[i] * scale : 0.f;
}

__global__ void k_t_step_advance(uint64_t* c) {
    if (threadIdx.x == 0 && blockIdx.x == 0) c[0] += 1;
}

__global__ __launch_bounds__(kThreads) void k_t_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                     int64_t n, float lr, float b1, float b2, float eps, float bc1, float bc2_sqrt,
                                                     const uint64_t* __restrict__ step_counter, float one_minus_b1, float one_minus_b2) {
    if (step_counter) {                                              // bias corrections of the step held on the device (graph replay)
        const double t = (double)step_counter[0];
        bc1 = (float)(1.0 - pow((double)b1, t));
        bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, t));
    }
    for (int64_t i = blockIdx.x * (int64_t)kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        const float gi = g[i];
        const float mi = m[i] + (gi - m[i]) * one_minus_b1;          // torch: exp_avg.lerp_(grad, 1 - beta1), 1 - beta formed in double
        const float vi = v[i] * b2 + one_minus_b2 * (gi * gi);       //        exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] -= (lr / bc1) * (mi / denom);
    }
}

// ------------------------------------------------------------------------------------------------------------ losses
// CoordLoss (lib/core/loss.py:10-25)
__global__ __launch_bounds__(kThreads) void k_t_coord_loss(const float* __restrict__ pred, const float* __restrict__ tgt, const float* __restrict__ valid,
                                                           Str4 sv, Idx4 n, int64_t total, float gscale, float* __restrict__ grad,
                                                           double* __restrict__ part) {
    __shared__ double sh[4];
    double acc = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
        const float v = valid ? valid[offset4(i, n, sv)] : 1.f;
        const float d = pred[i] * v - tgt[i] * v;
        acc += (double)fabsf(d);
        if (grad) grad[i] += gscale * v * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
    }
    acc = block_sum(acc, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

struct V3 {
    float x, y, z;
};
__device__ __forceinline__ V3 ld3(const float* p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ V3 sub3(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 scl3(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ V3 add3(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 cross3(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
// F.normalize(v, p=2): v / max(|v|, 1e-12)
__device__ __forceinline__ V3 normalize3(V3 v, float& len) {
    len = fmaxf(sqrtf(dot3(v, v)), 1e-12f);
    return scl3(v, 1.0f / len);
}
// d |u.n| / d v for u = v / |v|:  sign(u.n) * (n - u (u.n)) / |v|
__device__ __forceinline__ V3 dcos(V3 u, float len, V3 nrm, float c) {
    const float sg = c > 0.f ? 1.f : (c < 0.f ? -1.f : 0.f);
    return scl3(sub3(nrm, scl3(u, c)), sg / len);
}

// NormalVectorLoss (lib/core/loss.py:59-86): per (sample, face) the three |cos| terms and the gradient at the 3 corners
__global__ __launch_bounds__(kThreads) void k_t_normal_face(const float* __restrict__ pred, const float* __restrict__ tgt, const int32_t* __restrict__ faces,
                                                            int64_t B, int64_t V, int64_t F, float* __restrict__ fg, double* __restrict__ part) {
    __shared__ double sh[4];
    double acc = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)kThreads + threadIdx.x; i < B * F; i += (int64_t)gridDim.x * kThreads) {
        const int64_t b = i / F, f = i % F;
        const int i0 = faces[f * 3], i1 = faces[f * 3 + 1], i2 = faces[f * 3 + 2];
        const float *P = pred + b * V * 3, *T = tgt + b * V * 3;
        const V3 p0 = ld3(P + i0 * 3), p1 = ld3(P + i1 * 3), p2 = ld3(P + i2 * 3);
        const V3 t0 = ld3(T + i0 * 3), t1 = ld3(T + i1 * 3), t2 = ld3(T + i2 * 3);
        float l1, l2, l3, lt;
        const V3 u1 = normalize3(sub3(p1, p0), l1), u2 = normalize3(sub3(p2, p0), l2), u3 = normalize3(sub3(p2, p1), l3);
        const V3 g1 = normalize3(sub3(t1, t0), lt), g2 = normalize3(sub3(t2, t0), lt);
        const V3 nrm = normalize3(cross3(g1, g2), lt);
        const float c1 = dot3(u1, nrm), c2 = dot3(u2, nrm), c3 = dot3(u3, nrm);
        acc += (double)fabsf(c1) + (double)fabsf(c2) + (double)fabsf(c3);
        if (fg) {
            const V3 d1 = dcos(u1, l1, nrm, c1), d2 = dcos(u2, l2, nrm, c2), d3 = dcos(u3, l3, nrm, c3);
            const V3 q0 = scl3(add3(d1, d2), -1.f), q1 = sub3(d1, d3), q2 = add3(d2, d3);      // v1 = p1-p0, v2 = p2-p0, v3 = p2-p1
            float* o = fg + i * 9;
            o[0] = q0.x; o[1] = q0.y; o[2] = q0.z; o[3] = q1.x; o[4] = q1.y; o[5] = q1.z; o[6] = q2.x; o[7] = q2.y; o[8] = q2.z;
        }
    }
    acc = block_sum(acc, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

// EdgeLengthLoss (lib/core/loss.py:89-112)
__global__ __launch_bounds__(kThreads) void k_t_edge_face(const float* __restrict__ pred, const float* __restrict__ tgt, const int32_t* __restrict__ faces,
                                                          int64_t B, int64_t V, int64_t F, float* __restrict__ fg, double* __restrict__ part) {
    __shared__ double sh[4];
    double acc = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)kThreads + threadIdx.x; i < B * F; i += (int64_t)gridDim.x * kThreads) {
        const int64_t b = i / F, f = i % F;
        const int i0 = faces[f * 3], i1 = faces[f * 3 + 1], i2 = faces[f * 3 + 2];
        const float *P = pred + b * V * 3, *T = tgt + b * V * 3;
        const V3 p0 = ld3(P + i0 * 3), p1 = ld3(P + i1 * 3), p2 = ld3(P + i2 * 3);
        const V3 t0 = ld3(T + i0 * 3), t1 = ld3(T + i1 * 3), t2 = ld3(T + i2 * 3);
        const V3 e1 = sub3(p0, p1), e2 = sub3(p0, p2), e3 = sub3(p1, p2);
        const float d1 = sqrtf(dot3(e1, e1)), d2 = sqrtf(dot3(e2, e2)), d3 = sqrtf(dot3(e3, e3));
        const V3 h1 = sub3(t0, t1), h2 = sub3(t0, t2), h3 = sub3(t1, t2);
        const float r1 = d1 - sqrtf(dot3(h1, h1)), r2 = d2 - sqrtf(dot3(h2, h2)), r3 = d3 - sqrtf(dot3(h3, h3));
        acc += (double)fabsf(r1) + (double)fabsf(r2) + (double)fabsf(r3);
        if (fg) {
            auto sg = [](float r) { return r > 0.f ? 1.f : (r < 0.f ? -1.f : 0.f); };
            const V3 a1 = scl3(e1, d1 > 0.f ? sg(r1) / d1 : 0.f), a2 = scl3(e2, d2 > 0.f ? sg(r2) / d2 : 0.f),
                     a3 = scl3(e3, d3 > 0.f ? sg(r3) / d3 : 0.f);
            const V3 q0 = add3(a1, a2), q1 = sub3(a3, a1), q2 = scl3(add3(a2, a3), -1.f);
            float* o = fg + i * 9;
            o[0] = q0.x; o[1] = q0.y; o[2] = q0.z; o[3] = q1.x; o[4] = q1.y; o[5] = q1.z; o[6] = q2.x; o[7] = q2.y; o[8] = q2.z;
        }
    }
    acc = block_sum(acc, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

// per (sample, vertex): sum the vertex's corner gradients in list order
__global__ __launch_bounds__(kThreads) void k_t_face_gather(const float* __restrict__ fg, const int32_t* __restrict__ inc_ptr, const int32_t* __restrict__ inc_idx,
                                                            int64_t B, int64_t V, int64_t F, float gscale, float* __restrict__ grad) {
    for (int64_t i = blockIdx.x * (int64_t)kThreads + threadIdx.x; i < B * V; i += (int64_t)gridDim.x * kThreads) {
        const int64_t b = i / V, v = i % V;
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (int e = inc_ptr[v]; e < inc_ptr[v + 1]; ++e) {
            const float* o = fg + (b * F) * 9 + (int64_t)inc_idx[e] * 3;
            sx += o[0];
            sy += o[1];
            sz += o[2];
        }
        grad[i * 3] += gscale * sx;
        grad[i * 3 + 1] += gscale * sy;
        grad[i * 3 + 2] += gscale * sz;
    }
}

__global__ __launch_bounds__(kThreads) void k_t_loss_finish(const double* __restrict__ part, int n, double scale, float* __restrict__ out) {
    __shared__ double sh[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += kThreads) s += part[i];
    s = block_sum(s, sh);
    if (threadIdx.x == 0) out[0] = (float)(s * scale);
}

constexpr int kLossBlocks = 2048;

int grid_for(int64_t total) {
    int64_t b = (total + kThreads - 1) / kThreads;
    return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, "%s: %s", what, hipGetErrorString(e));
    return 0;
}

int face_loss(bool normal, const float* pred, const float* target, const int32_t* faces, const int32_t* inc_ptr, const int32_t* inc_idx,
              int64_t B, int64_t V, int64_t F, float weight, float* loss_out, float* grad, void* ws, hipStream_t st) {
    if (!pred || !target || !faces || !loss_out || !ws) return fail(1, "face loss: null argument");
    if (grad && (!inc_ptr || !inc_idx)) return fail(1, "face loss: the gradient needs the vertex incidence lists");
    double* part = static_cast<double*>(ws);
    float* fg = grad ? reinterpret_cast<float*>(static_cast<char*>(ws) + 65536) : nullptr;
    const int blocks = (int)std::min<int64_t>(kLossBlocks, (B * F + kThreads - 1) / kThreads);
    if (normal)
        hipLaunchKernelGGL(k_t_normal_face, dim3(blocks), dim3(kThreads), 0, st, pred, target, faces, B, V, F, fg, part);
    else
        hipLaunchKernelGGL(k_t_edge_face, dim3(blocks), dim3(kThreads), 0, st, pred, target, faces, B, V, F, fg, part);
    const double inv = 1.0 / ((double)B * 3.0 * (double)F);
    hipLaunchKernelGGL(k_t_loss_finish, dim3(1), dim3(kThreads), 0, st, part, blocks, (double)weight * inv, loss_out);
    if (grad)
        hipLaunchKernelGGL(k_t_face_gather, dim3(grid_for(B * V)), dim3(kThreads), 0, st, fg, inc_ptr, inc_idx, B, V, F, (float)(weight * inv), grad);
    return check_launch(normal ? "gator_t_normal_loss" : "gator_t_edge_loss");
}

// keep decision of element `idx` of a tensor, exactly as k_t_dropout draws it (quad idx/4, lane idx%4 of one Philox block)
__device__ __forceinline__ bool philox_keep(unsigned long long seed, unsigned long long offset, int64_t idx, uint32_t thresh) {
    const int64_t q = idx >> 2;
    uint32_t c[4] = {(uint32_t)q, (uint32_t)(q >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)};
    uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
    for (int r = 0; r < GATOR_PHILOX_ROUNDS; ++r) philox_round(c, k);
    return c[idx & 3] >= thresh;
}

#include "train_attn.inc"

}  // namespace
}  // namespace gator

using namespace gator;

extern "C" {

int gator_t_binary(int op, const float* a, const int64_t* sa, const float* b, const int64_t* sb, float* out, const int64_t* so,
                   const int64_t* shape, gator_stream stream) {
    if (!a || !b || !out || op < 0 || op > 3) return fail(1, "gator_t_binary: bad argument");
    Idx4 n;
    Str4 A, Bs, O;
    int64_t total = 1;
    for (int d = 0; d < 4; ++d) {
        n.n[d] = shape[d];
        A.s[d] = sa[d];
        Bs.s[d] = sb[d];
        O.s[d] = so[d];
        total *= shape[d];
    }
    if (total == 0) return 0;
    auto dense = [&](const Str4& s) {
        int64_t e = 1;
        for (int d = 3; d >= 0; --d) {
            if (n.n[d] != 1 && s.s[d] != e) return false;
            e *= n.n[d];
        }
        return true;
    };
    const bool aligned = (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) == 0;
    if (dense(A) && dense(Bs) && dense(O) && aligned)
        hipLaunchKernelGGL(k_t_binary_flat, dim3(grid_for((total + 3) / 4)), dim3(kThreads), 0, (hipStream_t)stream, op, a, b, out, total);
    else if (total < 0x7fffffffLL)
        hipLaunchKernelGGL(k_t_binary32, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream, op, a, A, b, Bs, out, O, n, (uint32_t)total);
    else
        hipLaunchKernelGGL(k_t_binary, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream, op, a, A, b, Bs, out, O, n, total);
    return check_launch("gator_t_binary");
}

int gator_t_add_n(const float* a, const float* b, const float* c, const float* d, float* out, int64_t n, gator_stream stream) {
    if (!a || !b || !out || (d && !c)) return fail(1, "gator_t_add_n: bad argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_t_add_n, dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream, a, b, c, d, out, n);
    return check_launch("gator_t_add_n");
}

int gator_t_unary(int op, const float* x, const int64_t* sx, float* out, const int64_t* so, const int64_t* shape, float p0, float p1,
                  gator_stream stream) {
    if (!x || !out || op < 0 || op > 11) return fail(1, "gator_t_unary: bad argument");
    Idx4 n;
    Str4 X, O;
    int64_t total = 1;
    for (int d = 0; d < 4; ++d) {
        n.n[d] = shape[d];
        X.s[d] = sx[d];
        O.s[d] = so[d];
        total *= shape[d];
    }
    if (total == 0) return 0;
    auto dense = [&](const Str4& s) {
        int64_t e = 1;
        for (int d = 3; d >= 0; --d) {
            if (n.n[d] != 1 && s.s[d] != e) return false;
            e *= n.n[d];
        }
        return true;
    };
    if (dense(X) && dense(O) && (((uintptr_t)x | (uintptr_t)out) & 15) == 0)
        hipLaunchKernelGGL(k_t_unary_flat, dim3(grid_for((total + 3) / 4)), dim3(kThreads), 0, (hipStream_t)stream, op, x, out, total, p0, p1);
    else if (total < 0x7fffffffLL)
        hipLaunchKernelGGL(k_t_unary32, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream, op, x, X, out, O, n, (uint32_t)total, p0, p1);
    else
        hipLaunchKernelGGL(k_t_unary, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream, op, x, X, out, O, n, total, p0, p1);
    return check_launch("gator_t_unary");
}

int64_t gator_t_reduce_ws_bytes(const int64_t* shape, const int32_t* red) {
    int64_t dummy[4] = {0, 0, 0, 0};
    ReducePlan p = plan_reduce(shape, dummy, red);      // without strides the general form is assumed: an upper bound on nsplit
    int ns = 1;
    while (ns < 1024 && p.R / (ns * 2) >= 32) ns *= 2;
    return p.n_out * ns * (int64_t)sizeof(double);
}

int gator_t_reduce_sum(const float* x, const int64_t* sx, const int64_t* shape, const int32_t* red, float* out, int accumulate, void* ws,
                       gator_stream stream) {
    if (!x || !out || !ws) return fail(1, "gator_t_reduce_sum: null argument");
    ReducePlan p = plan_reduce(shape, sx, red);
    if (p.n_out == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    double* part = static_cast<double*>(ws);
    if (p.cols)
        hipLaunchKernelGGL(k_t_reduce_cols, dim3((unsigned)((p.n_out + 63) / 64), p.nsplit), dim3(kThreads), 0, st, x, p.R, p.n_out, p.ld, p.nsplit, part, out,
                           accumulate);
    else {
        if (p.n_out > 0x7fffffffLL) return fail(1, "gator_t_reduce_sum: too many outputs");
        hipLaunchKernelGGL(k_t_reduce, dim3((unsigned)p.n_out, p.nsplit), dim3(kThreads), 0, st, x, p.s, p.keep, p.red, p.R, p.nsplit, part, out, accumulate);
    }
    if (p.nsplit > 1)
        hipLaunchKernelGGL(k_t_reduce_finish, dim3((unsigned)((p.n_out + 3) / 4)), dim3(kThreads), 0, st, part, p.nsplit, p.n_out, out,
                       accumulate, 1.0f);
    return check_launch("gator_t_reduce_sum");
}

int gator_t_gemm(const float* A, const float* B, float* C, int M, int N, int K, const int64_t* sa, const int64_t* sb, const int64_t* sc, int nb1,
                 int nb2, const int64_t* ba, const int64_t* bb, const int64_t* bc, const float* bias, float alpha, int accumulate, int ksplit,
                 float* ws, float* a_rowsum, gator_stream stream) {
    if (a_rowsum && (nb1 != 1 || nb2 != 1)) return fail(1, "gator_t_gemm: a_rowsum needs an unbatched product");
    if (!A || !B || !C || M <= 0 || N <= 0 || K < 0 || nb1 <= 0 || nb2 <= 0) return fail(1, "gator_t_gemm: bad argument");
    if (ksplit < 1) ksplit = 1;
    if (ksplit > 1 && (nb1 != 1 || nb2 != 1 || !ws)) return fail(1, "gator_t_gemm: split-K needs an unbatched product and a workspace");
    if ((int64_t)nb1 * nb2 > 65535) return fail(1, "gator_t_gemm: more than 65535 batched products");
    GemmArgs g;
    g.A = A; g.B = B; g.bias = bias; g.C = ksplit > 1 ? ws : C;
    g.M = M; g.N = N; g.K = K; g.nb2 = nb2; g.ksplit = ksplit;
    g.am = sa[0]; g.ak = sa[1]; g.bk = sb[0]; g.bn = sb[1];
    g.cm = ksplit > 1 ? N : sc[0]; g.cn = ksplit > 1 ? 1 : sc[1];
    g.a1 = ba[0]; g.a2 = ba[1]; g.b1 = bb[0]; g.b2 = bb[1]; g.c1 = bc[0]; g.c2 = bc[1];
    g.alpha = ksplit > 1 ? 1.f : alpha;
    g.accumulate = ksplit > 1 ? 0 : accumulate;
    g.rowsum = a_rowsum ? (ksplit > 1 ? ws + (int64_t)ksplit * M * N : a_rowsum) : nullptr;
    g.a_kfast = (g.ak == 1 || g.am != 1) ? 1 : 0;       // lanes run along whichever index is contiguous in memory
    g.b_nfast = (g.bn == 1 || g.bk != 1) ? 1 : 0;
    const unsigned tiles = (unsigned)(((M + 63) / 64) * ((N + 63) / 64));
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_t_gemm, dim3(tiles, nb1 * nb2, ksplit), dim3(kThreads), 0, st, g);
    if (ksplit > 1)
        hipLaunchKernelGGL(k_t_splitk_finish, dim3((unsigned)(((int64_t)M * N + 63) / 64 + (a_rowsum ? (M + 63) / 64 : 0))), dim3(kThreads), 0, st, ws, ksplit, M, N, C,
                           sc[0], sc[1], bias, alpha, accumulate, a_rowsum ? ws + (int64_t)ksplit * M * N : nullptr, a_rowsum);
    return check_launch("gator_t_gemm");
}

int64_t gator_t_gemm_grouped_prepare(gator_gemm_problem* pr, int n) {
    if (!pr || n <= 0) return -1;
    int64_t ws = 0;
    int wg = 0, fin = 0;
    for (int i = 0; i < n; ++i) {
        gator_gemm_problem& p = pr[i];
        if (p.M <= 0 || p.N <= 0 || p.K < 0 || p.ksplit < 1) return -1;
        const int tiles = ((p.M + 63) / 64) * ((p.N + 63) / 64);
        p.wg_begin = wg;
        p.fin_begin = fin;
        p.ws_off = ws;
        wg += tiles * p.ksplit;
        if (p.ksplit > 1) {
            fin += (int)(((int64_t)p.M * p.N + 63) / 64) + (p.a_rowsum ? (p.M + 63) / 64 : 0);
            ws += (int64_t)p.ksplit * ((int64_t)p.M * p.N + p.M);
        }
    }
    pr[0].total_wgs = wg;
    pr[0].total_fin = fin;
    return ws;
}

int gator_t_gemm_grouped(const gator_gemm_problem* table_host, int n, void* table_dev, float* ws, gator_stream stream) {
    if (!table_host || !table_dev || n <= 0) return fail(1, "gator_t_gemm_grouped: bad argument");
    static_assert(sizeof(gator_gemm_problem) == sizeof(GroupedProblem) && offsetof(gator_gemm_problem, ws_off) == offsetof(GroupedProblem, ws_off), "host / device problem layouts");
    hipStream_t st = (hipStream_t)stream;
    const int wgs = table_host[0].total_wgs, fin = table_host[0].total_fin;
    if (wgs <= 0) return fail(1, "gator_t_gemm_grouped: call gator_t_gemm_grouped_prepare first");
    if (fin > 0 && !ws) return fail(1, "gator_t_gemm_grouped: split-K problems need the workspace");
    const hipError_t e = hipMemcpyAsync(table_dev, table_host, (size_t)n * sizeof(gator_gemm_problem), hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return fail((int)e, "gator_t_gemm_grouped: table upload: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(k_t_gemm_grouped, dim3(wgs), dim3(kThreads), 0, st, static_cast<const GroupedProblem*>(table_dev), n, ws);
    if (fin > 0)
        hipLaunchKernelGGL(k_t_splitk_finish_grouped, dim3(fin), dim3(kThreads), 0, st, static_cast<const GroupedProblem*>(table_dev), n, ws);
    return check_launch("gator_t_gemm_grouped");
}

int gator_t_layernorm_fwd(const float* x, int64_t rows, int n, const float* w, const float* b, float eps, int mode, float* y, float* mean,
                          float* rinv, gator_stream stream) {
    if (!x || !y || !mean || !rinv || n < 1 || (mode == 1 && n < 2)) return fail(1, "gator_t_layernorm_fwd: bad argument");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(k_t_ln_fwd, dim3((unsigned)((rows + 3) / 4)), dim3(kThreads), 0, (hipStream_t)stream, x, rows, n, w, b, eps, mode, y, mean, rinv);
    return check_launch("gator_t_layernorm_fwd");
}

int gator_t_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rinv, const float* w, int64_t rows, int n, float eps,
                          int mode, float* dx, float* dy_xhat, const float* add, gator_stream stream) {
    if (!dy || !x || !mean || !rinv || !dx) return fail(1, "gator_t_layernorm_bwd: null argument");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(k_t_ln_bwd, dim3((unsigned)((rows + 3) / 4)), dim3(kThreads), 0, (hipStream_t)stream, dy, x, mean, rinv, w, rows, n, eps, mode, dx,
                       dy_xhat, add);
    return check_launch("gator_t_layernorm_bwd");
}

int gator_t_softmax_fwd(const float* x, int64_t rows, int n, float* p, gator_stream stream) {
    if (!x || !p || n < 1) return fail(1, "gator_t_softmax_fwd: bad argument");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(k_t_softmax_fwd, dim3((unsigned)((rows + 3) / 4)), dim3(kThreads), 0, (hipStream_t)stream, x, rows, n, p);
    return check_launch("gator_t_softmax_fwd");
}

int gator_t_softmax_bwd(const float* p, const float* dp, int64_t rows, int n, float* dx, gator_stream stream) {
    if (!p || !dp || !dx || n < 1) return fail(1, "gator_t_softmax_bwd: bad argument");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(k_t_softmax_bwd, dim3((unsigned)((rows + 3) / 4)), dim3(kThreads), 0, (hipStream_t)stream, p, dp, rows, n, dx);
    return check_launch("gator_t_softmax_bwd");
}

int gator_t_step_advance(uint64_t* step_counter, gator_stream stream) {
    if (!step_counter) return fail(1, "gator_t_step_advance: null counter");
    hipLaunchKernelGGL(k_t_step_advance, dim3(1), dim3(64), 0, (hipStream_t)stream, step_counter);
    return check_launch("gator_t_step_advance");
}

int gator_t_dropout(const float* x, int64_t n, float rate, uint64_t seed, uint64_t offset, const uint64_t* step_counter, float* out, uint8_t* mask,
                    gator_stream stream) {
    if (!out || !mask || rate < 0.f || rate >= 1.f) return fail(1, "gator_t_dropout: bad argument");
    if (n == 0) return 0;
    const double t = (double)rate * 4294967296.0;
    const uint32_t thresh = t >= 4294967295.0 ? 0xffffffffu : (uint32_t)t;
    const int64_t quads = (n + 3) / 4;
    hipLaunchKernelGGL(k_t_dropout, dim3((unsigned)((quads + kThreads - 1) / kThreads)), dim3(kThreads), 0, (hipStream_t)stream, x, n, thresh,
                       1.0f / (1.0f - rate), seed, offset, step_counter, out, mask);
    return check_launch("gator_t_dropout");
}

static uint32_t rate_thresh(float rate) {
    const double t = (double)rate * 4294967296.0;
    return t >= 4294967295.0 ? 0xffffffffu : (uint32_t)t;
}

int gator_t_drop_fused(const float* x, const float* res, int64_t n, int64_t per_sample, int gelu, float rate, uint64_t seed, uint64_t offset,
                       float path_rate, uint64_t path_offset, const uint64_t* step_counter, float* out, uint8_t* mask, float* path_factor,
                       gator_stream stream) {
    const bool drop_on = rate > 0.f && offset != 0, path_on = path_rate > 0.f && path_offset != 0;
    if (!x || !out || n <= 0 || per_sample <= 0 || rate < 0.f || rate >= 1.f || path_rate < 0.f || path_rate >= 1.f || (drop_on && !mask) ||
        (path_on && !path_factor))
        return fail(1, "gator_t_drop_fused: bad argument");
    const int64_t quads = (n + 3) / 4;
    hipLaunchKernelGGL(k_t_drop_fused, dim3((unsigned)((quads + kThreads - 1) / kThreads)), dim3(kThreads), 0, (hipStream_t)stream, x, res, n, per_sample, gelu,
                       rate_thresh(rate), 1.0f / (1.0f - rate), drop_on ? 1 : 0, seed, offset, rate_thresh(path_rate), 1.0f / (1.0f - path_rate),
                       path_on ? 1 : 0, path_offset, step_counter, out, mask, path_factor);
    return check_launch("gator_t_drop_fused");
}

int gator_t_drop_fused_bwd(const float* g, const float* x, const uint8_t* mask, const float* path_factor, int64_t n, int64_t per_sample, int gelu,
                           float rate, float* out, gator_stream stream) {
    if (!g || !out || n <= 0 || per_sample <= 0 || (gelu && !x)) return fail(1, "gator_t_drop_fused_bwd: bad argument");
    hipLaunchKernelGGL(k_t_drop_fused_bwd, dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream, g, x, mask, path_factor, n, per_sample, gelu,
                       1.0f / (1.0f - rate), out);
    return check_launch("gator_t_drop_fused_bwd");
}

int gator_t_mask_scale(const float* x, const uint8_t* mask, int64_t n, float scale, float* out, gator_stream stream) {
    if (!x || !mask || !out) return fail(1, "gator_t_mask_scale: null argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_t_mask_scale, dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream, x, mask, n, scale, out);
    return check_launch("gator_t_mask_scale");
}

int gator_t_adam(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr, double beta1, double beta2, double eps,
                 int step, const uint64_t* step_counter, gator_stream stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || (step < 1 && !step_counter)) return fail(1, "gator_t_adam: bad argument");
    if (step < 1) step = 1;
    if (n == 0) return 0;
    const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);      // scalars in double, as torch forms them on the host
    hipLaunchKernelGGL(k_t_adam, dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, (float)lr, (float)beta1,
                       (float)beta2, (float)eps, (float)bc1, (float)sqrt(bc2), step_counter, (float)(1.0 - beta1), (float)(1.0 - beta2));
    return check_launch("gator_t_adam");
}

static int attn_args(AttnArgs& a, const float* q, const float* k, const float* v, float* o, float* lse, int B, int H, int T, int Tk, int D, float scale,
                     float rate, uint64_t seed, uint64_t offset, const uint64_t* counter, const char* what) {
    if (!q || !k || !v || !o || !lse || B <= 0 || H <= 0 || T <= 0 || Tk <= 0) return fail(1, "%s: bad argument", what);
    if (D != kAD) return fail(1, "%s: head dim %d (built for %d)", what, D, kAD);
    if (rate < 0.f || rate >= 1.f || (int64_t)B * H > 65535) return fail(1, "%s: bad rate / too many heads", what);
    a.q = q; a.k = k; a.v = v; a.o = o; a.lse = lse;
    a.B = B; a.H = H; a.T = T; a.Tk = Tk; a.scale = scale;
    a.seed = seed; a.off = offset; a.counter = reinterpret_cast<const unsigned long long*>(counter); a.rate = rate;
    a.d_o = nullptr; a.dq = a.dk = a.dv = a.dsum = nullptr; a.qsplit = 1;
    return 0;
}

int gator_t_attn_fwd(const float* q, const float* k, const float* v, float* o, float* lse, int B, int H, int T, int Tk, int D, float scale, float rate,
                     uint64_t seed, uint64_t offset, const uint64_t* counter, gator_stream stream) {
    AttnArgs a;
    if (attn_args(a, q, k, v, o, lse, B, H, T, Tk, D, scale, rate, seed, offset, counter, "gator_t_attn_fwd")) return 1;
    hipLaunchKernelGGL(k_t_attn_fwd, dim3((T + 127) / 128, B * H), dim3(kThreads), 0, (hipStream_t)stream, a);
    return check_launch("gator_t_attn_fwd");
}

int gator_t_attn_bwd(const float* q, const float* k, const float* v, const float* o, const float* lse, const float* d_o, float* dq, float* dk,
                     float* dv, float* dsum, int B, int H, int T, int Tk, int D, float scale, float rate, uint64_t seed, uint64_t offset,
                     const uint64_t* counter, int qsplit, gator_stream stream) {
    AttnArgs a;
    if (attn_args(a, q, k, v, const_cast<float*>(o), const_cast<float*>(lse), B, H, T, Tk, D, scale, rate, seed, offset, counter, "gator_t_attn_bwd")) return 1;
    if (!d_o || !dq || !dk || !dv || !dsum) return fail(1, "gator_t_attn_bwd: null argument");
    a.d_o = d_o; a.dq = dq; a.dk = dk; a.dv = dv; a.dsum = dsum;
    a.qsplit = qsplit < 1 ? 1 : qsplit;
    hipStream_t st = (hipStream_t)stream;
    const int64_t rows = (int64_t)B * H * T;
    hipLaunchKernelGGL(k_t_attn_rowdot, dim3((unsigned)((rows + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, a);
    hipLaunchKernelGGL(k_t_attn_bwd_dq, dim3((T + 127) / 128, B * H), dim3(kThreads), 0, st, a);
    hipLaunchKernelGGL(k_t_attn_bwd_dkv, dim3((Tk + 127) / 128, B * H, a.qsplit), dim3(kThreads), 0, st, a);
    return check_launch("gator_t_attn_bwd");
}

static int attn_small_args(AttnSmallArgs& a, const float* qkv, const float* bias, float* o, float* P, int B, int H, int J, int D, float scale, float rate,
                           uint64_t seed, uint64_t offset, const uint64_t* counter, const char* what) {
    if (!qkv || !bias || !o || !P || B <= 0 || H <= 0 || J < 1 || J > 32) return fail(1, "%s: bad argument (J <= 32)", what);
    if (D != kSD || rate < 0.f || rate >= 1.f) return fail(1, "%s: head dim %d (built for %d) / bad rate", what, D, kSD);
    a.qkv = qkv; a.bias = bias; a.o = o; a.P = P; a.B = B; a.H = H; a.J = J; a.scale = scale;
    a.seed = seed; a.off = offset; a.counter = reinterpret_cast<const unsigned long long*>(counter); a.rate = rate;
    a.d_o = nullptr; a.dqkv = a.dS = nullptr;
    return 0;
}

int gator_t_attn_small_fwd(const float* qkv, const float* bias, float* o, float* P, int B, int H, int J, int D, float scale, float rate, uint64_t seed,
                           uint64_t offset, const uint64_t* counter, gator_stream stream) {
    AttnSmallArgs a;
    if (attn_small_args(a, qkv, bias, o, P, B, H, J, D, scale, rate, seed, offset, counter, "gator_t_attn_small_fwd")) return 1;
    hipLaunchKernelGGL(k_t_attn_small_fwd, dim3((B * H + 3) / 4), dim3(kThreads), 0, (hipStream_t)stream, a);
    return check_launch("gator_t_attn_small_fwd");
}

int gator_t_attn_small_bwd(const float* qkv, const float* bias, const float* P, const float* d_o, float* dqkv, float* dS, int B, int H, int J, int D,
                           float scale, float rate, uint64_t seed, uint64_t offset, const uint64_t* counter, gator_stream stream) {
    AttnSmallArgs a;
    if (attn_small_args(a, qkv, bias, dqkv, const_cast<float*>(P), B, H, J, D, scale, rate, seed, offset, counter, "gator_t_attn_small_bwd")) return 1;
    if (!d_o || !dqkv || !dS) return fail(1, "gator_t_attn_small_bwd: null argument");
    a.o = nullptr; a.d_o = d_o; a.dqkv = dqkv; a.dS = dS;
    hipLaunchKernelGGL(k_t_attn_small_bwd, dim3((B * H + 3) / 4), dim3(kThreads), 0, (hipStream_t)stream, a);
    return check_launch("gator_t_attn_small_bwd");
}

int gator_t_mgcn_fwd(const float* h0, const float* h1, const float* adj, const float* M, const float* bias, float* out, int B, int J, int C,
                     gator_stream stream) {
    if (!h0 || !h1 || !adj || !M || !bias || !out || B <= 0 || J <= 0 || C <= 0) return fail(1, "gator_t_mgcn_fwd: bad argument");
    MgcnArgs a;
    a.h0 = h0; a.h1 = h1; a.adj = adj; a.M = M; a.bias = bias; a.out = out; a.B = B; a.J = J; a.C = C;
    a.d_out = nullptr; a.dh0 = a.dh1 = a.pm = a.dadj = nullptr;
    hipLaunchKernelGGL(k_t_mgcn_fwd, dim3(B, (J * C + kThreads - 1) / kThreads), dim3(kThreads), 0, (hipStream_t)stream, a);
    return check_launch("gator_t_mgcn_fwd");
}

int gator_t_mgcn_bwd(const float* h0, const float* h1, const float* adj, const float* M, const float* d_out, float* dh0, float* dh1, float* pm,
                     float* dadj, int B, int J, int C, gator_stream stream) {
    if (!h0 || !h1 || !adj || !M || !d_out || !dh0 || !dh1 || !pm || !dadj || B <= 0 || J <= 0 || C <= 0) return fail(1, "gator_t_mgcn_bwd: bad argument");
    MgcnArgs a;
    a.h0 = h0; a.h1 = h1; a.adj = adj; a.M = M; a.bias = nullptr; a.out = nullptr; a.B = B; a.J = J; a.C = C;
    a.d_out = d_out; a.dh0 = dh0; a.dh1 = dh1; a.pm = pm; a.dadj = dadj;
    hipLaunchKernelGGL(k_t_mgcn_bwd, dim3(B, (J * C + kThreads - 1) / kThreads), dim3(kThreads), 0, (hipStream_t)stream, a);
    return check_launch("gator_t_mgcn_bwd");
}

int gator_t_batchnorm_fwd(const float* x, const float* w, const float* b, float* y, float* mean, float* rinv, float* run_mean, float* run_var, int B,
                          int C, int L, float eps, float momentum, gator_stream stream) {
    if (!x || !w || !b || !y || !mean || !rinv || B <= 0 || C <= 0 || L <= 0 || (run_mean && !run_var)) return fail(1, "gator_t_batchnorm_fwd: bad argument");
    BnArgs a;
    a.x = x; a.w = w; a.b = b; a.y = y; a.mean = mean; a.rinv = rinv; a.run_mean = run_mean; a.run_var = run_var; a.B = B; a.C = C; a.L = L;
    a.eps = eps; a.momentum = momentum; a.dy = nullptr; a.dx = a.dw = a.db = nullptr;
    hipLaunchKernelGGL(k_t_bn_fwd, dim3(C), dim3(kThreads), 0, (hipStream_t)stream, a);
    return check_launch("gator_t_batchnorm_fwd");
}

int gator_t_batchnorm_bwd(const float* dy, const float* x, const float* w, const float* mean, const float* rinv, float* dx, float* dw, float* db, int B,
                          int C, int L, gator_stream stream) {
    if (!dy || !x || !w || !mean || !rinv || !dx || !dw || !db || B <= 0 || C <= 0 || L <= 0) return fail(1, "gator_t_batchnorm_bwd: bad argument");
    BnArgs a;
    a.x = x; a.w = w; a.b = nullptr; a.y = nullptr; a.mean = const_cast<float*>(mean); a.rinv = const_cast<float*>(rinv); a.run_mean = a.run_var = nullptr;
    a.B = B; a.C = C; a.L = L; a.eps = 0.f; a.momentum = 0.f; a.dy = dy; a.dx = dx; a.dw = dw; a.db = db;
    hipLaunchKernelGGL(k_t_bn_bwd, dim3(C), dim3(kThreads), 0, (hipStream_t)stream, a);
    return check_launch("gator_t_batchnorm_bwd");
}

int64_t gator_t_struct_size(int which) {
    return which == 0 ? (int64_t)sizeof(gator_gemm_problem) : -1;
}

int64_t gator_t_loss_ws_bytes(int64_t B, int64_t F) { return 65536 + B * F * 9 * (int64_t)sizeof(float); }

int gator_t_coord_loss(const float* pred, const float* target, const float* valid, const int64_t* sv, const int64_t* shape, float weight,
                       float* loss_out, float* grad, void* ws, gator_stream stream) {
    if (!pred || !target || !loss_out || !ws) return fail(1, "gator_t_coord_loss: null argument");
    Idx4 n;
    Str4 S;
    int64_t total = 1;
    for (int d = 0; d < 4; ++d) {
        n.n[d] = shape[d];
        S.s[d] = (valid && sv) ? sv[d] : 0;
        total *= shape[d];
    }
    if (total == 0) return fail(1, "gator_t_coord_loss: empty input");
    const int blocks = (int)std::min<int64_t>(kLossBlocks, (total + kThreads - 1) / kThreads);
    hipStream_t st = (hipStream_t)stream;
    double* part = static_cast<double*>(ws);
    hipLaunchKernelGGL(k_t_coord_loss, dim3(blocks), dim3(kThreads), 0, st, pred, target, valid, S, n, total, (float)((double)weight / (double)total), grad,
                       part);
    hipLaunchKernelGGL(k_t_loss_finish, dim3(1), dim3(kThreads), 0, st, part, blocks, (double)weight / (double)total, loss_out);
    return check_launch("gator_t_coord_loss");
}

int gator_t_normal_loss(const float* pred, const float* target, const int32_t* faces, const int32_t* inc_ptr, const int32_t* inc_idx, int64_t B,
                        int64_t V, int64_t F, float weight, float* loss_out, float* grad, void* ws, gator_stream stream) {
    return face_loss(true, pred, target, faces, inc_ptr, inc_idx, B, V, F, weight, loss_out, grad, ws, (hipStream_t)stream);
}

int gator_t_edge_loss(const float* pred, const float* target, const int32_t* faces, const int32_t* inc_ptr, const int32_t* inc_idx, int64_t B,
                      int64_t V, int64_t F, float weight, float* loss_out, float* grad, void* ws, gator_stream stream) {
    return face_loss(false, pred, target, faces, inc_ptr, inc_idx, B, V, F, weight, loss_out, grad, ws, (hipStream_t)stream);
}

}  // extern "C"
