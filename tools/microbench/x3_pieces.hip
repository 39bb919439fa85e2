// Microbenchmark of the pieces of one MLP chunk of k_mdr_layer (one wave per SIMD, operands in registers):
//   a) 24 dependent bf16 MFMAs (fc1 chain)   b) gelu_tile   c) x3_split   d) 24 MFMAs on two accumulators (fc2)
//   e) the whole chunk a+b+c+d chained as in the kernel.   Cycles per repetition by s_memtime, median over waves.
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I gator_amd/csrc -I include tools/microbench/x3_pieces.hip -o /tmp/x3_pieces
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include "x3_common.h"
using namespace gator;

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const float* __restrict__ in, float* __restrict__ out, unsigned long long* cyc, int reps) {
    const int lane = threadIdx.x & 63;
    f32x16 h = load_block(in, lane), acc0 = zero16(), acc1 = zero16();
    X3 w0 = x3_split(load_block(in + 1024, lane)), w1 = x3_split(load_block(in + 2048, lane));
    X3 y0 = x3_split(load_block(in + 3072, lane)), y1 = x3_split(load_block(in + 4096, lane));
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        if (MODE == 0 || MODE == 4) {       // fc1: 24 dependent MFMAs
            f32x16 a = h * 1e-3f;
            a = x3_mma(w0, y0, a);
            a = x3_mma(w1, y1, a);
            h = a;
        }
        if (MODE == 1 || MODE == 4) gelu_tile(h);
        X3 hx;
        if (MODE == 2 || MODE == 4) {
            hx = x3_split(h);
            if (MODE == 2) { h[0] += (float)hx.p[2][0][0] + (float)hx.p[1][1][3] + (float)hx.p[0][0][1]; }
        }
        if (MODE == 3) hx = y0;
        if (MODE == 3 || MODE == 4) {
            acc0 = x3_mma(w0, hx, acc0);
            acc1 = x3_mma(w1, hx, acc1);
            if (MODE == 3) { y0.p[0][0][0] = (__bf16)acc0[0]; }
        }
        asm volatile("" ::: "memory");
    }
    f32x16 res = h + acc0 + acc1;
    store_block(out + (size_t)(blockIdx.x * 4 + (threadIdx.x >> 6)) * 1024, lane, res);
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// MODE 5: the whole chunk with its weights streamed as in k_mdr_layer: fc1 pair (2 tiles) and fc2 pair (2 tiles) of the NEXT
// chunk requested right after the MFMAs that used the current ones are queued; 16+16 X3 tiles per "layer", 8 chunks.
struct W2X { X3 t[2]; };
__device__ __forceinline__ W2X ldw2x(const float* __restrict__ Wx, int i0, int i1, int lane) {
    W2X w;
    w.t[0] = x3_load(Wx + (size_t)i0 * kTileX3, lane);
    w.t[1] = x3_load(Wx + (size_t)i1 * kTileX3, lane);
    return w;
}
#define PIN() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
template <int DEPTH>
__global__ __launch_bounds__(256, 2) void kw(const float* __restrict__ in, const float* __restrict__ wts, float* __restrict__ out,
                                             unsigned long long* cyc, int reps, int stagger) {
    const int lane = threadIdx.x & 63;
    f32x16 h = load_block(in, lane), acc0 = zero16(), acc1 = zero16();
    X3 y[2] = {x3_split(load_block(in + 3072, lane)), x3_split(load_block(in + 4096, lane))};
    const float* fc1 = wts;
    const float* fc2 = wts + 16 * kTileX3;
    W2X A = ldw2x(fc1, 0, 1, lane), B = ldw2x(fc2, 0, 8, lane);
    __builtin_amdgcn_s_waitcnt(0);
    // stagger: 1 = second half of the grid, 2 = odd workgroups: delayed by about half a chunk (VALU-only work)
    const bool late = stagger == 1 ? blockIdx.x >= gridDim.x / 2 : (stagger == 2 ? (blockIdx.x & 1) : false);
    if (late) { f32x16 d = h; gelu_tile(d); gelu_tile(d); h += d * 1e-9f; }
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int r = 0; r < reps; ++r) {
#pragma unroll 1
        for (int c = 0; c < 8; ++c) {
            const int cn = (c + 1) & 7;
            f32x16 a = zero16();
            a = x3_mma(A.t[0], y[0], a);
            a = x3_mma(A.t[1], y[1], a);
            A = ldw2x(fc1, 2 * cn, 2 * cn + 1, lane);
            PIN();
            a += h * 1e-3f;
            gelu_tile(a);
            const X3 hx = x3_split(a);
            acc0 = x3_mma(B.t[0], hx, acc0);
            acc1 = x3_mma(B.t[1], hx, acc1);
            B = ldw2x(fc2, cn, 8 + cn, lane);
            PIN();
            h = a;
        }
    }
    f32x16 res = h + acc0 + acc1;
    store_block(out + (size_t)(blockIdx.x * 4 + (threadIdx.x >> 6)) * 1024, lane, res);
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
// MODE 6: the same chunk loop software-pipelined inside ONE wave: fc1 of chunk c+1 and fc2 of chunk c-1 (48 independent MFMAs)
// are placed beside GELU + split of chunk c (VALU), so the in-order issue stream can alternate them.
template <int WPS, int HINT>
__global__ __launch_bounds__(256, WPS) void kp(const float* __restrict__ in, const float* __restrict__ wts, float* __restrict__ out,
                                               unsigned long long* cyc, int reps) {
    const int lane = threadIdx.x & 63;
    f32x16 h = load_block(in, lane), acc0 = zero16(), acc1 = zero16();
    X3 y[2] = {x3_split(load_block(in + 3072, lane)), x3_split(load_block(in + 4096, lane))};
    const float* fc1 = wts;
    const float* fc2 = wts + 16 * kTileX3;
    W2X A = ldw2x(fc1, 0, 1, lane), B = ldw2x(fc2, 0, 8, lane);
    f32x16 cur = x3_mma(A.t[1], y[1], x3_mma(A.t[0], y[0], zero16()));      // fc1(0)
    A = ldw2x(fc1, 2, 3, lane);
    X3 hxp = x3_split(h);                                                    // stands for split(gelu(fc1(-1)))
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int r = 0; r < reps; ++r) {
#pragma unroll 1
        for (int c = 0; c < 8; ++c) {
            const int cn = (c + 2) & 7, cb = (c + 1) & 7;
            // MFMA stream: fc1(c+1) -> nxt, fc2(c-1) with hxp
            f32x16 nxt = zero16();
            nxt = x3_mma(A.t[0], y[0], nxt);
            nxt = x3_mma(A.t[1], y[1], nxt);
            acc0 = x3_mma(B.t[0], hxp, acc0);
            acc1 = x3_mma(B.t[1], hxp, acc1);
            // VALU stream: gelu + split of chunk c
            f32x16 g = cur + h * 1e-3f;
            gelu_tile(g);
            const X3 hx = x3_split(g);
            if (HINT) {
#pragma unroll
                for (int i = 0; i < 48; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);     // 7 VALU
                }
            }
            A = ldw2x(fc1, 2 * cn, 2 * cn + 1, lane);
            B = ldw2x(fc2, cb, 8 + cb, lane);
            asm volatile("" ::: "memory");
            hxp = hx;
            cur = nxt;
            h = g;
        }
    }
    f32x16 res = h + acc0 + acc1 + cur;
    store_block(out + (size_t)(blockIdx.x * 4 + (threadIdx.x >> 6)) * 1024, lane, res);
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int WPS, int HINT>
double runp(const float* in, const float* wts, float* out, unsigned long long* cyc, int nwg, int reps, size_t lds) {
    kp<WPS, HINT><<<nwg, 256, lds>>>(in, wts, out, cyc, reps);
    kp<WPS, HINT><<<nwg, 256, lds>>>(in, wts, out, cyc, reps);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nwg * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    return (double)h[h.size() / 2] / reps / 8;
}

double runw(const float* in, const float* wts, float* out, unsigned long long* cyc, int nwg, int reps, size_t lds, int stagger = 0) {
    kw<1><<<nwg, 256, lds>>>(in, wts, out, cyc, reps, stagger);
    kw<1><<<nwg, 256, lds>>>(in, wts, out, cyc, reps, stagger);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nwg * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    return (double)h[h.size() / 2] / reps / 8;
}

template <int MODE> double run(const float* in, float* out, unsigned long long* cyc, int nwg, int reps) {
    k<MODE><<<nwg, 256>>>(in, out, cyc, reps);
    k<MODE><<<nwg, 256>>>(in, out, cyc, reps);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nwg * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    return (double)h[h.size() / 2] / reps;
}

int main() {
    const int nwg = 256, reps = 200;
    float *in, *out; unsigned long long* cyc;
    hipMalloc(&in, 5 * 1024 * 4); hipMalloc(&out, (size_t)2 * nwg * 4 * 1024 * 4); hipMalloc(&cyc, 2 * nwg * 4 * 8);
    std::vector<float> hin(5 * 1024);
    for (size_t i = 0; i < hin.size(); ++i) hin[i] = 0.37f * ((int)(i * 2654435761u % 2001) - 1000) / 1000.f;
    hipMemcpy(in, hin.data(), hin.size() * 4, hipMemcpyHostToDevice);
    printf("fc1 chain (24 dependent MFMA): %.0f cycles\n", run<0>(in, out, cyc, nwg, reps));
    printf("gelu_tile                    : %.0f cycles\n", run<1>(in, out, cyc, nwg, reps));
    printf("x3_split                     : %.0f cycles\n", run<2>(in, out, cyc, nwg, reps));
    printf("fc2 (24 MFMA, 2 accumulators): %.0f cycles\n", run<3>(in, out, cyc, nwg, reps));
    printf("whole chunk                  : %.0f cycles\n", run<4>(in, out, cyc, nwg, reps));
    float* wts;
    hipMalloc(&wts, 32 * kTileX3 * 4);
    hipMemset(wts, 0, 32 * kTileX3 * 4);
    printf("chunk + weight stream, 1 wave/SIMD : %.0f cycles\n", runw(in, wts, out, cyc, 256, 25, 60 * 1024));
    printf("chunk + weight stream, 2 waves/SIMD: %.0f cycles per wave-chunk\n", runw(in, wts, out, cyc, 512, 25, 0));
    printf("  ... second half of the grid delayed: %.0f\n", runw(in, wts, out, cyc, 512, 25, 0, 1));
    printf("  ... odd workgroups delayed         : %.0f\n", runw(in, wts, out, cyc, 512, 25, 0, 2));
    printf("software-pipelined, 1 wave/SIMD, compiler schedule : %.0f cycles per chunk\n", runp<1, 0>(in, wts, out, cyc, 256, 25, 0));
    printf("software-pipelined, 1 wave/SIMD, 1 MFMA : 7 VALU   : %.0f cycles per chunk\n", runp<1, 1>(in, wts, out, cyc, 256, 25, 0));
    printf("software-pipelined, 2 waves/SIMD, compiler schedule: %.0f cycles per wave-chunk\n", runp<2, 0>(in, wts, out, cyc, 512, 25, 0));
    printf("software-pipelined, 2 waves/SIMD, 1 MFMA : 7 VALU  : %.0f cycles per wave-chunk\n", runp<2, 1>(in, wts, out, cyc, 512, 25, 0));
    return 0;
}
