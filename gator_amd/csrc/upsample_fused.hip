// Vertex regressor: upsample_conv (Conv1d 431->6890, k=3, p=1 over the 3-long xyz axis) + bias + template add,
// lib/models/MDR.py:122,167-168.   out[b][o][l] = bias[o] + tpl[o][l] + sum_c sum_k w[o][c][k] * vc[b][c][l+k-1]
//
// Formulated as three accumulators (l = 0,1,2) per (32 samples x 32 vertices) wave tile that SHARE the weight operands:
// tap k of the weights meets input column l' = l+k-1, so each loaded weight fragment w[.][c][k] feeds up to three fp32
// MFMAs (7 per (c-step) instead of 9: the zero-padding taps are never multiplied -> 41.6 of the dense 53.45 MFLOP/mesh) and
// the 35.6 MB weight is streamed once per 128 samples.  A = packed vert431 (rows = samples), B = packed weights
// (cols = output vertices)  ->  accumulator: vertex on the lane, samples in registers, so each lane stores the 3
// contiguous floats out[b][o][0..2] and a wave row covers 384 contiguous bytes.
#include "fused_common.h"
#include "fused_state.h"

namespace gator {
namespace {

// vcp[mt][l'][cb][g][lane][j] = vc[32mt + (lane&31)][32cb + 8g + 4(lane>>5) + j][l']   (zero padded)
__global__ void k_pack_vc(const float* __restrict__ vc, int B, float* __restrict__ vcp, int64_t total) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int j = e & 3, lane = (e >> 2) & 63, g = (e >> 8) & 3;
    int64_t r = e >> 10;
    const int cb = r % kCB; r /= kCB;
    const int lp = r % 3;
    const int mt = (int)(r / 3);
    const int s = 32 * mt + (lane & 31), c = 32 * cb + 8 * g + 4 * (lane >> 5) + j;
    vcp[e] = (s < B && c < kV) ? vc[((int64_t)s * kV + c) * 3 + lp] : 0.f;
}

struct __attribute__((packed)) F3 { float x, y, z; };

// NT = 32-sample tiles per wave that share each weight fragment.  NT = 2 (64 samples x 32 vertices x 3 coords) halves the
// weight traffic and, at B=256, gives 864 waves - at most one per SIMD, so no co-resident partner competes for the MFMA pipe
// (142 -> 107 us); NT = 1 keeps small batches spread over the chip.
template <int NT>
__global__ __launch_bounds__(256, 1) void k_upsample(const float* __restrict__ vcp, const float* __restrict__ wp,
                                                     const float* __restrict__ bias, const float* __restrict__ tpl,
                                                     float* __restrict__ out, int B, int MT, int nwg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wg = xcd_remap(blockIdx.x, nwg);
    const int MP = (MT + NT - 1) / NT, mgroups = (MP + 3) >> 2;
    const int ob = wg / mgroups, mp = (wg % mgroups) * 4 + wave;
    if (mp >= MP) return;
    int mt[NT];
    const f32x4* a_base[NT];
#pragma unroll
    for (int m = 0; m < NT; ++m) {
        mt[m] = NT * mp + m < MT ? NT * mp + m : NT * mp;        // ragged tail: repeat the first tile (its result is not stored)
        a_base[m] = reinterpret_cast<const f32x4*>(vcp) + ((size_t)mt[m] * 3 * kCB * 4) * 64 + lane;
    }
    const f32x4* w_base = reinterpret_cast<const f32x4*>(wp) + ((size_t)ob * kCB * 4) * 64 + lane;
    const size_t w_tap = (size_t)kOB * kCB * 4 * 64, a_lp = (size_t)kCB * 4 * 64;
    f32x16 acc[NT][3], tot[NT][3];
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
        for (int l = 0; l < 3; ++l) { acc[m][l] = zero16(); tot[m][l] = zero16(); }
    // software pipeline over the 56 (cb,g) steps: the operand fragments of step s+1 are requested before the MFMAs of step s
    // are queued (the fence keeps that order), so their L2 latency hides behind 28*NT MFMAs
    f32x4 x[NT][3], w[3];
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
        for (int l = 0; l < 3; ++l) x[m][l] = a_base[m][l * a_lp];
#pragma unroll
    for (int k = 0; k < 3; ++k) w[k] = w_base[k * w_tap];
#pragma unroll 1
    for (int cb = 0; cb < kCB; ++cb) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int sn = cb * 4 + g + 1 < kCB * 4 ? cb * 4 + g + 1 : cb * 4 + g;     // next step (last one re-loads itself)
            const size_t o = (size_t)sn * 64;
            f32x4 nx[NT][3], nw[3];
#pragma unroll
            for (int m = 0; m < NT; ++m)
#pragma unroll
                for (int l = 0; l < 3; ++l) nx[m][l] = a_base[m][l * a_lp + o];
#pragma unroll
            for (int k = 0; k < 3; ++k) nw[k] = w_base[k * w_tap + o];
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // out l gets input l' = l + k - 1 (the zero-padding taps are never multiplied); ordered so that consecutive
                // MFMAs never write the same accumulator
#pragma unroll
                for (int m = 0; m < NT; ++m) acc[m][0] = GATOR_MFMA(x[m][0][j], w[1][j], acc[m][0]);
#pragma unroll
                for (int m = 0; m < NT; ++m) acc[m][1] = GATOR_MFMA(x[m][0][j], w[0][j], acc[m][1]);
#pragma unroll
                for (int m = 0; m < NT; ++m) acc[m][2] = GATOR_MFMA(x[m][1][j], w[0][j], acc[m][2]);
#pragma unroll
                for (int m = 0; m < NT; ++m) acc[m][0] = GATOR_MFMA(x[m][1][j], w[2][j], acc[m][0]);
#pragma unroll
                for (int m = 0; m < NT; ++m) acc[m][1] = GATOR_MFMA(x[m][1][j], w[1][j], acc[m][1]);
#pragma unroll
                for (int m = 0; m < NT; ++m) acc[m][2] = GATOR_MFMA(x[m][2][j], w[1][j], acc[m][2]);
#pragma unroll
                for (int m = 0; m < NT; ++m) acc[m][1] = GATOR_MFMA(x[m][2][j], w[2][j], acc[m][1]);
            }
#pragma unroll
            for (int m = 0; m < NT; ++m)
#pragma unroll
                for (int l = 0; l < 3; ++l) x[m][l] = nx[m][l];
#pragma unroll
            for (int k = 0; k < 3; ++k) w[k] = nw[k];
        }
        // two-level summation: chains of <= 96 products per 32-vertex block, then 14 partial sums (fp32 accuracy)
#pragma unroll
        for (int m = 0; m < NT; ++m)
#pragma unroll
            for (int l = 0; l < 3; ++l) { tot[m][l] += acc[m][l]; acc[m][l] = zero16(); }
    }
    const int o = 32 * ob + (lane & 31), h = lane >> 5;
    if (o >= kNV) return;
    const float bo = bias[o];
    const float t0 = tpl[o * 3] , t1 = tpl[o * 3 + 1], t2 = tpl[o * 3 + 2];
#pragma unroll
    for (int m = 0; m < NT; ++m) {
        if (m > 0 && NT * mp + m >= MT) break;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int s = 32 * mt[m] + kap(r) + 4 * h;
            if (s < B) {
                F3 v;
                v.x = (tot[m][0][r] + bo) + t0;     // conv output (+bias) first, then the template add: MDR.py:167-168
                v.y = (tot[m][1][r] + bo) + t1;
                v.z = (tot[m][2][r] + bo) + t2;
                *reinterpret_cast<F3*>(out + ((int64_t)s * kNV + o) * 3) = v;
            }
        }
    }
}

}  // namespace

int launch_pack_vc(const float* vc, int B, float* vcp, void* stream) {
    const int MT = (B + 31) / 32;
    const int64_t total = (int64_t)MT * 3 * kCB * kTile;
    k_pack_vc<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(vc, B, vcp, total);
    GATOR_HIP_CHECK(hipGetLastError());
    return GATOR_OK;
}

int launch_upsample(const FusedState* f, const gator_ctx* c, int B, float* verts, void* stream) {
    const int MT = (B + 31) / 32;
    if (MT >= 8) {          // >= 225 samples: two tiles per wave
        const int MP = (MT + 1) / 2, nwg = kOB * ((MP + 3) / 4);
        k_upsample<2><<<nwg, 256, 0, (hipStream_t)stream>>>(f->vcp, f->up_w, c->w.up_b, c->w.v6890, verts, B, MT, nwg);
        GATOR_HIP_CHECK(hipGetLastError());
        return GATOR_OK;
    }
    const int nwg = kOB * ((MT + 3) / 4);
    k_upsample<1><<<nwg, 256, 0, (hipStream_t)stream>>>(f->vcp, f->up_w, c->w.up_b, c->w.v6890, verts, B, MT, nwg);
    GATOR_HIP_CHECK(hipGetLastError());
    return GATOR_OK;
}

}  // namespace gator
