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

__global__ __launch_bounds__(256) void k_upsample(const float* __restrict__ vcp, const float* __restrict__ wp,
                                                  const float* __restrict__ bias, const float* __restrict__ tpl,
                                                  float* __restrict__ out, int B, int MT, int nwg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wg = xcd_remap(blockIdx.x, nwg);
    const int mgroups = (MT + 3) >> 2;
    const int ob = wg / mgroups, mt = (wg % mgroups) * 4 + wave;
    if (mt >= MT) return;
    const f32x4* a_base = reinterpret_cast<const f32x4*>(vcp) + ((size_t)mt * 3 * kCB * 4) * 64 + lane;
    const f32x4* w_base = reinterpret_cast<const f32x4*>(wp) + ((size_t)ob * kCB * 4) * 64 + lane;
    const size_t w_tap = (size_t)kOB * kCB * 4 * 64, a_lp = (size_t)kCB * 4 * 64;
    f32x16 acc[3], tot[3];
#pragma unroll
    for (int l = 0; l < 3; ++l) { acc[l] = zero16(); tot[l] = zero16(); }
    // software pipeline over the 56 (cb,g) steps: the 6 operand fragments of step s+1 are requested before the 28 MFMAs of
    // step s are queued (fence keeps the order), so their L2 latency hides behind ~1.8k MFMA cycles
    f32x4 a0 = a_base[0], a1 = a_base[a_lp], a2 = a_base[2 * a_lp];
    f32x4 w0 = w_base[0], w1 = w_base[w_tap], w2 = w_base[2 * w_tap];
#pragma unroll 1
    for (int cb = 0; cb < kCB; ++cb) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int sn = cb * 4 + g + 1 < kCB * 4 ? cb * 4 + g + 1 : cb * 4 + g;     // next step (last one re-loads itself)
            const size_t o = (size_t)sn * 64;
            const f32x4 na0 = a_base[o], na1 = a_base[a_lp + o], na2 = a_base[2 * a_lp + o];
            const f32x4 nw0 = w_base[o], nw1 = w_base[w_tap + o], nw2 = w_base[2 * w_tap + o];
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // out l gets input l' = l + k - 1
                // consecutive MFMAs never share an accumulator (a dependent 32x32x2 pair does not issue back to back)
                acc[0] = GATOR_MFMA(a0[j], w1[j], acc[0]);
                acc[1] = GATOR_MFMA(a0[j], w0[j], acc[1]);
                acc[2] = GATOR_MFMA(a1[j], w0[j], acc[2]);
                acc[0] = GATOR_MFMA(a1[j], w2[j], acc[0]);
                acc[1] = GATOR_MFMA(a1[j], w1[j], acc[1]);
                acc[2] = GATOR_MFMA(a2[j], w1[j], acc[2]);
                acc[1] = GATOR_MFMA(a2[j], w2[j], acc[1]);
            }
            a0 = na0; a1 = na1; a2 = na2;
            w0 = nw0; w1 = nw1; w2 = nw2;
        }
        // two-level summation: chains of <= 96 products per 32-vertex block, then 14 partial sums (fp32 accuracy)
#pragma unroll
        for (int l = 0; l < 3; ++l) { tot[l] += acc[l]; acc[l] = zero16(); }
    }
    const int o = 32 * ob + (lane & 31), h = lane >> 5;
    if (o >= kNV) return;
    const float bo = bias[o];
    const float t0 = tpl[o * 3] , t1 = tpl[o * 3 + 1], t2 = tpl[o * 3 + 2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int s = 32 * mt + kap(r) + 4 * h;
        if (s < B) {
            F3 v;
            v.x = (tot[0][r] + bo) + t0;     // conv output (+bias) first, then the template add: MDR.py:167-168
            v.y = (tot[1][r] + bo) + t1;
            v.z = (tot[2][r] + bo) + t2;
            *reinterpret_cast<F3*>(out + ((int64_t)s * kNV + o) * 3) = v;
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
    const int nwg = kOB * ((MT + 3) / 4);
    k_upsample<<<nwg, 256, 0, (hipStream_t)stream>>>(f->vcp, f->up_w, c->w.up_b, c->w.v6890, verts, B, MT, nwg);
    GATOR_HIP_CHECK(hipGetLastError());
    return GATOR_OK;
}

}  // namespace gator
