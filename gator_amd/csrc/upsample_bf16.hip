// bf16 variant of the vertex regressor (BASELINE config 3: "full GATOR forward bf16, MFMA vertex regressor"): the
// upsample_conv GEMM (lib/models/MDR.py:122,167-168) on v_mfma_f32_32x32x16_bf16 -- bf16 operands, fp32 accumulate, fp32
// bias/template epilogue and fp32 output.  Everything upstream stays fp32, so parity is MPJPE-level (bf16 rounding of the
// 431x3 coarse vertices and of the weights: ~0.2 mm rms on a vertex), not the 1e-3 mm of the fp32 path.
//
// Same tap-sharing formulation as upsample_fused.hip (7 MFMAs per loaded k-step, padding taps never multiplied) with
// K = 16 coarse vertices per MFMA.  At 16x the fp32 MFMA rate the kernel is bound by operand traffic, so one wave owns TWO
// 32-sample tiles per weight fragment (64 samples x 32 vertices x 3 coords).
#include "fused_common.h"
#include "fused_state.h"

namespace gator {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int kS16 = 28;      // 16-wide k steps over the 431 (->448) coarse vertices

#define GATOR_MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// dst[tap][ob][s][lane][j] = bf16( w[32ob + (lane&31)][16s + 8(lane>>5) + j][tap] )
__global__ void k_pack_up_bf16(const float* __restrict__ w, __bf16* __restrict__ dst, int64_t total) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int j = e & 7, lane = (e >> 3) & 63;
    int64_t r = e >> 9;
    const int s = r % kS16; r /= kS16;
    const int ob = r % kOB;
    const int tap = (int)(r / kOB);
    const int o = 32 * ob + (lane & 31), c = 16 * s + 8 * (lane >> 5) + j;
    dst[e] = (o < kNV && c < kV) ? (__bf16)w[((int64_t)o * kV + c) * 3 + tap] : (__bf16)0.f;
}

// vcp[mt][l'][s][lane][j] = bf16( vc[32mt + (lane&31)][16s + 8(lane>>5) + j][l'] )
__global__ void k_pack_vc_bf16(const float* __restrict__ vc, int B, __bf16* __restrict__ vcp, int64_t total) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int j = e & 7, lane = (e >> 3) & 63;
    int64_t r = e >> 9;
    const int s = r % kS16; r /= kS16;
    const int lp = r % 3;
    const int mt = (int)(r / 3);
    const int smp = 32 * mt + (lane & 31), c = 16 * s + 8 * (lane >> 5) + j;
    vcp[e] = (smp < B && c < kV) ? (__bf16)vc[((int64_t)smp * kV + c) * 3 + lp] : (__bf16)0.f;
}

struct __attribute__((packed)) F3 { float x, y, z; };

__global__ __launch_bounds__(256) void k_upsample_bf16(const __bf16* __restrict__ vcp, const __bf16* __restrict__ wp,
                                                       const float* __restrict__ bias, const float* __restrict__ tpl,
                                                       float* __restrict__ out, int B, int MT, int nwg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wg = xcd_remap(blockIdx.x, nwg);
    const int MP = (MT + 1) >> 1, mgroups = (MP + 3) >> 2;
    const int ob = wg / mgroups, mp = (wg % mgroups) * 4 + wave;          // mp: pair of 32-sample tiles
    if (mp >= MP) return;
    const int mt0 = 2 * mp, mt1 = (2 * mp + 1 < MT) ? 2 * mp + 1 : 2 * mp;   // odd tile count: the last pair repeats its tile
    const bf16x8* a0p = reinterpret_cast<const bf16x8*>(vcp) + ((size_t)mt0 * 3 * kS16) * 64 + lane;
    const bf16x8* a1p = reinterpret_cast<const bf16x8*>(vcp) + ((size_t)mt1 * 3 * kS16) * 64 + lane;
    const bf16x8* wq = reinterpret_cast<const bf16x8*>(wp) + ((size_t)ob * kS16) * 64 + lane;
    const size_t w_tap = (size_t)kOB * kS16 * 64, a_lp = (size_t)kS16 * 64;
    f32x16 acc[2][3];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int l = 0; l < 3; ++l) acc[m][l] = zero16();
#pragma unroll 2
    for (int s = 0; s < kS16; ++s) {
        const size_t o = (size_t)s * 64;
        const bf16x8 w0 = wq[o], w1 = wq[w_tap + o], w2 = wq[2 * w_tap + o];
        const bf16x8 x0 = a0p[o], x1 = a0p[a_lp + o], x2 = a0p[2 * a_lp + o];
        const bf16x8 y0 = a1p[o], y1 = a1p[a_lp + o], y2 = a1p[2 * a_lp + o];
        // out l gets input l' = l + k - 1 ; A = samples (rows), B = vertices (cols)
        acc[0][0] = GATOR_MFMA_BF16(x0, w1, acc[0][0]);
        acc[1][0] = GATOR_MFMA_BF16(y0, w1, acc[1][0]);
        acc[0][1] = GATOR_MFMA_BF16(x0, w0, acc[0][1]);
        acc[1][1] = GATOR_MFMA_BF16(y0, w0, acc[1][1]);
        acc[0][2] = GATOR_MFMA_BF16(x1, w0, acc[0][2]);
        acc[1][2] = GATOR_MFMA_BF16(y1, w0, acc[1][2]);
        acc[0][0] = GATOR_MFMA_BF16(x1, w2, acc[0][0]);
        acc[1][0] = GATOR_MFMA_BF16(y1, w2, acc[1][0]);
        acc[0][1] = GATOR_MFMA_BF16(x1, w1, acc[0][1]);
        acc[1][1] = GATOR_MFMA_BF16(y1, w1, acc[1][1]);
        acc[0][2] = GATOR_MFMA_BF16(x2, w1, acc[0][2]);
        acc[1][2] = GATOR_MFMA_BF16(y2, w1, acc[1][2]);
        acc[0][1] = GATOR_MFMA_BF16(x2, w2, acc[0][1]);
        acc[1][1] = GATOR_MFMA_BF16(y2, w2, acc[1][1]);
    }
    const int ov = 32 * ob + (lane & 31), h = lane >> 5;
    if (ov >= kNV) return;
    const float bo = bias[ov];
    const float t0 = tpl[ov * 3], t1 = tpl[ov * 3 + 1], t2 = tpl[ov * 3 + 2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        if (m == 1 && mt1 == mt0) break;
        const int mt = m ? mt1 : mt0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int smp = 32 * mt + kap(r) + 4 * h;
            if (smp < B) {
                F3 v;
                v.x = (acc[m][0][r] + bo) + t0;
                v.y = (acc[m][1][r] + bo) + t1;
                v.z = (acc[m][2][r] + bo) + t2;
                *reinterpret_cast<F3*>(out + ((int64_t)smp * kNV + ov) * 3) = v;
            }
        }
    }
}

}  // namespace

size_t upsample_bf16_weight_elems() { return (size_t)3 * kOB * kS16 * 512; }
size_t upsample_bf16_vcp_elems(int B) { return (size_t)((B + 31) / 32) * 3 * kS16 * 512; }

int pack_upsample_bf16(const float* up_w, void* dst, void* stream) {
    const int64_t total = (int64_t)upsample_bf16_weight_elems();
    k_pack_up_bf16<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(up_w, (__bf16*)dst, total);
    GATOR_HIP_CHECK(hipGetLastError());
    return GATOR_OK;
}

int launch_upsample_bf16(const FusedState* f, const gator_ctx* c, const float* vc, int B, float* verts, void* stream) {
    const int MT = (B + 31) / 32;
    const int64_t total = (int64_t)upsample_bf16_vcp_elems(B);
    k_pack_vc_bf16<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(vc, B, (__bf16*)f->vcp16, total);
    const int MP = (MT + 1) / 2;
    const int nwg = kOB * ((MP + 3) / 4);
    k_upsample_bf16<<<nwg, 256, 0, (hipStream_t)stream>>>((const __bf16*)f->vcp16, (const __bf16*)f->up_w16, c->w.up_b, c->w.v6890,
                                                         verts, B, MT, nwg);
    GATOR_HIP_CHECK(hipGetLastError());
    return GATOR_OK;
}

}  // namespace gator
