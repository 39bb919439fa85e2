// Where does k_upsample_x3 lose its time?  The product kernel (gator_amd/csrc/upsample_x3.hip) with parts switched off one at a
// time, on random planes at the headline shape (B = 256: 8 sample tiles x 216 vertex blocks).  Results are wrong with any
// flag set -- this is a stopwatch, not a test.
// Build: hipcc -O3 --offload-arch=gfx950 -Igator_amd/csrc -Iinclude tools/microbench/upsample_lab.hip -o tools/microbench/upsample_lab.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#include "fused_common.h"
#include "fused_state.h"
using namespace gator;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int kS16 = 28;
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
struct __attribute__((packed)) F3 { float x, y, z; };

enum { F_NOWLOAD = 1, F_NOALOAD = 2, F_NOLDSREAD = 4, F_NOSTORE = 8, F_NOFLUSH = 16, F_NOMFMA = 32, F_NOBAR = 64 };

constexpr int kX3Ring = 4;
template <int FL, int kX3Waves>
__global__ __launch_bounds__(64 * (kX3Waves + 1), 8 / kX3Waves) void k_up(const __bf16* __restrict__ vcp, const __bf16* __restrict__ wp,
                                                              const float* __restrict__ bias, const float* __restrict__ tpl,
                                                              float* __restrict__ out, int B, int MT, int nwg, int64_t a_plane,
                                                              int64_t w_plane) {
    __shared__ bf16x8 wl[2][9][64];
    __shared__ f32x4 tot[kX3Waves][12][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wg = xcd_remap(blockIdx.x, nwg);
    const int mgroups = (MT + kX3Waves - 1) / kX3Waves;
    const int ob = wg / mgroups;
    const bf16x8* wu = reinterpret_cast<const bf16x8*>(wp) + ((size_t)ob * kS16) * 64;
    const size_t w_tap = (size_t)kOB * kS16 * 64, wpl = (size_t)w_plane / 8;
#define W_AT(e, step) wu[(size_t)((e) % 3) * wpl + (size_t)((e) / 3) * w_tap + (size_t)(step) * 64 + lane]
    if (wave == kX3Waves) {
        bf16x8 ring[kX3Ring][9];
#pragma unroll
        for (int j = 0; j < kX3Ring; ++j)
#pragma unroll
            for (int e = 0; e < 9; ++e) ring[j][e] = W_AT(e, j);
#pragma unroll
        for (int e = 0; e < 9; ++e) {
            wl[0][e][lane] = ring[0][e];
            ring[0][e] = W_AT(e, kX3Ring);
        }
        __syncthreads();
#pragma unroll 1
        for (int s0 = 0; s0 < kS16; s0 += kX3Ring) {
#pragma unroll
            for (int j = 0; j < kX3Ring; ++j) {
                const int s = s0 + j, slot = (j + 1) % kX3Ring;
                const int nxt = s + 1 + kX3Ring < kS16 ? s + 1 + kX3Ring : kS16 - 1;
#pragma unroll
                for (int e = 0; e < 9; ++e) {
                    wl[(s + 1) & 1][e][lane] = ring[slot][e];
                    if (!(FL & F_NOWLOAD)) ring[slot][e] = W_AT(e, nxt);
                }
                if (!(FL & F_NOBAR)) __syncthreads();
            }
        }
        return;
    }
    const int mt_raw = (wg % mgroups) * kX3Waves + wave;
    const bool live = mt_raw < MT;
    const int mt = live ? mt_raw : MT - 1;
    const bf16x8* ab = reinterpret_cast<const bf16x8*>(vcp) + ((size_t)__builtin_amdgcn_readfirstlane(mt) * 3 * kS16) * 64 + lane;
    const size_t a_lp = (size_t)kS16 * 64, ap = (size_t)a_plane / 8;
    f32x16 big[3], sm[3];
#pragma unroll
    for (int l = 0; l < 3; ++l) { big[l] = zero16(); sm[l] = zero16(); }
    bf16x8 x[3][3];
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
        for (int p = 0; p < 3; ++p) x[l][p] = ab[p * ap + l * a_lp];
    __syncthreads();
    bf16x8 wc[3];
    if (FL & F_NOLDSREAD) { wc[0] = wl[0][0][lane]; wc[1] = wl[0][1][lane]; wc[2] = wl[0][2][lane]; }
#pragma unroll 1
    for (int s = 0; s < kS16; ++s) {
        const int cur = s & 1;
        const size_t on = (size_t)(s + 1 < kS16 ? s + 1 : s) * 64;
#pragma unroll
        for (int lp = 0; lp < 3; ++lp) {
#pragma unroll
            for (int k = 2; k >= 0; --k) {
                const int l = lp + 1 - k;
                if (l < 0 || l > 2) continue;
                bf16x8 w0, w1, w2;
                if (FL & F_NOLDSREAD) { w0 = wc[0]; w1 = wc[1]; w2 = wc[2]; }
                else { w0 = wl[cur][3 * k + 0][lane]; w1 = wl[cur][3 * k + 1][lane]; w2 = wl[cur][3 * k + 2][lane]; }
                if (!(FL & F_NOMFMA)) {
                    big[l] = MFMA_BF16(x[lp][0], w0, big[l]);
                    sm[l] = MFMA_BF16(x[lp][0], w1, sm[l]);
                    sm[l] = MFMA_BF16(x[lp][1], w0, sm[l]);
                    sm[l] = MFMA_BF16(x[lp][1], w1, sm[l]);
                    sm[l] = MFMA_BF16(x[lp][0], w2, sm[l]);
                    sm[l] = MFMA_BF16(x[lp][2], w0, sm[l]);
                } else {
                    big[l][0] += (float)w0[0] + (float)w1[1] + (float)w2[2] + (float)x[lp][0][0] + (float)x[lp][1][0] + (float)x[lp][2][0];
                }
            }
            asm volatile("" ::: "memory");
            if (!(FL & F_NOALOAD)) {
#pragma unroll
                for (int p = 0; p < 3; ++p) x[lp][p] = ab[p * ap + lp * a_lp + on];
            }
        }
        if (!(FL & F_NOFLUSH) && s % 7 == 6 && s != kS16 - 1) {
#pragma unroll
            for (int l = 0; l < 3; ++l)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v4[j] = big[l][4 * g + j];
                    if (s != 6) v4 += tot[wave][l * 4 + g][lane];
                    tot[wave][l * 4 + g][lane] = v4;
                }
#pragma unroll
            for (int l = 0; l < 3; ++l) big[l] = zero16();
        }
        if (!(FL & F_NOBAR)) __syncthreads();
    }
    if (!(FL & F_NOFLUSH)) {
#pragma unroll
        for (int l = 0; l < 3; ++l)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v4 = tot[wave][l * 4 + g][lane];
#pragma unroll
                for (int j = 0; j < 4; ++j) big[l][4 * g + j] += v4[j];
            }
    }
    const int ov = 32 * ob + (lane & 31), h = lane >> 5;
    if (ov >= kNV || !live) return;
    const float bo = bias[ov];
    const float t0 = tpl[ov * 3], t1 = tpl[ov * 3 + 1], t2 = tpl[ov * 3 + 2];
    if (FL & F_NOSTORE) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += big[0][r] + sm[0][r] + big[1][r] + sm[1][r] + big[2][r] + sm[2][r];
        if (s == 1234.5f) out[lane] = s + bo + t0 + t1 + t2;
        return;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int smp = 32 * mt + kap(r) + 4 * h;
        if (smp < B) {
            F3 v;
            v.x = ((big[0][r] + sm[0][r]) + bo) + t0;
            v.y = ((big[1][r] + sm[1][r]) + bo) + t1;
            v.z = ((big[2][r] + sm[2][r]) + bo) + t2;
            *reinterpret_cast<F3*>(out + ((int64_t)smp * kNV + ov) * 3) = v;
        }
    }
}

static __bf16 *g_vcp, *g_wp; static float *g_bias, *g_tpl, *g_out;
static int64_t g_aplane, g_wplane;
static int g_B = 256;

template <int FL, int kX3Waves = 8> void run(const char* what) {
    const int MT = (g_B + 31) / 32, nwg = kOB * ((MT + kX3Waves - 1) / kX3Waves);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) k_up<FL, kX3Waves><<<nwg, 64 * (kX3Waves + 1)>>>(g_vcp, g_wp, g_bias, g_tpl, g_out, g_B, MT, nwg, g_aplane, g_wplane);
    hipDeviceSynchronize();
    const int reps = 30;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) k_up<FL, kX3Waves><<<nwg, 64 * (kX3Waves + 1)>>>(g_vcp, g_wp, g_bias, g_tpl, g_out, g_B, MT, nwg, g_aplane, g_wplane);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s %7.1f us  (%s)\n", what, ms * 1000.f / reps, hipGetErrorString(hipGetLastError()));
}

int main(int argc, char** argv) {
    if (argc > 1) g_B = atoi(argv[1]);
    const int MT = (g_B + 31) / 32;
    g_wplane = (int64_t)3 * kOB * kS16 * 512; g_aplane = (int64_t)MT * 3 * kS16 * 512;
    std::vector<__bf16> hw(3 * g_wplane), ha(3 * g_aplane);
    srand(1);
    for (auto& v : hw) v = (__bf16)((rand() % 2001 - 1000) * 1e-4f);
    for (auto& v : ha) v = (__bf16)((rand() % 2001 - 1000) * 1e-3f);
    hipMalloc(&g_wp, hw.size() * 2); hipMalloc(&g_vcp, ha.size() * 2);
    hipMemcpy(g_wp, hw.data(), hw.size() * 2, hipMemcpyHostToDevice); hipMemcpy(g_vcp, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
    hipMalloc(&g_bias, kOB * 32 * 4); hipMalloc(&g_tpl, kOB * 32 * 12); hipMalloc(&g_out, (size_t)MT * 32 * kNV * 12 + 4096);
    hipMemset(g_bias, 0, kOB * 32 * 4); hipMemset(g_tpl, 0, kOB * 32 * 12);
    printf("B = %d, %d sample tiles x %d vertex blocks; MFMA floor on 216 CUs: 2 waves/SIMD x 1176 x 32 cycles = 75.3k cycles\n", g_B, MT, kOB);
    run<0>("product kernel");
    run<F_NOSTORE>("no output store");
    run<F_NOFLUSH>("no hi*hi flush");
    run<F_NOWLOAD>("loader issues no global loads in the loop");
    run<F_NOALOAD>("no activation reloads");
    run<F_NOLDSREAD>("weights from registers (no LDS reads)");
    run<F_NOBAR>("no per-step barrier (racy)");
    run<F_NOWLOAD | F_NOALOAD>("no global loads at all in the loop");
    run<F_NOWLOAD | F_NOALOAD | F_NOLDSREAD>("MFMA + barrier only");
    run<F_NOWLOAD | F_NOALOAD | F_NOLDSREAD | F_NOBAR | F_NOFLUSH | F_NOSTORE>("MFMA only");
    run<F_NOMFMA>("everything but the MFMAs");
    run<0>("product kernel again");
    run<0, 4>("4 compute waves + loader per workgroup, 2 workgroups per CU");
    run<F_NOSTORE, 4>("  same, no output store");
    run<F_NOWLOAD | F_NOALOAD | F_NOLDSREAD | F_NOBAR | F_NOFLUSH | F_NOSTORE, 4>("  same, MFMA only");
    run<0, 2>("2 compute waves + loader per workgroup, 4 workgroups per CU");
    return 0;
}
