// Playground for the two-plane (fp16 hi/lo) vertex regressor: 64 vertices x 128 samples per workgroup, every operand staged through
// LDS by LDS-DMA from all eight waves (no loader wave, no compiler-visible VMEM in the loop), four one-k-step stages in flight.
// Checks a few samples against a double-precision product of the same planes, then times it at the headline shape.
// Build: hipcc -O3 --offload-arch=gfx950 -Igator_amd/csrc -Iinclude tools/microbench/upsample_x2_lab.hip -o tools/microbench/upsample_x2_lab.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <cstdlib>
#include "fused_common.h"
#include "fused_state.h"
using namespace gator;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int kS16 = 28;
#define MFMA_F16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
struct __attribute__((packed)) F3 { float x, y, z; };

// one 1 KiB fragment global -> LDS; base in SGPRs, per-lane byte offset in one VGPR
__device__ __forceinline__ void dma_frag(const void* base, unsigned lane_off, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_off), "s"(base), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int kNBuf = 4, kStageFrags = 36;          // A: 4 sample tiles x (3 positions x 2 planes), W: 2 vertex blocks x (3 taps x 2 planes)
// ap: [mgroup][s][mt 4][lp 3][plane 2][lane][8]   wp: [ob pair][s][ob 2][tap 3][plane 2][lane][8]
enum { F_NODMA = 1, F_NOLDSREAD = 2, F_NOMFMA = 4, F_NOBAR = 8, F_NOSTORE = 16 };
template <int AHEAD, int FL = 0, int BARP = 1>
__global__ __launch_bounds__(512, 1) void k_up2(const _Float16* __restrict__ ap, const _Float16* __restrict__ wp,
                                                const float* __restrict__ bias, const float* __restrict__ tpl, float* __restrict__ out,
                                                int B, int MG, int nwg, float unscale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    f16x8 (*st)[kStageFrags][64] = reinterpret_cast<f16x8 (*)[kStageFrags][64]>(lds_raw);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wg = xcd_remap(blockIdx.x, nwg);
    const int obp = wg / MG, mg = wg % MG;
    const int mi = wave & 3, oi = wave >> 2;
    const unsigned lane_off = lane * 16;
    const unsigned lds0 = (unsigned)(unsigned long long)lds_raw;
    const char* a_src = reinterpret_cast<const char*>(ap) + (size_t)mg * kS16 * 24 * 1024;
    const char* w_src = reinterpret_cast<const char*>(wp) + (size_t)obp * kS16 * 12 * 1024;
    // fragments of a stage this wave copies: f = wave + 8 i (i < 5, f < 36); source base, per-stage stride and LDS offset of each
    const char* src[5]; unsigned stride[5], dsto[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int f = wave + 8 * i;
        src[i] = f < 24 ? a_src + f * 1024 : w_src + (f - 24) * 1024;
        stride[i] = f < 24 ? 24 * 1024 : 12 * 1024;
        dsto[i] = lds0 + f * 1024;
    }
    auto issue = [&](int s) {
        const unsigned dst = (unsigned)(s & (kNBuf - 1)) * kStageFrags * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_frag(src[i] + (size_t)s * stride[i], lane_off, dst + dsto[i]);
        if (wave < 4) dma_frag(src[4] + (size_t)s * stride[4], lane_off, dst + dsto[4]);
    };
    // own copies of stage `pub` must have landed before the barrier that publishes it; `younger` stages stay in flight
    auto wait_landed = [&](int younger) {
        if (younger >= 2) { if (wave < 4) wait_vm<10>(); else wait_vm<8>(); }
        else if (younger == 1) { if (wave < 4) wait_vm<5>(); else wait_vm<4>(); }
        else wait_vm<0>();
    };
    auto load_ops = [&](int s, f16x8 (&a)[3][2], f16x8 (&w)[3][2]) {
        const f16x8 (*cur)[64] = st[s & (kNBuf - 1)];
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                a[q][p] = cur[mi * 6 + q * 2 + p][lane];
                w[q][p] = cur[24 + oi * 6 + q * 2 + p][lane];
            }
    };
    f32x16 big[3], sm[3], tot[3];
#pragma unroll
    for (int l = 0; l < 3; ++l) { big[l] = zero16(); sm[l] = zero16(); tot[l] = zero16(); }
    auto mma = [&](int s, const f16x8 (&a)[3][2], const f16x8 (&w)[3][2]) {
#pragma unroll
        for (int lp = 0; lp < 3; ++lp)
#pragma unroll
            for (int k = 2; k >= 0; --k) {
                const int l = lp + 1 - k;
                if (l < 0 || l > 2) continue;
                if (FL & F_NOMFMA) { big[l][0] += (float)a[lp][0][0] + (float)a[lp][1][0] + (float)w[k][0][0] + (float)w[k][1][0]; continue; }
                big[l] = MFMA_F16(a[lp][0], w[k][0], big[l]);
                sm[l] = MFMA_F16(a[lp][0], w[k][1], sm[l]);
                sm[l] = MFMA_F16(a[lp][1], w[k][0], sm[l]);
            }
        if (s % 7 == 6) {
#pragma unroll
            for (int l = 0; l < 3; ++l) { tot[l] += big[l]; big[l] = zero16(); }
        }
    };
    issue(0); issue(1); issue(2);
    f16x8 a0[3][2], w0[3][2];
    if (AHEAD) {
        // Stage s + 1 is published by the barrier at the top of step s, so the operands step s + 1 needs first (position 0, taps 0
        // and 1) are read at the END of step s, behind its last MFMA: no LDS latency in front of a step's first MFMA.
        auto load_first = [&](int s) {
            const f16x8 (*cur)[64] = st[s & (kNBuf - 1)];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                a0[0][p] = cur[mi * 6 + p][lane];
                w0[0][p] = cur[24 + oi * 6 + p][lane];
                w0[1][p] = cur[24 + oi * 6 + 2 + p][lane];
            }
        };
        auto load_rest = [&](int s) {
            const f16x8 (*cur)[64] = st[s & (kNBuf - 1)];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                a0[1][p] = cur[mi * 6 + 2 + p][lane];
                w0[2][p] = cur[24 + oi * 6 + 4 + p][lane];
                a0[2][p] = cur[mi * 6 + 4 + p][lane];
            }
        };
        wait_landed(2);
        __builtin_amdgcn_s_barrier();
        load_first(0);
#pragma unroll 1
        for (int s = 0; s < kS16; ++s) {
            if (s + 1 < kS16) wait_landed(kS16 - 2 - s < 1 ? kS16 - 2 - s : 1);
            if (!(FL & F_NOBAR)) __builtin_amdgcn_s_barrier();
            if (!(FL & F_NODMA) && s + 3 < kS16) issue(s + 3);
            if (!(FL & F_NOLDSREAD) || s == 0) load_rest(s);
            mma(s, a0, w0);
            if (s + 1 < kS16 && !(FL & F_NOLDSREAD)) load_first(s + 1);
        }
    } else {
#pragma unroll 1
        for (int s = 0; s < kS16; ++s) {
            wait_landed(kS16 - 1 - s < 2 ? kS16 - 1 - s : 2);
            if (!(FL & F_NOBAR) && s % BARP == 0) __builtin_amdgcn_s_barrier();
            if (!(FL & F_NODMA) && s + 3 < kS16) issue(s + 3);
            if (!(FL & F_NOLDSREAD) || s == 0) load_ops(s, a0, w0);
            mma(s, a0, w0);
        }
    }
    const int ob = 2 * obp + oi, mt = 4 * mg + mi;
    const int ov = 32 * ob + (lane & 31), h = lane >> 5;
    if (ov >= kNV) return;
    if (FL & F_NOSTORE) { float q = 0.f; for (int l = 0; l < 3; ++l) for (int r = 0; r < 16; ++r) q += tot[l][r] + sm[l][r] + big[l][r]; if (q == 1234.5f) out[lane] = q; return; }
    const float bo = bias[ov];
    const float t0 = tpl[ov * 3], t1 = tpl[ov * 3 + 1], t2 = tpl[ov * 3 + 2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int smp = 32 * mt + kap(r) + 4 * h;
        if (smp < B) {
            F3 v;
            v.x = ((tot[0][r] + sm[0][r]) * unscale + bo) + t0;
            v.y = ((tot[1][r] + sm[1][r]) * unscale + bo) + t1;
            v.z = ((tot[2][r] + sm[2][r]) * unscale + bo) + t2;
            *reinterpret_cast<F3*>(out + ((int64_t)smp * kNV + ov) * 3) = v;
        }
    }
}

__global__ void k_denorm(float* out) {
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)9.5367431640625e-07f; b[j] = (_Float16)1.0f; }      // 2^-20: an fp16 subnormal
    f32x16 acc = zero16();
    acc = MFMA_F16(a, b, acc);
    if (threadIdx.x == 0) out[0] = acc[0];
}
int main(int argc, char** argv) {
    { float* d; hipMalloc(&d, 4); k_denorm<<<1, 64>>>(d); float h = -1; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
      printf("fp16 subnormal through the MFMA: 16 x 2^-20 x 1 = %.6e (kept: 1.525879e-05, flushed: 0)\n", h); }
    const int B = argc > 1 ? atoi(argv[1]) : 256;
    const int MT = (B + 31) / 32, MG = (MT + 3) / 4, OBP = kOB / 2;
    const float sx = 16.f, sw = 1024.f;
    std::vector<float> x((size_t)MG * 128 * kV * 3, 0.f), w((size_t)kNV * kV * 3);
    srand(3);
    for (size_t i = 0; i < (size_t)B * kV * 3; ++i) x[i] = (rand() % 20001 - 10000) * 1e-4f * 0.3f;
    for (auto& v : w) v = (rand() % 20001 - 10000) * 1e-4f * 0.03f;
    // planes
    const size_t a_elems = (size_t)MG * kS16 * 24 * 512, w_elems = (size_t)OBP * kS16 * 12 * 512;
    std::vector<_Float16> ha(a_elems), hw(w_elems);
    auto split = [](float v, _Float16& h, _Float16& l) { h = (_Float16)v; l = (_Float16)(v - (float)h); };
    for (int mg = 0; mg < MG; ++mg) for (int s = 0; s < kS16; ++s) for (int m = 0; m < 4; ++m) for (int lp = 0; lp < 3; ++lp)
        for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 8; ++j) {
            const int smp = 32 * (4 * mg + m) + (lane & 31), c = 16 * s + 8 * (lane >> 5) + j;
            const float v = (c < kV) ? x[((size_t)smp * kV + c) * 3 + lp] * sx : 0.f;
            _Float16 h, l; split(v, h, l);
            const size_t e = ((((size_t)mg * kS16 + s) * 4 + m) * 3 + lp) * 2;
            ha[(e + 0) * 512 + lane * 8 + j] = h; ha[(e + 1) * 512 + lane * 8 + j] = l;
        }
    for (int op = 0; op < OBP; ++op) for (int s = 0; s < kS16; ++s) for (int o2 = 0; o2 < 2; ++o2) for (int tap = 0; tap < 3; ++tap)
        for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 8; ++j) {
            const int o = 32 * (2 * op + o2) + (lane & 31), c = 16 * s + 8 * (lane >> 5) + j;
            const float v = (o < kNV && c < kV) ? w[((size_t)o * kV + c) * 3 + tap] * sw : 0.f;
            _Float16 h, l; split(v, h, l);
            const size_t e = ((((size_t)op * kS16 + s) * 2 + o2) * 3 + tap) * 2;
            hw[(e + 0) * 512 + lane * 8 + j] = h; hw[(e + 1) * 512 + lane * 8 + j] = l;
        }
    _Float16 *da, *dw; float *dbias, *dtpl, *dout;
    hipMalloc(&da, a_elems * 2); hipMalloc(&dw, w_elems * 2);
    hipMemcpy(da, ha.data(), a_elems * 2, hipMemcpyHostToDevice); hipMemcpy(dw, hw.data(), w_elems * 2, hipMemcpyHostToDevice);
    hipMalloc(&dbias, kOB * 32 * 4); hipMalloc(&dtpl, kOB * 32 * 12); hipMalloc(&dout, (size_t)MG * 128 * kNV * 12);
    hipMemset(dbias, 0, kOB * 32 * 4); hipMemset(dtpl, 0, kOB * 32 * 12);
    const int nwg = OBP * MG;
    const size_t lds = (size_t)kNBuf * kStageFrags * 1024;
    hipFuncSetAttribute((const void*)k_up2<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)k_up2<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    std::vector<float> ho((size_t)B * kNV * 3);
    for (int ahead = 0; ahead < 2; ++ahead) {
        hipMemset(dout, 0, (size_t)B * kNV * 12);
        if (ahead) k_up2<1><<<nwg, 512, lds>>>(da, dw, dbias, dtpl, dout, B, MG, nwg, 1.f / (sx * sw));
        else k_up2<0><<<nwg, 512, lds>>>(da, dw, dbias, dtpl, dout, B, MG, nwg, 1.f / (sx * sw));
        hipDeviceSynchronize();
        printf("launch: %s\n", hipGetErrorString(hipGetLastError()));
        hipMemcpy(ho.data(), dout, ho.size() * 4, hipMemcpyDeviceToHost);
        double worst = 0, worst_exact = 0;
        for (int smp : {0, 1, 37, B / 2 + 5, B - 1})
            for (int o = 0; o < kNV; o += 7)
                for (int l = 0; l < 3; ++l) {
                    double ref = 0, ex = 0;
                    for (int c = 0; c < kV; ++c)
                        for (int k = 0; k < 3; ++k) {
                            const int lp = l + k - 1;
                            if (lp < 0 || lp > 2) continue;
                            const float xv = x[((size_t)smp * kV + c) * 3 + lp] * sx, wv = w[((size_t)o * kV + c) * 3 + k] * sw;
                            _Float16 xh, xl, wh, wl; split(xv, xh, xl); split(wv, wh, wl);
                            ref += (double)xh * (double)wh + (double)xh * (double)wl + (double)xl * (double)wh;
                            ex += (double)xv * (double)wv;
                        }
                    const double got = ho[((size_t)smp * kNV + o) * 3 + l];
                    worst = fmax(worst, fabs(got - ref / (sx * sw)));
                    worst_exact = fmax(worst_exact, fabs(got - ex / (sx * sw)));
                }
        printf("ahead=%d  max |gpu - planes in double| = %.3e   max |gpu - exact product| = %.3e  (outputs ~0.3)\n", ahead, worst, worst_exact);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int reps = 30;
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) {
            if (ahead) k_up2<1><<<nwg, 512, lds>>>(da, dw, dbias, dtpl, dout, B, MG, nwg, 1.f / (sx * sw));
            else k_up2<0><<<nwg, 512, lds>>>(da, dw, dbias, dtpl, dout, B, MG, nwg, 1.f / (sx * sw));
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("ahead=%d  B=%d  %d workgroups: %.1f us per launch\n", ahead, B, nwg, ms * 1000.f / reps);
    }
    auto timeit = [&](auto kern, const char* what) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 3; ++i) kern<<<nwg, 512, lds>>>(da, dw, dbias, dtpl, dout, B, MG, nwg, 1.f / (sx * sw));
        float best = 1e9f, worst = 0.f;
        for (int rep = 0; rep < 7; ++rep) {
            hipEventRecord(e0);
            for (int i = 0; i < 20; ++i) kern<<<nwg, 512, lds>>>(da, dw, dbias, dtpl, dout, B, MG, nwg, 1.f / (sx * sw));
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            best = fminf(best, ms * 50.f); worst = fmaxf(worst, ms * 50.f);
        }
        printf("%-50s %6.1f .. %6.1f us (%s)\n", what, best, worst, hipGetErrorString(hipGetLastError()));
    };
    timeit(k_up2<0, 0>, "all on");
    timeit(k_up2<0, F_NOSTORE>, "no output store");
    timeit(k_up2<0, F_NODMA>, "no copies in the loop");
    timeit(k_up2<0, F_NOLDSREAD>, "no LDS reads");
    timeit(k_up2<0, F_NOBAR>, "no barrier (racy)");
    timeit(k_up2<0, F_NOMFMA>, "no MFMA");
    timeit(k_up2<0, F_NODMA | F_NOLDSREAD>, "MFMA + barrier");
    timeit(k_up2<0, F_NODMA | F_NOLDSREAD | F_NOBAR | F_NOSTORE>, "MFMA only");
    timeit(k_up2<0, F_NOMFMA | F_NOLDSREAD>, "copies + barrier only");
    timeit(k_up2<0, 0>, "all on again");
    timeit(k_up2<0, F_NODMA | F_NOLDSREAD | F_NOSTORE, 1>, "MFMA + barrier every step, no store");
    timeit(k_up2<0, F_NODMA | F_NOLDSREAD | F_NOSTORE, 2>, "MFMA + barrier every 2 steps, no store");
    timeit(k_up2<0, F_NODMA | F_NOLDSREAD | F_NOSTORE, 4>, "MFMA + barrier every 4 steps, no store");
    timeit(k_up2<0, F_NODMA | F_NOLDSREAD | F_NOSTORE, 14>, "MFMA + barrier every 14 steps, no store");
    timeit(k_up2<0, F_NODMA | F_NOLDSREAD | F_NOSTORE | F_NOBAR, 1>, "MFMA, no barrier, no store");
    timeit(k_up2<1, 0>, "AHEAD all on");
    timeit(k_up2<1, F_NOSTORE>, "AHEAD no output store");
    timeit(k_up2<1, F_NODMA>, "AHEAD no copies in the loop");
    timeit(k_up2<1, F_NOLDSREAD>, "AHEAD no LDS reads");
    timeit(k_up2<1, F_NOBAR>, "AHEAD no barrier (racy)");
    timeit(k_up2<1, F_NOMFMA>, "AHEAD no MFMA");
    timeit(k_up2<1, F_NODMA | F_NOLDSREAD>, "AHEAD MFMA + barrier");
    timeit(k_up2<1, 0>, "AHEAD all on again");
    return 0;
}
