// Bring-up path (GATOR_IMPL_BASIC): one simple HIP kernel per reference op, fp32 storage with fp64
// accumulation.  It is NOT the fast path; it exists (a) as the first end-to-end all-HIP forward and (b) as an
// accurate on-device cross-check for the fused MFMA kernels.  Op order follows the reference:
//   GAT.forward lib/models/GAT.py:133-152, GATBlock :33-43, modules.py (Attention :121-138, MGCN :243-255,
//   X_Feat :158-177, MLP :188-196), MDR.forward lib/models/MDR.py:124-170, vanilla_transformer_encoder.py:24-46,82-94.
#include <hip/hip_runtime.h>

#include <cmath>

#include "internal.h"

namespace gator {
namespace {

__device__ __forceinline__ float gelu_exact(double x) { return (float)(0.5 * x * (1.0 + erf(x * 0.70710678118654752440))); }

// C[m][n] = act( sum_k A[m*lda+k] * W[n*ldw+k] + bias[n] ) + res[(m % res_mod)*ldr + n]
// 64x64 tile, BK=16, 256 threads x (4x4) outputs, fp64 accumulation.
template <int ACT>
__global__ __launch_bounds__(256) void k_linear(const float* __restrict__ A, int lda, const float* __restrict__ W, int wsn, int wsk,
                                                const float* __restrict__ bias, const float* res, int ldr,
                                                int res_mod, float* C, int ldc, int M, int N, int K) {
    __shared__ float As[16][68];
    __shared__ float Ws[16][68];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    double acc[4][4] = {};
    for (int k0 = 0; k0 < K; k0 += 16) {
        for (int e = threadIdx.x; e < 64 * 16; e += 256) {
            const int r = e >> 4, kk = e & 15;
            const int m = m0 + r, n = n0 + r, k = k0 + kk;
            As[kk][r] = (m < M && k < K) ? A[(int64_t)m * lda + k] : 0.f;
            Ws[kk][r] = (n < N && k < K) ? W[(int64_t)n * wsn + (int64_t)k * wsk] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float a[4], w[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = As[kk][ty * 4 + i];
                w[i] = Ws[kk][tx * 4 + i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] += (double)a[i] * (double)w[j];
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty * 4 + i;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n >= N) continue;
            double v = acc[i][j] + (bias ? (double)bias[n] : 0.0);
            float o = (ACT == 1) ? gelu_exact(v) : (float)v;
            if (res) o = (float)((double)o + (double)res[(int64_t)(res_mod ? (m % res_mod) : m) * ldr + n]);
            C[(int64_t)m * ldc + n] = o;
        }
    }
}

// One wave per row.  MODE 0: nn.LayerNorm (biased var, eps inside sqrt).  MODE 1: Annotated-Transformer LayerNorm
// (unbiased std, eps added to std; vanilla_transformer_encoder.py:31-34).  GELU_AFTER: GAT tail (GAT.py:148-149).
template <int MODE, int GELU_AFTER>
__global__ __launch_bounds__(256) void k_layernorm(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                   const float* __restrict__ b, float eps, float* __restrict__ out, int ldo,
                                                   int M, int C) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    double v[4], s = 0.0;
    int cnt = 0;
    for (int c = lane; c < C; c += 64) {
        v[cnt] = (double)x[(int64_t)row * ldx + c];
        s += v[cnt++];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const double mean = s / C;
    double q = 0.0;
    for (int i = 0; i < cnt; ++i) q += (v[i] - mean) * (v[i] - mean);
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    cnt = 0;
    for (int c = lane; c < C; c += 64) {
        double y;
        if (MODE == 0) y = (v[cnt] - mean) / sqrt(q / C + (double)eps) * (double)w[c] + (double)b[c];
        else y = (double)w[c] * (v[cnt] - mean) / (sqrt(q / (C - 1)) + (double)eps) + (double)b[c];
        out[(int64_t)row * ldo + c] = GELU_AFTER ? gelu_exact(y) : (float)y;
        ++cnt;
    }
}

// GAT embedding: GraphLinear(2->64), GroupNorm(4,64), GELU, GraphLinear(64->128), + pos table.  GAT.py:69-72,135-144.
__global__ __launch_bounds__(128) void k_gat_embed(const float* __restrict__ pose2d, const float* __restrict__ W0,
                                                   const float* __restrict__ b0, const float* __restrict__ gnw,
                                                   const float* __restrict__ gnb, const float* __restrict__ W3,
                                                   const float* __restrict__ b3, const float* __restrict__ pos,
                                                   float* __restrict__ x, int J) {
    __shared__ double h[64][kMaxJ];
    __shared__ float g[64][kMaxJ];
    const int b = blockIdx.x, t = threadIdx.x;
    const float* p = pose2d + (int64_t)b * J * 2;
    if (t < 64)
        for (int j = 0; j < J; ++j) h[t][j] = (double)W0[t * 2] * p[j * 2] + (double)W0[t * 2 + 1] * p[j * 2 + 1] + (double)b0[t];
    __syncthreads();
    if (t < 64) {
        const int g0 = (t >> 4) << 4;
        double s = 0.0, q = 0.0;
        for (int c = g0; c < g0 + 16; ++c)
            for (int j = 0; j < J; ++j) s += h[c][j];
        const double mean = s / (16 * J);
        for (int c = g0; c < g0 + 16; ++c)
            for (int j = 0; j < J; ++j) q += (h[c][j] - mean) * (h[c][j] - mean);
        const double rstd = 1.0 / sqrt(q / (16 * J) + 1e-5);
        for (int j = 0; j < J; ++j) g[t][j] = gelu_exact((h[t][j] - mean) * rstd * (double)gnw[t] + (double)gnb[t]);
    }
    __syncthreads();
    for (int j = 0; j < J; ++j) {
        double a = (double)b3[t];
        for (int c = 0; c < 64; ++c) a += (double)W3[t * 64 + c] * (double)g[c][j];
        x[((int64_t)b * J + j) * kC + t] = (float)(a + (double)pos[j * kC + t]);
    }
}

// Attention core of modules.py:121-138: one block per sample, thread = (head, query).
__global__ void k_gat_attn(const float* __restrict__ qkv, const float* __restrict__ bias, float* __restrict__ out, int J) {
    const int b = blockIdx.x, t = threadIdx.x;
    if (t >= kH * J) return;
    const int h = t / J, i = t % J;
    const float* base = qkv + (int64_t)b * J * 3 * kC;
    const float* q = base + (int64_t)i * 3 * kC + h * 16;
    double s[kMaxJ], mx = -1e300;
    for (int j = 0; j < J; ++j) {
        const float* k = base + (int64_t)j * 3 * kC + kC + h * 16;
        double d = 0.0;
        for (int e = 0; e < 16; ++e) d += (double)q[e] * (double)k[e];
        s[j] = d * 0.25 + (double)bias[(h * J + i) * J + j];
        mx = fmax(mx, s[j]);
    }
    double l = 0.0;
    for (int j = 0; j < J; ++j) {
        s[j] = exp(s[j] - mx);
        l += s[j];
    }
    for (int e = 0; e < 16; ++e) {
        double o = 0.0;
        for (int j = 0; j < J; ++j) o += s[j] * (double)base[(int64_t)j * 3 * kC + 2 * kC + h * 16 + e];
        out[((int64_t)b * J + i) * kC + h * 16 + e] = (float)(o / l);
    }
}

// MGCN mix (modules.py:247-255): out = diag(A)*(M*h0) + offdiag(A)@(M*h1) + bias + add   (add = attention branch)
__global__ void k_mgcn_mix(const float* __restrict__ h0, const float* __restrict__ h1, const float* __restrict__ Mm,
                           const float* __restrict__ adiag, const float* __restrict__ aoff, const float* __restrict__ bias,
                           const float* __restrict__ add, float* __restrict__ out, int J) {
    const int b = blockIdx.x;
    for (int e = threadIdx.x; e < J * kC; e += blockDim.x) {
        const int i = e / kC, n = e % kC;
        double a = (double)adiag[i] * ((double)Mm[i * kC + n] * (double)h0[((int64_t)b * J + i) * kC + n]);
        for (int j = 0; j < J; ++j)
            a += (double)aoff[i * J + j] * (double)(Mm[j * kC + n] * h1[((int64_t)b * J + j) * kC + n]);
        a += (double)bias[n];
        out[((int64_t)b * J + i) * kC + n] = (float)(a + (double)add[((int64_t)b * J + i) * kC + n]);
    }
}

// out[b][i][n] = sum_j A[i][j] x[b][j][n]   (X_Feat hop masks, modules.py:163-173)
__global__ void k_adjmul(const float* __restrict__ A, const float* __restrict__ x, int ldx, float* __restrict__ out, int ldo,
                         int J, int N) {
    const int b = blockIdx.x;
    for (int e = threadIdx.x; e < J * N; e += blockDim.x) {
        const int i = e / N, n = e % N;
        double a = 0.0;
        for (int j = 0; j < J; ++j) a += (double)A[i * J + j] * (double)x[((int64_t)b * J + j) * ldx + n];
        out[((int64_t)b * J + i) * ldo + n] = (float)a;
    }
}

// pose_combine = cat(pose2d, pose3d/1000, feat) (GATOR.py:18-19) and pose3d output copy.
__global__ void k_build_pc(const float* __restrict__ pose2d, const float* __restrict__ xout, const float* __restrict__ feat,
                           float* __restrict__ pc, float* __restrict__ pose3d, int64_t rows) {
    const int64_t r = blockIdx.x;
    if (r >= rows) return;
    const int t = threadIdx.x;
    if (t < 2) pc[r * 133 + t] = pose2d[r * 2 + t];
    else if (t < 5) {
        const float v = xout[r * 3 + (t - 2)];
        pc[r * 133 + t] = v / 1000.f;
        if (pose3d) pose3d[r * 3 + (t - 2)] = v;
    } else if (t < 133) pc[r * 133 + t] = feat[r * kC + (t - 5)];
}

// verts-feature input [B,431,6] = cat(v431, pc[:, vj, 2:5])   (MDR.py:126-127)
__global__ void k_vert_in(const float* __restrict__ v431, const float* __restrict__ pc, const int32_t* __restrict__ vj,
                          float* __restrict__ out, int J, int64_t total) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int c = e % 6;
    const int64_t r = e / 6;
    const int v = r % kV;
    const int64_t b = r / kV;
    out[e] = c < 3 ? v431[v * 3 + c] : pc[(b * J + vj[v]) * 133 + 2 + (c - 3)];
}

// CrossAttention core (MDR.py:34-46): thread = (b, head, vertex); keys = J joints.
__global__ void k_cross_attn(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                             float* __restrict__ out, int J, int64_t total) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int h = e % 2;
    const int64_t r = e / 2;     // b*431 + vertex
    const int64_t b = r / kV;
    const float* qp = q + r * kE + h * 32;
    double s[kMaxJ], mx = -1e300;
    for (int j = 0; j < J; ++j) {
        const float* kp = k + (b * J + j) * kE + h * 32;
        double d = 0.0;
        for (int c = 0; c < 32; ++c) d += (double)qp[c] * (double)kp[c];
        s[j] = d * 0.17677669529663688110;   // 32 ** -0.5
        mx = fmax(mx, s[j]);
    }
    double l = 0.0;
    for (int j = 0; j < J; ++j) {
        s[j] = exp(s[j] - mx);
        l += s[j];
    }
    for (int c = 0; c < 32; ++c) {
        double o = 0.0;
        for (int j = 0; j < J; ++j) o += s[j] * (double)v[(b * J + j) * kE + h * 32 + c];
        out[r * kE + h * 32 + c] = (float)(o / l);
    }
}

// MultiHeadedAttention core (vanilla_transformer_encoder.py:36-46): block = (b, head), K/V of the head in LDS,
// thread = query.  q,k,v,out: [B,431,64] with head h at channels 32h..32h+31.
__global__ __launch_bounds__(448) void k_self_attn(const float* __restrict__ q, const float* __restrict__ k,
                                                   const float* __restrict__ v, float* __restrict__ out) {
    extern __shared__ float lds[];
    float* Ks = lds;
    float* Vs = lds + kV * 32;
    const int b = blockIdx.x >> 1, h = blockIdx.x & 1;
    for (int e = threadIdx.x; e < kV * 32; e += blockDim.x) {
        const int j = e >> 5, c = e & 31;
        Ks[e] = k[((int64_t)b * kV + j) * kE + h * 32 + c];
        Vs[e] = v[((int64_t)b * kV + j) * kE + h * 32 + c];
    }
    __syncthreads();
    const int i = threadIdx.x;
    if (i >= kV) return;
    float qr[32];
    for (int c = 0; c < 32; ++c) qr[c] = q[((int64_t)b * kV + i) * kE + h * 32 + c];
    const double inv = 1.0 / sqrt(32.0);
    double mx = -1e300;
    for (int j = 0; j < kV; ++j) {
        double d = 0.0;
        for (int c = 0; c < 32; ++c) d += (double)qr[c] * (double)Ks[j * 32 + c];
        mx = fmax(mx, d * inv);
    }
    double o[32] = {}, l = 0.0;
    for (int j = 0; j < kV; ++j) {
        double d = 0.0;
        for (int c = 0; c < 32; ++c) d += (double)qr[c] * (double)Ks[j * 32 + c];
        const double p = exp(d * inv - mx);
        l += p;
        for (int c = 0; c < 32; ++c) o[c] += p * (double)Vs[j * 32 + c];
    }
    for (int c = 0; c < 32; ++c) out[((int64_t)b * kV + i) * kE + h * 32 + c] = (float)(o[c] / l);
}

// bias_norm + GELU on mat_B [B,431,3]: BatchNorm1d(431) eval (channel = vertex) or LayerNorm(3).  MDR.py:158-160
__global__ void k_head_norm(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bb,
                            const float* __restrict__ mean, const float* __restrict__ var, int alpha,
                            float* __restrict__ out, int64_t rows) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const int v = r % kV;
    double a[3] = {(double)x[r * 3], (double)x[r * 3 + 1], (double)x[r * 3 + 2]};
    if (alpha) {
        const double m = (a[0] + a[1] + a[2]) / 3.0;
        const double q = ((a[0] - m) * (a[0] - m) + (a[1] - m) * (a[1] - m) + (a[2] - m) * (a[2] - m)) / 3.0;
        const double rs = 1.0 / sqrt(q + 1e-5);
        for (int c = 0; c < 3; ++c) out[r * 3 + c] = gelu_exact((a[c] - m) * rs * (double)w[c] + (double)bb[c]);
    } else {
        const double rs = 1.0 / sqrt((double)var[v] + 1e-5);
        for (int c = 0; c < 3; ++c) out[r * 3 + c] = gelu_exact((a[c] - (double)mean[v]) * rs * (double)w[v] + (double)bb[v]);
    }
}

// im2col of the zero-padded xyz axis: out[(b*3+l)][c*3+k] = src[b][c][l+k-1]   (Conv1d k=3 p=1 over L=3)
__global__ void k_im2col3(const float* __restrict__ src, float* __restrict__ out, int Cin, int64_t total) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int K = Cin * 3;
    const int col = e % K;
    const int64_t row = e / K;
    const int c = col / 3, kk = col % 3, l = row % 3;
    const int64_t b = row / 3;
    const int ll = l + kk - 1;
    out[e] = (ll >= 0 && ll < 3) ? src[(b * Cin + c) * 3 + ll] : 0.f;
}

// vert_coor = alpha * softmax(A) @ B + C   (MDR.py:165).  ac [B,431,23], bc [(b*3+l)][20], al [B,431] (or null)
__global__ void k_head_mix(const float* __restrict__ ac, const float* __restrict__ bc, const float* __restrict__ al,
                           float* __restrict__ vc, int64_t rows) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const int64_t b = r / kV;
    const float* a = ac + r * 23;
    double mx = -1e300, p[20], l = 0.0;
    for (int m = 0; m < 20; ++m) mx = fmax(mx, (double)a[m]);
    for (int m = 0; m < 20; ++m) {
        p[m] = exp((double)a[m] - mx);
        l += p[m];
    }
    const double sc = al ? pow(1.1, (double)al[r]) : 1.0;
    for (int c = 0; c < 3; ++c) {
        double o = 0.0;
        for (int m = 0; m < 20; ++m) o += (p[m] / l) * (double)bc[(b * 3 + c) * 20 + m];
        vc[r * 3 + c] = (float)(sc * o + (double)a[20 + c]);
    }
}

// verts[b][o][l] = tmp[(b*3+l)][o] + v6890[o][l]   (MDR.py:168)
__global__ void k_up_finish(const float* __restrict__ tmp, const float* __restrict__ tpl, float* __restrict__ out, int64_t total) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int l = e % 3;
    const int64_t r = e / 3;
    const int o = r % kNV;
    const int64_t b = r / kNV;
    out[e] = tmp[(b * 3 + l) * kNV + o] + tpl[o * 3 + l];
}

// hop/path bias fold (modules.py:98-107): bias[h][i][j] = emb[sp][h] + (sum_d W[h][i][j][d]*ea[d][h][i][j]) / max(sp-1,1)
__global__ void k_hop_bias(const int64_t* __restrict__ sp, const float* __restrict__ emb, const float* __restrict__ Wp,
                           const float* __restrict__ ea, float* __restrict__ bias, int J, int D) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= kH * J * J) return;
    const int h = e / (J * J), ij = e % (J * J);
    const int64_t s = sp[ij];
    float acc = 0.f;    // float32 like the reference (mul then sum over the last dim)
    for (int d = 0; d < D; ++d) acc += Wp[((int64_t)h * J * J + ij) * D + d] * ea[(int64_t)d * kH * J * J + h * J * J + ij];
    const float spatial = 1.0f / (float)(s - 1 > 0 ? s - 1 : 1);
    bias[e] = emb[s * kH + h] + acc * spatial;
}

__global__ void k_fold_graph(const float* __restrict__ A, const float* __restrict__ adj2, const int64_t* __restrict__ sp,
                             const float* __restrict__ pos_id, const float* __restrict__ pos_num, float* __restrict__ adiag,
                             float* __restrict__ aoff, float* __restrict__ m1, float* __restrict__ m2, float* __restrict__ pos,
                             int J, int blk) {
    for (int e = threadIdx.x; e < J * J; e += blockDim.x) {
        const int i = e / J, j = e % J;
        if (blk >= 0) {
            const float a = A[i * J + j] + adj2[i * J + j], at = A[j * J + i] + adj2[j * J + i];
            const float s = (at + a) / 2.f;                       // modules.py:247-248
            if (i == j) adiag[blk * J + i] = s;
            aoff[(blk * J + i) * J + j] = (i == j) ? 0.f : s;
        } else {
            m1[e] = sp[e] <= 1 ? 1.f : 0.f;
            m2[e] = sp[e] == 2 ? 1.f : 0.f;
        }
    }
    if (blk < 0)
        for (int e = threadIdx.x; e < J * kC; e += blockDim.x) {
            const int j = e / kC, n = e % kC;
            int deg = 0;
            for (int q = 0; q < J; ++q) deg += (int)(int64_t)A[j * J + q];   // graph_adj.long().sum(1), GAT.py:143
            pos[e] = pos_id[(j + 1) * kC + n] + pos_num[deg * kC + n];
        }
}

template <int ACT>
void linear(hipStream_t st, const float* A, int lda, const float* W, int ldw, const float* bias, const float* res, int ldr,
            int res_mod, float* C, int ldc, int64_t M, int N, int K, bool w_is_kn = false) {
    dim3 grid((N + 63) / 64, (unsigned)((M + 63) / 64));
    // w_is_kn: weight stored [K][N] and applied as x @ W (MGCN, modules.py:244-245) instead of nn.Linear's [N][K]
    k_linear<ACT><<<grid, 256, 0, st>>>(A, lda, W, w_is_kn ? 1 : ldw, w_is_kn ? ldw : 1, bias, res, ldr, res_mod, C, ldc, (int)M, N, K);
}

inline unsigned nblk(int64_t n, int t) { return (unsigned)((n + t - 1) / t); }

}  // namespace

// Workspace layout of the bring-up path (floats): [GAT activations | feat | x_out | pc | MDR activations | ... | col | tmp]
BasicLayout basic_layout(int J, int B) {
    BasicLayout L;
    const size_t T = (size_t)B * J, Vt = (size_t)B * kV;
    L.feat = T * (8 * 128 + 384 + 16 + 144 + 512);
    L.xout = L.feat + T * 128;
    L.pc = L.xout + T * 3;
    L.mdr = L.pc + T * 133;
    const size_t mdr_end = L.mdr + T * 64 * 4 + Vt * (8 * 64 + 256 + 6 + 23 + 3 + 3 + 1 + 3) + (size_t)B * 60 + 64;
    L.col = mdr_end;
    L.tmp = L.col + (size_t)B * 3 * 1293;
    L.total = L.tmp + (size_t)B * 3 * kNV;
    return L;
}

int ensure_workspace(gator_ctx* c, int B) {
    if (B <= c->cap_batch && c->ws) return GATOR_OK;
    if (c->ws) {
        GATOR_HIP_CHECK(hipDeviceSynchronize());
        GATOR_HIP_CHECK(hipFree(c->ws));
        c->ws = nullptr;
    }
    const size_t n = basic_layout(c->J, B).total;
    GATOR_HIP_CHECK(hipMalloc(&c->ws, n * sizeof(float)));
    c->ws_floats = n;
    c->cap_batch = B;
    return GATOR_OK;
}

int basic_fold_constants(gator_ctx* c, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int J = c->J, D = c->D;
    const Weights& w = c->w;
    // edge encoder: ea[d][h*J*J + ij] = Linear(J^2 -> 8J^2)(edge_input[:, :, d])      modules.py:100-101
    float *eaT = nullptr, *ea = nullptr;
    GATOR_HIP_CHECK(hipMalloc(&eaT, (size_t)D * J * J * sizeof(float)));
    GATOR_HIP_CHECK(hipMalloc(&ea, (size_t)D * kH * J * J * sizeof(float)));
    std::vector<float> h_ei((size_t)J * J * D), h_t((size_t)D * J * J);
    GATOR_HIP_CHECK(hipMemcpy(h_ei.data(), w.edge_input, h_ei.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (int ij = 0; ij < J * J; ++ij)
        for (int d = 0; d < D; ++d) h_t[(size_t)d * J * J + ij] = h_ei[(size_t)ij * D + d];
    GATOR_HIP_CHECK(hipMemcpy(eaT, h_t.data(), h_t.size() * sizeof(float), hipMemcpyHostToDevice));
    linear<0>(st, eaT, J * J, w.hp_ew, J * J, w.hp_eb, nullptr, 0, 0, ea, kH * J * J, D, kH * J * J, J * J);
    k_hop_bias<<<nblk(kH * J * J, 256), 256, 0, st>>>(w.sp, w.hp_emb, w.hp_W, ea, c->hop_bias, J, D);
    for (int b = 0; b < kDepth; ++b)
        k_fold_graph<<<1, 256, 0, st>>>(w.graph_adj, w.blk[b].gcn_adj2, w.sp, w.pos_id, w.pos_num, c->adj_diag, c->adj_off,
                                        c->mask1, c->mask2, c->pos_embed, J, b);
    k_fold_graph<<<1, 256, 0, st>>>(w.graph_adj, nullptr, w.sp, w.pos_id, w.pos_num, c->adj_diag, c->adj_off, c->mask1,
                                    c->mask2, c->pos_embed, J, -1);
    GATOR_HIP_CHECK(hipStreamSynchronize(st));
    GATOR_HIP_CHECK(hipFree(eaT));
    GATOR_HIP_CHECK(hipFree(ea));
    GATOR_HIP_CHECK(hipGetLastError());
    return GATOR_OK;
}

int basic_gat_forward(gator_ctx* c, const float* pose2d, int B, float* x_out, float* feat, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int J = c->J;
    const int64_t T = (int64_t)B * J;
    const Weights& w = c->w;
    float* p = c->ws;
    auto take = [&](size_t n) { float* r = p; p += n; return r; };
    float *x = take(T * 128), *y = take(T * 128), *att = take(T * 128), *a = take(T * 128), *h0 = take(T * 128),
          *h1 = take(T * 128), *s = take(T * 128), *u0 = take(T * 128), *qkv = take(T * 384), *u1 = take(T * 16),
          *cat = take(T * 144), *hid = take(T * 512);
    k_gat_embed<<<B, 128, 0, st>>>(pose2d, w.gl0_W, w.gl0_b, w.gn_w, w.gn_b, w.gl3_W, w.gl3_b, c->pos_embed, x, J);
    for (int i = 0; i < kDepth; ++i) {
        const GatBlockW& k = w.blk[i];
        k_layernorm<0, 0><<<nblk(T, 4), 256, 0, st>>>(x, 128, k.n1w, k.n1b, 1e-5f, y, 128, (int)T, 128);
        linear<0>(st, y, 128, k.qkv_w, 128, k.qkv_b, nullptr, 0, 0, qkv, 384, T, 384, 128);
        k_gat_attn<<<B, 192, 0, st>>>(qkv, c->hop_bias, att, J);
        linear<0>(st, att, 128, k.proj_w, 128, k.proj_b, nullptr, 0, 0, a, 128, T, 128, 128);
        // MGCN: h = y @ W[k]  (weight stored [in][out], modules.py:244-245)
        linear<0>(st, y, 128, k.gcn_W, 128, nullptr, nullptr, 0, 0, h0, 128, T, 128, 128, true);
        linear<0>(st, y, 128, k.gcn_W + 128 * 128, 128, nullptr, nullptr, 0, 0, h1, 128, T, 128, 128, true);
        k_mgcn_mix<<<B, 256, 0, st>>>(h0, h1, k.gcn_M, c->adj_diag + i * J, c->adj_off + (size_t)i * J * J, k.gcn_bias, a, s, J);
        linear<0>(st, s, 128, k.xl0_w, 128, k.xl0_b, nullptr, 0, 0, u0, 128, T, 128, 128);
        linear<0>(st, s, 128, k.xl1_w, 128, k.xl1_b, nullptr, 0, 0, u1, 16, T, 16, 128);
        k_adjmul<<<B, 256, 0, st>>>(c->mask1, u0, 128, cat, 144, J, 128);
        k_adjmul<<<B, 256, 0, st>>>(c->mask2, u1, 16, cat + 128, 144, J, 16);
        linear<0>(st, cat, 144, k.xlb_w, 144, k.xlb_b, x, 128, 0, x, 128, T, 128, 144);
        k_layernorm<0, 0><<<nblk(T, 4), 256, 0, st>>>(x, 128, k.n2w, k.n2b, 1e-5f, y, 128, (int)T, 128);
        linear<1>(st, y, 128, k.fc1_w, 128, k.fc1_b, nullptr, 0, 0, hid, 512, T, 512, 128);
        linear<0>(st, hid, 512, k.fc2_w, 512, k.fc2_b, x, 128, 0, x, 128, T, 128, 512);
    }
    k_layernorm<0, 1><<<nblk(T, 4), 256, 0, st>>>(x, 128, w.norm_w, w.norm_b, 1e-5f, feat, 128, (int)T, 128);
    linear<0>(st, feat, 128 * J, w.lifter_w, 128 * J, w.lifter_b, nullptr, 0, 0, x_out, 3 * J, B, 3 * J, 128 * J);
    GATOR_HIP_CHECK(hipGetLastError());
    return GATOR_OK;
}

int basic_build_pc(gator_ctx* c, const float* pose2d, const float* x_out, const float* feat, int B, float* pc, float* pose3d,
                   void* stream) {
    k_build_pc<<<(unsigned)((int64_t)B * c->J), 192, 0, (hipStream_t)stream>>>(pose2d, x_out, feat, pc, pose3d, (int64_t)B * c->J);
    GATOR_HIP_CHECK(hipGetLastError());
    return GATOR_OK;
}

int basic_upsample(gator_ctx* c, const float* vert431, int B, float* verts, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const Weights& w = c->w;
    const BasicLayout L = basic_layout(c->J, c->cap_batch);
    float* col = c->ws + L.col;
    float* tmp = c->ws + L.tmp;
    k_im2col3<<<nblk((int64_t)B * 3 * 1293, 256), 256, 0, st>>>(vert431, col, kV, (int64_t)B * 3 * 1293);
    linear<0>(st, col, 1293, w.up_w, 1293, w.up_b, nullptr, 0, 0, tmp, kNV, (int64_t)B * 3, kNV, 1293);
    k_up_finish<<<nblk((int64_t)B * kNV * 3, 256), 256, 0, st>>>(tmp, w.v6890, verts, (int64_t)B * kNV * 3);
    GATOR_HIP_CHECK(hipGetLastError());
    return GATOR_OK;
}

int basic_mdr_forward(gator_ctx* c, const float* pc, int B, float* verts, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    constexpr size_t kSelfAttnLds = 2 * kV * 32 * sizeof(float);   // 110 KB of the CU's 160 KB
    static bool attr_set = false;
    if (!attr_set) {
        GATOR_HIP_CHECK(hipFuncSetAttribute((const void*)k_self_attn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSelfAttnLds));
        attr_set = true;
    }
    const int J = c->J;
    const int64_t T = (int64_t)B * J, Vt = (int64_t)B * kV;
    const Weights& w = c->w;
    const BasicLayout L = basic_layout(c->J, c->cap_batch);
    float* p = c->ws + L.mdr;
    auto take = [&](size_t n) { float* r = p; p += n; return r; };
    float *jf = take(T * 64), *fzj = take(T * 64), *kj = take(T * 64), *vjv = take(T * 64);
    float *vf = take(Vt * 64), *fzv = take(Vt * 64), *q = take(Vt * 64), *o = take(Vt * 64), *qq = take(Vt * 64),
          *kk = take(Vt * 64), *vv = take(Vt * 64), *xo = take(Vt * 64), *hid = take(Vt * 256), *vin = take(Vt * 6),
          *ac = take(Vt * 23), *bm = take(Vt * 3), *bn = take(Vt * 3), *al = take(Vt), *vc = take(Vt * 3),
          *bc = take((size_t)B * 60 + 64);
    float* col = c->ws + L.col;   // shared with basic_upsample (used there only after this use is done)
    linear<0>(st, pc, 133, w.jfeat_w, 133, w.jfeat_b, w.pos_j + kE, kE, J, jf, 64, T, 64, 133);
    k_vert_in<<<nblk(Vt * 6, 256), 256, 0, st>>>(w.v431, pc, w.vj, vin, J, Vt * 6);
    linear<0>(st, vin, 6, w.vfeat_w, 6, w.vfeat_b, w.pos_v + kE, kE, kV, vf, 64, Vt, 64, 6);
    for (int li = 0; li < 3; ++li) {
        const MdrLayerW& k = w.lay[li];
        k_layernorm<0, 0><<<nblk(Vt, 4), 256, 0, st>>>(vf, 64, k.n1w, k.n1b, 1e-5f, fzv, 64, (int)Vt, 64);
        k_layernorm<0, 0><<<nblk(T, 4), 256, 0, st>>>(jf, 64, k.n1w, k.n1b, 1e-5f, fzj, 64, (int)T, 64);
        linear<0>(st, fzv, 64, k.wq, 64, nullptr, nullptr, 0, 0, q, 64, Vt, 64, 64);
        linear<0>(st, fzj, 64, k.wk, 64, nullptr, nullptr, 0, 0, kj, 64, T, 64, 64);
        linear<0>(st, fzj, 64, k.wv, 64, nullptr, nullptr, 0, 0, vjv, 64, T, 64, 64);
        k_cross_attn<<<nblk(Vt * 2, 256), 256, 0, st>>>(q, kj, vjv, o, J, Vt * 2);
        linear<0>(st, o, 64, k.proj_w, 64, k.proj_b, vf, 64, 0, vf, 64, Vt, 64, 64);
        k_layernorm<0, 0><<<nblk(Vt, 4), 256, 0, st>>>(vf, 64, k.n2w, k.n2b, 1e-5f, fzv, 64, (int)Vt, 64);
        linear<1>(st, fzv, 64, k.fc1_w, 64, k.fc1_b, nullptr, 0, 0, hid, 256, Vt, 256, 64);
        linear<0>(st, hid, 256, k.fc2_w, 256, k.fc2_b, vf, 64, 0, vf, 64, Vt, 64, 256);
        k_layernorm<1, 0><<<nblk(Vt, 4), 256, 0, st>>>(vf, 64, k.a2, k.b2, 1e-6f, vf, 64, (int)Vt, 64);
        linear<0>(st, vf, 64, k.sa_w[0], 64, k.sa_b[0], nullptr, 0, 0, qq, 64, Vt, 64, 64);
        linear<0>(st, vf, 64, k.sa_w[1], 64, k.sa_b[1], nullptr, 0, 0, kk, 64, Vt, 64, 64);
        linear<0>(st, vf, 64, k.sa_w[2], 64, k.sa_b[2], nullptr, 0, 0, vv, 64, Vt, 64, 64);
        k_self_attn<<<B * 2, 448, kSelfAttnLds, st>>>(qq, kk, vv, xo);
        linear<0>(st, xo, 64, k.sa_w[3], 64, k.sa_b[3], vf, 64, 0, vf, 64, Vt, 64, 64);
    }
    c->set_tap(TAP_MDR_LBF2, vf, Vt * 64);
    linear<0>(st, vf, 64, w.motion_w, 64, w.motion_b, nullptr, 0, 0, ac, 23, Vt, 23, 64);
    linear<0>(st, vf, 64, w.biasl_w, 64, w.biasl_b, nullptr, 0, 0, bm, 3, Vt, 3, 64);
    k_head_norm<<<nblk(Vt, 256), 256, 0, st>>>(bm, w.bn_w, w.bn_b, w.bn_mean, w.bn_var, c->alpha, bn, Vt);
    k_im2col3<<<nblk((int64_t)B * 3 * 1293, 256), 256, 0, st>>>(bn, col, kV, (int64_t)B * 3 * 1293);
    linear<0>(st, col, 1293, w.bconv_w, 1293, w.bconv_b, nullptr, 0, 0, bc, 20, (int64_t)B * 3, 20, 1293);
    if (c->alpha) linear<0>(st, vf, 64, w.scale_w, 64, w.scale_b, nullptr, 0, 0, al, 1, Vt, 1, 64);
    k_head_mix<<<nblk(Vt, 256), 256, 0, st>>>(ac, bc, c->alpha ? al : nullptr, vc, Vt);
    c->set_tap(TAP_VERT431, vc, Vt * 3);
    GATOR_HIP_CHECK(hipGetLastError());
    return basic_upsample(c, vc, B, verts, stream);
}

}  // namespace gator
