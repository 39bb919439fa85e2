// "Next" rows of the scope table: the caller-side pieces either side of GATOR.forward, on the device.
//   gator_preprocess_pose2d_f32 : raw 2D joints -> model input (demo/run.py:103-121,127-134; data/PW3D/dataset.py:168-183,241-250)
//   gator_rigid_align_f32       : per-sample similarity (Procrustes) alignment (lib/coord_utils.py:127-149), the PA-MPJPE kernel
// Both are tiny per-sample problems (<= 19 joints, one 3x3 SVD): one thread per sample, fp64 arithmetic, fp32 in / out.
#include <hip/hip_runtime.h>

#include "internal.h"

namespace gator {
namespace {

// With rot = 0 and flip = 0 (every evaluation path) the reference's bbox -> affine -> /[W,H] chain is a per-axis positive scale +
// shift, which cancels in the per-axis standardisation that follows it (checked on the demo input to 5e-8):
// out = (xy - mean) / std over the joints of the sample, population std (np.std).  COCO inputs first get pelvis = (L_Hip+R_Hip)/2
// and neck = (L_Shoulder+R_Shoulder)/2 appended (joints 11,12 and 5,6).
// element j of this thread's column of a [n][threads] LDS image
template <class T> struct ColT {
    T* base; int tid, stride;
    __device__ __forceinline__ T& operator[](int j) const { return base[j * stride + tid]; }
};
typedef ColT<double> Col;
typedef ColT<float> ColF;
__global__ void k_preprocess(const float* __restrict__ in, int B, int jin, int comps, int add_pn, float* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int jout = jin + (add_pn ? 2 : 0);
    const float* p = in + (size_t)b * jin * comps;
    // the joints of a sample as one COLUMN of a [32][blockDim] LDS image (run-time indexed private arrays would live in scratch)
    __shared__ double xs_[32 * 64], ys_[32 * 64];
    const Col x{xs_, (int)threadIdx.x, (int)blockDim.x}, y{ys_, (int)threadIdx.x, (int)blockDim.x};
    for (int j = 0; j < jin; ++j) { x[j] = p[j * comps]; y[j] = p[j * comps + 1]; }
    if (add_pn) {
        x[jin] = (x[11] + x[12]) * 0.5;     y[jin] = (y[11] + y[12]) * 0.5;          // pelvis
        x[jin + 1] = (x[5] + x[6]) * 0.5;   y[jin + 1] = (y[5] + y[6]) * 0.5;        // neck
    }
    double mx = 0.0, my = 0.0;
    for (int j = 0; j < jout; ++j) { mx += x[j]; my += y[j]; }
    mx /= jout; my /= jout;
    double vx = 0.0, vy = 0.0;
    for (int j = 0; j < jout; ++j) { vx += (x[j] - mx) * (x[j] - mx); vy += (y[j] - my) * (y[j] - my); }
    const double sx = sqrt(vx / jout), sy = sqrt(vy / jout);      // a degenerate axis gives inf/nan exactly as numpy does
    float* o = out + (size_t)b * jout * 2;
    for (int j = 0; j < jout; ++j) { o[j * 2] = (float)((x[j] - mx) / sx); o[j * 2 + 1] = (float)((y[j] - my) / sy); }
}

// 3x3 SVD by one-sided Jacobi (Hestenes) in fp64: G = H V is driven to orthogonal columns; s_i = |G_i|, U_i = G_i / s_i.
__device__ void svd3(const double H[3][3], double U[3][3], double s[3], double V[3][3]) {
    double G[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) { G[i][j] = H[i][j]; V[i][j] = i == j ? 1.0 : 0.0; }
#pragma unroll 1
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = p + 1; q < 3; ++q) {
                double a = 0.0, bq = 0.0, c = 0.0;
#pragma unroll
                for (int i = 0; i < 3; ++i) { a += G[i][p] * G[i][p]; bq += G[i][q] * G[i][q]; c += G[i][p] * G[i][q]; }
                off = fmax(off, fabs(c) / (sqrt(a * bq) + 1e-300));
                if (fabs(c) <= 1e-300) continue;
                const double zeta = (bq - a) / (2.0 * c);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const double gp = G[i][p], gq = G[i][q];
                    G[i][p] = cs * gp - sn * gq;  G[i][q] = sn * gp + cs * gq;
                    const double vp = V[i][p], vq = V[i][q];
                    V[i][p] = cs * vp - sn * vq;  V[i][q] = sn * vp + cs * vq;
                }
            }
        if (off < 1e-15) break;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) s[j] = sqrt(G[0][j] * G[0][j] + G[1][j] * G[1][j] + G[2][j] * G[2][j]);
    // sort descending (the reflection fix below must hit the SMALLEST singular value, as numpy's s[-1] does)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = i + 1; j < 3; ++j)
            if (s[j] > s[i]) {
                const double ts = s[i]; s[i] = s[j]; s[j] = ts;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const double tg = G[r][i]; G[r][i] = G[r][j]; G[r][j] = tg;
                    const double tv = V[r][i]; V[r][i] = V[r][j]; V[r][j] = tv;
                }
            }
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i) U[i][j] = s[j] > 1e-300 ? G[i][j] / s[j] : 0.0;
    if (s[2] <= 1e-14 * s[0]) {      // rank-deficient covariance (coplanar points): complete U with the cross product
        U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
        U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
        U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
    }
}

// rigid_transform_3D + rigid_align (lib/coord_utils.py:127-149): H = (A-ca)^T (B-cb) / n = U S V^T; R = V U^T, with the reflection
// fix (det R < 0: s3 := -s3, V[:,2] := -V[:,2]); c = sum(s) / sum_axis var(A); t = -(cR) ca + cb; out = cR A + t.
// similarity (Procrustes) fit of n points a onto t, fp64: aligned = c R a + tr   (lib/coord_utils.py:127-142)
// The point sets live in LDS, one column per thread ([point][axis][thread]: conflict-free, no per-thread private arrays)
struct Pts {
    double* p; int tid, nt;
    __device__ double& operator()(int i, int k) const { return p[(size_t)(i * 3 + k) * nt + tid]; }
};
__device__ void rigid_fit(const Pts& a, const Pts& t, int n, double& c, double (&R)[3][3], double (&tr)[3]) {
    double ca[3] = {0, 0, 0}, cb[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) { ca[k] += a(i, k); cb[k] += t(i, k); }
#pragma unroll
    for (int k = 0; k < 3; ++k) { ca[k] /= n; cb[k] /= n; }
    double H[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, var = 0.0;
    for (int i = 0; i < n; ++i) {
        double da[3], db[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { da[k] = a(i, k) - ca[k]; db[k] = t(i, k) - cb[k]; var += da[k] * da[k]; }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) H[r][cc] += da[r] * db[cc];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) H[r][cc] /= n;
    var /= n;
    double U[3][3], s[3], V[3][3];
    svd3(H, U, s, V);
    auto mkR = [&]() {
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) R[r][cc] = V[r][0] * U[cc][0] + V[r][1] * U[cc][1] + V[r][2] * U[cc][2];
    };
    mkR();
    const double det = R[0][0] * (R[1][1] * R[2][2] - R[1][2] * R[2][1]) - R[0][1] * (R[1][0] * R[2][2] - R[1][2] * R[2][0]) +
                       R[0][2] * (R[1][0] * R[2][1] - R[1][1] * R[2][0]);
    if (det < 0.0) {
        s[2] = -s[2];
#pragma unroll
        for (int r = 0; r < 3; ++r) V[r][2] = -V[r][2];
        mkR();
    }
    c = (s[0] + s[1] + s[2]) / var;
#pragma unroll
    for (int r = 0; r < 3; ++r) tr[r] = cb[r] - c * (R[r][0] * ca[0] + R[r][1] * ca[1] + R[r][2] * ca[2]);
}

constexpr int kMaxPts = 32;
constexpr size_t kPtsLds = (size_t)2 * kMaxPts * 3 * 64 * sizeof(double);      // 96 KB: the two point sets of a 64-thread block
__global__ void k_rigid_align(const float* __restrict__ A, const float* __restrict__ Bt, int nb, int n, float* __restrict__ out) {
    extern __shared__ double pts_lds[];                  // [2][kMaxPts][3][blockDim.x]
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    const float* a = A + (size_t)b * n * 3;
    const float* t = Bt + (size_t)b * n * 3;
    const Pts pa{pts_lds, (int)threadIdx.x, (int)blockDim.x}, pt{pts_lds + (size_t)kMaxPts * 3 * blockDim.x, (int)threadIdx.x, (int)blockDim.x};
    for (int i = 0; i < n; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) { pa(i, k) = a[i * 3 + k]; pt(i, k) = t[i * 3 + k]; }
    double c, R[3][3], tr[3];
    rigid_fit(pa, pt, n, c, R, tr);
    float* o = out + (size_t)b * n * 3;
    for (int i = 0; i < n; ++i)
#pragma unroll
        for (int r = 0; r < 3; ++r)
            o[i * 3 + r] = (float)(c * (R[r][0] * pa(i, 0) + R[r][1] * pa(i, 1) + R[r][2] * pa(i, 2)) + tr[r]);
}

// Per-sample evaluation errors in one launch (data/PW3D/dataset.py:273-286, 337-375): err[b][0] = mean distance of the root-aligned
// evaluation joints (MPJPE), err[b][1] = the same after the similarity alignment of the evaluation joints (PA-MPJPE).
__global__ void k_joint_errors(const float* __restrict__ P, const float* __restrict__ T, int nb, int nj, const int32_t* __restrict__ idx, int ne,
                               int root, float scale, float* __restrict__ err) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    extern __shared__ double pts_lds[];                  // [2][kMaxPts][3][blockDim.x]
    const float* p = P + (size_t)b * nj * 3;
    const float* t = T + (size_t)b * nj * 3;
    const Pts pa{pts_lds, (int)threadIdx.x, (int)blockDim.x}, pt{pts_lds + (size_t)kMaxPts * 3 * blockDim.x, (int)threadIdx.x, (int)blockDim.x};
    double e0 = 0.0;
    for (int i = 0; i < ne; ++i) {
        const int j = idx ? idx[i] : i;
        double d2 = 0.0;
        for (int k = 0; k < 3; ++k) {
            // float32 arithmetic where the reference has it: mesh * 1000, pred - pred[root], target - target[root], their difference
            const float pj = p[j * 3 + k] * scale, pr = p[root * 3 + k] * scale;
            const float dp = pj - pr, dt = t[j * 3 + k] - t[root * 3 + k];
            pa(i, k) = dp; pt(i, k) = dt;
            const float d = dp - dt;
            d2 += (double)d * (double)d;
        }
        e0 += sqrt(d2);
    }
    double c, R[3][3], tr[3];
    rigid_fit(pa, pt, ne, c, R, tr);
    double e1 = 0.0;
    for (int i = 0; i < ne; ++i) {
        double d2 = 0.0;
        for (int r = 0; r < 3; ++r) {
            const double al = (double)(float)(c * (R[r][0] * pa(i, 0) + R[r][1] * pa(i, 1) + R[r][2] * pa(i, 2)) + tr[r]);
            d2 += (al - pt(i, r)) * (al - pt(i, r));
        }
        e1 += sqrt(d2);
    }
    err[b * 2] = (float)(e0 / ne);
    err[b * 2 + 1] = (float)(e1 / ne);
}

}  // namespace
}  // namespace gator

using namespace gator;

namespace gator {
namespace {
// The GENERAL input chain (data/PW3D/dataset.py:236-250): tight bbox (lib/coord_utils.py:21-39) -> process_bbox (:42-66, which
// REJECTS a box narrower or lower than one pixel: valid = 0, the datasets drop such samples) -> get_affine_transform with rotation
// (lib/aug_utils.py:140-173: three float32 point pairs, solved as cv2.getAffineTransform does) -> affine per joint, optional
// flip_2d_joint (:31-38) -> float32 -> /[W,H] -> per-axis standardisation.  One thread per sample; float32 roundings where the
// reference has them (bbox, centre/scale, the point pairs, the transformed joints, the division), fp64 in between.
__global__ void k_preprocess_chain(const float* __restrict__ in, int B, int jin, int comps, int add_pn, const float* __restrict__ rot_deg,
                                   const int32_t* __restrict__ flip, const int32_t* __restrict__ pairs, int n_pairs, int res_w, int res_h,
                                   float* __restrict__ out, int32_t* __restrict__ valid) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int J = jin + (add_pn ? 2 : 0);
    const float* p = in + (size_t)b * jin * comps;
    __shared__ double xs_[32 * 64], ys_[32 * 64];          // one column per thread (48 KB with the float32 images below: 64-thread blocks)
    __shared__ float fxs_[32 * 64], fys_[32 * 64];
    const Col x{xs_, (int)threadIdx.x, (int)blockDim.x}, y{ys_, (int)threadIdx.x, (int)blockDim.x};
    const ColF fxn{fxs_, (int)threadIdx.x, (int)blockDim.x}, fyn{fys_, (int)threadIdx.x, (int)blockDim.x};
    for (int j = 0; j < jin; ++j) { x[j] = p[j * comps]; y[j] = p[j * comps + 1]; }
    if (add_pn) {
        x[jin] = (x[11] + x[12]) * 0.5;     y[jin] = (y[11] + y[12]) * 0.5;
        x[jin + 1] = (x[5] + x[6]) * 0.5;   y[jin + 1] = (y[5] + y[6]) * 0.5;
    }
    float* o = out + (size_t)b * J * 2;
    double xmin = x[0], xmax = x[0], ymin = y[0], ymax = y[0];
    for (int j = 1; j < J; ++j) { xmin = fmin(xmin, x[j]); xmax = fmax(xmax, x[j]); ymin = fmin(ymin, y[j]); ymax = fmax(ymax, y[j]); }
    // get_bbox: centre +- half extent, cast to float32
    const double xc = (xmin + xmax) / 2., bw0 = xmax - xmin, yc = (ymin + ymax) / 2., bh0 = ymax - ymin;
    const double bxmin = xc - 0.5 * bw0, bxmax = xc + 0.5 * bw0, bymin = yc - 0.5 * bh0, bymax = yc + 0.5 * bh0;
    const float bx = (float)bxmin, by = (float)bymin, bw = (float)(bxmax - bxmin), bh = (float)(bymax - bymin);
    // process_bbox (float32 arithmetic on the float32 box, as numpy does)
    const float x2 = bx + (bw - 1.f), y2 = by + (bh - 1.f);
    const bool ok = (bw * bh > 0.f) && x2 >= bx && y2 >= by;
    if (valid) valid[b] = ok ? 1 : 0;
    if (!ok) {
        for (int j = 0; j < 2 * J; ++j) o[j] = 0.f;
        return;
    }
    float w = x2 - bx, h = y2 - by;
    const float cx = bx + w / 2.f, cy = by + h / 2.f;
    const double ar = (double)res_w / (double)res_h;       // cfg.MODEL.input_shape[1] / [0]; python float (fp64) against float32 values
    if ((double)w > ar * (double)h) h = (float)((double)w / ar);
    else if ((double)w < ar * (double)h) w = (float)((double)h * ar);
    const float fx = cx - w / 2.f, fy = cy - h / 2.f;
    // get_center_scale + get_affine_transform: float32 point pairs
    const float cen0 = fx + w * 0.5f, cen1 = fy + h * 0.5f;
    const double rr = M_PI * (rot_deg ? (double)rot_deg[b] : 0.0) / 180.0, sn = sin(rr), cs = cos(rr);
    const double p1 = (double)(w * -0.5f);
    const double sd0 = 0.0 * cs - p1 * sn, sd1 = 0.0 * sn + p1 * cs;                      // get_dir([0, -w/2], rot)
    float src[3][2], dst[3][2];
    src[0][0] = cen0; src[0][1] = cen1;
    src[1][0] = (float)((double)cen0 + sd0); src[1][1] = (float)((double)cen1 + sd1);
    dst[0][0] = res_w * 0.5f; dst[0][1] = res_h * 0.5f;
    dst[1][0] = (float)((double)(res_w * 0.5) + 0.0); dst[1][1] = (float)((double)(res_h * 0.5) + (double)(res_w * -0.5f));
    for (int q = 0; q < 2; ++q) {
        float (*m)[2] = q ? dst : src;
        const float d0 = m[0][0] - m[1][0], d1 = m[0][1] - m[1][1];
        m[2][0] = m[1][0] + (-d1); m[2][1] = m[1][1] + d0;                                 // get_3rd_point
    }
    // solve [sx sy 1] T^T = [dx dy] for the 2x3 matrix (Cramer, fp64)
    const double a0 = src[0][0], b0 = src[0][1], a1 = src[1][0], b1 = src[1][1], a2 = src[2][0], b2 = src[2][1];
    const double det = a0 * (b1 - b2) - b0 * (a1 - a2) + (a1 * b2 - a2 * b1);
    double T[2][3];
    for (int r = 0; r < 2; ++r) {
        const double d0 = dst[0][r], d1 = dst[1][r], d2 = dst[2][r];
        T[r][0] = (d0 * (b1 - b2) - b0 * (d1 - d2) + (d1 * b2 - d2 * b1)) / det;
        T[r][1] = (a0 * (d1 - d2) - d0 * (a1 - a2) + (a1 * d2 - a2 * d1)) / det;
        T[r][2] = (a0 * (b1 * d2 - b2 * d1) - b0 * (a1 * d2 - a2 * d1) + d0 * (a1 * b2 - a2 * b1)) / det;
    }
    for (int j = 0; j < J; ++j) {
        const double nx = T[0][0] * x[j] + T[0][1] * y[j] + T[0][2], ny = T[1][0] * x[j] + T[1][1] * y[j] + T[1][2];
        x[j] = nx; y[j] = ny;
    }
    if (flip && flip[b]) {
        for (int j = 0; j < J; ++j) x[j] = res_w - x[j] - 1;
        for (int q = 0; q < n_pairs; ++q) {
            const int u = pairs[2 * q], v = pairs[2 * q + 1];
            if (u < J && v < J) { const double tx = x[u], ty = y[u]; x[u] = x[v]; y[u] = y[v]; x[v] = tx; y[v] = ty; }
        }
    }
    for (int j = 0; j < J; ++j) { fxn[j] = (float)x[j] / (float)res_w; fyn[j] = (float)y[j] / (float)res_h; }   // astype(float32); /= [W,H]
    double mx = 0.0, my = 0.0;
    for (int j = 0; j < J; ++j) { mx += fxn[j]; my += fyn[j]; }
    mx /= J; my /= J;
    double vx = 0.0, vy = 0.0;
    for (int j = 0; j < J; ++j) { vx += (fxn[j] - mx) * (fxn[j] - mx); vy += (fyn[j] - my) * (fyn[j] - my); }
    const double sx = sqrt(vx / J), sy = sqrt(vy / J);
    for (int j = 0; j < J; ++j) { o[j * 2] = (float)((fxn[j] - mx) / sx); o[j * 2 + 1] = (float)((fyn[j] - my) / sy); }
}
}  // namespace
}  // namespace gator

extern "C" int gator_preprocess_chain_f32(const float* joints, int32_t batch, int32_t num_joint_in, int32_t comps, int32_t add_pelvis_neck,
                                          const float* rot_deg, const int32_t* flip, const int32_t* flip_pairs, int32_t n_pairs,
                                          int32_t res_w, int32_t res_h, float* pose2d, int32_t* valid, void* stream) {
    using namespace gator;
    if (!joints || !pose2d || batch <= 0 || num_joint_in <= 0 || comps < 2 || res_w <= 0 || res_h <= 0 || n_pairs < 0 || (n_pairs > 0 && !flip_pairs))
        return fail(GATOR_EINVAL, "gator_preprocess_chain_f32: bad arguments");
    if (num_joint_in + (add_pelvis_neck ? 2 : 0) > 32) return fail(GATOR_EINVAL, "gator_preprocess_chain_f32: at most 32 joints");
    if (add_pelvis_neck && num_joint_in < 13) return fail(GATOR_EINVAL, "gator_preprocess_chain_f32: pelvis/neck need the COCO joint order (>= 13 joints)");
    k_preprocess_chain<<<(batch + 63) / 64, 64, 0, (hipStream_t)stream>>>(joints, batch, num_joint_in, comps, add_pelvis_neck ? 1 : 0, rot_deg, flip,
                                                                       flip_pairs, n_pairs, res_w, res_h, pose2d, valid);
    GATOR_HIP_CHECK(hipGetLastError());
    return GATOR_OK;
}

extern "C" int gator_preprocess_pose2d_f32(const float* joints, int32_t batch, int32_t num_joint_in, int32_t comps,
                                           int32_t add_pelvis_neck, float* pose2d, void* stream) {
    if (!joints || !pose2d || batch <= 0 || num_joint_in <= 0 || num_joint_in + (add_pelvis_neck ? 2 : 0) > 32 || comps < 2)
        return fail(GATOR_EINVAL, "gator_preprocess_pose2d_f32: bad arguments");
    if (add_pelvis_neck && num_joint_in < 13) return fail(GATOR_EINVAL, "gator_preprocess_pose2d_f32: pelvis/neck need the COCO joint order (>= 13 joints)");
    k_preprocess<<<(batch + 63) / 64, 64, 0, (hipStream_t)stream>>>(joints, batch, num_joint_in, comps, add_pelvis_neck, pose2d);      // 64-thread blocks: the kernel's LDS images have 64 columns
    GATOR_HIP_CHECK(hipGetLastError());
    return GATOR_OK;
}

// 96 KB of dynamic LDS > the 64 KB default: raised once PER DEVICE (the attribute belongs to the device's copy of the function; a
// process that evaluates on a second GPU, or whose first call failed transiently, must not inherit a cached answer)
static int raise_pts_lds(const void* fn, bool (&done)[64]) {
    int dev = 0;
    GATOR_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !done[dev]) {
        GATOR_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gator::kPtsLds));
        if (dev >= 0 && dev < 64) done[dev] = true;
    }
    return GATOR_OK;
}

extern "C" int gator_joint_errors_f32(const float* pred_joints, const float* target_joints, int32_t batch, int32_t n_joint,
                                      const int32_t* eval_joints, int32_t n_eval, int32_t root, float pred_scale, float* errors, void* stream) {
    using namespace gator;
    if (!pred_joints || !target_joints || !errors || batch <= 0 || n_joint <= 0 || root < 0 || root >= n_joint)
        return fail(GATOR_EINVAL, "gator_joint_errors_f32: bad arguments");
    const int ne = eval_joints ? n_eval : n_joint;
    if (ne < 3 || ne > 32) return fail(GATOR_EINVAL, "gator_joint_errors_f32: 3..32 evaluation joints");
    static bool lds_raised_[64] = {};
    if (int rc = raise_pts_lds((const void*)k_joint_errors, lds_raised_)) return rc;
    k_joint_errors<<<(batch + 63) / 64, 64, kPtsLds, (hipStream_t)stream>>>(pred_joints, target_joints, batch, n_joint, eval_joints, ne, root, pred_scale, errors);
    GATOR_HIP_CHECK(hipGetLastError());
    return GATOR_OK;
}

extern "C" int gator_rigid_align_f32(const float* a, const float* b, int32_t batch, int32_t n_points, float* aligned, void* stream) {
    if (!a || !b || !aligned || batch <= 0 || n_points < 3 || n_points > 32) return fail(GATOR_EINVAL, "gator_rigid_align_f32: bad arguments (3..32 points)");
    static bool lds_raised_[64] = {};
    if (int rc = raise_pts_lds((const void*)gator::k_rigid_align, lds_raised_)) return rc;
    k_rigid_align<<<(batch + 63) / 64, 64, kPtsLds, (hipStream_t)stream>>>(a, b, batch, n_points, aligned);
    GATOR_HIP_CHECK(hipGetLastError());
    return GATOR_OK;
}

// ---- measurement helper: the device-side traffic of this rank's share of an all-gather (include/gator_hip.h) ------------------------------
namespace gator {
typedef float tr_f4 __attribute__((ext_vector_type(4)));
// persistent-style: every workgroup walks the source in 4 KiB steps, strided by the grid, once per copy; 16 B per lane, non-temporal stores
// (the data is not read again on this device, like a receive buffer)
__global__ __launch_bounds__(256) void k_gather_traffic(const tr_f4* __restrict__ src, tr_f4* __restrict__ dst, long long n16, int copies) {
    for (int c = 0; c < copies; ++c) {
        tr_f4* d = dst + (size_t)c * n16;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256)
            __builtin_nontemporal_store(src[i], d + i);
    }
}
}  // namespace gator

extern "C" int gator_emulate_gather_traffic(const void* src, void* dst, int64_t bytes, int32_t copies, int32_t n_workgroups, void* stream) {
    using namespace gator;
    if (!src || !dst || bytes < 16 || (bytes & 15) || copies <= 0 || n_workgroups <= 0) return fail(GATOR_EINVAL, "gator_emulate_gather_traffic: bad arguments (bytes: a positive multiple of 16)");
    k_gather_traffic<<<n_workgroups, 256, 0, (hipStream_t)stream>>>((const tr_f4*)src, (tr_f4*)dst, (long long)(bytes / 16), copies);
    GATOR_HIP_CHECK(hipGetLastError());
    return GATOR_OK;
}
