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
__global__ void k_preprocess(const float* __restrict__ in, int B, int jin, int comps, int add_pn, float* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int jout = jin + (add_pn ? 2 : 0);
    const float* p = in + (size_t)b * jin * comps;
    double x[32], y[32];
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
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) { G[i][j] = H[i][j]; V[i][j] = i == j ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double a = 0.0, bq = 0.0, c = 0.0;
                for (int i = 0; i < 3; ++i) { a += G[i][p] * G[i][p]; bq += G[i][q] * G[i][q]; c += G[i][p] * G[i][q]; }
                off = fmax(off, fabs(c) / (sqrt(a * bq) + 1e-300));
                if (fabs(c) <= 1e-300) continue;
                const double zeta = (bq - a) / (2.0 * c);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
                for (int i = 0; i < 3; ++i) {
                    const double gp = G[i][p], gq = G[i][q];
                    G[i][p] = cs * gp - sn * gq;  G[i][q] = sn * gp + cs * gq;
                    const double vp = V[i][p], vq = V[i][q];
                    V[i][p] = cs * vp - sn * vq;  V[i][q] = sn * vp + cs * vq;
                }
            }
        if (off < 1e-15) break;
    }
    for (int j = 0; j < 3; ++j) s[j] = sqrt(G[0][j] * G[0][j] + G[1][j] * G[1][j] + G[2][j] * G[2][j]);
    // sort descending (the reflection fix below must hit the SMALLEST singular value, as numpy's s[-1] does)
    for (int i = 0; i < 2; ++i)
        for (int j = i + 1; j < 3; ++j)
            if (s[j] > s[i]) {
                const double ts = s[i]; s[i] = s[j]; s[j] = ts;
                for (int r = 0; r < 3; ++r) {
                    const double tg = G[r][i]; G[r][i] = G[r][j]; G[r][j] = tg;
                    const double tv = V[r][i]; V[r][i] = V[r][j]; V[r][j] = tv;
                }
            }
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) U[i][j] = s[j] > 1e-300 ? G[i][j] / s[j] : 0.0;
    if (s[2] <= 1e-14 * s[0]) {      // rank-deficient covariance (coplanar points): complete U with the cross product
        U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
        U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
        U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
    }
}

// rigid_transform_3D + rigid_align (lib/coord_utils.py:127-149): H = (A-ca)^T (B-cb) / n = U S V^T; R = V U^T, with the reflection
// fix (det R < 0: s3 := -s3, V[:,2] := -V[:,2]); c = sum(s) / sum_axis var(A); t = -(cR) ca + cb; out = cR A + t.
__global__ void k_rigid_align(const float* __restrict__ A, const float* __restrict__ Bt, int nb, int n, float* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    const float* a = A + (size_t)b * n * 3;
    const float* t = Bt + (size_t)b * n * 3;
    double ca[3] = {0, 0, 0}, cb[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) { ca[k] += a[i * 3 + k]; cb[k] += t[i * 3 + k]; }
    for (int k = 0; k < 3; ++k) { ca[k] /= n; cb[k] /= n; }
    double H[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, var = 0.0;
    for (int i = 0; i < n; ++i) {
        double da[3], db[3];
        for (int k = 0; k < 3; ++k) { da[k] = a[i * 3 + k] - ca[k]; db[k] = t[i * 3 + k] - cb[k]; var += da[k] * da[k]; }
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) H[r][c] += da[r] * db[c];
    }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) H[r][c] /= n;
    var /= n;
    double U[3][3], s[3], V[3][3], R[3][3];
    svd3(H, U, s, V);
    auto mkR = [&]() {
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) R[r][c] = V[r][0] * U[c][0] + V[r][1] * U[c][1] + V[r][2] * U[c][2];
    };
    mkR();
    const double det = R[0][0] * (R[1][1] * R[2][2] - R[1][2] * R[2][1]) - R[0][1] * (R[1][0] * R[2][2] - R[1][2] * R[2][0]) +
                       R[0][2] * (R[1][0] * R[2][1] - R[1][1] * R[2][0]);
    if (det < 0.0) {
        s[2] = -s[2];
        for (int r = 0; r < 3; ++r) V[r][2] = -V[r][2];
        mkR();
    }
    const double c = (s[0] + s[1] + s[2]) / var;
    double tr[3];
    for (int r = 0; r < 3; ++r) tr[r] = cb[r] - c * (R[r][0] * ca[0] + R[r][1] * ca[1] + R[r][2] * ca[2]);
    float* o = out + (size_t)b * n * 3;
    for (int i = 0; i < n; ++i)
        for (int r = 0; r < 3; ++r)
            o[i * 3 + r] = (float)(c * (R[r][0] * a[i * 3] + R[r][1] * a[i * 3 + 1] + R[r][2] * a[i * 3 + 2]) + tr[r]);
}

}  // namespace
}  // namespace gator

using namespace gator;

extern "C" int gator_preprocess_pose2d_f32(const float* joints, int32_t batch, int32_t num_joint_in, int32_t comps,
                                           int32_t add_pelvis_neck, float* pose2d, void* stream) {
    if (!joints || !pose2d || batch <= 0 || num_joint_in <= 0 || num_joint_in + (add_pelvis_neck ? 2 : 0) > 32 || comps < 2)
        return fail(GATOR_EINVAL, "gator_preprocess_pose2d_f32: bad arguments");
    if (add_pelvis_neck && num_joint_in < 13) return fail(GATOR_EINVAL, "gator_preprocess_pose2d_f32: pelvis/neck need the COCO joint order (>= 13 joints)");
    k_preprocess<<<(batch + 127) / 128, 128, 0, (hipStream_t)stream>>>(joints, batch, num_joint_in, comps, add_pelvis_neck, pose2d);
    GATOR_HIP_CHECK(hipGetLastError());
    return GATOR_OK;
}

extern "C" int gator_rigid_align_f32(const float* a, const float* b, int32_t batch, int32_t n_points, float* aligned, void* stream) {
    if (!a || !b || !aligned || batch <= 0 || n_points < 3) return fail(GATOR_EINVAL, "gator_rigid_align_f32: bad arguments");
    k_rigid_align<<<(batch + 63) / 64, 64, 0, (hipStream_t)stream>>>(a, b, batch, n_points, aligned);
    GATOR_HIP_CHECK(hipGetLastError());
    return GATOR_OK;
}
