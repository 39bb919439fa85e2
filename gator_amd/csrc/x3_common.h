// Split-precision ("x3") operands: fp32-accurate products on the bf16 MFMA (v_mfma_f32_32x32x16_bf16, 32 cycles for a 16-deep
// k-step against 8 x 64 cycles on the fp32-input MFMA).
//
// Every fp32 value is split EXACTLY into three bf16 planes, x = hi + mid + lo (8+8+8 significand bits; each remainder is
// exactly representable in fp32, so nothing is lost).  A product a*w is the sum of nine exact partial products; the six
// largest -- hi*hi | hi*mid, mid*hi | mid*mid, hi*lo, lo*hi -- go through the MFMA (fp32 accumulate), the three dropped
// ones are below 2^-24 of the product, i.e. under the rounding of the fp32 accumulation itself.  A 32x32 (K=32) tile
// product is 12 bf16 MFMAs (384 cycles) instead of 16 fp32-input MFMAs (1024 cycles).
//
// Operand tile ("X3 tile", 6 KiB = kTileX3 floats): [plane 3][k-step 2][lane 64][8 bf16].  Element j of lane l (i = l&31,
// h = l>>5) of k-step s is k index 16s + 8(j>>2) + 4h + (j&3): exactly registers 8s..8s+7 of a fused_common.h T/C-layout
// block, so an accumulator tile becomes the next product's operand by converting its registers pairwise, with no lane
// movement, and a packed fp32 weight tile [g][lane][j] maps to it by g = 2s + (j>>2).
#pragma once
#include "fused_common.h"

namespace gator {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define GATOR_MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

constexpr int kTileX3 = 1536;       // floats-equivalent of one X3 operand tile

struct X3 { bf16x8 p[3][2]; };      // [plane hi/mid/lo][k-step]: 24 VGPRs

__device__ __forceinline__ X3 x3_split(const f32x16& v) {
    X3 o;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = v[8 * s + j];
            const __bf16 h = (__bf16)x;
            const float r = x - (float)h;
            const __bf16 m = (__bf16)r;
            o.p[0][s][j] = h;
            o.p[1][s][j] = m;
            o.p[2][s][j] = (__bf16)(r - (float)m);
        }
    return o;
}

__device__ __forceinline__ X3 x3_load(const float* __restrict__ tile, int lane) {
    const bf16x8* q = reinterpret_cast<const bf16x8*>(tile) + lane;
    X3 o;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int s = 0; s < 2; ++s) o.p[pl][s] = q[(pl * 2 + s) * 64];
    return o;
}
__device__ __forceinline__ void x3_store(float* __restrict__ tile, int lane, const X3& v) {
    bf16x8* q = reinterpret_cast<bf16x8*>(tile) + lane;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int s = 0; s < 2; ++s) q[(pl * 2 + s) * 64] = v.p[pl][s];
}

// acc += A . B over the tile's 32-deep k (rows of the result = A's lane index, columns = B's lane index); small terms first
__device__ __forceinline__ f32x16 x3_mma(const X3& A, const X3& B, f32x16 acc) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        acc = GATOR_MFMA_BF16(A.p[2][s], B.p[0][s], acc);      // lo*hi
        acc = GATOR_MFMA_BF16(A.p[0][s], B.p[2][s], acc);      // hi*lo
        acc = GATOR_MFMA_BF16(A.p[1][s], B.p[1][s], acc);      // mid*mid
        acc = GATOR_MFMA_BF16(A.p[1][s], B.p[0][s], acc);      // mid*hi
        acc = GATOR_MFMA_BF16(A.p[0][s], B.p[1][s], acc);      // hi*mid
        acc = GATOR_MFMA_BF16(A.p[0][s], B.p[0][s], acc);      // hi*hi
    }
    return acc;
}

// ---- two-plane fp16 operands ("X2"): the 431x431 self-attention of the MDR layers (Q, K, V, P), their cross-attention over the J
// joint tokens (q, K, V, P; round 3) and the vertex regressor (upsample_x2.hip) -- the places where a result averages many terms ----
// x = hi + lo with hi = fp16(x), lo = fp16(x - hi): 22 significant bits, three partial products (hi*hi, hi*lo, lo*hi) per k-step
// instead of six.  Unlike the exact three-way bf16 split this ROUNDS the operands (2^-23 relative), which is only acceptable
// where the result is an average over many terms: measured in the fp64 oracle (two real fp16 planes, all other arithmetic exact)
// the vertices move by 8e-6 mm max / 1.4e-6 mm rms -- 1 % of the path's own fp32 noise -- whereas the same treatment of BOTH operands
// of the token-wise linears costs 4e-4 mm (as much as the whole budget): those keep their WEIGHTS exact and round only the
// activations (the "4-product" form further down, 1.3e-4 mm; GATOR_MDR_X3=1 / GATOR_GAT8_H4=0 keep the exact six).  Operands are pre-scaled
// by powers of two (Q, K, V x 16; P x 64 through the softmax offset) so that the low planes stay out of fp16's subnormal range.
// Tile = [plane 2][k-step 2][lane 64][8 halves] = 4 KiB (the size of the fp32 tile).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define GATOR_MFMA_F16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
struct X2 { f16x8 p[2][2]; };       // [plane hi/lo][k-step]: 16 VGPRs

__device__ __forceinline__ X2 x2_split(const f32x16& v) {
    X2 o;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = v[8 * s + j];
            const _Float16 h = (_Float16)x;
            o.p[0][s][j] = h;
            o.p[1][s][j] = (_Float16)(x - (float)h);
        }
    return o;
}
__device__ __forceinline__ X2 x2_load(const float* __restrict__ tile, int lane) {
    const f16x8* q = reinterpret_cast<const f16x8*>(tile) + lane;
    X2 o;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int s = 0; s < 2; ++s) o.p[pl][s] = q[(pl * 2 + s) * 64];
    return o;
}
__device__ __forceinline__ void x2_store(float* __restrict__ tile, int lane, const X2& v) {
    f16x8* q = reinterpret_cast<f16x8*>(tile) + lane;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int s = 0; s < 2; ++s) q[(pl * 2 + s) * 64] = v.p[pl][s];
}
// Order of the partial products (round 4).  Every MFMA rounds the fp32 accumulator once AT THE ACCUMULATOR'S MAGNITUDE, whatever the
// size of what it adds; in-product accumulation is where this path's fp32 noise comes from (profiles/r04_error_budget.md: rounding
// only the tensors an op writes gives a quarter of it).  So the cross products (2^-11 of the result) of ALL k-steps go first, while the
// accumulator is still small, and the hi*hi products last: a fresh 32-deep product then rounds twice at full magnitude instead of six
// times.  Same instructions, same count.
__device__ __forceinline__ f32x16 x2_mma_small(const X2& A, const X2& B, f32x16 acc) {      // the four cross products only
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        acc = GATOR_MFMA_F16(A.p[1][s], B.p[0][s], acc);      // lo*hi
        acc = GATOR_MFMA_F16(A.p[0][s], B.p[1][s], acc);      // hi*lo
    }
    return acc;
}
__device__ __forceinline__ f32x16 x2_mma_main(const X2& A, const X2& B, f32x16 acc) {       // the two hi*hi products only
#pragma unroll
    for (int s = 0; s < 2; ++s) acc = GATOR_MFMA_F16(A.p[0][s], B.p[0][s], acc);
    return acc;
}
__device__ __forceinline__ f32x16 x2_mma(const X2& A, const X2& B, f32x16 acc) { return x2_mma_main(A, B, x2_mma_small(A, B, acc)); }

// ---- "4-product" token-wise linears (round 3, MDR with GATOR_MDR_X3=2): activations on TWO fp16 planes, weights on THREE ---------
// The weight side dominates the error of a rounded linear because its rounding is the same for every token and sample (emulated in
// the fp64 oracle: all GAT linears with both operands on two planes cost 3.7e-4 mm, with only the activations rounded 9.5e-5).  So
// the weights stay EXACT -- fp16 x 3 planes, w 2^k = hi + mid + lo with 11 + 11 + >= 2 bits, one power-of-two scale for the whole
// weight set chosen at pack time so that max|w| 2^k < 2^14 -- and only the activations are rounded to 22 bits (two planes of
// 16 x value).  Products: a_hi (w_hi + w_mid + w_lo) + a_lo w_hi -- 4 MFMAs per k-step instead of 6; dropped: a_lo (w_mid + w_lo),
// 2^-22 of the product.  Emulated in the oracle for every token-wise linear of the MDR layers: vertices move by 1.3e-4 mm max /
// 2.4e-5 mm rms (the path's own fp32 noise: 6 - 7e-4 / 8 - 9e-5).  Results come out scaled by S = 2^(4 + k); every consumer folds
// 1 / S into a multiplication it does anyway (power of two: no rounding changes).  Same tile geometry as an X3 tile.
struct H3 { f16x8 p[3][2]; };       // [plane hi/mid/lo][k-step]: 24 VGPRs
__device__ __forceinline__ H3 h3_load(const float* __restrict__ tile, int lane) {
    const f16x8* q = reinterpret_cast<const f16x8*>(tile) + lane;
    H3 o;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int s = 0; s < 2; ++s) o.p[pl][s] = q[(pl * 2 + s) * 64];
    return o;
}
// two planes of scale * v (scale: a power of two that also carries whatever factor the producer left in v)
__device__ __forceinline__ X2 x2_split_scaled(const f32x16& v, float scale) { return x2_split(v * scale); }
// acc += W . a (rows of the result = the weight's lane index), in two parts so that a caller chaining several tiles into one
// accumulator can run the cross products of ALL its tiles first and the hi*hi products last (see x2_mma)
__device__ __forceinline__ f32x16 h3_mma_wa_small(const H3& W, const X2& a, f32x16 acc) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        acc = GATOR_MFMA_F16(W.p[2][s], a.p[0][s], acc);      // w lo  * a hi
        acc = GATOR_MFMA_F16(W.p[0][s], a.p[1][s], acc);      // w hi  * a lo
        acc = GATOR_MFMA_F16(W.p[1][s], a.p[0][s], acc);      // w mid * a hi
    }
    return acc;
}
__device__ __forceinline__ f32x16 h3_mma_wa_main(const H3& W, const X2& a, f32x16 acc) {
#pragma unroll
    for (int s = 0; s < 2; ++s) acc = GATOR_MFMA_F16(W.p[0][s], a.p[0][s], acc);      // w hi  * a hi
    return acc;
}
__device__ __forceinline__ f32x16 h3_mma_wa(const H3& W, const X2& a, f32x16 acc) { return h3_mma_wa_main(W, a, h3_mma_wa_small(W, a, acc)); }
// acc += a . W (rows of the result = the activation's lane index)
__device__ __forceinline__ f32x16 h3_mma_aw_small(const X2& a, const H3& W, f32x16 acc) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        acc = GATOR_MFMA_F16(a.p[0][s], W.p[2][s], acc);
        acc = GATOR_MFMA_F16(a.p[1][s], W.p[0][s], acc);
        acc = GATOR_MFMA_F16(a.p[0][s], W.p[1][s], acc);
    }
    return acc;
}
__device__ __forceinline__ f32x16 h3_mma_aw_main(const X2& a, const H3& W, f32x16 acc) {
#pragma unroll
    for (int s = 0; s < 2; ++s) acc = GATOR_MFMA_F16(a.p[0][s], W.p[0][s], acc);
    return acc;
}
__device__ __forceinline__ f32x16 h3_mma_aw(const X2& a, const H3& W, f32x16 acc) { return h3_mma_aw_main(a, W, h3_mma_aw_small(a, W, acc)); }


// ---- ONE fp16 plane ("X1"; BASELINE config 3, the 16-bit operand mode: DESIGN.md 4e) -------------------------------------------------
// Activations (and the attention operands of the MDR layers) as one fp16 plane of 16 x value: no split, 2 KiB tiles; weights as the two
// leading planes (hi, mid: 22 bits) of the same H3 tiles the fp32 configuration packs ("G2") -- two MFMAs per k-step instead of four.
struct X1 { f16x8 p[2]; };          // [k-step]: 8 VGPRs
__device__ __forceinline__ X1 x1_cvt(const f32x16& v) {
    X1 o;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) o.p[s][j] = (_Float16)v[8 * s + j];
    return o;
}
__device__ __forceinline__ X1 x1_load(const float* __restrict__ tile, int lane) {      // also: the hi plane of an X2 tile
    const f16x8* q = reinterpret_cast<const f16x8*>(tile) + lane;
    X1 o;
    o.p[0] = q[0];
    o.p[1] = q[64];
    return o;
}
__device__ __forceinline__ void x1_store(float* __restrict__ tile, int lane, const X1& v) {
    f16x8* q = reinterpret_cast<f16x8*>(tile) + lane;
    q[0] = v.p[0];
    q[64] = v.p[1];
}
__device__ __forceinline__ f32x16 x1_mma(const X1& A, const X1& B, f32x16 acc) {
    acc = GATOR_MFMA_F16(A.p[0], B.p[0], acc);
    return GATOR_MFMA_F16(A.p[1], B.p[1], acc);
}
struct G2 { f16x8 hi[2], mid[2]; };      // [k-step]
__device__ __forceinline__ G2 g2_load(const float* __restrict__ tile, int lane) {
    const f16x8* q = reinterpret_cast<const f16x8*>(tile) + lane;
    G2 o;
#pragma unroll
    for (int s = 0; s < 2; ++s) { o.hi[s] = q[s * 64]; o.mid[s] = q[(2 + s) * 64]; }
    return o;
}
__device__ __forceinline__ f32x16 g2_mma_wa(const G2& W, const X1& a, f32x16 acc) {      // acc += W . a, the small plane first
    acc = GATOR_MFMA_F16(W.mid[0], a.p[0], acc);
    acc = GATOR_MFMA_F16(W.mid[1], a.p[1], acc);
    acc = GATOR_MFMA_F16(W.hi[0], a.p[0], acc);
    return GATOR_MFMA_F16(W.hi[1], a.p[1], acc);
}

// fused_pack.hip: fp32 packed tiles [g][lane][4] -> X3 tiles, same tile indices
int fused_repack_x3(const float* src_tiles, float* dst_tiles, int64_t ntiles, void* stream);
// ... -> H3 tiles (three fp16 planes of 2^shift * w); *shift is chosen from max|w| over the tiles with (tile % period) < live (the
// weight grids; period 0 = every tile), so that 2^13 <= max|2^shift w| < 2^14.  The three planes hold 2^shift w down to 2^-24
// absolute, i.e. every weight to 2^-38 of the largest one ("exact" in this code base means that); *residual is the largest
// |2^shift w - (hi + mid + lo)| relative to max|2^shift w| -- at most 2^-39 by construction, reported as a sanity value.
int fused_repack_h3(const float* src_tiles, float* dst_tiles, int64_t ntiles, int* shift, float* residual, void* stream,
                    int64_t period = 0, int64_t live = 0);

}  // namespace gator
