// Device helpers of the "2 x f16" split-precision projection kernel (k1_pair.hip): the precision
// policy, the register split of float32 operands into two float16 pieces, the phase-table tile
// image.
#pragma once
#include <type_traits>

#include "psa_ctx.h"

namespace psa {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Both operands are first multiplied by a power of two that puts them at the top of the float16
// range (P' by 2^14; d by 2^(14-e), 2^e >= max|d| of the resident array, found once per upload by
// absmax_bits_kernel), then x = x1 + x2, x1 = f16(x), x2 = f16(x - x1), round-to-nearest, the
// residual exact in float32.  Each piece carries 11 significant bits plus the sign of the
// residual: |x - x1 - x2| <= 2^-24 |x| down to |x| = 2^-17 of the maximum, and never worse than
// 2^-39 of the maximum below that (f16 subnormals; were the matrix core to flush them: 2^-28).
//     x*y ~= x1*y1 + [ x1*y2 + x2*y1 ]                        (dropped: x2*y2 <= 2^-24 |x*y|)
// Three v_mfma_f32_16x16x32_f16 replace sixteen float32 MFMA-equivalents.  The scales are powers
// of two, so removing them in the epilogue is exact.
struct F16x2 {
    typedef _Float16 elem;
    typedef _Float16 v2 __attribute__((ext_vector_type(2)));
    typedef _Float16 v4 __attribute__((ext_vector_type(4)));
    typedef _Float16 v8 __attribute__((ext_vector_type(8)));
    static constexpr int   NP = 2;               // pieces per operand
    static constexpr int   NTERM = 3;            // MFMAs per (row tile, component, 32 atoms)
    static constexpr float P_SCALE = 16384.f;    // |P'| <= 1 -> top of the f16 range
    static __device__ __forceinline__ f32x4 mma(v8 a, v8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};

// two floats (times the power of two s) -> their leading and residual float16 pairs:
// v_mul_f32 x2, v_cvt_pk_f16_f32, v_fma_mix_f32 x2 (x * s - x1, the f16 operand read in place),
// v_cvt_pk_f16_f32 -- single-issue VALU only (packed-f32 ops are slow beside MFMAs)
__device__ __forceinline__ void split_pair(float x0, float x1, float s, F16x2::v2& lead, F16x2::v2& rest) {
    lead = __builtin_convertvector(f32x2{x0 * s, x1 * s}, F16x2::v2);
    const float r0 = __builtin_fmaf(x0, s, -(float)lead[0]);
    const float r1 = __builtin_fmaf(x1, s, -(float)lead[1]);
    rest = __builtin_convertvector(f32x2{r0, r1}, F16x2::v2);
}
__device__ __forceinline__ F16x2::v8 cat4(F16x2::v2 a, F16x2::v2 b, F16x2::v2 c, F16x2::v2 d) {
    const F16x2::v4 lo = __builtin_shufflevector(a, b, 0, 1, 2, 3);
    const F16x2::v4 hi = __builtin_shufflevector(c, d, 0, 1, 2, 3);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// raw: a lane's 8 atoms x 3 components as they lie in HBM (24 floats, component-minor, in 6
// register quads).  B fragments of component CC: b[piece] = the 8 atoms of this lane's frame
template <int I, int CC>
__device__ __forceinline__ float relem(const f32x4 (&raw)[6]) {
    return raw[(3 * I + CC) >> 2][(3 * I + CC) & 3];
}
template <int CC>
__device__ __forceinline__ void split_component(const f32x4 (&raw)[6], float s, F16x2::v8 (&b)[2]) {
    F16x2::v2 lead[4], rest[4];
    split_pair(relem<0, CC>(raw), relem<1, CC>(raw), s, lead[0], rest[0]);
    split_pair(relem<2, CC>(raw), relem<3, CC>(raw), s, lead[1], rest[1]);
    split_pair(relem<4, CC>(raw), relem<5, CC>(raw), s, lead[2], rest[2]);
    split_pair(relem<6, CC>(raw), relem<7, CC>(raw), s, lead[3], rest[3]);
    b[0] = cat4(lead[0], lead[1], lead[2], lead[3]);
    b[1] = cat4(rest[0], rest[1], rest[2], rest[3]);
}
// the same from the gathered image: one (x, y, z, -) quad per atom
template <int CC>
__device__ __forceinline__ void split_component(const f32x4 (&raw)[8], float s, F16x2::v8 (&b)[2]) {
    F16x2::v2 lead[4], rest[4];
    split_pair(raw[0][CC], raw[1][CC], s, lead[0], rest[0]);
    split_pair(raw[2][CC], raw[3][CC], s, lead[1], rest[1]);
    split_pair(raw[4][CC], raw[5][CC], s, lead[2], rest[2]);
    split_pair(raw[6][CC], raw[7][CC], s, lead[3], rest[3]);
    b[0] = cat4(lead[0], lead[1], lead[2], lead[3]);
    b[1] = cat4(rest[0], rest[1], rest[2], rest[3]);
}

// index of element (piece, row m, atom a) in the phase-table image: [M block][atom stage]
// [piece][row][32 atoms]; the four 16-byte slots of a row (8 atoms each) are XOR-swizzled by
// g((row>>2)&3), g = {0,2,3,1} packed as 0x78, which makes the A-fragment ds_read_b128 conflict-free
__host__ __device__ inline size_t pf16_tile_index(int piece, int m, int a, int m_blk, int n_stage) {
    const int    row = m % m_blk, al = a % K1_BA;
    const size_t tile = ((size_t)(m / m_blk) * n_stage + a / K1_BA) * ((size_t)F16x2::NP * m_blk * K1_BA);
    const int    sw = (0x78 >> (2 * ((row >> 2) & 3))) & 3;
    return tile + ((size_t)piece * m_blk + row) * K1_BA + (((al >> 3) ^ sw) << 3) + (al & 7);
}

}  // namespace psa
