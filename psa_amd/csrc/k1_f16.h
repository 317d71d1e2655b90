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

typedef __attribute__((address_space(3))) unsigned char   lds_u8;
typedef const __attribute__((address_space(3))) F16x2::v8 lds_cv8;
typedef const __attribute__((address_space(3))) f32x4     lds_cf32x4;

__device__ __forceinline__ int vs_phys_slot(int s, int row) { return (s & ~7) | ((s & 7) ^ (row & 7)); }

// One LDS-DMA instruction (64 lanes x 16 bytes, global -> LDS at dst + lane * 16), issued from
// inline assembly: hipcc's wait-count pass books a global_load_lds as a FLAT access that may touch
// LDS and, while one is pending, turns every LDS-data wait into lgkmcnt(0) and knows no partial
// vmcnt.  Its completion is awaited explicitly (s_waitcnt vmcnt below); the "memory" clobber keeps
// LDS accesses from being moved across it.  M0 is a reserved register the compiler re-materialises
// in front of its own uses.
__device__ __forceinline__ void lds_dma16(const void* g, unsigned lds_byte_addr) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_byte_addr);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(dst) : "memory");
}
// N consecutive 1-KiB pieces in ONE addressing setup: uniform 64-bit base in SGPRs + a 32-bit
// per-lane offset; piece i adds the instruction offset 1024 i, which the hardware applies to the
// global address AND to the LDS address (tools/probes/dma_offset.hip).  One M0 write, no VALU.
// NT: non-temporal cache policy (aux nt) -- for bytes this launch reads exactly once, streamed from HBM
// (MI355X_MICROARCH.md "nt-weights": issued -> landed -18 %); never for data other workgroups re-read
template <int N, bool NT = false>
__device__ __forceinline__ void lds_dma16_group(const void* sbase, unsigned voff, unsigned lds_byte_addr) {
    static_assert(N >= 1 && N <= 4, "instruction offsets reach 4095");
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_byte_addr);
#define PSA_DMA_ASM(P)                                                                                                         \
    if constexpr (N == 1)                                                                                                      \
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" P ::"v"(voff), "s"(sbase), "s"(dst) : "memory"); \
    else if constexpr (N == 2)                                                                                                 \
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" P "\n\t"                                \
                     "global_load_lds_dwordx4 %0, %1 offset:1024" P ::"v"(voff), "s"(sbase), "s"(dst) : "memory");             \
    else if constexpr (N == 3)                                                                                                 \
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" P "\n\t"                                \
                     "global_load_lds_dwordx4 %0, %1 offset:1024" P "\n\tglobal_load_lds_dwordx4 %0, %1 offset:2048" P         \
                     ::"v"(voff), "s"(sbase), "s"(dst) : "memory");                                                            \
    else                                                                                                                       \
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" P "\n\t"                                \
                     "global_load_lds_dwordx4 %0, %1 offset:1024" P "\n\tglobal_load_lds_dwordx4 %0, %1 offset:2048" P "\n\t"  \
                     "global_load_lds_dwordx4 %0, %1 offset:3072" P ::"v"(voff), "s"(sbase), "s"(dst) : "memory");
    if constexpr (NT) {
        PSA_DMA_ASM(" nt")
    } else {
        PSA_DMA_ASM("")
    }
#undef PSA_DMA_ASM
}
// one piece of such a group on its own (instruction offset OFF applied to the global and the LDS address)
template <int OFF, bool NT = false>
__device__ __forceinline__ void lds_dma16_at(const void* sbase, unsigned voff, unsigned lds_byte_addr) {
    static_assert(OFF >= 0 && OFF <= 3072 && OFF % 1024 == 0, "instruction offsets reach 4095");
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_byte_addr);
    if constexpr (NT)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3 nt" ::"v"(voff), "s"(sbase), "s"(dst), "n"(OFF) : "memory");
    else
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" ::"v"(voff), "s"(sbase), "s"(dst), "n"(OFF) : "memory");
}
// 12 bytes per lane, landing at dst + lane * 16
__device__ __forceinline__ void lds_dma12(const void* g, unsigned lds_byte_addr) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_byte_addr);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx3 %0, off" ::"v"(g), "s"(dst) : "memory");
}
// 4 bytes per lane, landing at dst + lane * 4
__device__ __forceinline__ void lds_dma4(const void* g, unsigned lds_byte_addr) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_byte_addr);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(g), "s"(dst) : "memory");
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

// ---------------------------------------------------------------------------------------------
// Split planes of a trajectory group (k1_planes.hip): the group's data d[t, a, c], already scaled
// and split into its two float16 pieces, in the image the planes kernel DMAs into LDS --
//     [frame group of 16][atom stage of 32][component][piece][frame 16][32 atoms]   (float16)
// i.e. one 1-KiB block per (frame group, stage, component, piece) = one LDS-DMA instruction and,
// read back with the slot swizzle below, the B fragments of v_mfma_f32_16x16x32_f16 as they
// stand: lane (frame r, atom octet q) takes 16 bytes at r * 64 + ((q ^ g(r)) << 4).  Same bytes
// per value as the float32 array (2 + 2).  Frames and atoms past the group's end are zero.
// ---------------------------------------------------------------------------------------------
constexpr int PL_BLOCK_ELEMS = 16 * K1_BA;                 // one (component, piece) block: 1 KiB
constexpr int PL_STAGE_ELEMS = 3 * F16x2::NP * PL_BLOCK_ELEMS;   // 6 KiB per (frame group, stage)
__host__ __device__ inline int pl_swizzle(int r) { return (0x78 >> (2 * ((r >> 2) & 3))) & 3; }
__host__ __device__ inline size_t plane_index(int64_t fg, int stage, int comp, int piece, int r, int al, int n_stage) {
    return ((size_t)(fg * n_stage + stage) * (3 * F16x2::NP) + (size_t)(comp * F16x2::NP + piece)) * PL_BLOCK_ELEMS +
           (size_t)r * K1_BA + (size_t)((((al >> 3) ^ pl_swizzle(r)) << 3) + (al & 7));
}
// (+ 4 stages of padding: the planes kernel prefetches up to RING <= 4 stages past a frame group's end)
inline size_t plane_bytes(int64_t n_fg, int n_stage) {
    return ((size_t)n_fg * n_stage + 4) * PL_STAGE_ELEMS * sizeof(_Float16);
}

}  // namespace psa
