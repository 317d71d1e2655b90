// K1, split-precision form: the same projection GEMM as k1_mfma.hip,
//     D[m, (c,t)] = sum_a P'[m, a] * d[t, a, c]
// evaluated on the bf16 matrix cores with fp32-equivalent accuracy ("3 x bf16"):
// every float32 operand x is written x = x1 + x2 + x3 with x1 = bf16(x),
// x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)  (round-to-nearest, residuals exact in
// float32, |x - x1 - x2 - x3| <= 2^-27 |x|), and a product keeps the six leading terms
//     x*y ~= x1*y1 + [ x1*y2 + x2*y1 + x2*y2 + x1*y3 + x3*y1 ]     (dropped: <= 2^-26 |x*y|)
// on v_mfma_f32_16x16x32_bf16: six bf16 MFMAs replace sixteen float32 MFMA-equivalents,
// 2.67x the fp32 matrix rate (SURVEY.md section 7-2 option (ii)).
//
// TWO ACCUMULATORS.  The leading term x1*y1 goes to `hi`, the five small terms to `lo`, and
// hi + lo is formed once in the epilogue.  This is not optional: the bf16 MFMA aligns its 32
// products to the exponent of the accumulator input and keeps only ~2-3 bits below the
// accumulator's ulp (measured: tools/probes/mfma_rounding.hip), so a correction term that is
// small against the running sum is truncated to nothing when it is added to it.  For a
// coherent phonon mode the running sum grows to ~N*A while the x2*y1 corrections stay at the
// 2^-9 level of single products: with one accumulator their coherent part (3e-6 .. 1e-5 of
// the peak on the silicon lattice of configuration 3) was silently lost.  Kept apart, `lo`
// never exceeds ~2^-8 |hi| and every term lands above its ulp.
//
//  - P' is split ONCE by the phase kernel and stored as three bf16 planes in the tile image
//    this kernel DMAs into LDS ([piece][M_BLK][32 atoms], 16-byte slots XOR-swizzled so that
//    the A-fragment ds_read_b128 are conflict-free).
//  - d stays float32 in HBM and in LDS (native (T,N,3) rows, same swizzled image as
//    k1_mfma.hip).  A lane reads its 8 atoms x 3 components = 96 contiguous bytes and splits
//    them in registers (v_cvt_pk_bf16_f32 / shift / subtract: ~4.5 VALU ops per value), which
//    also resolves the (atom, component) interleave.
//  - Work decomposition: 4 wavefronts side by side along t, each owning 16 frames x all M_BLK
//    rows (M_BLK/16 row tiles x 3 components x {hi, lo} accumulators of 16x16 = 192 AGPRs at
//    M_BLK = 128).  A wavefront's V rows are written (by its own DMA) and read by itself only.
//  - 3-deep LDS ring, one barrier per 32-atom stage: while stage s is multiplied, stage s+1
//    (already landed) is read and split in the MFMAs' shadow and stage s+2 travels by LDS-DMA.
//    No LDS read is ever issued while a DMA that could alias it is in flight -- hipcc would
//    guard it with vmcnt(0) (see k1_mfma.hip) -- because reads happen at the stage top, DMA
//    issue after them, and the stage ends with vmcnt(0) + barrier.
//  - MFMA shape 16x16x32 rather than 32x32x16: equal cycles per flop, but the chip holds a higher
//    clock on it (MI355X_MICROARCH.md, DVFS give-back item 7; measured here +5 %).
#include <type_traits>

#include "psa_ctx.h"

namespace psa {

typedef float  f32x4 __attribute__((ext_vector_type(4)));
typedef float  f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

template <int MT16, bool GATHER>
struct K1sCfg {
    static constexpr int M_BLK = 16 * MT16;
    static constexpr int T_BLK = 64;                             // 4 wavefronts x 16 frames
    // Contiguous groups: M_BLK = 128 fills the register file (1 workgroup/CU) and runs a 3-deep
    // ring with the next stage split in the MFMAs' shadow; the small-K variants (HBM-bound) fit two
    // workgroups per CU instead, 2-deep ring each, the other workgroup covering this one's reads,
    // split and barrier.  Gathered groups have a larger V image (16 bytes per atom): 2-deep ring,
    // one workgroup per CU.
    static constexpr int RING = (MT16 > 4 && !GATHER) ? 3 : 2;
    static constexpr int WG_PER_CU = (MT16 > 4 || GATHER) ? 1 : 2;
    // V tile of one wavefront: contiguous = 16 rows x 384 B (swizzled 16-byte slots);
    // gathered = 8 DMA pieces (2 frames x 32 atoms x 16 B, the 12-byte LDS-DMA element lands on a
    // 16-byte pitch) + 16 B of padding per piece, which keeps the raw reads at 2-way conflicts
    static constexpr int V_PIECE_BYTES = GATHER ? 1024 + 16 : 1024;
    static constexpr int V_DMA = GATHER ? 8 : 6;
    static constexpr int V_WAVE_BYTES = V_DMA * V_PIECE_BYTES;
    static constexpr int V_STAGE_BYTES = 4 * V_WAVE_BYTES;
    static constexpr int P_STAGE_BYTES = 3 * M_BLK * K1_BA * 2;  // three bf16 planes
    static constexpr int STAGE_BYTES = V_STAGE_BYTES + P_STAGE_BYTES;
    static constexpr int LDS_BYTES = RING * STAGE_BYTES;
    static constexpr int RAWN = GATHER ? 8 : 6;                  // 16-byte reads per lane per stage
    static constexpr int P_CHUNKS = P_STAGE_BYTES / 16;
    static constexpr int P_DMA = (P_CHUNKS + 255) / 256;
    static_assert(P_CHUNKS % 64 == 0, "P' tile must be whole wave-instructions");
    static_assert(WG_PER_CU * LDS_BYTES <= 160 * 1024, "LDS budget");
};

__device__ __forceinline__ int vs_phys_slot(int s, int row) { return (s & ~7) | ((s & 7) ^ (row & 7)); }

// One 16-byte LDS read at a byte address (+ compile-time offset).  hipcc (ROCm 7.2) puts
// s_waitcnt vmcnt(0) in front of every ds_read it can see while an LDS-DMA is in flight (it cannot
// prove they do not alias); the stage schedules below therefore issue all LDS reads of a stage at
// its top, before that stage's DMA, when the previous DMA has already been drained.
template <int OFF, class T>
__device__ __forceinline__ void lds_read128(T& dst, unsigned addr) {
    dst = *reinterpret_cast<const __attribute__((address_space(3))) T*>(
        (const __attribute__((address_space(3))) unsigned char*)(size_t)(addr + OFF));
}

// x (two floats) -> three packed bf16 pairs
__device__ __forceinline__ void split3(f32x2 x, bf16x2& p1, bf16x2& p2, bf16x2& p3) {
    p1 = __builtin_convertvector(x, bf16x2);
    const f32x2 r1 = x - __builtin_convertvector(p1, f32x2);
    p2 = __builtin_convertvector(r1, bf16x2);
    const f32x2 r2 = r1 - __builtin_convertvector(p2, f32x2);
    p3 = __builtin_convertvector(r2, bf16x2);
}
__device__ __forceinline__ bf16x8 cat4(bf16x2 a, bf16x2 b, bf16x2 c, bf16x2 d) {
    const bf16x4 lo = __builtin_shufflevector(a, b, 0, 1, 2, 3);
    const bf16x4 hi = __builtin_shufflevector(c, d, 0, 1, 2, 3);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// raw holds a lane's 8 atoms x 3 components: packed (24 floats in 6 register quads, component-
// minor as in HBM) or padded (one x,y,z,- quad per atom: the gathered image).  Component CC of
// atom I:
template <int N, int I, int CC>
__device__ __forceinline__ float relem(const f32x4 (&raw)[N]) {
    if constexpr (N == 8) return raw[I][CC];
    else return raw[(3 * I + CC) >> 2][(3 * I + CC) & 3];
}
template <int CC, int N>
__device__ __forceinline__ void split_component(const f32x4 (&raw)[N], bf16x8& b1, bf16x8& b2, bf16x8& b3) {
    bf16x2 p1[4], p2[4], p3[4];
    split3(f32x2{relem<N, 0, CC>(raw), relem<N, 1, CC>(raw)}, p1[0], p2[0], p3[0]);
    split3(f32x2{relem<N, 2, CC>(raw), relem<N, 3, CC>(raw)}, p1[1], p2[1], p3[1]);
    split3(f32x2{relem<N, 4, CC>(raw), relem<N, 5, CC>(raw)}, p1[2], p2[2], p3[2]);
    split3(f32x2{relem<N, 6, CC>(raw), relem<N, 7, CC>(raw)}, p1[3], p2[3], p3[3]);
    b1 = cat4(p1[0], p1[1], p1[2], p1[3]);
    b2 = cat4(p2[0], p2[1], p2[2], p2[3]);
    b3 = cat4(p3[0], p3[1], p3[2], p3[3]);
}

// GATHER = false: the group is the whole trajectory in order (N % 4 == 0): V rows are copied in
//                  16-byte pieces into the swizzled image.
// GATHER = true : arbitrary index list (any order, duplicates) or any N: one (frame, atom) triple
//                  = one 12-byte LDS-DMA element, 64 atoms per instruction, into a
//                  [frame][atom][x,y,z,-] image (the hardware places 12-byte elements on a 16-byte
//                  pitch: tools/probes/dma12.hip).  Each lane serves one atom column of the stage,
//                  so it needs ONE index per stage.
template <int MT16, bool GATHER>
__global__ void __launch_bounds__(256, (K1sCfg<MT16, GATHER>::WG_PER_CU))
k1_split_kernel(const float* __restrict__ V, const __bf16* __restrict__ Pb, const int* __restrict__ idx,
                float2* __restrict__ Q, int64_t T, int64_t q_stride, int64_t N_tot, int n_g, int A_pad, int K,
                int n_mblk, int n_tblk) {
    using C = K1sCfg<MT16, GATHER>;
    // ring slot r: [V tile: T_BLK rows x 96 float32][P' tile: 3 planes x M_BLK rows x 32 bf16]
    __shared__ __attribute__((aligned(16))) unsigned char smem[C::LDS_BYTES];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

    // XCD-aware block map: blocks b and b+8 share an XCD (and its L2); they get the M-blocks
    // of one frame tile, so V is fetched from HBM once per frame tile.
    const int b  = blockIdx.x;
    const int r8 = b >> 3;
    const int mb = r8 % n_mblk;
    const int tb = (r8 / n_mblk) * 8 + (b & 7);
    if (tb >= n_tblk) return;

    const int     tid = threadIdx.x, lane = tid & 63;
    const int     wn = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: DMA targets in SGPRs
    const int     r16 = lane & 15, q = lane >> 4;                  // frame / row lane, 8-atom group
    const int64_t t0 = (int64_t)tb * C::T_BLK + wn * 16;          // this wavefront's first frame
    const int     n_stage = A_pad / K1_BA;

    // ---- DMA sources: this wavefront's 16 V rows + its share of the P' tile -------------------
    // Contiguous form: one pointer per 1-KiB piece (6), advanced by one stage (32 atoms = 96
    // floats) after each issue.  The last stage of a trajectory whose atom count is not a multiple
    // of 32 reads up to 31 atoms past the row end: the next frame's data or the zeroed slack
    // behind the array, times P' = 0.
    // Gather form: piece j covers frames 2j, 2j+1 of the wavefront x the stage's 32 atoms; lane l
    // serves atom column l & 31 of frame 2j + (l >> 5), so it keeps 8 row pointers and one index.
    constexpr int VP = C::V_DMA;
    const float*  vp[VP];
#pragma unroll
    for (int j = 0; j < VP; ++j) {
        if constexpr (GATHER) {
            int64_t t = t0 + 2 * j + (lane >> 5);
            if (t >= T) t = T - 1;
            vp[j] = V + t * 3 * N_tot;
        } else {
            const int L = j * 64 + lane;
            const int row = L / 24, phys = L - row * 24;
            int64_t   t = t0 + row;
            if (t >= T) t = T - 1;                                // rows past the end: finite filler
            vp[j] = V + t * 3 * N_tot + 4 * vs_phys_slot(phys, row);
        }
    }
    // atom served by this lane in stage st (columns past the group's end carry P' = 0: any valid atom)
    auto atom_of = [&](int st) {
        int pos = st * K1_BA + (lane & 31);
        if (pos >= n_g) pos = n_g - 1;
        return idx ? idx[pos] : pos;
    };
    int atom_dma = 0, atom_next = 0;        // GATHER: index for the next DMA / the one after
    if constexpr (GATHER) {
        atom_dma = atom_of(0);
        atom_next = atom_of(n_stage > 1 ? 1 : 0);
    }
    int dma_count = 0;                      // stages issued so far
    const unsigned char* pp = reinterpret_cast<const unsigned char*>(Pb) +
                              (size_t)mb * n_stage * C::P_STAGE_BYTES + 16 * (wn * C::P_DMA * 64 + lane);
    auto dma_p = [&](int slot) {
        unsigned char* dst = smem + slot * C::STAGE_BYTES + C::V_STAGE_BYTES;
#pragma unroll
        for (int j = 0; j < C::P_DMA; ++j) {
            const int chunk = (wn * C::P_DMA + j) * 64;
            if constexpr (C::P_CHUNKS % 256 == 0) {
                __builtin_amdgcn_global_load_lds((gbl_void*)(pp + 1024 * j), (lds_void*)(dst + 16 * chunk),
                                                 16, 0, 0);
            } else {
                if (chunk < C::P_CHUNKS)      // wave-uniform
                    __builtin_amdgcn_global_load_lds((gbl_void*)(pp + 1024 * j),
                                                     (lds_void*)(dst + 16 * chunk), 16, 0, 0);
            }
        }
        pp += C::P_STAGE_BYTES;
    };
    // one M block: every byte of the trajectory is read once by this launch -> non-temporal policy
    // (aux 2, as for the planes in k1_planes.hip); several M blocks re-read it through L2
    const bool nt = n_mblk == 1;
    auto dma_v = [&](int slot) {
        unsigned char* vd = smem + slot * C::STAGE_BYTES + wn * C::V_WAVE_BYTES;
        if constexpr (GATHER) {
            // the index of this stage was loaded a stage ago; fetch the one after next now
            const int a = atom_dma;
#pragma unroll
            for (int j = 0; j < VP; ++j) {
                if (nt)
                    __builtin_amdgcn_global_load_lds((gbl_void*)(vp[j] + 3 * (int64_t)a),
                                                     (lds_void*)(vd + j * C::V_PIECE_BYTES), 12, 0, 2);
                else
                    __builtin_amdgcn_global_load_lds((gbl_void*)(vp[j] + 3 * (int64_t)a),
                                                     (lds_void*)(vd + j * C::V_PIECE_BYTES), 12, 0, 0);
            }
            ++dma_count;
            atom_dma = atom_next;
            atom_next = atom_of(dma_count + 1 < n_stage ? dma_count + 1 : n_stage - 1);
        } else {
#pragma unroll
            for (int j = 0; j < VP; ++j) {
                if (nt)
                    __builtin_amdgcn_global_load_lds((gbl_void*)vp[j], (lds_void*)(vd + j * C::V_PIECE_BYTES), 16, 0, 2);
                else
                    __builtin_amdgcn_global_load_lds((gbl_void*)vp[j], (lds_void*)(vd + j * C::V_PIECE_BYTES), 16, 0, 0);
                vp[j] += K1_VROW;
            }
        }
    };

    f32x4 hi[MT16][3], lo[MT16][3];          // [row tile][component]: leading term / correction terms
#pragma unroll
    for (int mt = 0; mt < MT16; ++mt)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            hi[mt][c] = f32x4{0.f, 0.f, 0.f, 0.f};
            lo[mt][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

    // P' slot swizzle: slot q of row r is stored at q ^ g((r>>2)&3), g = {0,2,3,1} packed two bits
    // each = 0x78 (makes the four 16-lane ds_read_b128 groups of the A fragment conflict-free)
    const int      gsw = (0x78 >> (2 * ((r16 >> 2) & 3))) & 3;
    // this lane's frame row (gathered: its 8-atom group too) in the wavefront's V tile
    const unsigned v_lane = lds0 + wn * C::V_WAVE_BYTES +
                            (GATHER ? (r16 >> 1) * C::V_PIECE_BYTES + (r16 & 1) * 512 + q * 128
                                    : r16 * (K1_VROW * 4));
    const unsigned p_lane = lds0 + C::V_STAGE_BYTES + r16 * (K1_BA * 2) + ((q ^ gsw) << 4);

    // raw float32 d of one stage: atoms 8q .. 8q+7 x 3 components = 96 contiguous bytes in HBM,
    // six swizzled 16-byte slots in LDS
    auto read_raw = [&](int slot, f32x4 (&raw)[C::RAWN]) {
        const unsigned base = v_lane + slot * C::STAGE_BYTES;
        if constexpr (GATHER) {                  // 8 atoms x (x,y,z,-): 128 contiguous bytes
            lds_read128<0>(raw[0], base);
            lds_read128<16>(raw[1], base);
            lds_read128<32>(raw[2], base);
            lds_read128<48>(raw[3], base);
            lds_read128<64>(raw[4], base);
            lds_read128<80>(raw[5], base);
            lds_read128<96>(raw[6], base);
            lds_read128<112>(raw[7], base);
        } else {                                 // 96 contiguous bytes of HBM, six swizzled slots
            const int s0 = 6 * q;
            lds_read128<0>(raw[0], base + 16 * vs_phys_slot(s0 + 0, r16));
            lds_read128<0>(raw[1], base + 16 * vs_phys_slot(s0 + 1, r16));
            lds_read128<0>(raw[2], base + 16 * vs_phys_slot(s0 + 2, r16));
            lds_read128<0>(raw[3], base + 16 * vs_phys_slot(s0 + 3, r16));
            lds_read128<0>(raw[4], base + 16 * vs_phys_slot(s0 + 4, r16));
            lds_read128<0>(raw[5], base + 16 * vs_phys_slot(s0 + 5, r16));
        }
    };
    // A fragments of the stage (one MFMA K = 32 atoms): three planes x MT16 row tiles
    auto read_a = [&](int slot, bf16x8 (&a)[3][MT16]) {
        const unsigned base = p_lane + slot * C::STAGE_BYTES;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int mt = 0; mt < MT16; ++mt)
                a[p][mt] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(
                    (const __attribute__((address_space(3))) unsigned char*)(size_t)(base + (p * C::M_BLK + mt * 16) * 64));
    };
    // one component: the leading product into `hi`, the five corrections (smallest first) into `lo`
    auto mfma_comp = [&](auto cc, const bf16x8 (&a)[3][MT16], const bf16x8 (&b1), const bf16x8 (&b2),
                         const bf16x8 (&b3)) {
        constexpr int c = decltype(cc)::value;
#pragma unroll
        for (int mt = 0; mt < MT16; ++mt) {
            f32x4 l = lo[mt][c];
            l = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2][mt], b1, l, 0, 0, 0);
            l = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][mt], b3, l, 0, 0, 0);
            l = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][mt], b2, l, 0, 0, 0);
            l = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][mt], b1, l, 0, 0, 0);
            l = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][mt], b2, l, 0, 0, 0);
            lo[mt][c] = l;
            hi[mt][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][mt], b1, hi[mt][c], 0, 0, 0);
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    // scheduler hints for one chunk (6*MT16 MFMAs): after every MFMA up to `valu` VALU ops, and
    // after each of the first `vmem` MFMAs one LDS-DMA piece
    auto interleave = [&](auto valu_c, auto vmem_c) {
        constexpr int valu = decltype(valu_c)::value, vmem = decltype(vmem_c)::value;
#pragma unroll
        for (int i = 0; i < 6 * MT16; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if constexpr (valu > 0) __builtin_amdgcn_sched_group_barrier(0x002, valu, 0);
            if constexpr (vmem > 0)
                if (i < vmem) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        }
    };
    using N0 = std::integral_constant<int, 0>;
    using NP = std::integral_constant<int, C::P_DMA>;
    using NV = std::integral_constant<int, VP>;
    using V2 = std::integral_constant<int, 2>;

    // split fragments of the CURRENT stage: b[piece][component]
    bf16x8 b1[3], b2[3], b3[3];
    f32x4  raw[C::RAWN];

    if constexpr (C::RING == 3) {
        // ---- 3-deep ring: stages 0 and 1 in flight, stage 0 split -----------------------------
        dma_p(0);
        dma_v(0);
        if (n_stage > 1) {
            dma_p(1);
            dma_v(1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        read_raw(0, raw);
        __builtin_amdgcn_sched_barrier(0);
        split_component<0>(raw, b1[0], b2[0], b3[0]);
        split_component<1>(raw, b1[1], b2[1], b3[1]);
        split_component<2>(raw, b1[2], b2[2], b3[2]);

        // One stage.  NEXT: stage s+1 exists (read + split it); NEXT2: stage s+2 exists (DMA it).
        // Compile-time flags keep the body branch-free so that DMA pieces and split VALU can be
        // interleaved with the MFMAs of the same basic block.
        auto stage = [&](auto next_c, auto next2_c, int slot) {
            constexpr bool next = decltype(next_c)::value, next2 = decltype(next2_c)::value;
            const int      slot1 = slot == C::RING - 1 ? 0 : slot + 1;
            const int      slot2 = slot1 == C::RING - 1 ? 0 : slot1 + 1;
            bf16x8         a[3][MT16];
            bf16x8         n1[3], n2[3], n3[3];
            read_a(slot, a);
            if constexpr (next) read_raw(slot1, raw);
            __builtin_amdgcn_sched_barrier(0);
            // chunk 0
            if constexpr (next2) dma_p(slot2);
            if constexpr (next) split_component<0>(raw, n1[0], n2[0], n3[0]);
            mfma_comp(I0{}, a, b1[0], b2[0], b3[0]);
            interleave(V2{}, NP{});
            __builtin_amdgcn_sched_barrier(0);
            // chunk 1
            if constexpr (next2) dma_v(slot2);
            if constexpr (next) split_component<1>(raw, n1[1], n2[1], n3[1]);
            mfma_comp(I1{}, a, b1[1], b2[1], b3[1]);
            interleave(V2{}, NV{});
            __builtin_amdgcn_sched_barrier(0);
            // chunk 2
            if constexpr (next) split_component<2>(raw, n1[2], n2[2], n3[2]);
            mfma_comp(I2{}, a, b1[2], b2[2], b3[2]);
            interleave(V2{}, N0{});
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (next) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    b1[c] = n1[c];
                    b2[c] = n2[c];
                    b3[c] = n3[c];
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // stage s+2 landed (own pieces)
                __syncthreads();                                   // ... everyone's; slot s is free
            }
        };
        int slot = 0;
        for (int s = 0; s + 2 < n_stage; ++s) {
            stage(std::true_type{}, std::true_type{}, slot);
            slot = slot == C::RING - 1 ? 0 : slot + 1;
        }
        if (n_stage > 1) {
            stage(std::true_type{}, std::false_type{}, slot);
            slot = slot == C::RING - 1 ? 0 : slot + 1;
        }
        stage(std::false_type{}, std::false_type{}, slot);
    } else {
        // ---- 2-deep ring, two workgroups per CU --------------------------------------------
        // read this stage | DMA the next one into the other slot | split | MFMA | wait + barrier.
        // The exposed front of a stage (LDS latency, split, barrier) is covered by the other
        // workgroup resident on the CU.
        dma_p(0);
        dma_v(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        auto stage = [&](auto next_c, int slot) {
            constexpr bool next = decltype(next_c)::value;
            bf16x8         a[3][MT16];
            read_a(slot, a);
            read_raw(slot, raw);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (next) {
                dma_p(slot ^ 1);
                dma_v(slot ^ 1);
            }
            split_component<0>(raw, b1[0], b2[0], b3[0]);
            mfma_comp(I0{}, a, b1[0], b2[0], b3[0]);
            split_component<1>(raw, b1[1], b2[1], b3[1]);
            interleave(V2{}, std::integral_constant<int, C::P_DMA + VP>{});
            __builtin_amdgcn_sched_barrier(0);
            mfma_comp(I1{}, a, b1[1], b2[1], b3[1]);
            split_component<2>(raw, b1[2], b2[2], b3[2]);
            interleave(V2{}, N0{});
            __builtin_amdgcn_sched_barrier(0);
            mfma_comp(I2{}, a, b1[2], b2[2], b3[2]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (next) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
        };
        for (int s = 0; s + 1 < n_stage; ++s) stage(std::true_type{}, s & 1);
        stage(std::false_type{}, (n_stage - 1) & 1);
    }

    // epilogue: register j of lane (r16, q) is row 4q + j, column r16 of its 16x16 tile; rows
    // 2p, 2p+1 are the cos / sin rows of one k -> one complex64 per lane and register pair
    const int     m0 = mb * C::M_BLK;
    const int64_t t = t0 + r16;
    if (t < T) {
#pragma unroll
        for (int mt = 0; mt < MT16; ++mt) {
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const int k = (m0 + mt * 16 + 4 * q + 2 * pr) >> 1;
                if (k < K) {
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        Q[((int64_t)k * 3 + c) * q_stride + t] =
                            make_float2(hi[mt][c][2 * pr] + lo[mt][c][2 * pr],
                                        hi[mt][c][2 * pr + 1] + lo[mt][c][2 * pr + 1]);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Phase table in split form: three bf16 planes in the tile image above.
// Same float32 argument / sincos as phase_table_kernel (kernels_misc.hip); only the storage differs.
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline size_t pb_tile_index(int piece, int m, int a, int m_blk, int n_stage) {
    const int    row = m % m_blk, al = a % K1_BA;
    const size_t tile = ((size_t)(m / m_blk) * n_stage + a / K1_BA) * (3 * (size_t)m_blk * K1_BA);
    const int    sw = (0x78 >> (2 * ((row >> 2) & 3))) & 3;           // g = {0,2,3,1}
    return tile + ((size_t)piece * m_blk + row) * K1_BA + (((al >> 3) ^ sw) << 3) + (al & 7);
}

__global__ void __launch_bounds__(256)
phase_table_split_kernel(const float* __restrict__ kvec, const float* __restrict__ mean_all,
                         const int* __restrict__ idx, __bf16* __restrict__ Pb, int K, int n_g,
                         int A_pad, int M_pad, int m_blk) {
    const int a = blockIdx.y * 256 + threadIdx.x;
    const int k = blockIdx.x;
    if (a >= A_pad || 2 * k >= M_pad) return;
    float cs[2] = {0.f, 0.f};
    if (k < K && a < n_g) {
        const int   src = idx ? idx[a] : a;
        const float rx = mean_all[3 * (size_t)src + 0], ry = mean_all[3 * (size_t)src + 1],
                    rz = mean_all[3 * (size_t)src + 2];
        const float kx = kvec[3 * k + 0], ky = kvec[3 * k + 1], kz = kvec[3 * k + 2];
        const float arg = __fmaf_rn(kz, rz, __fmaf_rn(ky, ry, __fmul_rn(kx, rx)));
        sincosf(arg, &cs[1], &cs[0]);
    }
    const int n_stage = A_pad / K1_BA;
#pragma unroll
    for (int ri = 0; ri < 2; ++ri) {
        const float  x = cs[ri];
        const __bf16 p1 = (__bf16)x;
        const float  r1 = x - (float)p1;
        const __bf16 p2 = (__bf16)r1;
        const float  r2 = r1 - (float)p2;
        const __bf16 p3 = (__bf16)r2;
        Pb[pb_tile_index(0, 2 * k + ri, a, m_blk, n_stage)] = p1;
        Pb[pb_tile_index(1, 2 * k + ri, a, m_blk, n_stage)] = p2;
        Pb[pb_tile_index(2, 2 * k + ri, a, m_blk, n_stage)] = p3;
    }
}

size_t pb_table_bytes(int M_pad, int A_pad) { return (size_t)M_pad * A_pad * 6; }

int launch_phase_table_split(psa_ctx* c, const float* d_kvec, const float* d_mean_all, const int* d_idx,
                             void* d_phase, const ProjGeom& g) {
    dim3 grid(g.M_pad / 2, (g.A_pad + 255) / 256);
    hipLaunchKernelGGL(phase_table_split_kernel, grid, dim3(256), 0, c->stream, d_kvec, d_mean_all, d_idx,
                       (__bf16*)d_phase, g.K, g.n_g, g.A_pad, g.M_pad, g.m_blk);
    PSA_HIP_CHECK(hipGetLastError());
    return PSA_OK;
}

template <int MT16>
static int launch_split_variant(psa_ctx* c, const float* d_v, const void* d_phase, const int* d_idx,
                                float2* d_q, const ProjGeom& g) {
    using C = K1sCfg<MT16, false>;
    const int     n_mblk = g.M_pad / C::M_BLK;
    const int64_t n_tblk = (g.T + C::T_BLK - 1) / C::T_BLK;
    const int64_t grid = ((n_tblk + 7) / 8) * 8 * n_mblk;
    PSA_REQUIRE(grid < (1ll << 31) && n_tblk < (1ll << 31), "projection grid too large");
    const bool contiguous = d_idx == nullptr && g.N_tot % 4 == 0 && g.n_g == g.N_tot;
    if (contiguous)
        hipLaunchKernelGGL((k1_split_kernel<MT16, false>), dim3((unsigned)grid), dim3(256), 0, c->stream, d_v,
                           (const __bf16*)d_phase, d_idx, d_q, g.T, g.q_stride, g.N_tot, g.n_g, g.A_pad, g.K, n_mblk,
                           (int)n_tblk);
    else
        hipLaunchKernelGGL((k1_split_kernel<MT16, true>), dim3((unsigned)grid), dim3(256), 0, c->stream, d_v,
                           (const __bf16*)d_phase, d_idx, d_q, g.T, g.q_stride, g.N_tot, g.n_g, g.A_pad, g.K, n_mblk,
                           (int)n_tblk);
    PSA_HIP_CHECK(hipGetLastError());
    return PSA_OK;
}

// every velocity-mode group; displacement mode (positions - mean while staging) stays on the
// exact-fp32 kernel's register loader
bool k1_split_eligible(const int* d_idx, int64_t N_tot, int64_t n_g, bool displacements) {
    (void)d_idx; (void)N_tot; (void)n_g;
    return !displacements;
}

int k1_split_block_rows(int K) {
    const int M = 2 * K;
    if (M <= 32) return 32;
    if (M <= 64) return 64;
    return 128;
}

int launch_k1_split(psa_ctx* c, const float* d_v, const void* d_phase, const int* d_idx, float2* d_q,
                    const ProjGeom& g) {
    PSA_REQUIRE(g.A_pad % K1_BA == 0 && g.A_pad >= K1_BA, "A_pad must be a positive multiple of %d", K1_BA);
    PSA_REQUIRE(g.M_pad % g.m_blk == 0, "M_pad not a multiple of the M block");
    switch (g.m_blk) {
        case 32:  return launch_split_variant<2>(c, d_v, d_phase, d_idx, d_q, g);
        case 64:  return launch_split_variant<4>(c, d_v, d_phase, d_idx, d_q, g);
        case 128: return launch_split_variant<8>(c, d_v, d_phase, d_idx, d_q, g);
    }
    set_error("no split projection variant for M block %d", g.m_blk);
    return PSA_EINVAL;
}

}  // namespace psa
