// K1, split-precision form: the same projection GEMM as k1_mfma.hip,
//     D[m, (c,t)] = sum_a P'[m, a] * d[t, a, c]
// evaluated on the bf16 matrix cores with fp32-equivalent accuracy ("3 x bf16"):
// every float32 operand x is written x = x1 + x2 + x3 with x1 = bf16(x),
// x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)  (round-to-nearest, residuals exact in
// float32, |x - x1 - x2 - x3| <= 2^-27 |x|), and a product keeps the six leading terms
//     x*y ~= x1*y1 + x1*y2 + x2*y1 + x2*y2 + x1*y3 + x3*y1        (dropped: <= 2^-26 |x*y|)
// accumulated in float32 by v_mfma_f32_16x16x32_bf16.  Six bf16 MFMAs replace sixteen
// float32 MFMA-equivalents: 2.67x the fp32 matrix rate, with one float32 rounding per
// 32-atom MFMA instead of one per FMA.  (SURVEY.md section 7-2 option (ii); accuracy vs
// the exact-fp32 kernel and the oracle: tests/test_gpu_parity.py.)
//
//  - P' is split ONCE by the phase kernel and stored as three bf16 planes in the tile image
//    this kernel DMAs into LDS ([piece][M_BLK][32 atoms], 16-byte slots XOR-swizzled so that
//    the A-fragment ds_read_b128 are conflict-free).
//  - d stays float32 in HBM and in LDS (native (T,N,3) rows, same swizzled image as
//    k1_mfma.hip).  A lane reads its 8 atoms x 3 components = 96 contiguous bytes and splits
//    them in registers (v_cvt_pk_bf16_f32 / shift / subtract: ~4.5 VALU ops per value), which
//    also resolves the (atom, component) interleave.
//  - Every wavefront owns 32 frames x all M_BLK rows (4 wavefronts side by side along t): a V
//    element is split once per workgroup, and because a wavefront's V rows are written (by
//    its own DMA) and read by itself only, the next stage's rows can be read and split BEFORE
//    the stage barrier, in the shadow of the current MFMAs; only the shared P' tile needs the
//    barrier.
//  - MFMA shape: 16x16x32 (one MFMA K = the whole 32-atom stage) rather than 32x32x16: equal
//    cycles per flop, but the chip holds a higher clock on it (measured here: 26.9 vs 28.3 ms
//    on configuration 3; MI355X_MICROARCH.md, DVFS give-back item 7).  The kernel is
//    power/clock-bound: PMC shows the MFMA pipe 73 % busy at ~1.7 GHz.
//  - LDS ring, LDS-DMA, XCD-aware block map are those of k1_mfma.hip.
#include <type_traits>

#include "psa_ctx.h"

namespace psa {

typedef float  f32x16 __attribute__((ext_vector_type(16)));
typedef float  f32x4 __attribute__((ext_vector_type(4)));
typedef float  f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

// Every wavefront owns 32 frames x all M_BLK rows of the workgroup tile (4 wavefronts side by
// side along t), so a V element is split exactly once per workgroup and a wavefront's V rows
// are written (by its own DMA) and read by that wavefront only.
template <int MT>
struct K1sCfg {
    static constexpr int M_BLK = 32 * MT;
    static constexpr int T_BLK = 128;
    static constexpr int V_STAGE_BYTES = T_BLK * K1_VROW * 4;
    static constexpr int P_STAGE_BYTES = 3 * M_BLK * K1_BA * 2;  // three bf16 planes
    static constexpr int LDS_BYTES = 2 * (V_STAGE_BYTES + P_STAGE_BYTES);
    static constexpr int V_DMA = 12;                             // 32 rows x 24 slots / 64 lanes
    static constexpr int P_CHUNKS = P_STAGE_BYTES / 16;
    static constexpr int P_DMA = (P_CHUNKS + 255) / 256;
    static_assert(P_CHUNKS % 64 == 0, "P' tile must be whole wave-instructions");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

__device__ __forceinline__ int vs_phys_slot(int s, int row) { return (s & ~7) | ((s & 7) ^ (row & 7)); }

// hipcc (ROCm 7.2) puts s_waitcnt vmcnt(0) in front of every ds_read it can see while an LDS-DMA is
// in flight (it cannot prove they do not alias).  The stage schedule below therefore keeps every
// LDS read outside the window in which the next stage's DMA is in flight: reads at the stage top
// (previous DMA drained before the barrier) and the early read of the next stage's V rows after
// the explicit vmcnt(0).
template <int OFF, class T>
__device__ __forceinline__ void lds_read128(T& dst, unsigned addr) {
    dst = *reinterpret_cast<const __attribute__((address_space(3))) T*>(
        (const __attribute__((address_space(3))) unsigned char*)(size_t)(addr + OFF));
}
__device__ __forceinline__ void lgkm_wait0() { __builtin_amdgcn_sched_barrier(0); }

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
// raw = 8 atoms x 3 components, component-minor (24 floats).  Component CC of atoms (2p, 2p+1)
template <int E>
__device__ __forceinline__ float relem(const f32x4 (&raw)[6]) { return raw[E >> 2][E & 3]; }
template <int CC>
__device__ __forceinline__ void split_component(const f32x4 (&raw)[6], bf16x8& b1, bf16x8& b2, bf16x8& b3) {
    bf16x2 p1[4], p2[4], p3[4];
    split3(f32x2{relem<0 + CC>(raw), relem<3 + CC>(raw)}, p1[0], p2[0], p3[0]);
    split3(f32x2{relem<6 + CC>(raw), relem<9 + CC>(raw)}, p1[1], p2[1], p3[1]);
    split3(f32x2{relem<12 + CC>(raw), relem<15 + CC>(raw)}, p1[2], p2[2], p3[2]);
    split3(f32x2{relem<18 + CC>(raw), relem<21 + CC>(raw)}, p1[3], p2[3], p3[3]);
    b1 = cat4(p1[0], p1[1], p1[2], p1[3]);
    b2 = cat4(p2[0], p2[1], p2[2], p2[3]);
    b3 = cat4(p3[0], p3[1], p3[2], p3[3]);
}

template <int MT>
__global__ void __launch_bounds__(256, 1)
k1_split_kernel(const float* __restrict__ V, const __bf16* __restrict__ Pb,
                float2* __restrict__ Q, int64_t T, int64_t N_tot, int n_g, int A_pad, int K,
                int n_mblk, int n_tblk) {
    using C = K1sCfg<MT>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[C::LDS_BYTES];
    // [2][T_BLK][96] float32 V, then [2][3][M_BLK][32] bf16 P'
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

    const int b  = blockIdx.x;
    const int r8 = b >> 3;
    const int mb = r8 % n_mblk;
    const int tb = (r8 / n_mblk) * 8 + (b & 7);
    if (tb >= n_tblk) return;

    const int     tid = threadIdx.x, lane = tid & 63;
    const int     wn = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: LDS-DMA
                                                                   // destinations stay in SGPRs
    const int     r16 = lane & 15, q = lane >> 4;     // 16x16x32 MFMA: row/col lane, 8-atom group
    const int64_t t0 = (int64_t)tb * C::T_BLK + wn * 32;          // this wavefront's first frame
    const int     n_stage = A_pad / K1_BA;
    const unsigned char* Pt = reinterpret_cast<const unsigned char*>(Pb) +
                              (size_t)mb * n_stage * C::P_STAGE_BYTES;

    // ---- DMA sources: this wavefront's 32 V rows (12 pieces) + its share of the P' tile -----
    // One pointer per piece, advanced by one stage (32 atoms = 96 floats) after each issue.  The
    // last stage of a trajectory whose atom count is not a multiple of 32 reads up to 31 atoms
    // past the row end: the next frame's data or the zeroed slack behind the array, multiplied by
    // P' = 0.
    const float* vp[C::V_DMA];
#pragma unroll
    for (int j = 0; j < C::V_DMA; ++j) {
        const int L = j * 64 + lane;
        const int row = L / 24, phys = L - row * 24;
        int64_t   t = t0 + row;
        if (t >= T) t = T - 1;                                    // rows past the end: finite filler
        vp[j] = V + t * 3 * N_tot + 4 * vs_phys_slot(phys, row);
    }
    const unsigned char* pp = Pt + 16 * (wn * C::P_DMA * 64 + lane);

    constexpr int MT16 = 2 * MT;                          // 16-row tiles of P'
    f32x4 acc[MT16][2][3];                                // [row tile][frame tile][component]
#pragma unroll
    for (int mt = 0; mt < MT16; ++mt)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[mt][tt][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    // P' slot swizzle for this shape: slot q of row r is stored at q ^ g((r>>2)&3), g = {0,2,3,1}
    // (makes the four 16-lane ds_read_b128 groups of the 16x16x32 A fragment conflict-free)
    const int      gsw = (0x78 >> (2 * ((r16 >> 2) & 3))) & 3;      // g packed two bits each: 0b01_11_10_00 = 0x78
    const unsigned v_lane = lds0 + (wn * 32 + r16) * (K1_VROW * 4);  // row of frame tile 0
    const unsigned p_lane = lds0 + 2 * C::V_STAGE_BYTES + r16 * (K1_BA * 2) + ((q ^ gsw) << 4);

    // raw float32 d of frame tile TT: atoms 8q .. 8q+7 x 3 components = 96 contiguous bytes
    auto read_raw = [&](int buf, int tt, f32x4 (&raw)[6]) {
        const unsigned base = v_lane + buf * C::V_STAGE_BYTES + tt * (16 * K1_VROW * 4);
        const int      s0 = 6 * q;
        lds_read128<0>(raw[0], base + 16 * vs_phys_slot(s0 + 0, r16));
        lds_read128<0>(raw[1], base + 16 * vs_phys_slot(s0 + 1, r16));
        lds_read128<0>(raw[2], base + 16 * vs_phys_slot(s0 + 2, r16));
        lds_read128<0>(raw[3], base + 16 * vs_phys_slot(s0 + 3, r16));
        lds_read128<0>(raw[4], base + 16 * vs_phys_slot(s0 + 4, r16));
        lds_read128<0>(raw[5], base + 16 * vs_phys_slot(s0 + 5, r16));
    };
    // A fragments of the stage (K = 32 atoms): three planes x MT16 row tiles
    auto read_a = [&](int buf, bf16x8 (&a)[3][MT16]) {
        const unsigned base = p_lane + buf * C::P_STAGE_BYTES;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int mt = 0; mt < MT16; ++mt)
                a[p][mt] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(
                    (const __attribute__((address_space(3))) unsigned char*)(size_t)(base + (p * C::M_BLK + mt * 16) * 64));
    };
    // six products per (row tile, component CC), smallest terms first
    auto mfma_comp = [&](auto cc, auto ttc, const bf16x8 (&a)[3][MT16], const bf16x8 (&bq)[3][3]) {
        constexpr int c = decltype(cc)::value, tt = decltype(ttc)::value;
#pragma unroll
        for (int mt = 0; mt < MT16; ++mt) {
            f32x4 d = acc[mt][tt][c];
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2][mt], bq[0][c], d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][mt], bq[2][c], d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][mt], bq[1][c], d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][mt], bq[0][c], d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][mt], bq[1][c], d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][mt], bq[0][c], d, 0, 0, 0);
            acc[mt][tt][c] = d;
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    // interleave hints for the scheduler inside one chunk (= 6*MT MFMAs): after every MFMA
    // allow `valu` VALU ops, and after every `per`-th MFMA one LDS-DMA piece
    auto interleave = [&](auto valu_c, auto vmem_c) {
        constexpr int valu = decltype(valu_c)::value, vmem = decltype(vmem_c)::value;
#pragma unroll
        for (int i = 0; i < 12 * MT; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if constexpr (valu > 0) __builtin_amdgcn_sched_group_barrier(0x002, valu, 0);
            if constexpr (vmem > 0)
                if (i < vmem) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        }
    };
    using N0 = std::integral_constant<int, 0>;

    // DMA of the next stage, in three parts so that each can ride in the shadow of one chunk
    auto dma_p = [&](int buf) {
        unsigned char* dst = smem + 2 * C::V_STAGE_BYTES + buf * C::P_STAGE_BYTES;
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
    auto dma_v = [&](int buf, auto j0c) {
        constexpr int j0 = decltype(j0c)::value;
        float* vd = reinterpret_cast<float*>(smem + buf * C::V_STAGE_BYTES) + wn * 32 * K1_VROW;
#pragma unroll
        for (int j = j0; j < j0 + 6; ++j) {
            __builtin_amdgcn_global_load_lds((gbl_void*)vp[j], (lds_void*)(vd + j * 256), 16, 0, 0);
            vp[j] += K1_VROW;
        }
    };
    using J0 = std::integral_constant<int, 0>;
    using J6 = std::integral_constant<int, 6>;
    using NP = std::integral_constant<int, C::P_DMA>;
    using N6 = std::integral_constant<int, 6>;
    using V3 = std::integral_constant<int, 3>;
    using V5 = std::integral_constant<int, 5>;

    // ---- software pipeline ----------------------------------------------------------------
    // A wavefront's V rows are its own (written by its own DMA, read by itself), so the raw
    // data of the NEXT stage's k-step 0 can be read and split before the stage barrier, in the
    // shadow of this stage's last MFMAs; only the shared P' tile needs the barrier.
    //   [barrier]  read A(ks0), A(ks1)
    //   chunk A: MFMA ks0 c0 | DMA P'(s+1)            | read raw(ks1)
    //   chunk B: MFMA ks0 c1 | DMA V(s+1) pieces 0-5  | split ks1 comp 0
    //   chunk C: MFMA ks0 c2 | DMA V(s+1) pieces 6-11 | split ks1 comp 1,2
    //   chunk D: MFMA ks1 c0
    //   vmcnt(0): own DMA landed -> read raw(ks0) of stage s+1
    //   chunk E: MFMA ks1 c1
    //   chunk F: MFMA ks1 c2 | split ks0 of stage s+1
    //   [barrier]
    // same pipeline as the 32x32x16 form with "k-step" replaced by "frame tile": one MFMA K
    // covers the whole 32-atom stage, the wavefront's 32 frames are two 16-column tiles.
    bf16x8 bq0[3][3], bq1[3][3];
    f32x4  raw[6];

    dma_p(0);
    dma_v(0, J0{});
    dma_v(0, J6{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    read_raw(0, 0, raw);
    __builtin_amdgcn_sched_barrier(0);
    split_component<0>(raw, bq0[0][0], bq0[1][0], bq0[2][0]);
    split_component<1>(raw, bq0[0][1], bq0[1][1], bq0[2][1]);
    split_component<2>(raw, bq0[0][2], bq0[1][2], bq0[2][2]);

    auto stage = [&](auto more_c, int s) {
        constexpr bool more = decltype(more_c)::value;
        const int      buf = s & 1;
        bf16x8         a[3][MT16];
        read_a(buf, a);
        read_raw(buf, 1, raw);
        __builtin_amdgcn_sched_barrier(0);
        // chunk A
        if constexpr (more) dma_p(buf ^ 1);
        mfma_comp(I0{}, I0{}, a, bq0);
        interleave(N0{}, NP{});
        __builtin_amdgcn_sched_barrier(0);
        // chunk B
        if constexpr (more) dma_v(buf ^ 1, J0{});
        split_component<0>(raw, bq1[0][0], bq1[1][0], bq1[2][0]);
        mfma_comp(I1{}, I0{}, a, bq0);
        interleave(V3{}, N6{});
        __builtin_amdgcn_sched_barrier(0);
        // chunk C
        if constexpr (more) dma_v(buf ^ 1, J6{});
        split_component<1>(raw, bq1[0][1], bq1[1][1], bq1[2][1]);
        split_component<2>(raw, bq1[0][2], bq1[1][2], bq1[2][2]);
        mfma_comp(I2{}, I0{}, a, bq0);
        interleave(V5{}, N6{});
        __builtin_amdgcn_sched_barrier(0);
        // chunk D
        mfma_comp(I0{}, I1{}, a, bq1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (more) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // own DMA of stage s+1 landed
            read_raw(buf ^ 1, 0, raw);                            // own V rows: no barrier needed
        }
        __builtin_amdgcn_sched_barrier(0);
        // chunk E
        mfma_comp(I1{}, I1{}, a, bq1);
        __builtin_amdgcn_sched_barrier(0);
        // chunk F
        if constexpr (more) {
            split_component<0>(raw, bq0[0][0], bq0[1][0], bq0[2][0]);
            split_component<1>(raw, bq0[0][1], bq0[1][1], bq0[2][1]);
            split_component<2>(raw, bq0[0][2], bq0[1][2], bq0[2][2]);
        }
        mfma_comp(I2{}, I1{}, a, bq1);
        interleave(V5{}, N0{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (more) __syncthreads();
    };
    for (int s = 0; s + 1 < n_stage; ++s) stage(std::true_type{}, s);
    stage(std::false_type{}, n_stage - 1);

    // epilogue: accumulator register j of lane (r16, q) is row 4q + j, column r16 of its tile
    const int m0 = mb * C::M_BLK;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int64_t t = t0 + tt * 16 + r16;
        if (t < T) {
#pragma unroll
            for (int mt = 0; mt < MT16; ++mt) {
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    const int k = (m0 + mt * 16 + 4 * q + 2 * pr) >> 1;
                    if (k < K) {
#pragma unroll
                        for (int c = 0; c < 3; ++c)
                            Q[((int64_t)k * 3 + c) * T + t] =
                                make_float2(acc[mt][tt][c][2 * pr], acc[mt][tt][c][2 * pr + 1]);
                    }
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

template <int MT>
static int launch_split_variant(psa_ctx* c, const float* d_v, const void* d_phase, float2* d_q,
                                const ProjGeom& g) {
    using C = K1sCfg<MT>;
    const int     n_mblk = g.M_pad / C::M_BLK;
    const int64_t n_tblk = (g.T + C::T_BLK - 1) / C::T_BLK;
    const int64_t grid = ((n_tblk + 7) / 8) * 8 * n_mblk;
    PSA_REQUIRE(grid < (1ll << 31) && n_tblk < (1ll << 31), "projection grid too large");
    hipLaunchKernelGGL((k1_split_kernel<MT>), dim3((unsigned)grid), dim3(256), 0, c->stream, d_v,
                       (const __bf16*)d_phase, d_q, g.T, g.N_tot, g.n_g, g.A_pad, g.K, n_mblk, (int)n_tblk);
    PSA_HIP_CHECK(hipGetLastError());
    return PSA_OK;
}

bool k1_split_eligible(const int* d_idx, int64_t N_tot, int64_t n_g, bool displacements) {
    return !displacements && d_idx == nullptr && (N_tot % 4 == 0) && n_g == N_tot;
}

int k1_split_block_rows(int K) {
    const int M = 2 * K;
    if (M <= 32) return 32;
    if (M <= 64) return 64;
    return 128;
}

int launch_k1_split(psa_ctx* c, const float* d_v, const void* d_phase, float2* d_q, const ProjGeom& g) {
    PSA_REQUIRE(g.A_pad % K1_BA == 0 && g.A_pad >= K1_BA, "A_pad must be a positive multiple of %d", K1_BA);
    PSA_REQUIRE(g.M_pad % g.m_blk == 0, "M_pad not a multiple of the M block");
    switch (g.m_blk) {
        case 32:  return launch_split_variant<1>(c, d_v, d_phase, d_q, g);
        case 64:  return launch_split_variant<2>(c, d_v, d_phase, d_q, g);
        case 128: return launch_split_variant<4>(c, d_v, d_phase, d_q, g);
    }
    set_error("no split projection variant for M block %d", g.m_blk);
    return PSA_EINVAL;
}

}  // namespace psa
