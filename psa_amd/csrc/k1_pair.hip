// K1, split-precision form for whole-trajectory groups with 2K > 64: the same projection GEMM as
// k1_mfma.hip, D[m, (c,t)] = sum_a P'[m, a] * d[t, a, c], on the float16 matrix cores with
// fp32-equivalent accuracy ("2 x f16": k1_f16.h), in a workgroup of eight wavefronts, two per SIMD.
//
// Why eight.  Measured on MI355X (tools/k1_experiments.sh, tools/pmc_k1.sh, configuration 3): a
// wavefront streams memory at about 10 bytes per cycle -- every 1-KiB memory instruction (LDS-DMA
// piece or global_load_dwordx4) holds the wavefront that issues it for 100-190 cycles, and nothing
// else of that wavefront issues meanwhile.  A 128 x 64 tile of 32 atoms needs 40 KiB: with four
// wavefronts per CU that is ten instructions each, ~1900 cycles per stage against 1152 of matrix
// work (72 MFMAs), whatever the schedule.  With eight wavefronts it is five each, and while one
// wavefront of a SIMD sits in a memory instruction the other one feeds the matrix pipe.
//
//  - Wavefront w = 4 h + f owns rows [64 h, 64 h + 64) x frames [16 f, 16 f + 16) of the workgroup's
//    128-row x 64-frame tile: 4 row tiles x 3 components, one MFMA chain and one float32 running
//    sum per output (48 + 48 registers).
//  - Per 32-atom stage and wavefront: 2 LDS-DMA pieces of P' (16 KiB, shared by all) and 3 of the
//    6 pieces of its frame group's V rows (6 KiB, shared with the wavefront of the other row half).
//    Three 40-KiB LDS slots; a DMA is issued two stages ahead and awaited with a counted vmcnt
//    that leaves the youngest batch in flight; one s_barrier per stage.
//  - During stage s a wavefront multiplies operands that are in registers, reads its V rows of
//    stage s+1 from LDS and splits them into float16 pieces (k1_f16.h), and refills each A
//    fragment with the next stage's right behind the MFMAs that consumed it.
//  - Accumulation: one MFMA chain per output, corrections first inside a stage; every FOLD = 8
//    stages the chains are added to the float32 sums by VALU and restart from zero (the f16 MFMA
//    keeps ~3 bits below the accumulator's ulp: a 22-bit product added to a sum thousands of
//    times larger is truncated -- tools/probes/mfma_f16_chain.hip; a chain of S stages is biased
//    by <= S * 2^-24 of a fully coherent sum).
//  - The atom axis is padded to 2 stages (64 atoms, zero phase columns); loads past the last stage
//    are clamped to it.
#include "k1_f16.h"

// Timing experiments (tools/k1_experiments.sh builds side libraries with -DPSA_K1_EXPERIMENT=bits;
// results are WRONG by construction, only the kernel time is of interest):
//   1: no DMA in the main loop   2: no MFMAs (operands kept alive)   4: no split   8: no fold
//  16: P' pieces re-read stage 0 (L2 hits)   32: V pieces re-read stage 0
#ifndef PSA_K1_EXPERIMENT
#define PSA_K1_EXPERIMENT 0
#endif

namespace psa {

template <int MT16_, bool GATHER_>
struct K1qCfg {
    static constexpr bool GATHER = GATHER_;        // index-list group (any order, duplicates) or N % 4 != 0
    static constexpr int MT16 = MT16_;             // row tiles of 16 per wavefront: 64 rows (or 32 for short k-lists)
    static constexpr int M_BLK = 32 * MT16;        // two row halves
    static constexpr int T_BLK = 64;               // four frame groups of 16
    static constexpr int FOLD = 8;                 // stages per MFMA chain
    // LDS slots: stage s+1 being read, s+2 .. s+RING in flight (a fourth slot measured no gain:
    // 4.87 vs 4.84 ms on the HBM-bound 64-row variant, 18.6 vs 18.6 ms on configuration 3)
    static constexpr int RING = 3;
    static constexpr int P_STAGE_BYTES = F16x2::NP * M_BLK * K1_BA * 2;    // 16 KiB
    static constexpr int P_DMA = P_STAGE_BYTES / 1024 / 8;                 // pieces per wavefront: 2 (or 1)
    // V image of a frame group.  Whole trajectory: 16 rows x 384 B in 6 pieces of 1 KiB.  Gathered:
    // 8 pieces of (2 frames x 32 atoms x 16 B) -- one (frame, atom) triple is a 12-byte LDS-DMA
    // element, which the hardware lays on a 16-byte pitch (tools/probes/dma12.hip) -- with 16 B of
    // padding per piece that keeps the raw reads at 2-way conflicts.  Either way each wavefront of
    // the pair copies half of the pieces.
    static constexpr int V_PIECE_BYTES = GATHER ? 1024 + 16 : 1024;
    static constexpr int V_DMA = GATHER ? 4 : 3;
    static constexpr int RAW_GROUP_BYTES = 2 * V_DMA * V_PIECE_BYTES;
    static constexpr int RAW_STAGE_BYTES = 4 * RAW_GROUP_BYTES;
    static constexpr int STAGE_BYTES = P_STAGE_BYTES + RAW_STAGE_BYTES;    // 40 (32) KiB; gathered 48.5 (40.5)
    // gathered: a private ring of index rows (64 lanes x 4 B) per wavefront behind the stage slots
    static constexpr int IDX_RING = 3;
    static constexpr int IDX_BASE = RING * STAGE_BYTES;
    static constexpr int LDS_BYTES = IDX_BASE + (GATHER ? 8 * IDX_RING * 256 : 0);
    static constexpr int BATCH = P_DMA + V_DMA + (GATHER ? 1 : 0);    // VMEM instructions per stage and wavefront
    static constexpr int RAWN = GATHER ? 8 : 6;    // 16-byte reads per lane and stage
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

template <int MT16_, bool GATHER_>
__global__ void __launch_bounds__(512, 1)
k1_pair_kernel(const float* __restrict__ V, const _Float16* __restrict__ Pb, const int* __restrict__ idx,
               float2* __restrict__ Q, int64_t T, int64_t q_stride, int64_t N_tot, int n_g, int n_stage, int K, int n_mblk, int n_tblk,
               float vscale, float qscale) {
    using C = K1qCfg<MT16_, GATHER_>;
    constexpr bool GATHER = GATHER_;
    using PR = F16x2;
    using E8 = PR::v8;
    constexpr int NP = PR::NP, MT16 = C::MT16;
    // slot r: [P' tile: NP planes x 128 rows x 32 x 16 bit][V rows: 4 frame groups x 16 rows x 96 float32]
    __shared__ __attribute__((aligned(16))) unsigned char smem[C::LDS_BYTES];
    const unsigned lds0 = (unsigned)(size_t)(lds_u8*)smem;

    // XCD-aware block map: blocks b and b+8 share an XCD (and its L2); they get the M-blocks
    // of one frame tile, so V is fetched from HBM once per frame tile.
    const int b  = blockIdx.x;
    const int r8 = b >> 3;
    const int mb = r8 % n_mblk;
    const int tb = (r8 / n_mblk) * 8 + (b & 7);
    if (tb >= n_tblk) return;

    const int     tid = threadIdx.x, lane = tid & 63;
    const int     w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int     wh = w >> 2, wf = w & 3;                         // row half, frame group
    const int     r16 = lane & 15, q = lane >> 4;                  // frame / row lane, 8-atom group
    const int64_t t0 = (int64_t)tb * C::T_BLK + wf * 16;
    const int     last = n_stage - 1;

    // ---- DMA sources ----------------------------------------------------------------------------
    // Whole trajectory: the frame group's 16 x 24 image of 16-byte slots, pieces 3 wh .. 3 wh + 2 of
    // its 6; the slots of a row are XOR-swizzled on the way in (the source address carries the
    // swizzle, the LDS image is written linearly) so that the 96-byte reads below are conflict-free.
    // Gathered: piece j covers frames 2j, 2j+1 x the stage's 32 atoms; lane l serves atom column
    // l & 31 of frame 2j + (l >> 5), so it keeps 4 row pointers and needs ONE atom index per stage.
    // Rows past the end: a finite filler row.
    const float* vp[C::V_DMA];
#pragma unroll
    for (int j = 0; j < C::V_DMA; ++j) {
        if constexpr (GATHER) {
            int64_t t = t0 + 2 * (wh * C::V_DMA + j) + (lane >> 5);
            if (t >= T) t = T - 1;
            vp[j] = V + t * 3 * N_tot;
        } else {
            const int L = (wh * C::V_DMA + j) * 64 + lane;
            const int row = L / 24, phys = L - row * 24;
            int64_t   t = t0 + row;
            if (t >= T) t = T - 1;
            vp[j] = V + t * 3 * N_tot + 4 * vs_phys_slot(phys, row);
        }
    }
    const unsigned char* pp = reinterpret_cast<const unsigned char*>(Pb) + (size_t)mb * n_stage * C::P_STAGE_BYTES +
                              16 * (w * C::P_DMA * 64 + lane);
    // Gathered: position of this lane's atom in the group at stage st (columns past the group's end
    // carry P' = 0: any valid atom), its index straight from HBM (prologue), and the private LDS
    // ring the index rows of later stages travel through (row k in slot k % 3).
    auto pos_of = [&](int st) {
        int pos = (st < last ? st : last) * K1_BA + (lane & 31);
        return pos < n_g ? pos : n_g - 1;
    };
    const unsigned idx_ring = lds0 + C::IDX_BASE + w * (C::IDX_RING * 256);
    // this wavefront's BATCH for stage st (clamped) -> slot: P' pieces first, then V (gathered: with
    // the atom index `atom`), then the index row of stage st + 2
    auto dma_stage = [&](int st, int slot, int atom) {
        const int      sc = st < last ? st : last;
        const unsigned dst = lds0 + slot * C::STAGE_BYTES;
        const int      scp = (PSA_K1_EXPERIMENT & 16) ? 0 : sc, scv = (PSA_K1_EXPERIMENT & 32) ? 0 : sc;
#pragma unroll
        for (int i = 0; i < C::P_DMA; ++i)
            lds_dma16(pp + (size_t)scp * C::P_STAGE_BYTES + 1024 * i, dst + 1024 * (w * C::P_DMA + i));
        const unsigned vdst = dst + C::P_STAGE_BYTES + wf * C::RAW_GROUP_BYTES + wh * C::V_DMA * C::V_PIECE_BYTES;
#pragma unroll
        for (int j = 0; j < C::V_DMA; ++j) {
            if constexpr (GATHER) lds_dma12(vp[j] + 3 * (int64_t)atom, vdst + j * C::V_PIECE_BYTES);
            else lds_dma16(vp[j] + (size_t)scv * K1_VROW, vdst + j * C::V_PIECE_BYTES);
        }
        if constexpr (GATHER) {
            const int* src = idx ? idx + pos_of(st + 2) : nullptr;
            // no index list (N % 4 != 0): the "index" is the position itself; keep the DMA count uniform
            lds_dma4(idx ? (const void*)src : (const void*)(V + (lane & 31)), idx_ring + ((st + 2) % C::IDX_RING) * 256);
        }
    };
    auto atom_from_ring = [&](int st) -> int {       // index of this lane's atom at stage st
        if constexpr (!GATHER) return 0;
        if (!idx) return pos_of(st);
        return *reinterpret_cast<const __attribute__((address_space(3))) int*>(
            (const lds_u8*)(size_t)(idx_ring + (st % C::IDX_RING) * 256 + 4 * lane));
    };

    // ---- LDS read addresses ---------------------------------------------------------------------
    const int      gsw = (0x78 >> (2 * ((r16 >> 2) & 3))) & 3;     // P' slot swizzle: k1_f16.h
    const unsigned p_lane = lds0 + (wh * (C::M_BLK / 2) + r16) * (K1_BA * 2) + ((q ^ gsw) << 4);
    const unsigned raw_lane = lds0 + C::P_STAGE_BYTES + wf * C::RAW_GROUP_BYTES +
                              (GATHER ? (r16 >> 1) * C::V_PIECE_BYTES + (r16 & 1) * 512 + q * 128 : r16 * (K1_VROW * 4));
    E8             a[NP][MT16];
    E8             bs[2][3][NP];                       // B fragments of stage k: bs[k & 1][component][piece]
    f32x4          raw[C::RAWN];
    f32x4          hi[MT16][3], lo[MT16][3];           // the running MFMA chains / the float32 sums
    auto read_a_tile = [&](int mt, int slot) {
        const unsigned base = p_lane + slot * C::STAGE_BYTES;
#pragma unroll
        for (int p = 0; p < NP; ++p)
            a[p][mt] = *reinterpret_cast<lds_cv8*>((const lds_u8*)(size_t)(base + (p * C::M_BLK + mt * 16) * 64));
    };
    // atoms 8q .. 8q+7 of frame r16: six swizzled 16-byte slots, or (gathered) eight (x,y,z,-) quads
    auto read_raw = [&](int slot) {
#pragma unroll
        for (int j = 0; j < C::RAWN; ++j)
            raw[j] = *reinterpret_cast<lds_cf32x4*>((const lds_u8*)(size_t)(
                raw_lane + slot * C::STAGE_BYTES + (GATHER ? 16 * j : 16 * vs_phys_slot(6 * q + j, r16))));
    };
#pragma unroll
    for (int mt = 0; mt < MT16; ++mt)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            hi[mt][c] = f32x4{0.f, 0.f, 0.f, 0.f};
            lo[mt][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

    // ---- prologue: stages 0 .. RING-1 in flight; stage 0 into registers --------------------------
    {
        // gathered: the first index rows come straight from HBM (plain loads, before any DMA); the
        // batch of stage k carries the index row of stage k + 2
        int a0[C::RING];
#pragma unroll
        for (int k = 0; k < C::RING; ++k) {
            a0[k] = 0;
            if constexpr (GATHER) a0[k] = idx ? idx[pos_of(k)] : pos_of(k);
        }
#pragma unroll
        for (int k = 0; k < C::RING; ++k) dma_stage(k, k, a0[k]);
    }
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((C::RING - 1) * C::BATCH) : "memory");      // stage 0 landed
    read_raw(0);
#pragma unroll
    for (int mt = 0; mt < MT16; ++mt) read_a_tile(mt, 0);
    split_component<0>(raw, vscale, bs[0][0]);
    split_component<1>(raw, vscale, bs[0][1]);
    split_component<2>(raw, vscale, bs[0][2]);
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((C::RING - 2) * C::BATCH) : "memory");   // stage 1 landed, slot 0 read

    // All products of one row tile; MFMAs that depend on one another are two instructions apart.
    auto mfma_tile = [&](int mt, int par, bool restart) {
        if constexpr ((PSA_K1_EXPERIMENT & 2) != 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) asm volatile("" ::"v"(a[0][mt]), "v"(a[1][mt]), "v"(bs[par][c][0]), "v"(bs[par][c][1]));
            return;
        }
        f32x4 ch[3];
#pragma unroll
        for (int c = 0; c < 3; ++c)
            ch[c] = PR::mma(a[1][mt], bs[par][c][0], restart ? f32x4{0.f, 0.f, 0.f, 0.f} : hi[mt][c]);
#pragma unroll
        for (int c = 0; c < 3; ++c) ch[c] = PR::mma(a[0][mt], bs[par][c][1], ch[c]);
#pragma unroll
        for (int c = 0; c < 3; ++c) hi[mt][c] = PR::mma(a[0][mt], bs[par][c][0], ch[c]);
    };
    // One stage (slot = s % RING holds stage s, already in registers; slot1 holds stage s+1, landed).
    //   top: DMA of stage s+RING into slot (nobody reads it any more), V rows of stage s+1 read;
    //   region mt: the tile's 9 MFMAs with a share of the split beside them, then its A fragments
    //   are refilled with those of stage s+1.  (Placing the DMA pieces between the regions, or half
    //   a stage apart in the two wavefronts of a SIMD, measured 3-5 % slower; a fourth slot that
    //   lets the last refill stay in flight across the barrier changed nothing.)
    auto stage = [&](auto par_c, auto restart_c, int s, int slot) {
        constexpr int  par = decltype(par_c)::value;
        constexpr bool restart = decltype(restart_c)::value;
        const int      slot1 = slot == C::RING - 1 ? 0 : slot + 1;
        // gathered: the index row of stage s+RING came with the batch issued two stages ago,
        // awaited by the counted vmcnt at the end of the last stage
        if constexpr ((PSA_K1_EXPERIMENT & 1) == 0) dma_stage(s + C::RING, slot, atom_from_ring(s + C::RING));
        read_raw(slot1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < MT16; ++mt) {
            if constexpr ((PSA_K1_EXPERIMENT & 4) == 0) {
                // the three components beside regions 1, 2, 3 (four row tiles) or 0, 1, 1 (two)
                if (mt == (MT16 >= 4 ? 1 : 0)) split_component<0>(raw, vscale, bs[par ^ 1][0]);
                if (mt == (MT16 >= 4 ? 2 : 1)) split_component<1>(raw, vscale, bs[par ^ 1][1]);
                if (mt == (MT16 >= 4 ? 3 : 1)) split_component<2>(raw, vscale, bs[par ^ 1][2]);
            }
            mfma_tile(mt, par, restart);
#pragma unroll
            for (int i = 0; i < 3 * PR::NTERM; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);      // up to three VALU
            }
            __builtin_amdgcn_sched_barrier(0);
            read_a_tile(mt, slot1);
            __builtin_amdgcn_sched_barrier(0);
        }
        // own pieces of stage s+2 landed (the batch just issued may stay in flight), own LDS reads
        // returned (the next stage's DMA overwrites the slot they read); then everyone's
        if constexpr ((PSA_K1_EXPERIMENT & 1) == 0)
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((C::RING - 2) * C::BATCH) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    int slot = 0;                                      // s % RING
    auto next_slot = [&]() { slot = slot == C::RING - 1 ? 0 : slot + 1; };
    for (int s = 0; s < n_stage;) {                    // n_stage is even; a chain is an even number of stages
        const int len = n_stage - s < C::FOLD ? n_stage - s : C::FOLD;
        stage(I0{}, std::true_type{}, s, slot);
        next_slot();
        stage(I1{}, std::false_type{}, s + 1, slot);
        next_slot();
        for (int i = 2; i < len; i += 2) {
            stage(I0{}, std::false_type{}, s + i, slot);
            next_slot();
            stage(I1{}, std::false_type{}, s + i + 1, slot);
            next_slot();
        }
        if constexpr ((PSA_K1_EXPERIMENT & 8) == 0) {
#pragma unroll
            for (int mt = 0; mt < MT16; ++mt)
#pragma unroll
                for (int c = 0; c < 3; ++c) lo[mt][c] += hi[mt][c];
        }
        s += len;
    }

    // epilogue: register j of lane (r16, q) is row 4q + j, column r16 of its 16x16 tile; rows
    // 2p, 2p+1 are the cos / sin rows of one k -> one complex64 per lane and register pair
    const int     m0 = mb * C::M_BLK + wh * (C::M_BLK / 2);
    const int64_t t = t0 + r16;
    if (t < T) {
#pragma unroll
        for (int mt = 0; mt < MT16; ++mt) {
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const int k = (m0 + mt * 16 + 4 * q + 2 * pr) >> 1;
                if (k < K) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const f32x4 sum = (PSA_K1_EXPERIMENT & 8) ? hi[mt][c] : lo[mt][c];
                        Q[((int64_t)k * 3 + c) * q_stride + t] = make_float2(sum[2 * pr] * qscale, sum[2 * pr + 1] * qscale);
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Phase table in split form: two float16 planes in the tile image the kernel reads.
// Same float32 argument / sincos as phase_table_kernel (kernels_misc.hip); only the storage differs.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
phase_table_f16_kernel(const float* __restrict__ kvec, const float* __restrict__ mean_all, const int* __restrict__ idx,
                       _Float16* __restrict__ Pb, int K, int n_g, int A_pad, int M_pad, int m_blk) {
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
        const float    x = cs[ri] * F16x2::P_SCALE;               // power of two: exact
        const _Float16 lead = (_Float16)x;
        Pb[pf16_tile_index(0, 2 * k + ri, a, m_blk, n_stage)] = lead;
        Pb[pf16_tile_index(1, 2 * k + ri, a, m_blk, n_stage)] = (_Float16)(x - (float)lead);
    }
}

// Padding behind the table: the 128-row kernels prefetch up to RING <= 4 stages past an M block's end (and never
// read them); the 256-row kernel (k1_planes_wide.hip) runs whole periods of 20 stages and MULTIPLIES up to 21
// stages of whatever follows its last M block by zero planes -- those bytes must be finite float16: zeroed here.
constexpr size_t PF16_PAD_BYTES = 22 * (size_t)F16x2::NP * 256 * K1_BA * 2;          // 704 KiB
size_t pf16_table_bytes(int M_pad, int A_pad) { return (size_t)M_pad * A_pad * 2 * F16x2::NP + PF16_PAD_BYTES; }

int launch_phase_table_f16(psa_ctx* c, const float* d_kvec, const float* d_mean_all, const int* d_idx, void* d_phase,
                           const ProjGeom& g) {
    dim3 grid(g.M_pad / 2, (g.A_pad + 255) / 256);
    hipLaunchKernelGGL(phase_table_f16_kernel, grid, dim3(256), 0, c->stream, d_kvec, d_mean_all, d_idx, (_Float16*)d_phase,
                       g.K, g.n_g, g.A_pad, g.M_pad, g.m_blk);
    PSA_HIP_CHECK(hipGetLastError());
    if (g.m_blk == 256)
        PSA_HIP_CHECK(hipMemsetAsync((char*)d_phase + (size_t)g.M_pad * g.A_pad * 2 * F16x2::NP, 0, PF16_PAD_BYTES, c->stream));
    return PSA_OK;
}

// 2^(14-e) with 2^e >= absmax > 2^(e-1): the array's largest magnitude lands in (2^13, 2^14], a
// factor 4 under the float16 maximum.  Returns 0 when the array holds a NaN or Inf (the caller
// then takes a kernel that propagates them as the reference does).
float k1_f16_vscale(unsigned absmax_bits) {
    if (absmax_bits >= 0x7f800000u) return 0.f;
    if (absmax_bits == 0) return 1.f;
    int e = (int)(absmax_bits >> 23) - 127;               // floor(log2), subnormals: -127
    if (absmax_bits & 0x007fffffu) ++e;                    // ceil
    int se = 14 - e;
    if (se > 126) se = 126;                                // tiny data: stay finite (still exact)
    if (se < -126) se = -126;
    union { unsigned u; float f; } s;
    s.u = (unsigned)(se + 127) << 23;
    return s.f;
}

// every velocity-mode group with more than 16 k-vectors (whole trajectory in its own order: row
// DMA; index lists or N % 4 != 0: per-atom gather DMA)
bool k1_pair_eligible(const int* d_idx, int64_t N_tot, int64_t n_g, int64_t K, bool displacements) {
    (void)d_idx; (void)N_tot; (void)n_g;
    return !displacements && 2 * K > 32;
}

int k1_pair_atom_pad(int64_t n_g) { return (int)((n_g + 63) / 64 * 64); }

template <int MT16, bool GATHER>
static int launch_pair_variant(psa_ctx* c, const float* d_v, const void* d_phase, const int* d_idx, float2* d_q,
                               const ProjGeom& g) {
    using C = K1qCfg<MT16, GATHER>;
    const int     n_mblk = g.M_pad / C::M_BLK;
    const int64_t n_tblk = (g.T + C::T_BLK - 1) / C::T_BLK;
    const int64_t grid = ((n_tblk + 7) / 8) * 8 * n_mblk;
    PSA_REQUIRE(grid < (1ll << 31) && n_tblk < (1ll << 31), "projection grid too large");
    const float qscale = 1.f / (g.vscale * F16x2::P_SCALE);           // powers of two: exact
    hipLaunchKernelGGL((k1_pair_kernel<MT16, GATHER>), dim3((unsigned)grid), dim3(512), 0, c->stream, d_v,
                       (const _Float16*)d_phase, d_idx, d_q, g.T, g.q_stride, g.N_tot, g.n_g, g.A_pad / K1_BA, g.K, n_mblk, (int)n_tblk,
                       g.vscale, qscale);
    PSA_HIP_CHECK(hipGetLastError());
    return PSA_OK;
}

// 128-row M blocks; 64-row blocks for k-lists of at most 32 (a one-block launch wastes nothing)
int k1_pair_block_rows(int K) { return 2 * K <= 64 ? 64 : 128; }

int launch_k1_pair(psa_ctx* c, const float* d_v, const void* d_phase, const int* d_idx, float2* d_q, const ProjGeom& g) {
    PSA_REQUIRE((g.m_blk == 128 || g.m_blk == 64) && g.M_pad % g.m_blk == 0, "projection kernel needs 64- or 128-row M blocks");
    PSA_REQUIRE(g.A_pad % (2 * K1_BA) == 0 && g.A_pad > 0, "projection kernel needs the atom axis padded to %d", 2 * K1_BA);
    PSA_REQUIRE(g.vscale > 0.f, "f16 split kernel needs the array's scale");
    const bool contiguous = d_idx == nullptr && g.N_tot % 4 == 0 && g.n_g == g.N_tot;
    if (g.m_blk == 128)
        return contiguous ? launch_pair_variant<4, false>(c, d_v, d_phase, d_idx, d_q, g)
                          : launch_pair_variant<4, true>(c, d_v, d_phase, d_idx, d_q, g);
    return contiguous ? launch_pair_variant<2, false>(c, d_v, d_phase, d_idx, d_q, g)
                      : launch_pair_variant<2, true>(c, d_v, d_phase, d_idx, d_q, g);
}

}  // namespace psa
