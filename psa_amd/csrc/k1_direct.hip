// K1, split-precision form for whole-trajectory groups with 2K > 64: the same projection GEMM as
// k1_mfma.hip, D[m, (c,t)] = sum_a P'[m, a] * d[t, a, c], on the float16 matrix cores with
// fp32-equivalent accuracy ("2 x f16": k1_f16.h), organised around what one wavefront per SIMD
// can issue.
//
// Measured on MI355X (tools/k1_experiments.sh, MI355X_MICROARCH.md "Per-instruction cycle
// constants"): with one 512-register wavefront per SIMD nothing overlaps for free -- every
// instruction of the wavefront costs issue cycles (VALU / SALU / s_nop 4, MFMA 8 of its 16, an
// LDS-DMA piece 60-185), and the matrix pipe only stays busy while the instructions between two
// MFMAs fit in the 8 cycles the MFMA leaves.  The LDS-staged kernel spent as long issuing its ten
// DMA pieces per stage as on its 72 MFMAs.  Hence:
//
//  - d (V) never touches LDS.  A lane needs, per 32-atom stage, the 96 contiguous bytes
//    (8 atoms x 3 components) of ITS frame row: six global_load_dwordx4 straight into registers,
//    issued two stages ahead (the wavefront's 16 rows x 384 B are read as 64-byte pieces; both
//    halves of a 128-byte line are fetched by neighbouring instructions and meet in the L1).
//    No DMA, no LDS round trip, no LDS space, nothing shared between wavefronts.
//  - P' (shared by the four wavefronts) goes global -> registers -> LDS: four global_load_dwordx4
//    + four ds_write_b128 per wavefront and stage, ~17 issue cycles per KiB instead of 60-185.
//    Three 16-KiB slots; one s_barrier per stage.
//  - A fragments are read through a window of 4 row tiles, refilled behind the MFMAs that consumed
//    them (tile mt+4 of this stage or tile mt-4 of the next): every LDS read has >= 2 tiles
//    (18 MFMAs) to return and A costs 32 registers instead of 64.
//  - Split of the next stage's V rows (v_mul, v_cvt_pk_f16_f32, v_fma_mix_f32: 3 single-issue ops
//    per value; packed-f32 VALU is an anti-lever beside MFMAs, so the file is built without SLP
//    vectorisation) and the chain folds are dealt out to the four 2-tile regions of a stage.
//  - One MFMA chain per output, corrections first inside a stage; every FOLD = 8 stages the chain
//    is added to a float32 running sum and restarted from zero (the f16 MFMA keeps ~3 bits below
//    the accumulator's ulp: a 22-bit product added to a sum thousands of times larger is
//    truncated -- tools/probes/mfma_f16_chain.hip; a chain of S stages is biased by <= S * 2^-24
//    of a fully coherent sum -- measured on the long chain: a quarter of the bound).  All chains
//    start in the same stage and end in the same stage, so no accumulator is live across the
//    loop's back edge (a staggered schedule made the register allocator shuffle accumulators
//    there).
//  - The running sums live in LDS (96 KiB: 96 floats per lane).  The last MFMA of a chain is
//    consumed by VALU right behind it, so the compiler lets it write VGPRs; the fold is
//    ds_read_b128 + 4 v_add_f32 + ds_write_b128 per (tile, component).  Two things NOT to do
//    (tools/k1_experiments.sh, configuration 3): reading long-lived accumulators with
//    v_accvgpr_read for a VALU add cost 5.5 ms of 23.5; ds_add_f32 straight from the accumulator
//    registers took the kernel from 23 to 141 ms (LDS float atomics run at about a lane per cycle).
//  - The atom axis is padded to 8 stages (256 atoms, zero phase columns) and loads past the last
//    stage are clamped to it: the main loop has no tail code.
#include "k1_f16.h"

// Timing experiments (tools/k1_experiments.sh builds side libraries with -DPSA_K1_EXPERIMENT=bits;
// results are WRONG by construction, only the kernel time is of interest):
//   1: V always loaded from stage 0 (cache hits)   2: no MFMAs (operands kept alive)
//   4: P' always loaded from stage 0               8: no fold
//  16: no split (B fragments opaque, V loads kept) 32: no s_barrier
//  64: no V loads at all (rows opaque, split kept)
#ifndef PSA_K1_EXPERIMENT
#define PSA_K1_EXPERIMENT 0
#endif

namespace psa {

struct K1dCfg {
    static constexpr int MT16 = 8;                 // row tiles of 16: M_BLK = 128
    static constexpr int M_BLK = 16 * MT16;
    static constexpr int T_BLK = 64;               // 4 wavefronts x 16 frames
    static constexpr int AW = 4;                   // A window, row tiles
    static constexpr int FOLD = 8;                 // stages per MFMA chain (two groups of 4)
    static constexpr int P_RING = 3;
    static constexpr int P_STAGE_BYTES = F16x2::NP * M_BLK * K1_BA * 2;    // 16 KiB
    static constexpr int P_LOADS = P_STAGE_BYTES / 16 / 256;               // per lane and stage: 4
    static constexpr int SUM_BYTES = MT16 * 3 * 256 * 16;                  // running sums: [tile, component][thread] x 16 B
    static constexpr int LDS_BYTES = P_RING * P_STAGE_BYTES + SUM_BYTES;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

typedef const __attribute__((address_space(3))) F16x2::v8 lds_v8;
typedef __attribute__((address_space(3))) f32x4            lds_f32x4;

__global__ void __launch_bounds__(256, 1)
k1_direct_kernel(const float* __restrict__ V, const _Float16* __restrict__ Pb, float2* __restrict__ Q, int64_t T,
                 int64_t N_tot, int n_stage, int K, int n_mblk, int n_tblk, float vscale, float qscale) {
    using C = K1dCfg;
    using PR = F16x2;
    using E8 = PR::v8;
    constexpr int NP = PR::NP, MT16 = C::MT16, AW = C::AW;
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
    const int     wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int     r16 = lane & 15, q = lane >> 4;                  // frame / row lane, 8-atom group
    const int64_t t0 = (int64_t)tb * C::T_BLK + wn * 16;          // this wavefront's first frame
    const int     last = n_stage - 1;

    // this lane's V row (frames past the end: a finite filler row) at its 8-atom group
    int64_t trow = t0 + r16;
    if (trow >= T) trow = T - 1;
    const f32x4* vrow = reinterpret_cast<const f32x4*>(V + trow * 3 * N_tot + 24 * q);
    // this wavefront's quarter of the P' tile images of M-block mb (16-byte chunks)
    const f32x4* ptile = reinterpret_cast<const f32x4*>(Pb) + (size_t)mb * n_stage * (C::P_STAGE_BYTES / 16) +
                         wn * (C::P_LOADS * 64) + lane;
    const unsigned p_store = lds0 + 16 * (wn * C::P_LOADS * 64 + lane);

    auto load_v = [&](int st, f32x4 (&raw)[6]) {       // stage st (clamped): 96 contiguous bytes
        const f32x4* src = vrow + (size_t)(st < last ? st : last) * (K1_VROW / 4);
#pragma unroll
        for (int j = 0; j < 6; ++j) raw[j] = src[j];
    };
    auto load_p = [&](int st, f32x4 (&pst)[C::P_LOADS]) {
        const f32x4* src = ptile + (size_t)(st < last ? st : last) * (C::P_STAGE_BYTES / 16);
#pragma unroll
        for (int j = 0; j < C::P_LOADS; ++j) pst[j] = src[64 * j];
    };
    auto store_p = [&](int slot, const f32x4 (&pst)[C::P_LOADS]) {
#pragma unroll
        for (int j = 0; j < C::P_LOADS; ++j)
            *reinterpret_cast<lds_f32x4*>((__attribute__((address_space(3))) unsigned char*)(size_t)(
                p_store + slot * C::P_STAGE_BYTES + 1024 * j)) = pst[j];
    };

    // P' slot swizzle: see k1_f16.h (pf16_tile_index); this lane's row r16 / k-group q of a row tile
    const int      gsw = (0x78 >> (2 * ((r16 >> 2) & 3))) & 3;
    const unsigned p_lane = lds0 + r16 * (K1_BA * 2) + ((q ^ gsw) << 4);
    E8             a[NP][AW];                          // A window: tile mt lives in entry mt % AW
    auto read_a_tile = [&](int mt, int slot) {
        const unsigned base = p_lane + slot * C::P_STAGE_BYTES;
#pragma unroll
        for (int p = 0; p < NP; ++p)
            a[p][mt % AW] = *reinterpret_cast<lds_v8*>(
                (const __attribute__((address_space(3))) unsigned char*)(size_t)(base + (p * C::M_BLK + mt * 16) * 64));
    };

    // hi: the running MFMA chains; sums: the float32 sums of the finished chains, in LDS as
    // [tile, component][thread] x 16 bytes (conflict-free b128 accesses, private to the thread)
    f32x4          hi[MT16][3];
    const unsigned sums = lds0 + C::P_RING * C::P_STAGE_BYTES + 16 * tid;
    auto           sum_at = [&](int mt, int c) {
        return reinterpret_cast<lds_f32x4*>(
            (__attribute__((address_space(3))) unsigned char*)(size_t)(sums + (mt * 3 + c) * (256 * 16)));
    };
#pragma unroll
    for (int mt = 0; mt < MT16; ++mt)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            hi[mt][c] = f32x4{0.f, 0.f, 0.f, 0.f};
            *sum_at(mt, c) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    auto fold_tile = [&](int mt) {                     // per component: ds_read_b128, 4 v_add, ds_write_b128
#pragma unroll
        for (int c = 0; c < 3; ++c) *sum_at(mt, c) = *sum_at(mt, c) + hi[mt][c];
    };
    f32x4 raws[4][6];                                  // V rows of stage k live in raws[k & 3]
    E8    bcs[2][3][NP];                               // split B of stage k: bcs[k & 1][component][piece]

    // ---- prologue: P'(0), P'(1) into LDS, P'(2) and V(0..2) into registers, stage 0 split --------
    f32x4 psts[2][C::P_LOADS];                         // P' of stage k travels in psts[k & 1]
    {
        f32x4 p1[C::P_LOADS];
        load_p(0, psts[0]);
        load_p(1, p1);
        load_v(0, raws[0]);
        load_v(1, raws[1]);
        load_v(2, raws[2]);
        store_p(0, psts[0]);
        store_p(1, p1);
        load_p(2, psts[0]);
    }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < AW; ++mt) read_a_tile(mt, 0);
    split_component<0>(raws[0], vscale, bcs[0][0]);
    split_component<1>(raws[0], vscale, bcs[0][1]);
    split_component<2>(raws[0], vscale, bcs[0][2]);

    // All products of one row tile; MFMAs that depend on one another are two instructions apart.
    // RESTART: the chain begins at zero (its previous value has just been folded).
    auto mfma_tile = [&](int mt, int par, bool restart) {
        if constexpr ((PSA_K1_EXPERIMENT & 2) != 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
                asm volatile("" ::"v"(a[0][mt % AW]), "v"(a[1][mt % AW]), "v"(bcs[par][c][0]), "v"(bcs[par][c][1]));
            return;
        }
        f32x4 ch[3];
#pragma unroll
        for (int c = 0; c < 3; ++c)
            ch[c] = PR::mma(a[1][mt % AW], bcs[par][c][0], restart ? f32x4{0.f, 0.f, 0.f, 0.f} : hi[mt][c]);
#pragma unroll
        for (int c = 0; c < 3; ++c) ch[c] = PR::mma(a[0][mt % AW], bcs[par][c][1], ch[c]);
#pragma unroll
        for (int c = 0; c < 3; ++c) hi[mt][c] = PR::mma(a[0][mt % AW], bcs[par][c][0], ch[c]);
    };

    // One stage (PH = s % 4, compile-time: register sets, which chains restart, where the work goes).
    //   region 0: tiles 0,1 | P'(s+3) loads issued                  |
    //   region 1: tiles 2,3 | V(s+3) loads issued                   | split component 0 of stage s+1
    //   region 2: tiles 4,5 | P'(s+2) (loaded a stage ago) -> LDS   | split component 1
    //   region 3: tiles 6,7 |                                       | split component 2
    // RESTART (a group's first stage, PH == 0): every chain starts from zero; FOLD (its last stage,
    // PH == 3): the tiles of region r are folded in region r + 1, those of region 3 behind it.
    auto stage = [&](auto ph_c, auto restart_c, auto fold_c, int s, int slot) {
        constexpr int  ph = decltype(ph_c)::value, par = ph & 1;
        constexpr bool restart = decltype(restart_c)::value && ph == 0;
        constexpr bool fold = decltype(fold_c)::value && ph == 3 && (PSA_K1_EXPERIMENT & 8) == 0;
        const int     slot1 = slot == 2 ? 0 : slot + 1, slot2 = slot1 == 2 ? 0 : slot1 + 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int vmem = 0, ds = 0;
            if (r == 0) {
                load_p((PSA_K1_EXPERIMENT & 4) ? 0 : s + 3, psts[par ^ 1]);
                vmem = C::P_LOADS;
            }
            if (r == 1) {
                if constexpr ((PSA_K1_EXPERIMENT & 64) != 0) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) asm volatile("" : "+v"(raws[(ph + 3) & 3][j]));
                } else {
                    load_v((PSA_K1_EXPERIMENT & 1) ? 0 : s + 3, raws[(ph + 3) & 3]);
                    vmem = 6;
                }
            }
            if (r == 2) {
                store_p(slot2, psts[par]);
                ds = C::P_LOADS;
            }
            if (fold && r > 0) {
                fold_tile(2 * r - 2);
                fold_tile(2 * r - 1);
                ds += 12;
            }
            if constexpr ((PSA_K1_EXPERIMENT & 16) == 0) {
                if (r == 1) split_component<0>(raws[(ph + 1) & 3], vscale, bcs[par ^ 1][0]);
                if (r == 2) split_component<1>(raws[(ph + 1) & 3], vscale, bcs[par ^ 1][1]);
                if (r == 3) split_component<2>(raws[(ph + 1) & 3], vscale, bcs[par ^ 1][2]);
            } else if (r == 3) {
#pragma unroll
                for (int j = 0; j < 6; ++j) asm volatile("" ::"v"(raws[(ph + 1) & 3][j]));
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int i = 0; i < NP; ++i) asm volatile("" : "+v"(bcs[par ^ 1][c][i]));
            }
            mfma_tile(2 * r, par, restart);
            mfma_tile(2 * r + 1, par, restart);
#pragma unroll
            for (int i = 0; i < 2 * 3 * PR::NTERM; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                 // up to two VALU
                if (i < vmem) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // a VMEM read
                if (i < ds) __builtin_amdgcn_sched_group_barrier(0x080, 1, 0);     // a DS op (store / fold)
            }
            __builtin_amdgcn_sched_barrier(0);
            // window refill behind the two tiles: same stage (slot) for r < 2, next stage after
            if (r < 2) {
                read_a_tile(2 * r + AW, slot);
                read_a_tile(2 * r + 1 + AW, slot);
            } else {
                read_a_tile(2 * r - AW, slot1);
                read_a_tile(2 * r + 1 - AW, slot1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (fold) {
            fold_tile(MT16 - 2);
            fold_tile(MT16 - 1);
        }
        // P'(s+2) stores of this wavefront are done (they precede the refills of regions 2, 3 and
        // the folds in the LDS queue: at least those 2 * 2 * NP reads are behind them); then
        // everyone's
        if constexpr ((PSA_K1_EXPERIMENT & 32) != 0)
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(4 * NP) : "memory");
        else
            asm volatile("s_waitcnt lgkmcnt(%0)\n\ts_barrier" ::"n"(4 * NP) : "memory");
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    int slot = 0;
    auto next_slot = [&]() { slot = slot == 2 ? 0 : slot + 1; };
    auto group = [&](auto restart_c, auto fold_c, int s) {     // four stages: one turn of the register sets
        stage(I0{}, restart_c, fold_c, s, slot);
        next_slot();
        stage(I1{}, restart_c, fold_c, s + 1, slot);
        next_slot();
        stage(I2{}, restart_c, fold_c, s + 2, slot);
        next_slot();
        stage(I3{}, restart_c, fold_c, s + 3, slot);
        next_slot();
    };
    static_assert(C::FOLD == 8, "one chain = a restarting and a folding group");
    for (int s = 0; s < n_stage; s += C::FOLD) {
        group(std::true_type{}, std::false_type{}, s);
        group(std::false_type{}, std::true_type{}, s + 4);
    }

    // epilogue: register j of lane (r16, q) is row 4q + j, column r16 of its 16x16 tile; rows
    // 2p, 2p+1 are the cos / sin rows of one k -> one complex64 per lane and register pair.
    // The last stage has folded every chain; a lane reads back its own sums.
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
                    for (int c = 0; c < 3; ++c) {
                        const f32x4 sum = *sum_at(mt, c);
                        const float re = sum[2 * pr], im = sum[2 * pr + 1];
                        Q[((int64_t)k * 3 + c) * T + t] = make_float2(re * qscale, im * qscale);
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
phase_table_f16_kernel(const float* __restrict__ kvec, const float* __restrict__ mean_all, _Float16* __restrict__ Pb,
                       int K, int n_g, int A_pad, int M_pad, int m_blk) {
    const int a = blockIdx.y * 256 + threadIdx.x;
    const int k = blockIdx.x;
    if (a >= A_pad || 2 * k >= M_pad) return;
    float cs[2] = {0.f, 0.f};
    if (k < K && a < n_g) {
        const float rx = mean_all[3 * (size_t)a + 0], ry = mean_all[3 * (size_t)a + 1], rz = mean_all[3 * (size_t)a + 2];
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

size_t pf16_table_bytes(int M_pad, int A_pad) { return (size_t)M_pad * A_pad * 2 * F16x2::NP; }

int launch_phase_table_f16(psa_ctx* c, const float* d_kvec, const float* d_mean_all, void* d_phase, const ProjGeom& g) {
    dim3 grid(g.M_pad / 2, (g.A_pad + 255) / 256);
    hipLaunchKernelGGL(phase_table_f16_kernel, grid, dim3(256), 0, c->stream, d_kvec, d_mean_all, (_Float16*)d_phase,
                       g.K, g.n_g, g.A_pad, g.M_pad, g.m_blk);
    PSA_HIP_CHECK(hipGetLastError());
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

// whole trajectory in its own order (the kernel reads a lane's 96 bytes straight from its frame row)
bool k1_direct_eligible(const int* d_idx, int64_t N_tot, int64_t n_g, int64_t K, bool displacements) {
    return !displacements && d_idx == nullptr && n_g == N_tot && N_tot % 4 == 0 && 2 * K > 64;
}

int k1_direct_atom_pad(int64_t n_g) {
    constexpr int pad = K1_BA * K1dCfg::FOLD;
    return (int)((n_g + pad - 1) / pad * pad);
}

int launch_k1_direct(psa_ctx* c, const float* d_v, const void* d_phase, float2* d_q, const ProjGeom& g) {
    using C = K1dCfg;
    PSA_REQUIRE(g.m_blk == C::M_BLK && g.M_pad % C::M_BLK == 0, "direct projection kernel needs 128-row M blocks");
    PSA_REQUIRE(g.A_pad % (K1_BA * C::FOLD) == 0 && g.A_pad > 0, "direct projection kernel needs the atom axis padded to %d",
                K1_BA * C::FOLD);
    PSA_REQUIRE(g.n_g == g.N_tot && g.N_tot % 4 == 0, "direct projection kernel takes whole-trajectory groups, N %% 4 == 0");
    PSA_REQUIRE(g.vscale > 0.f, "f16 split kernel needs the array's scale");
    const int     n_mblk = g.M_pad / C::M_BLK;
    const int64_t n_tblk = (g.T + C::T_BLK - 1) / C::T_BLK;
    const int64_t grid = ((n_tblk + 7) / 8) * 8 * n_mblk;
    PSA_REQUIRE(grid < (1ll << 31) && n_tblk < (1ll << 31), "projection grid too large");
    const float qscale = 1.f / (g.vscale * F16x2::P_SCALE);           // powers of two: exact
    hipLaunchKernelGGL(k1_direct_kernel, dim3((unsigned)grid), dim3(256), 0, c->stream, d_v, (const _Float16*)d_phase,
                       d_q, g.T, g.N_tot, g.A_pad / K1_BA, g.K, n_mblk, (int)n_tblk, g.vscale, qscale);
    PSA_HIP_CHECK(hipGetLastError());
    return PSA_OK;
}

}  // namespace psa
