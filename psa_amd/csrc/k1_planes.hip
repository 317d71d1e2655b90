// K1 on pre-split planes: the projection GEMM of k1_pair.hip ("2 x f16", k1_f16.h),
//     D[m, (c,t)] = sum_a P'[m, a] * d[t, a, c],
// for a group whose data has been scaled, split into its two float16 pieces and laid out in
// MFMA-fragment order ONCE (split_planes_kernel below; plane_index in k1_f16.h), instead of on
// every launch in the inner loop.  What that removes from the stage of k1_pair_kernel: the raw
// 96-byte de-interleaving LDS reads, 3 VALU operations per value (v_mul, v_cvt_pk_f16, v_fma_mix)
// and their registers; what it adds: nothing -- the planes have the float32 array's 4 bytes per
// value, a stage is still 40 KiB of LDS-DMA (16 KiB P' + 4 x 6 KiB V), and the B fragments are
// ds_read_b128 of the image as it lands.  An index-list or type group is COMPACTED by the split
// (its atoms become contiguous columns), so every group runs the same row-DMA kernel: no gather
// variant, no per-stage index traffic.
//
// Work decomposition as in k1_pair.hip: eight wavefronts, two per SIMD; wavefront w = 4 h + f owns
// rows [M_BLK/2 * h, +M_BLK/2) x frames [16 f, 16 f + 16); per 32-atom stage it copies its share of
// the P' tile and 3 of the 6 blocks of its frame group's V planes (the other row half copies the
// rest), multiplies from registers while the next stage's fragments are read, and folds its MFMA
// chains into float32 sums every FOLD stages (f16 MFMA truncation: k1_pair.hip).  RING LDS slots,
// one s_barrier per stage, counted vmcnt.
//
// Frames: the planes hold whole frame groups of 16 (zero-padded); T bounds the stores.  q is
// written with row stride q_stride so that a launch may cover a frame sub-range of a longer slab.
#include "k1_f16.h"

// Schedule experiments (tools/k1_experiments.sh builds side libraries with -DPSA_K1P_X=bits):
//   1: the second row half (waves 4-7, the SIMD partners of 0-3) issues its DMA half a stage later
//   2: s_setprio 1 for waves 4-7      4: DMA pieces spread over the row-tile regions
//   8: B-fragment reads spread over the regions   16: no sched_barrier between the regions
//  32: DMA addressed as SGPR base + per-lane 32-bit offset, the pieces of a group told apart by the
//      instruction offset (it moves the global AND the LDS address: tools/probes/dma_offset.hip) --
//      one M0 write per group and no 64-bit VALU address arithmetic
//  64: main loop unrolled over the slot ring (LDS addresses become instruction offsets)
// Tried and removed (configuration 3, same box, 14.9 ms shipped): A fragments double-buffered with all
// 14 LDS reads of the next stage at the top of the stage 16.4 ms (an in-order wavefront cannot issue
// its MFMAs behind a read burst that fills the LDS queue); the same with the reads paced one per two
// MFMAs (sched_group_barrier) 14.9 ms -- no schedule moves the launch any more: it is power-limited.
// 256: no non-temporal policy on the planes of a one-M-block launch    512: running source pointers
// Timing-only experiments (WRONG results): 128 every V read an L2 hit; 1024 no MFMAs; 2048 the P' tile
// always stage 0 (L2-hot); 8192 no LDS fragment reads; 16384 no LDS-DMA in the main loop; 32768 no s_barrier
// Operand-order experiments (results stay right): 65536 / 131072 / 262144, see mfma_tile / mfma_stage_ordered --
// 1.3 % between the best and the worst order: not a lever (profiles/r3_k1_experiments.txt)
// 524288: a wavefront's DMA pieces issued one by one, spread over the stage (no gain either)
#ifndef PSA_K1P_X
#define PSA_K1P_X 563       // product build: 1 + 2 + 16 + 32 + 512 (64 measured 3 % slower)
#endif
#ifndef PSA_K1P_POS
#define PSA_K1P_POS -1      // row tile after which waves 4-7 issue their DMA (-1: the middle one)
#endif
#ifndef PSA_K1P_RING1
#define PSA_K1P_RING1 3     // ring slots of the 128-row variant when the launch has one M block (4 = all 160 KiB of LDS)
#endif
#ifndef PSA_K1P_PRIO
#define PSA_K1P_PRIO 1      // which row half runs at s_setprio 1 (bit 2 of PSA_K1P_X)
#endif

namespace psa {

template <int MT16_, int RING_>
struct K1pCfg {
    static constexpr int MT16 = MT16_;             // row tiles of 16 per wavefront
    static constexpr int M_BLK = 32 * MT16;        // 128 / 64 / 32 rows (64 / 32 / 16 k-vectors) per workgroup
    static constexpr int T_BLK = 64;
    static constexpr int FOLD = 8;
    static constexpr int RING = RING_;
    static constexpr int P_STAGE_BYTES = F16x2::NP * M_BLK * K1_BA * 2;      // 16 / 8 / 4 KiB
    static constexpr int P_PIECES = P_STAGE_BYTES / 1024;
    static constexpr int P_DMA = P_PIECES >= 8 ? P_PIECES / 8 : 1;           // per wavefront (4 pieces: waves 4-7 repeat 0-3)
    static constexpr int V_GROUP_BYTES = PL_STAGE_ELEMS * 2;                 // 6 KiB
    static constexpr int V_DMA = 3;
    static constexpr int STAGE_BYTES = P_STAGE_BYTES + 4 * V_GROUP_BYTES;    // 40 / 32 / 28 KiB
    static constexpr int LDS_BYTES = RING * STAGE_BYTES;
    static constexpr int BATCH = P_DMA + V_DMA;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static_assert(RING >= 3, "stage s+1 read while s+2 .. s+RING travel");
};

// NT_V: the launch has ONE M block, so every byte of the planes is read exactly once -> streamed with
// the non-temporal policy (PSA_K1P_X bit 256 switches it off for comparison)
template <int MT16_, int RING_, bool NT_V>
__global__ void __launch_bounds__(512, 1)
k1_planes_kernel(const _Float16* __restrict__ planes, const _Float16* __restrict__ Pb, float2* __restrict__ Q,
                 int64_t T, int64_t q_stride, int n_fg, int n_stage, int K, int n_mblk, int n_tblk, float qscale) {
    using C = K1pCfg<MT16_, RING_>;
    using PR = F16x2;
    using E8 = PR::v8;
    constexpr int NP = PR::NP, MT16 = C::MT16;
    __shared__ __attribute__((aligned(16))) unsigned char smem[C::LDS_BYTES];
    const unsigned lds0 = (unsigned)(size_t)(lds_u8*)smem;

    // XCD-aware block map (k1_pair.hip): blocks b and b+8 share an XCD and get the M blocks of one
    // frame tile, so the tile's planes come from HBM once.
    const int b  = blockIdx.x;
    const int r8 = b >> 3;
    const int mb = r8 % n_mblk;
    const int tb = (r8 / n_mblk) * 8 + (b & 7);
    if (tb >= n_tblk) return;

    const int     tid = threadIdx.x, lane = tid & 63;
    const int     w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int     wh = w >> 2, wf = w & 3;
    const int     r16 = lane & 15, q = lane >> 4;
    const int64_t t0 = (int64_t)tb * C::T_BLK + wf * 16;
    const int     last = n_stage - 1;
    int           fg = tb * 4 + wf;                                  // frame group (past the end: the last one, never stored)
    if (fg >= n_fg) fg = n_fg - 1;
    if constexpr ((PSA_K1P_X & 128) != 0) fg &= 63;   // experiment (WRONG results): every V read is an L2 / MALL hit

    // ---- DMA sources: this wavefront's blocks of the V planes and of the P' tile ----------------
    const unsigned char* vp = reinterpret_cast<const unsigned char*>(planes) +
                              (size_t)fg * n_stage * C::V_GROUP_BYTES + 1024 * (wh * C::V_DMA) + 16 * lane;
    const int            pw = C::P_PIECES >= 8 ? w * C::P_DMA : (w & 3);
    const unsigned char* pp = reinterpret_cast<const unsigned char*>(Pb) + (size_t)mb * n_stage * C::P_STAGE_BYTES +
                              1024 * pw + 16 * lane;
    // the same as (uniform base, per-lane offset) pairs for the SGPR-base form
    const unsigned char* vbase = reinterpret_cast<const unsigned char*>(planes) + (size_t)fg * n_stage * C::V_GROUP_BYTES;
    const unsigned char* pbase = reinterpret_cast<const unsigned char*>(Pb) + (size_t)mb * n_stage * C::P_STAGE_BYTES;
    const unsigned       v_voff = 1024 * (wh * C::V_DMA) + 16 * lane, p_voff = 1024 * pw + 16 * lane;
    // piece i of this wavefront's BATCH for stage st (clamped) -> slot: P' pieces first, then V
    auto dma_piece = [&](int i, int st, int slot) {
        const int      sc = st < last ? st : last;
        const unsigned dst = lds0 + slot * C::STAGE_BYTES;
        if (i < C::P_DMA) {
            lds_dma16(pp + (size_t)sc * C::P_STAGE_BYTES + 1024 * i, dst + 1024 * (pw + i));
        } else {
            const int      j = i - C::P_DMA;
            const unsigned vdst = dst + C::P_STAGE_BYTES + wf * C::V_GROUP_BYTES + 1024 * (wh * C::V_DMA);
            lds_dma16(vp + (size_t)sc * C::V_GROUP_BYTES + 1024 * j, vdst + 1024 * j);
        }
    };
    // bit 512: the sources of the next stage to fetch as running uniform pointers (one s_add_u32 /
    // s_addc_u32 pair each per stage instead of clamp + 64-bit multiply-add); stages past the end are
    // fetched from the bytes that follow (the next frame group / M block, or the RING stages of padding
    // behind the buffers: k1_planes_tail_pad) and never read
    const unsigned char* p_next = pbase;
    const unsigned char* v_next = vbase;
    auto dma_stage = [&](int st, int slot) {
        if constexpr ((PSA_K1P_X & 512) != 0) {
            const unsigned dst = lds0 + slot * C::STAGE_BYTES;
            if constexpr ((PSA_K1P_X & 4096) != 0)          // experiment: the HBM-served pieces first
                lds_dma16_group<C::V_DMA, NT_V && (PSA_K1P_X & 256) == 0>(
                    v_next, v_voff, dst + C::P_STAGE_BYTES + wf * C::V_GROUP_BYTES + 1024 * (wh * C::V_DMA));
            lds_dma16_group<C::P_DMA>(p_next, p_voff, dst + 1024 * pw);
            if constexpr ((PSA_K1P_X & 4096) == 0)
                lds_dma16_group<C::V_DMA, NT_V && (PSA_K1P_X & 256) == 0>(
                    v_next, v_voff, dst + C::P_STAGE_BYTES + wf * C::V_GROUP_BYTES + 1024 * (wh * C::V_DMA));
            if constexpr ((PSA_K1P_X & 2048) == 0) p_next += C::P_STAGE_BYTES;
            v_next += C::V_GROUP_BYTES;
        } else if constexpr ((PSA_K1P_X & 32) != 0) {
            const int      sc = st < last ? st : last;
            const unsigned dst = lds0 + slot * C::STAGE_BYTES;
            lds_dma16_group<C::P_DMA>(pbase + (size_t)sc * C::P_STAGE_BYTES, p_voff, dst + 1024 * pw);
            lds_dma16_group<C::V_DMA, NT_V && (PSA_K1P_X & 256) == 0>(
                vbase + (size_t)sc * C::V_GROUP_BYTES, v_voff, dst + C::P_STAGE_BYTES + wf * C::V_GROUP_BYTES + 1024 * (wh * C::V_DMA));
        } else {
#pragma unroll
            for (int i = 0; i < C::BATCH; ++i) dma_piece(i, st, slot);
        }
    };

    // bit 524288 (experiment): the BATCH pieces of a wavefront issued one by one, spread over the stage
    // (running pointers advance behind the last piece)
    auto dma_one = [&](auto i_c, int slot) {
        constexpr int  i = decltype(i_c)::value;
        const unsigned dst = lds0 + slot * C::STAGE_BYTES;
        if constexpr (i < C::P_DMA) {
            lds_dma16_at<1024 * i>(p_next, p_voff, dst + 1024 * pw);
        } else {
            lds_dma16_at<1024 * (i - C::P_DMA), NT_V && (PSA_K1P_X & 256) == 0>(
                v_next, v_voff, dst + C::P_STAGE_BYTES + wf * C::V_GROUP_BYTES + 1024 * (wh * C::V_DMA));
        }
        if constexpr (i == C::BATCH - 1) {
            p_next += C::P_STAGE_BYTES;
            v_next += C::V_GROUP_BYTES;
        }
    };

    // ---- LDS read addresses: both images use the 64-byte rows / swizzled 16-byte slots of k1_f16.h
    const int      gsw = pl_swizzle(r16);
    const unsigned p_lane = lds0 + (wh * (C::M_BLK / 2) + r16) * (K1_BA * 2) + ((q ^ gsw) << 4);
    const unsigned v_lane = lds0 + C::P_STAGE_BYTES + wf * C::V_GROUP_BYTES + r16 * (K1_BA * 2) + ((q ^ gsw) << 4);
    E8    a[NP][MT16];
    E8    bs[2][3][NP];                            // B fragments of stage k: bs[k & 1][component][piece]
    f32x4 hi[MT16][3], lo[MT16][3];                // the running MFMA chains / the float32 sums
    if constexpr ((PSA_K1P_X & 8192) != 0) {       // (timing experiment without LDS reads: operands that are
        // defined, different from register to register and random-looking, so that the matrix pipe sees the
        // switching activity of real data -- constant operands would let the chip clock up)
        auto junk = [&](unsigned salt) {
            E8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                unsigned h = (unsigned)(lane * 8 + e) * 2654435761u ^ (salt * 40503u + 0x9E3779B9u);
                h ^= h >> 15;
                h *= 2246822519u;
                h ^= h >> 13;
                v[e] = (_Float16)((float)(int)(h & 0xFFFF) * (1.f / 8.f) - 4096.f);
            }
            return v;
        };
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#pragma unroll
            for (int mt = 0; mt < MT16; ++mt) a[p][mt] = junk(p * 8 + mt);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                bs[0][c][p] = junk(100 + p * 8 + c);
                bs[1][c][p] = junk(200 + p * 8 + c);
            }
        }
    }
    auto  read_a_tile = [&](int mt, int slot) {
        if constexpr ((PSA_K1P_X & 8192) != 0) return;
        const unsigned base = p_lane + slot * C::STAGE_BYTES;
#pragma unroll
        for (int p = 0; p < NP; ++p)
            a[p][mt] = *reinterpret_cast<lds_cv8*>((const lds_u8*)(size_t)(base + (p * C::M_BLK + mt * 16) * 64));
    };
    auto read_b1 = [&](int par, int slot, int i) {
        if constexpr ((PSA_K1P_X & 8192) != 0) return;
        bs[par][i >> 1][i & 1] = *reinterpret_cast<lds_cv8*>((const lds_u8*)(size_t)(v_lane + slot * C::STAGE_BYTES + i * 1024));
    };
    auto read_b = [&](int par, int slot) {
#pragma unroll
        for (int i = 0; i < 3 * NP; ++i) read_b1(par, slot, i);
    };
#pragma unroll
    for (int mt = 0; mt < MT16; ++mt)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            hi[mt][c] = f32x4{0.f, 0.f, 0.f, 0.f};
            lo[mt][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

    if constexpr ((PSA_K1P_X & 2) != 0) {
        if (wh == PSA_K1P_PRIO) __builtin_amdgcn_s_setprio(1);
    }
    // ---- prologue: stages 0 .. RING-1 in flight; stage 0 into registers --------------------------
#pragma unroll
    for (int k = 0; k < C::RING; ++k) dma_stage(k, k);
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((C::RING - 1) * C::BATCH) : "memory");      // stage 0 landed
    read_b(0, 0);
#pragma unroll
    for (int mt = 0; mt < MT16; ++mt) read_a_tile(mt, 0);
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((C::RING - 2) * C::BATCH) : "memory");   // stage 1 landed, slot 0 read

    auto mfma_tile = [&](int mt, int par, bool restart) {
        if constexpr ((PSA_K1P_X & 1024) != 0) {        // timing experiment: operands consumed, no matrix work
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                asm volatile("" ::"v"(a[0][mt]), "v"(a[1][mt]), "v"(bs[par][c][0]), "v"(bs[par][c][1]));
                if (restart) hi[mt][c] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            return;
        }
        f32x4 ch[3];
        if constexpr ((PSA_K1P_X & 65536) != 0) {
            // experiment: every MFMA differs from its predecessor in ONE operand register (the component
            // order snakes: up, down, up within a row tile, mirrored on odd row tiles); the order of the
            // three terms of a chain is kept
            const bool up = (mt & 1) == 0;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int c = up ? i : 2 - i;
                ch[c] = PR::mma(a[1][mt], bs[par][c][0], restart ? f32x4{0.f, 0.f, 0.f, 0.f} : hi[mt][c]);
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int c = up ? 2 - i : i;
                ch[c] = PR::mma(a[0][mt], bs[par][c][1], ch[c]);
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int c = up ? i : 2 - i;
                hi[mt][c] = PR::mma(a[0][mt], bs[par][c][0], ch[c]);
            }
            return;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c)
            ch[c] = PR::mma(a[1][mt], bs[par][c][0], restart ? f32x4{0.f, 0.f, 0.f, 0.f} : hi[mt][c]);
#pragma unroll
        for (int c = 0; c < 3; ++c) ch[c] = PR::mma(a[0][mt], bs[par][c][1], ch[c]);
#pragma unroll
        for (int c = 0; c < 3; ++c) hi[mt][c] = PR::mma(a[0][mt], bs[par][c][0], ch[c]);
    };
    // experiments 131072 / 262144: all 36 MFMAs of a stage in one block, term by term over the twelve
    // (row tile, component) pairs -- 131072: snake order, one operand register changes per MFMA;
    // 262144: diagonal order, both change every time.  Same instructions, same reads (all behind the block).
    auto mfma_stage_ordered = [&](int par, bool restart, bool snake) {
#pragma unroll
        for (int term = 0; term < 3; ++term) {
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                int mt, c;
                if (snake) {
                    mt = j / 3;
                    c = (mt & 1) ? 2 - j % 3 : j % 3;
                    if (term == 1) {             // walk back so that phase B ends where phase C starts
                        mt = 3 - mt;
                        c = 2 - c;
                    }
                } else {
                    mt = j % 4;
                    c = (j + j / 4) % 3;
                }
                if (term == 0)
                    hi[mt][c] = PR::mma(a[1][mt], bs[par][c][0], restart ? f32x4{0.f, 0.f, 0.f, 0.f} : hi[mt][c]);
                else if (term == 1)
                    hi[mt][c] = PR::mma(a[0][mt], bs[par][c][1], hi[mt][c]);
                else
                    hi[mt][c] = PR::mma(a[0][mt], bs[par][c][0], hi[mt][c]);
            }
        }
    };
    // One stage: slot holds stage s (in registers already), slot1 stage s+1 (landed).  The DMA of
    // stage s+RING goes into slot; the B fragments of stage s+1 are read at the top, each row tile's
    // A fragments right behind the MFMAs that consumed the old ones.
    auto stage = [&](auto par_c, auto restart_c, int s, auto slot_c) {
        constexpr int  par = decltype(par_c)::value;
        constexpr bool restart = decltype(restart_c)::value;
        const int      slot = slot_c;                  // an int, or an integral_constant (unrolled loop)
        const int      slot1 = slot == C::RING - 1 ? 0 : slot + 1;
        constexpr bool STAGGER = (PSA_K1P_X & 1) != 0, SPREAD_DMA = (PSA_K1P_X & 4) != 0,
                       SPREAD_B = (PSA_K1P_X & 8) != 0, FREE = (PSA_K1P_X & 16) != 0;
        constexpr bool NO_DMA = (PSA_K1P_X & 16384) != 0;
        constexpr bool SPREAD_RP = (PSA_K1P_X & 524288) != 0 && MT16 == 4 && C::BATCH == 5;
        if constexpr (SPREAD_RP) {
            dma_one(std::integral_constant<int, 0>{}, slot);
        } else if constexpr (!SPREAD_DMA && !NO_DMA) {
            if (!STAGGER || wh == 0) dma_stage(s + C::RING, slot);
        }
        if constexpr (!SPREAD_B) read_b(par ^ 1, slot1);
        if constexpr (!FREE) __builtin_amdgcn_sched_barrier(0);
        if constexpr ((PSA_K1P_X & (131072 | 262144)) != 0 && MT16 == 4) {
            mfma_stage_ordered(par, restart, (PSA_K1P_X & 131072) != 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < MT16; ++mt) {
                read_a_tile(mt, slot1);
                if constexpr (STAGGER && !NO_DMA) {
                    if (mt == 1 && wh == 1) dma_stage(s + C::RING, slot);
                }
            }
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((C::RING - 2) * C::BATCH) : "memory");
            return;
        }
#pragma unroll
        for (int mt = 0; mt < MT16; ++mt) {
            if constexpr (SPREAD_B) {
#pragma unroll
                for (int i = mt * 6 / MT16; i < (mt + 1) * 6 / MT16; ++i) read_b1(par ^ 1, slot1, i);
            }
            if constexpr (SPREAD_DMA) {
#pragma unroll
                for (int i = mt * C::BATCH / MT16; i < (mt + 1) * C::BATCH / MT16; ++i) dma_piece(i, s + C::RING, slot);
            }
            mfma_tile(mt, par, restart);
            if constexpr (!FREE) __builtin_amdgcn_sched_barrier(0);
            read_a_tile(mt, slot1);
            if constexpr (SPREAD_RP) {
                if (mt == 0) dma_one(std::integral_constant<int, 1>{}, slot);
                if (mt == 1) dma_one(std::integral_constant<int, 2>{}, slot);
                if (mt == 2) dma_one(std::integral_constant<int, 3>{}, slot);
                if (mt == 3) dma_one(std::integral_constant<int, 4>{}, slot);
            }
            if constexpr (STAGGER && !SPREAD_DMA && !NO_DMA && !SPREAD_RP) {
                if (mt == (PSA_K1P_POS < 0 ? (MT16 - 1) / 2 : (PSA_K1P_POS < MT16 ? PSA_K1P_POS : MT16 - 1)) && wh == 1)
                    dma_stage(s + C::RING, slot);
            }
            if constexpr (!FREE) __builtin_amdgcn_sched_barrier(0);
        }
        // own blocks of stage s+2 landed (younger batches stay in flight), own LDS reads returned
        if constexpr ((PSA_K1P_X & 32768) != 0)
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((C::RING - 2) * C::BATCH) : "memory");
        else if constexpr (NO_DMA)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((C::RING - 2) * C::BATCH) : "memory");
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    int s0 = 0;
    if constexpr ((PSA_K1P_X & 64) != 0) {
        // main loop: one fold period per iteration, slots and parities compile-time
        // (RING 3: 6 stages, RING 4: 8 -- both within the <= FOLD stages a chain may run)
        constexpr int PERIOD = C::RING == 3 ? 6 : 8;
        static_assert(PERIOD <= C::FOLD && PERIOD % C::RING == 0 && PERIOD % 2 == 0, "period");
        for (; s0 + PERIOD <= n_stage; s0 += PERIOD) {
            [&]<int... I>(std::integer_sequence<int, I...>) {
                (stage(std::integral_constant<int, (I & 1)>{}, std::bool_constant<I == 0>{}, s0 + I,
                       std::integral_constant<int, I % C::RING>{}), ...);
            }(std::make_integer_sequence<int, PERIOD>{});
#pragma unroll
            for (int mt = 0; mt < MT16; ++mt)
#pragma unroll
                for (int c = 0; c < 3; ++c) lo[mt][c] += hi[mt][c];
        }
    }
    int  slot = 0;                                     // s % RING  (s0 is a multiple of RING)
    auto next_slot = [&]() { slot = slot == C::RING - 1 ? 0 : slot + 1; };
    for (int s = s0; s < n_stage;) {                   // n_stage is even; a chain is an even number of stages
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
#pragma unroll
        for (int mt = 0; mt < MT16; ++mt)
#pragma unroll
            for (int c = 0; c < 3; ++c) lo[mt][c] += hi[mt][c];
        s += len;
    }
    // the clamped prefetches of stages >= n_stage are still in flight: let them land before the
    // workgroup's LDS is handed to the next one
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

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
                    for (int c = 0; c < 3; ++c)
                        Q[((int64_t)k * 3 + c) * q_stride + t] =
                            make_float2(lo[mt][c][2 * pr] * qscale, lo[mt][c][2 * pr + 1] * qscale);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The split itself: d (float32, native (T, N, 3) rows; atoms through an index list if there is
// one) -> the two scaled float16 planes in fragment order.  One pass: 12 bytes read and 12 written
// per (frame, atom).  A block of three wavefronts handles one (frame group, stage) tile: the 16 x 96
// floats go through LDS (coalesced reads along the row), then wavefront c converts component c,
// each lane one 8-atom fragment, written as two 16-byte stores.
// ---------------------------------------------------------------------------------------------
// VEC: whole trajectory in its own order with N % 4 == 0 -- every 96-float row segment is 16-byte
// aligned and is read as 24 float4 (two loads per thread instead of eight)
template <bool VEC>
__global__ void __launch_bounds__(192)
split_planes_kernel(const float* __restrict__ x, const float* __restrict__ mean, const int* __restrict__ idx,
                    _Float16* __restrict__ planes, int64_t T, int64_t N_tot, int n_g, int n_stage, int64_t n_fg, float vscale) {
    // mean (N_tot,3), may be null: the planes hold x - mean, one float32 subtraction per element like
    // the reference's temporary (sed_calculator.py:70-72)
    __shared__ __attribute__((aligned(16))) float raw[16][100];
    const int s = blockIdx.x;
    const int tid = threadIdx.x;
    for (int64_t fg = blockIdx.y; fg < n_fg; fg += gridDim.y) {
        const int64_t t0 = fg * 16;
        if constexpr (VEC) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int     e = tid + 192 * j;                 // float4 index in the 16 x 24 tile
                const int     row = e / 24, c4 = e - row * 24;
                const int64_t t = t0 + row;
                f32x4         v = {0.f, 0.f, 0.f, 0.f};
                // (atoms past n_g = N_tot: only the last stage can be short, and then by whole float4s
                //  only if N % 32 is a multiple of 4/3 atoms -- handled per element below)
                if (t < T) {
                    const int64_t col = (int64_t)s * 96 + 4 * c4;
                    if (col + 3 < 3 * N_tot) {
                        v = *reinterpret_cast<const f32x4*>(x + t * 3 * N_tot + col);
                        if (mean) {
                            const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + col);
#pragma unroll
                            for (int u = 0; u < 4; ++u) v[u] = __fsub_rn(v[u], mu[u]);
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (col + u < 3 * N_tot) v[u] = __fsub_rn(x[t * 3 * N_tot + col + u], mean ? mean[col + u] : 0.f);
                    }
                }
                *reinterpret_cast<f32x4*>(&raw[row][4 * c4]) = v;
            }
        } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int     e = tid + 192 * j;
            const int     row = e / 96, col = e - row * 96;
            const int     al = col / 3, comp = col - 3 * al;
            const int     pos = s * K1_BA + al;
            const int64_t t = t0 + row;
            float         v = 0.f;
            if (t < T && pos < n_g) {
                const int64_t atom = idx ? idx[pos] : pos;
                v = x[(t * N_tot + atom) * 3 + comp];
                if (mean) v = __fsub_rn(v, mean[atom * 3 + comp]);
            }
            raw[row][col] = v;
        }
        }
        __syncthreads();
        const int comp = tid >> 6, r = (tid >> 2) & 15, oct = tid & 3;
        F16x2::v2 lead[4], rest[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            split_pair(raw[r][(8 * oct + 2 * i) * 3 + comp], raw[r][(8 * oct + 2 * i + 1) * 3 + comp], vscale, lead[i],
                       rest[i]);
        const size_t o = plane_index(fg, s, comp, 0, r, 8 * oct, n_stage);
        *reinterpret_cast<F16x2::v8*>(planes + o) = cat4(lead[0], lead[1], lead[2], lead[3]);
        *reinterpret_cast<F16x2::v8*>(planes + o + PL_BLOCK_ELEMS) = cat4(rest[0], rest[1], rest[2], rest[3]);
        __syncthreads();
    }
}

int launch_split_planes(psa_ctx* c, const float* d_x, const float* d_mean, const int* d_idx, void* d_planes, int64_t T,
                        int64_t N_tot, int n_g, int A_pad, float vscale) {
    const int     n_stage = A_pad / K1_BA;
    const int64_t n_fg = (T + 15) / 16;
    PSA_REQUIRE(n_stage > 0 && n_fg > 0 && vscale > 0.f, "bad split geometry");
    dim3 grid((unsigned)n_stage, (unsigned)(n_fg < 4096 ? n_fg : 4096));
    if (d_idx == nullptr && n_g == N_tot && N_tot % 4 == 0)
        hipLaunchKernelGGL(split_planes_kernel<true>, grid, dim3(192), 0, c->stream, d_x, d_mean, d_idx, (_Float16*)d_planes, T,
                           N_tot, n_g, n_stage, n_fg, vscale);
    else
        hipLaunchKernelGGL(split_planes_kernel<false>, grid, dim3(192), 0, c->stream, d_x, d_mean, d_idx, (_Float16*)d_planes, T,
                           N_tot, n_g, n_stage, n_fg, vscale);
    PSA_HIP_CHECK(hipGetLastError());
    return PSA_OK;
}

// rows per workgroup for a k-list: 128 beyond 32 k-vectors, 64 for 17..32, 32 up to 16
// 256-row blocks (k1_planes_wide.hip, 3-6 % faster per row) where they cost no extra rows: an even number of
// 128-row blocks.  (192 k-vectors = three 128-row blocks would become two 256-row blocks: a third more MFMAs.)
int k1_planes_block_rows(int K, bool wide) {
    if (2 * K <= 32) return 32;
    if (2 * K <= 64) return 64;
    const int n128 = (2 * K + 127) / 128;
    return wide && n128 % 2 == 0 ? 256 : 128;
}

template <int MT16, int RING>
static int launch_planes_variant(psa_ctx* c, const void* d_planes, const void* d_phase, float2* d_q, const ProjGeom& g,
                                 int64_t n_fg) {
    using C = K1pCfg<MT16, RING>;
    const int     n_mblk = g.M_pad / C::M_BLK;
    const int64_t n_tblk = (g.T + C::T_BLK - 1) / C::T_BLK;
    const int64_t grid = ((n_tblk + 7) / 8) * 8 * n_mblk;
    PSA_REQUIRE(grid < (1ll << 31) && n_tblk < (1ll << 29) && n_fg < (1ll << 31), "projection grid too large");
    const float qscale = 1.f / (g.vscale * F16x2::P_SCALE);           // powers of two: exact
    if (n_mblk == 1)
        hipLaunchKernelGGL((k1_planes_kernel<MT16, RING, true>), dim3((unsigned)grid), dim3(512), 0, c->stream,
                           (const _Float16*)d_planes, (const _Float16*)d_phase, d_q, g.T, g.q_stride, (int)n_fg,
                           g.A_pad / K1_BA, g.K, n_mblk, (int)n_tblk, qscale);
    else
        hipLaunchKernelGGL((k1_planes_kernel<MT16, RING, false>), dim3((unsigned)grid), dim3(512), 0, c->stream,
                           (const _Float16*)d_planes, (const _Float16*)d_phase, d_q, g.T, g.q_stride, (int)n_fg,
                           g.A_pad / K1_BA, g.K, n_mblk, (int)n_tblk, qscale);
    PSA_HIP_CHECK(hipGetLastError());
    return PSA_OK;
}

// d_planes: the group's planes from the frame group that holds the launch's first frame on
// (g.T frames from there, n_fg frame groups available)
int launch_k1_planes(psa_ctx* c, const void* d_planes, const void* d_phase, float2* d_q, const ProjGeom& g, int64_t n_fg) {
    PSA_REQUIRE((g.m_blk == 128 || g.m_blk == 64 || g.m_blk == 32) && g.M_pad % g.m_blk == 0,
                "planes kernel needs 32-, 64- or 128-row M blocks");
    PSA_REQUIRE(g.A_pad % (2 * K1_BA) == 0 && g.A_pad > 0, "planes kernel needs the atom axis padded to %d", 2 * K1_BA);
    PSA_REQUIRE(g.vscale > 0.f && n_fg * 16 >= g.T, "planes do not cover the launch");
    if (g.m_blk == 128 && g.M_pad == 128) return launch_planes_variant<4, PSA_K1P_RING1>(c, d_planes, d_phase, d_q, g, n_fg);
    if (g.m_blk == 128) return launch_planes_variant<4, 3>(c, d_planes, d_phase, d_q, g, n_fg);
    if (g.m_blk == 64) return launch_planes_variant<2, 4>(c, d_planes, d_phase, d_q, g, n_fg);
    return launch_planes_variant<1, 4>(c, d_planes, d_phase, d_q, g, n_fg);
}

}  // namespace psa
