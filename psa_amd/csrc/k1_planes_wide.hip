// K1 on pre-split planes with a 256-ROW WORKGROUP TILE: the projection GEMM of k1_planes.hip (same
// "2 x f16" arithmetic, same plane image, same float32-fold rule), for k-lists that fill an even number of
// 128-row blocks (k1_planes_block_rows).
//
// Why: round 3 measured (profiles/r3_k1_experiments.txt, "less LDS-DMA") that what the 128-row kernels
// wait for is the LDS-DMA pipeline as much as the matrix cores -- a CU takes in a 1-KiB LDS-DMA
// instruction per 20-40 ns and a 128-row x 64-frame tile needs 40 of them per 32-atom stage; with 28 per
// stage (timing-only build) the same launch took 14.5 % less.  28 KiB per 128 rows is what a 256-row tile
// needs: 32 KiB of phase tile + 24 KiB of planes for twice the MFMAs.
//
// Registers decide the shape.  A 256-row x 64-frame x 3-component tile is 192 KiB of MFMA chains plus
// 192 KiB of float32 sums -- three quarters of the CU's register file -- and every wavefront of a kernel
// gets the same allocation, so there is no room for wavefronts that only load (k1_planes_lw.hip): EIGHT
// wavefronts, two per SIMD, <= 256 VGPRs each; wavefront w = 4 h + f owns rows [128 h, +128) x frames
// [16 f, +16) x 3 components = 24 accumulator tiles (96 + 96 registers), keeps the stage's B fragments
// (24 registers) and streams the A fragments one row tile ahead (2 x 8 registers).
//
// LDS decides the ring.  A stage is 56 KiB; 160 KiB hold 2.86 of them, so the ring is not made of stage
// slots but of 1-KiB UNITS (one LDS-DMA instruction = one MFMA fragment block): stage s, unit u lives at
// ring position (56 s + u) mod 160.  The wavefronts issue the stream in order, wavefront w the units
// u = 8 k + w (k = 0..6) of every stage: k = 0..2 are the planes (unit 8 c + 4 p + j = component c, piece p,
// frame group j: a wavefront copies its own frame group's piece h), k = 3..6 the phase tile in the order
// the row tiles are consumed (unit 24 + 4 mt + 2 h + p).  Entering stage s (barrier s-1 passed; stage s-1's 56
// units are free) a wavefront issues (s+1, k=6) and (s+2, k=0..5) -- in the shipped form waves 0-3 issue their
// SIMD partners' units too (PSA_K1W_SOLO); the barrier that ends a stage is preceded by vmcnt(2) (SOLO: 4):
// everything but (s+2, k=4..5) has landed -- all of stage s+1 and the first 32 units of stage s+2, i.e. its planes
// and row tiles 0-1, which is what the last row tile of a stage prefetches.  Every unit has at least one
// stage time (~1.7 us) between issue and first use.  Ring positions repeat every 20 stages (7 blocks of
// 8 KiB per stage, 20 blocks): the main loop is unrolled 20 times and every LDS address is a constant.
// Chains are folded into the float32 sums every 10 stages (a divisor of 20; bound 10 x 2^-24).
#include <utility>

#include "k1_f16.h"

// Schedule switches (tools/k1_experiments.sh with KERNEL=k1_planes_wide MACRO=PSA_K1W_X, EXTRA_DEFS for the others; results
// stay right; measurements in profiles/r3_k1_experiments.txt):
//   PSA_K1W_X bits   1: chains folded every 20 stages instead of every 10 (no gain)   2: s_setprio 1 for waves 4-7 (no gain)
//                    4: the stages past the group's last one are skipped instead of multiplying zeros (-1 %: configuration 3
//                       has 1024 stages = 51 periods + 4)
//                    8: the next row tile's A fragments read before the tile's first MFMA (left alone hipcc reads them behind
//                       the third and reuses the dead low piece's registers; no gain)
//   PSA_K1W_SOLO 1   waves 0-3 issue ALL LDS-DMA -- their own units and their SIMD partners' --, waves 4-7 none: a wavefront is
//                    held ~70 cycles per LDS-DMA instruction, its partner multiplies meanwhile (-1.8 % against the next line)
//   PSA_K1W_POS p    (SOLO 0) waves 4-7 issue their LDS-DMA behind row tile p instead of at the top of the stage, where their
//                    partners are issuing too (p = 1: -2 %; p = 3: +5 %, the units arrive late)
//   PSA_K1W_STAMP 1  DIAGNOSTIC build: every wavefront of workgroups 0-7 sums s_memtime differences over its stages (top of the
//                    stage -> LDS-DMA issued -> row tile 0 -> tiles 1-3 -> tiles 4-7 -> fold + vmcnt -> barrier) and prints them
// Issuing part of the LDS-DMA from inside the row-tile sequence (the units that have two stages to land, or the loads spread
// over the stage) could not be measured: every such build spills 6-350 VGPRs -- any asm statement between the row tiles does
// it, a C++ branch around a role's loads too -- and scratch traffic both costs time and breaks the counted vmcnt
// (tests/test_kernel_resources.py guards the product build).  The kernel sits at 248 of 256 registers.
// Tried with the loads of a role as ONE asm statement that the other role jumps over (no spills): waves 0-3 the ten
// instructions that have one stage to land at the top of the stage, waves 4-7 the four that have two at its end: +4 %
// (waves 4-7 are not early at the barrier: from the moment waves 0-3 compute too, the two halves share the matrix pipe);
// the same split with the loads under an EXEC mask instead of a jump: +13 % (a load under EXEC = 0 still costs its issue).
#ifndef PSA_K1W_X
#define PSA_K1W_X 4
#endif
#ifndef PSA_K1W_SOLO
#define PSA_K1W_SOLO 1
#endif
#ifndef PSA_K1W_POS
#define PSA_K1W_POS 1
#endif
#ifndef PSA_K1W_STAMP
#define PSA_K1W_STAMP 0
#endif

namespace psa {

namespace {
constexpr int W_M_BLK = 256, W_T_BLK = 64, W_FOLD = (PSA_K1W_X & 1) ? 20 : 10, W_PERIOD = 20;
constexpr int W_STAGE_UNITS = 56, W_RING_UNITS = 160;
constexpr int W_P_STAGE_BYTES = F16x2::NP * W_M_BLK * K1_BA * 2;        // 32 KiB
constexpr int W_V_GROUP_BYTES = PL_STAGE_ELEMS * 2;                     // 6 KiB
// byte offset in LDS of unit `unit` of a stage whose number is s20 (mod 20)
constexpr unsigned w_unit_off(int s20, int unit) { return (unsigned)((W_STAGE_UNITS * (s20 % W_PERIOD) + unit) % W_RING_UNITS) * 1024u; }

// one LDS-DMA instruction: uniform 64-bit base + per-lane offset -> LDS at wbase + OFF (+ lane * 16)
template <unsigned OFF, bool NT>
__device__ __forceinline__ void w_dma(const void* sbase, unsigned voff, unsigned wbase) {
    if constexpr (NT)
        asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(voff), "s"(sbase), "s"(wbase), "n"(OFF) : "memory", "scc");
    else
        asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(wbase), "n"(OFF) : "memory", "scc");
}
}  // namespace

template <bool NT_V>
__global__ void __launch_bounds__(512, 2)
k1_planes_wide_kernel(const _Float16* __restrict__ planes, const _Float16* __restrict__ Pb, const void* __restrict__ zeros,
                      float2* __restrict__ Q, int64_t T, int64_t q_stride, int n_fg, int n_stage, int K, int n_mblk, int n_tblk, float qscale) {
    using PR = F16x2;
    using E8 = PR::v8;
    constexpr int NP = PR::NP, MT = 8;
    __shared__ __attribute__((aligned(16))) unsigned char smem[W_RING_UNITS * 1024];
    const unsigned lds0 = (unsigned)(size_t)(lds_u8*)smem;

    // XCD-aware block map (k1_planes.hip): blocks b and b+8 share an XCD and get the M blocks of one frame tile
    const int b = blockIdx.x;
    const int r8 = b >> 3;
    const int mb = r8 % n_mblk;
    const int tb = (r8 / n_mblk) * 8 + (b & 7);
    if (tb >= n_tblk) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wh = w >> 2, wf = w & 3;
    const int r16 = lane & 15, q = lane >> 4;

    // ---- this wavefront's seven source streams (one per k), advanced by a stage after each use ---------
    const unsigned char* src[7];
    const unsigned char* pl0 = reinterpret_cast<const unsigned char*>(planes);
    int                  fg = tb * 4 + wf;                              // frame group (past the end: the last one, never stored)
    if (fg >= n_fg) fg = n_fg - 1;
#pragma unroll
    for (int k = 0; k < 3; ++k)                                         // unit 8 k + w = component k, piece wh, frame group wf
        src[k] = pl0 + (size_t)fg * n_stage * W_V_GROUP_BYTES + 1024 * (2 * k + wh);
    const unsigned char* ph0 = reinterpret_cast<const unsigned char*>(Pb) + (size_t)mb * n_stage * W_P_STAGE_BYTES;
#pragma unroll
    for (int k = 3; k < 7; ++k) {
        const int y = 8 * (k - 3) + w, mt = y >> 2, h = (y >> 1) & 1, p = y & 1;
        src[k] = ph0 + 1024 * (p * 16 + h * 8 + mt);                    // image [piece][256 rows][32 atoms]: 1 KiB per row tile
    }
    // (PSA_K1W_SOLO: the partner's streams, units 8 k + w + 4 -- piece 1 of the same planes, the other half's row tiles)
    const unsigned char* src2[7];
#pragma unroll
    for (int k = 0; k < 3; ++k) src2[k] = pl0 + (size_t)fg * n_stage * W_V_GROUP_BYTES + 1024 * (2 * k + 1);
#pragma unroll
    for (int k = 3; k < 7; ++k) {
        const int y = 8 * (k - 3) + w + 4, mt = y >> 2, h = (y >> 1) & 1, p = y & 1;
        src2[k] = ph0 + 1024 * (p * 16 + h * 8 + mt);
    }
    const unsigned dma_voff = 16 * lane;
    int            to_issue = n_stage;                                  // stages whose planes have not been issued yet
    const unsigned wbase = lds0 + 1024 * w;                             // unit 8 k + w of a stage: 1024 w past the block of k
    // instruction k of the stage whose number is S20 (mod 20)
    auto dma = [&](auto s20_c, auto k_c) __attribute__((always_inline)) {
        constexpr int      S20 = decltype(s20_c)::value, KK = decltype(k_c)::value;
        constexpr unsigned OFF = w_unit_off(S20, 8 * KK);
        if constexpr (KK < 3) {
            // stages past the group's last one (the loop runs whole periods of 20) multiply zeros: their
            // planes come from a block of zeros, their phase tile is whatever follows (finite float16)
            w_dma<OFF, NT_V>(to_issue > 0 ? src[KK] : reinterpret_cast<const unsigned char*>(zeros), dma_voff, wbase);
            src[KK] += W_V_GROUP_BYTES;
            if constexpr (PSA_K1W_SOLO != 0) {
                w_dma<OFF + 4096, NT_V>(to_issue > 0 ? src2[KK] : reinterpret_cast<const unsigned char*>(zeros), dma_voff, wbase);
                src2[KK] += W_V_GROUP_BYTES;
            }
            if constexpr (KK == 2) --to_issue;
        } else {
            w_dma<OFF, false>(src[KK], dma_voff, wbase);
            src[KK] += W_P_STAGE_BYTES;
            if constexpr (PSA_K1W_SOLO != 0) {
                w_dma<OFF + 4096, false>(src2[KK], dma_voff, wbase);
                src2[KK] += W_P_STAGE_BYTES;
            }
        }
    };
    using std::integral_constant;
    auto dma_range = [&]<int S20, int... Ks>(integral_constant<int, S20>, std::integer_sequence<int, Ks...>) __attribute__((always_inline)) {
        (dma(integral_constant<int, S20>{}, integral_constant<int, Ks>{}), ...);
    };

    // ---- fragment reads: every unit is 16 rows (frames) x 64 bytes, lane (r16, q) takes 16 bytes ----------
    const unsigned lane_off = lds0 + r16 * (K1_BA * 2) + ((q ^ pl_swizzle(r16)) << 4);
    // ds_read offsets reach 64 KiB: one base register per 64-KiB window of the ring and operand, opaque to the
    // optimizer (left to itself it forms a new base for almost every constant and spills them)
    unsigned lane_a[3], lane_b[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        lane_a[i] = lane_off + 2048 * wh + 65536 * i;
        lane_b[i] = lane_off + 1024 * wf + 65536 * i;
        asm volatile("" : "+v"(lane_a[i]), "+v"(lane_b[i]));
    }
    auto lds_frag = [&](const unsigned (&base)[3], unsigned off) __attribute__((always_inline)) {
        return *reinterpret_cast<lds_cv8*>((const lds_u8*)(size_t)(base[off >> 16] + (off & 0xFFFFu)));
    };
    E8    a[2][NP];                                // A fragments: row tile in work / the next one
    E8    bf[3][NP];                               // B fragments of the stage in work
    f32x4 hi[MT][3], lo[MT][3];
    auto  read_a = [&](auto s20_c, auto buf_c, auto mt_c) __attribute__((always_inline)) {
        constexpr int S20 = decltype(s20_c)::value, MTI = decltype(mt_c)::value, buf = decltype(buf_c)::value;
#pragma unroll
        for (int p = 0; p < NP; ++p)
            a[buf][p] = lds_frag(lane_a, w_unit_off(S20, 24 + 4 * MTI) + 1024 * p);
    };
    auto read_b = [&](auto s20_c, auto c_c) __attribute__((always_inline)) {
        constexpr int S20 = decltype(s20_c)::value, CC = decltype(c_c)::value;
#pragma unroll
        for (int p = 0; p < NP; ++p)
            bf[CC][p] = lds_frag(lane_b, w_unit_off(S20, 8 * CC) + 4096 * p);
    };
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            hi[mt][c] = f32x4{0.f, 0.f, 0.f, 0.f};
            lo[mt][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    auto fold = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int c = 0; c < 3; ++c) lo[mt][c] += hi[mt][c];
    };

    // ---- prologue: stage 0 (k = 0..6) and stage 1 (k = 0..5) ----------------------------------------------
    using I0 = integral_constant<int, 0>;
    using I1 = integral_constant<int, 1>;
    constexpr int PENDING = PSA_K1W_SOLO ? 4 : 2;                          // DMA instructions a barrier leaves in flight per issuing wavefront
    if (PSA_K1W_SOLO == 0 || wh == 0) {
        dma_range(I0{}, std::make_integer_sequence<int, 7>{});
        dma_range(I1{}, std::make_integer_sequence<int, 6>{});
    }
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PENDING) : "memory");   // stage 0 and the first 32 units of stage 1 landed
    read_b(I0{}, I0{});
    read_b(I0{}, I1{});
    read_b(I0{}, integral_constant<int, 2>{});
    read_a(I0{}, I0{}, I0{});

    // ---- one stage; returns false behind the last one -------------------------------------------------------
    int  left = n_stage;                                                   // stages to go when the period began
#if PSA_K1W_STAMP
    unsigned st_acc[6] = {0, 0, 0, 0, 0, 0}, st_prev = 0, st_n = 0;
    auto     stamp_now = [&]() __attribute__((always_inline)) {
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        return (unsigned)t;
    };
    auto stamp = [&](int i) __attribute__((always_inline)) {
        const unsigned t = stamp_now();
        st_acc[i] += t - st_prev;
        st_prev = t;
    };
    st_prev = stamp_now();
#define PSA_STAMP(i) stamp(i)
#else
#define PSA_STAMP(i)
#endif
    auto stage = [&](auto s20_c) __attribute__((always_inline)) {
        constexpr int  S20 = decltype(s20_c)::value;
        if constexpr ((PSA_K1W_X & 4) != 0) {
            if (left <= S20) return;
        }
        constexpr bool restart = S20 % W_FOLD == 0, folds = S20 % W_FOLD == W_FOLD - 1;
        using SN = integral_constant<int, (S20 + 1) % W_PERIOD>;
        using SNN = integral_constant<int, (S20 + 2) % W_PERIOD>;
        // stage s-1's units are free: (s+1, k=6), (s+2, k=0..5)
        auto issue = [&]() __attribute__((always_inline)) {
            dma(SN{}, integral_constant<int, 6>{});
            dma_range(SNN{}, std::make_integer_sequence<int, 6>{});
        };
        if constexpr (PSA_K1W_SOLO != 0) {
            if (wh == 0) issue();
        } else if constexpr (PSA_K1W_POS < 0) {
            issue();
        } else {
            if (wh == 0) issue();
        }
        PSA_STAMP(0);
        auto tile = [&](auto mt_c) __attribute__((always_inline)) {
            constexpr int MTI = decltype(mt_c)::value, cur = MTI & 1;
            if constexpr (MTI < MT - 1)
                read_a(s20_c, integral_constant<int, (cur ^ 1)>{}, integral_constant<int, MTI + 1>{});
            else
                read_a(SN{}, integral_constant<int, (cur ^ 1)>{}, I0{});
            if constexpr ((PSA_K1W_X & 8) != 0) __builtin_amdgcn_sched_barrier(0);      // the reads stay in front of the tile's MFMAs                          // row tile 0 of the next stage (landed: k = 3)
            auto comp = [&](auto c_c) __attribute__((always_inline)) {
                constexpr int CC = decltype(c_c)::value;
                f32x4 ch = PR::mma(a[cur][1], bf[CC][0], restart ? f32x4{0.f, 0.f, 0.f, 0.f} : hi[MTI][CC]);
                ch = PR::mma(a[cur][0], bf[CC][1], ch);
                hi[MTI][CC] = PR::mma(a[cur][0], bf[CC][0], ch);
                if constexpr (MTI == MT - 1) read_b(SN{}, c_c);       // behind the component's last use in this stage
            };
            comp(I0{});
            comp(I1{});
            comp(integral_constant<int, 2>{});
            if constexpr (PSA_K1W_SOLO == 0 && PSA_K1W_POS >= 0 && MTI == PSA_K1W_POS) {
                if (wh != 0) issue();
            }
            __builtin_amdgcn_sched_barrier(0);     // 232 registers are live by design: nothing moves across a row tile
            if constexpr (MTI == 0) PSA_STAMP(1);
            if constexpr (MTI == 3) PSA_STAMP(2);
            if constexpr (MTI == 7) PSA_STAMP(3);
        };
        [&]<int... Ms>(std::integer_sequence<int, Ms...>) __attribute__((always_inline)) { (tile(integral_constant<int, Ms>{}), ...); }(std::make_integer_sequence<int, MT>{});
        if constexpr (folds) fold();
        else if constexpr ((PSA_K1W_X & 4) != 0) {
            if (left == S20 + 1) fold();                                  // the last stage of the group
        }
        // What this stage read from its own units has been consumed by the MFMAs above (it has returned); the
        // reads still in flight come from stage s+1's units, which nothing overwrites before barrier s+1.
#if PSA_K1W_STAMP
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PENDING) : "memory");
        PSA_STAMP(4);
        asm volatile("s_barrier" ::: "memory");
        PSA_STAMP(5);
        ++st_n;
#else
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PENDING) : "memory");
#endif
    };
    if constexpr ((PSA_K1W_X & 2) != 0) {
        if (wh) __builtin_amdgcn_s_setprio(1);
    }
    for (; left > 0; left -= W_PERIOD)
        [&]<int... Ss>(std::integer_sequence<int, Ss...>) __attribute__((always_inline)) { (stage(integral_constant<int, Ss>{}), ...); }(
            std::make_integer_sequence<int, W_PERIOD>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // nothing in flight when LDS is handed on
#if PSA_K1W_STAMP
    if (blockIdx.x < 8 && lane == 0 && st_n)
        printf("[k1w stamp] block %d wave %d stages %u: dma %u  tile0 %u  tiles1-3 %u  tiles4-7 %u  fold+vmcnt %u  barrier %u  (cycles per stage)\n",
               (int)blockIdx.x, w, st_n, st_acc[0] / st_n, st_acc[1] / st_n, st_acc[2] / st_n, st_acc[3] / st_n, st_acc[4] / st_n, st_acc[5] / st_n);
#endif

    // ---- epilogue (k1_planes.hip): register j of lane (r16, q) is row 4q + j, column r16 of its 16x16 tile ----
    // (the lane's coordinates are taken afresh: carried through the loop they are two more live registers there)
    const int     lane_e = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int     r16_e = lane_e & 15, q_e = lane_e >> 4;
    const int     m0 = mb * W_M_BLK + wh * (W_M_BLK / 2);
    const int64_t t = (int64_t)tb * W_T_BLK + wf * 16 + r16_e;
    if (t < T) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const int k = (m0 + mt * 16 + 4 * q_e + 2 * pr) >> 1;
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

int launch_k1_planes_wide(psa_ctx* c, const void* d_planes, const void* d_phase, float2* d_q, const ProjGeom& g, int64_t n_fg) {
    PSA_REQUIRE(g.m_blk == W_M_BLK && g.M_pad % W_M_BLK == 0, "wide planes kernel: 256-row M blocks only");
    PSA_REQUIRE(g.A_pad % (2 * K1_BA) == 0 && g.A_pad > 0, "planes kernel needs the atom axis padded to %d", 2 * K1_BA);
    PSA_REQUIRE(g.vscale > 0.f && n_fg * 16 >= g.T, "planes do not cover the launch");
    const int     n_mblk = g.M_pad / W_M_BLK;
    const int64_t n_tblk = (g.T + W_T_BLK - 1) / W_T_BLK;
    const int64_t grid = ((n_tblk + 7) / 8) * 8 * n_mblk;
    PSA_REQUIRE(grid < (1ll << 31) && n_tblk < (1ll << 29) && n_fg < (1ll << 31), "projection grid too large");
    const float qscale = 1.f / (g.vscale * F16x2::P_SCALE);           // powers of two: exact
    if (!c->d_zeros.ptr) {
        PSA_TRY(c->d_zeros.reserve(1024));
        PSA_HIP_CHECK(hipMemsetAsync(c->d_zeros.ptr, 0, 1024, c->stream));
    }
    if (n_mblk == 1)
        hipLaunchKernelGGL((k1_planes_wide_kernel<true>), dim3((unsigned)grid), dim3(512), 0, c->stream, (const _Float16*)d_planes,
                           (const _Float16*)d_phase, c->d_zeros.ptr, d_q, g.T, g.q_stride, (int)n_fg, g.A_pad / K1_BA, g.K, n_mblk,
                           (int)n_tblk, qscale);
    else
        hipLaunchKernelGGL((k1_planes_wide_kernel<false>), dim3((unsigned)grid), dim3(512), 0, c->stream, (const _Float16*)d_planes,
                           (const _Float16*)d_phase, c->d_zeros.ptr, d_q, g.T, g.q_stride, (int)n_fg, g.A_pad / K1_BA, g.K, n_mblk,
                           (int)n_tblk, qscale);
    PSA_HIP_CHECK(hipGetLastError());
    return PSA_OK;
}

}  // namespace psa
