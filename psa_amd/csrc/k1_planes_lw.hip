// K1 on pre-split planes with DEDICATED LOADER WAVEFRONTS: the projection GEMM of k1_planes.hip
// (same arithmetic, same plane and phase-tile images, same 128-row x 64-frame workgroup tile, same
// float32-fold rule), restructured around what round 3 measured on that kernel
// (profiles/r3_k1_experiments.txt): with its LDS-DMA compiled out of the main loop the launch takes
// 11.3 ms instead of 14.8 on the same random operands -- a wavefront that issues a 1-KiB LDS-DMA
// instruction is held for 60-190 cycles and issues no MFMA meanwhile, and every one of the eight
// wavefronts has to issue five of them per 32-atom stage.
//
// Here a workgroup has TWELVE wavefronts, three per SIMD:
//   * wavefronts 0-7 compute, exactly the 64-row x 16-frame x 3-component tiles of k1_planes.hip, and
//     issue NO vector-memory instruction in the main loop: LDS fragment reads, MFMAs, one barrier
//     per stage;
//   * wavefronts 8-11 (one per SIMD) only load: wavefront 8+j copies a quarter of the stage's phase
//     tile (4 KiB) and frame group j's 6 KiB of planes -- ten LDS-DMA instructions per stage --
//     then waits (counted vmcnt) for the stage that must have landed and joins the barrier.
//
// Three wavefronts per SIMD need <= 168 VGPRs each, so the compute wavefronts hold fewer fragments:
// the MFMAs run COMPONENT-major -- for component c the two B fragments (8 registers, the next pair
// is fetched meanwhile: 16 in all instead of 48) meet all four row tiles' A fragments (32 registers,
// reloaded tile by tile for the next stage behind their last use, as before); the three MFMAs of a
// chain follow each other on one accumulator.  Because a stage's B fragments for c = 1, 2 are still
// being read from its slot during the stage, the loaders write one slot further back: a FOUR-slot
// ring (4 x 40 KiB = all 160 KiB of LDS), stage s+3 travelling into the slot of stage s-1 while
// stage s is computed -- two stages in flight, like the three-slot ring of k1_planes.hip.
//
// Per stage and CU: the same 40 KiB of LDS-DMA, 112 KiB of fragment reads, 288 MFMAs.
#include "k1_f16.h"

// Timing-only experiment (tools/k1_experiments.sh with KERNEL=k1_planes_lw MACRO=PSA_K1LW_X; results are WRONG):
//   1 = 28 KiB of LDS-DMA per stage instead of 40 (what a 256-row workgroup tile would need per 128 rows):
//       each loader skips half of its phase quarter and its last plane piece, the compute wavefronts read
//       only what was refreshed (row tiles 0, 1 twice; component 2's second piece from component 1)
//   2 = every stage fetches the bytes of stage 0 (all LDS-DMA served by L2)
#ifndef PSA_K1LW_X
#define PSA_K1LW_X 0
#endif

namespace psa {

template <bool NT_V>
__global__ void __launch_bounds__(768, 3)
k1_planes_lw_kernel(const _Float16* __restrict__ planes, const _Float16* __restrict__ Pb, float2* __restrict__ Q,
                    int64_t T, int64_t q_stride, int n_fg, int n_stage, int K, int n_mblk, int n_tblk, float qscale) {
    using PR = F16x2;
    using E8 = PR::v8;
    constexpr int NP = PR::NP, MT16 = 4, M_BLK = 128, T_BLK = 64, FOLD = 8, RING = 4;
    constexpr int P_STAGE_BYTES = NP * M_BLK * K1_BA * 2;                   // 16 KiB
    constexpr int V_GROUP_BYTES = PL_STAGE_ELEMS * 2;                       // 6 KiB
    constexpr int STAGE_BYTES = P_STAGE_BYTES + 4 * V_GROUP_BYTES;          // 40 KiB
    constexpr bool LESS_DMA = (PSA_K1LW_X & 1) != 0;
    constexpr int  BATCH = LESS_DMA ? 7 : 10;                               // LDS-DMA instructions per loader and stage
    static_assert(RING * STAGE_BYTES == 160 * 1024, "the ring is all of LDS");
    __shared__ __attribute__((aligned(16))) unsigned char smem[RING * STAGE_BYTES];
    const unsigned lds0 = (unsigned)(size_t)(lds_u8*)smem;

    // XCD-aware block map (k1_planes.hip): blocks b and b+8 share an XCD and get the M blocks of one frame tile
    const int b = blockIdx.x;
    const int r8 = b >> 3;
    const int mb = r8 % n_mblk;
    const int tb = (r8 / n_mblk) * 8 + (b & 7);
    if (tb >= n_tblk) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);

    if (w >= 8) {
        // ================================ loader wavefront j ================================
        const int j = w - 8;
        int       fg = tb * 4 + j;                                         // frame group (past the end: the last one, never stored)
        if (fg >= n_fg) fg = n_fg - 1;
        const unsigned char* p_next = reinterpret_cast<const unsigned char*>(Pb) + (size_t)mb * n_stage * P_STAGE_BYTES;
        const unsigned char* v_next = reinterpret_cast<const unsigned char*>(planes) + (size_t)fg * n_stage * V_GROUP_BYTES;
        const unsigned       p_voff = 4096 * j + 16 * lane, v_voff = 16 * lane;
        // stages past the end are fetched from the bytes that follow (next M block / frame group, or the
        // padding behind the buffers) and never read -- as in k1_planes.hip
        auto dma_stage = [&](int slot) {
            const unsigned dst = lds0 + slot * STAGE_BYTES;
            lds_dma16_group<LESS_DMA ? 2 : 4>(p_next, p_voff, dst + 4096 * j);
            lds_dma16_group<3, NT_V>(v_next, v_voff, dst + P_STAGE_BYTES + j * V_GROUP_BYTES);
            lds_dma16_group<LESS_DMA ? 2 : 3, NT_V>(v_next, v_voff + 3072, dst + P_STAGE_BYTES + j * V_GROUP_BYTES + 3072);
            if constexpr (!(PSA_K1LW_X & 2)) {
                p_next += P_STAGE_BYTES;
                v_next += V_GROUP_BYTES;
            }
        };
        __builtin_amdgcn_s_setprio(2);     // a loader's few instructions go first (priorities 0 and 3 measured the same)
        dma_stage(0);
        dma_stage(1);
        dma_stage(2);
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * BATCH) : "memory");     // stage 0 landed
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(1 * BATCH) : "memory");     // stage 1 landed
        int slot = 3;                                                      // stage s + 3 goes to slot (s + 3) % 4
        for (int s = 0; s < n_stage; ++s) {
            dma_stage(slot);
            slot = slot == RING - 1 ? 0 : slot + 1;
            // all but the batch just issued complete: stage s + 2 has landed
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(BATCH) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // nothing in flight when LDS is handed on
        return;
    }

    // ==================================== compute wavefront ====================================
    const int     wh = w >> 2, wf = w & 3;
    const int     r16 = lane & 15, q = lane >> 4;
    const int64_t t0 = (int64_t)tb * T_BLK + wf * 16;
    const int      gsw = pl_swizzle(r16);
    const unsigned p_lane = lds0 + (wh * (M_BLK / 2) + r16) * (K1_BA * 2) + ((q ^ gsw) << 4);
    const unsigned v_lane = lds0 + P_STAGE_BYTES + wf * V_GROUP_BYTES + r16 * (K1_BA * 2) + ((q ^ gsw) << 4);
    E8    a[NP][MT16];                             // A fragments of the stage in work
    E8    bb[2][NP];                               // B fragments: component in work / the next one
    f32x4 hi[MT16][3], lo[MT16][3];
    auto  read_a_tile = [&](int mt, int slot) {
        const unsigned base = p_lane + slot * STAGE_BYTES;
#pragma unroll
        for (int p = 0; p < NP; ++p)
            a[p][mt] = *reinterpret_cast<lds_cv8*>((const lds_u8*)(size_t)(base + (p * M_BLK + (LESS_DMA ? mt & 1 : mt) * 16) * 64));
    };
    auto read_b = [&](int buf, int slot, int c) {
#pragma unroll
        for (int p = 0; p < NP; ++p)
            bb[buf][p] = *reinterpret_cast<lds_cv8*>(
                (const lds_u8*)(size_t)(v_lane + slot * STAGE_BYTES + ((LESS_DMA && c == 2 && p == 1 ? 1 : c) * NP + p) * 1024));
    };
#pragma unroll
    for (int mt = 0; mt < MT16; ++mt)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            hi[mt][c] = f32x4{0.f, 0.f, 0.f, 0.f};
            lo[mt][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

    asm volatile("s_barrier" ::: "memory");                                // stage 0 landed (loaders waited for it)
    read_b(0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < MT16; ++mt) read_a_tile(mt, 0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");        // stage 1 landed

    // One stage: A of stage s and B (component 0) of stage s are in registers, in bb[par]; slot holds the
    // stage's other B fragments, slot1 stage s + 1.
    auto stage = [&](auto par_c, auto restart_c, int slot) {
        constexpr int  par = decltype(par_c)::value;        // buffer of component 0; the components alternate from there
        constexpr bool restart = decltype(restart_c)::value;
        const int      slot1 = slot == RING - 1 ? 0 : slot + 1;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int cur = (par + c) & 1;
            if (c < 2)
                read_b(cur ^ 1, slot, c + 1);
            else
                read_b(cur ^ 1, slot1, 0);
#pragma unroll
            for (int mt = 0; mt < MT16; ++mt) {
                f32x4 ch = PR::mma(a[1][mt], bb[cur][0], restart ? f32x4{0.f, 0.f, 0.f, 0.f} : hi[mt][c]);
                ch = PR::mma(a[0][mt], bb[cur][1], ch);
                hi[mt][c] = PR::mma(a[0][mt], bb[cur][0], ch);
                if (c == 2) read_a_tile(mt, slot1);        // behind the tile's last MFMAs of this stage
            }
        }
        // No wait for the fragment reads here.  What this stage read from its OWN slot (B of c = 1, 2) has been
        // consumed by the MFMAs above, so it has returned; that slot is overwritten right after this barrier.
        // What it read from the NEXT slot (A, and B of c = 0, of stage s + 1) is consumed by the MFMAs of
        // stage s + 1 -- behind the compiler's own counted waits -- and that slot is overwritten only after the
        // barrier that ends stage s + 1.  (The three-slot ring of k1_planes.hip overwrites the next slot at
        // once and has to wait here.)
#ifdef PSA_K1LW_WAIT_READS
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
        asm volatile("s_barrier" ::: "memory");
#endif
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    int  slot = 0;
    auto next_slot = [&]() { slot = slot == RING - 1 ? 0 : slot + 1; };
    for (int s = 0; s < n_stage;) {                    // n_stage is even; three components per stage flip the buffer parity
        const int len = n_stage - s < FOLD ? n_stage - s : FOLD;
        stage(I0{}, std::true_type{}, slot);
        next_slot();
        stage(I1{}, std::false_type{}, slot);
        next_slot();
        for (int i = 2; i < len; i += 2) {
            stage(I0{}, std::false_type{}, slot);
            next_slot();
            stage(I1{}, std::false_type{}, slot);
            next_slot();
        }
#pragma unroll
        for (int mt = 0; mt < MT16; ++mt)
#pragma unroll
            for (int c = 0; c < 3; ++c) lo[mt][c] += hi[mt][c];
        s += len;
    }

    // epilogue (k1_planes.hip): register j of lane (r16, q) is row 4q + j, column r16 of its 16x16 tile
    const int     m0 = mb * M_BLK + wh * (M_BLK / 2);
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

// the 128-row shapes of launch_k1_planes (k1_planes.hip)
int launch_k1_planes_lw(psa_ctx* c, const void* d_planes, const void* d_phase, float2* d_q, const ProjGeom& g, int64_t n_fg) {
    PSA_REQUIRE(g.m_blk == 128 && g.M_pad % 128 == 0, "loader-wave planes kernel: 128-row M blocks only");
    PSA_REQUIRE(g.A_pad % (2 * K1_BA) == 0 && g.A_pad > 0, "planes kernel needs the atom axis padded to %d", 2 * K1_BA);
    PSA_REQUIRE(g.vscale > 0.f && n_fg * 16 >= g.T, "planes do not cover the launch");
    const int     n_mblk = g.M_pad / 128;
    const int64_t n_tblk = (g.T + 63) / 64;
    const int64_t grid = ((n_tblk + 7) / 8) * 8 * n_mblk;
    PSA_REQUIRE(grid < (1ll << 31) && n_tblk < (1ll << 29) && n_fg < (1ll << 31), "projection grid too large");
    const float qscale = 1.f / (g.vscale * F16x2::P_SCALE);           // powers of two: exact
    if (n_mblk == 1)        // (nt with several M blocks: the siblings start to miss -- 38.7 GB fetched instead of 34.4, +2 % time)
        hipLaunchKernelGGL((k1_planes_lw_kernel<true>), dim3((unsigned)grid), dim3(768), 0, c->stream, (const _Float16*)d_planes,
                           (const _Float16*)d_phase, d_q, g.T, g.q_stride, (int)n_fg, g.A_pad / K1_BA, g.K, n_mblk, (int)n_tblk,
                           qscale);
    else
        hipLaunchKernelGGL((k1_planes_lw_kernel<false>), dim3((unsigned)grid), dim3(768), 0, c->stream, (const _Float16*)d_planes,
                           (const _Float16*)d_phase, d_q, g.T, g.q_stride, (int)n_fg, g.A_pad / K1_BA, g.K, n_mblk, (int)n_tblk,
                           qscale);
    PSA_HIP_CHECK(hipGetLastError());
    return PSA_OK;
}

}  // namespace psa
