// K1, exact-fp32 form -- the k-projection on the fp32 matrix cores.  It serves every atom
// group the split-precision kernel (k1_split.hip, the default for whole-trajectory groups) does
// not take: index lists, displacement mode, atom counts that are not a multiple of 4; and any
// group when PSA_K1_MFMA32 is selected.
//
//   q[k, c, t] = sum_a  d[t, a, c] * P[k, a]          d real float32, P complex64
//
// ref: src/psa/core/sed_calculator.py:80-81 (three einsum('ta,ak->tk') calls, i.e.
// a complex GEMM per Cartesian component).  Written here as ONE real GEMM
//     D[m, (c,t)] = sum_a P'[m, a] * d[t, a, c],   m = 2k (cos row) | 2k+1 (sin row)
// on the fp32 matrix cores: v_mfma_f32_32x32x2_f32, exact float32 FMA chains, so
// the result has the reference's precision class (no bf16/tf32 shortcuts).
//
// Work decomposition (256 threads = 4 wavefronts of 64 per workgroup, 1 workgroup/CU):
//   workgroup tile : M_BLK = 32*MT*WM rows of P'  x  T_BLK = 32*WN frames x 3 components
//   wavefront tile : MT row-tiles x 1 frame-tile x 3 components  -> MT*3 accumulators
//                    of 32x32 (16 AGPRs each); MT=4 -> 192 accumulator registers
//   atom loop      : stages of BA = 32 atoms in a 2-deep LDS ring.  Stage s+1 travels
//                    HBM -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging VGPRs)
//                    while the MFMAs of stage s run; one vmcnt(0) + barrier per stage.
//
// LDS images (per stage)
//   Vs[T_BLK][24 slots of 16 B] : row t = the 96 floats d[t, a0..a0+31, 0..2] exactly as
//       they lie in HBM (component-minor); slot s of row t is stored at physical slot
//       (s & ~7) | ((s & 7) ^ (t & 7))  (XOR swizzle: LDS-DMA writes 1 KiB linearly per
//       wave-instruction, so rows cannot be padded; the swizzle goes on the per-lane
//       SOURCE address and on the read).  A lane's B fragment for four MFMA k-steps x
//       three components = logical slots 6g+3h .. +2 of its row (48 contiguous bytes in
//       HBM): three ds_read_b128, at most 2-way bank conflicts.
//   Ps[M_BLK][36 floats]        : row m = P'[m, a0..a0+31] + 4 pad floats.  The phase kernel
//       writes P' to HBM already in this tile image, so the DMA is a linear copy; the odd
//       9-slot pitch makes the A-fragment ds_read_b128 conflict-free.
//
// MFMA operand map (32x32x2 f32): lane l supplies A[i = l&31][kk = l>>5] and
// B[kk = l>>5][j = l&31].  The contraction index may be permuted freely, so k-step s
// of k-group g uses atom a0 + 8g + 4*(l>>5) + s on both operands: the four atoms a
// lane needs are contiguous in both LDS images.
//
// Accumulator map: register r of lane l is row (r&3) + 8*(r>>2) + 4*(l>>5), column l&31.
// Rows 2p, 2p+1 (cos, sin of one k) sit in registers 2p, 2p+1 of the same lane, so the
// epilogue stores float2 = one complex64 per lane, 32 consecutive frames per half-wave:
// q is written k-major (K,3,T) with t contiguous, which is what the batched FFT wants.
//
// Roofline: 12 flop per (k,t,atom); V is read once per M-block (12 B per (t,atom)).
// At K >= ~25 k-points per device the kernel is bound by the fp32 MFMA rate
// (157.3 TFLOP/s), below that by HBM.
#include "psa_ctx.h"

namespace psa {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

template <int MT, int WM, int WN>
struct K1Cfg {
    static constexpr int M_BLK = 32 * MT * WM;
    static constexpr int T_BLK = 32 * WN;
    static constexpr int V_STAGE = T_BLK * K1_VROW;   // floats
    static constexpr int P_STAGE = M_BLK * K1_PROW;
    static constexpr int LDS_BYTES = 2 * (V_STAGE + P_STAGE) * 4;
    static constexpr int V_DMA = T_BLK * 24 / 256;      // V DMA instructions per wave per stage
    static constexpr int P_CHUNKS = P_STAGE / 4;        // 16-byte chunks in a P' tile
    static constexpr int P_DMA = (P_CHUNKS + 255) / 256;
    static constexpr int V_ITEMS = T_BLK * K1_BA / 256; // (t,atom) items per thread (register loader)
    static_assert(WM * WN == 4, "4 wavefronts per workgroup");
    static_assert((T_BLK * 24) % 256 == 0, "V tile must be whole wave-instructions");
};

__device__ __forceinline__ int v_phys_slot(int s, int row) { return (s & ~7) | ((s & 7) ^ (row & 7)); }

// VDMA  : the group is "all atoms in order", N % 4 == 0, no displacement: the V tile is
//         copied HBM -> LDS by LDS-DMA.
// !VDMA : arbitrary index list (duplicates, any order), unaligned N, or displacement mode
//         (positions - mean, sed_calculator.py:70): one (t, atom) item = three dword loads
//         through registers.  P' always arrives by DMA.
template <int MT, int WM, int WN, bool VDMA, bool DISP>
__global__ void __launch_bounds__(256, 1)
k1_mfma_kernel(const float* __restrict__ V, const float* __restrict__ P,
               const int* __restrict__ idx, const float* __restrict__ mean_g,
               float2* __restrict__ Q, int64_t T, int64_t q_stride, int64_t N_tot, int n_g, int A_pad, int K,
               int n_mblk, int n_tblk) {
    using C = K1Cfg<MT, WM, WN>;
    __shared__ __attribute__((aligned(16))) float smem[C::LDS_BYTES / 4];
    float* Vs = smem;                       // [2][T_BLK][96]   swizzled slots
    float* Ps = smem + 2 * C::V_STAGE;      // [2][M_BLK][36]

    // XCD-aware block map: blocks b and b+8 share an XCD (and its L2); give them the
    // M-blocks of one frame tile so the second read of that V tile is an L2 hit.
    const int b  = blockIdx.x;
    const int r8 = b >> 3;
    const int mb = r8 % n_mblk;
    const int tb = (r8 / n_mblk) * 8 + (b & 7);
    if (tb >= n_tblk) return;

    const int     tid  = threadIdx.x;
    const int     lane = tid & 63;
    const int     wave = tid >> 6;
    const int     wm = wave % WM, wn = wave / WM;
    const int     l31 = lane & 31, h = lane >> 5;
    const int64_t t0 = (int64_t)tb * C::T_BLK;
    const int     n_stage = A_pad / K1_BA;
    const float*  Pt = P + (size_t)mb * n_stage * C::P_STAGE;     // this M-block's tile images

    // ---- V by LDS-DMA: per-lane source offsets (loop invariant) -------------------------
    // DMA instruction j of this wave fills LDS slots L = (wave*V_DMA + j)*64 + lane.
    int64_t v_row[VDMA ? C::V_DMA : 1];
    int     v_s4[VDMA ? C::V_DMA : 1];
    if constexpr (VDMA) {
#pragma unroll
        for (int j = 0; j < C::V_DMA; ++j) {
            const int L = (wave * C::V_DMA + j) * 64 + lane;
            const int row = L / 24, phys = L - row * 24;
            int64_t   t = t0 + row;
            if (t >= T) t = T - 1;                              // rows past the end: finite filler
            v_row[j] = t * 3 * N_tot;
            v_s4[j] = 4 * v_phys_slot(phys, row);               // the swizzle is an involution
        }
    }
    float vx[VDMA ? 1 : C::V_ITEMS], vy[VDMA ? 1 : C::V_ITEMS], vz[VDMA ? 1 : C::V_ITEMS];

    auto dma_stage = [&](int st, int buf) {
        // P': linear copy of the tile image, 1 KiB per wave-instruction
        const float* src = Pt + (size_t)st * C::P_STAGE;
        float*       dst = Ps + buf * C::P_STAGE;
#pragma unroll
        for (int j = 0; j < C::P_DMA; ++j) {
            const int chunk = (wave * C::P_DMA + j) * 64;        // first 16-byte chunk of this instruction
            if constexpr (C::P_CHUNKS % 256 == 0) {
                // every wave issues P_DMA full instructions: straight-line code
                __builtin_amdgcn_global_load_lds((gbl_void*)(src + 4 * (chunk + lane)),
                                                 (lds_void*)(dst + 4 * chunk), 16, 0, 0);
            } else {
                // wave-uniform outer test; the last instruction of a 32-row tile is half
                // full, its upper lanes stay masked off (a masked lane writes nothing)
                if (chunk < C::P_CHUNKS && chunk + lane < C::P_CHUNKS)
                    __builtin_amdgcn_global_load_lds((gbl_void*)(src + 4 * (chunk + lane)),
                                                     (lds_void*)(dst + 4 * chunk), 16, 0, 0);
            }
        }
        if constexpr (VDMA) {
            const int a0 = st * K1_BA;
            float*    vd = Vs + buf * C::V_STAGE;
#pragma unroll
            for (int j = 0; j < C::V_DMA; ++j) {
                // atoms past the group's end carry P' = 0: feed them the stage's first chunk
                const int s4 = (3 * a0 + v_s4[j] + 3 < 3 * n_g) ? v_s4[j] : 0;
                __builtin_amdgcn_global_load_lds((gbl_void*)(V + v_row[j] + 3 * a0 + s4),
                                                 (lds_void*)(vd + (wave * C::V_DMA + j) * 256), 16, 0, 0);
            }
        }
    };

    auto load_items = [&](int a0) {
        if constexpr (!VDMA) {
#pragma unroll
            for (int j = 0; j < C::V_ITEMS; ++j) {
                const int     q = tid + 256 * j;
                const int     row = q >> 5, al = q & 31;
                const int     a = a0 + al;
                const int64_t t = t0 + row;
                float x = 0.f, y = 0.f, z = 0.f;
                if (t < T && a < n_g) {
                    const int64_t src = idx ? idx[a] : a;
                    const float*  p = V + (t * N_tot + src) * 3;
                    x = p[0]; y = p[1]; z = p[2];
                    if constexpr (DISP) {
                        x -= mean_g[3 * a + 0]; y -= mean_g[3 * a + 1]; z -= mean_g[3 * a + 2];
                    }
                }
                vx[j] = x; vy[j] = y; vz[j] = z;
            }
        }
    };

    auto store_items = [&](int buf) {
        if constexpr (!VDMA) {
            float* vs = Vs + buf * C::V_STAGE;
#pragma unroll
            for (int j = 0; j < C::V_ITEMS; ++j) {
                const int q = tid + 256 * j;
                const int row = q >> 5, al = q & 31;
                const float val[3] = {vx[j], vy[j], vz[j]};
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int e = 3 * al + c;                      // float index in the row
                    vs[row * K1_VROW + 4 * v_phys_slot(e >> 2, row) + (e & 3)] = val[c];
                }
            }
        }
    };

    f32x16 acc[MT][3];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][c][r] = 0.f;

    const int vrow = wn * 32 + l31;
    const int prow = wm * MT * 32 + l31;

    auto load_frags = [&](int buf, int g, float4 (&a)[MT], float4 (&bv)[3]) {
        const float* ps = Ps + buf * C::P_STAGE + prow * K1_PROW + 4 * h + 8 * g;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            a[mt] = *reinterpret_cast<const float4*>(ps + mt * 32 * K1_PROW);
        const float* vs = Vs + buf * C::V_STAGE + vrow * K1_VROW;
#pragma unroll
        for (int j = 0; j < 3; ++j)
            bv[j] = *reinterpret_cast<const float4*>(vs + 4 * v_phys_slot(6 * g + 3 * h + j, vrow));
    };

    auto mfma_group = [&](const float4 (&a)[MT], const float4 (&bv)[3]) {
        const float bb[12] = {bv[0].x, bv[0].y, bv[0].z, bv[0].w, bv[1].x, bv[1].y, bv[1].z, bv[1].w,
                              bv[2].x, bv[2].y, bv[2].z, bv[2].w};
#pragma unroll
        for (int st = 0; st < 4; ++st) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const float av = st == 0 ? a[mt].x : st == 1 ? a[mt].y : st == 2 ? a[mt].z : a[mt].w;
                    acc[mt][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bb[3 * st + c], acc[mt][c], 0, 0, 0);
                }
            }
        }
    };

    // One stage = 4 k-groups of 8 atoms.  Schedule (pinned with sched_barrier, because the
    // compiler otherwise sinks the LDS reads next to their first use):
    //   read g0,g1 | MFMA g0 | read g2 | MFMA g1 | read g3 | DMA(next stage) + MFMA g2 | MFMA g3
    // - the fragments of group g+1 are in flight behind the 12*MT MFMAs of group g;
    // - the next stage's LDS-DMA is issued only after this stage's LAST LDS read: hipcc
    //   (ROCm 7.2) cannot prove that an in-flight LDS-DMA does not alias a later ds_read and
    //   puts s_waitcnt vmcnt(0) in front of it, so issuing earlier would serialise load and
    //   compute.  Issued here, the DMA still has two MFMA groups (~6k cycles) to land, and
    //   its 1-KiB pieces are interleaved one per MFMA (sched_group_barrier) so that their
    //   issue cost hides in the 64-cycle MFMA shadow.
    constexpr int N_DMA = (VDMA ? C::V_DMA : 0) + C::P_DMA;
    auto compute_stage = [&](int buf, bool prefetch, int next_stage) {
        float4 a0[MT], b0[3], a1[MT], b1[3];
        load_frags(buf, 0, a0, b0);
        load_frags(buf, 1, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        load_frags(buf, 2, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        load_frags(buf, 3, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        if (prefetch) {
            dma_stage(next_stage, buf ^ 1);
            load_items(next_stage * K1_BA);
        }
        mfma_group(a0, b0);
#pragma unroll
        for (int i = 0; i < N_DMA; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);   // one VMEM (an LDS-DMA piece)
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
    };

    // prologue: stage 0
    dma_stage(0, 0);
    load_items(0);
    store_items(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int s = 0; s + 1 < n_stage; ++s) {
        const int buf = s & 1;
        compute_stage(buf, true, s + 1);
        store_items(buf ^ 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    compute_stage((n_stage - 1) & 1, false, 0);

    // epilogue: complex64 q[k][c][t], 32 consecutive frames per half-wave
    const int64_t t = t0 + wn * 32 + l31;
    const int     m0 = mb * C::M_BLK;
    if (t < T) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int r0 = 2 * p;
                const int i = (r0 & 3) + 8 * (r0 >> 2) + 4 * h;
                const int k = (m0 + (wm * MT + mt) * 32 + i) >> 1;
                if (k < K) {
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        Q[((int64_t)k * 3 + c) * q_stride + t] = make_float2(acc[mt][c][r0], acc[mt][c][r0 + 1]);
                }
            }
        }
    }
}

int k1_mfma_block_rows(int K) {
    const int M = 2 * K;
    if (M <= 32) return 32;
    if (M <= 64) return 64;
    if (M <= 128) return 128;
    return 256;
}

template <int MT, int WM, int WN, bool VDMA, bool DISP>
static int launch_variant(psa_ctx* c, const float* d_v, const float* d_phase, const int* d_idx,
                          const float* d_mean_g, float2* d_q, const ProjGeom& g) {
    using C = K1Cfg<MT, WM, WN>;
    auto kern = k1_mfma_kernel<MT, WM, WN, VDMA, DISP>;
    const int     n_mblk = g.M_pad / C::M_BLK;
    const int64_t n_tblk = (g.T + C::T_BLK - 1) / C::T_BLK;
    const int64_t grid = ((n_tblk + 7) / 8) * 8 * n_mblk;
    PSA_REQUIRE(grid < (1ll << 31) && n_tblk < (1ll << 31), "projection grid too large");
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), 0, c->stream, d_v, d_phase,
                       d_idx, d_mean_g, d_q, g.T, g.q_stride, g.N_tot, g.n_g, g.A_pad, g.K, n_mblk, (int)n_tblk);
    PSA_HIP_CHECK(hipGetLastError());
    return PSA_OK;
}

template <int MT, int WM, int WN>
static int launch_shape(psa_ctx* c, const float* d_v, const float* d_phase, const int* d_idx,
                        const float* d_mean_g, float2* d_q, const ProjGeom& g, bool disp) {
    const bool vdma = !disp && (d_idx == nullptr) && (g.N_tot % 4 == 0) && (g.n_g == g.N_tot);
    if (vdma) return launch_variant<MT, WM, WN, true, false>(c, d_v, d_phase, d_idx, d_mean_g, d_q, g);
    return disp ? launch_variant<MT, WM, WN, false, true>(c, d_v, d_phase, d_idx, d_mean_g, d_q, g)
                : launch_variant<MT, WM, WN, false, false>(c, d_v, d_phase, d_idx, d_mean_g, d_q, g);
}

int launch_k1_mfma(psa_ctx* c, const float* d_v, const float* d_phase, const int* d_idx,
                   const float* d_mean_g, float2* d_q, const ProjGeom& g, bool displacements) {
    PSA_REQUIRE(g.A_pad % K1_BA == 0 && g.A_pad >= K1_BA, "A_pad must be a positive multiple of %d", K1_BA);
    PSA_REQUIRE(g.M_pad % g.m_blk == 0, "M_pad not a multiple of the M block");
    switch (g.m_blk) {
        case 32:  return launch_shape<1, 1, 4>(c, d_v, d_phase, d_idx, d_mean_g, d_q, g, displacements);
        case 64:  return launch_shape<2, 1, 4>(c, d_v, d_phase, d_idx, d_mean_g, d_q, g, displacements);
        case 128: return launch_shape<4, 1, 4>(c, d_v, d_phase, d_idx, d_mean_g, d_q, g, displacements);
        case 256: return launch_shape<4, 2, 2>(c, d_v, d_phase, d_idx, d_mean_g, d_q, g, displacements);
    }
    set_error("no projection variant for M block %d", g.m_blk);
    return PSA_EINVAL;
}

}  // namespace psa
