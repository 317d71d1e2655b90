// K1 -- the k-projection kernel (the hot kernel of the path).
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
//                    of 32x32 (16 VGPRs each); MT=4 -> 192 accumulator VGPRs
//   atom loop      : stages of BA = 32 atoms, double-buffered in LDS; the next stage's
//                    global loads are issued before the MFMAs of the current one and
//                    written to the other LDS buffer after them (one barrier per stage)
//
// LDS images (per stage)
//   Vs[T_BLK][100] : row t holds the 96 floats d[t, a0..a0+31, 0..2] exactly as they lie
//                    in HBM (component-minor), +4 floats of padding.  A lane reads 48
//                    contiguous bytes = 4 atoms x 3 components with three ds_read_b128:
//                    that is the B fragment of all three components for four MFMA
//                    k-steps -- the (atom, component) interleave costs nothing.
//   Ps[M_BLK][36]  : row m holds P'[m, a0..a0+31], +4 floats of padding; one ds_read_b128
//                    = the A fragment of four k-steps.
//   Row pitches of 25 and 9 sixteen-byte slots (odd) keep every 16-lane ds_read_b128
//   group on 16 distinct slots of the 256-byte bank row: conflict-free.
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

constexpr int BA   = 32;           // atoms per LDS stage
constexpr int VROW = 3 * BA + 4;   // floats per Vs row (100 -> 25 slots of 16 B)
constexpr int PROW = BA + 4;       // floats per Ps row (36 -> 9 slots)

template <int MT, int WM, int WN>
struct K1Cfg {
    static constexpr int M_BLK = 32 * MT * WM;
    static constexpr int T_BLK = 32 * WN;
    static constexpr int V_STAGE = T_BLK * VROW;   // floats
    static constexpr int P_STAGE = M_BLK * PROW;
    static constexpr int LDS_BYTES = 2 * (V_STAGE + P_STAGE) * 4;
    static constexpr int V_CHUNKS = T_BLK * 24 / 256;   // 16-byte chunks per thread (VEC loader)
    static constexpr int V_ITEMS  = T_BLK * BA / 256;   // (t,atom) items per thread (ATOM loader)
    static constexpr int P_CHUNKS = M_BLK * 8 / 256;
    static_assert(WM * WN == 4, "4 wavefronts per workgroup");
    static_assert(P_CHUNKS >= 1, "P tile smaller than one chunk per thread");
};

// VEC   : the group is "all atoms in order" and N % 4 == 0 -> rows are 16-byte aligned,
//         the tile is copied with global_load_dwordx4.
// !VEC  : arbitrary index list (duplicates, any order) or unaligned N: one (t, atom)
//         item = three dword loads.
// DISP  : subtract the group's mean positions while staging (sed_calculator.py:70).
template <int MT, int WM, int WN, bool VEC, bool DISP>
__global__ void __launch_bounds__(256, 1)
k1_mfma_kernel(const float* __restrict__ V, const float* __restrict__ P,
               const int* __restrict__ idx, const float* __restrict__ mean_g,
               float2* __restrict__ Q, int64_t T, int64_t N_tot, int n_g, int A_pad, int K,
               int n_mblk, int n_tblk) {
    using C = K1Cfg<MT, WM, WN>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Vs = smem;                       // [2][T_BLK][VROW]
    float* Ps = smem + 2 * C::V_STAGE;      // [2][M_BLK][PROW]

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
    const int     m0 = mb * C::M_BLK;
    const int64_t row_floats = 3 * N_tot;

    float4 vreg[VEC ? C::V_CHUNKS : 1];
    float  vx[VEC ? 1 : C::V_ITEMS], vy[VEC ? 1 : C::V_ITEMS], vz[VEC ? 1 : C::V_ITEMS];
    float4 preg[C::P_CHUNKS];

    auto load_stage = [&](int a0) {
        if constexpr (VEC) {
#pragma unroll
            for (int j = 0; j < C::V_CHUNKS; ++j) {
                const int     q = tid + 256 * j;
                const int     row = q / 24, c16 = q - row * 24;
                const int64_t t = t0 + row;
                const int     off = 3 * a0 + 4 * c16;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (t < T && off < 3 * n_g) {
                    v = *reinterpret_cast<const float4*>(V + t * row_floats + off);
                    if constexpr (DISP) {
                        const float4 m = *reinterpret_cast<const float4*>(mean_g + off);
                        v.x -= m.x; v.y -= m.y; v.z -= m.z; v.w -= m.w;
                    }
                }
                vreg[j] = v;
            }
        } else {
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
#pragma unroll
        for (int j = 0; j < C::P_CHUNKS; ++j) {
            const int q = tid + 256 * j;
            const int row = q >> 3, c16 = q & 7;
            preg[j] = *reinterpret_cast<const float4*>(P + (size_t)(m0 + row) * A_pad + a0 + 4 * c16);
        }
    };

    auto store_stage = [&](int buf) {
        float* vs = Vs + buf * C::V_STAGE;
        float* ps = Ps + buf * C::P_STAGE;
        if constexpr (VEC) {
#pragma unroll
            for (int j = 0; j < C::V_CHUNKS; ++j) {
                const int q = tid + 256 * j;
                const int row = q / 24, c16 = q - row * 24;
                *reinterpret_cast<float4*>(vs + row * VROW + 4 * c16) = vreg[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < C::V_ITEMS; ++j) {
                const int q = tid + 256 * j;
                const int row = q >> 5, al = q & 31;
                float* d = vs + row * VROW + 3 * al;
                d[0] = vx[j]; d[1] = vy[j]; d[2] = vz[j];
            }
        }
#pragma unroll
        for (int j = 0; j < C::P_CHUNKS; ++j) {
            const int q = tid + 256 * j;
            const int row = q >> 3, c16 = q & 7;
            *reinterpret_cast<float4*>(ps + row * PROW + 4 * c16) = preg[j];
        }
    };

    f32x16 acc[MT][3];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][c][r] = 0.f;

    auto compute_stage = [&](int buf) {
        const float* vs = Vs + buf * C::V_STAGE + (wn * 32 + l31) * VROW + 12 * h;
        const float* ps = Ps + buf * C::P_STAGE + (wm * MT * 32 + l31) * PROW + 4 * h;
#pragma unroll
        for (int g = 0; g < BA / 8; ++g) {
            float4 a[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                a[mt] = *reinterpret_cast<const float4*>(ps + mt * 32 * PROW + 8 * g);
            const float4 b0 = *reinterpret_cast<const float4*>(vs + 24 * g);
            const float4 b1 = *reinterpret_cast<const float4*>(vs + 24 * g + 4);
            const float4 b2 = *reinterpret_cast<const float4*>(vs + 24 * g + 8);
            const float bb[12] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w,
                                  b2.x, b2.y, b2.z, b2.w};
#pragma unroll
            for (int st = 0; st < 4; ++st) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const float av = st == 0 ? a[mt].x : st == 1 ? a[mt].y : st == 2 ? a[mt].z : a[mt].w;
                        acc[mt][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bb[3 * st + c],
                                                                           acc[mt][c], 0, 0, 0);
                    }
                }
            }
        }
    };

    // atom loop: stage s is computed from LDS buffer s&1 while stage s+1 travels
    // HBM -> registers -> the other buffer; the last stage is peeled so that the
    // staging registers are never conditionally live.
    const int n_stage = A_pad / BA;
    load_stage(0);
    store_stage(0);
    __syncthreads();
    for (int s = 0; s + 1 < n_stage; ++s) {
        load_stage((s + 1) * BA);
        compute_stage(s & 1);
        store_stage((s & 1) ^ 1);
        __syncthreads();
    }
    compute_stage((n_stage - 1) & 1);

    // epilogue: complex64 q[k][c][t], 32 consecutive frames per half-wave
    const int64_t t = t0 + wn * 32 + l31;
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
                        Q[((int64_t)k * 3 + c) * T + t] = make_float2(acc[mt][c][r0], acc[mt][c][r0 + 1]);
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

template <int MT, int WM, int WN, bool VEC, bool DISP>
static int launch_variant(psa_ctx* c, const float* d_v, const float* d_phase, const int* d_idx,
                          const float* d_mean_g, float2* d_q, const ProjGeom& g) {
    using C = K1Cfg<MT, WM, WN>;
    auto kern = k1_mfma_kernel<MT, WM, WN, VEC, DISP>;
    PSA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
    const int     n_mblk = g.M_pad / C::M_BLK;
    const int64_t n_tblk = (g.T + C::T_BLK - 1) / C::T_BLK;
    const int64_t grid = ((n_tblk + 7) / 8) * 8 * n_mblk;
    PSA_REQUIRE(grid < (1ll << 31) && n_tblk < (1ll << 31), "projection grid too large");
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), C::LDS_BYTES, c->stream, d_v, d_phase,
                       d_idx, d_mean_g, d_q, g.T, g.N_tot, g.n_g, g.A_pad, g.K, n_mblk, (int)n_tblk);
    PSA_HIP_CHECK(hipGetLastError());
    return PSA_OK;
}

template <int MT, int WM, int WN>
static int launch_shape(psa_ctx* c, const float* d_v, const float* d_phase, const int* d_idx,
                        const float* d_mean_g, float2* d_q, const ProjGeom& g, bool disp) {
    const bool vec = (d_idx == nullptr) && (g.N_tot % 4 == 0) && (g.n_g == g.N_tot);
    if (vec)
        return disp ? launch_variant<MT, WM, WN, true, true>(c, d_v, d_phase, d_idx, d_mean_g, d_q, g)
                    : launch_variant<MT, WM, WN, true, false>(c, d_v, d_phase, d_idx, d_mean_g, d_q, g);
    return disp ? launch_variant<MT, WM, WN, false, true>(c, d_v, d_phase, d_idx, d_mean_g, d_q, g)
                : launch_variant<MT, WM, WN, false, false>(c, d_v, d_phase, d_idx, d_mean_g, d_q, g);
}

int launch_k1_mfma(psa_ctx* c, const float* d_v, const float* d_phase, const int* d_idx,
                   const float* d_mean_g, float2* d_q, const ProjGeom& g, bool displacements) {
    PSA_REQUIRE(g.A_pad % BA == 0 && g.A_pad >= BA, "A_pad must be a positive multiple of %d", BA);
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
