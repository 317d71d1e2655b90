// K1, shuffle form -- the projection as BASELINE's north_star words it: stage a
// velocity tile in LDS, let each lane run the complex multiply-accumulate over its share
// of the atoms for a handful of (k, component) outputs, and finish with wavefront
// __shfl_down sums.  It is the diagnostic twin of the MFMA tile kernel (k1_mfma.hip):
// same inputs, same output layout, completely different arithmetic schedule, so the two
// check each other on the GPU.  VALU-bound by construction -- selected only with
// psa_set_k1(ctx, PSA_K1_WAVE).
//
//   workgroup = one frame t  x  16 k-points (4 wavefronts x 4 k-points)
//   atom loop = chunks of 256 atoms staged as d[t, a, 0..2] (768 floats) in LDS
//   lane      = atoms lane, lane+64, lane+128, lane+192 of the chunk
#include "psa_ctx.h"

namespace psa {

constexpr int WK_K_PER_WAVE = 4;
constexpr int WK_K_PER_BLOCK = 16;
constexpr int WK_ATOMS = 256;

template <bool DISP>
__global__ void __launch_bounds__(256)
k1_wave_kernel(const float* __restrict__ V, const float* __restrict__ P,
               const int* __restrict__ idx, const float* __restrict__ mean_g,
               float2* __restrict__ Q, int64_t T, int64_t q_stride, int64_t N_tot, int n_g, int A_pad, int K, int m_blk) {
    __shared__ float vs[WK_ATOMS * 3];
    const int64_t t = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k0 = blockIdx.y * WK_K_PER_BLOCK + wave * WK_K_PER_WAVE;

    float re[WK_K_PER_WAVE][3], im[WK_K_PER_WAVE][3];
#pragma unroll
    for (int j = 0; j < WK_K_PER_WAVE; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) re[j][c] = im[j][c] = 0.f;

    for (int a0 = 0; a0 < n_g; a0 += WK_ATOMS) {
        // stage: thread tid owns atom a0+tid of the chunk
        {
            const int a = a0 + tid;
            float x = 0.f, y = 0.f, z = 0.f;
            if (a < n_g) {
                const int64_t src = idx ? idx[a] : a;
                const float*  p = V + (t * N_tot + src) * 3;
                x = p[0]; y = p[1]; z = p[2];
                if constexpr (DISP) {
                    x -= mean_g[3 * a + 0]; y -= mean_g[3 * a + 1]; z -= mean_g[3 * a + 2];
                }
            }
            vs[3 * tid + 0] = x; vs[3 * tid + 1] = y; vs[3 * tid + 2] = z;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < WK_ATOMS / 64; ++u) {
            const int al = lane + 64 * u;
            const int a = a0 + al;
            if (a < n_g) {
                const float x = vs[3 * al], y = vs[3 * al + 1], z = vs[3 * al + 2];
#pragma unroll
                for (int j = 0; j < WK_K_PER_WAVE; ++j) {
                    const int k = k0 + j;
                    if (k < K) {
                        const float pc = P[p_tile_index(2 * k, a, m_blk, A_pad / K1_BA)];
                        const float ps = P[p_tile_index(2 * k + 1, a, m_blk, A_pad / K1_BA)];
                        re[j][0] = fmaf(pc, x, re[j][0]); im[j][0] = fmaf(ps, x, im[j][0]);
                        re[j][1] = fmaf(pc, y, re[j][1]); im[j][1] = fmaf(ps, y, im[j][1]);
                        re[j][2] = fmaf(pc, z, re[j][2]); im[j][2] = fmaf(ps, z, im[j][2]);
                    }
                }
            }
        }
        __syncthreads();
    }

    // wavefront (64-lane) shuffle sums
#pragma unroll
    for (int j = 0; j < WK_K_PER_WAVE; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float r = re[j][c], i = im[j][c];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                r += __shfl_down(r, off, 64);
                i += __shfl_down(i, off, 64);
            }
            const int k = k0 + j;
            if (lane == 0 && k < K) Q[((int64_t)k * 3 + c) * q_stride + t] = make_float2(r, i);
        }
}

int launch_k1_wave(psa_ctx* c, const float* d_v, const float* d_phase, const int* d_idx,
                   const float* d_mean_g, float2* d_q, const ProjGeom& g, bool displacements) {
    const int64_t ky = (g.K + WK_K_PER_BLOCK - 1) / WK_K_PER_BLOCK;
    PSA_REQUIRE(g.T < (1ll << 31) && ky <= 65535, "shape too large for the shuffle kernel");
    dim3 grid((unsigned)g.T, (unsigned)ky);
    if (displacements)
        hipLaunchKernelGGL(k1_wave_kernel<true>, grid, dim3(256), 0, c->stream, d_v, d_phase, d_idx,
                           d_mean_g, d_q, g.T, g.q_stride, g.N_tot, g.n_g, g.A_pad, g.K, g.m_blk);
    else
        hipLaunchKernelGGL(k1_wave_kernel<false>, grid, dim3(256), 0, c->stream, d_v, d_phase, d_idx,
                           d_mean_g, d_q, g.T, g.q_stride, g.N_tot, g.n_g, g.A_pad, g.K, g.m_blk);
    PSA_HIP_CHECK(hipGetLastError());
    return PSA_OK;
}

}  // namespace psa
