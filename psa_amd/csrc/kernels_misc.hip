// Small streaming kernels around the projection: phase table, mean gather,
// synthetic-trajectory fill, mean over frames.  All are HBM/latency-bound
// elementwise work; one thread per output element, coalesced along the fastest axis.
#include "psa_ctx.h"

namespace psa {

// ---------------------------------------------------------------------------
// Phase table                                   ref: sed_calculator.py:78
//   P[k,a] = exp(1j * np.dot(k_vectors, mean_pos_group.T))
// np.dot on float32 is sgemm; with an inner dimension of 3 OpenBLAS evaluates each
// entry as the FMA chain fma(kz,rz, fma(ky,ry, kx*rx)) (checked bit-for-bit on the
// build host, tests/test_oracle_golden.py::test_phase_argument_is_fma_chain), so the
// argument below is the same float32 number the reference feeds to exp.  Only the
// sin/cos evaluation differs (<= 2 ulp, ocml vs libm).
//
// Layout written: the LDS tile images of the projection kernel (p_tile_index in psa_ctx.h):
// row m = 2k holds cos, m = 2k+1 holds sin; everything outside (K, n_g) is zero so the
// tile kernel needs no bounds checks on P' (the 4 pad floats per row are never read).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
phase_table_kernel(const float* __restrict__ kvec, const float* __restrict__ mean_all,
                   const int* __restrict__ idx, float* __restrict__ P,
                   int K, int n_g, int A_pad, int M_pad, int m_blk) {
    const int a = blockIdx.y * 256 + threadIdx.x;      // grid.x runs over k (can be > 65535)
    const int k = blockIdx.x;
    if (a >= A_pad || 2 * k >= M_pad) return;
    float c = 0.f, s = 0.f;
    if (k < K && a < n_g) {
        const int   src = idx ? idx[a] : a;
        const float rx = mean_all[3 * (size_t)src + 0];
        const float ry = mean_all[3 * (size_t)src + 1];
        const float rz = mean_all[3 * (size_t)src + 2];
        const float kx = kvec[3 * k + 0], ky = kvec[3 * k + 1], kz = kvec[3 * k + 2];
        const float arg = __fmaf_rn(kz, rz, __fmaf_rn(ky, ry, __fmul_rn(kx, rx)));
        sincosf(arg, &s, &c);
    }
    const int n_stage = A_pad / K1_BA;
    P[p_tile_index(2 * k, a, m_blk, n_stage)]     = c;
    P[p_tile_index(2 * k + 1, a, m_blk, n_stage)] = s;
}

int launch_phase_table(psa_ctx* c, const float* d_kvec, const float* d_mean_all, const int* d_idx,
                       float* d_phase, const ProjGeom& g) {
    dim3 grid(g.M_pad / 2, (g.A_pad + 255) / 256);
    hipLaunchKernelGGL(phase_table_kernel, grid, dim3(256), 0, c->stream, d_kvec, d_mean_all, d_idx,
                       d_phase, g.K, g.n_g, g.A_pad, g.M_pad, g.m_blk);
    PSA_HIP_CHECK(hipGetLastError());
    return PSA_OK;
}

// mean positions of the group's atoms, in group order, padded with zeros to A_pad:
// the tile loader subtracts them from the staged positions (sed_calculator.py:70).
__global__ void __launch_bounds__(256)
gather_mean_kernel(const float* __restrict__ mean_all, const int* __restrict__ idx,
                   float* __restrict__ mean_g, int n_g, int A_pad) {
    const int j = blockIdx.x * 256 + threadIdx.x;      // float index into (A_pad,3)
    if (j >= 3 * A_pad) return;
    const int a = j / 3, comp = j - 3 * a;
    float v = 0.f;
    if (a < n_g) v = mean_all[3 * (size_t)(idx ? idx[a] : a) + comp];
    mean_g[j] = v;
}

int launch_gather_mean(psa_ctx* c, const float* d_mean_all, const int* d_idx, float* d_mean_g,
                       const ProjGeom& g) {
    hipLaunchKernelGGL(gather_mean_kernel, dim3((3 * g.A_pad + 255) / 256), dim3(256), 0, c->stream,
                       d_mean_all, d_idx, d_mean_g, g.n_g, g.A_pad);
    PSA_HIP_CHECK(hipGetLastError());
    return PSA_OK;
}

// ---------------------------------------------------------------------------
// Synthetic trajectory (bench / parity inputs).  NumPy twin: psa_amd/synth.py.
// Integer hash -> sum of four 16-bit uniforms (Irwin-Hall, ~N(0,1)) -> one float32
// multiply; plane-wave modes from host-built cos/sin tables with explicitly
// unfused multiply/add, so both sides produce the same bits.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ void __launch_bounds__(256)
fill_synthetic_kernel(float* __restrict__ v, int64_t T, int64_t N, uint64_t seed, int64_t t_offset, int n_modes,
                      const float* __restrict__ amp, const int* __restrict__ comp,
                      const float* __restrict__ ct, const float* __restrict__ st,
                      const float* __restrict__ ca, const float* __restrict__ sa) {
    const int64_t total = T * N * 3;
    const uint64_t key = seed * 0xD1B54A32D192ED03ull;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * 256) {
        const int64_t ta = i / 3;
        const int     cc = (int)(i - 3 * ta);
        const int64_t t = ta / N, a = ta - t * N;
        // the counter is the element's index in the WHOLE trajectory: a rank that holds frames
        // [t_offset, t_offset + T) generates exactly its slice of the one synthetic array
        const uint64_t h = splitmix64((uint64_t)(i + t_offset * N * 3) ^ key);
        const int s = (int)(h & 0xFFFF) + (int)((h >> 16) & 0xFFFF) + (int)((h >> 32) & 0xFFFF) +
                      (int)(h >> 48);
        float val = __fmul_rn((float)(s - 131070), 1.0f / 37837.0f);
        for (int m = 0; m < n_modes; ++m) {
            if (comp[m] == cc) {
                const float w = __fadd_rn(__fmul_rn(ct[m * T + t], ca[m * N + a]),
                                          __fmul_rn(st[m * T + t], sa[m * N + a]));
                val = __fadd_rn(val, __fmul_rn(amp[m], w));
            }
        }
        v[i] = val;
    }
}

int launch_fill_synthetic(psa_ctx* c, float* d_v, int64_t T, int64_t N, uint64_t seed, int64_t t_offset, int n_modes,
                          const float* d_amp, const int* d_comp, const float* d_ct,
                          const float* d_st, const float* d_ca, const float* d_sa) {
    const int64_t total = T * N * 3;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(fill_synthetic_kernel, dim3((unsigned)blocks), dim3(256), 0, c->stream, d_v, T,
                       N, seed, t_offset, n_modes, d_amp, d_comp, d_ct, d_st, d_ca, d_sa);
    PSA_HIP_CHECK(hipGetLastError());
    return PSA_OK;
}

// ---------------------------------------------------------------------------
// mean over frames                               ref: sed_calculator.py:205
// np.mean(positions, axis=0, dtype=float32) adds frame after frame into one float32
// accumulator per (atom, component) and divides by T once.  One thread per column,
// frames in order, adds kept unfused and in sequence; loads are issued 8 frames
// ahead so the dependent add chain does not serialise HBM.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
mean_over_frames_kernel(const float* __restrict__ x, int64_t T, int64_t cols, float* __restrict__ mean) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= cols) return;
    float acc = 0.f;
    int64_t t = 0;
    for (; t + 8 <= T; t += 8) {
        float r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) r[u] = x[(t + u) * cols + j];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __fadd_rn(acc, r[u]);
    }
    for (; t < T; ++t) acc = __fadd_rn(acc, x[t * cols + j]);
    mean[j] = __fdiv_rn(acc, (float)T);
}

int launch_mean_over_frames(psa_ctx* c, const float* d_x, int64_t T, int64_t N, float* d_mean) {
    const int64_t cols = 3 * N;
    hipLaunchKernelGGL(mean_over_frames_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0,
                       c->stream, d_x, T, cols, d_mean);
    PSA_HIP_CHECK(hipGetLastError());
    return PSA_OK;
}

// ---------------------------------------------------------------------------
// displacements                                   ref: sed_calculator.py:70-72
// positions - mean_pos, one float32 subtraction per element, materialised like the reference's
// temporary (so the fast projection kernels read it like velocities)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
subtract_mean_kernel(const float* __restrict__ x, const float* __restrict__ mean, float* __restrict__ out, int64_t T,
                     int64_t cols) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= cols) return;
    const float m = mean[j];
    for (int64_t t = blockIdx.y; t < T; t += gridDim.y) out[t * cols + j] = __fsub_rn(x[t * cols + j], m);
}

int launch_subtract_mean(psa_ctx* c, const float* d_x, const float* d_mean, float* d_out, int64_t T, int64_t N) {
    const int64_t cols = 3 * N;
    dim3 grid((unsigned)((cols + 255) / 256), (unsigned)(T < 256 ? T : 256));
    hipLaunchKernelGGL(subtract_mean_kernel, grid, dim3(256), 0, c->stream, d_x, d_mean, d_out, T, cols);
    PSA_HIP_CHECK(hipGetLastError());
    return PSA_OK;
}

// ---------------------------------------------------------------------------
// largest |x| of a resident array, as float bits (non-negative floats order like unsigned
// integers; a NaN or Inf anywhere makes the result >= 0x7f800000).  The split-precision
// projection kernel derives from it the power of two that places the data at the top of the
// float16 range.  One pass at HBM rate, run once per upload.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
absmax_bits_kernel(const float4* __restrict__ x4, int64_t n4, const float* __restrict__ tail, int n_tail,
                   unsigned* __restrict__ out) {
    unsigned m = 0;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const float4 v = x4[i];
        m = max(max(m, __float_as_uint(v.x) & 0x7fffffffu), __float_as_uint(v.y) & 0x7fffffffu);
        m = max(max(m, __float_as_uint(v.z) & 0x7fffffffu), __float_as_uint(v.w) & 0x7fffffffu);
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < n_tail) m = max(m, __float_as_uint(tail[threadIdx.x]) & 0x7fffffffu);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// The same per column block of 32 atoms (over all frames): an index-list group takes its scale from
// the blocks its atoms sit in, so that a group of slow atoms is not scaled by a fast atom elsewhere.
__global__ void __launch_bounds__(256)
absmax_blocks_kernel(const float* __restrict__ x, const float* __restrict__ mean, int64_t T, int64_t N,
                     unsigned* __restrict__ out) {
    // mean (N,3), may be null: the magnitudes are those of x - mean (displacement mode, sed_calculator.py:70)
    const int     fr = threadIdx.x >> 7, e = threadIdx.x & 127;      // two frame rows per pass, 96 floats each
    const int64_t col = (int64_t)blockIdx.x * 96 + e;
    unsigned      m = 0;
    if (e < 96 && col < 3 * N) {
        const float mu = mean ? mean[col] : 0.f;
        for (int64_t t = 2 * (int64_t)blockIdx.y + fr; t < T; t += 2 * (int64_t)gridDim.y)
            m = max(m, __float_as_uint(__fsub_rn(x[t * 3 * N + col], mu)) & 0x7fffffffu);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out + blockIdx.x, m);
}

int launch_absmax_blocks(psa_ctx* c, const float* d_x, const float* d_mean, int64_t T, int64_t N, unsigned* d_out) {
    const int64_t n_blocks = (N + 31) / 32;
    PSA_REQUIRE(n_blocks < (1ll << 31), "too many atoms");
    PSA_HIP_CHECK(hipMemsetAsync(d_out, 0, (size_t)n_blocks * sizeof(unsigned), c->stream));
    const int64_t rows = (T + 1) / 2;
    dim3 grid((unsigned)n_blocks, (unsigned)(rows < 64 ? rows : 64));
    hipLaunchKernelGGL(absmax_blocks_kernel, grid, dim3(256), 0, c->stream, d_x, d_mean, T, N, d_out);
    PSA_HIP_CHECK(hipGetLastError());
    return PSA_OK;
}

// reset = false: fold another piece of the array into a running maximum (chunks of an upload)
int launch_absmax_bits(psa_ctx* c, const float* d_x, int64_t n, unsigned* d_out, bool reset) {
    if (reset) PSA_HIP_CHECK(hipMemsetAsync(d_out, 0, sizeof(unsigned), c->stream));
    const int64_t n4 = n / 4;
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(absmax_bits_kernel, dim3((unsigned)blocks), dim3(256), 0, c->stream,
                       reinterpret_cast<const float4*>(d_x), n4, d_x + 4 * n4, (int)(n - 4 * n4), d_out);
    PSA_HIP_CHECK(hipGetLastError());
    return PSA_OK;
}

}  // namespace psa
