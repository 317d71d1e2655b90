// K2 -- epilogues after the batched FFT.  All HBM-streaming: every kernel reads and
// writes each element once, coalesced on both sides (LDS tile where a transpose is
// needed).
//
//   scale_transpose_c64   S[k,c,w]/T -> out[w,k,c]      ref: sed_calculator.py:83-84, 311
//   intensity_accumulate  I[k,w] (+)= sum_c |S[k,c,w]/T|^2          ref: :325
//   transpose_f32         I[k,w] -> out[w,k]                         ref: :327
//   result_intensity      sum_c |out[w,k,c]|^2                       ref: core/sed.py:22-24
//   result_chiral_c       folded phase difference of two components  ref: :344-350
//   dft_bin               one frequency bin of FFT_t(q)/T                 ref: :83-84 as used by :494-499
#include "psa_ctx.h"

namespace psa {

constexpr int TT = 64;   // frames per tile
constexpr int KT = 16;   // k-points per tile

// Column maps (psa_ctx.h: KMAP_MIRROR).  Entry i of a launch writes output column cols[i] (null:
// col_first + i) from slab row srcs[i] & ~KMAP_MIRROR (null: src_first + i).  With KMAP_MIRROR set the
// column is the partner -k of the k-vector that row was projected for: S(-k)[w] = conj S(k)[(T-w) mod T]
// (the float32 phase argument is odd in k, cos even, sin odd, the data real: q(-k) = conj q(k) bit for
// bit), so the row is read backwards in frequency and conjugated instead of being projected and
// transformed a second time.
__device__ __forceinline__ int64_t mirrored_frame(int64_t t, int64_t T) { return t == 0 ? 0 : T - t; }

// S[k,c,w]/T -> out[w,col,c]; inten (may be null): sum_c |out[w,col,c]|^2 -> inten[w,col], taken from
// the tile while it is in LDS (core/sed.py:22-24 without a second pass over the result)
__global__ void __launch_bounds__(256)
scale_transpose_c64_kernel(const float2* __restrict__ slab, float2* __restrict__ out, float* __restrict__ inten, int64_t T,
                           int64_t n, int64_t K_pitch, int64_t col_first, int64_t src_first, const int32_t* __restrict__ cols,
                           const int32_t* __restrict__ srcs) {
    __shared__ float2 tile[KT * 3][TT + 1];
    __shared__ int    col_s[KT], src_s[KT];
    const int64_t t0 = (int64_t)blockIdx.x * TT;
    const int64_t i0 = (int64_t)blockIdx.y * KT;
    const int     tid = threadIdx.x;
    const float   n_t = (float)T;
    if (tid < KT) {
        const int64_t i = i0 + tid;
        col_s[tid] = i < n ? (cols ? cols[i] : (int)(col_first + i)) : -1;
        src_s[tid] = i < n ? (srcs ? srcs[i] : (int)(src_first + i)) : 0;
    }
    __syncthreads();
    {
        const int     tl = tid & 63;
        const int64_t t = t0 + tl;
#pragma unroll
        for (int j = 0; j < KT * 3 / 4; ++j) {
            const int r = (tid >> 6) + 4 * j;      // row = entry*3 + c
            const int e = r / 3;
            float2    v = make_float2(0.f, 0.f);
            if (col_s[e] >= 0 && t < T) {
                const int     src = src_s[e];
                const bool    mirror = (src & KMAP_MIRROR) != 0;
                const int64_t row = src & ~KMAP_MIRROR;
                v = slab[(row * 3 + r % 3) * T + (mirror ? mirrored_frame(t, T) : t)];
                // complex64 / int in NumPy is a true division of both parts (:83)
                v.x = __fdiv_rn(v.x, n_t);
                v.y = __fdiv_rn(v.y, n_t);
                if (mirror) v.y = -v.y;
            }
            tile[r][tl] = v;
        }
    }
    __syncthreads();
    for (int item = tid; item < TT * KT * 3; item += 256) {
        const int tl = item / (KT * 3), e = item - tl * (KT * 3);
        const int col = col_s[e / 3];
        if (col >= 0 && t0 + tl < T) out[((t0 + tl) * K_pitch + col) * 3 + e % 3] = tile[e][tl];
    }
    if (inten) {
        for (int item = tid; item < TT * KT; item += 256) {
            const int tl = item / KT, e = item - tl * KT;
            const int col = col_s[e];
            if (col >= 0 && t0 + tl < T) {
                float s = 0.f;
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) {
                    const float2 v = tile[e * 3 + cc][tl];
                    s += v.x * v.x + v.y * v.y;
                }
                inten[(t0 + tl) * K_pitch + col] = s;
            }
        }
    }
}

int launch_scale_transpose_c64(psa_ctx* c, const float2* d_slab, float2* d_out, float* d_inten, int64_t T, int64_t n,
                               int64_t K_pitch, int64_t col_first, int64_t src_first, const int32_t* d_cols,
                               const int32_t* d_srcs) {
    if (n == 0) return PSA_OK;
    const int64_t gy = (n + KT - 1) / KT;
    PSA_REQUIRE(gy <= 65535, "too many k-points for one transpose launch");
    dim3 grid((unsigned)((T + TT - 1) / TT), (unsigned)gy);
    hipLaunchKernelGGL(scale_transpose_c64_kernel, grid, dim3(256), 0, c->stream, d_slab, d_out, d_inten, T, n, K_pitch,
                       col_first, src_first, d_cols, d_srcs);
    PSA_HIP_CHECK(hipGetLastError());
    return PSA_OK;
}

__global__ void __launch_bounds__(256)
intensity_accumulate_kernel(const float2* __restrict__ q, float* __restrict__ acc, int64_t T,
                            int64_t K, int first) {
    const int64_t total = T * K;
    const float   n_t = (float)T;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * 256) {
        const int64_t k = i / T, t = i - k * T;
        float s = 0.f;
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
            const float2 v = q[(k * 3 + cc) * T + t];
            const float  re = __fdiv_rn(v.x, n_t), im = __fdiv_rn(v.y, n_t);
            s += re * re + im * im;
        }
        acc[i] = first ? s : acc[i] + s;
    }
}

int launch_intensity_accumulate(psa_ctx* c, const float2* d_q, float* d_slab_rows, int64_t T,
                                int64_t K_local, bool first_group) {
    int64_t blocks = (T * K_local + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(intensity_accumulate_kernel, dim3((unsigned)blocks), dim3(256), 0, c->stream,
                       d_q, d_slab_rows, T, K_local, first_group ? 1 : 0);
    PSA_HIP_CHECK(hipGetLastError());
    return PSA_OK;
}

// I[row,w] -> out[w,k]; srcs (may be null: row = k) as above -- a mirrored column reads its partner's
// row backwards in frequency: I(-k)[w] = I(k)[(T-w) mod T]
__global__ void __launch_bounds__(256)
transpose_f32_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t T, int64_t K,
                     const int32_t* __restrict__ srcs) {
    __shared__ float tile[32][33];
    const int64_t t0 = (int64_t)blockIdx.x * 32, k0 = (int64_t)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t k = k0 + ty + 8 * j, t = t0 + tx;
        float v = 0.f;
        if (k < K && t < T) {
            const int src = srcs ? srcs[k] : (int)k;
            v = in[(int64_t)(src & ~KMAP_MIRROR) * T + ((src & KMAP_MIRROR) ? mirrored_frame(t, T) : t)];
        }
        tile[ty + 8 * j][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t t = t0 + ty + 8 * j, k = k0 + tx;
        if (k < K && t < T) out[t * K + k] = tile[tx][ty + 8 * j];
    }
}

int launch_transpose_f32(psa_ctx* c, const float* d_slab, float* d_out, int64_t T, int64_t K, const int32_t* d_srcs) {
    if (K == 0) return PSA_OK;
    const int64_t gy = (K + 31) / 32;
    PSA_REQUIRE(gy <= 65535, "too many k-points for one transpose launch");
    dim3 grid((unsigned)((T + 31) / 32), (unsigned)gy);
    hipLaunchKernelGGL(transpose_f32_kernel, grid, dim3(256), 0, c->stream, d_slab, d_out, T, K, d_srcs);
    PSA_HIP_CHECK(hipGetLastError());
    return PSA_OK;
}

__global__ void __launch_bounds__(256)
result_intensity_kernel(const float2* __restrict__ out, float* __restrict__ inten, int64_t n_tk) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_tk;
         i += (int64_t)gridDim.x * 256) {
        float s = 0.f;
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
            const float2 v = out[3 * i + cc];
            s += v.x * v.x + v.y * v.y;
        }
        inten[i] = s;
    }
}

int launch_result_intensity(psa_ctx* c, const float2* d_out, float* d_int, int64_t n_tk) {
    int64_t blocks = (n_tk + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(result_intensity_kernel, dim3((unsigned)blocks), dim3(256), 0, c->stream, d_out,
                       d_int, n_tk);
    PSA_HIP_CHECK(hipGetLastError());
    return PSA_OK;
}

// Option "C" of calculate_chiral_phase (:344-350), all in float32 like the reference:
//   d = angle(Z1) - angle(Z2);  d = (d + pi) % (2 pi) - pi;  fold |d| > pi/2 back.
__global__ void __launch_bounds__(256)
result_chiral_c_kernel(const float2* __restrict__ out, float* __restrict__ phase, int64_t n_tk, int c1,
                       int c2) {
    const float PI = 3.14159265358979323846f, TWO_PI = 6.28318530717958647692f;
    const float HALF_PI = 1.57079632679489661923f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_tk;
         i += (int64_t)gridDim.x * 256) {
        const float2 z1 = out[3 * i + c1], z2 = out[3 * i + c2];
        float d = atan2f(z1.y, z1.x) - atan2f(z2.y, z2.x);
        d = d + PI;
        float m = fmodf(d, TWO_PI);             // Python-style %: result takes the divisor's sign
        if (m != 0.f && m < 0.f) m += TWO_PI;
        d = m - PI;
        if (d > HALF_PI) d = PI - d;
        else if (d < -HALF_PI) d = -PI - d;
        phase[i] = d;
    }
}

int launch_result_chiral_c(psa_ctx* c, const float2* d_out, float* d_phase, int64_t n_tk, int c1, int c2) {
    int64_t blocks = (n_tk + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(result_chiral_c_kernel, dim3((unsigned)blocks), dim3(256), 0, c->stream, d_out,
                       d_phase, n_tk, c1, c2);
    PSA_HIP_CHECK(hipGetLastError());
    return PSA_OK;
}

// One frequency bin of the time FFT of a single k-vector's projection q (3, T):
//   S_c = (1/T) sum_t q[c,t] exp(-2 pi i bin t / T)
// -- all that iSED consumes of a group's spectrum (sed_calculator.py:483, :494-499).  The twiddle
// angle is reduced exactly in integers (bin * t mod T) and evaluated in double, the sum is kept
// in double: the result is the correctly rounded value the float32 FFT approximates.
__global__ void __launch_bounds__(1024)
dft_bin_kernel(const float2* __restrict__ q, int64_t T, int64_t bin, float2* __restrict__ out3) {
    __shared__ double red[16][6];
    double acc[6] = {0, 0, 0, 0, 0, 0};
    const double w = -6.283185307179586476925286766559 / (double)T;
    for (int64_t t = threadIdx.x; t < T; t += 1024) {
        const int64_t m = (int64_t)(((unsigned __int128)(uint64_t)bin * (uint64_t)t) % (uint64_t)T);
        double sn, cs;
        sincos(w * (double)m, &sn, &cs);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float2 v = q[c * T + t];
            acc[2 * c] += (double)v.x * cs - (double)v.y * sn;
            acc[2 * c + 1] += (double)v.x * sn + (double)v.y * cs;
        }
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[j] += __shfl_xor(acc[j], o);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0)
        for (int j = 0; j < 6; ++j) red[wave][j] = acc[j];
    __syncthreads();
    if (threadIdx.x < 3) {
        double re = 0, im = 0;
        for (int k = 0; k < 16; ++k) {
            re += red[k][2 * threadIdx.x];
            im += red[k][2 * threadIdx.x + 1];
        }
        out3[threadIdx.x] = make_float2((float)(re / (double)T), (float)(im / (double)T));
    }
}

int launch_dft_bin(psa_ctx* c, const float2* d_q, int64_t T, int64_t bin, float2* d_out3) {
    hipLaunchKernelGGL(dft_bin_kernel, dim3(1), dim3(1024), 0, c->stream, d_q, T, bin, d_out3);
    PSA_HIP_CHECK(hipGetLastError());
    return PSA_OK;
}

}  // namespace psa
