// Probe: error of a long v_mfma_f32_16x16x32_f16 accumulation chain whose running sum grows
// coherently (the `hi` accumulator of the 2 x f16 split kernel), and of the same chain folded
// into a float32 VALU sum every S instructions.
// build: hipcc --offload-arch=gfx950 -O2 mfma_f16_chain.hip -o mfma_f16_chain
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// A: [L][16 rows][32 k] f16, B: [L][32 k][16 cols] f16;  D[flush]: 16x16
__global__ void k(const _Float16* a, const _Float16* b, float* d, int L, int S) {
    const int l = threadIdx.x, r = l & 15, q = l >> 4;
    f32x4 master = {0, 0, 0, 0}, acc = {0, 0, 0, 0};
    for (int s = 0; s < L; ++s) {
        f16x8 av, bv;
        for (int j = 0; j < 8; ++j) {
            av[j] = a[((size_t)s * 16 + r) * 32 + 8 * q + j];
            bv[j] = b[((size_t)s * 32 + 8 * q + j) * 16 + r];
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc, 0, 0, 0);
        if (S > 0 && (s + 1) % S == 0) {
            master += acc;
            acc = f32x4{0, 0, 0, 0};
        }
    }
    master += acc;
    for (int i = 0; i < 4; ++i) d[(4 * q + i) * 16 + r] = master[i];
}

int main() {
    const int L = 1024;
    std::vector<_Float16> a((size_t)L * 512), b((size_t)L * 512);
    std::mt19937 rng(3);
    std::uniform_real_distribution<float> ph(0.f, 6.2831853f);
    std::normal_distribution<float> nd(0.f, 1.f);
    // rows: cos(phi_a + row offset) * 2^14; cols: (amp * cos(phi_a + col offset) + noise) scaled to 2^13
    for (int s = 0; s < L; ++s)
        for (int kk = 0; kk < 32; ++kk) {
            const float p = ph(rng);
            for (int r = 0; r < 16; ++r) a[((size_t)s * 16 + r) * 32 + kk] = (_Float16)(16384.f * cosf(p + 0.1f * r));
            for (int c = 0; c < 16; ++c)
                b[((size_t)s * 32 + kk) * 16 + c] = (_Float16)(2048.f * (1.0f * cosf(p + 0.1f * c) + (c & 1 ? 1.f : 0.05f) * nd(rng)));
        }
    std::vector<double> ex(256, 0.0);
    for (int s = 0; s < L; ++s)
        for (int r = 0; r < 16; ++r)
            for (int c = 0; c < 16; ++c) {
                double t = 0;
                for (int kk = 0; kk < 32; ++kk)
                    t += (double)(float)a[((size_t)s * 16 + r) * 32 + kk] * (double)(float)b[((size_t)s * 32 + kk) * 16 + c];
                ex[r * 16 + c] += t;
            }
    _Float16 *da, *db;
    float* dd;
    hipMalloc(&da, a.size() * 2); hipMalloc(&db, b.size() * 2); hipMalloc(&dd, 1024);
    hipMemcpy(da, a.data(), a.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), b.size() * 2, hipMemcpyHostToDevice);
    double mx = 0;
    for (double v : ex) mx = fmax(mx, fabs(v));
    for (int S : {0, 1, 2, 4, 8, 16, 64}) {
        std::vector<float> d(256);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd, L, S);
        hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
        double worst = 0, bias = 0;
        for (int i = 0; i < 256; ++i) {
            const double e = ((double)d[i] - ex[i]) / mx;
            worst = fmax(worst, fabs(e));
            bias += e * (ex[i] > 0 ? 1 : -1);
        }
        printf("flush every %3d MFMAs: max |err|/max|sum| = %.3e   mean signed (toward larger |sum|) = %+.3e\n", S, worst, bias / 256);
    }
    return 0;
}
