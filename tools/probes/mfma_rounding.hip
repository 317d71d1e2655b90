// Probe: how does v_mfma_f32_16x16x32_bf16 round its 32-term dot product + C ?
// build: hipcc --offload-arch=gfx950 -O2 mfma_rounding.hip -o mfma_rounding
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k(const float* a, const float* b, const float* c, float* d) {
    // lane l: A[row l&15][k = 8(l>>4)+j], B[k][col l&15]; C/D: col = l&15, row = 4(l>>4)+reg
    const int l = threadIdx.x, r = l & 15, q = l >> 4;
    bf16x8 av, bv;
    for (int j = 0; j < 8; ++j) {
        av[j] = (__bf16)a[r * 32 + 8 * q + j];
        bv[j] = (__bf16)b[(8 * q + j) * 16 + r];
    }
    f32x4 cv;
    for (int i = 0; i < 4; ++i) cv[i] = c[(4 * q + i) * 16 + r];
    f32x4 dv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, cv, 0, 0, 0);
    for (int i = 0; i < 4; ++i) d[(4 * q + i) * 16 + r] = dv[i];
}

static float bf(float x) { return (float)(__bf16)x; }

int main() {
    float *da, *db, *dc, *dd;
    hipMalloc(&da, 16 * 32 * 4); hipMalloc(&db, 32 * 16 * 4); hipMalloc(&dc, 256 * 4); hipMalloc(&dd, 256 * 4);
    std::vector<float> a(512), b(512), c(256), d(256);
    auto run = [&]() {
        hipMemcpy(da, a.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 2048, hipMemcpyHostToDevice);
        hipMemcpy(dc, c.data(), 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
        hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
    };
    // 1. tiny equal-sign products under a big C
    for (int sgn = -1; sgn <= 1; sgn += 2) {
        for (auto& x : a) x = ldexpf(1.f, -12);
        for (auto& x : b) x = sgn * ldexpf(1.f, -13);
        for (auto& x : c) x = 1.0f;
        run();
        printf("C=1, 32 products of %+g each: D-1 = %g   (exact %g)\n", sgn * ldexp(1., -25), (double)d[0] - 1.0, sgn * ldexp(1., -20));
    }
    // 2. one product just below half an ulp of C, others zero
    for (auto& x : a) x = 0; for (auto& x : b) x = 0; for (auto& x : c) x = 1.0f;
    a[0] = ldexpf(1.f, -12); b[0] = ldexpf(1.5f, -13);      // 1.5 * 2^-25 = 0.75 ulp(1)/... ulp(1)=2^-23
    run(); printf("C=1 + one product 1.5*2^-25: D-1 = %g (RNE of exact 4.47e-8 -> 0 or 2^-23=1.19e-7?)\n", (double)d[0] - 1.0);
    a[0] = ldexpf(1.f, -12); b[0] = ldexpf(1.f, -11);       // 2^-23 = exactly one ulp
    run(); printf("C=1 + one product 2^-23: D-1 = %g\n", (double)d[0] - 1.0);
    a[0] = ldexpf(1.f, -12); b[0] = ldexpf(1.5f, -12);      // 1.5 ulp -> RNE gives 2 ulp; truncation 1 ulp
    run(); printf("C=1 + one product 1.5*2^-23: D-1 = %g   (RNE 2.38e-7, RZ 1.19e-7)\n", (double)d[0] - 1.0);
    for (auto& x : c) x = -1.0f;
    run(); printf("C=-1 + one product 1.5*2^-23: D+1 = %g   (exact 1.79e-7: RNE 2.38e-7 (to -1+2ulp'), RZ/RM?)\n", (double)d[0] + 1.0);
    // 3. random data: bias of (D - exact)
    std::mt19937 rng(1); std::normal_distribution<float> nd(0.f, 1.f);
    double bias = 0, rel = 0, mag = 0; int cnt = 0;
    for (int rep = 0; rep < 200; ++rep) {
        for (auto& x : a) x = bf(nd(rng)); for (auto& x : b) x = bf(nd(rng));
        for (auto& x : c) x = 200.f * nd(rng);
        run();
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            double ex = c[i * 16 + j];
            for (int kk = 0; kk < 32; ++kk) ex += (double)a[i * 32 + kk] * (double)b[kk * 16 + j];
            double e = (double)d[i * 16 + j] - ex;
            bias += e * (ex > 0 ? 1 : -1); rel += fabs(e); mag += fabs(ex); ++cnt;
        }
    }
    printf("random: mean error toward +|x| = %.3e, mean |err| = %.3e, mean |x| = %.3e  (ulp at 200 ~ 1.5e-5)\n", bias / cnt, rel / cnt, mag / cnt);
    return 0;
}
