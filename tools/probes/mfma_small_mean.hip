// Probe: does a bf16 MFMA preserve the MEAN of small addends accumulated onto a large C?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* a, const float* b, const float* c, float* d) {
    const int l = threadIdx.x, r = l & 15, q = l >> 4;
    bf16x8 av, bv;
    for (int j = 0; j < 8; ++j) { av[j] = (__bf16)a[r * 32 + 8 * q + j]; bv[j] = (__bf16)b[(8 * q + j) * 16 + r]; }
    f32x4 cv;
    for (int i = 0; i < 4; ++i) cv[i] = c[(4 * q + i) * 16 + r];
    f32x4 dv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, cv, 0, 0, 0);
    for (int i = 0; i < 4; ++i) d[(4 * q + i) * 16 + r] = dv[i];
}
static float bf(float x) { return (float)(__bf16)x; }
int main() {
    float *da, *db, *dc, *dd;
    (void)hipMalloc(&da, 2048); (void)hipMalloc(&db, 2048); (void)hipMalloc(&dc, 1024); (void)hipMalloc(&dd, 1024);
    std::vector<float> a(512), b(512), c(256), d(256);
    std::mt19937 rng(7); std::normal_distribution<float> nd(0.f, 1.f);
    for (float Cval : {30000.f, -30000.f, 100.f}) {
        for (float scale : {1.f / 512, 1.f / 64}) {
            double sum_true = 0, sum_got = 0; long cnt = 0;
            for (int rep = 0; rep < 400; ++rep) {
                for (auto& x : a) x = bf(scale * nd(rng)); for (auto& x : b) x = bf(3.f * nd(rng));
                for (auto& x : c) x = Cval;
                (void)hipMemcpy(da, a.data(), 2048, hipMemcpyHostToDevice); (void)hipMemcpy(db, b.data(), 2048, hipMemcpyHostToDevice);
                (void)hipMemcpy(dc, c.data(), 1024, hipMemcpyHostToDevice);
                hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
                (void)hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
                for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
                    double p = 0; for (int kk = 0; kk < 32; ++kk) p += (double)a[i * 32 + kk] * (double)b[kk * 16 + j];
                    sum_true += p; sum_got += (double)d[i * 16 + j] - (double)Cval; ++cnt;
                }
            }
            printf("C=%8.0f ulp=%.4g  addend sigma~%.3g : mean true %+.4e  mean got %+.4e  (diff %+.3e = %+.3f ulp)\n", Cval,
                   ldexp(1., ilogb(fabs(Cval)) - 23), scale * 3 * sqrt(32.), sum_true / cnt, sum_got / cnt, (sum_got - sum_true) / cnt,
                   (sum_got - sum_true) / cnt / ldexp(1., ilogb(fabs(Cval)) - 23));
        }
    }
    // sign test
    for (auto& x : a) x = 0; for (auto& x : b) x = 0;
    a[0] = ldexpf(1.f, -12); b[0] = -ldexpf(1.5f, -12); for (auto& x : c) x = 1.f;
    (void)hipMemcpy(da, a.data(), 2048, hipMemcpyHostToDevice); (void)hipMemcpy(db, b.data(), 2048, hipMemcpyHostToDevice); (void)hipMemcpy(dc, c.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dc, dd); (void)hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
    printf("C=1 + product -1.5*2^-23: D-1 = %g ulp   (toward zero of addend: -1, floor: -2, RNE: -2 (tie->even))\n", ((double)d[0] - 1.0) / ldexp(1., -23));
    b[0] = -ldexpf(1.25f, -12);
    (void)hipMemcpy(db, b.data(), 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dc, dd); (void)hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
    printf("C=1 + product -1.25*2^-23: D-1 = %g ulp   (toward zero: -1, floor: -2, RNE: -1)\n", ((double)d[0] - 1.0) / ldexp(1., -23));
    return 0;
}
