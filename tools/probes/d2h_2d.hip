// Probe: device -> pinned-host bandwidth of hipMemcpy2DAsync for the row widths a k-block of the
// (T, K, 3) complex64 result has (24 bytes per k-point), against one linear copy.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
int main() {
    const size_t T = 65536, K = 256, row = K * 24, total = T * row;
    char *d, *h;
    (void)hipMalloc(&d, total);
    (void)hipHostMalloc(&h, total, hipHostMallocDefault);
    (void)hipMemset(d, 1, total);
    hipStream_t s;
    (void)hipStreamCreate(&s);
    auto time = [&](auto fn) {
        fn(); (void)hipStreamSynchronize(s);
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 3; ++i) fn();
        (void)hipStreamSynchronize(s);
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / 3;
    };
    double ms = time([&] { (void)hipMemcpyAsync(h, d, total, hipMemcpyDeviceToHost, s); });
    printf("linear %zu MB: %.2f ms = %.1f GB/s\n", total >> 20, ms, total / ms / 1e6);
    for (size_t kb : {32, 64, 128}) {
        const size_t w = kb * 24;
        ms = time([&] {
            for (size_t k0 = 0; k0 < K; k0 += kb)
                (void)hipMemcpy2DAsync(h + k0 * 24, row, d + k0 * 24, row, w, T, hipMemcpyDeviceToHost, s);
        });
        printf("2-D, %zu blocks of width %zu B x %zu rows: %.2f ms = %.1f GB/s\n", K / kb, w, T, ms, total / ms / 1e6);
    }
    return 0;
}
