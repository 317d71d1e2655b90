// Probe: device -> pinned-host bandwidth with ONE copy stream against TWO concurrent ones (two halves of
// the buffer), and a kernel that writes straight into mapped pinned memory: is 55 GB/s the link or the engine?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void store_to_host(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
int main() {
    const size_t total = (size_t)470 << 20;
    char *d, *h;
    (void)hipMalloc(&d, total);
    (void)hipHostMalloc(&h, total, hipHostMallocDefault);
    (void)hipMemset(d, 1, total);
    hipStream_t s[2];
    (void)hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking);
    (void)hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking);
    auto time = [&](auto fn) {
        fn(); (void)hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 5; ++i) fn();
        (void)hipDeviceSynchronize();
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / 5;
    };
    double ms = time([&] { (void)hipMemcpyAsync(h, d, total, hipMemcpyDeviceToHost, s[0]); });
    printf("one stream, %zu MB: %.2f ms = %.1f GB/s\n", total >> 20, ms, total / ms / 1e6);
    ms = time([&] {
        (void)hipMemcpyAsync(h, d, total / 2, hipMemcpyDeviceToHost, s[0]);
        (void)hipMemcpyAsync(h + total / 2, d + total / 2, total / 2, hipMemcpyDeviceToHost, s[1]);
    });
    printf("two streams, halves: %.2f ms = %.1f GB/s\n", ms, total / ms / 1e6);
    ms = time([&] {
        for (int c = 0; c < 8; ++c)
            (void)hipMemcpyAsync(h + c * (total / 8), d + c * (total / 8), total / 8, hipMemcpyDeviceToHost, s[c & 1]);
    });
    printf("two streams, 8 chunks alternating: %.2f ms = %.1f GB/s\n", ms, total / ms / 1e6);
    ms = time([&] { hipLaunchKernelGGL(store_to_host, dim3(1024), dim3(256), 0, s[0], (const float4*)d, (float4*)h, total / 16); });
    printf("kernel storing to mapped host memory: %.2f ms = %.1f GB/s\n", ms, total / ms / 1e6);
    ms = time([&] {
        (void)hipMemcpyAsync(h, d, total / 2, hipMemcpyDeviceToHost, s[0]);
        hipLaunchKernelGGL(store_to_host, dim3(1024), dim3(256), 0, s[1], (const float4*)(d + total / 2), (float4*)(h + total / 2), total / 32);
    });
    printf("copy engine + kernel stores, halves: %.2f ms = %.1f GB/s\n", ms, total / ms / 1e6);
    return 0;
}
