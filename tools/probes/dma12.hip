// Probe: LDS layout produced by global_load_lds with 12-byte elements
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
__global__ void k(const float* src, float* out) {
    __shared__ __attribute__((aligned(16))) float lds[512];
    for (int i = threadIdx.x; i < 512; i += 64) lds[i] = -1.f;
    __syncthreads();
    __builtin_amdgcn_global_load_lds((gbl_void*)(src + 3 * threadIdx.x), (lds_void*)lds, 12, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}
int main() {
    float *ds, *dout;
    (void)hipMalloc(&ds, 4096); (void)hipMalloc(&dout, 2048);
    std::vector<float> s(1024), o(512);
    for (int i = 0; i < 1024; ++i) s[i] = (float)i;
    (void)hipMemcpy(ds, s.data(), 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, ds, dout);
    (void)hipMemcpy(o.data(), dout, 2048, hipMemcpyDeviceToHost);
    for (int i = 0; i < 40; ++i) printf("%g ", o[i]);
    printf("\n...\n");
    for (int i = 180; i < 200; ++i) printf("%g ", o[i]);
    printf("\nlast written index: ");
    int last = -1; for (int i = 0; i < 512; ++i) if (o[i] >= 0) last = i;
    printf("%d\n", last);
    return 0;
}
