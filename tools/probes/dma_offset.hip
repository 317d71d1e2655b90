// Probe: global_load_lds_dwordx4 in its SGPR-base + VGPR-offset form with an instruction offset --
// which addresses does `offset:N` move (global source, LDS destination, or both)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* src, float* out) {
    __shared__ __attribute__((aligned(16))) float lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = -1.f;
    __syncthreads();
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds;
    const unsigned voff = threadIdx.x * 16;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024"
                 :: "v"(voff), "s"(src), "s"(lds0) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 64) out[i] = lds[i];
}
int main() {
    float *ds, *dout;
    (void)hipMalloc(&ds, 16384); (void)hipMalloc(&dout, 8192);
    std::vector<float> s(4096), o(2048);
    for (int i = 0; i < 4096; ++i) s[i] = (float)i;
    (void)hipMemcpy(ds, s.data(), 16384, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, ds, dout);
    (void)hipMemcpy(o.data(), dout, 8192, hipMemcpyDeviceToHost);
    int first = -1, last = -1;
    for (int i = 0; i < 2048; ++i) if (o[i] >= 0) { if (first < 0) first = i; last = i; }
    printf("LDS floats written: [%d, %d]; first values %g %g %g %g ... value at last %g\n", first, last,
           first >= 0 ? o[first] : -1, first >= 0 ? o[first + 1] : -1, first >= 0 ? o[first + 4] : -1,
           first >= 0 ? o[first + 5] : -1, last >= 0 ? o[last] : -1);
    printf("(source element 256 = byte 1024: offset moved the global address if first value is 256;\n"
           " LDS float 256 = byte 1024: offset moved the LDS address if first written index is 256)\n");
    return 0;
}
