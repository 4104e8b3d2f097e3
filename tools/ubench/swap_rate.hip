// Micro-benchmark: issue cost of v_swap_b32 against v_mov_b32 and v_cndmask_b32 (gfx950, 4 waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
constexpr int N = 4096;
template <int KIND>
__global__ __launch_bounds__(1024) void k(unsigned* out, unsigned long long* cyc) {
    unsigned u[8]; for (int i = 0; i < 8; i++) u[i] = threadIdx.x * 2654435761u + i;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < N / 32; it++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (KIND == 0) asm volatile("v_mov_b32 %0, %1" : "+v"(u[i]) : "v"(u[(i+1)&7]));
            if (KIND == 1) asm volatile("v_swap_b32 %0, %1" : "+v"(u[i]), "+v"(u[(i+1)&7]));
            if (KIND == 2) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(u[i]) : "v"(u[(i+1)&7]));
        }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned su = 0;
    for (int i = 0; i < 8; i++) su += u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = su;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}
template <int KIND> void run(const char* name, unsigned* out, unsigned long long* cyc) {
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(1024), 0, 0, out, cyc);
    (void)hipDeviceSynchronize();
    const int nw = 256 * 16;
    std::vector<unsigned long long> h(nw);
    (void)hipMemcpy(h.data(), cyc, nw * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-16s 4w: %5.2f cycles /SIMD\n", name, (double)h[nw / 2] / N / 4);
}
int main() {
    unsigned* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 1024 * sizeof(unsigned));
    (void)hipMalloc(&cyc, 256 * 16 * sizeof(unsigned long long));
    run<0>("v_mov_b32", out, cyc);
    run<1>("v_swap_b32", out, cyc);
    run<2>("v_cndmask_b32", out, cyc);
    return 0;
}
