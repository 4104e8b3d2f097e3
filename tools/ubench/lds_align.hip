// What does ds_read_b32 return for an address that is not a multiple of 4 on this device / runtime configuration: the aligned
// dword (low address bits ignored) or the unaligned one?  (Decides whether the swizzle-table index of rt_pslot.hpp needs its mask.)
// Build: hipcc --offload-arch=gfx950 -O3 lds_align.hip -o lds_align
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    __shared__ unsigned w[64];
    w[threadIdx.x] = 0x11000000u * (threadIdx.x & 7u) + threadIdx.x * 0x0101u + 0x00AB0000u;
    __syncthreads();
    unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned*)w + 16u + (threadIdx.x & 3u);   // word 4, byte offset 0..3
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
    out[threadIdx.x] = v;
    if (threadIdx.x == 0) { out[64] = w[4]; out[65] = w[5]; }
}
int main() {
    unsigned* d; (void)hipMalloc(&d, 66 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[66]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("w[4] = %08x  w[5] = %08x\n", h[64], h[65]);
    for (int o = 0; o < 4; o++) printf("ds_read_b32 at byte offset +%d: %08x  (%s)\n", o, h[o], h[o] == h[64] ? "aligned word" : "NOT the aligned word");
    return 0;
}
