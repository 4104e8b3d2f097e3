// Cache-policy bits on scattered byte loads (round 4): tools/ubench/gather_latency.hip showed that a CU serves ~one scattered lane
// per 2.7 shader cycles however few waves ask (every lane = one 128-byte line into L1 for one byte).  Does a load that does not
// allocate in L1 (sc0 / sc1 / nt in the instruction's cache-policy field) cost less?  Same kernel shape as gather_latency (one
// 1024-thread workgroup per CU, 120 KiB of LDS held), 4 loads per round, 20 or 64 active lanes, 16 MiB and 1 GiB footprints.
// Build: hipcc --offload-arch=gfx950 -O3 gather_policy.hip -o gather_policy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int AUX>
__global__ __launch_bounds__(1024) void k_gather(const unsigned char* data, unsigned foot_mask, int rounds, int active, unsigned long long* out, unsigned* sink) {
    __shared__ unsigned char s_pad[120 * 1024];
    if (threadIdx.x == 0) s_pad[120 * 1024 - 1] = 1;
    const unsigned lane = threadIdx.x & 63u;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(data), (short)0, (int)(foot_mask + 1u), 0x00020000);
    unsigned x[4];
    for (int k = 0; k < 4; k++) x[k] = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u * (k + 1);
    unsigned acc = 0;
    unsigned long long t0 = 0;
    for (int r = -16; r < rounds; r++) {
        if (r == 0) t0 = clock64();
        unsigned v[4];
        for (int k = 0; k < 4; k++) v[k] = __builtin_amdgcn_raw_buffer_load_b8(rsrc, (int)lane < active ? (x[k] & foot_mask) : 0xFFFFFFFFu, 0, AUX);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int k = 0; k < 4; k++) { x[k] = x[k] * 1664525u + 1013904223u + v[k]; acc += v[k]; }
    }
    const unsigned long long t1 = clock64();
    if (lane == 0) atomicAdd(out, t1 - t0);
    if (acc == 0x7FFFFFFFu) sink[0] = acc;
}

template <int AUX>
void run(const unsigned char* d, size_t foot, unsigned long long* d_out, unsigned* d_sink) {
    const int rounds = 2000;
    for (int waves : {4, 16})
        for (int active : {20, 64}) {
            (void)hipMemset(d_out, 0, 8);
            hipLaunchKernelGGL(k_gather<AUX>, dim3(256), dim3(64 * waves), 0, 0, d, (unsigned)(foot - 1), rounds, active, d_out, d_sink);
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); exit(1); }
            unsigned long long o; (void)hipMemcpy(&o, d_out, 8, hipMemcpyDeviceToHost);
            printf("aux %2d  footprint %4zu MiB  waves/CU %2d  active lanes %2d : %7.0f ticks per round of 4 loads\n", AUX, foot >> 20, waves, active,
                   (double)o / (256.0 * waves) / rounds);
        }
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    unsigned long long* d_out; (void)hipMalloc(&d_out, 8);
    unsigned* d_sink; (void)hipMalloc(&d_sink, 4);
    for (size_t foot : {(size_t)16 << 20, (size_t)1 << 30}) {
        unsigned char* d; if (hipMalloc(&d, foot) != hipSuccess) { printf("no memory\n"); return 1; }
        std::vector<unsigned char> h(foot);
        for (size_t i = 0; i < foot; i++) h[i] = (unsigned char)(rand() & 7);
        (void)hipMemcpy(d, h.data(), foot, hipMemcpyHostToDevice);
        run<0>(d, foot, d_out, d_sink);    // default
        run<1>(d, foot, d_out, d_sink);    // sc0
        run<2>(d, foot, d_out, d_sink);    // nt
        run<3>(d, foot, d_out, d_sink);    // sc0 nt
        run<16>(d, foot, d_out, d_sink);   // sc1
        run<17>(d, foot, d_out, d_sink);   // sc0 sc1
        run<18>(d, foot, d_out, d_sink);   // sc1 nt
        (void)hipFree(d);
    }
    return 0;
}
