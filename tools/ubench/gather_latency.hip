// What does ONE round of scattered byte loads cost a wave (round 4: the step loop's byte loads take ~3800 shader cycles per step in
// k_paths' drain — tools/lab/r4/step_times.py — against ~200 for a lone pointer chase that hits L2, tools/ubench/chase_latency.hip)?
// A wave issues, per round, `ninst` buffer_load_ubyte instructions in which `active` lanes read one random byte each of a `foot`-byte
// array (the others pass an out-of-range offset, as k_paths does) and waits for all of them (s_waitcnt vmcnt(0)); the next round's
// addresses depend on the bytes read.  Ticks of s_memtime per round, for: lanes active, instructions per round, waves per CU (one
// 64- ... 1024-thread workgroup per CU, k_paths' shape), CUs busy, footprint.
// Build: hipcc --offload-arch=gfx950 -O3 gather_latency.hip -o gather_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(1024) void k_gather(const unsigned char* data, unsigned foot_mask, int rounds, int active, int ninst, int lds_bytes,
                                                 unsigned long long* out, unsigned* sink) {
    __shared__ unsigned char s_pad[120 * 1024];   // occupies LDS like k_paths' nibble map does (one workgroup per CU)
    if (lds_bytes > 0 && threadIdx.x == 0) s_pad[lds_bytes - 1] = 1;
    const unsigned lane = threadIdx.x & 63u;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(data), (short)0, (int)(foot_mask + 1u), 0x00020000);
    unsigned x[4];
    for (int k = 0; k < 4; k++) x[k] = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u * (k + 1);
    unsigned acc = 0;
    unsigned long long t0 = 0;
    for (int r = -16; r < rounds; r++) {
        if (r == 0) t0 = clock64();
        unsigned v[4] = {0, 0, 0, 0};
        for (int k = 0; k < 4; k++)
            if (k < ninst) v[k] = __builtin_amdgcn_raw_buffer_load_b8(rsrc, (int)lane < active ? (x[k] & foot_mask) : 0xFFFFFFFFu, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int k = 0; k < 4; k++) { x[k] = x[k] * 1664525u + 1013904223u + v[k]; acc += v[k]; }
    }
    const unsigned long long t1 = clock64();
    if (lane == 0) atomicAdd(out, t1 - t0);
    if (acc == 0x7FFFFFFFu) sink[0] = acc;
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    unsigned long long* d_out; (void)hipMalloc(&d_out, 8);
    unsigned* d_sink; (void)hipMalloc(&d_sink, 4);
    const int rounds = 2000;
    for (size_t foot : {(size_t)1 << 20, (size_t)16 << 20, (size_t)128 << 20, (size_t)1 << 30}) {
        unsigned char* d; if (hipMalloc(&d, foot) != hipSuccess) { printf("no memory\n"); return 1; }
        std::vector<unsigned char> h(foot);
        for (size_t i = 0; i < foot; i++) h[i] = (unsigned char)(rand() & 7);
        (void)hipMemcpy(d, h.data(), foot, hipMemcpyHostToDevice);
        printf("## footprint %zu MiB\n", foot >> 20);
        for (int cus : {1, 256})
            for (int waves : {1, 4, 16})
                for (int ninst : {1, 4})
                    for (int active : {1, 8, 20, 64}) {
                        if (foot != ((size_t)16 << 20) && (ninst != 4 || (active != 20 && active != 64))) continue;   // the full sweep at 16 MiB only
                        (void)hipMemset(d_out, 0, 8);
                        hipLaunchKernelGGL(k_gather, dim3(cus), dim3(64 * waves), 0, 0, d, (unsigned)(foot - 1), rounds, active, ninst, 120 * 1024, d_out, d_sink);
                        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
                        unsigned long long o; (void)hipMemcpy(&o, d_out, 8, hipMemcpyDeviceToHost);
                        printf("CUs %3d  waves/CU %2d  loads/round %d  active lanes %2d : %7.0f ticks per round\n", cus, waves, ninst, active,
                               (double)o / ((double)cus * waves) / rounds);
                    }
        (void)hipFree(d);
    }
    return 0;
}
