// Dependent-load latency (a pointer chase by one wave) with the rest of the GPU idle and with it streaming: does a lone chain of
// loads see a LONGER latency when little else is in flight (what the path kernel's drain looks like: DESIGN.md 5 "The drain")?
// Footprints: 1 MiB (L2), 48 MiB (beyond one XCD's 4 MiB L2: Infinity Cache), 2 GiB (HBM).  One hop = one 4-byte load whose
// address depends on the previous one; every launch walks a stretch of the chain no launch before it touched; 100 MHz clock.
// Build: hipcc --offload-arch=gfx950 -O3 chase_latency.hip -o chase_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <numeric>
#include <random>
__global__ void k_chase(const unsigned* next, unsigned start, unsigned hops, unsigned long long* out, const uint4* stream, size_t stream_n, unsigned* sink,
                        volatile unsigned* stop) {
    if (blockIdx.x == 0) {
        if (threadIdx.x >= 64) return;
        unsigned p = start;
        // warm-up hops (TLB), then the timed ones
        for (unsigned i = 0; i < 256; i++) { p = next[p]; asm volatile("" : "+v"(p)); }
        const unsigned long long t0 = wall_clock64();
        for (unsigned i = 0; i < hops; i++) { p = next[p]; asm volatile("" : "+v"(p)); }   // a vector load per hop
        const unsigned long long t1 = wall_clock64();
        if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = p; *stop = 1u; }
    } else {
        // background traffic until block 0 is done — and for 100 ms at most (10^7 ticks of the 100 MHz clock): every wave ends
        uint4 acc = make_uint4(0, 0, 0, 0);
        size_t j = (size_t)(blockIdx.x - 1) * blockDim.x + threadIdx.x;
        const size_t step = (size_t)(gridDim.x - 1) * blockDim.x;
        const unsigned long long tb = wall_clock64();
        while (!*stop && wall_clock64() - tb < 10000000ull) {
            for (int u = 0; u < 8; u++) {
                uint4 v = stream[j];
                acc.x += v.x; acc.y ^= v.y; acc.z += v.z; acc.w ^= v.w;
                j += step; if (j >= stream_n) j -= stream_n;
            }
        }
        if (acc.x == 0x12345678u) sink[0] = acc.y + acc.z + acc.w;
    }
}
int main() {
    unsigned long long* d_out; (void)hipMalloc(&d_out, 16);
    unsigned* d_sink; (void)hipMalloc(&d_sink, 4);
    unsigned* d_stop; (void)hipMalloc(&d_stop, 4);
    const size_t stream_bytes = (size_t)4 << 30;
    uint4* d_stream; if (hipMalloc(&d_stream, stream_bytes) != hipSuccess) { printf("no memory\n"); return 1; }
    (void)hipMemset(d_stream, 1, stream_bytes);
    setvbuf(stdout, nullptr, _IONBF, 0);
    const size_t foot[3] = {(size_t)1 << 20, (size_t)48 << 20, (size_t)2 << 30};
    const char* name[3] = {"1 MiB (L2)", "48 MiB (Infinity Cache)", "2 GiB (HBM)"};
    for (int f = 0; f < 3; f++) {
        // one random cycle over lines of 128 bytes (32 words): every hop is a new line
        const size_t lines = foot[f] / 128;
        std::vector<unsigned> order(lines);
        std::iota(order.begin(), order.end(), 0u);
        std::mt19937 rng(1234);
        std::shuffle(order.begin() + 1, order.end(), rng);
        std::vector<unsigned> h(foot[f] / 4, 0u);
        for (size_t i = 0; i < lines; i++) h[(size_t)order[i] * 32] = order[(i + 1) % lines] * 32u;
        { unsigned q = 0; std::vector<unsigned> seen; for (unsigned i = 0; i < 256 + 4000; i++) { q = h[q]; seen.push_back(q); }
          std::sort(seen.begin(), seen.end()); printf("  (host walk of the chain: %zu distinct lines in 4256 hops)\n", (size_t)(std::unique(seen.begin(), seen.end()) - seen.begin())); }
        unsigned* d_next; if (hipMalloc(&d_next, foot[f]) != hipSuccess) { printf("no memory\n"); return 1; }
        (void)hipMemcpy(d_next, h.data(), foot[f], hipMemcpyHostToDevice);
        const unsigned hops = 4000;
        for (int busy = 0; busy < 3; busy++) {
            const int grid = busy == 0 ? 1 : (busy == 1 ? 65 : 2049);   // 0, 64, 2048 streaming workgroups of 256 threads
            for (int rep = 0; rep < 2; rep++) {
                (void)hipMemset(d_stop, 0, 4);
                hipLaunchKernelGGL(k_chase, dim3(grid), dim3(256), 0, 0, d_next, order[((size_t)(busy * 2 + rep) * 20011u) % lines] * 32u, hops, d_out, d_stream, stream_bytes / 16, d_sink, d_stop);
                if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
                unsigned long long o[2]; (void)hipMemcpy(o, d_out, 16, hipMemcpyDeviceToHost);
                if (rep == 1) printf("%-24s %5d streaming workgroups: %7.1f ns per dependent load\n", name[f], grid - 1, (double)o[0] * 10.0 / hops);
            }
        }
        (void)hipFree(d_next);
    }
    return 0;
}
