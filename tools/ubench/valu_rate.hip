// Micro-benchmark: issue cost of wave64 VALU instructions on gfx950 with 1, 2 and 4 waves per SIMD.
// Prints shader cycles (s_memtime) per instruction per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float float2v __attribute__((ext_vector_type(2)));
constexpr int N = 4096;   // instructions of the kind under test per wave (8 independent chains)

template <int KIND>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* cyc, float seed) {
    __shared__ unsigned lds[4096];
    float a[8]; float2v p[8];
    for (int i = 0; i < 8; i++) { a[i] = seed + i + threadIdx.x; p[i] = float2v{seed + i, seed - i} + (float)threadIdx.x; }
    unsigned u[8]; for (int i = 0; i < 8; i++) u[i] = threadIdx.x * 2654435761u + i;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i;
    unsigned lds_addr = (threadIdx.x * 4u) & 16383u;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < N / 32; it++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i+1)&7]), "v"(a[(i+2)&7]));
            if (KIND == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i+1)&7]));
            if (KIND == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i+1)&7]));
            if (KIND == 3) asm volatile("v_fmaak_f32 %0, %0, %1, 0x44000000" : "+v"(a[i]) : "v"(a[(i+1)&7]));
            if (KIND == 4) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i+1)&7]));
            if (KIND == 5) asm volatile("v_max_f32 %0, |%0|, |%1|" : "+v"(a[i]) : "v"(a[(i+1)&7]));
            if (KIND == 6) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i+1)&7]), "v"(a[(i+2)&7]));
            if (KIND == 7) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i+1)&7]), "v"(a[(i+2)&7]));
            if (KIND == 8) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i+1)&7]), "v"(a[(i+2)&7]));
            if (KIND == 9) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));
            if (KIND == 10) asm volatile("v_fract_f32 %0, %0" : "+v"(a[i]));
            if (KIND == 11) asm volatile("v_trunc_f32 %0, %0" : "+v"(a[i]));
            if (KIND == 12) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i]));
            if (KIND == 13) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(a[i]));
            if (KIND == 14) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[i]));
            if (KIND == 15) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(u[i]) : "v"(a[(i+1)&7]));
            if (KIND == 16) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a[i]) : "v"(u[(i+1)&7]));
            if (KIND == 17) asm volatile("v_mov_b32 %0, %1" : "+v"(u[i]) : "v"(u[(i+1)&7]));
            if (KIND == 18) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i+1)&7]));
            if (KIND == 19) asm volatile("v_and_b32 %0, 0x7fc, %0" : "+v"(u[i]));
            if (KIND == 20) asm volatile("v_or_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i+1)&7]));
            if (KIND == 21) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i+1)&7]));
            if (KIND == 22) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i+1)&7]), "v"(u[(i+2)&7]));
            if (KIND == 23) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i+1)&7]), "v"(u[(i+2)&7]));
            if (KIND == 24) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(u[i]));
            if (KIND == 25) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(u[i]));
            if (KIND == 26) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(u[i]) : "v"(u[(i+1)&7]));
            if (KIND == 27) asm volatile("v_lshl_or_b32 %0, %0, 2, %1" : "+v"(u[i]) : "v"(u[(i+1)&7]));
            if (KIND == 28) asm volatile("v_bfe_u32 %0, %0, %1, 4" : "+v"(u[i]) : "v"(u[(i+1)&7]));
            if (KIND == 29) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(u[i]) : "v"(u[(i+1)&7]), "v"(u[(i+2)&7]));
            if (KIND == 30) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i+1)&7]), "v"(u[(i+2)&7]));
            if (KIND == 31) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i+1)&7]));
            if (KIND == 32) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i+1)&7]));
            if (KIND == 33) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i+1)&7]), "v"(u[(i+2)&7]));
            if (KIND == 34) asm volatile("v_min_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i+1)&7]));
            if (KIND == 35) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i+1)&7]));
            if (KIND == 36) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i+1)&7]), "v"(u[(i+2)&7]));
            if (KIND == 37) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i+1)&7]));
            if (KIND == 38) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(u[(i+1)&7]));
            if (KIND == 39) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(u[i]) : "v"(u[(i+1)&7]));
            if (KIND == 40) asm volatile("v_cndmask_b32_e64 %0, 0, %0, s[20:21]" : "+v"(u[i]));
            if (KIND == 41) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a[i]), "v"(a[(i+1)&7]) : "vcc");
            if (KIND == 42) asm volatile("v_cmp_lt_f32_e64 s[22:23], %0, %1" :: "v"(a[i]), "v"(a[(i+1)&7]) : "s22", "s23");
            if (KIND == 43) asm volatile("v_cmp_eq_u32_e64 s[22:23], %0, %1" :: "v"(u[i]), "v"(u[(i+1)&7]) : "s22", "s23");
            if (KIND == 44) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(p[(i+1)&7]), "v"(p[(i+2)&7]));
            if (KIND == 45) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i+1)&7]));
            if (KIND == 46) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i+1)&7]));
            if (KIND == 47) asm volatile("s_and_b64 s[24:25], s[20:21], s[22:23]" ::: "s24", "s25");
            if (KIND == 48) asm volatile("ds_read_b32 %0, %1" : "=v"(u[i]) : "v"(lds_addr));
        }
        }
        if (KIND == 48) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; unsigned su = 0;
    for (int i = 0; i < 8; i++) { s += a[i] + p[i].x + p[i].y; su += u[i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)su + lds[threadIdx.x];
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int KIND>
void run(const char* name, float* out, unsigned long long* cyc) {
    const int threads[3] = {256, 512, 1024};
    printf("%-22s", name);
    for (int t = 0; t < 3; t++) {
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads[t]), 0, 0, out, cyc, 1.0f);
        (void)hipDeviceSynchronize();
        int nw = 256 * threads[t] / 64;
        std::vector<unsigned long long> h(nw);
        (void)hipMemcpy(h.data(), cyc, nw * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        double med = (double)h[nw / 2];
        int wps = threads[t] / 256;
        printf("  %dw: %5.2f /wave %5.2f /SIMD", wps, med / N, med / N / wps);
    }
    printf("\n");
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);   // every result reaches the file even if a later kind dies
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 1024 * sizeof(float));
    (void)hipMalloc(&cyc, 256 * 16 * sizeof(unsigned long long));
    run<0>("v_fma_f32", out, cyc);
    run<1>("v_mul_f32", out, cyc);
    run<2>("v_add_f32", out, cyc);
    run<3>("v_fmaak_f32", out, cyc);
    run<4>("v_max_f32", out, cyc);
    run<5>("v_max_f32 |abs|", out, cyc);
    run<6>("v_min3_f32", out, cyc);
    run<7>("v_max3_f32", out, cyc);
    run<8>("v_med3_f32", out, cyc);
    run<9>("v_floor_f32", out, cyc);
    run<10>("v_fract_f32", out, cyc);
    run<11>("v_trunc_f32", out, cyc);
    run<12>("v_cvt_i32_f32", out, cyc);
    run<13>("v_cvt_u32_f32", out, cyc);
    run<14>("v_cvt_f32_u32", out, cyc);
    run<15>("v_cvt_pk_u8_f32", out, cyc);
    run<16>("v_ldexp_f32", out, cyc);
    run<17>("v_mov_b32", out, cyc);
    run<18>("v_and_b32", out, cyc);
    run<19>("v_and_b32 lit", out, cyc);
    run<20>("v_or_b32", out, cyc);
    run<21>("v_xor_b32", out, cyc);
    run<22>("v_or3_b32", out, cyc);
    run<23>("v_and_or_b32", out, cyc);
    run<24>("v_lshlrev_b32", out, cyc);
    run<25>("v_lshrrev_b32", out, cyc);
    run<26>("v_lshl_add_u32", out, cyc);
    run<27>("v_lshl_or_b32", out, cyc);
    run<28>("v_bfe_u32", out, cyc);
    run<29>("v_bfi_b32", out, cyc);
    run<30>("v_perm_b32", out, cyc);
    run<31>("v_add_u32", out, cyc);
    run<32>("v_sub_u32", out, cyc);
    run<33>("v_add3_u32", out, cyc);
    run<34>("v_min_u32", out, cyc);
    run<35>("v_mul_u32_u24", out, cyc);
    run<36>("v_mad_u32_u24", out, cyc);
    run<37>("v_mul_lo_u32", out, cyc);
    run<38>("v_cndmask e32(vcc)", out, cyc);
    run<39>("v_cndmask e64(sgpr)", out, cyc);
    run<40>("v_cndmask e64 lit", out, cyc);
    run<41>("v_cmp_lt_f32 e32", out, cyc);
    run<42>("v_cmp_lt_f32 e64", out, cyc);
    run<43>("v_cmp_eq_u32 e64", out, cyc);
    run<44>("v_pk_fma_f32", out, cyc);
    run<45>("v_pk_mul_f32", out, cyc);
    run<46>("v_pk_add_f32", out, cyc);
    // (kinds 47 "s_and_b64" and 48 "ds_read_b32" are not part of the VALU table; the SALU-only loop did not finish within 100 s at
    // two waves per SIMD on the round-3 box and is left out of the default run)
    return 0;
}
