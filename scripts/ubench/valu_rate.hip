// Issue cost of single VALU instructions for ONE wave per SIMD (the attention v4 situation): cycles per instruction of a stream of
// independent instructions of one kind, by s_memtime.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int KIND>
__global__ __launch_bounds__(256, 1) void k(unsigned long long* out, float* sink, float seed) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)sink, 0, 256 * 256 * 4, 0x00020000);
    const int voff = (threadIdx.x & 63) * 16;
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned long long t0, t1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int it = 0; it < 16; ++it) {
        if (KIND == 0) asm volatile(REP64("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        if (KIND == 1) asm volatile(REP64("v_mul_f32 %0, %0, %0\n\tv_mul_f32 %1, %1, %1\n\tv_mul_f32 %2, %2, %2\n\tv_mul_f32 %3, %3, %3\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        if (KIND == 2) asm volatile(REP64("v_cvt_pk_bf16_f32 %0, %4, %5\n\tv_cvt_pk_bf16_f32 %1, %5, %6\n\tv_cvt_pk_bf16_f32 %2, %6, %7\n\tv_cvt_pk_bf16_f32 %3, %7, %4\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7));
        if (KIND == 3) asm volatile(REP64("v_max3_f32 %0, %0, %4, %5\n\tv_max3_f32 %1, %1, %5, %6\n\tv_max3_f32 %2, %2, %6, %7\n\tv_max3_f32 %3, %3, %7, %4\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7));
        if (KIND == 4) asm volatile(REP64("v_exp_f32 %0, %0\n\tv_mul_f32 %1, %1, %1\n\tv_mul_f32 %2, %2, %2\n\tv_mul_f32 %3, %3, %3\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        if (KIND == 5) asm volatile(REP64("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_mul_f32 %2, %2, %2\n\tv_mul_f32 %3, %3, %3\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        if (KIND == 6) asm volatile(REP64("v_pk_mul_f32 %0, %0, %0\n\tv_pk_mul_f32 %1, %1, %1\n\tv_pk_mul_f32 %0, %0, %0\n\tv_pk_mul_f32 %1, %1, %1\n\t") : "+v"(*(double*)&a0), "+v"(*(double*)&a2));
        if (KIND == 7) asm volatile(REP64("v_exp_f16 %0, %0\n\tv_exp_f16 %1, %1\n\tv_exp_f16 %2, %2\n\tv_exp_f16 %3, %3\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        if (KIND == 8) asm volatile(REP64("v_exp_f32 %0, %0\n\ts_nop 0\n\tv_exp_f32 %1, %1\n\ts_nop 0\n\t") : "+v"(a0), "+v"(a1));
        if (KIND == 9) asm volatile(REP64("v_exp_f32 %0, %0\n\tv_mfma_f32_32x32x16_bf16 a[0:15], v[200:203], v[204:207], a[0:15]\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\t") : "+v"(a0), "+v"(a1), "+v"(a2) :: "a0","a15","v200","v207");
        if (KIND >= 10 && KIND < 20) {  // one 32x32x16 MFMA + (KIND - 10) + 3 v_mul per group
            constexpr int NM = KIND - 10 + 3;
            for (int r = 0; r < 64; ++r) {
                asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], v[200:203], v[204:207], a[0:15]" ::: "a0", "a15", "v200", "v207");
#pragma unroll
                for (int m = 0; m < NM; ++m) {  // independent fillers (eight registers in rotation)
                    float& x = (m & 7) == 0 ? a0 : (m & 7) == 1 ? a1 : (m & 7) == 2 ? a2 : (m & 7) == 3 ? a3 : (m & 7) == 4 ? a4 : (m & 7) == 5 ? a5 : (m & 7) == 6 ? a6 : a7;
                    asm volatile("v_mul_f32 %0, %0, %0" : "+v"(x));
                }
            }
        }
        if (KIND >= 20 && KIND < 30) {  // two 16x16x32 MFMAs + (KIND - 20) + 2 v_mul per group
            constexpr int NM = KIND - 20 + 2;
            for (int r = 0; r < 64; ++r) {
                asm volatile("v_mfma_f32_16x16x32_bf16 a[0:3], v[200:203], v[204:207], a[0:3]\n\tv_mfma_f32_16x16x32_bf16 a[4:7], v[200:203], v[208:211], a[4:7]" ::: "a0", "a7", "v200", "v211");
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    float& x = (m & 7) == 0 ? a0 : (m & 7) == 1 ? a1 : (m & 7) == 2 ? a2 : (m & 7) == 3 ? a3 : (m & 7) == 4 ? a4 : (m & 7) == 5 ? a5 : (m & 7) == 6 ? a6 : a7;
                    asm volatile("v_mul_f32 %0, %0, %0" : "+v"(x));
                }
            }
        }
        if (KIND >= 30 && KIND < 40) {  // MFMAs of one kind (+ one LDS-DMA piece per group for the even kinds): what a buffer_load ... lds costs the stream
            for (int r = 0; r < 64; ++r) {
                if (KIND < 32) asm volatile("v_mfma_f32_16x16x32_bf16 a[0:3], v[200:203], v[204:207], a[0:3]\n\tv_mfma_f32_16x16x32_bf16 a[4:7], v[200:203], v[208:211], a[4:7]\n\t"
                                            "v_mfma_f32_16x16x32_bf16 a[8:11], v[200:203], v[204:207], a[8:11]\n\tv_mfma_f32_16x16x32_bf16 a[12:15], v[200:203], v[208:211], a[12:15]\n\t"
                                            "v_mfma_f32_16x16x32_bf16 a[16:19], v[200:203], v[204:207], a[16:19]\n\tv_mfma_f32_16x16x32_bf16 a[20:23], v[200:203], v[208:211], a[20:23]\n\t"
                                            "v_mfma_f32_16x16x32_bf16 a[24:27], v[200:203], v[204:207], a[24:27]\n\tv_mfma_f32_16x16x32_bf16 a[28:31], v[200:203], v[208:211], a[28:31]" ::: "a0", "a31", "v200", "v211");
                else asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], v[200:203], v[204:207], a[0:15]\n\tv_mfma_f32_32x32x16_bf16 a[16:31], v[200:203], v[208:211], a[16:31]\n\t"
                                  "v_mfma_f32_32x32x16_bf16 a[32:47], v[200:203], v[204:207], a[32:47]\n\tv_mfma_f32_32x32x16_bf16 a[48:63], v[200:203], v[208:211], a[48:63]" ::: "a0", "a63", "v200", "v211");
                if ((KIND & 1) == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(smem + (r & 7) * 1024), 16, voff, (r & 15) * 1024, 0, 0);
                if ((r & 7) == 7) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int KIND>
void run(const char* name, int per_iter) {
    unsigned long long* d; float* s;
    hipMalloc(&d, 256 * 8); hipMalloc(&s, 256 * 256 * 4);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256), 8192, 0, d, s, 0.001f);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256), 8192, 0, d, s, 0.001f);
    hipDeviceSynchronize();
    unsigned long long h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double n = 16.0 * 64 * per_iter;
    printf("%-44s %7.2f ticks / instruction (block 0: %llu ticks for %.0f instructions)\n", name, h[0] / n, h[0], n);
    hipFree(d); hipFree(s);
}
int main() {
    run<1>("v_mul_f32", 4);
    run<0>("v_exp_f32", 4);
    run<2>("v_cvt_pk_bf16_f32", 4);
    run<3>("v_max3_f32", 4);
    run<4>("1 v_exp_f32 + 3 v_mul_f32", 4);
    run<5>("2 v_exp_f32 + 2 v_mul_f32", 4);
    run<6>("v_pk_mul_f32", 4);
    run<7>("v_exp_f16", 4);
    run<8>("v_exp_f32 + s_nop 0 (pairs)", 4);
    run<9>("1 MFMA 32x32x16 + 3 v_exp_f32 (per 4)", 4);
    // ticks per GROUP = 4 x the printed value when per_iter is set to 4: print per group instead
    run<10>("[per group/4] MFMA32 + 3 v_mul", 4);
    run<12>("[per group/4] MFMA32 + 5 v_mul", 4);
    run<13>("[per group/4] MFMA32 + 6 v_mul", 4);
    run<14>("[per group/4] MFMA32 + 7 v_mul", 4);
    run<15>("[per group/4] MFMA32 + 8 v_mul", 4);
    run<17>("[per group/4] MFMA32 + 10 v_mul", 4);
    run<20>("[per group/4] 2 MFMA16 + 2 v_mul", 4);
    run<22>("[per group/4] 2 MFMA16 + 4 v_mul", 4);
    run<23>("[per group/4] 2 MFMA16 + 5 v_mul", 4);
    run<24>("[per group/4] 2 MFMA16 + 6 v_mul", 4);
    run<26>("[per group/4] 2 MFMA16 + 8 v_mul", 4);
    // groups of 128 matrix-pipe cycles: ticks per group = 4 x the printed value
    run<31>("[per group/4] 8 MFMA16", 4);
    run<30>("[per group/4] 8 MFMA16 + 1 buffer_load lds", 4);
    run<33>("[per group/4] 4 MFMA32", 4);
    run<32>("[per group/4] 4 MFMA32 + 1 buffer_load lds", 4);
    return 0;
}
