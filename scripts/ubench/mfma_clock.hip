// Microbenchmark (GPU box): shader clock (s_memtime ticks / wall time) and MFMA throughput of a register-only
// v_mfma_f32_32x32x16_bf16 loop, with zero vs random operands, and of a VALU-only loop.  Calibrates the power wall that
// DESIGN.md 5.1 describes:  hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_clock.hip -o /tmp/mfma_clock && /tmp/mfma_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ __launch_bounds__(256) void mfma_loop(const unsigned* seed, unsigned long long* cyc, float* sink, int iters, int mode) {
    const int tid = threadIdx.x + blockIdx.x * blockDim.x;
    unsigned s = seed[tid % 4096];
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 8; ++e) {
            s = s * 1664525u + 1013904223u;
            float fa = (mode == 0 || mode == 3) ? 0.f : ((int)(s >> 8) % 2001 - 1000) * 1e-3f;
            s = s * 1664525u + 1013904223u;
            float fb = (mode == 0 || mode == 3) ? 0.f : ((int)(s >> 8) % 2001 - 1000) * 1e-3f;
            a[i][e] = (__bf16)fa;
            b[i][e] = (__bf16)fb;
        }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mode < 2) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[i], acc[i], 0, 0, 0);
        }
    } else if (mode == 3 || mode == 4) {  // the K = 32 shape: 16x16x32, same flops per issue slot (2 instructions per 32x32x16)
        f32x4 c4[8];
        for (int i = 0; i < 8; ++i)
            for (int r = 0; r < 4; ++r) c4[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)  // in-place accumulators (the builtin form made hipcc rotate overlapping AGPR tuples)
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c4[i]) : "v"(a[i & 3]), "v"(b[(i + (i >> 2)) & 3]));
        }
        for (int i = 0; i < 8; ++i)
            for (int r = 0; r < 4; ++r) acc[i & 3][r] += c4[i][r];
    } else {
        float x = acc[0][0] + (float)s * 1e-9f, y = 1.0001f;
        for (int it = 0; it < iters * 16; ++it) {
            x = x * y + 0.5f;
            y = y * 0.9999f + 1e-4f;
        }
        acc[0][0] = x + y;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int k = 0; k < 16; ++k) r += acc[i][k];
    sink[tid] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    const int blocks = 256 * 2, threads = 256, iters = 20000;
    unsigned* seed;
    unsigned long long* cyc;
    float* sink;
    hipMalloc(&seed, 4096 * 4);
    hipMalloc(&cyc, blocks * 8);
    hipMalloc(&sink, blocks * threads * 4);
    std::vector<unsigned> h(4096);
    for (auto& v : h) v = rand();
    hipMemcpy(seed, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const char* names[5] = {"mfma 32x32x16 zero", "mfma 32x32x16 random", "valu fma chain", "mfma 16x16x32 zero", "mfma 16x16x32 random"};
    for (int mode = 0; mode < 5; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(threads), 0, 0, seed, cyc, sink, iters, mode);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> c(blocks);
            hipMemcpy(c.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
            double avg = 0;
            for (auto v : c) avg += v;
            avg /= blocks;
            const double flops = mode < 2 ? 2.0 * 32 * 32 * 16 * 4.0 * iters * (blocks * threads / 64)
                                 : (mode >= 3 ? 2.0 * 16 * 16 * 32 * 8.0 * iters * (blocks * threads / 64) : 0);
            // 2 workgroups of 4 waves per CU -> 2 waves per SIMD; kernel wall ~ per-wave loop time
            printf("%-22s rep %d: %.3f ms, %.0f loop cycles/wave -> clock %.2f GHz, %.0f TFLOP/s\n", names[mode], rep, ms, avg,
                   avg / (ms * 1e6), flops / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
