// Diagnostics (GPU box) without the Python start-up cost: s_memtime breakdowns of the GEMM trace builds through the C ABI
// (lt_op_gemm_trace) next to event timings of the product kernels.  Variants: 3 = 8-wave ping-pong, 10 = 4 waves LDS-DMA,
// 12 = 4 waves VGPR-staged.  Build (here or on the box), from the repo root:
//   hipcc -O2 scripts/ubench/gemm_trace_native.cpp -Iinclude -Llumina-t2x_amd/lib -llumina_dit \
//         -Wl,-rpath,'$ORIGIN/../../lumina-t2x_amd/lib' -o scripts/ubench/gemm_trace_native
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "lumina_dit.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define LT(x) do { int r_ = (x); if (r_ != 0) { printf("lt error %d (%s) at line %d\n", r_, lt_last_error(), __LINE__); return 1; } } while (0)

static void fill_bf16(std::vector<uint16_t>& v, unsigned seed, int exp_lo) {  // random sign / mantissa, 7 exponents from exp_lo
    unsigned s = seed;
    for (auto& x : v) {
        s = s * 1664525u + 1013904223u;
        x = (uint16_t)(((s >> 31) << 15) | (((unsigned)exp_lo + ((s >> 8) % 7u)) << 7) | ((s >> 16) & 0x7fu));
    }
}

int main(int argc, char** argv) {
    struct Shape { const char* name; int M, N, K; };
    const Shape shapes[] = {{"w13", 8192, 12288, 2304}, {"qkv", 8192, 6912, 2304}};
    const int nshape = argc > 1 ? atoi(argv[1]) : 2;
    struct Var { int id, nw; const char* names[6]; };
    const Var vars[] = {
        {3, 8, {"ds_issue", "vm_wait", "lgkm_wait", "bar1", "mfma", "bar2"}},
        {10, 4, {"stream", "vm_wait", "lgkm_wait", "barrier", "-", "-"}},
        {12, 4, {"h0_reads", "vm_wait", "h0_writes", "lgkm_wait", "barrier", "h1"}},
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int si = 0; si < nshape && si < 2; ++si) {
        const Shape sh = shapes[si];
        std::vector<uint16_t> hA((size_t)sh.M * sh.K), hW((size_t)sh.N * sh.K);
        fill_bf16(hA, 1u + si, 121);  // |a| in [2^-6, 2)
        fill_bf16(hW, 77u + si, 115);  // |w| in [2^-12, 2^-5)
        void *A, *W, *C, *T;
        CK(hipMalloc(&A, hA.size() * 2));
        CK(hipMalloc(&W, hW.size() * 2));
        CK(hipMalloc(&C, (size_t)sh.M * sh.N * 2));
        CK(hipMalloc(&T, 64 * 12 * 8 * 8));
        CK(hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
        const double flops = 2.0 * sh.M * sh.N * sh.K;
        for (const Var& v : vars) {
            float ms_plain = 0.f, ms_trace = 0.f;
            for (int pass = 0; pass < 2; ++pass) {
                CK(hipMemset(T, 0, 64 * 12 * 8 * 8));
                for (int i = 0; i < 13; ++i) {
                    if (i == 3) CK(hipEventRecord(e0, 0));
                    if (pass == 0) LT(lt_op_gemm_bf16(A, W, nullptr, 1, C, sh.M, sh.N, sh.K, 0, v.id, nullptr));
                    else LT(lt_op_gemm_trace(A, W, C, sh.M, sh.N, sh.K, v.id, T, nullptr));
                }
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(pass == 0 ? &ms_plain : &ms_trace, e0, e1));
            }
            std::vector<unsigned long long> t(64 * 12 * 8);
            CK(hipMemcpy(t.data(), T, t.size() * 8, hipMemcpyDeviceToHost));
            const unsigned long long v6 = t[6], v7 = t[7];
            const double ns = (double)(v6 >> 32), pro = (double)(v6 & 0xffffffffu), loop = (double)(v7 >> 20), epi = (double)(v7 & 0xfffff);
            const int tiles = ((sh.M + 255) / 256) * ((sh.N + 255) / 256);
            printf("== %s M%d N%d K%d variant %d: %.1f us (%.0f TFLOP/s), trace build %.1f us; tiles %d (%.2f per CU); wave 0 of block 5: "
                   "prologue %.0f + main loop %.0f + epilogue %.0f ticks, %.0f slabs -> %.0f ticks per slab\n",
                   sh.name, sh.M, sh.N, sh.K, v.id, ms_plain * 100.0, flops / (ms_plain * 1e-4) / 1e12, ms_trace * 100.0, tiles, tiles / 256.0,
                   pro, loop, epi, ns, ns > 0 ? loop / ns : 0.0);
            for (int blk = 0; blk < 2; ++blk)
                for (int w = 0; w < v.nw; ++w) {
                    const unsigned long long* o = &t[((size_t)blk * v.nw + w) * 8];
                    const double nsl = (double)(o[6] >> 32);
                    if (nsl == 0) continue;
                    double sum = 0;
                    printf("   blk %3d wave %2d:", blk * 64 + 5, w);
                    for (int i = 0; i < 6; ++i) { printf(" %s %.0f", v.names[i], (double)o[i] / nsl); sum += (double)o[i] / nsl; }
                    printf(" | sum %.0f; pro %.0f loop %.0f epi %.0f\n", sum, (double)(o[6] & 0xffffffffu), (double)(o[7] >> 20), (double)(o[7] & 0xfffff));
                }
        }
        CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(C)); CK(hipFree(T));
    }
    return 0;
}
