// Per-CU operand FILL rate on gfx950: how fast can ONE compute unit pull bytes, by transport and by the number of issuing waves?
// (round 3: the 512-row GEMMs of cfg 1 / cfg 5 run at ~50 GB/s per busy CU whatever the prefetch depth - DESIGN.md 9.1.)
//   transport 0: buffer_load_dwordx4 ... lds  (LDS-DMA, 1 KiB per wave-instruction into a 64 KiB LDS ring)
//   transport 1: buffer_load_dwordx4 -> VGPR  (plain vector load, 16 B per lane, results xor-ed into a sink register)
// Grid = `wgs` workgroups of W waves, each workgroup streams its own slice; `span` bytes per workgroup decide the source: a slice
// re-read many times from L2 (span 256 KiB) or streamed once from HBM (span = whole buffer / wgs).
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/fill_rate.hip -o /tmp/fill_rate && /tmp/fill_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int TRANSPORT, int DEPTH>
__global__ void fill(const char* src, long long span, int iters, unsigned* sink, int shared) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    // shared != 0: the 32 workgroups of an XCD (blockIdx % 8) share 8 slices (2 MiB per XCD: L2-resident with every CU busy)
    const long long slice = shared ? (blockIdx.x % 8) * 8 + (blockIdx.x / 8) % 8 : blockIdx.x;
    const char* base = src + slice * span;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(span > 0x7fffffffLL ? 0x7fffffffLL : span), 0x00020000);
    const int pieces = (int)(span / 1024);   // 1 KiB pieces in the slice
    u32x4 acc = {0, 0, 0, 0};
    int q = wave;                            // this wave's next piece
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int off = q * 1024;
            if (TRANSPORT == 0) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(smem + ((wave * DEPTH + d) & 63) * 1024), 16, lane * 16, off, 0, 0);
            } else {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, off, 0);
                acc ^= v;
            }
            q += nw;
            if (q >= pieces) q -= pieces;
        }
        if (TRANSPORT == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH / 2) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (TRANSPORT == 1 && (acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[threadIdx.x] = acc[0];
}

template <int TRANSPORT>
double run(const char* d, long long total, int wgs, int waves, long long span, unsigned* sink, int shared) {
    const int DEPTH = 8;
    const long long bytes_per_wg = 64LL << 20;  // every workgroup moves 64 MiB in total
    const int iters = (int)(bytes_per_wg / (1024LL * DEPTH * waves));
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    auto k = fill<TRANSPORT, DEPTH>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL(k, dim3(wgs), dim3(64 * waves), 65536, 0, d, span, iters / 8, sink, shared);
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(wgs), dim3(64 * waves), 65536, 0, d, span, iters, sink, shared);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return (double)iters * DEPTH * waves * 1024.0 / (ms * 1e-3) / 1e9;  // GB/s per workgroup (= per CU at one workgroup per CU)
}

int main() {
    const long long total = 4LL << 30;
    char* d; unsigned* sink;
    hipMalloc(&d, total); hipMalloc(&sink, 4096);
    hipMemset(d, 1, total);
    printf("# GB/s per workgroup (one workgroup per CU: 64 KiB LDS each), 64 MiB moved per workgroup\n");
    printf("%-44s %8s %8s %8s %8s %8s\n", "case", "1 wave", "2", "4", "8", "16");
    struct { const char* name; int wgs; long long span; int shared; } cases[] = {
        {"L2-resident 256 KiB slice, 1 CU busy", 1, 256 << 10, 0}, {"L2-resident 256 KiB slice, 64 CUs busy", 64, 256 << 10, 0},
        {"L2-resident, 256 CUs share 2 MiB per XCD", 256, 256 << 10, 1},
        {"MALL-resident 256 KiB slices, 256 CUs busy", 256, 256 << 10, 0},
        {"HBM stream (16 MiB slices), 64 CUs busy", 64, 16 << 20, 0}, {"HBM stream (16 MiB slices), 256 CUs busy", 256, 16 << 20, 0}};
    for (int tr = 0; tr < 2; ++tr) {
        printf("%s\n", tr == 0 ? "-- buffer_load_dwordx4 ... lds (LDS-DMA)" : "-- buffer_load_dwordx4 -> VGPR");
        for (auto& c : cases) {
            printf("%-44s", c.name);
            for (int w : {1, 2, 4, 8, 16}) printf(" %8.1f", tr == 0 ? run<0>(d, total, c.wgs, w, c.span, sink, c.shared) : run<1>(d, total, c.wgs, w, c.span, sink, c.shared));
            printf("\n");
            fflush(stdout);
        }
    }
    return 0;
}
