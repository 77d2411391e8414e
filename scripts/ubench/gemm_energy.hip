// Microbenchmark (GPU box): what does each ingredient of a GEMM main loop cost under the MI355X power-managed clock?
// the engine's GEMMs run at ~1150-1220 TFLOP/s whatever their schedule, the vendor's 4-wave kernel at 1490 on the
// same problem, and a register-only MFMA loop tops out at ~1600 (random operands).  This probe starts from that register-only loop
// and adds the ingredients one at a time - fragment reads from LDS, LDS-DMA refills, VGPR-staged refills (buffer_load + ds_write),
// a barrier per slab, idle gaps - in the two wave shapes the GEMMs use:
//     NT = 4: one wave per SIMD, 4x4 MFMA tiles (128x128 per wave, 256 accumulator registers)      -> variants 10 / 12, vendor
//     NT = 2: two waves per SIMD, 4x2 MFMA tiles (128x64 per wave, 128 accumulator registers)      -> the 8-wave kernels
// and prints TFLOP/s (wall clock) plus s_memtime ticks per k-step (relative duty).  No result is checked: operands are random
// bf16 in [-1, 1], addresses are masked into their buffers.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/gemm_energy.hip -o /tmp/gemm_energy && /tmp/gemm_energy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// F_HBM16: every 16th k-step streams its refill from a 1 GiB region no cache holds (the GEMMs' tile reuse through L2 leaves
//          about that share of their fetches to the fabric); F_TILES: a tile boundary every 144 k-steps (72 slabs): the 256
//          accumulator registers go out as bf16 (128 KiB per workgroup, to the big region), then the loop drains every
//          outstanding load and meets at a barrier - the epilogue + cold prologue of a real tile.
// F_STAGGER: the workgroups of an XCD start in eight phases, an eighth of a tile (~10 k cycles) apart, so that the chip is never
//          in its tile boundary all at once.
enum { F_READS = 1, F_DMA = 2, F_STAGED = 4, F_BARRIER = 8, F_SHARE_A = 16, F_HBM16 = 32, F_TILES = 64, F_STAGGER = 128 };
// idle gap per k-step (SLEEP > 0), by kind: 0 s_sleep SLEEP (wave descheduled), 1 s_nop spin of the same length (wave issuing),
// 2 s_waitcnt vmcnt(0) on a fresh load from the big region (wave parked on a counter), 3 waves 1-3 parked at s_barrier while
// wave 0 sleeps
enum { IDLE_SLEEP = 0, IDLE_NOP = 1, IDLE_VMCNT = 2, IDLE_BARRIER = 3 };

constexpr int FRAG_BYTES = 32768, LAND_BYTES = 16384, SMEM_BYTES = FRAG_BYTES + LAND_BYTES;
constexpr int GBYTES = 8 << 20;       // streamed global region (L2 / MALL resident)
constexpr unsigned BIGBYTES = 1u << 30;  // the region nothing caches; also receives the tile stores

template <int NT, int FLAGS, int SLEEP, int IDLE = IDLE_SLEEP>
__global__ __launch_bounds__(256, NT == 2 ? 2 : 1) void probe(const unsigned short* __restrict__ g, float* sink, unsigned long long* cyc, int iters,
                                                              unsigned short* __restrict__ big) {
    constexpr int MT = 4, NM = MT * NT, NR = MT + NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < FRAG_BYTES / 16; i += 256)
        ((u32x4*)smem)[i] = ((const u32x4*)g)[(blockIdx.x * 2048 + i) & (GBYTES / 16 - 1)];
    __syncthreads();
    bf16x8 a[2][MT], b[2][NT];
#pragma unroll
    for (int j = 0; j < MT; ++j) a[0][j] = a[1][j] = *(const bf16x8*)(smem + j * 1024 + lane * 16);
#pragma unroll
    for (int j = 0; j < NT; ++j) b[0][j] = b[1][j] = *(const bf16x8*)(smem + (MT + j) * 1024 + lane * 16);
    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // ONE allocation and descriptor: [ 8 MiB cached region | 1 GiB big region ] - the region is chosen through the scalar offset,
    // so the MFMA stream stays one basic block
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, GBYTES + BIGBYTES, 0x00020000);
    const unsigned gwave = (unsigned)(blockIdx.x * 4 + wave);  // chip-wide wave id (< 2048)
    const int voff = lane * 16;
    char* land = smem + FRAG_BYTES + wave * 4096;
    u32x4 st[2][4];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) st[c][q] = u32x4{0u, 0u, 0u, 0u};

    // first = the k-step that opens a tile: its MFMAs take the constant 0 as accumulator input (as a real tile's first k-step)
    auto step = [&](int it, auto cur, auto first) __attribute__((always_inline)) {
        constexpr int c = decltype(cur)::value, n = c ^ 1;
        constexpr bool FIRST = decltype(first)::value;
        const char* sb = smem + (it & 3) * 8192;
        // streamed bytes: every wave of the chip walks its own 4 KiB window per step through the 8 MiB region
        // big region: every wave owns 512 KiB and walks it 4 KiB per visit (a visit every 16 k-steps: wraps after 2048 k-steps)
        const bool hbm = (FLAGS & F_HBM16) != 0 && (it & 15) == 0;
        const unsigned off_small = (gwave * 65536u + (unsigned)it * 4096u) & (GBYTES - 1);
        const unsigned off_big = (unsigned)GBYTES + gwave * 524288u + (((unsigned)it >> 4) * 4096u & 524287u);
        const int soff = __builtin_amdgcn_readfirstlane((int)(hbm ? off_big : off_small));
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            const int mt = (FLAGS & F_SHARE_A) ? i % MT : i / NT, nt = (FLAGS & F_SHARE_A) ? i / MT : i % NT;
            if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(acc[mt][nt]) : "v"(b[c][nt]), "v"(a[c][mt]));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[mt][nt]) : "v"(b[c][nt]), "v"(a[c][mt]));
            if constexpr ((FLAGS & F_READS) != 0) {
                if (i < NR) {
                    if (i < NT) b[n][i] = *(const bf16x8*)(sb + (MT + i) * 1024 + lane * 16);
                    else a[n][i - NT] = *(const bf16x8*)(sb + (i - NT) * 1024 + lane * 16);
                }
            }
            if (i % 4 == 3) {  // one 1 KiB refill piece per four MFMAs, as in the GEMMs
                const int q = (i / 4) & 3;
                if constexpr ((FLAGS & F_DMA) != 0)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(land + q * 1024), 16, voff, soff + q * 1024, 0, 0);
                if constexpr ((FLAGS & F_STAGED) != 0) {
                    *(u32x4*)(land + q * 1024 + lane * 16) = st[c][q];  // loaded two steps ago
                    st[c][q] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff + q * 1024, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr ((FLAGS & F_BARRIER) != 0) {
            if (c == 1) {  // one barrier per two k-steps (= one 32-deep slab)
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_s_barrier();
            }
        }
        if constexpr (SLEEP > 0) {
            if constexpr (IDLE == IDLE_SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
            if constexpr (IDLE == IDLE_NOP) {
#pragma unroll
                for (int q = 0; q < 4 * SLEEP; ++q) asm volatile("s_nop 15");  // 64 cycles per unit, like s_sleep
            }
            if constexpr (IDLE == IDLE_VMCNT) {
                const int poff = __builtin_amdgcn_readfirstlane((int)((unsigned)GBYTES + ((gwave * 524288u + 262144u + (unsigned)it * 256u) & (BIGBYTES - 1))));
                st[0][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff & 255, poff, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::"v"(st[0][0]) : "memory");
            }
            if constexpr (IDLE == IDLE_BARRIER) {
                if (wave == 0) __builtin_amdgcn_s_sleep(SLEEP);
                __builtin_amdgcn_s_barrier();
            }
        }
    };
    // tile boundary: epilogue stores (16 fp32 -> 16 bf16 = two 16-byte stores per lane and MFMA tile), then the cold start of
    // the next tile (every outstanding load drained, barrier)
    auto boundary = [&](unsigned tile) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    u32x4 pk;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned lo = __builtin_bit_cast(unsigned short, (__bf16)acc[i][j][h * 8 + 2 * e]);
                        const unsigned hi = __builtin_bit_cast(unsigned short, (__bf16)acc[i][j][h * 8 + 2 * e + 1]);
                        pk[e] = lo | (hi << 16);
                    }
                    const int ooff = __builtin_amdgcn_readfirstlane((int)((unsigned)GBYTES + gwave * 524288u + (tile & 7u) * 65536u + (unsigned)((i * NT + j) * 2 + h) * 1024u));
                    __builtin_amdgcn_raw_buffer_store_b128(pk, rs, voff, ooff, 0);
                }
                __builtin_amdgcn_sched_barrier(0);  // one MFMA tile at a time
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    constexpr bool TILES = (FLAGS & F_TILES) != 0;
    const int ntile = TILES ? iters / 144 : 1, per = TILES ? 144 : iters;
    if constexpr ((FLAGS & F_STAGGER) != 0) {
        const int reps = ((blockIdx.x >> 3) & 7) * 10;
        for (int q = 0; q < reps; ++q) __builtin_amdgcn_s_sleep(16);
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    int it = 0;
    for (int t = 0; t < ntile; ++t) {
        const int e = it + per;
        if constexpr (TILES) {
            step(it, std::integral_constant<int, 0>{}, std::true_type{});
            step(it + 1, std::integral_constant<int, 1>{}, std::false_type{});
            it += 2;
        }
        for (; it < e; it += 2) {
            step(it, std::integral_constant<int, 0>{}, std::false_type{});
            step(it + 1, std::integral_constant<int, 1>{}, std::false_type{});
        }
        if constexpr (TILES) boundary((unsigned)t);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) r += acc[i][j][k];
    r += (float)st[0][0][0] + (float)st[1][3][1] + (float)smem[FRAG_BYTES + (tid & 1023)];
    sink[blockIdx.x * 256 + tid] = r;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

struct Case {
    const char* name;
    int nt, wps;  // MFMA columns per wave, waves per SIMD (workgroups per CU)
    void (*fn)(const unsigned short*, float*, unsigned long long*, int, unsigned short*);
};
#define CASE(label, NT, WPS, FLAGS, SLEEP) Case{label, NT, WPS, probe<NT, FLAGS, SLEEP>}
#define CASEI(label, NT, WPS, FLAGS, SLEEP, IDLE) Case{label, NT, WPS, probe<NT, FLAGS, SLEEP, IDLE>}

int main(int argc, char** argv) {
    int cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) == hipSuccess) cus = prop.multiProcessorCount;
    const int iters = argc > 1 ? atoi(argv[1]) / 288 * 288 : 8064, reps = 3;  // whole tiles of 144 k-steps, even
    unsigned short *g, *big;
    unsigned long long* cyc;
    float* sink;
    hipMalloc(&g, (size_t)GBYTES + BIGBYTES);
    big = g + GBYTES / 2;
    hipMemset(big, 0x3c, BIGBYTES);
    hipMalloc(&cyc, cus * 2 * 8);
    hipMalloc(&sink, cus * 2 * 256 * 4);
    std::vector<unsigned short> h(GBYTES / 2);
    unsigned s = 12345u;
    for (auto& v : h) {  // bf16 in [-1, 1]: random sign, exponent 120..126, random mantissa
        s = s * 1664525u + 1013904223u;
        v = (unsigned short)(((s >> 31) << 15) | ((120u + ((s >> 8) % 7u)) << 7) | ((s >> 16) & 0x7fu));
    }
    hipMemcpy(g, h.data(), GBYTES, hipMemcpyHostToDevice);
    const Case cases[] = {
        CASE("mfma only", 4, 1, 0, 0),
        CASE("mfma only", 2, 2, 0, 0),
        CASE("mfma only", 2, 1, 0, 0),
        CASE("mfma only, A-major order", 4, 1, F_SHARE_A, 0),
        CASE("+ fragment reads", 4, 1, F_READS, 0),
        CASE("+ fragment reads", 2, 2, F_READS, 0),
        CASE("+ reads + LDS-DMA", 4, 1, F_READS | F_DMA, 0),
        CASE("+ reads + LDS-DMA", 2, 2, F_READS | F_DMA, 0),
        CASE("+ reads + VGPR-staged", 4, 1, F_READS | F_STAGED, 0),
        CASE("+ reads + VGPR-staged", 2, 2, F_READS | F_STAGED, 0),
        CASE("+ reads + LDS-DMA + barrier", 4, 1, F_READS | F_DMA | F_BARRIER, 0),
        CASE("+ reads + LDS-DMA + barrier", 2, 2, F_READS | F_DMA | F_BARRIER, 0),
        CASE("+ reads + VGPR-staged + barrier", 4, 1, F_READS | F_STAGED | F_BARRIER, 0),
        CASE("LDS-DMA only (no reads)", 4, 1, F_DMA, 0),
        CASE("mfma only, s_sleep 1 per k-step", 4, 1, 0, 1),
        CASE("mfma only, s_sleep 2 per k-step", 4, 1, 0, 2),
        CASE("mfma only, s_sleep 4 per k-step", 4, 1, 0, 4),
        CASE("mfma only, s_sleep 2 per k-step", 2, 2, 0, 2),
        // what an idle wave costs, by the way it idles (same gap length as s_sleep 4 where the kind allows)
        CASEI("mfma only, s_nop spin 4 per k-step", 4, 1, 0, 4, IDLE_NOP),
        CASEI("mfma only, vmcnt stall per k-step", 4, 1, 0, 1, IDLE_VMCNT),
        CASEI("mfma only, barrier park 4 per k-step", 4, 1, 0, 4, IDLE_BARRIER),
        // towards the real kernel: fabric traffic and tile boundaries on top of the full loop
        CASE("full loop + 1/16 of refills from HBM", 4, 1, F_READS | F_DMA | F_BARRIER | F_HBM16, 0),
        CASE("full loop + tile boundaries", 4, 1, F_READS | F_DMA | F_BARRIER | F_TILES, 0),
        CASE("full loop + HBM + tile boundaries", 4, 1, F_READS | F_DMA | F_BARRIER | F_HBM16 | F_TILES, 0),
        CASE("full loop + HBM + tile boundaries", 2, 2, F_READS | F_DMA | F_BARRIER | F_HBM16 | F_TILES, 0),
        CASE("full loop + HBM + tiles, s_sleep 2", 4, 1, F_READS | F_DMA | F_BARRIER | F_HBM16 | F_TILES, 2),
        CASE("full loop + HBM + tiles, staggered", 4, 1, F_READS | F_DMA | F_BARRIER | F_HBM16 | F_TILES | F_STAGGER, 0),
    };
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("%-36s %3s %4s %10s %12s %10s %8s\n", "loop", "NT", "w/SD", "TFLOP/s", "ticks/kstep", "mfma cyc", "ms");
    for (const Case& c : cases) {
        const int blocks = cus * c.wps;
        hipFuncSetAttribute((const void*)c.fn, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
        float ms = 0.f;
        double avg = 0;
        for (int rep = 0; rep < reps; ++rep) {  // the last repetition is reported (clocks settled)
            hipEventRecord(e0);
            hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(256), SMEM_BYTES, 0, g, sink, cyc, iters, big);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        if (hipGetLastError() != hipSuccess) { printf("%-36s launch failed\n", c.name); continue; }
        std::vector<unsigned long long> t(blocks);
        hipMemcpy(t.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        for (auto v : t) avg += (double)v;
        avg /= blocks;
        const double flops = 2.0 * 32 * 32 * 16 * 4.0 * c.nt * (double)iters * blocks * 4;
        printf("%-36s %3d %4d %10.0f %12.1f %10d %8.3f\n", c.name, c.nt, c.wps, flops / (ms * 1e-3) / 1e12, avg / iters, 4 * c.nt * 32 * c.wps, ms);
    }
    return 0;
}
