// EXPERIMENTAL GEMM variants (round-1 A/B record; none is on the product path): persistent 8-wave ping-pong (variant 9),
// AGPR accumulators (11), 4-wave VGPR-staged (12), 4 x (128 x 128) per-tile form (10), ping-pong tail overlap, trace builds.
// Built only with `make EXPERIMENTAL=1` (-DLT_EXPERIMENTAL); measurements in profiles/r01/opbench_gemm_*.log and DESIGN.md 5.1.
#include "../gemm_device.h"
#include "gemm_w4p.h"
#include "gemm_small_m.h"

namespace lt_gemm {

// ---- persistent ping-pong kernel -------------------------------------------------------------------------
// The two-barrier ping-pong loop of gemm_bf16_pp (MODE 0, 32-deep slabs, 8 waves, 256x256 tile), but one workgroup per CU
// walks SEVERAL tiles and the 4-slot LDS ring never restarts: the LDS-DMA of the next tile's first three slabs is issued during
// the last three MFMA segments of the current tile, so the ~4.6 k-cycle cold prologue (3 slabs of DMA latency with an idle
// matrix pipe) is paid once per workgroup instead of once per tile, and each wave group's epilogue (pack + 8-16 global stores
// per wave, no barriers) overlaps the OTHER group's last / first MFMA segment.  For GEMMs with several tile rounds per CU
// (SwiGLU: 6) the per-tile prologue + epilogue was ~10 % of the kernel (profiles/r01/gemm_pingpong_cycle_trace.log).
// MEASURED (profiles/r01/opbench_gemm_persistent.log): 398.4 us vs 400.2 us on the SwiGLU GEMM - no gain.  Removing idle
// time from a kernel that already runs against the power-managed clock buys nothing; kept as variant 9 /
// option "gemm_persist" (parity-tested), not the default.
// vmcnt bookkeeping: loads and stores retire in issue order on gfx9 (one counter, the compiler relies on it too), so after an
// epilogue the NST stores of this wave sit between the prefetched slabs and the new tile's own LDS-DMA; the two READ segments
// that follow allow NST more outstanding operations.
template <int N>
__device__ __forceinline__ void wait_vmcnt_n() { wait_vmcnt<N>(); }

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_bf16_pp_persist(GemmArgs p) {
    constexpr int WM = 2, WN = 4, MT = 4, NT = 2, NW = 8;
    constexpr int BM = 256, BN = 256;
    constexpr int PA = BM / 16, NP = (BM + BN) / 16, IP = NP / NW;  // 32 pieces of 1 KiB per slab, 4 per wave
    constexpr int SLAB = (BM + BN) * 64, W_OFF = BM * 64;
    constexpr int NM = 2 * MT * NT;
    constexpr int NST = EPI == 0 ? MT * NT * 2 : MT * (NT / 2) * 2;  // global stores per wave and tile (store_tile)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int wm = wave / WN, wn = wave % WN;
    const int hi = lane >> 5, l31 = lane & 31;
    const int TM = (p.M + BM - 1) / BM, TN = (p.N + BN - 1) / BN;
    const int ntiles = TM * TN;
    const int ns = p.K / 32;

    // staging: wave w copies pieces w, w + 8 (A rows 16 w.., 16 (w + 8)..) and w + 16, w + 24 (the same rows of W); the
    // per-lane offsets do not depend on the tile - a tile only changes the two buffer descriptors (base = the tile's first row,
    // num_records = bytes left => rows past M / N read as zero)
    const int sswz = ((lane & 3) ^ ((lane >> 4) & 3)) * 16;
    const int srow = 16 * wave + (lane >> 2);
    int voff[IP], ldsoff[IP];
    voff[0] = srow * p.lda * 2 + sswz;
    voff[1] = (srow + 128) * p.lda * 2 + sswz;
    voff[2] = srow * p.ldw * 2 + sswz;
    voff[3] = (srow + 128) * p.ldw * 2 + sswz;
#pragma unroll
    for (int i = 0; i < IP; ++i) ldsoff[i] = (wave + NW * i) * 1024;
    static_assert(IP == 4 && PA == 16, "piece map above");
    auto setup = [&](int v, __amdgpu_buffer_rsrc_t& rA, __amdgpu_buffer_rsrc_t& rW, int& m0, int& n0) __attribute__((always_inline)) {
        int tm, tn;
        tile_coords(v, ntiles, TM, TN, tm, tn);
        m0 = tm * BM; n0 = tn * BN;
        const long long a_left = (long long)(p.M - m0) * p.lda * 2;
        const long long w_left = (long long)(p.N - n0) * p.ldw * 2;
        rA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (size_t)m0 * p.lda), 0, (int)(a_left > 0x7fffffffLL ? 0x7fffffffLL : a_left), 0x00020000);
        rW = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (size_t)n0 * p.ldw), 0, (int)(w_left > 0x7fffffffLL ? 0x7fffffffLL : w_left), 0x00020000);
    };
    __amdgpu_buffer_rsrc_t rAC, rWC, rAN, rWN;
    int m0 = 0, n0 = 0, m0n = 0, n0n = 0;

    const int fswz = (l31 >> 2) & 3;
    const int a_row_off = (wm * MT * 32 + l31) * 64;
    const int w_row_off = W_OFF + (wn * NT * 32 + l31) * 64;
    int coff[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) coff[k] = ((2 * k + hi) ^ fswz) << 4;

    f32x16 acc[MT][NT];
    bf16x8 wf[2][NT], af[2][MT];

    int v = blockIdx.x;
    if (v >= ntiles) return;  // uniform
    const int my_tiles = (ntiles - 1 - v) / gridDim.x + 1;
    const int total = my_tiles * ns;  // slabs this workgroup consumes
    setup(v, rAC, rWC, m0, n0);
    bool has_next = v + (int)gridDim.x < ntiles;
    // no next tile: the last three segments still issue their LDS-DMA (one code path), from empty descriptors - every lane is
    // out of range, the ring slots they zero-fill hold slabs that were consumed already
    const __amdgpu_buffer_rsrc_t r_null = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0, 0x00020000);
    if (has_next) setup(v + gridDim.x, rAN, rWN, m0n, n0n);
    else { rAN = r_null; rWN = r_null; }

    auto stage_from = [&](int g, int slab_in_tile, __amdgpu_buffer_rsrc_t rA, __amdgpu_buffer_rsrc_t rW) __attribute__((always_inline)) {
        char* base = smem + (g & 3) * SLAB;
        const int soff = slab_in_tile * 64;
#pragma unroll
        for (int i = 0; i < IP; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(i < 2 ? rA : rW, LDS_PTR(base + ldsoff[i]), 16, voff[i], soff, 0, 0);
    };
    // prologue (once per workgroup): global slabs 0..2 = this tile's slabs 0..2 (ns >= 3 is required by the launcher)
    stage_from(0, 0, rAC, rWC);
    stage_from(1, 1, rAC, rWC);
    stage_from(2, 2, rAC, rWC);
    wait_vmcnt_n<2 * IP>();
    pp_barrier();
    for (int g_ = 0; g_ < grp; ++g_) pp_barrier();

    int g = 0;                // global slab index of this wave's stream
    auto read_seg = [&]() __attribute__((always_inline)) {
        const char* sb = smem + (g & 3) * SLAB;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wf[k][nt] = *(const bf16x8*)(sb + w_row_off + nt * 2048 + coff[k]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[k][mt] = *(const bf16x8*)(sb + a_row_off + mt * 2048 + coff[k]);
        }
        // slab g+1 landed (slab g+2 - and, right after an epilogue, this wave's NST stores - may still be outstanding)
        // (the null-descriptor DMAs of the last tile count like real ones, so the counts are uniform to the very end)
        // (right after an epilogue this also waits for the wave's own stores: store_tile's `if (m < M && col < N)` may skip a
        //  store instruction on an edge tile, so a literal that lets "NST stores" pass could let a DMA pass instead)
        wait_vmcnt_n<IP>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        pp_barrier();
    };
    auto mfma_all = [&](auto do_stage, int slab_in_tile, __amdgpu_buffer_rsrc_t rA, __amdgpu_buffer_rsrc_t rW) __attribute__((always_inline)) {
        __builtin_amdgcn_s_setprio(1);
        constexpr int EVERY = NM / IP;
        char* base = smem + ((g + 3) & 3) * SLAB;
        const int soff = slab_in_tile * 64;
        int issued = 0;
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            const int k = i / (MT * NT), mt = (i / NT) % MT, nt = i % NT;
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[k][nt], af[k][mt], acc[mt][nt], 0, 0, 0);
            if constexpr (decltype(do_stage)::value) {
                if ((i + 1) % EVERY == 0 && issued < IP) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(issued < 2 ? rA : rW, LDS_PTR(base + ldsoff[issued]), 16, voff[issued], soff, 0, 0);
                    ++issued;
                }
            }
        }
        if constexpr (decltype(do_stage)::value) {
#pragma unroll
            for (int i = 0; i < IP; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x8, EVERY, 0);
                __builtin_amdgcn_sched_group_barrier(0x10, 1, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto handover = [&]() __attribute__((always_inline)) {
        pp_barrier();
        __builtin_amdgcn_s_setprio(0);
    };

    for (int t = 0; t < my_tiles; ++t) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        int s = 0;
        for (; s + 3 < ns; ++s) {  // LDS-DMA of this tile's slab s+3
            read_seg();
            mfma_all(std::true_type{}, s + 3, rAC, rWC);
            handover();
            ++g;
        }
        for (; s < ns; ++s) {      // last three segments: the next tile's slabs 0..2 (if there is a next tile)
            read_seg();
            mfma_all(std::true_type{}, s + 3 - ns, rAN, rWN);
            if (g + 1 < total) handover();
            else __builtin_amdgcn_s_setprio(0);
            ++g;
        }
        store_tile<MT, NT, EPI>(acc, p, m0, n0, wm, wn, hi, l31);
        if (has_next) {
            rAC = rAN; rWC = rWN;
            m0 = m0n; n0 = n0n;
            v += gridDim.x;
            has_next = v + (int)gridDim.x < ntiles;
            if (has_next) setup(v + gridDim.x, rAN, rWN, m0n, n0n);
            else { rAN = r_null; rWN = r_null; }
        }
    }
    wait_vmcnt_n<0>();  // no LDS-DMA (the null ones of the last segments included) may outlive the workgroup's LDS allocation
    for (int g_ = grp; g_ < 1; ++g_) pp_barrier();  // group 0 passes the barrier group 1 still executes after its last READ
}

// ---- 4-wave kernel with VGPR staging (EXPERIMENTAL, variant 12 - written at the end of round 1; first hardware run:
//      bit-identical to variant 1 on 6 shapes, 9 % SLOWER than the 8-wave ping-pong: profiles/r01/opbench_gemm_vgpr_staged.log) ---
// Why: the PMC comparison with the vendor library shows that kernels with 4 waves per workgroup run ~25 % higher
// clocks than the 8 / 12-wave kernels at the same MFMA work, and that our 4-wave loop (variant 10) loses that again to a 58 %
// duty cycle - with one wave per SIMD every slow-issuing instruction is a matrix-pipe bubble, and a `buffer_load ... lds` costs
// its wave 60-180 cycles, eight times per slab.  This kernel keeps the 2x2 waves of 128x128 (32 MFMAs per 32-deep slab, AGPR
// accumulators) but stages global -> VGPR -> LDS with plain buffer loads (cheap to issue) and ds_write_b128:
//   * three staging register sets: the loads of slab s+4 are issued right after slab s+1 left its set for the LDS, i.e. about
//     2.5 slabs (~2500 cycles) before they are needed;
//   * two LDS buffers of 32 KiB; ONE barrier per slab, in the middle of it:
//       H0(k): MFMAs of k-step 0 | fragment reads of (k, k-step 1) | wait for slab k+1's loads, ds_write it to buffer (k+1)&1
//              lgkmcnt(0), s_barrier  -> slab k+1 visible, every wave done with buffer (k+1)&1's previous content (slab k-1)
//       H1(k): MFMAs of k-step 1 | fragment reads of (k+1, k-step 0) | buffer loads of slab k+4
//     so the fragments of the next k-step are always read while the current one multiplies, the barrier's stall is the only
//     bubble, and the LDS write of a slab never races a read of the same buffer:
//       WAR  buffer (k+1)&1 held slab k-1, read in H1(k-2) and H0(k-1); every wave waited lgkmcnt(0) before B(k-1).
//       RAW  fragments of (k+1, 0) are read in H1(k), after B(k), which follows every wave's ds_write of slab k+1.
//   * LDS image, swizzle and fragment reads are those of gemm_bf16_pp (64-byte rows, chunk c of row r at position c ^ ((r>>2)&3)).
// TRACE build (lt_op_gemm_trace, variant 12): s_memtime stamps T0 | 8 MFMA + 8 fragment reads | Tv0 | vmcnt wait | Tv1 | 8 MFMA +
// 8 ds_write | T1 | lgkmcnt(0) | T2 | barrier | T3 | H1 | next T0.  The stamps are inline assembly (invisible to the compiler's
// waitcnt pass, so its counted LDS waits stay as in the product build) and are only consumed behind this kernel's own
// lgkmcnt(0): those of the first half of a slab right after the barrier, those behind it one slab later.
template <int EPI, bool TRACE = false>
__global__ __launch_bounds__(256, 1) void gemm_bf16_w4s(GemmArgs p) {
    constexpr int MT = 4, NT = 4, BM = 256, BN = 256;
    unsigned long long t_entry = 0;
    if constexpr (TRACE) t_entry = __builtin_amdgcn_s_memtime();
    constexpr int BUF = (BM + BN) * 64, W_OFF = BM * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int hi = lane >> 5, l31 = lane & 31;
    const int TM = (p.M + BM - 1) / BM, TN = (p.N + BN - 1) / BN;
    int tm, tn;
    tile_coords(blockIdx.x, gridDim.x, TM, TN, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const long long a_left = (long long)(p.M - m0) * p.lda * 2;
    const long long w_left = (long long)(p.N - n0) * p.ldw * 2;
    // buffer descriptors as plain SGPR quads (base, stride 0, num_records = bytes left in the panel => rows past M / N read 0):
    // the loads below are inline assembly, so that the compiler's waitcnt pass does not see them - it guarded the ds_writes of
    // the staged data with vmcnt(0), i.e. it drained the two younger slabs every iteration; the waits are counted by hand
    auto make_desc = [](const void* base, long long left) __attribute__((always_inline)) {
        const unsigned long long a = (unsigned long long)base;
        u32x4 d;
        d[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
        d[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
        d[2] = __builtin_amdgcn_readfirstlane((unsigned)(left > 0x7fffffffLL ? 0x7fffffffLL : (left < 0 ? 0 : left)));
        d[3] = 0x00020000u;
        return d;
    };
    const u32x4 rA = make_desc(p.A + (size_t)m0 * p.lda, a_left);
    const u32x4 rW = make_desc(p.W + (size_t)n0 * p.ldw, w_left);

    // staging: thread t moves chunk (row = t/4 + 64 i, c = t%4) of A (i = 0..3) and of W (i = 0..3) of every slab
    const int srow = tid >> 2, sc = tid & 3;
    const int ga = srow * p.lda * 2 + sc * 16, gw = srow * p.ldw * 2 + sc * 16;   // byte offsets inside the tile panels
    const int ga_step = 64 * p.lda * 2, gw_step = 64 * p.ldw * 2;
    const int lds_w = srow * 64 + ((sc ^ ((srow >> 2) & 3)) << 4);                 // (row + 64 i) keeps (row >> 2) & 3
    u32x4 st[3][8];
    int gvo[8];  // per-lane byte offsets of the eight chunks inside the A / W panels
#pragma unroll
    for (int i = 0; i < 4; ++i) { gvo[i] = ga + i * ga_step; gvo[4 + i] = gw + i * gw_step; }
    auto gload1 = [&](int slab, u32x4 (&s)[8], int i) __attribute__((always_inline)) {
        const int soff = __builtin_amdgcn_readfirstlane(slab * 64);
        if (i < 4) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(s[i]) : "v"(gvo[i]), "s"(rA), "s"(soff) : "memory");
        else asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(s[i]) : "v"(gvo[i]), "s"(rW), "s"(soff) : "memory");
    };
    auto gload = [&](int slab, u32x4 (&s)[8]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) gload1(slab, s, i);
    };
    auto lwrite1 = [&](int slab, const u32x4 (&s)[8], int i) __attribute__((always_inline)) {
        char* b = smem + (slab & 1) * BUF + lds_w;
        if (i < 4) *(u32x4*)(b + i * 64 * 64) = s[i];
        else *(u32x4*)(b + W_OFF + (i - 4) * 64 * 64) = s[i];
    };
    auto lwrite = [&](int slab, const u32x4 (&s)[8]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) lwrite1(slab, s, i);
    };

    // fragment reads (as gemm_bf16_pp, KS = 2)
    const int fswz = (l31 >> 2) & 3;
    const int a_row_off = (wm * MT * 32 + l31) * 64;
    const int w_row_off = W_OFF + (wn * NT * 32 + l31) * 64;
    int coff[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) coff[k] = ((2 * k + hi) ^ fswz) << 4;
    bf16x8 wf[2][NT], af[2][MT];  // [k-step parity]
    auto fread1 = [&](int slab, int ks, int j) __attribute__((always_inline)) {  // j = 0..3: W fragments, 4..7: A fragments
        const char* sb = smem + (slab & 1) * BUF;
        if (j < NT) wf[ks][j] = *(const bf16x8*)(sb + w_row_off + j * 2048 + coff[ks]);
        else af[ks][j - NT] = *(const bf16x8*)(sb + a_row_off + (j - NT) * 2048 + coff[ks]);
    };
    auto fread = [&](int slab, int ks) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 8; ++j) fread1(slab, ks, j);
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // MFMA i (0..15) of k-step ks, in place on AGPR accumulators (inline assembly: the builtin form let the register
    // allocator rotate the 256 accumulator registers through copies under this kernel's pressure)
    auto mfma1 = [&](int ks, int i) __attribute__((always_inline)) {
        const int mt = i / NT, nt = i % NT;
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[mt][nt]) : "v"(wf[ks][nt]), "v"(af[ks][mt]));
    };
    auto fence = []() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };  // pins the written order

    const int ns = p.K / 32;
    unsigned tr[6] = {0, 0, 0, 0, 0, 0};              // TRACE: h0a, vm, h0b, lgkm, bar, h1 cycle totals of this wave
    unsigned long long s0 = 0, sv0 = 0, sv1 = 0, s1 = 0, s2 = 0, s3 = 0;
    unsigned q1 = 0;                                  // low word of the previous slab's T1
    auto mt = [](unsigned long long& t) __attribute__((always_inline)) {
        if constexpr (TRACE) asm volatile("s_memtime %0" : "=s"(t));
    };
    // prologue: slabs 0..2 on their way, slab 0 in LDS and visible, slab 3 requested, fragments (0, k-step 0) in registers
    gload(0, st[0]);
    if (ns > 1) gload(1, st[1]);
    if (ns > 2) gload(2, st[2]);
    if (ns > 2) wait_vmcnt<16>();
    else if (ns > 1) wait_vmcnt<8>();
    else wait_vmcnt<0>();
    lwrite(0, st[0]);
    if (ns > 3) gload(3, st[0]);
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    pp_barrier();
    fread(0, 0);

    // one slab; snext = the staging set that holds slab k+1 (written to the LDS here, refilled with slab k+4).
    // steady = std::true_type: slabs k+1 .. k+4 all exist (no branches in the stream)
    auto slab_step = [&](int k, u32x4 (&snext)[8], auto steady) __attribute__((always_inline)) {
        constexpr bool ST = decltype(steady)::value;
        // ---- H0(k): first 8 MFMAs beside the 8 fragment reads of (k, 1); last 8 beside the 8 ds_writes of slab k+1
        mt(s0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            mfma1(0, j);
            fread1(k, 1, j);
            fence();
        }
        mt(sv0);
        // slab k+1 has arrived: its loads are older than those of slabs k+2, k+3 (8 each); slab k+4 is requested in H1(k)
        if (ST || k + 3 < ns) wait_vmcnt<16>();
        else if (k + 2 < ns) wait_vmcnt<8>();
        else wait_vmcnt<0>();
        mt(sv1);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            mfma1(0, 8 + j);
            if (ST || k + 1 < ns) lwrite1(k + 1, snext, j);
            fence();
        }
        __builtin_amdgcn_s_setprio(0);
        mt(s1);
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): fragments (k, 1) in registers, this wave's ds_writes done
        fence();
        // TRACE: T0 .. T1 of this slab and T2, T3 of the previous one have all returned (they are older than the wait above)
        const unsigned c0 = (unsigned)s0, cv0 = (unsigned)sv0, cv1 = (unsigned)sv1, c1 = (unsigned)s1, c2 = (unsigned)s2, c3 = (unsigned)s3;
        mt(s2);
        pp_barrier();
        mt(s3);
        // ---- H1(k): 16 MFMAs beside the 8 fragment reads of (k+1, 0) and the 8 buffer loads of slab k+4
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            mfma1(1, j);
            if ((j & 1) == 0) { if (ST || k + 1 < ns) fread1(k + 1, 0, j >> 1); }
            else if (ST || k + 4 < ns) gload1(k + 4, snext, j >> 1);
            if constexpr (TRACE) {
                if (j == 1) {  // scalar bookkeeping under the MFMAs
                    tr[0] += cv0 - c0; tr[1] += cv1 - cv0; tr[2] += c1 - cv1;
                    tr[3] += c2 - q1; tr[4] += c3 - c2; tr[5] += c0 - c3;  // previous slab's second half (first slab: ~0)
                    q1 = c1;
                }
            }
            fence();
        }
        __builtin_amdgcn_s_setprio(0);
    };
    unsigned long long tstart = 0;
    if constexpr (TRACE) { tstart = __builtin_amdgcn_s_memtime(); s2 = tstart; s3 = tstart; q1 = (unsigned)tstart; }
    int k = 0;
    for (; k + 6 < ns; k += 3) {  // steady state: the last step of the triple (k + 2) still has slab k + 6 to request
        slab_step(k, st[1], std::true_type{});      // staging sets rotate with period 3: slab s lives in set s % 3
        slab_step(k + 1, st[2], std::true_type{});
        slab_step(k + 2, st[0], std::true_type{});
    }
    for (; k < ns; k += 3) {
        slab_step(k, st[1], std::false_type{});
        if (k + 1 < ns) slab_step(k + 1, st[2], std::false_type{});
        if (k + 2 < ns) slab_step(k + 2, st[0], std::false_type{});
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    unsigned long long t_loop_end = 0;
    if constexpr (TRACE) t_loop_end = __builtin_amdgcn_s_memtime();
    store_tile<MT, NT, EPI>(acc, p, m0, n0, wm, wn, hi, l31);
    if constexpr (TRACE) {  // same record as gemm_bf16_pp's trace build (scripts/gemm_trace.py)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        if (p.trace && lane == 0 && (blockIdx.x & 63) == 5) {
            unsigned long long* o = p.trace + ((size_t)(blockIdx.x >> 6) * 4 + wave) * 8;
#pragma unroll
            for (int i = 0; i < 6; ++i) o[i] = tr[i];
            o[6] = ((unsigned long long)ns << 32) | (unsigned)(tstart - t_entry);
            o[7] = ((t_loop_end - tstart) << 20) | ((t_end - t_loop_end) & 0xfffff);
        }
    }
}

// ---- experimental instantiations ----------------------------------------------------------------------------------------
template __global__ void gemm_bf16_pp<4, 3, 2, 3, 0>(GemmArgs);                        // 256 x 288 ping-pong (variant 4)
template __global__ void gemm_bf16_pp<2, 4, 4, 2, 0, true>(GemmArgs);                  // trace builds
template __global__ void gemm_bf16_pp<4, 3, 2, 3, 0, true>(GemmArgs);
template __global__ void gemm_bf16_pp<2, 2, 4, 4, 0, false, 0, 2, 2>(GemmArgs);        // variant 10: 4 waves x (128 x 128), one tile per workgroup
template __global__ void gemm_bf16_pp<2, 2, 4, 4, 1, false, 0, 2, 2>(GemmArgs);
template __global__ void gemm_bf16_pp<2, 2, 4, 4, 0, true, 0, 2, 2>(GemmArgs);
template __global__ void gemm_bf16_pp<2, 4, 4, 2, 0, false, 0, 0, 2, true>(GemmArgs);  // variant 11: AGPR accumulators
template __global__ void gemm_bf16_pp<2, 4, 4, 2, 1, false, 0, 0, 2, true>(GemmArgs);
template __global__ void gemm_bf16_w4s<0>(GemmArgs);                                   // variant 12
template __global__ void gemm_bf16_w4s<1>(GemmArgs);
template __global__ void gemm_bf16_w4s<0, true>(GemmArgs);
template __global__ void gemm_bf16_pp_persist<0>(GemmArgs);                            // variant 9
template __global__ void gemm_bf16_pp_persist<1>(GemmArgs);
template __global__ void gemm_bf16_pp<2, 4, 4, 2, 0, false, 0, 1>(GemmArgs);           // variants 5 / 6: single-barrier rendezvous
template __global__ void gemm_bf16_pp<2, 4, 4, 2, 0, true, 0, 1>(GemmArgs);
template __global__ void gemm_bf16_pp<2, 4, 4, 2, 1, false, 0, 1>(GemmArgs);
template __global__ void gemm_bf16_pp<4, 3, 2, 3, 0, false, 0, 1>(GemmArgs);

}  // namespace lt_gemm

namespace {
using namespace lt_gemm;

int exp_num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

template <typename Kern>
int launch_plain(Kern kern, int smem, dim3 grid, dim3 block, const GemmArgs& a, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    LT_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    if (ev0) hipExtLaunchKernelGGL(kern, grid, block, smem, stream, ev0, ev1, 0, a);
    else hipLaunchKernelGGL(kern, grid, block, smem, stream, a);
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}
}  // namespace

// variants 4 (256x288 ping-pong), 5 / 6 (rendezvous 256 / 288), 9 (persistent ping-pong), 10 (4 x (128 x 128)), 11 (AGPR), 12
// (VGPR-staged), and the s_memtime trace builds of variants 1 / 2 / 5 / 10 / 12 (a.trace != null)
int launch_gemm_experimental(const GemmArgs& a, int epilogue, int variant, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    const int t256 = ((a.M + 255) / 256) * ((a.N + 255) / 256), t288 = ((a.M + 255) / 256) * ((a.N + 287) / 288);
    constexpr int S1 = 4 * 512 * 64, S2 = 4 * 544 * 64;
    if (a.trace) {
        if (variant == 15 && epilogue == 1)
            return launch_plain(gemm_bf16_w4q<1, 8, true>, 4 * 512 * 64, dim3(std::min(t256, exp_num_cus())), dim3(256), a, stream, nullptr, nullptr);
        LT_REQUIRE(epilogue == 0, "gemm trace: plain epilogue");
        if (variant == 15 || variant == 16) {  // persistent 16x16x32 kernel: one record per workgroup
            const int nt = variant == 16 ? t288 : t256;
            const dim3 grid(std::min(nt, exp_num_cus()));
            return variant == 16 ? launch_plain(gemm_bf16_w4q<0, 9, true>, 4 * 544 * 64, grid, dim3(256), a, stream, nullptr, nullptr)
                                 : launch_plain(gemm_bf16_w4q<0, 8, true>, 4 * 512 * 64, grid, dim3(256), a, stream, nullptr, nullptr);
        }
        if (variant == 12) return launch_plain(gemm_bf16_w4s<0, true>, 2 * 512 * 64, dim3(t256), dim3(256), a, stream, nullptr, nullptr);
        if (variant == 10) return launch_plain(gemm_bf16_pp<2, 2, 4, 4, 0, true, 0, 2, 2>, S1, dim3(t256), dim3(256), a, stream, nullptr, nullptr);
        if (variant == 5) return launch_plain(gemm_bf16_pp<2, 4, 4, 2, 0, true, 0, 1>, S1, dim3(t256), dim3(512), a, stream, nullptr, nullptr);
        if (variant == 1 || variant == 3) return launch_plain(gemm_bf16_pp<2, 4, 4, 2, 0, true>, S1, dim3(t256), dim3(512), a, stream, nullptr, nullptr);
        if (variant == 2 || variant == 4) return launch_plain(gemm_bf16_pp<4, 3, 2, 3, 0, true>, S2, dim3(t288), dim3(768), a, stream, nullptr, nullptr);
        lt_set_error("gemm trace: variants 1 / 2 (ping-pong), 5, 10, 12 are instrumented");
        return 2;
    }
    LT_REQUIRE(epilogue == 0 || epilogue == 1, "experimental gemm variants: plain or SwiGLU epilogue");
    switch (variant) {
        case 4:
            LT_REQUIRE(epilogue == 0, "gemm variant 4: plain epilogue");
            return launch_plain(gemm_bf16_pp<4, 3, 2, 3, 0>, S2, dim3(t288), dim3(768), a, stream, ev0, ev1);
        case 5:
            return epilogue == 1 ? launch_plain(gemm_bf16_pp<2, 4, 4, 2, 1, false, 0, 1>, S1, dim3(t256), dim3(512), a, stream, ev0, ev1)
                                 : launch_plain(gemm_bf16_pp<2, 4, 4, 2, 0, false, 0, 1>, S1, dim3(t256), dim3(512), a, stream, ev0, ev1);
        case 6:
            LT_REQUIRE(epilogue == 0, "gemm variant 6: plain epilogue");
            return launch_plain(gemm_bf16_pp<4, 3, 2, 3, 0, false, 0, 1>, S2, dim3(t288), dim3(768), a, stream, ev0, ev1);
        case 9: {
            LT_REQUIRE(!a.tile_expert && a.K >= 96, "gemm variant 9: dense problems with K >= 96 only");
            const dim3 grid(std::min(t256, exp_num_cus()));
            return epilogue == 1 ? launch_plain(gemm_bf16_pp_persist<1>, S1, grid, dim3(512), a, stream, ev0, ev1)
                                 : launch_plain(gemm_bf16_pp_persist<0>, S1, grid, dim3(512), a, stream, ev0, ev1);
        }
        case 10:
            return epilogue == 1 ? launch_plain(gemm_bf16_pp<2, 2, 4, 4, 1, false, 0, 2, 2>, S1, dim3(t256), dim3(256), a, stream, ev0, ev1)
                                 : launch_plain(gemm_bf16_pp<2, 2, 4, 4, 0, false, 0, 2, 2>, S1, dim3(t256), dim3(256), a, stream, ev0, ev1);
        case 11:
            return epilogue == 1 ? launch_plain(gemm_bf16_pp<2, 4, 4, 2, 1, false, 0, 0, 2, true>, S1, dim3(t256), dim3(512), a, stream, ev0, ev1)
                                 : launch_plain(gemm_bf16_pp<2, 4, 4, 2, 0, false, 0, 0, 2, true>, S1, dim3(t256), dim3(512), a, stream, ev0, ev1);
        case 12:
            LT_REQUIRE(!a.tile_expert, "gemm variant 12: dense problems only");
            return epilogue == 1 ? launch_plain(gemm_bf16_w4s<1>, 2 * 512 * 64, dim3(t256), dim3(256), a, stream, ev0, ev1)
                                 : launch_plain(gemm_bf16_w4s<0>, 2 * 512 * 64, dim3(t256), dim3(256), a, stream, ev0, ev1);
        case 13:
        case 14: {  // persistent 4 waves x (128 x 128) on 32x32x16 MFMAs (14: a tile's epilogue rides in the next tile's first slab)
            LT_REQUIRE(!a.tile_expert && a.bias_dtype < 0 && a.K % 64 == 0 && a.K >= 128, "gemm variants 13 / 14: dense, no bias, K %% 64 == 0, K >= 128");
            const dim3 grid(std::min(t256, exp_num_cus()));
            if (variant == 14) return epilogue == 1 ? launch_plain(gemm_bf16_w4p<1, true>, S1, grid, dim3(256), a, stream, ev0, ev1)
                                                    : launch_plain(gemm_bf16_w4p<0, true>, S1, grid, dim3(256), a, stream, ev0, ev1);
            return epilogue == 1 ? launch_plain(gemm_bf16_w4p<1, false>, S1, grid, dim3(256), a, stream, ev0, ev1)
                                 : launch_plain(gemm_bf16_w4p<0, false>, S1, grid, dim3(256), a, stream, ev0, ev1);
        }
        case 17:
        case 18: {  // deep-ring small-M kernel (gemm_small_m.h): 128 x 128 (17, also SwiGLU) / 64 x 128 (18) tiles
            LT_REQUIRE(a.bias_dtype < 0 && a.K % 64 == 0 && a.K >= 32 * 13, "gemm variants 17 / 18: no bias, K %% 64 == 0, K >= 416");
            LT_REQUIRE(variant == 17 || epilogue == 0, "gemm variant 18: plain epilogue");
            const int t128 = ((a.M + 127) / 128) * ((a.N + 127) / 128), t64 = ((a.M + 63) / 64) * ((a.N + 127) / 128);
            if (variant == 18) return launch_plain(gemm_bf16_sm<0, 2, 4, 12>, 12 * 192 * 64, dim3(std::min(t64, exp_num_cus())), dim3(256), a, stream, ev0, ev1);
            return epilogue == 1 ? launch_plain(gemm_bf16_sm<1, 4, 4, 8>, 8 * 256 * 64, dim3(std::min(t128, exp_num_cus())), dim3(256), a, stream, ev0, ev1)
                                 : launch_plain(gemm_bf16_sm<0, 4, 4, 8>, 8 * 256 * 64, dim3(std::min(t128, exp_num_cus())), dim3(256), a, stream, ev0, ev1);
        }
        default:
            lt_set_error("gemm: variant %d is not an experimental kernel", variant);
            return 2;
    }
}
