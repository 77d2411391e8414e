// Device code of the bf16 MFMA GEMM family (kernel templates + shared epilogue); instantiated by gemm_bf16.hip (product
// kernels) and experimental/gemm_experimental.hip (A/B variants that lost or were never adopted, built with EXPERIMENTAL=1).
#pragma once
#include "common.h"
#include "kernels.h"
#include "tile_order.h"
#include <hip/hip_ext.h>
#include <algorithm>
#include <type_traits>

namespace {

constexpr int BK = 64;

__device__ __forceinline__ float load_bias(const void* bias, int dt, int n) {
    return dt == 0 ? ((const float*)bias)[n] : bf2f(((const u16*)bias)[n]);
}

}  // namespace

// (named namespace: a __global__ template with internal linkage that is only instantiated from another
//  template loses its host stub with hipcc 7.2)
namespace lt_gemm {

// ---- epilogue shared by both GEMM kernels: lane holds, per 32x32 tile, row m = l31 and columns 8q + 4hi + j (reg 4q+j) ----
template <int MT, int NT, int EPI>
__device__ __forceinline__ void store_tile(f32x16 (&acc)[MT][NT], const GemmArgs& p, int m0, int n0, int wm, int wn,
                                           int hi, int l31) {
    if constexpr (EPI == 2) {
        // V^T epilogue.  The MFMAs ran with swapped operands (D = A_frag x W_frag), so here a lane holds, per 32x32 tile, COLUMN
        // n = l31 and rows m = 8 q + 4 hi + j (reg 4 q + j).  Destination: vt[b][kv head][d][token'] (AttnArgs::vt) with the
        // keys of every group of 16 permuted (bits 2 and 3 of the position swapped - the order in which a lane of the swapped
        // QK^T MFMA holds its P values, see v_transpose in qkv_post.hip).  Rows 8 q + 4 hi + j of one 16-group (q = 2 g, 2 g + 1)
        // land on positions 8 hi + 4 (q & 1) + j: the lane's two register quads are 8 CONSECUTIVE permuted keys = one 16-byte
        // store, no cross-lane exchange.  Tokens per sample are a multiple of 64 (launcher), so a 32-row tile lies in one sample.
        const int ntok = p.vt_tokens, hd = p.vt_hd, kvh = p.N / hd;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = n0 + wn * NT * 32 + nt * 32 + l31;
            const int head = n / hd, d = n - head * hd;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int mtile = m0 + wm * MT * 32 + mt * 32;  // first row of the 32-row tile (wave-uniform)
                const int b = mtile / ntok, tok0 = mtile - b * ntok;
                u16* drow = p.C + (((size_t)b * kvh + head) * hd + d) * (size_t)p.vt_npad + tok0 + 8 * hi;
                if (mtile < p.M && n < p.N) {
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        const u32x4 o = {pack2bf_pk(acc[mt][nt][8 * g + 0], acc[mt][nt][8 * g + 1]),
                                         pack2bf_pk(acc[mt][nt][8 * g + 2], acc[mt][nt][8 * g + 3]),
                                         pack2bf_pk(acc[mt][nt][8 * g + 4], acc[mt][nt][8 * g + 5]),
                                         pack2bf_pk(acc[mt][nt][8 * g + 6], acc[mt][nt][8 * g + 7])};
                        *(u32x4*)(drow + 16 * g) = o;
                    }
                }
            }
        }
        return;
    }
    const size_t ldc = p.ldc;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = m0 + wm * MT * 32 + mt * 32 + l31;
        u16* crow = p.C + (size_t)m * ldc;
        if (EPI == 0) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int nbase = n0 + wn * NT * 32 + nt * 32;
#pragma unroll
                for (int qp = 0; qp < 2; ++qp) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = acc[mt][nt][8 * qp + j];
                    if (p.bias_dtype >= 0) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            int n = nbase + 16 * qp + 8 * (j >> 2) + 4 * hi + (j & 3);
                            n = n < p.N ? n : p.N - 1;  // clamped (branch-free); out-of-range columns are not stored
                            v[j] += load_bias(p.bias, p.bias_dtype, n);
                        }
                    }
                    unsigned ax = pack2bf_pk(v[0], v[1]), ay = pack2bf_pk(v[2], v[3]);
                    unsigned bx = pack2bf_pk(v[4], v[5]), by = pack2bf_pk(v[6], v[7]);
                    auto r0 = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
                    auto r1 = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
                    const int col = nbase + 16 * qp + 8 * hi;
                    if (m < p.M && col < p.N) {
                        u32x4 o = {r0[0], r1[0], r0[1], r1[1]};
                        *(u32x4*)(crow + col) = o;
                    }
                }
            }
        } else {
#pragma unroll
            for (int np = 0; np < NT / 2; ++np) {
                const int obase = (n0 + wn * NT * 32 + np * 64) / 2;
#pragma unroll
                for (int qp = 0; qp < 2; ++qp) {
                    unsigned w[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        // reference rounding points (model.py:497-502 under bf16): w1 x, w3 x, silu, product
                        const f32x2 a = {acc[mt][2 * np][8 * qp + 2 * j], acc[mt][2 * np][8 * qp + 2 * j + 1]};
                        const f32x2 b = {acc[mt][2 * np + 1][8 * qp + 2 * j], acc[mt][2 * np + 1][8 * qp + 2 * j + 1]};
                        w[j] = pk_bf(swiglu2(a, b));
                    }
                    unsigned ax = w[0], ay = w[1], bx = w[2], by = w[3];
                    auto r0 = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
                    auto r1 = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
                    const int col = obase + 16 * qp + 8 * hi;
                    if (m < p.M && col < p.N / 2) {
                        u32x4 o = {r0[0], r1[0], r0[1], r1[1]};
                        *(u32x4*)(crow + col) = o;
                    }
                }
            }
        }
    }
}

// WM x WN waves, each owning an (MT*32) x (NT*32) block of C.  Tile = (WM*MT*32) x (WN*NT*32) x 64.
//   <2,4,4,2>: 256 x 256, 8 waves  (128 accumulators / lane)  - default and the SwiGLU epilogue
//   <4,3,2,3>: 256 x 288, 12 waves ( 96 accumulators / lane)  - N = 2304 / 6912: 8192 x 2304 is exactly
//              256 tiles = one round of the 256 CUs (256-wide tiles need 288 = 1.125 rounds -> 2 rounds)
template <int WM, int WN, int MT, int NT, int EPI>
__global__ __launch_bounds__(WM* WN * 64, (WM * WN + 3) / 4) void gemm_bf16_tn(GemmArgs p) {
    constexpr int NW = WM * WN;
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    constexpr int PA = BM / 8, PW = BN / 8;           // 1-KiB staging pieces (8 rows x 128 B)
    constexpr int IA = (PA + NW - 1) / NW, IW = (PW + NW - 1) / NW;
    constexpr int W_OFF = BM * 128;
    constexpr int STAGE_BYTES = (BM + BN) * 128;
    static_assert(EPI != 1 || NT % 2 == 0, "SwiGLU epilogue pairs accumulator tiles");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int hi = lane >> 5, l31 = lane & 31;

    const int TM = (p.M + BM - 1) / BM, TN = (p.N + BN - 1) / BN;
    int tm, tn;
    tile_coords(blockIdx.x, gridDim.x, TM, TN, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const u16* Wg = p.W;
    if (p.tile_expert) {  // grouped mode: the 256-row segment this tile lies in belongs to one expert (or is padding)
        const int ex = p.tile_expert[(tm * BM) >> 8];
        if (ex < 0) return;
        Wg += (size_t)ex * p.w_expert_stride;
    }

    // descriptors based at the tile's first row; num_records = bytes left => rows past the end read 0
    const long long a_left = (long long)(p.M - m0) * p.lda * 2;
    const long long w_left = (long long)(p.N - n0) * p.ldw * 2;
    __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.A + (size_t)m0 * p.lda), 0, (int)(a_left > 0x7fffffffLL ? 0x7fffffffLL : a_left), 0x00020000);
    __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(Wg + (size_t)n0 * p.ldw), 0, (int)(w_left > 0x7fffffffLL ? 0x7fffffffLL : w_left), 0x00020000);

    // staging: wave w copies pieces w, w + NW, ... of the A tile and of the W tile.  Piece j holds rows
    // 8j..8j+7; the lane's 16-byte chunk c of row r is fetched from source chunk c ^ ((r >> 1) & 7).
    const int srow = wave * 8 + (lane >> 3);
    const int sswz = ((lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7)) * 16;  // NW is even: parity of j = parity of w
    static_assert(IA <= 4 && IW <= 4, "staging pieces per wave");
    int a_voff[4], w_voff[4];  // fixed size: a dependent bound here breaks host-side substitution (hipcc 7.2)
#pragma unroll
    for (int i = 0; i < IA; ++i) a_voff[i] = (srow + 8 * NW * i) * p.lda * 2 + sswz;
#pragma unroll
    for (int i = 0; i < IW; ++i) w_voff[i] = (srow + 8 * NW * i) * p.ldw * 2 + sswz;
    auto stage = [&](int buf, int kt) {
        const int soff = kt * BK * 2;
        char* base = smem + buf * STAGE_BYTES + wave * 1024;
#pragma unroll
        for (int i = 0; i < IA; ++i)
            if (wave + NW * i < PA)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, LDS_PTR(base + i * NW * 1024), 16, a_voff[i], soff, 0, 0);
#pragma unroll
        for (int i = 0; i < IW; ++i)
            if (wave + NW * i < PW)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, LDS_PTR(base + W_OFF + i * NW * 1024), 16, w_voff[i], soff, 0, 0);
    };

    // fragment read offsets (row ≡ l31 mod 32 in every sub-tile, so the swizzle key is per lane)
    const int fswz = (l31 >> 1) & 7;
    const int a_row_off = (wm * MT * 32 + l31) * 128;
    const int w_row_off = W_OFF + (wn * NT * 32 + l31) * 128;
    int coff[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) coff[s] = ((2 * s + hi) ^ fswz) << 4;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = p.K / BK;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* sb = smem + cur * STAGE_BYTES;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8 wf[NT], af[MT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wf[nt] = *(const bf16x8*)(sb + w_row_off + nt * 4096 + coff[s]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[mt] = *(const bf16x8*)(sb + a_row_off + mt * 4096 + coff[s]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = EPI == 2 ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt], wf[nt], acc[mt][nt], 0, 0, 0)   // rows = m
                                           : __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nt], af[mt], acc[mt][nt], 0, 0, 0);  // rows = n
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    store_tile<MT, NT, EPI>(acc, p, m0, n0, wm, wn, hi, l31);
}


// ---- ping-pong kernel ---------------------------------------------------------------------------------
// Same tile shapes and fragment/epilogue layout as gemm_bf16_tn, different time structure.  The workgroup's
// waves form G = NW/4 groups (group = wave / 4, i.e. the G waves that share one SIMD belong to G different
// groups).  K is consumed in 32-deep slabs held in a 4-slot LDS ring (64-byte rows, XOR swizzle on the two
// chunk-index bits).  Per slab every wave runs
//        READ  (fragment ds_reads of slab s, counted vmcnt for slab s+1, lgkmcnt(0))   | s_barrier
//        MFMA  (all MFMAs of slab s, with the LDS-DMA of slab s+3 issued between them)   | s_barrier  [+ G-2 idle]
// and group g starts g barrier intervals late, so on every SIMD exactly one wave is in its MFMA segment while
// the others read / wait: the matrix pipe sees back-to-back MFMA segments and no wave ever drains vmcnt to 0
// in the main loop (LDS-DMA stays in flight across barriers; guide T3/T4, "Pipelining across barriers").
//
// Hazards, in barrier-interval units (READ(s) of group g runs in interval G*s + g, MFMA(s) one later):
//   RAW  slab s+1 is waited for (each wave: its own pieces) in READ(s), interval G*s+g, and first read in
//        READ(s+1), interval G*s+G+g' > G*s+g for all g, g'  -> a barrier every wave has passed lies between.
//   WAR  slab s+4 reuses the slot of slab s; it is issued in MFMA(s+1), interval G*s+G+g+1, while the last
//        read of slab s completed (lgkmcnt(0) before the barrier) in interval G*s+g' <= G*s+G-1.
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void pp_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// TAILN: the barrier that hands the matrix pipe to the other group sits TAILN MFMAs before the end of the MFMA segment; those
// last MFMAs (k-step 1 fragments, registers only) are issued after the barrier INSIDE the group's next READ segment, between
// its k-step 0 fragment reads, so the barrier's release latency (~95 cycles) is covered by this group's MFMA work while the
// other group starts.  (First attempt, tail issued BEFORE the next READ: 5-8 % slower, profiles/r01/opbench_pp_tail_ab.log -
// READ then started TAILN MFMAs late and READ + tail, not the MFMA segment, set the barrier interval.)
// Measured (profiles/r01/opbench_gemm_pipelines.log): no gain either - a wave parked in s_barrier cannot issue, so the pipe
// still idles for the release latency, and the tail MFMAs simply come out of the other group's segment (505 instead of 416
// cycles for its 13 MFMAs).  Default stays TAILN = 0; option "gemm_pp_tail" keeps the A/B.
//
// MODE 1 ("rendezvous"): ONE barrier per slab.  Between two barriers group 0 runs MFMA(k) then READ(k+1), the other groups run
// READ(k) then MFMA(k): matrix work sits beside memory work in both halves of the interval without a hand-over barrier in the
// middle (an in-order wave whose MFMA finds the pipe busy simply waits for it), so the pipe idles for one barrier release per
// slab instead of one per segment.  Hazards (interval k = after barrier k): slab k+1 is read in interval k (group 0) or k+1
// (others) and every wave waited for its pieces of slab k+1 before barrier k; slab k+3 is issued in interval k into the slot
// of slab k-1, whose last reads (other groups, interval k-1) completed before barrier k.
// KS: MFMA k-steps per slab (2 = 32-deep slabs, 64-byte LDS rows; 4 = 64-deep, 128-byte rows).  The deep form halves the
// number of barrier intervals of a K loop; the small-M tiles use it, where an interval holds only 2-4 MFMAs per wave and the
// loop is barrier-latency bound (the 256-wide tiles cannot: 4 slots x 64 KiB exceed the LDS).
// AGPR: issue the MFMAs as inline assembly with the accumulators constrained to the AGPR file.  (The builtin lets the compiler
// use the unified-VGPR form whenever the kernel fits 256 registers, which the 8-wave kernels do; the vendor library's kernels
// keep their accumulators in AGPRs and run ~25 % faster clocks on the same problem - profiles/r01/vendor_vs_engine_pmc.log.)
// Experiment knob of the 4-wave kernels (variants 10, 13, 14; lt_set_option("gemm_stagger", n)): workgroup b sleeps
// ((b >> 3) & 7) * n * ~256 cycles before its first load, which spreads the CUs of an XCD over eight tile phases.  All tiles of
// a GEMM take the same time, so without it every CU of the chip is in its prologue / epilogue at the same moment; whether that
// synchronised idle phase is what keeps the clock low is one of the next round's questions.  A __device__ word
// instead of a GemmArgs field: the kernels of the product path do not read it and keep their argument layout.
__device__ __forceinline__ void stagger_start(int n) {
    if (n > 0) {
        const int reps = ((blockIdx.x >> 3) & 7) * n;
        for (int i = 0; i < reps; ++i) __builtin_amdgcn_s_sleep(4);
    }
}

template <int WM, int WN, int MT, int NT, int EPI, bool TRACE = false, int TAILN = 0, int MODE = 0, int KS = 2, bool AGPR = false>
__global__ __launch_bounds__(WM* WN * 64, (WM * WN + 3) / 4) void gemm_bf16_pp(GemmArgs p) {
    static_assert(MODE == 0 || TAILN == 0, "rendezvous mode has no hand-over barrier");
    static_assert(KS == 2 || KS == 4, "slab depth 32 or 64");
    static_assert(TAILN == 0 || KS == 2, "tail overlap is written for 32-deep slabs");
    constexpr int RB = KS * 32;          // bytes per LDS row
    constexpr int RPP = 1024 / RB;       // rows per 1-KiB staging piece
    constexpr int LPR = RB / 16;         // lanes (16-byte chunks) per row
    constexpr int NW = WM * WN, G = NW / 4;
    static_assert(NW % 4 == 0 && ((G >= 2 && G <= 3) || (MODE == 2 && G == 1)), "ping-pong needs 2 or 3 waves per SIMD");
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    constexpr int PA = BM / RPP, PW = BN / RPP, NP = PA + PW;  // 1-KiB pieces (RPP rows x RB bytes) per slab
    constexpr int IP = (NP + NW - 1) / NW;                     // pieces per wave per slab (same for every wave)
    constexpr int SLAB = (BM + BN) * RB, W_OFF = BM * RB;
    static_assert(EPI != 1 || NT % 2 == 0, "SwiGLU epilogue pairs accumulator tiles");
    static_assert(IP <= 8, "staging pieces per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long t_entry = 0;
    if constexpr (TRACE) t_entry = __builtin_amdgcn_s_memtime();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int wm = wave / WN, wn = wave % WN;
    const int hi = lane >> 5, l31 = lane & 31;

    const int TM = (p.M + BM - 1) / BM, TN = (p.N + BN - 1) / BN;
    int tm, tn;
    // split-K (GemmArgs::split_k = S = 2 or 4): workgroups [j tiles, (j + 1) tiles) take the j-th of S equal K ranges of the same tiles
    const int ntile = TM * TN;
    const int S = p.split_k >= 2 ? p.split_k : 1;
    const int ksplit = S > 1 ? (int)blockIdx.x / ntile : 0;
    const int bid = S > 1 ? (int)blockIdx.x - ksplit * ntile : (int)blockIdx.x;
    tile_coords(bid, S > 1 ? ntile : (int)gridDim.x, TM, TN, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    const long long a_left = (long long)(p.M - m0) * p.lda * 2;
    const long long w_left = (long long)(p.N - n0) * p.ldw * 2;
    const int a_bytes = (int)(a_left > 0x7fffffffLL ? 0x7fffffffLL : a_left);
    const int w_bytes = (int)(w_left > 0x7fffffffLL ? 0x7fffffffLL : w_left);
    const u16* Wg = p.W;
    if (p.tile_expert) {  // grouped mode (see GemmArgs; the table is per 256 rows); uniform exit before any barrier
        const int ex = p.tile_expert[(tm * BM) >> 8];
        if (ex < 0) return;
        Wg += (size_t)ex * p.w_expert_stride;
    }
    const u16* a_base = p.A + (size_t)m0 * p.lda;
    const u16* w_base = Wg + (size_t)n0 * p.ldw;

    // staging: wave w owns pieces w, w + NW, ... (a surplus slot re-loads the wave's previous piece: same bytes
    // to the same place, so every wave issues exactly IP loads per slab and one vmcnt literal fits all).
    // Piece q holds rows RPP q .. RPP q + RPP - 1 of A (q < PA) or of W; lane -> row RPP q + lane / LPR, 16-byte position
    // lane % LPR, fetched from source chunk pos ^ key(row): key = (row >> 2) & 3 for 64-byte rows, (row >> 1) & 7 for
    // 128-byte rows (the same keys the fragment reads apply, so a 32x32x16 fragment read is bank-conflict free).
    __amdgpu_buffer_rsrc_t rs[8];  // fixed size: a dependent bound here breaks host-side substitution (hipcc 7.2)
    int voff[8], ldsoff[8];
#pragma unroll
    for (int i = 0; i < IP; ++i) {
        int q = wave + NW * i;
        if (q >= NP) q -= NW;
        const bool isA = q < PA;
        const int r0 = RPP * (isA ? q : q - PA) + lane / LPR;
        const int key = KS == 2 ? (r0 >> 2) & 3 : (r0 >> 1) & 7;
        if (isA && p.a_row_map) {  // gather-on-load: this lane's LDS row r0 of the tile comes from row a_row_map[m0 + r0] of A (or is zero)
            const int src = m0 + r0 < p.M ? p.a_row_map[m0 + r0] : -1;
            rs[i] = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.a_map_rows * p.lda * 2, 0x00020000);
            voff[i] = (src >= 0 ? src * p.lda * 2 : 0x40000000) + (((lane % LPR) ^ key) << 4);
        } else {
            rs[i] = __builtin_amdgcn_make_buffer_rsrc((void*)(isA ? a_base : w_base), 0, isA ? a_bytes : w_bytes, 0x00020000);
            voff[i] = r0 * (isA ? p.lda : p.ldw) * 2 + (((lane % LPR) ^ key) << 4);
        }
        ldsoff[i] = q * 1024;
    }
    const int kbase = S > 1 ? ksplit * (p.K / S) * 2 : 0;  // byte offset along K of this workgroup's range (K / S elements)
    auto stage = [&](int slab) {
        char* base = smem + (slab & 3) * SLAB;
        const int soff = kbase + slab * RB;
#pragma unroll
        for (int i = 0; i < IP; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs[i], LDS_PTR(base + ldsoff[i]), 16, voff[i], soff, 0, 0);
    };

    const int fswz = KS == 2 ? (l31 >> 2) & 3 : (l31 >> 1) & 7;
    const int a_row_off = (wm * MT * 32 + l31) * RB;
    const int w_row_off = W_OFF + (wn * NT * 32 + l31) * RB;
    constexpr int TSTRIDE = 32 * RB;  // LDS bytes between two 32-row fragment tiles
    int coff[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) coff[s] = ((2 * s + hi) ^ fswz) << 4;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int ns = (p.K / S) / (16 * KS);
    if constexpr (MODE == 2 && G == 1) stagger_start(p.stagger);  // 4-wave kernel only (variant 10)
    // prologue: slabs 0..2 in flight, slab 0 landed and visible
    stage(0);
    if (ns > 1) stage(1);
    if (ns > 2) stage(2);
    if (ns > 2) wait_vmcnt<2 * IP>();
    else if (ns > 1) wait_vmcnt<IP>();
    else wait_vmcnt<0>();
    pp_barrier();
    if constexpr (MODE == 0)
        for (int g = 0; g < grp; ++g) pp_barrier();

    bf16x8 wf[KS][NT], af[KS][MT];
    // TRACE build only: per-wave cycle totals of the six sub-segments of a step (s_memtime stamps)
    unsigned long long tr[6] = {0, 0, 0, 0, 0, 0}, tprev = 0, ta = 0, tb = 0, tc = 0;
    unsigned long long tstart = 0;
    if constexpr (TRACE) { tprev = __builtin_amdgcn_s_memtime(); tstart = tprev; }
    constexpr int NM = KS * MT * NT;  // MFMAs of one segment
    static_assert(TAILN >= 0 && TAILN < MT * NT, "tail MFMAs must all belong to k-step 1");
    auto one_mfma = [&](int idx) __attribute__((always_inline)) {
        const int k = idx / (MT * NT), mt = (idx / NT) % MT, nt = idx % NT;
        if constexpr (AGPR)
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[mt][nt]) : "v"(wf[k][nt]), "v"(af[k][mt]));
        else
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[k][nt], af[k][mt], acc[mt][nt], 0, 0, 0);
    };
    // READ(s).  with_tail: the last TAILN MFMAs of the previous slab (k-step 1 fragments, registers only) are issued here,
    // AFTER the hand-over barrier, interleaved with the k-step 0 fragment reads of slab s - so the barrier's release latency
    // is covered by MFMA work of this group and this group's reads start at the hand-over, not TAILN MFMAs later.
    auto read_seg = [&](int s, auto with_tail) __attribute__((always_inline)) {
        const char* sb = smem + (s & 3) * SLAB;
        constexpr bool WT = decltype(with_tail)::value && TAILN > 0;
        constexpr int R0 = NT + MT;
        auto read0 = [&](int r) __attribute__((always_inline)) {
            if (r < NT) wf[0][r] = *(const bf16x8*)(sb + w_row_off + r * TSTRIDE + coff[0]);
            else af[0][r - NT] = *(const bf16x8*)(sb + a_row_off + (r - NT) * TSTRIDE + coff[0]);
        };
        if constexpr (WT) {
            constexpr int RPT = R0 / (TAILN > 0 ? TAILN : 1);  // k-step 0 reads per tail MFMA
            static_assert(R0 % (TAILN > 0 ? TAILN : 1) == 0, "tail interleave");
            // the tail must win the matrix pipe against the other group's freshly started segment (same-priority arbitration
            // is oldest-wave-first: the younger group's tail would sit behind the older group's whole segment and hold up
            // this in-order wave's fragment reads behind it)
            __builtin_amdgcn_s_setprio(3);
#pragma unroll
            for (int i = 0; i < TAILN; ++i) {
                one_mfma(NM - TAILN + i);
#pragma unroll
                for (int r = i * RPT; r < (i + 1) * RPT; ++r) read0(r);
            }
#pragma unroll
            for (int i = 0; i < TAILN; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, RPT, 0);
            }
            __builtin_amdgcn_s_setprio(0);
        } else {
#pragma unroll
            for (int r = 0; r < R0; ++r) read0(r);
        }
#pragma unroll
        for (int k = 1; k < KS; ++k) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wf[k][nt] = *(const bf16x8*)(sb + w_row_off + nt * TSTRIDE + coff[k]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[k][mt] = *(const bf16x8*)(sb + a_row_off + mt * TSTRIDE + coff[k]);
        }
        if constexpr (TRACE) ta = __builtin_amdgcn_s_memtime();
        if (s + 2 < ns) wait_vmcnt<IP>();  // slab s+1 landed (slab s+2 may still be in flight)
        else wait_vmcnt<0>();
        if constexpr (TRACE) tb = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (TRACE) {
            tc = __builtin_amdgcn_s_memtime();
            tr[0] += ta - tprev; tr[1] += tb - ta; tr[2] += tc - tb;
        }
        pp_barrier();
        if constexpr (TRACE) { tprev = __builtin_amdgcn_s_memtime(); }
    };
    auto mfma_seg = [&](int s, auto do_stage) __attribute__((always_inline)) {  // MFMAs [0, NM - TAILN) + the LDS-DMA of slab s+3
        __builtin_amdgcn_s_setprio(1);
        constexpr int HEAD = NM - TAILN;
        constexpr int EVERY = HEAD / IP > 0 ? HEAD / IP : 1;
        int issued = 0;
        char* base = smem + ((s + 3) & 3) * SLAB;
        const int soff = kbase + (s + 3) * RB;
#pragma unroll
        for (int i = 0; i < HEAD; ++i) {
            one_mfma(i);
            if constexpr (decltype(do_stage)::value) {
                if ((i + 1) % EVERY == 0 && issued < IP) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs[issued], LDS_PTR(base + ldsoff[issued]), 16, voff[issued], soff, 0, 0);
                    ++issued;
                }
            }
        }
        if constexpr (decltype(do_stage)::value) {
            // pin the interleave: EVERY MFMAs, one LDS-DMA issue, ... (a clustered burst of DMA issues would
            // starve the matrix pipe of this in-order wave for a few hundred cycles)
#pragma unroll
            for (int i = 0; i < IP; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x8, EVERY, 0);
                __builtin_amdgcn_sched_group_barrier(0x10, 1, 0);
            }
            if constexpr (HEAD - IP * EVERY > 0) __builtin_amdgcn_sched_group_barrier(0x8, HEAD - IP * EVERY, 0);
        }
        if constexpr (TRACE) {
            __builtin_amdgcn_sched_barrier(0);
            ta = __builtin_amdgcn_s_memtime();
            if constexpr (MODE == 0) tr[3] += tprev - tc;
            tr[4] += ta - tprev;
            tprev = ta;
        }
    };
    auto mfma_tail = [&]() __attribute__((always_inline)) {  // the last slab's tail (no hand-over follows)
#pragma unroll
        for (int i = NM - TAILN; i < NM; ++i) one_mfma(i);
        __builtin_amdgcn_s_setprio(0);
    };
    auto trace_gap = [&]() {  // after the post-MFMA barrier(s)
        if constexpr (TRACE) {
            ta = __builtin_amdgcn_s_memtime();
            tr[5] += ta - tprev;
            tprev = ta;
        }
    };

    // every MFMA segment but the last one ends with the hand-over barrier; its tail is issued by the next READ
    auto step = [&](int s, auto do_stage, auto with_tail) __attribute__((always_inline)) {
        read_seg(s, with_tail);
        mfma_seg(s, do_stage);
        if (s + 1 < ns) {
            pp_barrier();
            if constexpr (TAILN == 0) __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int g = 0; g < G - 2; ++g) pp_barrier();
        } else {
            mfma_tail();
        }
        trace_gap();
    };
    if constexpr (MODE == 0) {
        int s = 0;
        if (ns > 3) { step(0, std::true_type{}, std::false_type{}); s = 1; }
        for (; s + 3 < ns; ++s) step(s, std::true_type{}, std::true_type{});
        if (s == 0) { step(0, std::false_type{}, std::false_type{}); s = 1; }
        for (; s < ns; ++s) step(s, std::false_type{}, std::true_type{});
        for (int g = grp; g < G - 1; ++g) pp_barrier();  // equalise barrier counts before the (barrier-free) epilogue
    } else if constexpr (MODE == 2) {
        // MODE 2 ("register pipeline", small tiles): every wave runs the same stream, fragments double-buffered in registers -
        // the reads of slab k+1 are issued BEFORE the MFMAs of slab k, so the LDS latency hides behind them and an interval
        // is max(reads, MFMAs) + one barrier.  For tiles whose interval holds only 2-8 MFMAs per wave nothing is gained by
        // giving the matrix pipe to one wave group at a time; the serial READ -> MFMA dependency per interval is what costs.
        bf16x8 wf2[KS][NT], af2[KS][MT];
        auto reads_to = [&](int s, bf16x8 (&w)[KS][NT], bf16x8 (&a)[KS][MT]) __attribute__((always_inline)) {
            const char* sb = smem + (s & 3) * SLAB;
#pragma unroll
            for (int k = 0; k < KS; ++k) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) w[k][nt] = *(const bf16x8*)(sb + w_row_off + nt * TSTRIDE + coff[k]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a[k][mt] = *(const bf16x8*)(sb + a_row_off + mt * TSTRIDE + coff[k]);
            }
        };
        auto body = [&](int k, bf16x8 (&wc)[KS][NT], bf16x8 (&ac)[KS][MT], bf16x8 (&wn_)[KS][NT], bf16x8 (&an)[KS][MT])
                        __attribute__((always_inline)) {
            if (k + 1 < ns) reads_to(k + 1, wn_, an);  // slab k+1: waited for + barrier at the end of interval k-1
            const bool st = k + 3 < ns;
            if (st) stage(k + 3);                      // slot of slab k-1: its reads completed before MFMA(k-1), two barriers ago
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < KS; ++kk)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[kk][nt], ac[kk][mt], acc[mt][nt], 0, 0, 0);
            // the builtin (not an asm string): the compiler's own waitcnt pass must see that the next slab's fragments have
            // landed here, otherwise it guards the next interval's MFMAs with lgkmcnt waits that drain that interval's fresh reads
            __builtin_amdgcn_sched_barrier(0);   // keep the wait behind the MFMAs
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
            if (k + 1 < ns) {
                if (st) wait_vmcnt<IP>();  // slab k+2 landed, slab k+3 in flight
                else wait_vmcnt<0>();
                pp_barrier();
            }
        };
        // Large wave tiles (one wave per SIMD, e.g. 2x2 waves of 128x128): the interval's work must be ONE interleaved stream -
        // a burst of 16 ds_reads + 8 LDS-DMA issues in front of 32 MFMAs would leave the matrix pipe idle for hundreds of
        // cycles - so the steady-state body pins "2 MFMA, 1 fragment read, 2 MFMA, 1 fragment read, 1 LDS-DMA" groups.
        constexpr int RD = KS * (MT + NT);
        constexpr bool PINNED = (G == 1) && (NM == 4 * IP) && (RD == 2 * IP);
        auto body_pinned = [&](int k, bf16x8 (&wc)[KS][NT], bf16x8 (&ac)[KS][MT], bf16x8 (&wn_)[KS][NT], bf16x8 (&an)[KS][MT])
                               __attribute__((always_inline)) {
            // steady state only: slabs k+1 (read) and k+3 (staged) exist
            const char* sb = smem + ((k + 1) & 3) * SLAB;
            char* db = smem + ((k + 3) & 3) * SLAB;
            const int soff = kbase + (k + 3) * RB;
            auto rd = [&](int r) __attribute__((always_inline)) {  // fragment read r of slab k+1, k-step major
                const int kk = r / (MT + NT), j = r % (MT + NT);
                if (j < NT) wn_[kk][j] = *(const bf16x8*)(sb + w_row_off + j * TSTRIDE + coff[kk]);
                else an[kk][j - NT] = *(const bf16x8*)(sb + a_row_off + (j - NT) * TSTRIDE + coff[kk]);
            };
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                const int kk = i / (MT * NT), mt = (i / NT) % MT, nt = i % NT;
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[kk][nt], ac[kk][mt], acc[mt][nt], 0, 0, 0);
                // all fragment reads of the next slab in the FIRST half of the MFMA stream (one per MFMA): by the end of the
                // stream they have landed, so the lgkmcnt(0) in front of the barrier does not expose an LDS round trip
                if (i < RD) rd(i);
                if (i % 4 == 3)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs[i / 4], LDS_PTR(db + ldsoff[i / 4]), 16, voff[i / 4], soff, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < IP; ++j) {
                if (4 * j < RD) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x10, 1, 0);
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (TRACE) asm volatile("s_memtime %0" : "=s"(ta));   // T1: MFMA stream issued
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): slab k+1's fragments are in registers
            if constexpr (TRACE) asm volatile("s_memtime %0" : "=s"(tb));   // T2
            wait_vmcnt<IP>();                    // slab k+2 landed, slab k+3 in flight
            if constexpr (TRACE) asm volatile("s_memtime %0" : "=s"(tc));   // T3
            pp_barrier();
            if constexpr (TRACE) {  // buckets: 0 = MFMA / read / DMA stream, 2 = lgkmcnt wait, 1 = vmcnt wait, 3 = barrier
                unsigned long long td;
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(td));  // T4 (the wait is the trace build's overhead)
                tr[0] += ta - tprev; tr[2] += tb - ta; tr[1] += tc - tb; tr[3] += td - tc;
                tprev = td;
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        reads_to(0, wf, af);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        if (ns > 2) wait_vmcnt<IP>();  // slab 1 landed
        else wait_vmcnt<0>();
        pp_barrier();
        int k = 0;
        if constexpr (TRACE) { tprev = __builtin_amdgcn_s_memtime(); }
        if constexpr (PINNED) {
            for (; k + 4 < ns; k += 2) {  // both bodies of the pair are steady state: k + 1 + 3 < ns
                body_pinned(k, wf, af, wf2, af2);
                body_pinned(k + 1, wf2, af2, wf, af);
            }
        }
        for (; k < ns; k += 2) {
            body(k, wf, af, wf2, af2);
            if (k + 1 < ns) body(k + 1, wf2, af2, wf, af);
        }
    } else {
        auto reads = [&](int s) __attribute__((always_inline)) {
            const char* sb = smem + (s & 3) * SLAB;
#pragma unroll
            for (int k = 0; k < KS; ++k) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) wf[k][nt] = *(const bf16x8*)(sb + w_row_off + nt * TSTRIDE + coff[k]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) af[k][mt] = *(const bf16x8*)(sb + a_row_off + mt * TSTRIDE + coff[k]);
            }
        };
        const bool lead = __builtin_amdgcn_readfirstlane(grp == 0 ? 1 : 0) != 0;
        auto stamp = [&](int i) __attribute__((always_inline)) {  // TRACE: cycles since the previous stamp -> bucket i
            if constexpr (TRACE) {
                __builtin_amdgcn_sched_barrier(0);
                ta = __builtin_amdgcn_s_memtime();
                tr[i] += ta - tprev;
                tprev = ta;
            }
        };
        auto sync = [&](int k, auto do_stage) __attribute__((always_inline)) {
            if (k + 1 < ns) {
                if constexpr (decltype(do_stage)::value) wait_vmcnt<IP>();  // slab k+2 landed, slab k+3 in flight
                else wait_vmcnt<0>();
                stamp(1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                stamp(2);
                pp_barrier();
                stamp(5);
            }
        };
        if (lead) {  // MFMA(k) then READ(k+1)
            reads(0);
            if (ns > 2) wait_vmcnt<IP>();  // slab 1 landed
            else wait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            pp_barrier();
            auto interval = [&](int k, auto do_stage) __attribute__((always_inline)) {
                mfma_seg(k, do_stage);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                if (k + 1 < ns) reads(k + 1);
                stamp(0);
                sync(k, do_stage);
            };
            int k = 0;
            for (; k + 3 < ns; ++k) interval(k, std::true_type{});
            for (; k < ns; ++k) interval(k, std::false_type{});
        } else {     // READ(k) then MFMA(k)
            if (ns > 2) wait_vmcnt<IP>();
            else wait_vmcnt<0>();
            pp_barrier();
            auto interval = [&](int k, auto do_stage) __attribute__((always_inline)) {
                reads(k);
                stamp(0);
                mfma_seg(k, do_stage);
                __builtin_amdgcn_s_setprio(0);
                sync(k, do_stage);
            };
            int k = 0;
            for (; k + 3 < ns; ++k) interval(k, std::true_type{});
            for (; k < ns; ++k) interval(k, std::false_type{});
        }
    }
    unsigned long long t_loop_end = 0;
    if constexpr (TRACE) t_loop_end = __builtin_amdgcn_s_memtime();

    if constexpr (MODE == 1 && KS == 4 && !TRACE)  // (the small-M tiles only: keeps the 256-wide instantiations' code as it was)
    if (S > 1) {
        // Every part parks its fp32 partial (register image, [register quad][thread] = coalesced 16-byte accesses); the LAST arriver at
        // the tile's counter adds the others' and stores the tile (S = 2: a + b commutes; S = 4, round 5: summed in K-range order whichever
        // part arrives last, so the result does not depend on the arrival order either).  The two workgroups may sit on different XCDs, whose
        // L2s are not coherent: the partials are stored and loaded SYSTEM-coherent (sc0 sc1: written through to / read from the memory
        // side) and the counter is a system-scope atomic.  NOT __threadfence(): an agent-scope release / acquire on this part is
        // buffer_wbl2 + buffer_inv - a write-back and invalidate of the XCD's whole 4 MiB L2 - and made every split GEMM 40 us slower.
        // What the hand-off relies on: an sc0 sc1 store is written THROUGH the L2 and its vmcnt credit returns only with the
        // memory side's acknowledgement, so after `s_waitcnt vmcnt(0)` (inline asm: the compiler cannot drop or move it) + the
        // workgroup barrier every lane's partial is visible device-wide BEFORE lane 0 touches the counter; the reader's sc0 sc1 loads
        // miss its own L2 by definition and are issued after the counter's returning atomic has told it that it is second.  This is the
        // "drained sc1 payload -> asm vmcnt(0) -> sc1 flag" form of /opt/skills/guides/MI355X_MICROARCH.md (price list, row
        // handoff-flag; the failure it warns about - the flag overtaking the write-back - needs a compiler-visible fence to be dropped,
        // which the asm wait is not).  tests/test_gpu_ops.py::test_gemm_splitk_handoff_stress runs 3000 split launches against a
        // concurrent L2-thrashing stream and checks every output word of every launch.
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4_;
        constexpr int PART_BYTES = BM * BN * 4, CP = 17;  // aux bits: sc0 | sc1
        const __amdgpu_buffer_rsrc_t rMine = __builtin_amdgcn_make_buffer_rsrc((void*)(p.splitk_part + ((size_t)bid * S + ksplit) * (BM * BN)), 0, PART_BYTES, 0x00020000);
        // (all S partials of the tile behind one descriptor: part j at byte offset j * PART_BYTES)
        const __amdgpu_buffer_rsrc_t rAll = __builtin_amdgcn_make_buffer_rsrc((void*)(p.splitk_part + (size_t)bid * S * (BM * BN)), 0, S * PART_BYTES, 0x00020000);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = {acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1], acc[mt][nt][4 * q + 2], acc[mt][nt][4 * q + 3]};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, v), rMine, (((mt * NT + nt) * 4 + q) * (NW * 64) + tid) * 16, 0, CP);
                }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave's partial has reached the memory side ...
        __syncthreads();                                   // ... before the workgroup reports in
        int* flag = (int*)smem;  // (the slab ring is idle: every wave is past its last fragment read)
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(p.splitk_cnt + bid, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (old == (unsigned)(S - 1)) __hip_atomic_store(p.splitk_cnt + bid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // all arrived: free for the next launch
            *flag = (int)old;
        }
        __syncthreads();
        if (*flag != S - 1) return;  // not the last arriver (uniform)
        if (S == 2) {  // own + other (fp32 addition commutes)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 o = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rAll, (((mt * NT + nt) * 4 + q) * (NW * 64) + tid) * 16, (1 - ksplit) * PART_BYTES, CP));
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[mt][nt][4 * q + j] += o[j];
                    }
        } else {  // ((p0 + p1) + p2) + p3 with this workgroup's own part taken from its registers, whichever position it has
            f32x16 tot[MT][NT];
#pragma unroll
            for (int part = 0; part < 4; ++part) {
                if (part >= S) break;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            f32x4 o = {acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1], acc[mt][nt][4 * q + 2], acc[mt][nt][4 * q + 3]};
                            if (part != ksplit)  // (uniform)
                                o = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rAll, (((mt * NT + nt) * 4 + q) * (NW * 64) + tid) * 16, part * PART_BYTES, CP));
#pragma unroll
                            for (int j = 0; j < 4; ++j) tot[mt][nt][4 * q + j] = part == 0 ? o[j] : tot[mt][nt][4 * q + j] + o[j];
                        }
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = tot[mt][nt];
        }
    }
    store_tile<MT, NT, EPI>(acc, p, m0, n0, wm, wn, hi, l31);

    if constexpr (MODE == 1 && KS == 4 && EPI == 0 && !TRACE) {
        // GemmArgs::rowstat (round 5, the 512-row-class QKV projection): per row and per BN-column tile, (sum, sum of squares) of the
        // tile's bf16-ROUNDED outputs -> rowstat[row][n0 / BN].  The fused small-N attention kernel reduces the tiles of the Q (or K)
        // columns to the LayerNorm statistics of q_norm / k_norm itself, so no q / k post-processing launch runs in between.
        // v_dot2_f32_bf16 of each packed output pair with (1, 1) and with itself; row halves meet through lane ^ 32, the WN waves of a
        // row through the (idle) slab ring.  Columns past N are zero accumulators (their W rows read as zero) and add nothing.
        if (p.rowstat) {
            typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
            const bf16x2_ ones = __builtin_bit_cast(bf16x2_, 0x3F803F80u);
            __syncthreads();  // every wave is past its last fragment read: the ring becomes the exchange buffer
            float2* ex = (float2*)smem;  // [WN][BM]
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const bf16x2_ v = __builtin_bit_cast(bf16x2_, pack2bf_pk(acc[mt][nt][2 * j], acc[mt][nt][2 * j + 1]));
                        s1 = __builtin_amdgcn_fdot2_f32_bf16(v, ones, s1, false);
                        s2 = __builtin_amdgcn_fdot2_f32_bf16(v, v, s2, false);
                    }
                s1 += __shfl_xor(s1, 32, 64);
                s2 += __shfl_xor(s2, 32, 64);
                if (hi == 0) ex[wn * BM + wm * MT * 32 + mt * 32 + l31] = float2{s1, s2};
            }
            __syncthreads();
            if (tid < BM && m0 + tid < p.M) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int w = 0; w < WN; ++w) { s1 += ex[w * BM + tid].x; s2 += ex[w * BM + tid].y; }
                ((float2*)p.rowstat)[(size_t)(m0 + tid) * p.rowstat_slots + tn] = float2{s1, s2};
            }
        }
    }

    if constexpr (TRACE) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // stores acknowledged
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        if (p.trace && lane == 0 && (blockIdx.x & 63) == 5) {
            unsigned long long* o = p.trace + ((size_t)(blockIdx.x >> 6) * NW + wave) * 8;
#pragma unroll
            for (int i = 0; i < 6; ++i) o[i] = tr[i];
            o[6] = ((unsigned long long)ns << 32) | (unsigned)(tstart - t_entry);   // slabs | prologue cycles
            o[7] = ((t_loop_end - tstart) << 20) | ((t_end - t_loop_end) & 0xfffff);  // main-loop cycles | epilogue cycles
        }
    }
}

// ---- persistent 4-wave kernel on 16x16x32 MFMAs: 256 x 256 or 256 x 288 tiles (variants 15 / 16) ----------------------------------
// What round 2's measurements asked for (profiles/r02/opbench_gemm_probe.log): the 4-wave kernels run their main loop
// at 0.65-0.8 us per 256x256x32 slab (1.3-1.65 PFLOP/s) and the persistent form carries it across tiles, but with 32x32 MFMAs a
// wave's 128-column half tile only comes in multiples of 32 columns, i.e. 256-wide tiles - and 8192 x 6912 / 2304 (QKV, O, W2 of the
// 2B model) are 3.375 / 1.125 rounds of 256-wide tiles over 256 CUs.  With v_mfma_f32_16x16x32_bf16 the wave tile is 128 x 144 just as
// well (8 x 9 accumulator tiles of 4 registers = 288 AGPRs, fragments double-buffered in 136 VGPRs): 256 x 288 tiles make those
// shapes exactly 3 / 1 tiles per CU.  Structure = gemm_bf16_w4p (one workgroup per CU walks its tiles; 4-slot LDS ring of 32-deep
// slabs filled by LDS-DMA three slabs ahead, running across tile boundaries; fragments of slab g+1 read while slab g multiplies;
// one barrier per slab; epilogue stores issued and not waited for), with these differences:
//  * one MFMA covers the slab's whole depth (K = 32): 8 x NT MFMAs of 16 cycles per slab, one fragment read per 4 MFMAs (8 + NT reads
//    spread over the whole body since round 3: all four waves leave the barrier together, and 68 KiB of reads in the first quarter
//    of the body queued behind each other - GEMM class +1 %, profiles/r03/bench_ab_w4q_fragment_read_spacing.log), one LDS-DMA per 8;
//  * a 16-row fragment read (lane l: row l & 15, 16-byte chunk l >> 4 of the 64-byte row) is bank-conflict free when chunk c of row
//    r sits at position c ^ (3 * ((r >> 3) & 1)) - on the LDS-DMA's source address and on the read (guide rule 21);
//  * D' = W_frag x A_frag puts 4 consecutive columns of one C row in a lane; v_permlane16_swap of two neighbouring accumulator
//    tiles widens that to 8 consecutive columns = one 16-byte store per lane and tile pair;
//  * accumulation order inside a slab differs from the 32x32x16 kernels (one K = 32 MFMA instead of two K = 16), so results are
//    equal to fp32 rounding, not bit-identical to variant 1.
// EPI 1 (SwiGLU) needs NT % 4 == 0 (w1 / w3 interleaved in 32-row groups = pairs of 16-column tiles).  K % 64 == 0, K >= 128, no bias.
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
// TRACE (experimental build, lt_op_gemm_trace variants 15 / 16): per workgroup 8 x u64 = s_memrealtime (100 MHz) at entry | prologue
// done | last main loop done | last epilogue issued | exit (stores acknowledged), shader clocks of the first tile's main loop,
// HW_ID, tiles walked.
// GROUPED (round 4; the experts' GEMMs of Next-DiT-MoE at 1024^2, Next-DiT-MoE/models/models2.py:459-506 - 16 384 routed rows per MoE
// FFN is the MFMA-bound regime): the persistent walk runs over the VALID row tiles only (tile_expert[tm] >= 0, compacted into an LDS
// list once per workgroup), every tile multiplies with W + tile_expert[tm] * w_expert_stride, and - gather-on-load, a_row_map - row r
// of a tile is row a_row_map[m0 + r] of A (-1: reads as zero).  The A stream then goes through ONE descriptor over all of A with
// per-tile lane offsets; the 256 map entries of the tile after next are fetched by a 4-byte LDS-DMA at every tile boundary (counted
// in the hand-kept vmcnt like the slabs) and turned into lane offsets where the DMA stream crosses into that tile.  K >= 256.
// LT_W4Q_PD (round 6): how many slabs ahead of the one being multiplied the LDS-DMA stream runs.  Body g multiplies slab g from
// registers and reads slab g + 1's fragments, so slot g & 3 has been free since the barrier that ended body g - 1: the four-slot ring
// carries a distance of 4 as well as the 3 it was built with (rounds 2-5), with one more body of cover for every fill (slab g + 2 must have
// landed at the end of body g in both forms; 3: issued during body g - 1, 4: during body g - 2) and one more slab in flight.
#ifndef LT_W4Q_PD
#define LT_W4Q_PD 3
#endif
template <int EPI, int NW16, bool TRACE = false, bool GROUPED = false>
__global__ __launch_bounds__(256, 1) void gemm_bf16_w4q(GemmArgs p) {
    constexpr int PD = LT_W4Q_PD;
    static_assert(PD == 3 || PD == 4, "the DMA stream runs 3 or 4 slabs ahead (4 ring slots, fragments of slab g + 1 in registers during body g + 1)");
    constexpr int MT = 8, NT = NW16, NW = 4, BM = 256, BN = 2 * NW16 * 16;
    constexpr int PA = BM / 16, PW = BN / 16, NP = PA + PW;   // 1-KiB staging pieces (16 rows x 64 B) per slab
    constexpr int IP = (NP + NW - 1) / NW;                    // pieces per wave and slab (a surplus slot re-loads the wave's last piece)
    constexpr int SLAB = (BM + BN) * 64, W_OFF = BM * 64;
    constexpr int NM = MT * NT, RD = MT + NT, EVERY = NM / IP, RS = NM / RD;
    constexpr int NST = EPI != 1 ? MT * (NT / 2) + (NT % 2 ? MT : 0) : MT * (NT / 4);  // store instructions per wave and tile
    constexpr int NST_V = (MT / 2) * NT;  // ... of a V^T tile (EPI 3)
    // round 6: the grouped W2 launch of the MoE at 1024^2 is 1.5-1.6 rounds of 256 x 256 tiles - the second round a
    // third to two thirds empty.  TSPLIT: the tiles of a partial LAST round are cut along K into S parts (GemmArgs::tail_*; S picked on the
    // device from the number of valid tiles), every part is one item of the persistent walk, parks its fp32 accumulators in a workspace and
    // reports in; the last arriver of a tile sums the S parts in K order and runs the normal epilogue (the small-M kernels' hand-off).
    constexpr bool TSPLIT = GROUPED && EPI == 0;
    constexpr bool YST = EPI == 0 && !GROUPED;  // round 6: the plain dense kernels can emit per-row sums of squares of their outputs (GemmArgs::ystat)
    constexpr int NST_Q = (EPI == 3 || YST) ? MT : 0;  // plain tiles of EPI 3 / YST: + the partial-sum stores (one per row tile and wave, always issued)
    static_assert(NM % IP == 0 && RS >= 1 && (RD - 1) * RS + 4 <= NM, "one LDS-DMA per EVERY MFMAs, one fragment read per RS MFMAs, the last one >= 4 MFMAs before the wait");
    static_assert(EPI != 1 || NT % 4 == 0, "SwiGLU pairs 32-column groups");
    static_assert(PA % NW == 0, "A pieces first: slot i < PA / NW is an A piece for every wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, q4 = lane >> 4;
    const int TMall = (p.M + BM - 1) / BM, TN = (p.N + BN - 1) / BN;
    int TM = TMall;
    // GROUPED: behind the slab ring - two 1-KiB slots of gather-map entries (tile parity), the list of valid row tiles (<= 1024) + count
    char* const g_map = smem + 4 * SLAB;
    int* const g_list = (int*)(smem + 4 * SLAB + 2048);
    const bool gather = GROUPED && p.a_row_map != nullptr;
    if constexpr (GROUPED) {
        if (wave == 0) {
            int cnt = 0;
            for (int base = 0; base < TMall; base += 64) {
                const int idx = base + lane;
                const int ex = idx < TMall ? p.tile_expert[idx] : -1;
                const bool ok = ex >= 0;
                const unsigned long long b = __ballot(ok);
                if (ok) g_list[cnt + __popcll(b & ((1ull << lane) - 1ull))] = idx | (ex << 16);  // row tile | its expert
                cnt += __popcll(b);
            }
            if (lane == 0) g_list[1024] = cnt;
        }
        __syncthreads();
        TM = __builtin_amdgcn_readfirstlane(g_list[1024]);
    }
    const int ntiles = TM * TN;
    const int ns = p.K / 32;
    // virtual items of the walk: the whole rounds' tiles as they are, then S parts per tile of the partial last round
    int nv = ntiles, tail0 = ntiles, tsl = 0;  // tsl: log2 of the parts per tail tile
    if constexpr (TSPLIT) {
        const int G = (int)gridDim.x, R = ntiles % G;
        if (p.tail_part && ntiles > G && R != 0) {
            int best = 8;  // eighths of a tile time the partial round costs: ceil(R S / G) / S, unsplit = 1
#pragma unroll
            for (int S = 2; S <= (p.tail_max_parts >= 4 ? 4 : 2); S *= 2) {  // (8 parts: 32 registers of partials in flight per accumulator tile - the epilogue spilled)
                if (ns % (2 * S) != 0 || ns / S < 8 || (long long)R * S > p.tail_cap_parts) continue;
                const int c = ((R * S + G - 1) / G) * (8 / S);
                if (c < best) { best = c; tsl = S == 2 ? 1 : 2; }
            }
            if (tsl) { tail0 = ntiles - R; nv = tail0 + (R << tsl); }
        }
    }
    unsigned long long tr_entry = 0, tr_pro = 0, tr_loop = 0, tr_epi = 0, tr_clk = 0;
    if constexpr (TRACE) tr_entry = __builtin_amdgcn_s_memrealtime();

    // staging: wave w copies pieces w + 4 i; piece q < PA = A rows 16 q .., else W rows 16 (q - PA) ..; lane -> row lane >> 2, 16-byte
    // position lane & 3, fetched from source chunk pos ^ (3 * ((row >> 3) & 1))
    const int sswz = ((lane & 3) ^ (((lane >> 5) & 1) * 3)) * 16;
    // pair_ab bit 0: A, bit 1: W in the pair layout (the dense engine path sets both; the grouped expert GEMMs: W always, A where it is not
    // gathered row by row - a gathered row's line mate is not the next row of the tile)
    const bool pair_a = (p.pair_ab & 1) != 0 && !(GROUPED && p.a_row_map != nullptr), pair_w = (p.pair_ab & 2) != 0;
    const int psh_a = pair_a ? 1 : 0, psh_w = pair_w ? 1 : 0;  // a slab's step along K in bytes: 64 << psh
    static_assert(IP <= 9, "staging slots per wave");
    int voff[9], ldsoff[9];  // fixed size: a dependent bound here breaks host-side substitution (hipcc 7.2)
#pragma unroll
    for (int i = 0; i < IP; ++i) {
        int q = wave + NW * i;
        if (q >= NP) q -= NW;
        const bool isA = i < PA / NW;
        const int r0 = 16 * (isA ? q : q - PA) + (lane >> 2);
        // GemmArgs::pair_ab (round 6): A and W are stored row-pair-interleaved per 32-deep K chunk ([rows / 2][K / 32][2][32]: the two rows'
        // 64-byte slab pieces side by side = ONE 128-byte line), so an LDS-DMA instruction touches 8 whole lines instead of 16 half lines -
        // half the requests into the L2 for the same bytes; the LDS image, the fragment reads and the arithmetic are the same
        voff[i] = (isA ? pair_a : pair_w) ? (r0 >> 1) * (isA ? p.lda : p.ldw) * 4 + (r0 & 1) * 64 + sswz : r0 * (isA ? p.lda : p.ldw) * 2 + sswz;
        ldsoff[i] = q * 1024;
    }
    const int ncols_out = EPI == 1 ? p.N / 2 : p.N;
    struct Tile { const u16* a; const u16* w; u16* c; int a_bytes, w_bytes, c_bytes, n0, m0, part; };
    // bytes along K an item multiplies: everything (part < 0), or part `part & (parts - 1)` of its tail tile
    auto kbeg_of = [&](int part) __attribute__((always_inline)) { return part < 0 ? 0 : (part & ((1 << tsl) - 1)) * ((ns >> tsl) * 64); };
    auto kend_of = [&](int part) __attribute__((always_inline)) { return part < 0 ? ns * 64 : ((part & ((1 << tsl) - 1)) + 1) * ((ns >> tsl) * 64); };
    auto setup = [&](int v) __attribute__((always_inline)) {
        int tm, tn;
        int part_ = -1;  // >= 0: part `part_ & (parts - 1)` of tail tile `part_ >> tsl`
        if constexpr (TSPLIT) {
            if (v >= tail0) {
                part_ = v - tail0;
                v = tail0 + (part_ >> tsl);
            }
        }
        tile_coords(v, ntiles, TM, TN, tm, tn, p.group_rows > 0 ? p.group_rows : 4);
        const u16* wbase = p.W;
        if constexpr (GROUPED) {
            // (from LDS, not from the table in memory: a global load at a tile boundary is followed by s_waitcnt vmcnt(0) = a drain of
            //  the slabs in flight and of the epilogue's stores; readfirstlane: a W descriptor derived from a VGPR costs a waterfall
            //  loop around every LDS-DMA)
            const int e = __builtin_amdgcn_readfirstlane(g_list[tm]);
            tm = e & 0xffff;
            wbase += (size_t)(e >> 16) * p.w_expert_stride;
        }
        const int m0 = tm * BM, n0_ = tn * BN;
        // GROUPED: one descriptor over ALL of A (the lanes' offsets carry the rows: a_row_map[m0 + r] or m0 + r)
        const long long a_left = GROUPED ? (long long)(p.a_row_map ? p.a_map_rows : p.M) * p.lda * 2 : (long long)(p.M - m0) * p.lda * 2;
        const long long w_left = (long long)(p.N - n0_) * p.ldw * 2;
        const long long c_left = (long long)(p.M - m0) * p.ldc * 2;
        Tile t;
        t.a = GROUPED ? p.A : p.A + (size_t)m0 * p.lda; t.w = wbase + (size_t)n0_ * p.ldw; t.c = p.C + (size_t)m0 * p.ldc;
        t.a_bytes = (int)(a_left > 0x7fffffffLL ? 0x7fffffffLL : a_left);
        t.w_bytes = (int)(w_left > 0x7fffffffLL ? 0x7fffffffLL : w_left);
        t.c_bytes = (int)(c_left > 0x7fffffffLL ? 0x7fffffffLL : c_left);
        t.n0 = n0_;
        t.m0 = m0;
        t.part = part_;
        return t;
    };
    const Tile t_null = {p.A, p.W, p.C, 0, 0, 0, 0, 0, -1};  // no next tile: the last bodies' DMAs read nothing (all lanes out of range)

    // fragment reads: lane -> row l15 of the 16-row tile, chunk q4 (swizzled)
    const int csw = (q4 ^ (((l15 >> 3) & 1) * 3)) << 4;
    const int a_row_off = (wm * (MT * 16) + l15) * 64 + csw;          // + mt * 1024
    const int w_row_off = W_OFF + (wn * (NT * 16) + l15) * 64 + csw;  // + nt * 1024

    f32x4 acc[MT][NT];
    bf16x8 wf[NT], af[MT], wf2[NT], af2[MT];

    int v = blockIdx.x;
    if (v >= nv) return;  // uniform
    const int my_tiles = (nv - 1 - v) / (int)gridDim.x + 1;
    Tile cur = setup(v);
    bool has_next = v + (int)gridDim.x < nv;
    Tile nxt = has_next ? setup(v + gridDim.x) : t_null;

    // GROUPED: byte offsets of this lane's four A-piece rows in the tile the DMA stream is in (ga) - dense: voff[0..3], tile-invariant
    constexpr int NGA = PA / NW;
    int ga[NGA];
    // (the lane id comes from lane_now(): an opaque, fresh value - terms derived from the kernel's `lane` were hoisted out of the tile
    //  loop, spilled, and re-loaded from scratch in every tile's first body with an s_waitcnt vmcnt(0) behind the re-load)
    auto lane_now = [&]() __attribute__((always_inline)) {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    };
    auto ga_of_tile = [&](const Tile& t, const int* map, int (&o)[NGA]) __attribute__((always_inline)) {  // map: the tile's 256 entries, or null
        const int ln = lane_now();
        const int swz = ((ln & 3) ^ (((ln >> 5) & 1) * 3)) * 16;
#pragma unroll
        for (int i = 0; i < NGA; ++i) {
            const int r = 16 * (wave + NW * i) + (ln >> 2);
            const int src = t.c_bytes == 0 ? -1 : (map ? map[r] : t.m0 + r);  // the null tile behind the last one: every lane out of range
            o[i] = (src >= 0 ? (pair_a ? (src >> 1) * p.lda * 4 + (src & 1) * 64 : src * p.lda * 2) : 0x40000000) + swz;
        }
    };
    // GROUPED: ONE loop-invariant descriptor over all of A (< 2^30 bytes, launcher) - a descriptor carried from tile to tile ended up in
    // VGPRs here (four waterfall loops per slab)
    const __amdgpu_buffer_rsrc_t gA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, GROUPED ? (p.a_row_map ? p.a_map_rows : p.M) * p.lda * 2 : 0, 0x00020000);
    // the 256 gather-map entries of tile `t` -> LDS slot `slot` (one 4-byte LDS-DMA per wave: 64 entries); a null tile reads nothing
    auto map_dma = [&](const Tile& t, int slot, bool real) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rM = __builtin_amdgcn_make_buffer_rsrc((void*)(p.a_row_map + t.m0), 0, real ? 1024 : 0, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rM, LDS_PTR(g_map + slot * 1024 + wave * 256), 4, (wave * 64 + lane_now()) * 4, 0, 0, 0);
    };
    if constexpr (GROUPED) {
        ga_of_tile(cur, gather ? p.a_row_map + cur.m0 : nullptr, ga);  // (plain loads: the compiler waits for them right here)
        if (gather) map_dma(nxt, 1, has_next);  // oldest vmcnt entry of the prologue: every wait below covers it
    } else {
#pragma unroll
        for (int i = 0; i < NGA; ++i) ga[i] = voff[i];
    }
    int map_slot = 1;  // LDS slot holding the map of the tile AFTER the one the DMA stream is in
    auto stage_from = [&](int g, int slab_in_tile, const Tile& t) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rA = GROUPED ? gA : __builtin_amdgcn_make_buffer_rsrc((void*)t.a, 0, t.a_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)t.w, 0, t.w_bytes, 0x00020000);
        char* base = smem + (g & 3) * SLAB;
        const int soff = (TSPLIT ? kbeg_of(t.part) : 0) + slab_in_tile * 64;
#pragma unroll
        for (int i = 0; i < IP; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(i < PA / NW ? rA : rW, LDS_PTR(base + ldsoff[i]), 16, i < NGA ? ga[i] : voff[i], soff << (i < PA / NW ? psh_a : psh_w), 0, 0);
    };
    stagger_start(p.stagger);
    // prologue (once per workgroup): slabs 0..2 in flight, slab 0 read into the first fragment set, slab 1 landed and visible
    stage_from(0, 0, cur);
    stage_from(1, 1, cur);
    stage_from(2, 2, cur);
    if constexpr (PD == 4) stage_from(3, 3, cur);
    wait_vmcnt<(PD - 1) * IP>();
    pp_barrier();
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) wf[nt] = *(const bf16x8*)(smem + w_row_off + nt * 1024);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) af[mt] = *(const bf16x8*)(smem + a_row_off + mt * 1024);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    wait_vmcnt<(PD - 2) * IP>();
    pp_barrier();

    int after_epilogue = 0;  // stores of the previous tile's epilogue still in the queue: 0 none, 1 NST, 2 NST_V (EPI 3: a V^T tile)
    if constexpr (TRACE) { tr_pro = __builtin_amdgcn_s_memrealtime(); tr_clk = __builtin_amdgcn_s_memtime(); }
    // scalar state of the two streams, carried from body to body.  Each body computes the NEXT body's values inside its own MFMA
    // stream (the fences below pin them there): between the barrier and the first MFMA of a slab there is nothing but one address
    // add.  (Round 2's first form rebuilt descriptors, slot offsets and the tile selects at the top of every body: ~25 dependent
    // scalar instructions per slab in front of an idle matrix pipe - profiles/r02/gemm_trace_w4q.log, 80-87 % main-loop duty.)
    int rd_off = SLAB;       // LDS offset of slab g + 1 (fragment reads of this body)
    int wr_off = (PD & 3) * SLAB;   // LDS offset of slab g + PD (LDS-DMA destination of this body)
    int d_soff = (TSPLIT ? kbeg_of(cur.part) : 0) + PD * 64;     // byte offset along K of the slab the DMA stream fetches next ...
    int d_kend = TSPLIT ? kend_of(cur.part) : ns * 64;    // ... and where the stream's tile ends along K
    __amdgpu_buffer_rsrc_t dA = __builtin_amdgcn_make_buffer_rsrc((void*)cur.a, 0, cur.a_bytes, 0x00020000);   // ... in this tile
    __amdgpu_buffer_rsrc_t dW = __builtin_amdgcn_make_buffer_rsrc((void*)cur.w, 0, cur.w_bytes, 0x00020000);
    // one slab: MFMAs of slab g from (wc, ac) | fragment reads of slab g+1 into (wn_, an) | LDS-DMA of the slab three ahead.
    // FIRST (a tile's slab 0): C = 0 forms, no accumulator clears.  Accumulator tiles 0..63 live in AGPRs, the rest (NT = 9: 8 tiles)
    // in arch VGPRs; inline assembly gives each ONE home (the builtin bounced tiles through spare AGPRs: 272 us instead of 216 us on
    // the QKV GEMM) and, with a scheduling fence per MFMA, pins the written interleave.
    // SWAP (EPI 3, the V columns of a fused QKV projection): D = A_frag x W_frag instead of W_frag x A_frag - the same products,
    // but a lane then holds 4 consecutive ROWS (tokens) of one column, which is what the transposed V image wants.
    auto body = [&](auto first_tag, auto swap_tag, bf16x8 (&wc)[NT], bf16x8 (&ac)[MT], bf16x8 (&wn_)[NT], bf16x8 (&an)[MT]) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_tag)::value, SWAP = decltype(swap_tag)::value;
        const char* sb = smem + rd_off;
        char* db = smem + wr_off;
        const int d_soff_a = d_soff << psh_a, d_soff_w = d_soff << psh_w;  // (pair layout: 128 bytes per slab)
        int n_rd = rd_off, n_wr = wr_off, n_soff = d_soff, n_kend = d_kend;
        __amdgpu_buffer_rsrc_t nA = dA, nW = dW;
        int n_ga[NGA];
#pragma unroll
        for (int i = 0; i < NGA; ++i) n_ga[i] = ga[i];
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            const int mt = i / NT, nt = i % NT;
            const int id = mt * NT + nt;  // accumulator tile: the first 64 live in AGPRs
            if constexpr (FIRST && !SWAP) {
                if (id < 64) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc[mt][nt]) : "v"(wc[nt]), "v"(ac[mt]));
                else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=v"(acc[mt][nt]) : "v"(wc[nt]), "v"(ac[mt]));
            } else if constexpr (!FIRST && !SWAP) {
                if (id < 64) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[mt][nt]) : "v"(wc[nt]), "v"(ac[mt]));
                else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[mt][nt]) : "v"(wc[nt]), "v"(ac[mt]));
            } else if constexpr (FIRST) {
                if (id < 64) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc[mt][nt]) : "v"(ac[mt]), "v"(wc[nt]));
                else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=v"(acc[mt][nt]) : "v"(ac[mt]), "v"(wc[nt]));
            } else {
                if (id < 64) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[mt][nt]) : "v"(ac[mt]), "v"(wc[nt]));
                else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[mt][nt]) : "v"(ac[mt]), "v"(wc[nt]));
            }
            if (i % RS == 0 && i / RS < RD) {  // one fragment read per RS MFMAs, spread over the whole body
                const int j = i / RS;
                if (j < NT) wn_[j] = *(const bf16x8*)(sb + w_row_off + j * 1024);
                else an[j - NT] = *(const bf16x8*)(sb + a_row_off + (j - NT) * 1024);
            }
            if (i % EVERY == EVERY / 2)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(i / EVERY < PA / NW ? (GROUPED ? gA : dA) : dW, LDS_PTR(db + ldsoff[i / EVERY]), 16,
                                                         i / EVERY < NGA ? ga[i / EVERY < NGA ? i / EVERY : 0] : voff[i / EVERY], i / EVERY < PA / NW ? d_soff_a : d_soff_w, 0, 0);
            // the next body's scalar state, a few instructions under each of the last MFMAs
            if (i == NM - 4) { n_rd = rd_off + SLAB; n_rd = n_rd == 4 * SLAB ? 0 : n_rd; }
            if (i == NM - 3) { n_wr = wr_off + SLAB; n_wr = n_wr == 4 * SLAB ? 0 : n_wr; n_soff = d_soff + 64; }
            if (i == NM - 2) {
                if (n_soff == d_kend) {  // the DMA stream moves on to the next tile (its last three slabs ride in this tile's bodies)
                    n_soff = TSPLIT ? kbeg_of(nxt.part) : 0; n_kend = TSPLIT ? kend_of(nxt.part) : ns * 64;
                    if constexpr (!GROUPED) nA = __builtin_amdgcn_make_buffer_rsrc((void*)nxt.a, 0, nxt.a_bytes, 0x00020000);
                    nW = __builtin_amdgcn_make_buffer_rsrc((void*)nxt.w, 0, nxt.w_bytes, 0x00020000);
                    // GROUPED: the next tile's lane offsets (its map entries landed in LDS >= 2 barriers ago: K >= 256, launcher)
                    if constexpr (GROUPED) ga_of_tile(nxt, gather ? (const int*)(g_map + map_slot * 1024) : nullptr, n_ga);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
        rd_off = n_rd; wr_off = n_wr; d_soff = n_soff; d_kend = n_kend; dA = nA; dW = nW;
        if constexpr (GROUPED) {
#pragma unroll
            for (int i = 0; i < NGA; ++i) ga[i] = n_ga[i];
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): slab g+1's fragments are in registers
        // slab g+2 landed; still allowed in flight: this body's IP DMAs and, right after a tile boundary, the NST stores issued
        // between slab g+2's DMAs and them (loads and stores retire in issue order, one counter)
        // (PD - 2 slabs of DMAs: slabs g + 3 .. g + PD; the counter has 6 bits - a smaller bound only waits for more than it needs)
        constexpr int INF = (PD - 2) * IP;
        constexpr auto cap = [](int n) constexpr { return n < 63 ? n : 63; };
        if constexpr (FIRST) {
            if (GROUPED && after_epilogue == 1 && gather) wait_vmcnt<cap(INF + NST + 1)>();  // + the map LDS-DMA issued behind the stores
            else if (after_epilogue == 1) wait_vmcnt<cap(INF + NST + NST_Q)>();
            else if (EPI == 3 && after_epilogue == 2) wait_vmcnt<cap(INF + NST_V)>();
            else wait_vmcnt<INF>();
            after_epilogue = 0;
        } else {
            wait_vmcnt<INF>();
        }
        pp_barrier();
    };
    // epilogue through the tile's C descriptor (rows past M fall outside num_records, columns past N get an out-of-range offset):
    // every wave issues exactly NST store instructions per tile.  Lane holds, per 16x16 accumulator tile, C row l15 and columns
    // 4 q4 + r (register r).
    // (round 4) The epilogues take their lane coordinates from a fresh, opaque lane id: derived from the kernel's `lane` the compiler
    // hoisted the tile-invariant address terms out of the tile loop, spilled them (the main loop owns all 512 registers) and
    // re-loaded them from scratch here - and a scratch re-load is followed by s_waitcnt vmcnt(0), i.e. by a drain of the next tile's
    // three slabs in flight AND of every store this epilogue has issued so far ("stores issued and not waited for" was not true).
    // (lane_now(): defined above, next to the grouped mode's lane offsets)
    auto store_out = [&](const Tile& t) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc((void*)t.c, 0, t.c_bytes, 0x00020000);
        const int lane_e = lane_now(), l15 = lane_e & 15, q4 = lane_e >> 4;  // shadow the kernel-scope values
        const int nbase = t.n0 + wn * (NT * 16);
        // EPI 3: LayerNorm partial sums of the Q columns (GemmArgs::qstat) - v_dot2_f32_bf16 of each packed output pair with (1, 1) and
        // with itself: the sums run over the ROUNDED values, which is what the reference normalises (nn.LayerNorm of the bf16 Linear output)
        const bool stats = EPI == 3 && p.qstat != nullptr && t.n0 < p.qstat_cols;
        const bool ystats = YST && p.ystat != nullptr;
        typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
        const bf2_t ones2 = __builtin_bit_cast(bf2_t, 0x3F803F80u);
        float st1[MT], st2[MT];
        auto stat_add = [&](int mt, unsigned w) __attribute__((always_inline)) {
            const bf2_t v = __builtin_bit_cast(bf2_t, w);
            st1[mt] = __builtin_amdgcn_fdot2_f32_bf16(v, ones2, st1[mt], false);
            st2[mt] = __builtin_amdgcn_fdot2_f32_bf16(v, v, st2[mt], false);
        };
        auto stat_sq = [&](int mt, unsigned w) __attribute__((always_inline)) {
            const bf2_t v = __builtin_bit_cast(bf2_t, w);
            st2[mt] = __builtin_amdgcn_fdot2_f32_bf16(v, v, st2[mt], false);
        };
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int row_off = (wm * (MT * 16) + mt * 16 + l15) * p.ldc * 2;  // bytes from the tile's first C row (< 2^31: launcher)
            const int row_off_pc = ((wm * (MT * 16) + mt * 16 + l15) >> 1) * p.ldc * 4 + (l15 & 1) * 64;  // ... in the pair layout (tiles start on even rows)
            st1[mt] = 0.f; st2[mt] = 0.f;
            if constexpr (EPI != 1) {
#pragma unroll
                for (int np = 0; np < NT / 2; ++np) {
                    const f32x4 a = acc[mt][2 * np], b = acc[mt][2 * np + 1];
                    const unsigned a0 = pack2bf_pk(a[0], a[1]), a1 = pack2bf_pk(a[2], a[3]);
                    const unsigned b0 = pack2bf_pk(b[0], b[1]), b1 = pack2bf_pk(b[2], b[3]);
                    if constexpr (EPI == 3) { if (stats) { stat_add(mt, a0); stat_add(mt, a1); stat_add(mt, b0); stat_add(mt, b1); } }
                    if constexpr (YST) { if (ystats) { stat_sq(mt, a0); stat_sq(mt, a1); stat_sq(mt, b0); stat_sq(mt, b1); } }
                    auto r0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
                    auto r1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
                    // lane rows 0 / 2 hold tile 2 np, columns 0..7 / 8..15; lane rows 1 / 3 tile 2 np + 1
                    const int col = nbase + (2 * np + (q4 & 1)) * 16 + (q4 >> 1) * 8;
                    const u32x4 o = {r0[0], r1[0], r0[1], r1[1]};
                    const int off = col < ncols_out ? row_off + col * 2 : (int)0x80000000u;
                    __builtin_amdgcn_raw_buffer_store_b128(o, rC, off, 0, 0);
                }
                if constexpr (NT % 2 == 1) {  // the unpaired last tile: 8 bytes per lane
                    const f32x4 a = acc[mt][NT - 1];
                    const u32x2_t o = {pack2bf_pk(a[0], a[1]), pack2bf_pk(a[2], a[3])};
                    if constexpr (EPI == 3) { if (stats) { stat_add(mt, o[0]); stat_add(mt, o[1]); } }
                    if constexpr (YST) { if (ystats) { stat_sq(mt, o[0]); stat_sq(mt, o[1]); } }
                    const int col = nbase + (NT - 1) * 16 + 4 * q4;
                    const int off = col < ncols_out ? row_off + col * 2 : (int)0x80000000u;
                    __builtin_amdgcn_raw_buffer_store_b64(o, rC, off, 0, 0);
                }
            } else {
#pragma unroll
                for (int j = 0; j < NT / 4; ++j) {  // 64 input columns = 32 of w1 | 32 of w3 -> 32 output columns
                    unsigned pk[2][2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const f32x4 x = acc[mt][4 * j + u], y = acc[mt][4 * j + 2 + u];
                        // reference rounding points (model.py:497-502 under bf16): w1 x, w3 x, silu, product
                        pk[u][0] = pk_bf(swiglu2(f32x2{x[0], x[1]}, f32x2{y[0], y[1]}));
                        pk[u][1] = pk_bf(swiglu2(f32x2{x[2], x[3]}, f32x2{y[2], y[3]}));
                    }
                    auto r0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
                    auto r1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
                    const int col = nbase / 2 + j * 32 + (q4 & 1) * 16 + (q4 >> 1) * 8;
                    const u32x4 o = {r0[0], r1[0], r0[1], r1[1]};
                    // GemmArgs::pair_c: the output (the W2 projection's A operand) in the row-pair-interleaved layout
                    const int cb = p.pair_c ? row_off_pc + col * 2 + (col >> 5) * 64 : row_off + col * 2;
                    const int off = col < ncols_out ? cb : (int)0x80000000u;
                    __builtin_amdgcn_raw_buffer_store_b128(o, rC, off, 0, 0);
                }
            }
        }
        if constexpr (EPI == 3) {
            // a row's columns of this wave are spread over the four lanes l15 + 16 q4: add them up, lane q4 == 0 stores the pair.
            // ALWAYS MT store instructions per wave and plain tile (lanes / tiles that have nothing to say store out of range): the
            // hand-kept vmcnt of the next tile's first body counts them (NST_Q)
            const long long q_all = (long long)p.M * p.qstat_slots * 8;
            const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc((void*)p.qstat, 0, stats ? (int)(q_all > 0x7fffffffLL ? 0x7fffffffLL : q_all) : 0, 0x00020000);
            const int slot = 2 * (t.n0 / BN) + wn;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float a = st1[mt], b = st2[mt];
                a += __shfl_xor(a, 16, 64); b += __shfl_xor(b, 16, 64);
                a += __shfl_xor(a, 32, 64); b += __shfl_xor(b, 32, 64);
                const int row = t.m0 + wm * (MT * 16) + mt * 16 + l15;
                const u32x2_t o = {__float_as_uint(a), __float_as_uint(b)};
                const int off = (q4 == 0 && row < p.M) ? (row * p.qstat_slots + slot) * 8 : (int)0x80000000u;
                __builtin_amdgcn_raw_buffer_store_b64(o, rS, off, 0, 0);
            }
        }
        if constexpr (YST) {
            // GemmArgs::ystat: sum of squares of this wave's ROUNDED outputs of every row of the tile -> ystat[row][2 * column tile + wn]
            // (what the RMSNorm of the consuming row kernel needs of the row before it can start: norm.hip, GatedResArgs::ystat).  Always
            // MT store instructions per wave and tile, like EPI 3's (NST_Q)
            const long long y_all = (long long)p.M * p.ystat_slots * 4;
            const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc((void*)p.ystat, 0, ystats ? (int)(y_all > 0x7fffffffLL ? 0x7fffffffLL : y_all) : 0, 0x00020000);
            const int slot = 2 * (t.n0 / BN) + wn;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float b = st2[mt];
                b += __shfl_xor(b, 16, 64);
                b += __shfl_xor(b, 32, 64);
                const int row = t.m0 + wm * (MT * 16) + mt * 16 + l15;
                const int off = (q4 == 0 && row < p.M) ? (row * p.ystat_slots + slot) * 4 : (int)0x80000000u;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(b), rS, off, 0, 0);
            }
        }
    };
    // EPI 3, a V tile (column block at or past p.vt_split): the transposed, key-permuted V image vt[b][kv head][d][token'] of the
    // attention kernels (AttnArgs::vt; positions inside every group of 16 tokens: 0-3, 8-11, 4-7, 12-15).  After the swapped MFMAs a
    // lane holds, per 16x16 accumulator tile, column l15 and rows 4 q4 + r = one quad of the group.  v_permlane32_swap of two row
    // tiles' packed registers puts quads (0, 2) resp. (1, 3) of ONE tile side by side: 8 consecutive positions = one 16-byte store.
    auto store_vt = [&](const Tile& t) __attribute__((always_inline)) {
        const int hd = p.vt_hd, kvh = (p.N - p.vt_split) / hd;
        const long long vt_all = (long long)(p.M / p.vt_tokens) * kvh * hd * p.vt_npad * 2;
        const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc((void*)p.VT, 0, (int)(vt_all > 0x7fffffffLL ? 0x7fffffffLL : vt_all), 0x00020000);
        const int lane_e = lane_now(), l15 = lane_e & 15, q4 = lane_e >> 4;  // (see store_out)
        // a pair of 16-row tiles = 32 consecutive tokens lies inside one sample (tokens per sample % 32 == 0, launcher), the 256-row
        // tile need not (Flag-DiT: 4160 tokens per sample); rows past M land past the image's last sample = outside num_records
        int pair_off[MT / 2];
#pragma unroll
        for (int mp = 0; mp < MT / 2; ++mp) {
            const int r = t.m0 + wm * (MT * 16) + mp * 32;
            const int b = r / p.vt_tokens;
            pair_off[mp] = (b * kvh * hd * p.vt_npad + (r - b * p.vt_tokens)) * 2;  // bytes (< 2^31: launcher)
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int c = t.n0 - p.vt_split + wn * (NT * 16) + nt * 16 + l15;
            const int head = c / hd, d = c - head * hd;
            const int col_off = ((head * hd + d) * p.vt_npad + (q4 >> 1) * 16 + (q4 & 1) * 8) * 2;
#pragma unroll
            for (int mp = 0; mp < MT / 2; ++mp) {
                const f32x4 a = acc[2 * mp][nt], bb = acc[2 * mp + 1][nt];
                const unsigned a0 = pack2bf_pk(a[0], a[1]), a1 = pack2bf_pk(a[2], a[3]);
                const unsigned b0 = pack2bf_pk(bb[0], bb[1]), b1 = pack2bf_pk(bb[2], bb[3]);
                auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                // lane rows 0 / 1: row tile 2 mp, positions 0-7 / 8-15; lane rows 2 / 3: row tile 2 mp + 1
                const u32x4 o = {r0[0], r1[0], r0[1], r1[1]};
                __builtin_amdgcn_raw_buffer_store_b128(o, rV, col_off + pair_off[mp], 0, 0);
            }
        }
    };
    auto advance = [&]() __attribute__((always_inline)) {
        if (has_next) {
            cur = nxt;
            v += gridDim.x;
            has_next = v + (int)gridDim.x < nv;
            nxt = has_next ? setup(v + gridDim.x) : t_null;
        }
        if constexpr (GROUPED) {
            if (gather) {  // the new nxt's map -> the slot the stream's crossing into `cur` has just finished with
                map_slot ^= 1;
                map_dma(nxt, map_slot, has_next);
            }
        }
    };
    // one tile: slab s prefetches slab s + PD (the last PD: the next tile's first slabs)
    auto run_tile = [&](auto swap_tag) __attribute__((always_inline)) {
        body(std::true_type{}, swap_tag, wf, af, wf2, af2);
        body(std::false_type{}, swap_tag, wf2, af2, wf, af);
        const int ns_t = (TSPLIT && cur.part >= 0) ? ns >> tsl : ns;  // slabs of this item
        for (int s = 2; s < ns_t; s += 2) {
            body(std::false_type{}, swap_tag, wf, af, wf2, af2);
            body(std::false_type{}, swap_tag, wf2, af2, wf, af);
        }
    };
    // EPI 3: runs of plain tiles and runs of V^T tiles are two separate inner loops.  (As one loop with an if / else per tile the two
    // bodies met at a join in EVERY iteration and the register allocator reconciled their fragment-register assignments through
    // scratch: 68 VGPRs stored after every plain tile's second body and re-loaded - with an s_waitcnt vmcnt(0) behind them - before
    // every V^T tile, profiles/r03 VERDICT.  Now a reconciliation can only sit on the edge between two runs; at cfg 2 a
    // workgroup's tiles are Q, K, V in this order: one edge per launch.)
    // TSPLIT: a part's accumulators -> the workspace (written through to the memory side), the workgroup reports in; true for the LAST part
    // of the tile to arrive, whose accumulators then hold the sum of all TS parts in K order (independent of the arrival order).  The
    // small-M kernels' hand-off (gemm_bf16_pp above: sc0 sc1 payload, asm vmcnt(0), workgroup barrier, system-scope counter).  vmcnt(0)
    // also waits for the next item's slabs in flight - once per tail part, not per tile.
    auto handoff = [&](const Tile& tl) __attribute__((always_inline)) -> bool {
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4_;
        constexpr int PART_BYTES = BM * BN * 4, CP = 17;  // aux bits: sc0 | sc1
        const int TS = 1 << tsl, tile = tl.part >> tsl, mine = tl.part & (TS - 1);
        const int lane_e = lane_now(), tid_e = wave * 64 + lane_e;
        const __amdgpu_buffer_rsrc_t rAll = __builtin_amdgcn_make_buffer_rsrc((void*)(p.tail_part + (size_t)tile * TS * (BM * BN)), 0, TS * PART_BYTES, 0x00020000);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, acc[mt][nt]), rAll, ((mt * NT + nt) * 256 + tid_e) * 16, mine * PART_BYTES, CP);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave's partial has reached the memory side ...
        __syncthreads();                                   // ... before the workgroup reports in
        int* flag = g_list + 1026;  // (behind the tile list and its count; the slab ring holds the next item's slabs)
        if (tid_e == 0) {
            const unsigned old = __hip_atomic_fetch_add(p.tail_cnt + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (old == (unsigned)(TS - 1)) __hip_atomic_store(p.tail_cnt + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // all arrived: free for the next launch
            *flag = (int)old;
        }
        __syncthreads();
        const bool last = __builtin_amdgcn_readfirstlane(*flag) == TS - 1;
        __syncthreads();  // (the flag word is free again before a later part of this workgroup writes it)
        if (!last) return false;
        // parts in K order: ((p0 + p1) + p2) + p3, the own part re-read like the others (a select "own part from its registers at its own
        // position" made the compiler hold all 64 accumulator tiles in VGPRs: 151 spilled registers), a part past TS reads out of range =
        // zeros.  Six accumulator tiles' loads (24 x 16 bytes per lane) in flight between fences: two made the sum latency-bound (+47 us
        // per launch, the first form of this code)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x4 o[4];
#pragma unroll
                for (int part = 0; part < 4; ++part) {
                    const int off = ((mt * NT + nt) * 256 + tid_e) * 16 + (part < TS ? 0 : (int)0x40000000);
                    o[part] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rAll, off, part * PART_BYTES, CP));
                }
                acc[mt][nt] = ((o[0] + o[1]) + o[2]) + o[3];
                asm volatile("" : "+a"(acc[mt][nt]));  // back to its AGPR home right away (GROUPED: NT = 8, all 64 tiles live in AGPRs): as plain values the 64 sums sat in VGPRs and spilled
                if ((mt * NT + nt) % 6 == 5) __builtin_amdgcn_sched_barrier(0);
            }
        return true;
    };
    int t = 0;
    auto plain_tile = [&]() __attribute__((always_inline)) {
        run_tile(std::false_type{});
        if constexpr (TRACE) { tr_loop = __builtin_amdgcn_s_memrealtime(); if (t == 0) tr_clk = __builtin_amdgcn_s_memtime() - tr_clk; }
        bool emit = true;
        if constexpr (TSPLIT) { if (cur.part >= 0) emit = handoff(cur); }
        if (emit) store_out(cur);
        after_epilogue = emit ? 1 : 0;
        if constexpr (TRACE) tr_epi = __builtin_amdgcn_s_memrealtime();
        advance();
        ++t;
    };
    if constexpr (EPI == 3) {
        while (t < my_tiles) {
            while (t < my_tiles && cur.n0 < p.vt_split) plain_tile();
            while (t < my_tiles && cur.n0 >= p.vt_split) {
                run_tile(std::true_type{});
                if constexpr (TRACE) { tr_loop = __builtin_amdgcn_s_memrealtime(); if (t == 0) tr_clk = __builtin_amdgcn_s_memtime() - tr_clk; }
                store_vt(cur);
                after_epilogue = 2;
                if constexpr (TRACE) tr_epi = __builtin_amdgcn_s_memrealtime();
                advance();
                ++t;
            }
        }
    } else {
        while (t < my_tiles) plain_tile();
    }
    wait_vmcnt<0>();  // no LDS-DMA (the null ones of the last bodies included) may outlive the workgroup's LDS allocation
    if constexpr (TRACE) {
        if (p.trace && tid == 0) {
            unsigned long long* o = p.trace + (size_t)blockIdx.x * 8;
            o[0] = tr_entry; o[1] = tr_pro; o[2] = tr_loop; o[3] = tr_epi; o[4] = __builtin_amdgcn_s_memrealtime(); o[5] = tr_clk;
            o[6] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) |       // HW_ID
                   ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 32);  // XCC_ID[3:0]
            o[7] = (unsigned long long)my_tiles;
        }
    }
}

}  // namespace lt_gemm
