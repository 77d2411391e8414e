// bf16 MFMA GEMM for gfx950:  C[M,N] = A[M,K] * W[N,K]^T (+bias) ; fp32 accumulate, bf16 out.
//
// Replaces every fairscale Column/RowParallelLinear == F.linear at mp=1 on the denoising path
// (lumina_next_t2i/models/model.py:165-209 wq/wk/wv/wo, :475-495 w1/w2/w3).
//
// Design (CDNA4-first, not a CUDA tiling):
//  * 256x256x64 macro tile, 8 waves (2 along M x 4 along N), each wave owns 128x64 of C as
//    4x2 tiles of v_mfma_f32_32x32x16_bf16 (128 accumulator registers / lane); or 256x288x64 with
//    12 waves (4 x 3, 64x96 per wave) where that removes tile-count quantisation over the 256 CUs.
//  * A and W tiles go HBM -> LDS with buffer_load ... lds (16 B / lane, no VGPR round trip),
//    double-buffered (2 x 64 KiB of the CU's 160 KiB).  The LDS image is lane-linear, so the bank
//    swizzle is applied to the per-lane SOURCE address and again on the ds_read_b128 (guide rule 21):
//    chunk' = chunk ^ ((row >> 1) & 7) makes the 32x32x16 fragment reads conflict-free.
//  * Buffer descriptors carry the row bound: rows >= M (or >= N for W) read as zero and their
//    stores are predicated, so M and N need not be tile multiples.  K must be a multiple of 64.
//  * The MFMA is issued as D'[n][m] = W_frag x A_frag so that each lane ends up holding 4 consecutive
//    N for one row of C; a v_permlane32_swap pair widens that to one 16-byte store per lane.
//  * Epilogue 1 fuses SwiGLU (model.py:497-502): W is w1/w3 interleaved in 32-row groups, so the
//    two accumulator tiles of a wave hold silu-input and gate for the same (m, n).
//  * Workgroup -> tile map is XCD-aware: each XCD (private 4 MiB L2) gets a contiguous run of a
//    grouped (4 tile-rows, column-major) order so neighbouring tiles share A/W panels in one L2.
#include "common.h"
#include "kernels.h"
#include <hip/hip_ext.h>
#include <algorithm>
#include <type_traits>

namespace {

constexpr int BK = 64;

__device__ __forceinline__ void tile_coords(int bid, int nwg, int TM, int TN, int& tm, int& tn) {
    const int NX = 8;
    const int xcd = bid % NX, idx = bid / NX;
    const int q = nwg / NX, r = nwg % NX;
    const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int G = 4;
    const int per_group = G * TN;
    const int g = L / per_group;
    const int first_m = g * G;
    const int gsz = min(G, TM - first_m);
    const int in = L - g * per_group;
    tm = first_m + in % gsz;
    tn = in / gsz;
}

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
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        // reference rounding points (model.py:497-502 under bf16): w1 x, w3 x, silu, product
                        const float a = bfr(acc[mt][2 * np][8 * qp + j]);
                        const float b = bfr(acc[mt][2 * np + 1][8 * qp + j]);
                        v[j] = bfr(silu_f(a)) * b;
                    }
                    unsigned ax = pack2bf_pk(v[0], v[1]), ay = pack2bf_pk(v[2], v[3]);
                    unsigned bx = pack2bf_pk(v[4], v[5]), by = pack2bf_pk(v[6], v[7]);
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
    static_assert(EPI == 0 || NT % 2 == 0, "SwiGLU epilogue pairs accumulator tiles");
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
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nt], af[mt], acc[mt][nt], 0, 0, 0);
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
// ((b >> 3) & 7) * n * ~1024 cycles before its first load, which spreads the CUs of an XCD over eight tile phases.  All tiles of
// a GEMM take the same time, so without it every CU of the chip is in its prologue / epilogue at the same moment; whether that
// synchronised idle phase is what keeps the clock low is one of the next round's questions.  A __device__ word
// instead of a GemmArgs field: the kernels of the product path do not read it and keep their argument layout.
__device__ int g_dev_gemm_stagger = 0;
__device__ __forceinline__ void stagger_start() {
    const int n = g_dev_gemm_stagger;
    if (n > 0) {
        const int reps = ((blockIdx.x >> 3) & 7) * n;
        for (int i = 0; i < reps; ++i) __builtin_amdgcn_s_sleep(16);
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
    static_assert(EPI == 0 || NT % 2 == 0, "SwiGLU epilogue pairs accumulator tiles");
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
    tile_coords(blockIdx.x, gridDim.x, TM, TN, tm, tn);
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
        rs[i] = __builtin_amdgcn_make_buffer_rsrc((void*)(isA ? a_base : w_base), 0, isA ? a_bytes : w_bytes, 0x00020000);
        voff[i] = r0 * (isA ? p.lda : p.ldw) * 2 + (((lane % LPR) ^ key) << 4);
        ldsoff[i] = q * 1024;
    }
    auto stage = [&](int slab) {
        char* base = smem + (slab & 3) * SLAB;
        const int soff = slab * RB;
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

    const int ns = p.K / (16 * KS);
    if constexpr (MODE == 2 && G == 1) stagger_start();  // 4-wave kernel only (variant 10)
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
        const int soff = (s + 3) * RB;
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
            const int soff = (k + 3) * RB;
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

    store_tile<MT, NT, EPI>(acc, p, m0, n0, wm, wn, hi, l31);

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

// ---- persistent 4-wave kernel (EXPERIMENTAL, variant 13 - written without GPU time at the end of round 1) -----------------
// Variant 10's steady-state body (one wave per SIMD, 128x128 per wave, LDS-DMA, fragments double-buffered in registers, one
// barrier per slab: 83.5 % matrix-pipe duty in the s_memtime trace) loses 16 % of every tile at its two ends - a cold 3-slab
// prologue (5.3 k ticks) and an epilogue (12.7 k) with nothing else resident on the CU (profiles/r01/gemm_trace_4wave_native.log).
// Here one workgroup per CU walks its tiles with the slab stream running across tile boundaries, as in gemm_bf16_pp_persist:
// the last three bodies of a tile issue the LDS-DMA of the next tile's slabs 0..2 and the last body reads the next tile's
// first fragments, so a boundary is: epilogue stores (issued, not waited for), accumulators cleared, next body.
//   * the epilogue stores go through a buffer descriptor over the tile's C rows (rows past M fall outside num_records, columns
//     past N get an out-of-range offset): EVERY wave issues exactly NST store instructions per tile, so the one vmcnt literal
//     that has to let them pass (first body after a boundary) is exact - a branchy `if (m < M)` store could issue fewer and
//     the wait would then be too weak;
//   * slabs are consumed in pairs (fragment register sets alternate), so K % 64 == 0; K >= 128.  No bias epilogue.
//   * OVL (variant 14): a tile's epilogue is not a phase of its own but rides in the FIRST body of the next tile - that body's
//     k-step-0 MFMAs take the constant 0 as accumulator input, each right after the old contents of its 32x32 accumulator tile
//     were copied out, and the pack / store of that tile issues behind the MFMA.  Only the workgroup's last tile stores the plain way.
template <int EPI, bool OVL = false>
__global__ __launch_bounds__(256, 1) void gemm_bf16_w4p(GemmArgs p) {
    constexpr int MT = 4, NT = 4, NW = 4, BM = 256, BN = 256, IP = 8;
    constexpr int SLAB = (BM + BN) * 64, W_OFF = BM * 64, TSTRIDE = 2048;
    constexpr int NM = 2 * MT * NT, RD = 2 * (MT + NT);
    constexpr int NST = EPI == 0 ? MT * NT * 2 : MT * (NT / 2) * 2;  // 16-byte stores per wave and tile
    static_assert(NM == 4 * IP && RD == 2 * IP, "body: one read per MFMA in the first half, one DMA per four MFMAs");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int hi = lane >> 5, l31 = lane & 31;
    const int TM = (p.M + BM - 1) / BM, TN = (p.N + BN - 1) / BN;
    const int ntiles = TM * TN;
    const int ns = p.K / 32;

    // staging: wave w copies pieces w + 4 i: i < 4 rows 16 w + 64 i .. of A, i >= 4 the same rows of W (tile-independent
    // per-lane offsets; a tile only changes the descriptors)
    const int sswz = ((lane & 3) ^ ((lane >> 4) & 3)) * 16;
    const int srow = 16 * wave + (lane >> 2);
    int voff[IP], ldsoff[IP];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        voff[i] = (srow + 64 * i) * p.lda * 2 + sswz;
        voff[4 + i] = (srow + 64 * i) * p.ldw * 2 + sswz;
    }
#pragma unroll
    for (int i = 0; i < IP; ++i) ldsoff[i] = (wave + NW * i) * 1024;
    const int ncols_out = EPI == 0 ? p.N : p.N / 2;
    // a tile = its A / W / C panel pointers and the bytes left in each panel (descriptor num_records: rows past M / N read as
    // zero, C rows past M are not written).  Plain scalars: the body picks "this tile" or "the next one" with scalar selects and
    // builds the descriptor on the spot, so every slab of a tile runs through ONE loop body (a separate copy of the body for the
    // last slabs made the compiler re-shuffle half of the accumulator registers between the two copies)
    struct Tile { const u16* a; const u16* w; u16* c; int a_bytes, w_bytes, c_bytes, n0; };
    auto setup = [&](int v) __attribute__((always_inline)) {
        int tm, tn;
        tile_coords(v, ntiles, TM, TN, tm, tn);
        const int m0 = tm * BM, n0_ = tn * BN;
        const long long a_left = (long long)(p.M - m0) * p.lda * 2;
        const long long w_left = (long long)(p.N - n0_) * p.ldw * 2;
        const long long c_left = (long long)(p.M - m0) * p.ldc * 2;
        Tile t;
        t.a = p.A + (size_t)m0 * p.lda; t.w = p.W + (size_t)n0_ * p.ldw; t.c = p.C + (size_t)m0 * p.ldc;
        t.a_bytes = (int)(a_left > 0x7fffffffLL ? 0x7fffffffLL : a_left);
        t.w_bytes = (int)(w_left > 0x7fffffffLL ? 0x7fffffffLL : w_left);
        t.c_bytes = (int)(c_left > 0x7fffffffLL ? 0x7fffffffLL : c_left);
        t.n0 = n0_;
        return t;
    };
    // no next tile: the last three bodies still issue their LDS-DMA (one code path) from empty descriptors - every lane is out of
    // range, the ring slots they zero-fill hold slabs that were consumed already
    const Tile t_null = {p.A, p.W, p.C, 0, 0, 0, 0};

    const int fswz = (l31 >> 2) & 3;
    const int a_row_off = (wm * MT * 32 + l31) * 64;
    const int w_row_off = W_OFF + (wn * NT * 32 + l31) * 64;
    int coff[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) coff[k] = ((2 * k + hi) ^ fswz) << 4;

    f32x16 acc[MT][NT];
    bf16x8 wf[2][NT], af[2][MT], wf2[2][NT], af2[2][MT];

    int v = blockIdx.x;
    if (v >= ntiles) return;  // uniform
    const int my_tiles = (ntiles - 1 - v) / (int)gridDim.x + 1;
    Tile cur = setup(v);
    bool has_next = v + (int)gridDim.x < ntiles;
    Tile nxt = has_next ? setup(v + gridDim.x) : t_null;

    auto stage_from = [&](int g, int slab_in_tile, const Tile& t) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)t.a, 0, t.a_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)t.w, 0, t.w_bytes, 0x00020000);
        char* base = smem + (g & 3) * SLAB;
        const int soff = slab_in_tile * 64;
#pragma unroll
        for (int i = 0; i < IP; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(i < 4 ? rA : rW, LDS_PTR(base + ldsoff[i]), 16, voff[i], soff, 0, 0);
    };
    stagger_start();
    // prologue (once per workgroup): slabs 0..2 in flight, slab 0 read into the first fragment set, slab 1 landed and visible
    stage_from(0, 0, cur);
    stage_from(1, 1, cur);
    stage_from(2, 2, cur);
    wait_vmcnt<2 * IP>();
    pp_barrier();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wf[k][nt] = *(const bf16x8*)(smem + w_row_off + nt * TSTRIDE + coff[k]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) af[k][mt] = *(const bf16x8*)(smem + a_row_off + mt * TSTRIDE + coff[k]);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    wait_vmcnt<IP>();
    pp_barrier();

    int g = 0;               // global slab index of the stream
    bool after_epilogue = false;
    // one slab: MFMAs of slab g from (wc, ac) | fragment reads of slab g+1 into (wn_, an) | LDS-DMA of the slab three ahead
    // (slab_in_tile of the tile rA / rW describe) - the instruction mix of gemm_bf16_pp's body_pinned
    auto body = [&](int s3, bf16x8 (&wc)[2][NT], bf16x8 (&ac)[2][MT], bf16x8 (&wn_)[2][NT], bf16x8 (&an)[2][MT]) __attribute__((always_inline)) {
        // s3 = in-tile index of the slab three ahead; past the tile's end it is slab s3 - ns of the next tile
        const bool own = s3 < ns;
        const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)(own ? cur.a : nxt.a), 0, own ? cur.a_bytes : nxt.a_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)(own ? cur.w : nxt.w), 0, own ? cur.w_bytes : nxt.w_bytes, 0x00020000);
        const char* sb = smem + ((g + 1) & 3) * SLAB;
        char* db = smem + ((g + 3) & 3) * SLAB;
        const int soff = (own ? s3 : s3 - ns) * 64;
        auto rd = [&](int r) __attribute__((always_inline)) {
            const int kk = r / (MT + NT), j = r % (MT + NT);
            if (j < NT) wn_[kk][j] = *(const bf16x8*)(sb + w_row_off + j * TSTRIDE + coff[kk]);
            else an[kk][j - NT] = *(const bf16x8*)(sb + a_row_off + (j - NT) * TSTRIDE + coff[kk]);
        };
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            const int kk = i / (MT * NT), mt = (i / NT) % MT, nt = i % NT;
            // OVL: every MFMA of the kernel is the in-place inline-assembly form, so that the accumulators stay in one fixed set
            // of AGPRs through the boundary body as well (mixing it with the builtin made the allocator move them around the
            // loop); the written order is pinned by scheduling fences instead of sched_group_barrier masks
            if constexpr (OVL) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[mt][nt]) : "v"(wc[kk][nt]), "v"(ac[kk][mt]));
            else acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[kk][nt], ac[kk][mt], acc[mt][nt], 0, 0, 0);
            if (i < RD) rd(i);
            if (i % 4 == 3)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(i / 4 < 4 ? rA : rW, LDS_PTR(db + ldsoff[i / 4]), 16, voff[i / 4], soff, 0, 0);
            if constexpr (OVL) __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (!OVL) {
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
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): slab g+1's fragments are in registers
        // slab g+2 landed; still allowed in flight: this body's 8 DMAs and, right after a tile boundary, the NST stores
        // issued between slab g+2's DMAs and them (loads and stores retire in issue order, one counter)
        if (after_epilogue) wait_vmcnt<IP + NST>();
        else wait_vmcnt<IP>();
        after_epilogue = false;
        pp_barrier();
        ++g;
    };
    // epilogue through the tile's C descriptor: lane holds, per 32x32 MFMA tile, row l31 and columns 8 q + 4 hi + j (as store_tile).
    // emit: the two 16-byte stores of output group (mt, ng) - EPI 0: accumulator tile (mt, ng) in `x`; EPI 1: silu(x) * y of the
    // tile pair (mt, 2 ng), (mt, 2 ng + 1)
    auto emit = [&](__amdgpu_buffer_rsrc_t rC, int n0_, int mt, int ng, const f32x16& x, const f32x16& y) __attribute__((always_inline)) {
        const int row_off = (wm * MT * 32 + mt * 32 + l31) * p.ldc * 2;  // bytes from the tile's first C row (< 2^31: launcher)
        const int cbase = EPI == 0 ? n0_ + wn * NT * 32 + ng * 32 : (n0_ + wn * NT * 32 + ng * 64) / 2;
#pragma unroll
        for (int qp = 0; qp < 2; ++qp) {
            float vv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if constexpr (EPI == 0) {
                    vv[j] = x[8 * qp + j];
                } else {  // reference rounding points (model.py:497-502 under bf16): w1 x, w3 x, silu, product
                    const float a = bfr(x[8 * qp + j]);
                    const float b = bfr(y[8 * qp + j]);
                    vv[j] = bfr(silu_f(a)) * b;
                }
            }
            unsigned ax = pack2bf_pk(vv[0], vv[1]), ay = pack2bf_pk(vv[2], vv[3]);
            unsigned bx = pack2bf_pk(vv[4], vv[5]), by = pack2bf_pk(vv[6], vv[7]);
            auto r0 = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
            auto r1 = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
            const int col = cbase + 16 * qp + 8 * hi;
            const u32x4 o = {r0[0], r1[0], r0[1], r1[1]};
            // columns past the end: an offset no descriptor covers (the store is issued and dropped)
            const int off = col < ncols_out ? row_off + col * 2 : (int)0x80000000u;
            __builtin_amdgcn_raw_buffer_store_b128(o, rC, off, 0, 0);
        }
    };
    auto store_out = [&](const Tile& t) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc((void*)t.c, 0, t.c_bytes, 0x00020000);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            constexpr int NG = EPI == 0 ? NT : NT / 2;
#pragma unroll
            for (int ng = 0; ng < NG; ++ng) {
                if constexpr (EPI == 0) emit(rC, t.n0, mt, ng, acc[mt][ng], acc[mt][ng]);
                else emit(rC, t.n0, mt, ng, acc[mt][2 * ng], acc[mt][2 * ng + 1]);
            }
        }
    };
    // OVL: first body of a tile whose predecessor `done` still sits in the accumulators (see the kernel comment)
    auto body_first = [&](const Tile& done, bf16x8 (&wc)[2][NT], bf16x8 (&ac)[2][MT], bf16x8 (&wn_)[2][NT], bf16x8 (&an)[2][MT]) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)cur.a, 0, cur.a_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)cur.w, 0, cur.w_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc((void*)done.c, 0, done.c_bytes, 0x00020000);
        const char* sb = smem + ((g + 1) & 3) * SLAB;
        char* db = smem + ((g + 3) & 3) * SLAB;
        const int soff = 3 * 64;  // this tile's slab 3 (ns >= 4)
        f32x16 keep;
#pragma unroll
        for (int r = 0; r < 16; ++r) keep[r] = 0.f;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            const int kk = i / (MT * NT), mt = (i / NT) % MT, nt = i % NT;
            if (kk == 0) {
                // copy-out, pinned in front of its MFMA by an empty volatile asm that wants the copy in VGPRs right here (left to
                // itself the allocator hoisted the reads of eleven tiles to the top of the body and spilled around them)
                f32x16 old = acc[mt][nt];
                asm volatile("" : "+v"(old));
                // in place ("+a": same registers in and out, although the instruction only writes them) - the builtin form let
                // the allocator put the new tile into a different register tuple, and the loop then paid for rotating them back
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "+a"(acc[mt][nt]) : "v"(wc[0][nt]), "v"(ac[0][mt]));
                if constexpr (EPI == 0) {
                    emit(rC, done.n0, mt, nt, old, old);
                } else {
                    if ((nt & 1) == 0) keep = old;
                    else emit(rC, done.n0, mt, nt >> 1, keep, old);
                }
            } else {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[mt][nt]) : "v"(wc[kk][nt]), "v"(ac[kk][mt]));
            }
            if (i < RD) {
                const int rk = i / (MT + NT), j = i % (MT + NT);
                if (j < NT) wn_[rk][j] = *(const bf16x8*)(sb + w_row_off + j * TSTRIDE + coff[rk]);
                else an[rk][j - NT] = *(const bf16x8*)(sb + a_row_off + (j - NT) * TSTRIDE + coff[rk]);
            }
            if (i % 4 == 3)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(i / 4 < 4 ? rA : rW, LDS_PTR(db + ldsoff[i / 4]), 16, voff[i / 4], soff, 0, 0);
            __builtin_amdgcn_sched_barrier(0);  // one accumulator tile at a time (register budget), in this order
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        wait_vmcnt<IP + NST>();  // slab g+2 landed; younger: this body's 8 DMAs and NST stores, in whatever interleaving
        pp_barrier();
        ++g;
    };

    auto clear_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    auto advance = [&]() __attribute__((always_inline)) {
        if (has_next) {
            cur = nxt;
            v += gridDim.x;
            has_next = v + (int)gridDim.x < ntiles;
            nxt = has_next ? setup(v + gridDim.x) : t_null;
        }
    };
    if constexpr (OVL) {
        clear_acc();
        // ONE code path for every tile (two copies of the loop made the allocator permute the accumulator tuples between them): the
        // first tile runs the boundary body too, "storing" the cleared accumulators through the empty descriptor of t_null
        Tile done = t_null;
        for (int t = 0; t < my_tiles; ++t) {
            body_first(done, wf, af, wf2, af2);
            body(4, wf2, af2, wf, af);
            for (int s = 2; s < ns; s += 2) {
                body(s + 3, wf, af, wf2, af2);
                body(s + 4, wf2, af2, wf, af);
            }
            done = cur;
            advance();
        }
        store_out(done);  // the workgroup's last tile
    } else {
        for (int t = 0; t < my_tiles; ++t) {
            clear_acc();
            for (int s = 0; s < ns; s += 2) {  // slab s prefetches slab s + 3 (the last three: the next tile's slabs 0, 1, 2)
                body(s + 3, wf, af, wf2, af2);
                body(s + 4, wf2, af2, wf, af);
            }
            store_out(cur);
            after_epilogue = true;
            advance();
        }
    }
    wait_vmcnt<0>();  // no LDS-DMA (the null ones of the last bodies included) may outlive the workgroup's LDS allocation
}

// explicit instantiations (hipcc 7.2 does not emit the kernel body for address-only uses inside another template)
template __global__ void gemm_bf16_tn<2, 4, 4, 2, 0>(GemmArgs);
template __global__ void gemm_bf16_tn<2, 4, 4, 2, 1>(GemmArgs);
template __global__ void gemm_bf16_tn<4, 3, 2, 3, 0>(GemmArgs);
template __global__ void gemm_bf16_pp<2, 4, 4, 2, 0>(GemmArgs);
template __global__ void gemm_bf16_pp<2, 4, 4, 2, 1>(GemmArgs);
template __global__ void gemm_bf16_pp<4, 3, 2, 3, 0>(GemmArgs);
template __global__ void gemm_bf16_pp<2, 4, 4, 2, 0, true>(GemmArgs);
template __global__ void gemm_bf16_pp<2, 2, 4, 4, 0, false, 0, 2, 2>(GemmArgs);  // 4 waves x (128 x 128): one wave per SIMD
template __global__ void gemm_bf16_pp<2, 2, 4, 4, 1, false, 0, 2, 2>(GemmArgs);
template __global__ void gemm_bf16_pp<2, 2, 4, 4, 0, true, 0, 2, 2>(GemmArgs);  // ... its trace build
template __global__ void gemm_bf16_pp<2, 4, 4, 2, 0, false, 0, 0, 2, true>(GemmArgs);  // AGPR accumulators (variant 11)
template __global__ void gemm_bf16_pp<2, 4, 4, 2, 1, false, 0, 0, 2, true>(GemmArgs);
template __global__ void gemm_bf16_w4s<0>(GemmArgs);
template __global__ void gemm_bf16_w4s<1>(GemmArgs);
template __global__ void gemm_bf16_w4s<0, true>(GemmArgs);
template __global__ void gemm_bf16_w4p<0>(GemmArgs);
template __global__ void gemm_bf16_w4p<1>(GemmArgs);
template __global__ void gemm_bf16_w4p<0, true>(GemmArgs);
template __global__ void gemm_bf16_w4p<1, true>(GemmArgs);
template __global__ void gemm_bf16_pp_persist<0>(GemmArgs);
template __global__ void gemm_bf16_pp_persist<1>(GemmArgs);
template __global__ void gemm_bf16_pp<2, 4, 2, 1, 0>(GemmArgs);  // 128 x 128, small-M problems
template __global__ void gemm_bf16_pp<4, 2, 1, 2, 1>(GemmArgs);  // 128 x 128 with the SwiGLU epilogue (needs NT even)
template __global__ void gemm_bf16_pp<2, 4, 1, 1, 0>(GemmArgs);  //  64 x 128
template __global__ void gemm_bf16_pp<2, 4, 2, 1, 0, false, 0, 2, 4>(GemmArgs);  // ... register-pipelined, one barrier per slab
template __global__ void gemm_bf16_pp<4, 2, 1, 2, 1, false, 0, 2, 4>(GemmArgs);
template __global__ void gemm_bf16_pp<2, 4, 1, 1, 0, false, 0, 2, 4>(GemmArgs);
template __global__ void gemm_bf16_pp<2, 4, 2, 1, 0, false, 0, 1, 4>(GemmArgs);  // ... and with one barrier per (64-deep) slab
template __global__ void gemm_bf16_pp<4, 2, 1, 2, 1, false, 0, 1, 4>(GemmArgs);
template __global__ void gemm_bf16_pp<2, 4, 1, 1, 0, false, 0, 1, 4>(GemmArgs);
template __global__ void gemm_bf16_pp<2, 4, 4, 2, 0, false, 0, 1>(GemmArgs);
template __global__ void gemm_bf16_pp<2, 4, 4, 2, 0, true, 0, 1>(GemmArgs);
template __global__ void gemm_bf16_pp<2, 4, 4, 2, 1, false, 0, 1>(GemmArgs);
template __global__ void gemm_bf16_pp<4, 3, 2, 3, 0, false, 0, 1>(GemmArgs);
template __global__ void gemm_bf16_pp<2, 4, 4, 2, 0, false, 3>(GemmArgs);
template __global__ void gemm_bf16_pp<2, 4, 4, 2, 1, false, 3>(GemmArgs);
template __global__ void gemm_bf16_pp<2, 4, 4, 2, 0, true, 3>(GemmArgs);
template __global__ void gemm_bf16_pp<4, 3, 2, 3, 0, true>(GemmArgs);

}  // namespace lt_gemm

namespace {
using lt_gemm::gemm_bf16_tn;
using lt_gemm::gemm_bf16_pp;
using lt_gemm::gemm_bf16_pp_persist;
using lt_gemm::gemm_bf16_w4s;
using lt_gemm::gemm_bf16_w4p;

// w1/w3 -> 32-row interleaved packed weight (row P: block = P/64; P%64 < 32 -> w1 else w3)
__global__ void pack_w13_kernel(const u16* __restrict__ w1, const u16* __restrict__ w3, u16* __restrict__ out,
                                int F, int K) {
    const int chunks_per_row = K / 8;
    const long long total = (long long)2 * F * chunks_per_row;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int P = (int)(i / chunks_per_row), c = (int)(i % chunks_per_row);
        const int blk = P >> 6, w = P & 63;
        const u16* src = (w < 32 ? w1 : w3) + (size_t)(blk * 32 + (w & 31)) * K + c * 8;
        *(bf8_t*)(out + (size_t)P * K + c * 8) = *(const bf8_t*)src;
    }
}

}  // namespace

namespace {
// ev0 / ev1 (optional): start / stop events attached to THIS dispatch packet (hipExtLaunchKernelGGL) - the timestamps come
// from the dispatch's own completion signal, no extra barrier packets in the queue (event records around a launch cost
// tens of microseconds of queue idle time each on this stack)
template <int WM, int WN, int MT, int NT, int EPI, bool PP, int TAIL = 0, int MODE = 0, int KS = 2, bool AGPR = false>
int launch_cfg(const GemmArgs& a, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    constexpr int SMEM = PP ? 4 * (BM + BN) * 32 * KS : 2 * (BM + BN) * 128;
    const void* fn;
    if constexpr (PP) fn = (const void*)gemm_bf16_pp<WM, WN, MT, NT, EPI, false, TAIL, MODE, KS, AGPR>;
    else fn = (const void*)gemm_bf16_tn<WM, WN, MT, NT, EPI>;
    static bool attr_done = false;
    if (!attr_done) {
        LT_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
        attr_done = true;
    }
    const int TM = (a.M + BM - 1) / BM, TN = (a.N + BN - 1) / BN;
    const dim3 grid(TM * TN), block(WM * WN * 64);
    if constexpr (PP) {
        if (ev0) hipExtLaunchKernelGGL((gemm_bf16_pp<WM, WN, MT, NT, EPI, false, TAIL, MODE, KS, AGPR>), grid, block, SMEM, stream, ev0, ev1, 0, a);
        else hipLaunchKernelGGL((gemm_bf16_pp<WM, WN, MT, NT, EPI, false, TAIL, MODE, KS, AGPR>), grid, block, SMEM, stream, a);
    } else {
        if (ev0) hipExtLaunchKernelGGL((gemm_bf16_tn<WM, WN, MT, NT, EPI>), grid, block, SMEM, stream, ev0, ev1, 0, a);
        else hipLaunchKernelGGL((gemm_bf16_tn<WM, WN, MT, NT, EPI>), grid, block, SMEM, stream, a);
    }
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}

int num_cus();
template <int EPI, bool TRACE = false>
int launch_w4s(const GemmArgs& a, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    constexpr int SMEM = 2 * 512 * 64;
    static bool attr_done = false;
    if (!attr_done) {
        LT_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bf16_w4s<EPI, TRACE>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
        attr_done = true;
    }
    const dim3 grid(((a.M + 255) / 256) * ((a.N + 255) / 256)), block(256);
    if (ev0) hipExtLaunchKernelGGL((gemm_bf16_w4s<EPI, TRACE>), grid, block, SMEM, stream, ev0, ev1, 0, a);
    else hipLaunchKernelGGL((gemm_bf16_w4s<EPI, TRACE>), grid, block, SMEM, stream, a);
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}
template <int EPI, bool OVL = false>
int launch_w4p(const GemmArgs& a, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    constexpr int SMEM = 4 * 512 * 64;
    static bool attr_done = false;
    if (!attr_done) {
        LT_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bf16_w4p<EPI, OVL>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
        attr_done = true;
    }
    const int tiles = ((a.M + 255) / 256) * ((a.N + 255) / 256);
    const int cus = num_cus();
    const dim3 grid(tiles < cus ? tiles : cus), block(256);
    if (ev0) hipExtLaunchKernelGGL((gemm_bf16_w4p<EPI, OVL>), grid, block, SMEM, stream, ev0, ev1, 0, a);
    else hipLaunchKernelGGL((gemm_bf16_w4p<EPI, OVL>), grid, block, SMEM, stream, a);
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}
template <int EPI>
int launch_persist(const GemmArgs& a, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    constexpr int SMEM = 4 * 512 * 64;
    static bool attr_done = false;
    if (!attr_done) {
        LT_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bf16_pp_persist<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
        attr_done = true;
    }
    const int ntiles = ((a.M + 255) / 256) * ((a.N + 255) / 256);
    const dim3 grid(std::min(ntiles, num_cus())), block(512);
    if (ev0) hipExtLaunchKernelGGL((gemm_bf16_pp_persist<EPI>), grid, block, SMEM, stream, ev0, ev1, 0, a);
    else hipLaunchKernelGGL((gemm_bf16_pp_persist<EPI>), grid, block, SMEM, stream, a);
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}

int num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}
}  // namespace

static int g_gemm_variant = 0;   // tile shape when the caller passes 0: 0 auto, 1 = 256x256, 2 = 256x288
static int g_gemm_pipeline = 0;  // variant <= 2: 0 auto (ping-pong for the SwiGLU GEMM, classic elsewhere), 1 ping-pong, 2 classic, 3 rendezvous
void lt_set_gemm_variant(int v) { g_gemm_variant = v; }
void lt_set_gemm_pipeline(int v) { g_gemm_pipeline = v; }
static int g_gemm_persist = 0;   // 1: SwiGLU GEMMs with >= 2 tile rounds per CU run on the persistent ping-pong kernel
void lt_set_gemm_persist(int v) { g_gemm_persist = v; }
static int g_gemm_pp_tail = 0;   // 8-wave ping-pong kernel: 1 = tail MFMAs issued after the hand-over (TAILN = 3; measured no gain), 0 = plain
void lt_set_gemm_pp_tail(int v) { g_gemm_pp_tail = v; }
int lt_set_gemm_stagger(int v) {  // device-side word (see stagger_start)
    LT_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(lt_gemm::g_dev_gemm_stagger), &v, sizeof(int)));
    return 0;
}

// variant: 0 = pick the tile shape that minimises (rounds over the CUs) x (tile width); 1 = 256x256; 2 = 256x288;
//          3 / 4 = the same two shapes with the ping-pong kernel regardless of the process-wide pipeline option;
//          5 / 6 = the same two shapes with the single-barrier rendezvous kernel;
//          7 / 8 = 128x128 / 64x128 ping-pong tiles (picked automatically when 256-wide tiles would fill < half the CUs)
int launch_gemm_bf16(const GemmArgs& a, int epilogue, int variant, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    LT_REQUIRE(a.K % BK == 0 && a.K > 0, "gemm: K=%d must be a positive multiple of %d", a.K, BK);
    LT_REQUIRE(a.N % 8 == 0 && a.ldc % 8 == 0, "gemm: N=%d and ldc=%d must be multiples of 8", a.N, a.ldc);
    LT_REQUIRE(a.lda % 8 == 0 && a.ldw % 8 == 0, "gemm: lda/ldw must be multiples of 8");
    LT_REQUIRE(epilogue == 0 || (a.N % 64 == 0 && a.bias_dtype < 0), "gemm: swiglu epilogue needs N %% 64 == 0, no bias");
    LT_REQUIRE(variant >= 0 && variant <= 14, "gemm: unknown variant %d", variant);
    if (variant == 13 || variant == 14) {  // EXPERIMENTAL: persistent 4-wave kernel (see gemm_bf16_w4p; 14 = epilogue inside the next
                                           // tile's first body); not part of the parity suite yet
        LT_REQUIRE(!a.tile_expert && !a.trace && a.bias_dtype < 0, "gemm variant 13 / 14: dense problems without bias, no trace build");
        LT_REQUIRE(a.K % 64 == 0 && a.K >= 128, "gemm variant 13 / 14: K=%d must be a multiple of 64, >= 128 (slabs are consumed in pairs)", a.K);
        LT_REQUIRE(255LL * a.ldc * 2 + (long long)a.N * 2 < 0x7fffffffLL, "gemm variant 13 / 14: C row stride too large for 32-bit tile offsets");
        if (variant == 14) return epilogue == 1 ? launch_w4p<1, true>(a, stream, ev0, ev1) : launch_w4p<0, true>(a, stream, ev0, ev1);
        return epilogue == 1 ? launch_w4p<1>(a, stream, ev0, ev1) : launch_w4p<0>(a, stream, ev0, ev1);
    }
    if (variant == 12) {  // EXPERIMENTAL: 4 waves, VGPR-staged (see gemm_bf16_w4s); not part of the parity suite yet
        LT_REQUIRE(!a.tile_expert, "gemm variant 12: dense problems only");
        if (a.trace) {
            LT_REQUIRE(epilogue == 0, "gemm trace: plain epilogue");
            return launch_w4s<0, true>(a, stream, ev0, ev1);
        }
        return epilogue == 1 ? launch_w4s<1>(a, stream, ev0, ev1) : launch_w4s<0>(a, stream, ev0, ev1);
    }
    if (variant == 11) {  // ping-pong 256x256 with the accumulators held in AGPRs
        LT_REQUIRE(!a.trace, "gemm variant 11: no trace build");
        return epilogue == 1 ? launch_cfg<2, 4, 4, 2, 1, true, 0, 0, 2, true>(a, stream, ev0, ev1)
                             : launch_cfg<2, 4, 4, 2, 0, true, 0, 0, 2, true>(a, stream, ev0, ev1);
    }
    if (variant == 10) {  // 256x256 tile, 4 waves of 128x128 (one wave per SIMD, 256 accumulator registers), register-pipelined loop
        if (a.trace) {
            LT_REQUIRE(epilogue == 0, "gemm trace: plain epilogue");
            constexpr int S10 = 4 * 512 * 64;
            static bool done10 = false;
            if (!done10) {
                LT_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bf16_pp<2, 2, 4, 4, 0, true, 0, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, S10));
                done10 = true;
            }
            hipLaunchKernelGGL((gemm_bf16_pp<2, 2, 4, 4, 0, true, 0, 2, 2>), dim3(((a.M + 255) / 256) * ((a.N + 255) / 256)), dim3(256), S10, stream, a);
            LT_CHECK_HIP(hipGetLastError());
            return 0;
        }
        return epilogue == 1 ? launch_cfg<2, 2, 4, 4, 1, true, 0, 2, 2>(a, stream, ev0, ev1) : launch_cfg<2, 2, 4, 4, 0, true, 0, 2, 2>(a, stream, ev0, ev1);
    }
    if (variant == 9 || (variant == 0 && g_gemm_persist && epilogue == 1 && !a.tile_expert && !a.trace && a.K >= 96 &&
                         (long long)((a.M + 255) / 256) * ((a.N + 255) / 256) >= 2LL * num_cus())) {
        // persistent ping-pong kernel (256x256 tiles, several tiles per CU, LDS ring and DMA prefetch carried across tiles)
        LT_REQUIRE(!a.tile_expert && !a.trace && a.K >= 96, "gemm variant 9: dense problems with K >= 96 only");
        return epilogue == 1 ? launch_persist<1>(a, stream, ev0, ev1) : launch_persist<0>(a, stream, ev0, ev1);
    }
    const int cus = num_cus();
    const long long t256 = (long long)((a.M + 255) / 256) * ((a.N + 255) / 256);
    const long long t128 = (long long)((a.M + 127) / 128) * ((a.N + 127) / 128);
    // small-M problems (cfg 1 / cfg 5: 512 rows): 256-wide tiles leave most CUs idle and every workgroup is a long serial
    // K loop that streams weights nobody else re-uses; 128 x 128 (or 64 x 128) ping-pong tiles give 4-8x the workgroups,
    // each with its own 3-slab prefetch window.  Variants 7 / 8 force them.
    const bool small = !a.trace && (variant == 7 || variant == 8 || (variant == 0 && g_gemm_variant == 0 && 2 * t256 <= cus));
    if (small) {
        const bool tiny = variant == 8 || (variant == 0 && 2 * t128 <= cus);
        // 64-deep slabs in every default form (half the barrier intervals of the 32-deep loop).  Measured per cfg-1 layer
        // (profiles/r01/opbench_small_m.log): single-barrier rendezvous 72.9 us < two-barrier ping-pong 77.8 us < register-
        // pipelined loop 84-93 us (all waves read, then all waves multiply: the LDS-DMA issue cost of 3-4 pieces per wave and
        // interval is not covered by anything) < 32-deep ping-pong 93-100 us.  At this size the loop is bound by the
        // L2 -> LDS staging rate of the 96-144 busy CUs, not by the matrix pipe.
        if (g_gemm_pipeline == 2) {  // A/B: register-pipelined single-barrier loop
            if (epilogue == 1) return launch_cfg<4, 2, 1, 2, 1, true, 0, 2, 4>(a, stream, ev0, ev1);
            if (tiny) return launch_cfg<2, 4, 1, 1, 0, true, 0, 2, 4>(a, stream, ev0, ev1);
            return launch_cfg<2, 4, 2, 1, 0, true, 0, 2, 4>(a, stream, ev0, ev1);
        }
        if (g_gemm_pipeline != 1) {  // default
            if (epilogue == 1) return launch_cfg<4, 2, 1, 2, 1, true, 0, 1, 4>(a, stream, ev0, ev1);
            if (tiny) return launch_cfg<2, 4, 1, 1, 0, true, 0, 1, 4>(a, stream, ev0, ev1);
            return launch_cfg<2, 4, 2, 1, 0, true, 0, 1, 4>(a, stream, ev0, ev1);
        }
        if (epilogue == 1) return launch_cfg<4, 2, 1, 2, 1, true>(a, stream, ev0, ev1);
        if (tiny) return launch_cfg<2, 4, 1, 1, 0, true>(a, stream, ev0, ev1);
        return launch_cfg<2, 4, 2, 1, 0, true>(a, stream, ev0, ev1);
    }
    bool pp = g_gemm_pipeline == 1, rv = g_gemm_pipeline == 3;
    if (variant >= 5) { rv = true; pp = false; variant -= 4; }
    else if (variant >= 3) { pp = true; rv = false; variant -= 2; }
    if (a.trace) {  // diagnostic build of the ping-pong kernel with s_memtime stamps (scripts/gemm_trace.py)
        LT_REQUIRE(epilogue == 0 && (variant == 1 || variant == 2), "gemm trace: plain epilogue, explicit tile shape");
        constexpr int S1 = 4 * 512 * 64, S2 = 4 * 544 * 64;
        static bool done = false;
        if (!done) {
            LT_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bf16_pp<2, 4, 4, 2, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, S1));
            LT_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bf16_pp<2, 4, 4, 2, 0, true, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, S1));
            LT_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bf16_pp<2, 4, 4, 2, 0, true, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, S1));
            LT_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bf16_pp<4, 3, 2, 3, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, S2));
            done = true;
        }
        const int TMx = (a.M + 255) / 256;
        if (variant == 1 && rv) hipLaunchKernelGGL((gemm_bf16_pp<2, 4, 4, 2, 0, true, 0, 1>), dim3(TMx * ((a.N + 255) / 256)), dim3(512), S1, stream, a);
        else if (variant == 1 && g_gemm_pp_tail) hipLaunchKernelGGL((gemm_bf16_pp<2, 4, 4, 2, 0, true, 3>), dim3(TMx * ((a.N + 255) / 256)), dim3(512), S1, stream, a);
        else if (variant == 1) hipLaunchKernelGGL((gemm_bf16_pp<2, 4, 4, 2, 0, true>), dim3(TMx * ((a.N + 255) / 256)), dim3(512), S1, stream, a);
        else hipLaunchKernelGGL((gemm_bf16_pp<4, 3, 2, 3, 0, true>), dim3(TMx * ((a.N + 287) / 288)), dim3(768), S2, stream, a);
        LT_CHECK_HIP(hipGetLastError());
        return 0;
    }
    // SwiGLU GEMM (N = 2F = 12288 at cfg 2: whole rounds of 256x256 tiles): the ping-pong kernel measured +5 % over the classic
    // loop (opbench r01), so it is the default there; g_gemm_pipeline == 2 forces the classic loop everywhere (A/B).
    if (epilogue == 1) {
        if (rv) return launch_cfg<2, 4, 4, 2, 1, true, 0, 1>(a, stream, ev0, ev1);
        if (!(pp || g_gemm_pipeline == 0)) return launch_cfg<2, 4, 4, 2, 1, false>(a, stream, ev0, ev1);
        return g_gemm_pp_tail ? launch_cfg<2, 4, 4, 2, 1, true, 3>(a, stream, ev0, ev1) : launch_cfg<2, 4, 4, 2, 1, true>(a, stream, ev0, ev1);
    }
    if (variant == 0) variant = g_gemm_variant;
    if (variant == 0) {
        const long long tm = (a.M + 255) / 256;
        const long long t288 = tm * ((a.N + 287) / 288);
        const long long c256 = ((t256 + cus - 1) / cus) * 256, c288 = ((t288 + cus - 1) / cus) * 288;
        variant = c288 < c256 ? 2 : 1;
    }
    if (rv) return variant == 2 ? launch_cfg<4, 3, 2, 3, 0, true, 0, 1>(a, stream, ev0, ev1) : launch_cfg<2, 4, 4, 2, 0, true, 0, 1>(a, stream, ev0, ev1);
    if (variant == 2) return pp ? launch_cfg<4, 3, 2, 3, 0, true>(a, stream, ev0, ev1) : launch_cfg<4, 3, 2, 3, 0, false>(a, stream, ev0, ev1);
    if (!pp) return launch_cfg<2, 4, 4, 2, 0, false>(a, stream, ev0, ev1);
    return g_gemm_pp_tail ? launch_cfg<2, 4, 4, 2, 0, true, 3>(a, stream, ev0, ev1) : launch_cfg<2, 4, 4, 2, 0, true>(a, stream, ev0, ev1);
}

int launch_pack_w13(const u16* w1, const u16* w3, u16* out, int F, int K, hipStream_t stream) {
    LT_REQUIRE(F % 32 == 0 && K % 8 == 0, "pack_w13: F %% 32 and K %% 8 required");
    hipLaunchKernelGGL(pack_w13_kernel, dim3(1024), dim3(256), 0, stream, w1, w3, out, F, K);
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}
