// Self-attention for head_dim 48 (Next-DiT-ImageNet 600M, Next-DiT-MoE 600M: BASELINE configs[0] / [4]), round 4: the one-wave-per-SIMD
// structure of attention_v4.hip (hd 72) at the head dim where it is SOFTMAX-bound, not MFMA-bound.
//
// What carries over unchanged: 4 waves x 64 query rows (two 32-row blocks A, B per wave), swapped QK^T on v_mfma_f32_32x32x16_bf16 with
// the running maximum folded into the MFMA through pad slots, row sum from a row of ones behind V^T, P = the lane's own registers
// (key-permuted V^T image), O^T / Q / K / V^T fragments in asm-owned AGPRs (attention_v4_48_asm.inc, scripts/gen_attn_v4_48_asm.py),
// one software-pipelined instruction stream per 64-key tile with the other block's softmax as single-instruction fillers, ring-depth
// unrolled tile loop, one barrier per tile, fused text phase.  What head_dim 48 changes:
//  * 48 = three k-steps of 16 with NO spare slots: a FOURTH k-step carries only the pad slots (hi half: the key's constant chunk
//    (1, 1, mask, 0..) against Q's (-m_hi, -m_lo, 1, 0..); lo half: a chunk of zeros against zeros) - 8 QK^T MFMAs per block and tile;
//  * O^T has 64 rows = two 32-row blocks: 48 of data, the row of ones (row sum), 15 rows of zeros - 8 PV MFMAs per block and tile;
//    16 MFMAs per block and tile instead of hd 72's 22, for the SAME softmax work (32 exp2, 16 cvt_pk, 16 max3 per lane): a tile's
//    1024 matrix-pipe cycles stand against ~1300 issue cycles (32 v_exp_f32 at 8 cycles + the rest), so the fillers are spread
//    evenly - 3 exp2 per gap - over a window that reaches six gaps into the block's own PV segment (the PV MFMA of key group g
//    issues two gaps after the group's last cvt_pk);
//  * a (K, V^T) tile pair is 12 one-KiB staging pieces: three per wave, no duplicates.
// Replaces flash_attn_func of Next-DiT-ImageNet/models/models.py:389 / Next-DiT-MoE/models/models2.py:389 (exact softmax attention,
// bf16 in / fp32 accumulate / bf16 out); bit-identical to attn_fwd_kernel_v2<48> is NOT claimed (other summation order): the tests
// hold it to the fp32 softmax reference at the tolerances of the other attention kernels.
#include "common.h"
#include "kernels.h"
#include <type_traits>

namespace lt_attn48 {

#include "attention_v4_48_asm.inc"

__global__ __launch_bounds__(256, 1) void attn_fwd_kernel_v4h48(AttnArgs p) {
    constexpr int HD = 48, KS = 4, DT = 2, NB = 2;  // KS: three k-steps of data + the pad step; DT: O^T blocks of 32 rows
    constexpr int KTILE = 64 * HD * 2;        // 6144
    constexpr int VTILE = HD * 128 + 256;     // incl. the row of ones (d = HD) and a row of zeros (d = HD + 1)
    // LDS: K ring (4 slots) | V^T ring (4 slots) | the pad chunks, laid out at the K ring's (slot, sub-tile) strides so that every
    // fragment read is `lane address + immediate`, whatever the ring slot (the tile loop is unrolled by the ring depth)
    constexpr int K_BASE = 0, V_BASE = 4 * KTILE, CONST_OFF = V_BASE + 4 * VTILE;
    constexpr int NKP = 6, NVP = 6, IP = (NKP + NVP) / 4;  // 1-KiB staging pieces per (K, V^T) tile pair; per wave and tile
    constexpr int STG_OFF = CONST_OFF + 3 * KTILE + 32 * HD * 2 + 512 + 16 + 16;  // four wave-private strips of 64 rows x HD for the output rows
    constexpr int ZERO_CHUNK = 512;  // behind the 32 pad chunks of every (slot, sub-tile) block: 16 bytes of zeros (the pad step's lo half)
    constexpr int W_PV = 4 * DT, W_END = 4 * DT + 2 * KS + 6;  // softmax window: gaps [0, W_PV) under the other block's PV, [W_PV, W_PV + 2 KS) under its QK^T, then six gaps into the block's own PV
    constexpr float THR = 8.0f;
    constexpr int LI = HD % 32, L_DT = HD / 32, L_HI = (LI >> 2) & 1, L_REG = (LI & 3) + 4 * (LI >> 3);
    extern __shared__ __attribute__((aligned(16))) char smem[];

    unsigned long long tr_entry = 0, tr_loop0 = 0, tr_loop1 = 0, tr_clk = 0;  // diagnostics (lt_op_attention_trace): phase stamps
    if (p.trace) tr_entry = __builtin_amdgcn_s_memrealtime();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;

    const int nqb = (p.N + 255) / 256;
    const int BH = p.B * p.H;
    int bh, qb;
    if ((BH & 7) == 0) {  // XCD-aware: head bh lives on XCD bh % 8, its q-blocks run back to back (K/V stay in that L2)
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        bh = xcd + 8 * (idx / nqb);
        qb = idx % nqb;
    } else {
        bh = blockIdx.x / nqb;
        qb = blockIdx.x % nqb;
    }
    const int b = bh / p.H, h = bh - b * p.H;
    const int bhk = b * p.Hkv + h / (p.H / p.Hkv);

    // constants in LDS: ones rows behind each V^T slot, and the K-side pad chunks - one per key, at the K ring's (slot, sub-tile,
    // key) strides / 9: (1, 1, b, 0, 0, 0, 0, 0) with b = 0 for image keys and the key's additive mask (0 / -inf) for text keys
    // (O^T has 96 rows for 73 used ones: rows 73..95 multiply zeros - an all-zero operand row draws no toggling power - not ones)
    *(unsigned*)(smem + V_BASE + (tid >> 6) * VTILE + HD * 128 + (tid & 63) * 4) = (tid & 32) ? 0u : 0x3F803F80u;
    auto pad_chunk = [&](int slot, int kt2, int key) __attribute__((always_inline)) {
        return (u32x4*)(smem + CONST_OFF + slot * KTILE + kt2 * (32 * HD * 2) + key * 16);
    };
    *pad_chunk(tid >> 6, (tid >> 5) & 1, tid & 31) = u32x4{0x3F803F80u, 0u, 0u, 0u};
    if (tid < 8) *(u32x4*)(smem + CONST_OFF + (tid >> 1) * KTILE + (tid & 1) * (32 * HD * 2) + ZERO_CHUNK) = u32x4{0u, 0u, 0u, 0u};

    // ---- staging: 12 one-KiB pieces per (K, V^T) tile pair, three per wave, branch-free --------------------------------------------
    // piece q = wave + 4 i; q < 6: bytes [1024 q, +1024) of the K tile's LDS image; else V^T rows 8 (q - 6) .. +7 (128-byte rows,
    // chunk-swizzled on the source address).  Slot 0 is a K piece and slot 2 a V^T piece for every wave; slot 1 is K for waves 0, 1
    int st_lds[IP], st_voff[IP];
    const bool s1_is_k = (wave < 2);
    auto v_voff = [&](int j, int v_ld) __attribute__((always_inline)) {
        const int d = 8 * j + (lane >> 3);
        return d * v_ld * 2 + (((lane & 7) ^ ((d >> 1) & 7)) << 4);
    };
    __amdgpu_buffer_rsrc_t rK, rV, rS1;
    auto make_src = [&](const u16* k_head, int k_rows, const u16* v_head, int v_ld) __attribute__((always_inline)) {
        const int kbytes = (int)((size_t)k_rows * HD * 2), vbytes = (int)((size_t)HD * v_ld * 2);
        rK = __builtin_amdgcn_make_buffer_rsrc((void*)k_head, 0, kbytes, 0x00020000);
        rV = __builtin_amdgcn_make_buffer_rsrc((void*)v_head, 0, vbytes, 0x00020000);
        rS1 = __builtin_amdgcn_make_buffer_rsrc((void*)(s1_is_k ? k_head : v_head), 0, s1_is_k ? kbytes : vbytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < IP; ++i) {
            const int q = wave + 4 * i;
            const bool isk = q < NKP;
            const int j = isk ? q : q - NKP;
            st_lds[i] = (isk ? K_BASE : V_BASE) + j * 1024;
            st_voff[i] = isk ? j * 1024 + lane * 16 : v_voff(j, v_ld);
        }
    };
    // piece i of the batch {K(tk), V^T(tv)} into ring slots ks / vs (compile-time constants in the tile loop)
    auto dma = [&](int i, int tk, int tv, int ks, int vs) __attribute__((always_inline)) {
        const bool isk = i == 0 || (i == 1 && s1_is_k);
        const int dst = st_lds[i] + (isk ? ks * KTILE : vs * VTILE);
        const int soff = isk ? tk * KTILE : tv * 128;
        if (i == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rK, LDS_PTR(smem + dst), 16, st_voff[i], soff, 0, 0);
        else if (i == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rS1, LDS_PTR(smem + dst), 16, st_voff[i], soff, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rV, LDS_PTR(smem + dst), 16, st_voff[i], soff, 0, 0);
    };
    // ---- per-lane LDS read addresses (absolute LDS byte addresses: the fragment reads are inline assembly) -------------------------
    const unsigned lds0 = (unsigned)(size_t)LDS_PTR(smem);
    const int ka = (int)lds0 + K_BASE + l31 * (HD * 2) + hi * 16;  // + slot * KTILE + kt2 * 32 * HD * 2 + 32 s  (immediates)
    // pad k-step: the hi half reads its key's pad chunk (1, 1, mask, 0, ..), the lo half the block's chunk of zeros (its Q slots are 0
    // as well: 0 x 0, never 0 x -inf)
    const int kp = (int)lds0 + CONST_OFF + (hi ? l31 * 16 : ZERO_CHUNK);
    // V^T fragment (dt, g): row d = 32 dt + l31, chunk (2 g + hi) ^ ((d >> 1) & 7); dt = 1 (rows 32..63: hd 32..47, the row of ones at
    // d = 48, zeros past it) has its own addresses
    int va01[4], va2[4];
    {
        const int d2 = (32 * (DT - 1) + l31 > HD) ? HD + 1 : 32 * (DT - 1) + l31;  // d = HD: the row of ones (row sum); past it: the row of zeros
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            va01[g] = (int)lds0 + V_BASE + l31 * 128 + (((2 * g + hi) ^ ((l31 >> 1) & 7)) << 4);
            va2[g] = (int)lds0 + V_BASE + d2 * 128 + (((2 * g + hi) ^ ((d2 >> 1) & 7)) << 4);
        }
    }

    f32x16 sc[NB][2];     // scores of the current tile, two 32-key sub-tiles per block; exp2 in place
    u32x4 pa[NB][4];      // P of the current tile as bf16 pairs: the PV operands (16-key groups)
    float m_run[NB] = {0.f, 0.f}, mxv[NB] = {0.f, 0.f};
    v4_o_zero(0);
    v4_o_zero(1);
    v4_kc_init(hi ? 0x3F803F80u : 0u);

    auto fence = [&]() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };
    auto bar = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    // fragment reads: asm on both ends (the compiler neither copies the AGPR fragments nor knows about their latency: the consumers
    // wait with s_waitcnt lgkmcnt, see iter())
    auto read_k = [&](int n, int slot) __attribute__((always_inline)) { v4_read_k(n, slot, ka, kp); };
    auto read_v = [&](int n, int slot) __attribute__((always_inline)) { v4_read_v(n, slot, va01[n / DT], va2[n / DT]); };
    auto mfma_qk = [&](int blk, int n) __attribute__((always_inline)) { v4_mfma_qk(blk, n, sc[blk][n & 1]); };
    auto mfma_pv = [&](int blk, int n) __attribute__((always_inline)) { v4_mfma_pv(blk, n, pa[blk][n / DT]); };
    auto lgkm0 = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };

    // ---- softmax of one block's tile, cut into the 19 filler groups of its window ------------------------------------------------
    // Window gap w (0..21: the two segments a block's softmax hides under, + 2).  Nothing in gaps 0, 1 (the scores' last MFMAs are still
    // in flight); the rest is packed by measured issue cost
    // (scripts/ubench/valu_rate.hip: v_exp_f32 8 cycles, v_max3 / v_cvt_pk 4.5) to <= 25 cycles per gap beside the gap's one fragment
    // read or DMA piece and the MFMA's own issue slot (32 cycles per MFMA):
    //   2-5: 4 max3 | 6: fold, exchange with the row's other half | 7: move the folded maximum if the tile exceeds it by 2^THR (rare
    //   wave-uniform branch) | 8-22: per 16-key group 8 exp2 in place + 4 cvt_pk into the PV operand, three exp2 (or the equivalent) per gap
    // Every step is an inline-asm statement: the written order is the issue order.  (The hazard recognizer puts a wait state between
    // two dependent inline-asm statements with no compiler-visible instruction between them: the order below keeps such pairs rare.)
    // hot = true (tiles >= 1 of the self-attention loop): the rare branch touches the scores, O^T and Q only through inline assembly
    // (v4_score_shift: the move is one extra MFMA per sub-tile against the constant K fragment) - the hot path then has no register
    // copies at the join.  hot = false (tile 0, text tiles): plain VALU form, with the forced move of a first tile.
#define S_(g, i) sc[blk][(g) >> 1][8 * ((g) & 1) + (i)]
    // one asm statement per instruction: with several elements of a score tuple as operands of ONE statement the register allocator
    // stops treating them as sub-registers of the MFMA's tuple and copies 20 registers per block and tile
    auto exp1 = [&](int blk, int g, int i) __attribute__((always_inline)) {
        float x = S_(g, i);
        asm volatile("v_exp_f32 %0, %0" : "+v"(x));
        S_(g, i) = x;
    };
    auto cvt1 = [&](int blk, int g, int j) __attribute__((always_inline)) {
        const float x = S_(g, 2 * j), y = S_(g, 2 * j + 1);
        unsigned wd;
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(wd) : "v"(x), "v"(y));
        pa[blk][g][j] = wd;
    };
    auto exps = [&](int blk, int g, int i, int n) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < n; ++e) exp1(blk, g, i + e);
    };
    auto cvts = [&](int blk, int g, int j, int n) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < n; ++e) cvt1(blk, g, j + e);
    };
    // running tile max over score indices k0 .. k0 + 3 of both sub-tiles: four independent accumulators (a dependent pair of asm
    // statements costs a wait state; this way it is one per gap), folded and exchanged with the row's other half in gap 6
    float mx4[NB][4];
    auto max4 = [&](int blk, int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = sc[blk][0][k0 + e], b = sc[blk][1][k0 + e];
            if (k0 == 0) asm volatile("v_max_f32 %0, %1, %2" : "=v"(mx4[blk][e]) : "v"(a), "v"(b));
            else asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(mx4[blk][e]) : "v"(a), "v"(b));
        }
    };
    auto max_fold = [&](int blk) __attribute__((always_inline)) {
        float a, b;
        asm volatile("v_max3_f32 %0, %1, %2, %3\n\tv_max_f32 %0, %0, %4" : "=&v"(a) : "v"(mx4[blk][0]), "v"(mx4[blk][1]), "v"(mx4[blk][2]), "v"(mx4[blk][3]));
        b = a;
        // the other 16 keys of the row's 32-key sub-tiles live on lane ^ 32: v_permlane32_swap puts (own, partner) halves side by side
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_max_f32 %0, %0, %1" : "+v"(a), "+v"(b));
        mxv[blk] = a;
    };
    auto decide = [&](int blk, int t, bool hot) __attribute__((always_inline)) {
        const float mx = mxv[blk];
        const bool first = !hot && (t == 0);
        const bool raise = first || (mx > THR);
        if (__builtin_expect(__any(raise), 0)) {
            // new running max = m_run + mx (scores are relative to m_run already), rounded to the bf16 pair (hi + lo) the MFMA
            // will actually subtract from now on; delta is the step between the two REPRESENTED values
            const float nm = -(m_run[blk] + (raise ? mx : 0.f));
            const float nm_hi = bfr(nm);
            const float nm_lo = bfr(nm - nm_hi);
            const float m_new = -(nm_hi + nm_lo);
            const float delta = m_new - m_run[blk];
            const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);
            if (hot) {
                // the pair Q carries now (it represents -m_run exactly, so it can be recomputed)
                const float p_hi = bfr(-m_run[blk]);
                const float p_lo = bfr(-m_run[blk] - p_hi);
                v4_score_shift(hi ? pack2bf(nm_hi, nm_lo) : 0u, hi ? pack2bf(-p_hi, -p_lo) : 0u, sc[blk][0], sc[blk][1]);
            } else {
#pragma unroll
                for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sc[blk][kt2][r] -= delta;
            }
            m_run[blk] = m_new;
            v4_o_scale(blk, alpha);
            if (hi) v4_q_pad_write(blk, pack2bf(nm_hi, nm_lo));  // pad slots 0 and 1 of the last k-step live on the hi half
        }
    };
    // w = 0 .. 21.  Gaps 0 .. 7 lie under the other block's PV segment, 8 .. 15 under its QK^T segment, 16 .. 21 are the first six gaps of
    // the block's OWN PV segment (its MFMA i reads P of key group i / 2: group 2 from gap 20 on, group 3 from gap 22 on - two gaps
    // behind the group's last cvt_pk below); there they share a gap with the other block's gaps 0 .. 5 (empty, empty, 4 x max).
    // ~26 issue cycles of fillers per gap (3 exp2, or 2 exp2 + 2 cvt_pk, ...): this head dim is bound by them, not by the MFMAs.
    auto sm_fill = [&](int blk, int w, int t, bool hot) __attribute__((always_inline)) {
        switch (w) {
            case 2: case 3: case 4: case 5: max4(blk, 4 * (w - 2)); break;
            case 6: max_fold(blk); break;
            case 7: decide(blk, t, hot); break;
            case 8: exps(blk, 0, 0, 3); break;
            case 9: exps(blk, 0, 3, 3); break;
            case 10: exps(blk, 0, 6, 2); cvts(blk, 0, 0, 2); break;
            case 11: cvts(blk, 0, 2, 2); exps(blk, 1, 0, 2); break;
            case 12: exps(blk, 1, 2, 3); break;
            case 13: exps(blk, 1, 5, 3); break;
            case 14: cvts(blk, 1, 0, 4); exps(blk, 2, 0, 1); break;
            case 15: exps(blk, 2, 1, 3); break;
            case 16: exps(blk, 2, 4, 3); break;
            case 17: exps(blk, 2, 7, 1); cvts(blk, 2, 0, 4); break;
            case 18: exps(blk, 3, 0, 3); break;
            case 19: exps(blk, 3, 3, 3); break;
            case 20: exps(blk, 3, 6, 2); cvts(blk, 3, 0, 2); break;
            case 21: cvts(blk, 3, 2, 2); break;
            default: break;
        }
    };

    // one tile; J = t & 3 (ring slot of K(t) and V^T(t)) is a compile-time constant: the loop below is unrolled by the ring depth, so
    // every LDS offset is an immediate and the scalar work per tile is the DMA's m0 writes and two running offsets.
    // HAS_PREV = false: tile 0 (no softmax_B / PV_B of a previous tile).
    // Fragment reads are issued in the first half of a segment and waited for ONCE at the start of the next one (a counted wait per
    // MFMA cost an issue slot per MFMA: with one wave per SIMD every instruction is one of ~8 slots an MFMA covers).
    auto iter = [&](int t, auto has_prev_c, auto slot_c) __attribute__((always_inline)) {
        constexpr bool HAS_PREV = decltype(has_prev_c)::value;
        constexpr int J = decltype(slot_c)::value, J1 = (J + 1) & 3, J2 = (J + 2) & 3, J3 = (J + 3) & 3;
        // seg 1: QK^T_A(t) | softmax_B(t-1) gaps 8 .. 15 | the three DMA pieces (odd gaps)
        lgkm0();  // K(t) fragments (read during seg 3 / 4 of the previous tile, or the prologue)
#pragma unroll
        for (int i = 0; i < 2 * KS; ++i) {
            mfma_qk(0, i);
            if ((i & 1) && (i >> 1) < IP) dma(i >> 1, t + 3, t + 2, J3, J2);
            if constexpr (HAS_PREV) sm_fill(1, W_PV + i, t - 1, true);  // (no max move in these gaps)
            fence();
        }
        // seg 2: PV_B(t-1) | softmax_A(t) first half | V^T(t) fragment n two gaps after PV_B's MFMA n released its register
#pragma unroll
        for (int i = 0; i < 4 * DT; ++i) {
            if constexpr (HAS_PREV) mfma_pv(1, i);
            else if (i == 0) { asm volatile("s_nop 15"); asm volatile("s_nop 15"); }  // the scores' MFMAs complete uncovered
            if (i >= 2) read_v(i - 2, J);
            if constexpr (HAS_PREV) { if (i < 6) sm_fill(1, W_PV + 2 * KS + i, t - 1, true); }
            sm_fill(0, i, t, HAS_PREV);
            fence();
        }
        // seg 3: QK^T_B(t) | softmax_A(t) second half | last two V^T(t) fragments, then K(t+1) fragment n two gaps after its MFMA
#pragma unroll
        for (int i = 0; i < 2 * KS; ++i) {
            mfma_qk(1, i);
            if (i < 2) read_v(4 * DT - 2 + i, J);
            else read_k(i - 2, J1);
            sm_fill(0, W_PV + i, t, HAS_PREV);
            fence();
        }
        // seg 4: PV_A(t) | softmax_B(t) first half | last two K(t+1) fragments
        asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");  // V^T(t) fragments (the six K reads behind them may stay in flight)
#pragma unroll
        for (int i = 0; i < 4 * DT; ++i) {
            mfma_pv(0, i);
            if (i < 2) read_k(2 * KS - 2 + i, J1);
            if (i < 6) sm_fill(0, W_PV + 2 * KS + i, t, HAS_PREV);
            sm_fill(1, i, t, HAS_PREV);
            fence();
        }
        // K(t+2), V(t+1) (issued one tile ago) have landed; this tile's three pieces may stay in flight
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        bar();
    };
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    using C2 = std::integral_constant<int, 2>;
    using C3 = std::integral_constant<int, 3>;
    using T_ = std::true_type;

    // ---- two phases through the same tile pipeline: the image keys (self-attention), then - fused zero-init gated text
    //      cross-attention (model.py:420-434) - the text keys of the same Q rows (post-RoPE, :427) with fresh O^T / maximum -----------
    u32x2 res[NB][DT][4];  // result, bf16 (flash-attn output dtype, model.py:392-405)
    int qrow[NB];
    bool q_ok[NB];
    const int nph = p.tk ? 2 : 1;
    // the text keys' mask values (0 / -inf: scale free), one per thread = per text key, fetched now and written into the pad chunks
    // when the text phase is set up
    float text_bias = -INFINITY;
    if (p.tk && tid < p.Tkpad) text_bias = p.tbias[(size_t)b * p.Tkpad + tid];
    // sources of a phase + its first batches: {K(0)}, {K(1), V(0)}, {K(2), V(1)}  (what iterations -3, -2, -1 would have issued).
    // Phase 1 is started right after the last barrier of phase 0's tile loop - the ring and the pad chunks are idle from there on -
    // so that its tiles fly under phase 0's drain and epilogue.
    auto start_phase = [&](int ph) __attribute__((always_inline)) {
        if (ph == 0) {
            make_src(p.k + (size_t)bhk * p.Nk * HD, p.Nk, p.vt + (size_t)bhk * HD * p.Nkpad, p.Nkpad);
        } else {
            make_src(p.tk + (size_t)bhk * p.Tk * HD, p.Tk, p.tvt + (size_t)bhk * HD * p.Tkpad, p.Tkpad);
            // thread -> text key tid (tile tid >> 6, at most four tiles: launcher): (1, 1, mask, 0, ..)
            *pad_chunk(tid >> 6, (tid >> 5) & 1, tid & 31) = u32x4{0x3F803F80u, (unsigned)f2bf(text_bias), 0u, 0u};
        }
#pragma unroll
        for (int i = 0; i < IP; ++i) {
            const bool isk = i == 0 || (i == 1 && s1_is_k);
            if (isk) dma(i, 0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < IP; ++i) dma(i, 1, 0, 1, 0);
#pragma unroll
        for (int i = 0; i < IP; ++i) dma(i, 2, 1, 2, 1);
    };
    start_phase(0);
    for (int ph = 0; ph < nph; ++ph) {
        const int nt = ph == 0 ? p.Nk / 64 : (p.Tk + 63) / 64;
        if (ph == 1) {
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
                v4_o_zero(blk);
                m_run[blk] = 0.f;
                if (hi) v4_q_pad_write(blk, 0u);
            }
        }
        if (ph == 0) {
        // (after the first tiles' DMA is in flight: the Q rows' global loads and their conversion overlap it)
        // ---- Q fragments of both blocks, pre-scaled to the log2 domain unless K carries the scale --------------------------------
        const float sl2 = p.k_prescaled ? 1.0f : p.scale * 1.44269504088896340736f;
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
            qrow[blk] = qb * 256 + wave * 64 + blk * 32 + l31;
            q_ok[blk] = qrow[blk] < p.N;
            if (!q_ok[blk]) qrow[blk] = p.N - 1;
            const u16* qptr = p.q + ((size_t)bh * p.N + qrow[blk]) * HD;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int d0 = 16 * s + 8 * hi;
                // the pad k-step (s = KS - 1, d0 >= HD on both halves): hi half - slots 0, 1 become (-m_hi, -m_lo), slot 2 = 1.0 multiplies the
                // key's mask value; lo half - all zeros (against the chunk of zeros)
                unsigned w[4] = {0u, (d0 >= HD && hi) ? 0x00003F80u : 0u, 0u, 0u};
                if (d0 < HD) {
                    const bf8_t raw = *(const bf8_t*)(qptr + d0);
                    float f[8];
                    unpack8(raw, f);
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] = pack2bf(f[2 * e] * sl2, f[2 * e + 1] * sl2);
                }
                v4_q_write(blk, s, w[0], w[1], w[2], w[3]);
            }
        }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // DMA landed, constant rows / pad chunks written
        bar();
#pragma unroll
        for (int n = 0; n < 2 * KS; ++n) read_k(n, 0);

        if (p.trace && ph == 0) { tr_loop0 = __builtin_amdgcn_s_memrealtime(); tr_clk = __builtin_amdgcn_s_memtime(); }
        iter(0, std::false_type{}, C0{});
        if (nt > 1) iter(1, T_{}, C1{});
        if (nt > 2) iter(2, T_{}, C2{});
        if (nt > 3) iter(3, T_{}, C3{});
        int t = 4;
        for (; t + 3 < nt; t += 4) {
            iter(t, T_{}, C0{});
            iter(t + 1, T_{}, C1{});
            iter(t + 2, T_{}, C2{});
            iter(t + 3, T_{}, C3{});
        }
        if (t < nt) iter(t, T_{}, C0{});
        if (t + 1 < nt) iter(t + 1, T_{}, C1{});
        if (t + 2 < nt) iter(t + 2, T_{}, C2{});
        if (p.trace && ph == 0) { tr_loop1 = __builtin_amdgcn_s_memrealtime(); tr_clk = __builtin_amdgcn_s_memtime() - tr_clk; }
        if (ph + 1 < nph) start_phase(ph + 1);
        // drain: softmax_B(last) second half, PV_B(last)
#pragma unroll
        for (int w = W_PV; w < W_END; ++w) { sm_fill(1, w, nt - 1, nt > 1); fence(); }
#pragma unroll
        for (int i = 0; i < 4 * DT; ++i) { mfma_pv(1, i); fence(); }
        fence();

        // l = O^T[HD][q] lives in register L_REG of tile L_DT on the hi == L_HI lane of this query row
        const float gate = ph == 0 ? 0.f : bfr(tanhf(bf2f(p.tgate[h])));
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
            f32x16 ot[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) ot[dt] = v4_o_read(blk, dt);
            const float inv = 1.0f / __shfl(ot[L_DT][L_REG], l31 + 32 * L_HI, 64);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    if (ph == 0) {
                        res[blk][dt][q4][0] = pack2bf_pk(ot[dt][4 * q4] * inv, ot[dt][4 * q4 + 1] * inv);
                        res[blk][dt][q4][1] = pack2bf_pk(ot[dt][4 * q4 + 2] * inv, ot[dt][4 * q4 + 3] * inv);
                    } else {
                        // output + bf16(output_y * tanh(gate))  (model.py:433-434, bf16 rounding points)
                        float v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = bfr(bfr(ot[dt][4 * q4 + j] * inv) * gate);
                        res[blk][dt][q4][0] = pack2bf(bf_lo(res[blk][dt][q4][0]) + v[0], bf_hi(res[blk][dt][q4][0]) + v[1]);
                        res[blk][dt][q4][1] = pack2bf(bf_lo(res[blk][dt][q4][1]) + v[2], bf_hi(res[blk][dt][q4][1]) + v[3]);
                    }
                }
        }
    }

    // ---- output: through a wave-private LDS strip to row-contiguous 16-byte stores (store_rows_via_lds, common.h) ---------------------
    {
        const int row0 = qb * 256 + wave * 64;
        if (p.out_pair)
            store_rows_via_lds<HD, DT>(smem + STG_OFF + wave * (64 * HD * 2), res, lane, p.out, (size_t)p.H * HD, p.N - row0, 1, (size_t)b * p.N + row0, h * HD);
        else
            store_rows_via_lds<HD, DT>(smem + STG_OFF + wave * (64 * HD * 2), res, lane,
                                       p.out + ((size_t)b * p.N + row0) * ((size_t)p.H * HD) + (size_t)h * HD, (size_t)p.H * HD, p.N - row0);
    }
    if (p.trace && tid == 0) {  // per workgroup: s_memrealtime (100 MHz) at entry | loop start | loop end | exit, shader clocks of the loop
        unsigned long long* o = p.trace + (size_t)blockIdx.x * 8;
        o[0] = tr_entry; o[1] = tr_loop0; o[2] = tr_loop1; o[3] = __builtin_amdgcn_s_memrealtime(); o[4] = tr_clk; o[5] = (unsigned long long)(p.Nk / 64);
        o[6] = (unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));  // XCC_ID
        o[7] = 0;
    }
}

}  // namespace lt_attn48

int launch_attention_v4_hd48(const AttnArgs& a, hipStream_t stream) {
    // V^T ring | K ring | pad-chunk blocks at the K ring's (slot, sub-tile) strides: the last block ends 3 KTILE + 32 hd 2 + 512 + 16 in
    constexpr int SMEM = 4 * (48 * 128 + 256) + 4 * (64 * 48 * 2) + 3 * (64 * 48 * 2) + 32 * 48 * 2 + 512 + 16 + 16 + 4 * (64 * 48 * 2);  // (+ the waves' output strips)
    LT_REQUIRE(a.hd == 48 && !a.bias && !a.accumulate && !a.nk_batch && a.Nk % 64 == 0 && !a.tk,
               "attention v4 (hd 48): whole 64-key tiles, no per-sample key counts, no fused text keys (the inherited text phase is untested)");
    if (ensure_dynamic_lds((const void*)lt_attn48::attn_fwd_kernel_v4h48, SMEM)) return 1;  // per (device, kernel)
    const int nqb = (a.N + 255) / 256;
    hipLaunchKernelGGL(lt_attn48::attn_fwd_kernel_v4h48, dim3(a.B * a.H * nqb), dim3(256), SMEM, stream, a);
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}
