// EXPERIMENTAL (round 3, variants 17 / 18; built only with `make EXPERIMENTAL=1`): small-M bf16 GEMM with a DEEP LDS ring.
// C[M,N] = A[M,K] · W[N,K]^T when M is a few hundred rows (cfg 1 / cfg 5: 512 rows; MoE expert groups).
//
// Hypothesis it was written to test: every 512-row GEMM of the 600M models takes ~20 us whatever its shape, the round-1 small
// tiles prefetch two slabs ahead and weights stream from HBM, so a slab interval cannot be shorter than HBM latency / 2.
// MEASURED (profiles/r03/opbench_small_m_deep_ring_vs_round1_tiles.log): bit-correct (35 tests) and 15-25 % SLOWER than the
// round-1 tiles on all four shapes - prefetch depth is not the bound.  scripts/ubench/fill_rate.hip says what is
// (profiles/r03/ubench_fill_rate_per_cu.log): one CU keeps at most ~64 KiB of fills outstanding, i.e. 140 GB/s from its L2, ~50 GB/s
// from HBM, 27 GB/s when all 256 CUs pull from the MALL, and one wave issues LDS-DMA at ~20 GB/s - so four waves (this kernel)
// fill slower than eight (the round-1 tiles), and a ring deeper than 64 KiB buys nothing.  DESIGN.md 9.1.
//
// The kernel is the persistent 4-wave 16x16x32 structure of gemm_bf16_w4q sized for that regime:
//   * small tiles (2 MT x 2 NT sixteen-row blocks: 128 x 128 or 64 x 128) so that 512-row problems give 96 - 256 workgroups;
//   * a DEEP LDS ring: RING slots of one 32-deep slab (16 / 12 KiB), the LDS-DMA stream runs RING - 1 slabs (7 - 11) ahead of the
//     MFMAs and keeps running across tile boundaries (its own tile iterator), so ~100 KiB per CU are in flight all the time;
//   * one barrier per slab, fragments of slab g + 1 read while slab g multiplies (double-buffered in registers);
//   * grouped (mixture-of-experts) mode: a tile's 256-row segment selects the expert's weights, padding tiles are skipped by
//     both iterators;
//   * epilogues 0 (plain) and 1 (SwiGLU, NT % 4 == 0) with the reference's rounding points, 16-byte stores via permlane16_swap.
// Requirements (launcher): no bias, K % 64 == 0, K >= 32 (RING + 1), lda / ldw / ldc such that a tile's byte offsets fit 31 bits.
#pragma once

namespace lt_gemm {

// one int through the scalar cache (the tile -> expert table of a grouped problem).  A plain load compiles to a VECTOR load
// followed by s_waitcnt vmcnt(0), which would drain the whole LDS-DMA ring every time the stream moves to its next tile.
__device__ __forceinline__ int sload_i32(const int* ptr) {
    int v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(ptr) : "memory");
    return v;
}

template <int EPI, int MT, int NT, int RING>
__global__ __launch_bounds__(256, 1) void gemm_bf16_sm(GemmArgs p) {
    constexpr int NW = 4, BM = 2 * MT * 16, BN = 2 * NT * 16;
    constexpr int PA = BM / 16, PW = BN / 16, NP = PA + PW;  // 1-KiB staging pieces (16 rows x 64 B) per slab
    constexpr int IP = (NP + NW - 1) / NW;                   // pieces per wave and slab
    constexpr int SLAB = (BM + BN) * 64, W_OFF = BM * 64;
    constexpr int NM = MT * NT, RD = MT + NT;
    constexpr int P = RING - 1;                              // slabs the DMA stream runs ahead of the MFMAs
    constexpr int NST = EPI == 1 ? MT * (NT / 4) : MT * (NT / 2);  // store instructions per wave and tile
    constexpr int KEEP = (P - 2) * IP;                       // DMAs younger than slab g + 2's at the end of body g
    static_assert(RD <= NM && IP <= NM, "one fragment read per MFMA in the first RD, every DMA piece between two MFMAs");
    static_assert(PA % NW == 0 && NP % NW == 0, "A pieces first, no surplus slot");
    static_assert(NT % 2 == 0 && (EPI != 1 || NT % 4 == 0), "paired 16-column tiles in the epilogue");
    static_assert(P >= 3 && KEEP + NST <= 63, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, q4 = lane >> 4;
    const int TM = (p.M + BM - 1) / BM, TN = (p.N + BN - 1) / BN;
    const int ntiles = TM * TN;
    const int kbytes = (p.K / 32) * 64;  // bytes of one row of a tile's K panel = 64 per slab
    const int ns = p.K / 32;

    // staging: wave w copies pieces w + 4 i; piece q < PA = A rows 16 q .., else W rows 16 (q - PA) ..; lane -> row lane >> 2,
    // 16-byte position lane & 3 fetched from source chunk pos ^ (3 * ((row >> 3) & 1)) (bank-conflict-free fragment reads)
    const int sswz = ((lane & 3) ^ (((lane >> 5) & 1) * 3)) * 16;
    static_assert(IP <= 8, "staging slots per wave");
    int voff[8], ldsoff[8];  // fixed size: a dependent bound here breaks host-side substitution (hipcc 7.2)
#pragma unroll
    for (int i = 0; i < IP; ++i) {
        const int q = wave + NW * i;
        const bool isA = i < PA / NW;
        const int r0 = 16 * (isA ? q : q - PA) + (lane >> 2);
        voff[i] = r0 * (isA ? p.lda : p.ldw) * 2 + sswz;
        ldsoff[i] = q * 1024;
    }
    const int ncols_out = EPI == 1 ? p.N / 2 : p.N;
    struct Tile { const u16* a; const u16* w; u16* c; int a_bytes, w_bytes, c_bytes, n0; };
    const int G = TM < 8 ? TM : 8;  // tiles that share a W panel (same column block) sit next to each other on one XCD
    auto expert_of = [&](int v) __attribute__((always_inline)) {
        int tm, tn;
        tile_coords(v, ntiles, TM, TN, tm, tn, G);
        return sload_i32(p.tile_expert + ((tm * BM) >> 8));
    };
    // this workgroup's tile sequence: v, v + grid, ... without the padding tiles of a grouped problem
    auto next_valid = [&](int v) __attribute__((always_inline)) {
        if (p.tile_expert)
            while (v < ntiles && expert_of(v) < 0) v += (int)gridDim.x;
        return v;
    };
    auto setup = [&](int v) __attribute__((always_inline)) {
        int tm, tn;
        tile_coords(v, ntiles, TM, TN, tm, tn, G);
        const int m0 = tm * BM, n0_ = tn * BN;
        const u16* Wg = p.W;
        if (p.tile_expert) Wg += (size_t)sload_i32(p.tile_expert + (m0 >> 8)) * p.w_expert_stride;
        const long long a_left = (long long)(p.M - m0) * p.lda * 2;
        const long long w_left = (long long)(p.N - n0_) * p.ldw * 2;
        const long long c_left = (long long)(p.M - m0) * p.ldc * 2;
        Tile t;
        t.a = p.A + (size_t)m0 * p.lda; t.w = Wg + (size_t)n0_ * p.ldw; t.c = p.C + (size_t)m0 * p.ldc;
        t.a_bytes = (int)(a_left > 0x7fffffffLL ? 0x7fffffffLL : a_left);
        t.w_bytes = (int)(w_left > 0x7fffffffLL ? 0x7fffffffLL : w_left);
        t.c_bytes = (int)(c_left > 0x7fffffffLL ? 0x7fffffffLL : c_left);
        t.n0 = n0_;
        return t;
    };
    const Tile t_null = {p.A, p.W, p.C, 0, 0, 0, 0};  // past the last tile: the DMA stream reads nothing (every lane out of range)

    int cv = next_valid((int)blockIdx.x);  // compute iterator
    if (cv >= ntiles) return;              // uniform
    Tile cur = setup(cv);

    // ---- the DMA stream: its own tile iterator, P slabs ahead of the MFMAs --------------------------------------------------
    int dv = cv;
    int d_soff = 0;            // byte offset along K of the slab fetched next
    int wr_off = 0;            // LDS offset of the ring slot it lands in
    __amdgpu_buffer_rsrc_t dA = __builtin_amdgcn_make_buffer_rsrc((void*)cur.a, 0, cur.a_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t dW = __builtin_amdgcn_make_buffer_rsrc((void*)cur.w, 0, cur.w_bytes, 0x00020000);
    auto dma_piece = [&](int i) __attribute__((always_inline)) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(i < PA / NW ? dA : dW, LDS_PTR(smem + wr_off + ldsoff[i]), 16, voff[i], d_soff, 0, 0);
    };
    auto dma_advance = [&]() __attribute__((always_inline)) {
        wr_off += SLAB;
        wr_off = wr_off == RING * SLAB ? 0 : wr_off;
        d_soff += 64;
        if (d_soff == kbytes) {
            d_soff = 0;
            dv = dv < ntiles ? next_valid(dv + (int)gridDim.x) : dv;
            const Tile t = dv < ntiles ? setup(dv) : t_null;
            dA = __builtin_amdgcn_make_buffer_rsrc((void*)t.a, 0, t.a_bytes, 0x00020000);
            dW = __builtin_amdgcn_make_buffer_rsrc((void*)t.w, 0, t.w_bytes, 0x00020000);
        }
    };

    // fragment reads: lane -> row l15 of the 16-row block, 16-byte chunk q4 (swizzled)
    const int csw = (q4 ^ (((l15 >> 3) & 1) * 3)) << 4;
    const int a_row_off = (wm * (MT * 16) + l15) * 64 + csw;          // + mt * 1024
    const int w_row_off = W_OFF + (wn * (NT * 16) + l15) * 64 + csw;  // + nt * 1024

    f32x4 acc[MT][NT];
    bf16x8 wf[NT], af[MT], wf2[NT], af2[MT];

    // prologue (once per workgroup): slabs 0 .. P-1 in flight, slab 0 in the first fragment set, slab 1 landed and visible
#pragma unroll
    for (int s = 0; s < P; ++s) {  // (P < ns, launcher: the first P slabs lie inside the first tile, no ring wrap yet)
#pragma unroll
        for (int i = 0; i < IP; ++i) dma_piece(i);
        wr_off += SLAB;
        d_soff += 64;
    }
    wait_vmcnt<(P - 1) * IP>();
    pp_barrier();
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) wf[nt] = *(const bf16x8*)(smem + w_row_off + nt * 1024);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) af[mt] = *(const bf16x8*)(smem + a_row_off + mt * 1024);
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    wait_vmcnt<(P - 2) * IP>();
    pp_barrier();

    int rd_off = SLAB;          // LDS offset of slab g + 1 (this body's fragment reads)
    int since_epi = P;          // bodies since the last epilogue: its NST stores sit between the ring's DMAs for P - 2 bodies
    // one slab: MFMAs of slab g from (wc, ac) | fragment reads of slab g + 1 into (wn_, an) | LDS-DMA of slab g + P
    auto body = [&](auto first_tag, bf16x8 (&wc)[NT], bf16x8 (&ac)[MT], bf16x8 (&wn_)[NT], bf16x8 (&an)[MT]) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const char* sb = smem + rd_off;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            const int mt = i / NT, nt = i % NT;
            if constexpr (FIRST) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[nt], ac[mt], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            else acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[nt], ac[mt], acc[mt][nt], 0, 0, 0);
            if (i < NT) wn_[i] = *(const bf16x8*)(sb + w_row_off + i * 1024);
            else if (i < RD) an[i - NT] = *(const bf16x8*)(sb + a_row_off + (i - NT) * 1024);
#pragma unroll
            for (int j = 0; j < IP; ++j)
                if (i == ((2 * j + 1) * NM) / (2 * IP)) dma_piece(j);
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
        dma_advance();
        rd_off += SLAB;
        rd_off = rd_off == RING * SLAB ? 0 : rd_off;
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): slab g + 1's fragments are in registers
        // slab g + 2 landed.  Still allowed in flight: the DMAs of slabs g + 3 .. g + P and, for P - 2 bodies after a tile
        // boundary, the NST stores of that epilogue (loads and stores retire in issue order, one counter)
        if (since_epi < P - 2) { wait_vmcnt<KEEP + NST>(); ++since_epi; }
        else wait_vmcnt<KEEP>();
        pp_barrier();
    };
    // epilogue: lane holds, per 16x16 accumulator tile, C row l15 and columns 4 q4 + r (register r); v_permlane16_swap of two
    // neighbouring tiles widens that to 8 consecutive columns = one 16-byte store per lane and tile pair.  Rows past M fall
    // outside the descriptor, columns past N get an out-of-range offset: every wave issues exactly NST stores per tile.
    auto store_out = [&](const Tile& t) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc((void*)t.c, 0, t.c_bytes, 0x00020000);
        const int nbase = t.n0 + wn * (NT * 16);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int row_off = (wm * (MT * 16) + mt * 16 + l15) * p.ldc * 2;
            if constexpr (EPI != 1) {
#pragma unroll
                for (int np = 0; np < NT / 2; ++np) {
                    const f32x4 a = acc[mt][2 * np], b = acc[mt][2 * np + 1];
                    const unsigned a0 = pack2bf_pk(a[0], a[1]), a1 = pack2bf_pk(a[2], a[3]);
                    const unsigned b0 = pack2bf_pk(b[0], b[1]), b1 = pack2bf_pk(b[2], b[3]);
                    auto r0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
                    auto r1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
                    const int col = nbase + (2 * np + (q4 & 1)) * 16 + (q4 >> 1) * 8;
                    const u32x4 o = {r0[0], r1[0], r0[1], r1[1]};
                    const int off = col < ncols_out ? row_off + col * 2 : (int)0x80000000u;
                    __builtin_amdgcn_raw_buffer_store_b128(o, rC, off, 0, 0);
                }
            } else {
#pragma unroll
                for (int j = 0; j < NT / 4; ++j) {  // 64 input columns = 32 of w1 | 32 of w3 -> 32 output columns
                    unsigned pk[2][2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const f32x4 x = acc[mt][4 * j + u], y = acc[mt][4 * j + 2 + u];
                        float vv[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r)  // reference rounding points (model.py:497-502 under bf16): w1 x, w3 x, silu, product
                            vv[r] = bfr(silu_f(bfr(x[r]))) * bfr(y[r]);
                        pk[u][0] = pack2bf_pk(vv[0], vv[1]);
                        pk[u][1] = pack2bf_pk(vv[2], vv[3]);
                    }
                    auto r0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
                    auto r1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
                    const int col = nbase / 2 + j * 32 + (q4 & 1) * 16 + (q4 >> 1) * 8;
                    const u32x4 o = {r0[0], r1[0], r0[1], r1[1]};
                    const int off = col < ncols_out ? row_off + col * 2 : (int)0x80000000u;
                    __builtin_amdgcn_raw_buffer_store_b128(o, rC, off, 0, 0);
                }
            }
        }
    };
    while (true) {
        body(std::true_type{}, wf, af, wf2, af2);
        body(std::false_type{}, wf2, af2, wf, af);
        for (int s = 2; s < ns; s += 2) {
            body(std::false_type{}, wf, af, wf2, af2);
            body(std::false_type{}, wf2, af2, wf, af);
        }
        store_out(cur);
        since_epi = 0;
        cv = next_valid(cv + (int)gridDim.x);
        if (cv >= ntiles) break;
        cur = setup(cv);
    }
    wait_vmcnt<0>();  // no LDS-DMA (the null ones of the last bodies included) may outlive the workgroup's LDS allocation
}

}  // namespace lt_gemm
