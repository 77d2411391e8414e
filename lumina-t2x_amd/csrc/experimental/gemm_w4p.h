// EXPERIMENTAL (moved out of the product library in round 3): the persistent 4-wave kernel on 32x32x16 MFMAs, variants 13 / 14.
// Round 2 measured it against its 16x16x32 successor gemm_bf16_w4q (profiles/r02/vendor_vs_engine_pmc_r02.log: 428 us at 1.51 GHz vs
// 377 us at 1.82 GHz on the W1|W3 shape; profiles/r03/power_probe_gemm_attention_watts.log: 1.12 vs 0.99 J/TFLOP at the same
// 1350 W) and the dispatcher has not picked it since.  Included by experimental/gemm_experimental.hip after gemm_device.h.
#pragma once

namespace lt_gemm {

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
    stagger_start(p.stagger);
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

}  // namespace lt_gemm
