// Workgroup -> tile order of the GEMM kernels, and the weight-panel prefetch that walks the same order (shared by gemm_device.h, the
// stand-alone prefetch kernel in gemm_bf16.hip and the kernels that carry prefetch workgroups as riders in their own grid, norm.hip).
#pragma once
#include "common.h"
#include "kernels.h"

// XCD-aware order: workgroup bid runs on XCD bid % 8 (round-robin dispatch); each XCD walks a contiguous run of the (tile-row group,
// tile column) list, G tile rows per group, so that the tiles resident on one XCD share A row panels and W column panels in its L2.
__device__ __forceinline__ void tile_coords(int bid, int nwg, int TM, int TN, int& tm, int& tn, int G = 4) {
    const int NX = 8;
    const int xcd = bid % NX, idx = bid / NX;
    const int q = nwg / NX, r = nwg % NX;
    const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int per_group = G * TN;
    const int g = L / per_group;
    const int first_m = g * G;
    const int gsz = min(G, TM - first_m);
    const int in = L - g * per_group;
    tm = first_m + in % gsz;
    tn = in / gsz;
}

// One workgroup (256 threads) of the weight-panel prefetch of a small-M GEMM (round 4): `blk` is the blockIdx of the
// GEMM workgroup whose W panel is read - same tile, and, run from a block with the same index mod 8, the same XCD, i.e. the L2 that
// workgroup will stage from.  The workgroups of a tile column inside a tile-row group share the panel's rows.  Loads only.
__device__ __forceinline__ void prefetch_w_block(const PrefetchRider& r, int blk) {
    const int ntile = r.TM * r.TN;
    const int S = r.split >= 2 ? r.split : 1;  // K parts per tile (GemmArgs::split_k: 2 or 4); part ks of a tile is workgroup tile + ntile * ks
    const int ks = blk / ntile;
    const int bid = blk - ks * ntile;
    int tm, tn;
    tile_coords(bid, ntile, r.TM, r.TN, tm, tn);
    const int first_m = (tm / 4) * 4, gsz = min(4, r.TM - first_m), part = tm - first_m;
    const int n0 = tn * r.BN, n1 = min(r.N, n0 + r.BN);
    const int k0 = ks * (r.K / S), kw = r.K / S;  // elements
    // ONE 4-byte load per 128-byte line and lane: a wave instruction touches 64 lines (8 KB of panel), so a 128 x 1536 panel is three
    // instructions for each of the 16 waves that share it, all in flight at once - one memory round trip.  (First form: a linear
    // 16-byte-chunk index per thread, one dependent load and one integer division at a time - 12.5 us for a 14-25 MB panel set, longer
    // than the 7 us row kernel it rides in; second form, 16-byte loads with two rows in flight per wave: 9 us, four round trips.)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nrow = n1 - n0;
    const int lpr = (kw + 63) / 64;  // lines per row
    const int total = nrow * lpr, step = gsz * 4 * 64;
    const char* base = (const char*)(r.W + (size_t)n0 * r.ldw + k0);
    unsigned acc = 0;
    for (int i0 = (part * 4 + wave) * 64 + lane; i0 - lane < total; i0 += 8 * step) {
        unsigned v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * step;
            const int row = i / lpr, c = i - row * lpr;
            v[u] = i < total ? *(const unsigned*)(base + (size_t)row * r.ldw * 2 + (size_t)c * 128) : 0u;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u];
    }
    if (acc == 0x9e3779b9u) asm volatile("s_nop 0");  // keep the loads
}
