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
    const int ks = r.split == 2 ? blk / ntile : 0;
    const int bid = blk - ks * ntile;
    int tm, tn;
    tile_coords(bid, ntile, r.TM, r.TN, tm, tn);
    const int first_m = (tm / 4) * 4, gsz = min(4, r.TM - first_m), part = tm - first_m;
    const int n0 = tn * r.BN, n1 = min(r.N, n0 + r.BN);
    const int k0 = r.split == 2 ? ks * (r.K / 2) : 0, kw = r.split == 2 ? r.K / 2 : r.K;  // elements
    const int cpr = kw / 8;                                                               // 16-byte chunks per row
    const long long total = (long long)(n1 - n0) * cpr;
    unsigned acc = 0;
    for (long long i = (long long)part * 256 + threadIdx.x; i < total; i += (long long)gsz * 256) {
        const int row = (int)(i / cpr), c = (int)(i - (long long)row * cpr);
        const uint4 v = *(const uint4*)(r.W + (size_t)(n0 + row) * r.ldw + k0 + c * 8);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x9e3779b9u) asm volatile("s_nop 0");  // keep the loads
}
