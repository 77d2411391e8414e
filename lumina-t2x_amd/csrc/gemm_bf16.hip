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

#include "gemm_device.h"
#include "options.h"
#include <mutex>
#include <map>
#include <set>
#include <utility>

namespace lt_gemm {
// explicit instantiations of the PRODUCT kernels (hipcc 7.2 does not emit the kernel body for address-only uses inside another
// template).  Everything else this file used to carry lives in experimental/gemm_experimental.hip (make EXPERIMENTAL=1).
template __global__ void gemm_bf16_tn<2, 4, 4, 2, 0>(GemmArgs);   // 256 x 256, 8 waves, classic double-buffered loop
template __global__ void gemm_bf16_tn<4, 3, 2, 3, 0>(GemmArgs);   // 256 x 288, 12 waves
template __global__ void gemm_bf16_tn<2, 4, 4, 2, 2>(GemmArgs);   // ... with the V^T epilogue (EPI 2)
template __global__ void gemm_bf16_tn<4, 3, 2, 3, 2>(GemmArgs);
template __global__ void gemm_bf16_tn<2, 4, 4, 2, 1>(GemmArgs);   // SwiGLU on the classic loop (explicit variant 1)
template __global__ void gemm_bf16_pp<2, 4, 4, 2, 0>(GemmArgs);   // 8-wave ping-pong (explicit variant 3)
template __global__ void gemm_bf16_pp<2, 4, 4, 2, 1>(GemmArgs);   // ... grouped (MoE expert) SwiGLU GEMM
template __global__ void gemm_bf16_w4q<0, 8>(GemmArgs);           // persistent 4 waves on 16x16x32 MFMAs: 256 x 256 tiles
template __global__ void gemm_bf16_w4q<0, 9>(GemmArgs);           // ... 256 x 288 tiles (N = 2304 / 6912: whole rounds over 256 CUs)
template __global__ void gemm_bf16_w4q<1, 8>(GemmArgs);           // ... SwiGLU
template __global__ void gemm_bf16_w4q<3, 9>(GemmArgs);           // ... fused QKV projection: plain tiles for Q | K, V^T tiles for V
template __global__ void gemm_bf16_w4q<3, 8>(GemmArgs);           // ... the same on 256-wide tiles (Flag-DiT 5B: 3072-wide Q, K, V)
template __global__ void gemm_bf16_w4q<0, 8, false, true>(GemmArgs);  // ... grouped (MoE experts' W2: valid row tiles only, per-tile expert weights)
template __global__ void gemm_bf16_w4q<1, 8, false, true>(GemmArgs);  // ... grouped + gather-on-load + SwiGLU (the experts' w1 | w3)
template __global__ void gemm_bf16_pp<2, 4, 2, 1, 0, false, 0, 1, 4>(GemmArgs);  // 128 x 128, small-M problems, one barrier per 64-deep slab
template __global__ void gemm_bf16_pp<4, 2, 1, 2, 1, false, 0, 1, 4>(GemmArgs);  // 128 x 128 with the SwiGLU epilogue (needs NT even)
template __global__ void gemm_bf16_pp<2, 4, 1, 1, 0, false, 0, 1, 4>(GemmArgs);  //  64 x 128
}  // namespace lt_gemm

int num_cus() {
    static int per_dev[64] = {0};  // one entry per device id: a process may drive several devices
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return 256; }
    if (per_dev[dev] == 0) {
        hipDeviceProp_t prop;
        int n = 0;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        per_dev[dev] = n > 0 ? n : 256;
    }
    return per_dev[dev];
}


// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per (device, function): a process that drives several devices must set it on each.
// The (device, function) pair is recorded only AFTER the call succeeded (a failed first attempt used to mark the pair as done,
// every later launch then died with too little dynamic LDS and the original error was lost), keyed by the real device id.
// The pair remembers the LARGEST size it was raised to (attn_small_fused_kernel's LDS need grows with the token count - 55 KB at
// 256, 110 KB at 512 tokens - and a process that ran the small shape first never raised the attribute again): a larger request re-sets it.
int ensure_dynamic_lds(const void* fn, int bytes) {
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, int> done;
    int dev = 0;
    LT_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    auto it = done.find({dev, fn});
    if (it != done.end() && it->second >= bytes) return 0;
    LT_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done[{dev, fn}] = bytes;
    return 0;
}

// the fused QKV projection (epilogue 3) runs on the persistent kernel only: whole 288- or 256-wide tiles on both sides of the split,
// a 32-row pair of V^T tiles inside one sample, at least one tile per CU, offsets that fit the buffer instructions' 32-bit arithmetic.
// Returns the tile width (288 / 256) or 0.
int gemm_qkv_fused_tile(const GemmArgs& a) {
    if (a.tile_expert || a.trace || a.bias_dtype >= 0 || a.K % 64 != 0 || a.K < (LT_W4Q_PD == 4 ? 192 : 128) || !a.VT) return 0;
    if (a.vt_split <= 0 || a.N <= a.vt_split) return 0;
    int bn = 0;
    if (a.vt_split % 288 == 0 && (a.N - a.vt_split) % 288 == 0) bn = 288;
    else if (a.vt_split % 256 == 0 && (a.N - a.vt_split) % 256 == 0) bn = 256;
    else return 0;
    if (a.vt_hd <= 0 || (a.N - a.vt_split) % a.vt_hd != 0 || a.vt_tokens <= 0 || a.vt_tokens % 64 != 0 || a.M % a.vt_tokens != 0) return 0;
    if (a.vt_npad != a.vt_tokens) return 0;
    if (255LL * a.ldc * 2 + (long long)a.N * 2 >= 0x7fffffffLL) return 0;
    if ((long long)a.M * (a.N - a.vt_split) * 2 >= 0x7fffffffLL) return 0;
    return (long long)((a.M + 255) / 256) * (a.N / bn) >= num_cus() ? bn : 0;
}
bool gemm_qkv_fusable(const GemmArgs& a) { return gemm_qkv_fused_tile(a) != 0; }
int gemm_qkv_tile_width(const GemmArgs& a) { return gemm_qkv_fused_tile(a); }

namespace {
using lt_gemm::gemm_bf16_tn;
using lt_gemm::gemm_bf16_pp;
using lt_gemm::gemm_bf16_w4q;

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

// ev0 / ev1 (optional): start / stop events attached to THIS dispatch packet (hipExtLaunchKernelGGL) - the timestamps come
// from the dispatch's own completion signal, no extra barrier packets in the queue (event records around a launch cost
// tens of microseconds of queue idle time each on this stack)
template <int WM, int WN, int MT, int NT, int EPI, bool PP, int MODE = 0, int KS = 2>
int launch_cfg(const GemmArgs& a, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    constexpr int SMEM = PP ? 4 * (BM + BN) * 32 * KS : 2 * (BM + BN) * 128;
    const void* fn;
    if constexpr (PP) fn = (const void*)gemm_bf16_pp<WM, WN, MT, NT, EPI, false, 0, MODE, KS>;
    else fn = (const void*)gemm_bf16_tn<WM, WN, MT, NT, EPI>;
    if (ensure_dynamic_lds(fn, SMEM)) return 1;
    const int TM = (a.M + BM - 1) / BM, TN = (a.N + BN - 1) / BN;
    const dim3 grid(TM * TN * (a.split_k >= 2 ? a.split_k : 1)), block(WM * WN * 64);
    if constexpr (PP) {
        if (ev0) hipExtLaunchKernelGGL((gemm_bf16_pp<WM, WN, MT, NT, EPI, false, 0, MODE, KS>), grid, block, SMEM, stream, ev0, ev1, 0, a);
        else hipLaunchKernelGGL((gemm_bf16_pp<WM, WN, MT, NT, EPI, false, 0, MODE, KS>), grid, block, SMEM, stream, a);
    } else {
        if (ev0) hipExtLaunchKernelGGL((gemm_bf16_tn<WM, WN, MT, NT, EPI>), grid, block, SMEM, stream, ev0, ev1, 0, a);
        else hipLaunchKernelGGL((gemm_bf16_tn<WM, WN, MT, NT, EPI>), grid, block, SMEM, stream, a);
    }
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}


template <int EPI, int NW16, bool GROUPED = false>
int launch_w4q(const GemmArgs& a, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    // GROUPED: + two 1-KiB gather-map slots, the list of valid row tiles (<= 1024) and its count behind the slab ring
    constexpr int BN = 32 * NW16, SMEM = 4 * (256 + BN) * 64 + (GROUPED ? 2048 + 4096 + 16 : 0);
    if (ensure_dynamic_lds((const void*)gemm_bf16_w4q<EPI, NW16, false, GROUPED>, SMEM)) return 1;
    const int tiles = ((a.M + 255) / 256) * ((a.N + BN - 1) / BN);
    const int cus = num_cus();
    const dim3 grid(tiles < cus ? tiles : cus), block(256);
    if (ev0) hipExtLaunchKernelGGL((gemm_bf16_w4q<EPI, NW16, false, GROUPED>), grid, block, SMEM, stream, ev0, ev1, 0, a);
    else hipLaunchKernelGGL((gemm_bf16_w4q<EPI, NW16, false, GROUPED>), grid, block, SMEM, stream, a);
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---- kernel selection (one place: launch_gemm_bf16 launches what choose() names, lt_gemm_describe prints it) -------------
enum GemmKernel {
    GK_TN256, GK_TN288, GK_TN256_VT, GK_TN288_VT, GK_TN256_SWIGLU, GK_PP256, GK_PP256_SWIGLU,
    GK_S128, GK_S128_SWIGLU, GK_S64, GK_W4Q256, GK_W4Q288, GK_W4Q256_SWIGLU, GK_W4Q288_QKV, GK_W4Q256_QKV, GK_W4Q256_GROUPED,
    GK_W4Q256_SWIGLU_GROUPED, GK_REMOVED, GK_NONE
};
const char* const kGemmKernelName[] = {
    "gemm_bf16_tn<2,4,4,2,0> (256x256, 8 waves)", "gemm_bf16_tn<4,3,2,3,0> (256x288, 12 waves)",
    "gemm_bf16_tn<2,4,4,2,2> (256x256, V^T epilogue)", "gemm_bf16_tn<4,3,2,3,2> (256x288, V^T epilogue)",
    "gemm_bf16_tn<2,4,4,2,1> (256x256, SwiGLU)", "gemm_bf16_pp<2,4,4,2,0> (256x256 ping-pong)",
    "gemm_bf16_pp<2,4,4,2,1> (256x256 ping-pong, SwiGLU)",
    "gemm_bf16_pp<2,4,2,1,0,..,1,4> (128x128)", "gemm_bf16_pp<4,2,1,2,1,..,1,4> (128x128, SwiGLU)",
    "gemm_bf16_pp<2,4,1,1,0,..,1,4> (64x128)", "gemm_bf16_w4q<0,8> (persistent 4 waves, 16x16x32 MFMA, 256x256)",
    "gemm_bf16_w4q<0,9> (persistent 4 waves, 16x16x32 MFMA, 256x288)", "gemm_bf16_w4q<1,8> (persistent 4 waves, 16x16x32 MFMA, 256x256, SwiGLU)",
    "gemm_bf16_w4q<3,9> (persistent 4 waves, 16x16x32 MFMA, 256x288, fused QKV: plain Q|K tiles + V^T tiles)",
    "gemm_bf16_w4q<3,8> (persistent 4 waves, 16x16x32 MFMA, 256x256, fused QKV: plain Q|K tiles + V^T tiles)",
    "gemm_bf16_w4q<0,8,grouped> (persistent 4 waves, 16x16x32 MFMA, 256x256, expert segments)",
    "gemm_bf16_w4q<1,8,grouped> (persistent 4 waves, 16x16x32 MFMA, 256x256, expert segments, gather-on-load, SwiGLU)", "removed study kernel", "none"};

// Options (options.h; lt_set_option / lt_engine_set_option):
//   gemm_variant      tile shape when the caller passes 0: 0 auto, 1 = 256x256, 2 = 256x288
//   gemm_w4q          1 (default): large dense GEMMs (>= one tile per CU) run on the persistent 16x16x32 kernel (256 / 288-wide tiles)
//   gemm_splitk       split-K of the 512-row-class GEMMs (round 4).  A 64 x 128 workgroup of the O / W2 projections at 512 rows stages
//                     0.6 / 1.6 MB through a CU that fills at ~50 GB/s: cutting K in two halves the bytes per workgroup and
//                     doubles the busy CUs (96 -> 192); the second-arriving half adds the first one's fp32 partial itself (counter per tile)
//   gemm_w4q_grouped  1 (default): grouped (MoE expert) GEMMs with >= 1.5 tiles per CU as well (round 4; at one tile per CU - the 600M MoE
//                     at 256 tokens - the 8-wave tiles are 4 % faster); 2: from 2 tiles per CU on

// the persistent kernel's grouped mode: expert segments (and gather-on-load) - one descriptor over all of A, lane offsets < 2^31
// (gather: < 2^30, the out-of-range offset of a padding row is 2^30), <= 1024 row tiles in the LDS list, K >= 256 (the map of the
// next tile must have landed two barriers before the DMA stream crosses into it)
bool w4q_grouped_ok(const GemmArgs& a, int epilogue) {
    if (!a.tile_expert || a.trace || a.bias_dtype >= 0 || a.K % 64 != 0 || a.K < 256 || (epilogue != 0 && epilogue != 1)) return false;
    if ((a.M + 255) / 256 > 1024 || a.M % 256 != 0) return false;
    if (255LL * a.ldc * 2 + (long long)a.N * 2 >= 0x7fffffffLL) return false;
    return a.a_row_map ? a.a_map_rows > 0 && (long long)a.a_map_rows * a.lda * 2 < 0x40000000LL : (long long)a.M * a.lda * 2 < 0x40000000LL;
}

// variant: 0 = auto; 1 / 2 = 256x256 / 256x288 classic loop; 3 = 256x256 8-wave ping-pong; 7 / 8 = 128x128 / 64x128 small-M tiles;
//          15 / 16 = persistent 4 waves on 16x16x32 MFMAs, 256x256 / 256x288 tiles;
//          4, 5, 6, 9 .. 14, 17, 18 = study kernels of rounds 1-3, removed (git history)
GemmKernel choose(const GemmArgs& a, int epilogue, int variant) {
    const bool w4p_ok = !a.tile_expert && !a.trace && a.bias_dtype < 0 && a.K % 64 == 0 && a.K >= (LT_W4Q_PD == 4 ? 192 : 128) &&
                        255LL * a.ldc * 2 + (long long)a.N * 2 < 0x7fffffffLL && epilogue != 2;
    if (epilogue == 3) { const int bn = gemm_qkv_fused_tile(a); return bn == 288 ? GK_W4Q288_QKV : bn == 256 ? GK_W4Q256_QKV : GK_NONE; }
    if (a.tile_expert && (variant == 15 || (variant == 0 && lt_opt(OPT_GEMM_VARIANT) == 0 && lt_opt(OPT_GEMM_W4Q) && lt_opt(OPT_GEMM_W4Q_GROUPED) &&
                                             2LL * ((a.M + 255) / 256) * ((a.N + 255) / 256) >= (lt_opt(OPT_GEMM_W4Q_GROUPED) == 2 ? 4LL : 3LL) * num_cus()))) {
        if (w4q_grouped_ok(a, epilogue)) return epilogue == 1 ? GK_W4Q256_SWIGLU_GROUPED : GK_W4Q256_GROUPED;
        if (variant == 15) return GK_NONE;
    }
    if (variant == 15 || variant == 16) {
        if (!w4p_ok || (epilogue == 1 && variant == 16)) return GK_NONE;
        return epilogue == 1 ? GK_W4Q256_SWIGLU : (variant == 15 ? GK_W4Q256 : GK_W4Q288);
    }
    if (variant == 4 || variant == 5 || variant == 6 || (variant >= 9 && variant <= 14) || variant == 17 || variant == 18) return GK_REMOVED;
    const int cus = num_cus();
    const long long t256 = (long long)((a.M + 255) / 256) * ((a.N + 255) / 256);
    const long long t128 = (long long)((a.M + 127) / 128) * ((a.N + 127) / 128);
    if (epilogue == 2) {  // V^T epilogue: the two classic tile shapes
        int v = variant == 0 ? lt_opt(OPT_GEMM_VARIANT) : variant;
        if (v != 1 && v != 2) {
            const long long t288 = (long long)((a.M + 255) / 256) * ((a.N + 287) / 288);
            v = ((t288 + cus - 1) / cus) * 288 < ((t256 + cus - 1) / cus) * 256 ? 2 : 1;
        }
        return v == 2 ? GK_TN288_VT : GK_TN256_VT;
    }
    // small-M problems (cfg 1 / cfg 5: 512 rows): 256-wide tiles leave most CUs idle and every workgroup is a long serial
    // K loop that streams weights nobody else re-uses; 128 x 128 (or 64 x 128) tiles give 4-8x the workgroups, each with
    // its own 3-slab prefetch window, single-barrier rendezvous loop over 64-deep slabs (profiles/r01/opbench_small_m.log)
    const bool small = variant == 7 || variant == 8 || (variant == 0 && lt_opt(OPT_GEMM_VARIANT) == 0 && 2 * t256 <= cus);
    if (small) {
        if (epilogue == 1) return GK_S128_SWIGLU;
        return (variant == 8 || (variant == 0 && 2 * t128 <= cus)) ? GK_S64 : GK_S128;
    }
    if (variant == 0 && lt_opt(OPT_GEMM_VARIANT) == 0 && lt_opt(OPT_GEMM_W4Q) && w4p_ok && t256 >= cus) {
        if (epilogue == 1) return GK_W4Q256_SWIGLU;
        const long long t288 = (long long)((a.M + 255) / 256) * ((a.N + 287) / 288);
        return ((t288 + cus - 1) / cus) * 288 < ((t256 + cus - 1) / cus) * 256 ? GK_W4Q288 : GK_W4Q256;
    }
    if (epilogue == 1) {
        if (variant == 1) return GK_TN256_SWIGLU;
        return GK_PP256_SWIGLU;
    }
    if (variant == 3) return GK_PP256;
    if (variant == 0) variant = lt_opt(OPT_GEMM_VARIANT);
    if (variant == 0) {
        const long long t288 = (long long)((a.M + 255) / 256) * ((a.N + 287) / 288);
        const long long c256 = ((t256 + cus - 1) / cus) * 256, c288 = ((t288 + cus - 1) / cus) * 288;
        variant = c288 < c256 ? 2 : 1;
    }
    return variant == 2 ? GK_TN288 : GK_TN256;
}

// The ONE place that decides what a launch runs: the kernel choose() names, then the split-K overrides (the launcher, the
// weight-panel riders and lt_gemm_describe each modelled the split on their own, and the four-way split's kernel change was missed by two).
struct GemmPlan { GemmKernel k; int split_k; };
GemmPlan plan(const GemmArgs& a, int epilogue, int variant) {
    GemmPlan p{choose(a, epilogue, variant), 0};
    // split-K: dense plain-epilogue problems on the 64 x 128 tiles whose two halves still fit one round of the CUs, K >= 1024
    if (p.k == GK_S64 && lt_opt(OPT_GEMM_SPLITK) && (variant == 0 || variant == 8) && a.splitk_part && a.splitk_cnt && !a.tile_expert && !a.a_row_map && a.bias_dtype < 0 &&
        a.K >= 1024 && a.K % 512 == 0) {
        const int tiles = ((a.M + 63) / 64) * ((a.N + 127) / 128);
        if ((2 * tiles <= num_cus() || lt_opt(OPT_GEMM_SPLITK) == 2) && tiles <= a.splitk_tiles) p.split_k = 2;
    }
    // ... and four ways on the 128 x 128 tile where K is long (round 5, option gemm_splitk4; the 512-row w2 projection of the 600M models:
    // K = 4096): a 64 x 128 half stages (64 + 128) x 2048 x 2 = 786 KB through its CU, a quarter of a 128 x 128 tile (128 + 128) x 1024 x 2 =
    // 524 KB - the same 192 workgroups, a third less per workgroup (NOTEBOOK.md 9.3 priced it at 4 us per layer).  The last arriver sums
    // the four partials in K order, so the result is independent of the arrival order here too.
    if ((p.k == GK_S64 || p.k == GK_S128) && epilogue == 0 && variant == 0 && lt_opt(OPT_GEMM_SPLITK) && lt_opt(OPT_GEMM_SPLITK4) && a.splitk_part && a.splitk_cnt &&
        !a.tile_expert && !a.a_row_map && a.bias_dtype < 0 && !a.rowstat && a.K >= 4096 && a.K % 1024 == 0) {
        const int t128 = ((a.M + 127) / 128) * ((a.N + 127) / 128);
        if (4 * t128 <= num_cus() && 4 * t128 <= a.splitk_tiles) { p.k = GK_S128; p.split_k = 4; }  // (a 128 x 128 part takes two [2][64 x 128] slots, four parts per tile)
    }
    return p;
}
}  // namespace

const char* lt_gemm_describe(const GemmArgs& a, int epilogue, int variant) { return kGemmKernelName[plan(a, epilogue, variant).k]; }
bool gemm_runs_w4q_grouped(const GemmArgs& a, int epilogue) {
    const GemmKernel k = plan(a, epilogue, 0).k;
    return k == GK_W4Q256_GROUPED || k == GK_W4Q256_SWIGLU_GROUPED;
}
bool gemm_runs_w4q_dense(const GemmArgs& a, int epilogue) {
    const GemmKernel k = plan(a, epilogue, 0).k;
    return k == GK_W4Q256 || k == GK_W4Q288 || k == GK_W4Q256_SWIGLU || k == GK_W4Q288_QKV || k == GK_W4Q256_QKV;
}

int launch_gemm_bf16(const GemmArgs& a0, int epilogue, int variant, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    GemmArgs a = a0;
    a.stagger = lt_opt(OPT_GEMM_STAGGER);
    a.group_rows = lt_opt(OPT_GEMM_GROUP);
    LT_REQUIRE(a.K % BK == 0 && a.K > 0, "gemm: K=%d must be a positive multiple of %d", a.K, BK);
    LT_REQUIRE(a.lda % 8 == 0 && a.ldw % 8 == 0, "gemm: lda/ldw must be multiples of 8");
    LT_REQUIRE(epilogue >= 0 && epilogue <= 3, "gemm: unknown epilogue %d", epilogue);
    LT_REQUIRE(epilogue != 3 || gemm_qkv_fusable(a), "gemm: the fused QKV epilogue needs whole 288- or 256-column tiles on both sides of the split, tokens per sample %% 64 == 0 and at least one tile per CU");
    if (epilogue == 2) {
        LT_REQUIRE(a.bias_dtype < 0 && !a.tile_expert && !a.trace, "gemm: the V^T epilogue takes dense problems without bias");
        LT_REQUIRE(a.vt_hd > 0 && a.vt_hd % 8 == 0 && a.N % a.vt_hd == 0, "gemm: V^T epilogue: N=%d must be whole heads of vt_hd=%d", a.N, a.vt_hd);
        LT_REQUIRE(a.vt_tokens > 0 && a.vt_tokens % 64 == 0 && a.M % a.vt_tokens == 0 && a.vt_npad == a.vt_tokens,
                   "gemm: V^T epilogue needs tokens per sample %% 64 == 0 and no key padding (tokens %d, padded %d, M %d)", a.vt_tokens, a.vt_npad, a.M);
    } else {
        LT_REQUIRE(a.N % 8 == 0 && a.ldc % 8 == 0, "gemm: N=%d and ldc=%d must be multiples of 8", a.N, a.ldc);
    }
    LT_REQUIRE(epilogue != 1 || (a.N % 64 == 0 && a.bias_dtype < 0), "gemm: swiglu epilogue needs N %% 64 == 0, no bias");
    LT_REQUIRE(variant >= 0 && variant <= 18, "gemm: unknown variant %d", variant);
    const GemmPlan pl = plan(a, epilogue, variant);
    GemmKernel k = pl.k;
    a.split_k = pl.split_k;
    if (a.rowstat) {
        LT_REQUIRE((k == GK_S128 || k == GK_S64) && epilogue == 0, "gemm: rowstat is written by the small-M tiles' plain epilogue only (this problem runs %s)", kGemmKernelName[k]);
        LT_REQUIRE(a.rowstat_slots >= (a.N + 127) / 128, "gemm: rowstat_slots %d < %d column tiles", a.rowstat_slots, (a.N + 127) / 128);
    }
    if (a.ystat) {
        LT_REQUIRE((k == GK_W4Q256 || k == GK_W4Q288) && epilogue == 0, "gemm: ystat is written by the persistent kernel's plain dense tiles only (this problem runs %s)", kGemmKernelName[k]);
        const int want = 2 * ((a.N + (k == GK_W4Q288 ? 287 : 255)) / (k == GK_W4Q288 ? 288 : 256));
        LT_REQUIRE(a.ystat_slots == want && (long long)a.M * want * 4 < 0x7fffffffLL, "gemm: ystat_slots %d, this launch fills %d per row (gemm_ystat_slots)", a.ystat_slots, want);
    }
    if (a.a_row_map) {  // gather-on-load lives in the ping-pong kernels' staging (the grouped SwiGLU GEMM of the MoE layers)
        LT_REQUIRE(k == GK_PP256_SWIGLU || k == GK_PP256 || k == GK_S128 || k == GK_S128_SWIGLU || k == GK_S64 || k == GK_W4Q256_GROUPED ||
                   k == GK_W4Q256_SWIGLU_GROUPED,
                   "gemm: a_row_map is supported by the gemm_bf16_pp kernels and the grouped persistent kernel only (this problem runs %s)", kGemmKernelName[k]);
        LT_REQUIRE(a.a_map_rows > 0 && (long long)a.a_map_rows * a.lda * 2 < 0x40000000LL, "gemm: a_row_map needs 0 < a_map_rows * lda * 2 < 2^30");
    }
    if (a.pair_ab || a.pair_c) {
        LT_REQUIRE(k == GK_W4Q256 || k == GK_W4Q288 || k == GK_W4Q256_SWIGLU || k == GK_W4Q288_QKV || k == GK_W4Q256_QKV || k == GK_W4Q256_GROUPED ||
                   k == GK_W4Q256_SWIGLU_GROUPED,
                   "gemm: the row-pair-interleaved operand layout is read by the persistent kernel only (this problem runs %s)", kGemmKernelName[k]);
        LT_REQUIRE(!(a.pair_ab & 1) || !a.a_row_map, "gemm: a gathered A operand (a_row_map) cannot be in the pair layout");
        LT_REQUIRE(a.M % 2 == 0 && a.N % 2 == 0 && a.K % 32 == 0 && (!a.pair_c || (epilogue == 1 && a.ldc % 32 == 0)),
                   "gemm: pair layout needs even row counts, K %% 32 == 0 and (pair_c) the SwiGLU epilogue with ldc %% 32 == 0");
    }
    switch (k) {
        case GK_TN256: return launch_cfg<2, 4, 4, 2, 0, false>(a, stream, ev0, ev1);
        case GK_TN288: return launch_cfg<4, 3, 2, 3, 0, false>(a, stream, ev0, ev1);
        case GK_TN256_VT: return launch_cfg<2, 4, 4, 2, 2, false>(a, stream, ev0, ev1);
        case GK_TN288_VT: return launch_cfg<4, 3, 2, 3, 2, false>(a, stream, ev0, ev1);
        case GK_TN256_SWIGLU: return launch_cfg<2, 4, 4, 2, 1, false>(a, stream, ev0, ev1);
        case GK_PP256: return launch_cfg<2, 4, 4, 2, 0, true>(a, stream, ev0, ev1);
        case GK_PP256_SWIGLU: return launch_cfg<2, 4, 4, 2, 1, true>(a, stream, ev0, ev1);
        case GK_S128: return launch_cfg<2, 4, 2, 1, 0, true, 1, 4>(a, stream, ev0, ev1);
        case GK_S128_SWIGLU: return launch_cfg<4, 2, 1, 2, 1, true, 1, 4>(a, stream, ev0, ev1);
        case GK_S64: return launch_cfg<2, 4, 1, 1, 0, true, 1, 4>(a, stream, ev0, ev1);
        case GK_W4Q256: return launch_w4q<0, 8>(a, stream, ev0, ev1);
        case GK_W4Q288: return launch_w4q<0, 9>(a, stream, ev0, ev1);
        case GK_W4Q256_SWIGLU: return launch_w4q<1, 8>(a, stream, ev0, ev1);
        case GK_W4Q288_QKV: return launch_w4q<3, 9>(a, stream, ev0, ev1);
        case GK_W4Q256_QKV: return launch_w4q<3, 8>(a, stream, ev0, ev1);
        case GK_W4Q256_GROUPED: return launch_w4q<0, 8, true>(a, stream, ev0, ev1);
        case GK_W4Q256_SWIGLU_GROUPED: return launch_w4q<1, 8, true>(a, stream, ev0, ev1);
        case GK_REMOVED:
            lt_set_error("gemm: variant %d was a study kernel of csrc/experimental/ (rounds 1-3), removed in round 5; product variants: 0 auto, 1, 2, 3, 7, 8, 15, 16", variant);
            return 2;
        default:
            lt_set_error("gemm: variant %d cannot run this problem (persistent 4-wave kernels: dense, no bias, K %% 64 == 0, K >= 128)", variant);
            return 2;
    }
}

// ---- weight-panel prefetch for the 512-row-class GEMMs (round 4 experiment) ---------------------------------
// A small-M workgroup stages its operands at ~50 GB/s when they come from HBM and up to ~140 GB/s when they sit in ITS XCD's L2
// (scripts/ubench/fill_rate.hip).  This kernel walks the SAME workgroup -> tile map as the GEMM that follows (same grid, same
// tile_coords, i.e. same XCD per tile) and reads each tile's W panel - every workgroup of a tile column takes its share of the rows -
// so that the panel is resident in the L2 of the XCD whose workgroups will stage it.  Loads only, results discarded.
__global__ __launch_bounds__(256) void gemm_prefetch_w_kernel(PrefetchRider r) { prefetch_w_block(r, (int)blockIdx.x); }

// lt_set_option "gemm_prefetch": 3 (default) = the W panels of the 512-row-class GEMMs are read by rider workgroups of the row kernel
// that precedes the GEMM (cfg 1 -1.6 %, cfg 5 -1..3 % on a fast-class box, -6.5 % on a slow one); 0 = off; 1 = a prefetch launch right in
// front of every small-M GEMM (same stream: the upper bound experiment); 2 = on a side stream beside the preceding kernel (loses 33 %)

int gemm_ystat_slots(const GemmArgs& a, int epilogue) {
    if (epilogue != 0) return 0;
    const GemmKernel k = plan(a, epilogue, 0).k;
    if (k == GK_W4Q288) return 2 * ((a.N + 287) / 288);
    if (k == GK_W4Q256) return 2 * ((a.N + 255) / 256);
    return 0;
}

bool gemm_is_small_m(const GemmArgs& a, int epilogue) {
    const GemmKernel k = plan(a, epilogue, 0).k;
    return k == GK_S64 || k == GK_S128;
}

bool gemm_prefetch_rider(const GemmArgs& a0, int epilogue, PrefetchRider* r) {
    GemmArgs a = a0;
    const GemmPlan pl = plan(a, epilogue, 0);  // the kernel AND the split the launch will really use
    const GemmKernel k = pl.k;
    int BM = 0;
    if (k == GK_S64) BM = 64;
    else if (k == GK_S128 || k == GK_S128_SWIGLU) BM = 128;
    else return false;  // not a small-M launch: nothing to do
    if (a.tile_expert || a.a_row_map) return false;
    const int split = pl.split_k;
    r->W = a.W; r->N = a.N; r->K = a.K; r->ldw = a.ldw; r->BN = 128;
    r->TM = (a.M + BM - 1) / BM; r->TN = (a.N + 127) / 128; r->split = split;
    r->blocks = r->TM * r->TN * (split >= 2 ? split : 1);
    r->first = 0;
    return true;
}

int launch_gemm_prefetch_w(const GemmArgs& a, int epilogue, hipStream_t stream) {
    PrefetchRider r;
    if (!gemm_prefetch_rider(a, epilogue, &r)) return 0;
    hipLaunchKernelGGL(gemm_prefetch_w_kernel, dim3(r.blocks), dim3(256), 0, stream, r);
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_pack_w13(const u16* w1, const u16* w3, u16* out, int F, int K, hipStream_t stream) {
    LT_REQUIRE(F % 32 == 0 && K % 8 == 0, "pack_w13: F %% 32 and K %% 8 required");
    hipLaunchKernelGGL(pack_w13_kernel, dim3(1024), dim3(256), 0, stream, w1, w3, out, F, K);
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}
