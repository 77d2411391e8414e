/* lumina_dit_debug.h - A/B options and diagnostics of the MI355X (gfx950) Next-DiT denoising engine.
 *
 * NOT part of the drop-in boundary (include/lumina_dit.h): nothing here is needed to run the reference's path.  This header names
 * the options lt_set_option / lt_engine_set_option accept (A/B measurements, tests; every default is the measured-best setting) and
 * declares the one instrumented-kernel entry point.  The option table itself lives in lumina-t2x_amd/csrc/options.hip;
 * tests/test_abi.py keeps this text and the table in step (every name, its range and default).
 */
#ifndef LUMINA_DIT_DEBUG_H
#define LUMINA_DIT_DEBUG_H

#include "lumina_dit.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Options - name (range, default): meaning.
 *   "graph"             (0..2, 2): 1 one model evaluation is captured into a HIP graph per (shape, arguments, option generation) and replayed
 *                       | 2 the same above 1024 rows only: at 512 rows plain launches are 1.5-4.6 % faster (round 5 same-box A/B) | 0: eager
 *                       launches
 *   "attention_variant" (1..6, 4): 1 baseline | 2 VALU-diet | 3 ping-pong wave groups (hd 72 / 96) | 4 one wave per SIMD x 64 query
 *                       rows, asm-owned AGPRs (hd 72, 96 and 48 with whole 64-key tiles; variant 3 otherwise) | 6 = 4 with the hd-48
 *                       one-wave kernel forced at every size | 5 is refused (the PV-on-16x16x32 study kernel, removed in round 5)
 *   "qkv_post_fused"    (0..2, 2): 2 one launch for q / k post-processing + V transpose below 2048 rows (launch-bound regime), three
 *                       launches above | 1 always one launch | 0 always separate launches
 *   "qkv_vt_epilogue"   (0..1, 1): the V projection's epilogue writes the attention kernels' transposed, key-permuted V image directly
 *                       (no V transpose pass; needs tokens per sample % 64 == 0, large M) | 0: off
 *   "qkv_fused_gemm"    (0..1, 1): Q | K | V projection in one launch where the shapes are whole 256 x 288 (or 256 x 256) tiles
 *   "qk_post_pair"      (0..1, 1): q and k post-processing share one persistent launch (>= 2048 rows) | 0: two launches
 *   "attn_q_fused"      (0..1, 1): behind the fused QKV launch at head_dim 72 (2-D RoPE, qk_norm, whole 64-key tiles) q_norm + RoPE of
 *                       the queries happen in the attention kernel's prologue - the QKV GEMM's epilogue leaves per-row LayerNorm partial
 *                       sums, the K pass reduces them to (mean, rstd) - and q is never written head-major; the row statistics come from
 *                       (sum, sum of squares) instead of the two-pass form, so a few queries differ by one bf16 ulp from the "0" path
 *   "norm_specialize"   (0..1, 1): gated_residual_norm runs instantiations with its three mode switches fixed at compile time (the
 *                       engine's combinations at d = 1536 / 2304 / 3072; bit-identical, 34.4 -> 31.6 us) | 0: generic kernel
 *   "gemm_w4q"          (0..1, 1): large dense GEMMs run on the persistent 4-wave 16x16x32 kernel | 0: classic / ping-pong tiles
 *   "gemm_prefetch"     (0..3, 3): 3 at <= 1024 rows the weight panels of the QKV / O / W1|W3 projections are read into the L2 of the
 *                       XCDs that will stage them by extra workgroups of the row kernel in front of the GEMM | 0 off | 1 a serial
 *                       prefetch launch in front of every small-M GEMM (the upper-bound measurement form) | 2 is refused (the
 *                       side-stream form lost 33 % and was removed in round 5)
 *   "gemm_splitk"       (0..2, 1): 1 the 64 x 128 small-M tile splits K over two workgroups per tile when both halves fit one round of
 *                       the CUs (the 512-row O / W2 projections) | 2 whenever the workspace allows | 0 off
 *   "gemm_w4q_grouped"  (0..2, 1): 1 the grouped (mixture-of-experts) GEMMs run on the persistent kernel's grouped mode from 1.5 tiles
 *                       of 256 x 256 per CU on | 2 from 2 tiles per CU on (A/B) | 0 always the 8-wave ping-pong / classic tiles
 *   "gemm_group"        (0..64, 0): tile rows per group of the XCD-aware tile order, 0 = the built-in 4 (experiment, no measured effect)
 *   "gemm_stagger"      (0..256, 0): the persistent 4-wave GEMM kernels spread the start of the workgroups of an XCD over eight phases,
 *                       n * ~256 cycles apart (experiment: de-synchronises the tile-end store bursts; measured -0.5 % .. 0 by box)
 *   "gemm_variant"      (0..2, 0): tile shape of the classic kernels when the caller passes variant 0: 0 auto | 1 256x256 | 2 256x288
 *   "rmsnorm_apex"      (0..1, 0): rounding order of the weighted RMSNorms.  0 = the reference's vanilla class
 *                       (lumina_next_t2i/models/components.py:11-54): bf16(x * rstd) * w, two roundings.  1 = apex.FusedRMSNorm, which
 *                       the reference uses when apex is importable (components.py:6-9): bf16(x * rstd * w), the weight applied in fp32
 *                       before the one rounding - as SURVEY.md 8c describes it; DESIGN.md 6 explains why the default order is expected to
 *                       match an apex box too.  Meant for the text-conditional families (Next-DiT T2I, Flag-DiT: the only ones whose
 *                       reference imports apex) - set it per engine (lt_engine_set_option) and before lt_prepare_prompt, whose hoisted
 *                       text K / V go through the same norm
 *   "attn_small_fused"  (0..1, 1): class-conditional models at head_dim 48 and <= 512 tokens (the 600M ImageNet / MoE models at 256^2): q_norm,
 *                       k_norm, RoPE, the V transpose and the attention itself are ONE launch that reads the QKV projection's output and the
 *                       per-tile LayerNorm partials its GEMM epilogue left (round 5: one launch boundary less per layer where every launch
 *                       is latency-bound) | 0: q / k / v post-processing launch + attention launch
 *   "moe_route_fused"   (0..1, 1): Next-DiT-MoE with both MoE layers per block (models2.py): the row kernel between the time and the space layer
 *                       (combine + gated residual + pre-norm) also routes the space layer - it has the router's input row in registers:
 *                       bf16 logits, top-2, fp32 softmax, bf16 weights, moe_route_kernel's arithmetic statement for statement, bit-identical
 *                       selections (round 5: one launch less per layer) | 0: a separate routing launch
 *   "gemm_splitk4"      (0..1, 1): with "gemm_splitk" on, a dense small-M plain-epilogue GEMM with K >= 4096 (the 512-row w2 projection) runs on
 *                       128 x 128 tiles with K split FOUR ways when that still fits one round of the CUs; the last arriver sums the four
 *                       fp32 partials in K order (round 5) | 0: the two-way split on 64 x 128 tiles
 *   "moe_time_plan_hoist" (0..1, 1): the time-routed MoE layers' plans (selection, weights, expert-sorted positions, tile table) of ALL layers
 *                       are written by one launch at the top of the evaluation - the time router's logits depend on the timestep embedding
 *                       only and are already computed there for every layer (round 5: one launch less per layer); not used while routing is
 *                       forced by the parity hook or with more than 64 samples | 0: one plan launch per layer
 *   "grn_ystat"         (0..2, 1): the O and W2 projections (persistent kernel, plain dense tiles) leave the per-row sum of squares of their
 *                       outputs behind, in partial sums per column tile and wave half, and the sandwich-norm row kernel that follows runs
 *                       its streaming form on it (64 registers, all rows resident; round 6) | 2: the same with two rows per wave (the second row's loads
 *                       in flight under the first row's arithmetic and stores) | 0: the row kernel loads the row, reduces, applies
 *   "qk_wg_per_cu"      (1..8, 3): persistent 4-wave workgroups per CU of the stand-alone q / k post-processing launch (LayerNorm + RoPE + head-major
 *                       view; at cfg 2 the K rows of a layer: 8192 rows over 3 x 256 x 4 waves = 2.67 rows per wave)
 *   "prologue_fused"    (0..7, 0): bit mask - 1: the timestep features are formed inside the first t_embedder GEMV, 2: temb + caption / label
 *                       embedding inside the adaLN GEMV, 4: the per-NFE gate / scale preparation as that GEMV's epilogue; each removes one launch
 *                       and is bit-identical to the separate kernel, and each measured SLOWER than the launch it removes (round 6,
 *                       profiles/r06/bench_ab_prologue_merges_cfg1_cfg5_all_lose.log) | 0 (default): the separate kernels
 *   "gemm_tail_split"   (0..2, 0): 1: the experts' W2 GEMM on the grouped persistent kernel cuts the tiles of a partial last round of its walk
 *                       along K into 2 / 4 parts (the MoE at 1024^2: 1.5-1.6 rounds of 256 x 256 tiles run as 2); the last part of a tile to
 *                       arrive sums the fp32 parts in K order | 2: two-way splits only | 0 (default): whole tiles only.  Built for VERDICT r5
 *                       item 5b and measured SLOWER (cfg5-1024: 24.73 -> 26.8 ms four-way, 24.95 two-way): a 256 x 256 fp32 part is 256 KB
 *                       through a CU that moves ~50-100 GB/s - the hand-off costs what the split saves (profiles/r06)
 *   "attn_text_skip"    (0..1, 1): the one-wave attention kernels (head_dim 72 / 96) do not run text tiles behind a sample's last valid text
 *                       key (the unconditional half of a CFG pair: 8 valid keys of 128 -> one 64-key tile instead of two); the skipped keys
 *                       are all masked, so results are bit-identical (round 6) | 0: every text tile up to Tk
 *   "attn_tail_split"   (0..4, 4): head_dim 96 one-wave kernel, sequences with a partial last query block of 64 or 128 rows (Flag-DiT at
 *                       1024^2: 4160 tokens = 16 x 256 + 64, the 64 workgroups of a fifth round with one live wave each): that block runs as
 *                       this many workgroups over disjoint key ranges ((O^T, m, l) partials in fp32) and a merge launch writes its rows
 *                       (round 6) | 0, 1: one workgroup per block as before
 *   "pair_layout"       (0..1, 1): evaluations whose dense blocks run all four GEMMs on the persistent kernel (the fused QKV launch, O, W1 | W3,
 *                       W2: >= one 256-row tile per CU) and the attention on a one-wave kernel keep the GEMMs' A operands (pre-norm output,
 *                       attention output, SwiGLU output) and the four weights of every layer in the row-pair-interleaved layout (lumina_dit.h,
 *                       lt_op_pair_layout): the LDS-DMA stream then fetches whole 128-byte lines - half the requests into the L2, the
 *                       counter that separated this kernel from the vendor's (round 6).  The weights are converted in place when the
 *                       regime changes (a 256-token call after a 4096-token one) and before every lt_set_weight; bit-identical results
 *                       | 0: row-major everywhere
 * (the round-1 names gemm_pipeline / gemm_pp_tail / gemm_persist are accepted with value 0 only: the study kernels they selected were
 *  deleted with csrc/experimental/ in round 5) */

/* diagnostics: the hd-72 self-attention kernels built with clock stamps.  Variant 3 (ping-pong): trace_dev receives, for every 64th
 * workgroup and each of its 8 waves, 8 x uint64: cycle totals of {X phase (MFMA), DMA wait, barrier, Y phase (softmax + DMA issue),
 * barrier}, the tile count.  Variant 4 (one wave per SIMD): per workgroup 8 x uint64 = s_memrealtime (100 MHz) at entry | loop start |
 * loop end | exit, shader clocks of the loop.  Readers: scripts/attn_trace.py, scripts/attn_trace_v4.py. */
int lt_op_attention_trace(const void* q_dev, const void* k_dev, const void* vt_dev, void* out_dev, int32_t B,
                          int32_t H, int32_t Hkv, int32_t N, int32_t Nk, int32_t Nkpad, int32_t hd, float scale,
                          void* trace_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif
