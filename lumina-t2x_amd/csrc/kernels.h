// Internal launcher interface between the HIP kernel files and the engine / C-ABI layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.h"

struct GemmArgs {
    const u16* A;      // [M, lda] bf16
    const u16* W;      // [N, ldw] bf16 (K contiguous)
    u16* C;            // [M, ldc] bf16 (ldc >= N, or N/2 for the SwiGLU epilogue)
    const void* bias;  // [N] or null
    int M, N, K;
    int lda, ldw, ldc;
    int bias_dtype;    // -1 none, 0 f32, 1 bf16
    unsigned long long* trace = nullptr;  // diagnostics only (lt_op_gemm_trace): per-wave cycle totals
    // grouped (MoE expert) mode: M-tile tm (256 rows) multiplies with W + tile_expert[tm] * w_expert_stride elements;
    // tile_expert[tm] < 0 -> the tile is padding and the workgroup exits.  Device array of ceil(M / 256) ints.
    const int* tile_expert = nullptr;
    long long w_expert_stride = 0;
    // gather-on-load (grouped SwiGLU GEMM of the MoE layers, gemm_bf16_pp kernels only): row m of the problem is row a_row_map[m] of A
    // (-1: a padding row, reads as zero); A then has a_map_rows rows.  Replaces a gather pass + an expert-sorted copy of the FFN input.
    const int* a_row_map = nullptr;
    int a_map_rows = 0;
    // epilogue 2 (V^T): C is the attention kernels' transposed, key-permuted V image [M / vt_tokens][N / vt_hd][vt_hd][vt_npad]
    // (AttnArgs::vt) instead of a row-major matrix: the V projection lands in the layout the PV MFMA reads, no transpose pass
    int vt_tokens = 0, vt_hd = 0, vt_npad = 0;
    // epilogue 3 (fused QKV projection, persistent 16x16x32 kernel): columns < vt_split go to C as a plain GEMM, columns >= vt_split are
    // the V projection and land in VT as the transposed image (same layout as epilogue 2).  vt_split and N - vt_split are multiples of
    // the 288-column tile, vt_tokens of the 256-row tile.
    u16* VT = nullptr;
    int vt_split = 0;
    // split-K (round 4, the 64 x 128 small-M tiles of gemm_bf16_pp only).  The CALLER lends a workspace (per engine = per stream:
    // launches that share one must be ordered); launch_gemm_bf16 decides (split_k is its output): the K range is cut in two, both
    // halves of a tile run as separate workgroups, each writes its fp32 partial tile to `splitk_part` and bumps the tile's counter;
    // the second arriver adds the other half to its accumulators, resets the counter and runs the normal epilogue.  fp32 addition
    // commutes, so the result does not depend on which half arrives last.
    float* splitk_part = nullptr;    // [splitk_tiles][2][64 * 128] fp32, or null: never split
    unsigned* splitk_cnt = nullptr;  // [splitk_tiles], zero between launches
    int splitk_tiles = 0;
    int split_k = 0;                 // set by the launcher: 0 off, 2 two halves (64 x 128 tiles), 4 four quarters (128 x 128 tiles, round 5)
    // epilogue 3, round 4: LayerNorm statistics of the Q columns on the way out.  Every plain tile whose first column lies below
    // qstat_cols (= the Q width) adds, per row and per wave column-half, (sum, sum of squares) of its bf16-ROUNDED outputs to
    // qstat[row][slot] with slot = 2 * (n0 / BN) + wn  (float2; qstat_slots = 2 * qstat_cols / BN per row).  The attention kernel's
    // prologue reduces the slots to the row's mean / rstd and applies q_norm + RoPE itself (AttnArgs::q_raw): Q is never re-written.
    // round 6, persistent dense kernel only (gemm_runs_w4q_dense): A and W in the ROW-PAIR-INTERLEAVED layout - element (r, k) of a
    // [rows][ld] matrix at (r >> 1) * 2 * ld + (k >> 5) * 64 + (r & 1) * 32 + (k & 31): the 64-byte pieces a 32-deep slab takes from rows 2 i and
    // 2 i + 1 are one 128-byte line (launch_pair_layout converts in place; rows even, K % 32 == 0).  pair_c: epilogue 1 writes its output
    // (ldc = N / 2 columns) in that layout.  The products and their order do not depend on it: results are bit-identical.
    // pair_ab: bit 0 = A, bit 1 = W (3 = both: the dense block; the grouped expert GEMMs: W, and A where it is not gathered through a_row_map)
    int pair_ab = 0, pair_c = 0;
    float* qstat = nullptr;
    int qstat_cols = 0, qstat_slots = 0;
    // round 5 (small-M tiles of gemm_bf16_pp, plain epilogue): rowstat[row][column tile of 128] = (sum, sum of squares) of that tile's
    // bf16-rounded outputs of the row (float2; rowstat_slots = ceil(N / 128) per row) - the LayerNorm partials the fused small-N
    // attention kernel (AttnSmallArgs) reduces; the launcher refuses it on any other kernel
    float* rowstat = nullptr;
    int rowstat_slots = 0;
    // round 6 (the persistent 16x16x32 kernel's plain dense instantiations only; gemm_ystat_slots() says whether a launch takes one):
    // ystat[row][slot] = sum of squares of the bf16-ROUNDED outputs of the row in column tile n0 / BN, wave column-half wn
    // (slot = 2 * (n0 / BN) + wn; ystat_slots = 2 * ceil(N / BN) floats per row) - the RMSNorm statistic of the row kernel that
    // consumes C (GatedResArgs::ystat), which can then start on a row's first bytes instead of after its last
    float* ystat = nullptr;
    int ystat_slots = 0;
    // round 6, the grouped persistent kernel's plain epilogue (the MoE experts' W2): the tiles of a partial LAST round of the persistent walk
    // are cut along K into 2 / 4 parts (picked on the device from the number of valid tiles), parts hand fp32 accumulators through
    // tail_part ([tail tile][part][256 x 256] floats, tail_cap_parts parts in all) and count in on tail_cnt ([tail tile], zero between launches)
    float* tail_part = nullptr;
    unsigned* tail_cnt = nullptr;
    long long tail_cap_parts = 0;
    int tail_max_parts = 4;  // 2: two-way splits only
    int group_rows = 0;  // experiment knob (lt_set_option "gemm_group"): tile rows per group of the XCD-aware tile order (0 = 4)
    int stagger = 0;  // experiment knob of the 4-wave kernels (lt_set_option "gemm_stagger"), filled by the launcher
};
// ev0 / ev1: optional start / stop events carried by the dispatch packet itself (profiling without extra queue packets)
int launch_gemm_bf16(const GemmArgs& a, int epilogue, int variant, hipStream_t stream, hipEvent_t ev0 = nullptr,
                     hipEvent_t ev1 = nullptr);
// weight-panel prefetch of a small-M GEMM as a RIDER in the grid of the kernel that precedes it (option "gemm_prefetch" 3): blocks
// [first, first + blocks) of the host kernel run prefetch_w_block (tile_order.h) instead of the host's work.  first % 8 == 0, so that
// rider i runs on the XCD GEMM workgroup i will run on.  Filled by gemm_prefetch_rider (false: the GEMM is not a small-M launch).
struct PrefetchRider {
    const u16* W = nullptr;
    int N = 0, K = 0, ldw = 0, BN = 0, TM = 0, TN = 0, split = 0;
    int first = 0x7fffffff, blocks = 0;
};
bool gemm_prefetch_rider(const GemmArgs& a, int epilogue, PrefetchRider* r);
int gemm_ystat_slots(const GemmArgs& a, int epilogue);  // > 0: launch_gemm_bf16 would run this problem on a kernel that can fill GemmArgs::ystat, with this many slots per row
bool gemm_is_small_m(const GemmArgs& a, int epilogue);  // launch_gemm_bf16 would run this problem on the 128 x 128 / 64 x 128 small-M tiles (GemmArgs::rowstat needs them)
int launch_gemm_prefetch_w(const GemmArgs& a, int epilogue, hipStream_t stream);  // experiment: W panels of a small-M GEMM -> the L2 of the XCDs that will stage them
bool gemm_qkv_fusable(const GemmArgs& a);
int gemm_qkv_tile_width(const GemmArgs& a);  // 288 / 256 (the fused launch's tile width), 0 = not fusable  // epilogue 3 can take this problem (else: one plain launch for Q | K + one V^T launch)
int launch_pack_w13(const u16* w1, const u16* w3, u16* out, int F, int K, hipStream_t stream);

// ---- norm / residual kernels (norm.hip) --------------------------------------------------------
struct NormModArgs {
    const u16* x;      // [rows, d]
    const u16* w;      // [d] or null
    const u16* scale;  // [B, ld_mod] or null
    const u16* shift;  // [B, ld_mod] or null
    u16* out;          // [rows, d]
    int rows, rows_per_batch, d, ld_mod;
    float eps;
    int scale_pre = 0;  // 1: `scale` already holds bf16(1 + scale) (engine: prepared once per NFE by launch_prep_mod)
    int apex = 0;       // set by the launcher from option rmsnorm_apex: the weight multiplies in fp32 before the one rounding
    int out_pair = 0;   // `out` in the row-pair-interleaved layout (GemmArgs::pair_ab: it is a persistent GEMM's A operand); rows even, d % 32 == 0
};
int launch_rmsnorm_mod(const NormModArgs& a, hipStream_t stream);

struct GatedResArgs {
    u16* x;                 // residual stream [rows, d], updated in place
    const u16* y;           // branch output [rows, d]
    const u16* post_w;      // RMSNorm weight applied to y (post_mode 1)
    const u16* gate;        // [B, ld_mod] or null
    const u16* next_w;      // weight of the next pre-norm or null
    const u16* next_scale;  // [B, ld_mod] or null
    const u16* next_shift;  // [B, ld_mod] or null
    u16* h;                 // [rows, d] next branch input (next_mode != 0)
    int rows, rows_per_batch, d, ld_mod;
    int post_mode, gate_mode, next_mode;
    float eps, eps_next;
    int scale_pre = 0;  // 1: next_scale already holds bf16(1 + scale)
    int apex = 0;       // set by the launcher from option rmsnorm_apex (generic kernel only; the specialised instantiations are bypassed)
    // MoE layers: y is the top-2 combine of the experts' outputs, formed on the way in (y may then be null):
    // ys [sorted rows, d], pos [rows, 2] sorted row of each (token, expert) pair, wts [rows, 2] bf16 routing weights (MoeArgs)
    const u16* moe_ys = nullptr;
    const int* moe_pos = nullptr;
    const u16* moe_wts = nullptr;
    PrefetchRider pf;  // weight panels of the GEMM that follows, read by extra workgroups of this launch (512-row-class problems)
    // round 5 (MoE kernels, next_mode 1): h is the input of a token-routed MoE layer - the row kernel has the row in registers and routes it on
    // its way out: logits = bf16(h . route_w[e]) (nn.Linear under autocast), top-2, fp32 softmax, bf16 weights -> route_sel / route_wts
    // ([rows][2], MoeArgs::sel / wts), exactly moe_route_kernel's arithmetic; route_forced: the parity hook's [rows][2] expert ids or null
    const u16* route_w = nullptr;  // [route_E, d] bf16
    int route_E = 0;
    int* route_sel = nullptr;
    u16* route_wts = nullptr;
    const int* route_forced = nullptr;
    // round 6: the sum of squares of every row of y, in ystat_slots partial sums per row, left behind by the GEMM that wrote y
    // (GemmArgs::ystat).  With it the first RMSNorm needs nothing of the row but the element at hand: the dense post_mode 1 / gate_mode 0 /
    // next_mode 1 combination then runs a streaming kernel (8-byte steps, 64 registers, eight waves per SIMD: all rows of a 8192-row
    // launch resident at once) instead of "load the row, reduce, apply".  Ignored (the kernel reduces y itself) by every other combination.
    const float* ystat = nullptr;
    int ystat_slots = 0;
    int h_pair = 0;  // next_mode 1: `h` in the row-pair-interleaved layout (GemmArgs::pair_ab); rows even, d % 32 == 0
};
int launch_gated_residual_norm(const GatedResArgs& a, hipStream_t stream);

// ---- q/k/v post-processing (qkv_post.hip) ------------------------------------------------------
struct QkPostArgs {
    const u16* src;      // [B*N, ld_src]
    const u16* ln_w;     // [heads*hd] or null (no qk-norm)
    const u16* ln_b;
    u16* dst;            // [B, heads, N, hd]
    const float* cs;     // (cos,sin) table, see lt_op_qk_norm_rope
    const float* t;      // device timesteps (branch select) or null
    int ld_src, col0, B, N, heads, hd;
    int rope_mode, grid_w, cs_len;  // cs_len: positions per branch table
    float ln_eps, watershed;
    float out_scale = 1.0f;  // multiplies the result before its single bf16 rounding (folds softmax scale * log2 e into K)
    // packed variable-resolution batches (model.py:803-831): per-sample token count and latent-grid width (device arrays or
    // null); padded positions n >= n_tok_b[b] rotate like the sample's LAST token (item_freqs_cis[-1:] expand, :822-827)
    const int* n_tok_b = nullptr;
    const int* grid_w_b = nullptr;
    // round 4, the K pass of a layer whose queries are post-processed by the attention prologue (AttnArgs::q_raw): while it walks
    // row r it also reduces the fused QKV GEMM's LayerNorm partials of the SAME row of Q (GemmArgs::qstat: qstat_slots float2 of
    // (sum, sum of squares)) to the row's (mean, rstd) -> qstat_out[r] (float2), once per row instead of once per head and lane
    const float* qstat_in = nullptr;
    float* qstat_out = nullptr;
    int qstat_slots = 0, qstat_width = 0;
};
int launch_qk_norm_rope(const QkPostArgs& a, hipStream_t stream);
int launch_qk_norm_rope_pair(const QkPostArgs& q, const QkPostArgs& k, hipStream_t stream);  // both in one persistent launch
int launch_v_transpose(const u16* src, int ld_src, int col0, u16* dst, int B, int N, int Npad, int kv_heads,
                       int hd, hipStream_t stream);
// the three passes above in one launch (engine path)
struct QkvPostArgs {
    QkPostArgs q, k;
    const u16* v_src;
    u16* v_dst;
    int v_ld_src, v_col0, v_B, v_N, v_Npad, v_kv_heads, v_hd;
    int nq_blocks = 0, nk_blocks = 0;  // filled by the launcher
    PrefetchRider pf;  // weight panels of a later small-M GEMM (the O projection), read by extra workgroups of this launch
};
int launch_qkv_post(const QkvPostArgs& a, hipStream_t stream);

// ---- attention (attention.hip) -----------------------------------------------------------------
struct AttnArgs {
    const u16* q;       // [B, H, N, hd]
    const u16* k;       // [B, Hkv, Nk_rows, hd]
    const u16* vt;      // [B, Hkv, hd, Nkpad]
    const float* bias;  // [B, Nkpad] or null
    u16* out;           // [B, N, H*hd]
    const u16* gate;    // [H] bf16 (accumulate mode)
    int accumulate;
    int B, H, Hkv, N, Nk, Nkpad, hd;
    float scale;
    int k_prescaled = 0;  // 1: K already carries scale * log2(e) (qk_norm_rope out_scale): scores are in the log2 domain
    // fused gated text cross-attention (model.py:420-434), hd-72 ping-pong kernel only: after the self-attention loop the
    // same workgroup attends its (still resident) Q rows to the text keys and writes
    //   out = bf16(self) + bf16(bf16(text) * tanh(gate[h])).  Requires k_prescaled (both K carry their scale * log2 e).
    const u16* tk = nullptr;      // [B, Hkv, Tk, hd]
    const u16* tvt = nullptr;     // [B, Hkv, hd, Tkpad]
    const float* tbias = nullptr; // [B, Tkpad] 0 / -inf (padded with -inf)
    const u16* tgate = nullptr;   // [H] bf16
    int Tk = 0, Tkpad = 0;
    // one-wave kernels (hd 72 / 96), set by their launchers from option attn_text_skip: text tiles behind a sample's last valid key are
    // not run (all their keys are masked: bit-identical results)
    int text_skip = 0;
    // hd-96 one-wave kernel, set by its launcher (option attn_tail_split; attention_v4_96.hip): the partial last query block of every head
    // (N % 256 = 64 or 128 rows: one or two live waves of four) is run as tail_split workgroups over disjoint key ranges that leave
    // (O^T, m, l) partials in tail_ws; launch_attention_v4_hd96 follows with the merge kernel.  tail_ws == nullptr at the op-level
    // entries: the launcher's own per-device workspace.
    int tail_split = 0, tail_rows = 0;
    float* tail_ws = nullptr;
    size_t tail_ws_bytes = 0;
    int out_pair = 0;  // one-wave kernels: `out` ([B * N][H * hd]) in the row-pair-interleaved layout (the O projection's A operand, GemmArgs::pair_ab)
    unsigned long long* trace = nullptr;  // diagnostics only (lt_op_attention_trace)
    // round 4 (attn_fwd_kernel_v4<72> only; attention_takes_raw_q() says whether a call qualifies): q == nullptr and the workgroup
    // makes its 256 query rows itself from the QKV projection's row-major output - q_norm (full-width affine LayerNorm in fp32; the
    // row's (mean, rstd) come from q_stat, which the K pass of qk_norm_rope reduced from the fused QKV GEMM's partial sums,
    // GemmArgs::qstat / QkPostArgs::qstat_in) -> 2-D RoPE in fp32 -> ONE bf16 rounding (model.py:361-371), the arithmetic of
    // qk_norm_rope.  Saves the Q half of that kernel's round trip (2 x [M, d] bf16 per layer).
    const u16* q_raw = nullptr;     // [B * N, q_ld] bf16, this head's columns at q_col0 + h * hd
    int q_ld = 0, q_col0 = 0;
    const float* q_stat = nullptr;  // [B * N] float2 (mean, rstd)
    const u16* q_ln_w = nullptr;    // [H * hd] bf16
    const u16* q_ln_b = nullptr;
    const float* rope_cs = nullptr;   // (cos, sin) table of QkPostArgs::cs (rope_mode 1): [branch][pos][hd / 4], the ROW factors (one position per 32 lanes)
    const float* rope_cs_t = nullptr; // the same table as [branch][hd / 4][pos]: the COLUMN factors (32 consecutive positions per load)
    const float* rope_t = nullptr;    // branch = rope_t[0] < rope_watershed ? 0 : 1
    float rope_watershed = 0.f;
    int rope_cs_len = 0, rope_grid_w = 0;
    // packed variable-resolution batches (model.py:789-834): valid keys of sample b = nk_batch[b] <= Nk (device array, or null)
    const int* nk_batch = nullptr;
    // regional (compositional) text attention: key/value/bias/output batch b attends the queries of batch q_batch_map[b]
    // (several captions share one image's queries); device array [B] or null.  Masked (bias != null) kernels only.
    const int* q_batch_map = nullptr;
};
// out[b] = bf16(out[b] + bf16(sum_r region_r(n) * bf16(bf16(txt[r]) * tanh(gate[h]))))  - the caption sum of the compositional
// Next-DiT (lumina_next_compositional_generation/models/model.py:422-446); see misc.hip
int launch_region_text_combine(u16* out, const u16* txt, const u16* gate, int Y, int N, int H, int hd, int Hp, int Wp,
                               int h_split, int w_split, hipStream_t stream);
int launch_attention(const AttnArgs& a, hipStream_t stream);
// Fused q / k post-processing + V^T staging + attention for short sequences (attention_small.hip, round 5): one launch instead of
// qkv_post + attention at the 600M class-conditional models' 256 tokens.  Reads the QKV projection's row-major output and the per-tile
// LayerNorm partials its GEMM left (GemmArgs::rowstat).
struct AttnSmallArgs {
    const u16* qkv;        // [B * N, ld] bf16: q | k | v column blocks as the GEMM wrote them
    int ld, q_col0, k_col0, v_col0;
    const float* rowstat;  // [B * N][slots] float2 (sum, sum of squares) per 128-column tile of the projection
    int slots, q_slot0, q_nslot, k_slot0, k_nslot;
    const u16 *q_ln_w, *q_ln_b, *k_ln_w, *k_ln_b;  // affine LayerNorm over the full q / k width (bf16)
    float ln_eps;
    const float* cs;       // (cos, sin) table of QkPostArgs::cs, 2-D mode: [branch][pos][hd / 4]
    const float* t;        // device timesteps (branch = t[0] < watershed ? 0 : 1) or null (branch 1)
    float watershed;
    int cs_len, grid_w;
    float k_scale;         // softmax scale * log2(e), folded into K's one bf16 rounding
    u16* out;              // [B, N, H * hd]
    int B, H, Hkv, N, hd;
    PrefetchRider pf;      // weight panels of the GEMM that follows (the O projection), read by extra workgroups of this launch
};
bool attention_small_fusable(int hd, int N, int H, int Hkv, int q_width, int k_width);
int launch_attention_small(const AttnSmallArgs& a, hipStream_t stream);
int launch_attention_v4(const AttnArgs& a, hipStream_t stream);  // hd 72, 4 waves x 64 query rows (attention_v4.hip)
int launch_attention_v4_hd48(const AttnArgs& a, hipStream_t stream);  // hd 48, the same structure, softmax-bound (attention_v4_48.hip)
int launch_attention_v4_hd96(const AttnArgs& a, hipStream_t stream);  // hd 96, the same structure without pad slots (attention_v4_96.hip)
size_t attention_tail_ws_floats(int heads, int parts, int rows);  // AttnArgs::tail_ws of launch_attention_v4_hd96's tail split, in floats
bool attention_is_one_wave(const AttnArgs& a);  // launch_attention would run this call on a one-wave-per-SIMD kernel (hd 72 / 48 / 96): these write AttnArgs::out_pair
bool attention_takes_raw_q(const AttnArgs& a);  // launch_attention would run this call on attn_fwd_kernel_v4<72> (the kernel with the q_raw prologue)
bool attention_fuses_text(int hd);  // hd-72 ping-pong kernel: text cross-attention rides in the self-attention launch
int ensure_dynamic_lds(const void* fn, int bytes);  // hipFuncSetAttribute(MaxDynamicSharedMemorySize) per (device, kernel), raised whenever a larger size is asked for; records the pair only on success
bool gemm_runs_w4q_grouped(const GemmArgs& a, int epilogue);  // ... on the persistent kernel's grouped (expert) mode
bool gemm_runs_w4q_dense(const GemmArgs& a, int epilogue);  // launch_gemm_bf16 would run this (variant 0) call on the persistent dense kernel, the one that reads the pair layout
const char* lt_gemm_describe(const GemmArgs& a, int epilogue, int variant);  // name of the kernel launch_gemm_bf16 would run

// ---- mixture-of-experts routing (moe.hip; Next-DiT-MoE/models/models2.py:451-506) --------------------------------
struct MoeArgs {
    const u16* x;              // [rows, d] FFN input (after pre-norm + modulate)
    const u16* gate_w;         // space router weight [E, d] (per-token logits) or null
    const u16* sample_logits;  // time router logits [B, E] bf16 (every token of a sample shares them) or null; with them the plan
                               // kernel routes as well (no separate route launch)
    int sample_ld = 0;         // row stride of sample_logits in elements (0 = E): the engine computes every layer's time-router logits in ONE
                               // GEMV per evaluation ([B, L * E]) and hands each layer its E columns
    const int* forced;         // [rows, 2] expert ids that replace the top-2 choice (parity hook, lt_moe_routing_force) or null
    int rows, rows_per_sample, d, E;
    int* sel;                  // [rows, 2] selected experts, ascending expert id (= the reference's accumulation order)
    u16* wts;                  // [rows, 2] bf16 softmax weights aligned with sel
    int* pos;                  // [rows, 2] row of each (token, expert) pair in the expert-sorted buffers
    int* src;                  // [max_tiles * 256] inverse map: token row of each sorted position, -1 for padding (GemmArgs::a_row_map)
    int* tile_expert;          // [max_tiles] expert of each 256-row tile of the sorted buffers, -1 = padding
    int max_tiles;
    // round 5: the time router's plans of ALL layers in one launch (the logits of every layer exist before the first block runs):
    // layer l reads sample_logits + l * E and writes sel / pos at + l * layer_stride_rows, wts likewise (u16), src at + l * layer_stride_src,
    // tile_expert at + l * layer_stride_tiles.  layers <= 1 = one plan (the fields above as they are)
    int layers = 1;
    long long layer_stride_rows = 0, layer_stride_src = 0, layer_stride_tiles = 0;
};
constexpr int LT_MOE_PLAN_TIME_MAX_SAMPLES = 64;               // the closed-form time plan holds this many samples' routings in LDS
int launch_moe_route(const MoeArgs& a, hipStream_t stream);    // logits -> top-2, weights
int launch_moe_plan(const MoeArgs& a, hipStream_t stream);     // counts -> tile-aligned segments, pos, src, tile_expert

// ---- small kernels (misc.hip) ------------------------------------------------------------------
int launch_linear_small_m(const u16* a, const u16* w, const u16* b, u16* y, int M, int N, int K, int act_in,
                          hipStream_t stream);
// round 6 (option prologue_fused): the three element-wise launches around the conditioning GEMVs folded into them,
// each with the stand-alone kernel's own statements so that the results are bit-identical:
//   t        the input rows are the sinusoidal timestep features of t[m] (launch_timestep_features: K = feature dim; `a` is not read)
//   a2       a := bf16(a + a2) before the activation (launch_add_bf16: temb + caption / label embedding)
//   pm_*     launch_prep_mod's transform applied to the rounded output: column n lies in chunk (n / d) % chunks of layer n / (chunks d)
//            (tanh / one-plus by the masks) or, past the layers, in the final layer's vector (chunk pm_final -> one-plus)
struct LinearSmallMExtra {
    const float* t = nullptr;
    const u16* a2 = nullptr;
    int pm_L = 0, pm_chunks = 0, pm_d = 0, pm_final = -1;
    unsigned pm_tanh = 0, pm_scale = 0;
};
int launch_linear_small_m_ext(const u16* a, const u16* w, const u16* b, u16* y, int M, int N, int K, int act_in,
                              const LinearSmallMExtra& x, hipStream_t stream);
// weights upload: cast src (f32/bf16/f16) to bf16 rows at dst (+ row offset handled by caller)
int launch_cast_to_bf16(const void* src, int dtype, u16* dst, long long n, hipStream_t stream);
// x [B,C,H,W] (bf16/f32) -> patch rows [B*N, kpad] bf16, (c,ph,pw) order, zero padded (model.py:777)
// wp_stride: tokens per latent row in `out` (0 = W / patch; W / patch + 1 leaves one eol slot per row untouched)
int launch_patchify(const void* x, int x_dtype, u16* out, int B, int C, int H, int W, int patch, int kpad,
                    int dup_first_half, int wp_stride, hipStream_t stream);
// Flag-DiT: the last token of each latent row is the learned eol_token (lumina_t2i/models/model.py:779-786)
int launch_eol_fill(u16* x, const u16* eol, int rows_total, int Wp, int d, hipStream_t stream);
// class-conditional variants: out[b] = table[labels[b]] (bf16 [rows, d] table)
int launch_label_gather(const u16* table, const int32_t* labels, u16* out, int B, int rows, int d, hipStream_t stream);
// sinusoidal timestep features (model.py:63-82): t [B] f32 -> [B, dim] bf16 (cos block then sin block)
int launch_timestep_features(const float* t, int t_index, u16* out, int B, int dim, hipStream_t stream);
// masked mean pool + affine LayerNorm (model.py:847-849, cap_embedder.0): -> [B, C] bf16
int launch_cap_pool_ln(const void* cap, int cap_dtype, const int32_t* mask, const u16* ln_w, const u16* ln_b,
                       u16* out, int B, int T, int C, hipStream_t stream);
// in place, once per NFE, on every layer's adaLN vector (mod [B, ld_mod], `chunks` chunks of d per layer + final_chunks for
// the final layer): chunks whose bit is set in tanh_mask -> bf16(tanh(.)) (gates), in scale_mask -> bf16(1 + .) (scales);
// final_scale_chunk >= 0: that chunk of the final layer's vector -> bf16(1 + .).  Per (sample, channel) instead of per token.
int launch_prep_mod(u16* mod, int B, int ld_mod, int L, int chunks, int d, unsigned tanh_mask, unsigned scale_mask,
                    int final_scale_chunk, hipStream_t stream);
// c = bfr(a + b) elementwise bf16
int launch_add_bf16(const u16* a, const u16* b, u16* c, long long n, hipStream_t stream);
// cap_feats (any dtype) -> bf16 copy, and mask -> additive float bias (0 / -inf), padded to Tpad
int launch_mask_to_bias(const int32_t* mask, float* bias, int B, int T, int Tpad, hipStream_t stream);
// final projection rows [B*N, ld] bf16 -> unpatchify (model.py:749-755), keep first C channels,
// optional CFG combine on cfg_channels (model.py:908-913); out [B,C,H,W] bf16/f32
int launch_unpatchify_cfg(const u16* rows, int ld, void* out, int out_dtype, int B, int C, int out_ch, int H, int W,
                          int patch, int use_cfg, float cfg_scale, int cfg_channels, int wp_stride, hipStream_t stream);
// torchdiffeq fixed-grid state arithmetic (modes documented in misc.hip)
int launch_ode_combine(int mode, const void* y0, const void* k1, const void* k2, const void* k3, const void* k4,
                       void* out, int dtype, float dt, long long n, hipStream_t stream);
// weight upload: cast rows of src [rows, cols] to bf16 at dst rows (row_map 0: r0 + r; 1/2: w1/w3 slots of
// the 32-row interleaved SwiGLU layout), row stride dst_ld (>= cols; padding left untouched)
int launch_upload_rows(const void* src, int dtype, u16* dst, int rows, int cols, int dst_ld, int r0, int row_map,
                       hipStream_t stream);
int launch_rope_table_2d(float* out, int len, int hd, float theta, float scale_factor, hipStream_t stream, float* out_t = nullptr);
// general (cos,sin) factor table, two branches: out[b][pos][fi], fi < hd / step (step 4 = 2-D RoPE axis table, 2 = 1-D)
int launch_rope_table(float* out, int len, int hd, int step, float theta0, float lin0, float theta1, float lin1,
                      int lin_on_pos, hipStream_t stream, float* out_t = nullptr);  // out_t: the same as [b][fi][pos]
int launch_fill_rows_bf16(u16* dst, const u16* row, long long rows, int d, hipStream_t stream);
// in-place conversion of a dense [rows][cols] bf16 matrix to (to_pair 1) / from (0) the row-pair-interleaved layout of GemmArgs::pair_ab
int launch_pair_layout(u16* m, long long rows, int cols, int to_pair, hipStream_t stream);
