/*
 * lumina_dit.h -- C ABI of the MI355X (gfx950) Next-DiT / Flag-DiT denoising engine.
 *
 * The reference (Alpha-VLLM/Lumina-T2X) has NO plugin / FFI interface: the hot path sits behind two
 * Python conventions:
 *   1. the model-callable protocol  model_fn(x[B,C,H,W], t[B], **kw) -> [B,C,H,W]
 *      (lumina_next_t2i/transport/transport.py:192-195, integrators.py:104-116), concretely
 *      NextDiT.forward_with_cfg (lumina_next_t2i/models/model.py:866-913) and
 *      NextDiT.forward (model.py:836-864);
 *   2. module construction + state_dict key contract (lumina_next_t2i/sample.py:125-142).
 * This header is what a ctypes / cffi / pybind stub on the reference side would bind to replace the
 * body of those two methods and of transport/integrators.py:ode.sample (integrators.py:104-116).
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; lt_last_error() gives the message
 *     (thread-local, valid until the next call on the same thread).  No C++ exception crosses the ABI.
 *   - all pointers named *_dev are DEVICE pointers owned by the caller (PyTorch in our host code);
 *     the engine never frees or retains them beyond the call, except where stated.
 *   - `stream` is a hipStream_t passed as void* (0 = default stream).  No call synchronises the
 *     stream or the device; nothing inside the denoising loop reads device data on the host.
 *   - dtype codes: LT_F32 = 0, LT_BF16 = 1, LT_F16 = 2 (f16 only accepted for weights upload).
 */
#ifndef LUMINA_DIT_H
#define LUMINA_DIT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LT_F32 0
#define LT_BF16 1
#define LT_F16 2

/* model variants (which reference class the engine reproduces) */
#define LT_VARIANT_NEXT_T2I 0      /* lumina_next_t2i/models/model.py:665 NextDiT              */
#define LT_VARIANT_NEXT_IMAGENET 1 /* Next-DiT-ImageNet/models/models.py:836 DiT_Llama         */
#define LT_VARIANT_FLAG_T2I 2      /* lumina_t2i/models/model.py:661 DiT_Llama (Flag-DiT)      */
#define LT_VARIANT_NEXT_MOE 3      /* Next-DiT-MoE/models/models2.py:850 DiT_Llama (time+space MoE) */
#define LT_VARIANT_NEXT_MOE_TIME 4  /* Next-DiT-MoE/models/models.py:802 DiT_Llama: ONE MoE FFN per block, routed by the timestep
                                       embedding (every token of a sample visits the same two experts; models.py:459-477)  */
#define LT_VARIANT_NEXT_MOE_SPACE 5 /* Next-DiT-MoE/models/models1.py:802 DiT_Llama: ONE MoE FFN per block, routed per token   */

/* fixed-grid ODE methods, torchdiffeq names (lumina_next_t2i/transport/integrators.py:115) */
#define LT_ODE_EULER 0
#define LT_ODE_MIDPOINT 1
#define LT_ODE_RK4 2

typedef struct lt_engine lt_engine;

typedef struct lt_config {
    int32_t variant;       /* LT_VARIANT_*                                                        */
    int32_t dim;           /* model width d            (model.py:674)                             */
    int32_t n_layers;      /* L                        (model.py:675)                             */
    int32_t n_heads;       /* H                        (model.py:676)                             */
    int32_t n_kv_heads;    /* GQA kv heads, == n_heads for MHA (model.py:158)                     */
    int32_t ffn_hidden;    /* F, already rounded as in FeedForward.__init__ (model.py:469-473)    */
    int32_t patch_size;    /* 2                                                                   */
    int32_t in_channels;   /* 4                                                                   */
    int32_t out_channels;  /* 8 when learn_sigma (model.py:689)                                   */
    int32_t cap_feat_dim;  /* text feature width (2048 Gemma-2B); 0 for class-conditional         */
    int32_t adaln_dim;     /* min(dim, 1024)           (model.py:563)                             */
    int32_t qk_norm;       /* 1: affine LayerNorm over the full q/k projection (model.py:211-215) */
    int32_t num_classes;   /* class-conditional variants only (label table has num_classes+1 rows)*/
    float   norm_eps;      /* RMSNorm eps, 1e-5        (model.py:680)                             */
    int32_t max_batch;     /* max rows of the (cond+uncond) batch the workspace is sized for      */
    int32_t max_tokens;    /* max latent tokens per sample (N)                                    */
    int32_t max_text;      /* max text tokens per sample (T)                                      */
    int32_t rope_table_len;/* 384 (model.py:734); positions per axis in the 2-D RoPE table        */
    int32_t num_experts;   /* LT_VARIANT_NEXT_MOE*: experts per MoE layer (4 models2.py:696, 8 models.py / models1.py:666); top-2 */
} lt_config;

/* kwargs of NextDiT.forward_with_cfg (model.py:866-877) that are not tensors */
typedef struct lt_step_args {
    float   cfg_scale;         /* model.py:872                                                    */
    float   scale_factor;      /* model.py:873   RoPE extrapolation factor (rope_scaling_factor elsewhere) */
    float   scale_watershed;   /* model.py:874   t < watershed -> linear interp, else NTK          */
    int32_t base_seqlen;       /* model.py:875   0 = None                                          */
    int32_t proportional_attn; /* model.py:876                                                     */
    int32_t latent_h;          /* H of x[B,C,H,W]                                                  */
    int32_t latent_w;          /* W of x[B,C,H,W]                                                  */
    int32_t batch;             /* B (cond+uncond rows), even for forward_with_cfg                  */
    int32_t io_dtype;          /* LT_BF16 or LT_F32: dtype of x and out                            */
    int32_t cfg_channels;      /* 3 = reference quirk (model.py:908); in_channels = standard CFG   */
    float   ntk_factor;        /* ImageNet / Flag-DiT forward_with_cfg(ntk_factor=...) (models.py:946,  */
                               /* lumina_t2i model.py:866-875); there scale_factor = rope_scaling_factor; */
                               /* <= 0 means 1.0.  Ignored by LT_VARIANT_NEXT_T2I.                       */
} lt_step_args;

const char* lt_last_error(void);
/* library / build identification: "lumina_dit gfx950 r5" */
const char* lt_version(void);

/* ---- options ----------------------------------------------------------------------------------
 * The engine picks its kernels by problem shape; a handful of A/B and diagnostic options can override the choice.  They are NOT part
 * of the drop-in surface - an integrator never has to touch them - and are documented, one by one, in include/lumina_dit_debug.h.
 *   lt_set_option          sets the PROCESS DEFAULT of an option: what engines without an override of their own, and the lt_op_*
 *                          operator entry points, use
 *   lt_engine_set_option   overrides an option for ONE engine (every later call on that engine, from any thread; nothing else).
 *                          value LT_OPTION_INHERIT drops the override again.  Thread-safe against calls running on the same engine: every
 *                          engine entry point takes one snapshot of all options when it starts and finishes on that snapshot
 *   lt_engine_get_option   the value in effect for an engine (e == NULL: the process default)
 * Unknown names and out-of-range values are errors (lt_last_error names the range).  There is no other mutable state outside an
 * lt_engine. */
#define LT_OPTION_INHERIT INT32_MIN
int lt_set_option(const char* name, int32_t value);
int lt_engine_set_option(lt_engine* e, const char* name, int32_t value);
int lt_engine_get_option(lt_engine* e, const char* name, int32_t* value);

/* ---- engine lifetime ------------------------------------------------------------------------- */
int  lt_create(const lt_config* cfg, lt_engine** out);
void lt_destroy(lt_engine* e);

/* Upload one checkpoint tensor by its reference state_dict key, e.g.
 * "layers.3.attention.wq.weight".  The engine converts to bf16 and repacks into its own HBM arena
 * (QKV concatenated, w1/w3 interleaved in 32-row groups for the fused SwiGLU epilogue); the source
 * pointer is not retained.  Unknown keys are an error; lt_weights_ready() reports missing keys. */
int lt_set_weight(lt_engine* e, const char* key, const void* src_dev, int32_t dtype,
                  const int64_t* shape, int32_t ndim, void* stream);
int lt_weights_ready(lt_engine* e); /* 0 if every required tensor has been uploaded */

/* Step-invariant text work (model.py:421-422,602,847-850): masked mean pool + cap_embedder, and per
 * layer RMSNorm_y -> wk_y/wv_y -> ky_norm K/V.  cap_feats [B,T,cap_dim], cap_mask int32 [B,T]. */
int lt_prepare_prompt(lt_engine* e, const void* cap_feats_dev, int32_t cap_dtype,
                      const int32_t* cap_mask_dev, int32_t B, int32_t T, void* stream);
/* Compositional (regional) conditioning - lumina_next_compositional_generation/models/model.py:852-890, :422-446: Y captions
 * [Y,T,cap_dim] with masks; captions 0..Y-2 condition regions of the COND row (the latent grid cut into h_split x w_split cells,
 * cell (i,j) -> caption (i+1)(j+1)-1 as in the reference), caption Y-1 conditions the whole UNCOND row; the adaLN conditioning
 * is pooled from the one-row global caption [1,Tg,cap_dim].  Steps that follow run one image (batch 2).  A later
 * lt_prepare_prompt switches back to plain per-row captions.  Needs max_batch >= Y. */
int lt_prepare_prompt_regional(lt_engine* e, const void* cap_feats_dev, int32_t cap_dtype, const int32_t* cap_mask_dev,
                               int32_t Y, int32_t T, const void* global_feats_dev, const int32_t* global_mask_dev,
                               int32_t Tg, int32_t h_split, int32_t w_split, void* stream);
/* class-conditional variants: labels int32 [B] (null class = num_classes) */
int lt_prepare_labels(lt_engine* e, const int32_t* labels_dev, int32_t B, void* stream);

/* NextDiT.forward (model.py:836-864): x [B,C,H,W] -> out [B,C,H,W] (first in_channels kept). */
int lt_forward(lt_engine* e, const void* x_dev, const float* t_dev, void* out_dev,
               const lt_step_args* a, void* stream);
/* NextDiT.forward with a LIST of differently sized samples (patchify_and_embed list branch, model.py:789-834; unpatchify
 * :757-768): x_ptrs / out_ptrs are HOST arrays of a->batch device pointers, sample b is [in_channels, H_b, W_b] with
 * (H_b, W_b) = hw_host[2b], hw_host[2b+1] (a->latent_h / latent_w are ignored).  Sequences are padded to the longest one with
 * pad_token; padded keys are masked; each output has its own sample's shape (sigma half dropped).  Text-conditional
 * Next-DiT, plain forward only (the reference's forward_with_cfg takes tensors only). */
int lt_forward_packed(lt_engine* e, const void* const* x_ptrs, const int32_t* hw_host, const float* t_dev,
                      void* const* out_ptrs, const lt_step_args* a, void* stream);
/* NextDiT.forward_with_cfg (model.py:866-913): duplicates the first half, CFG on cfg_channels. */
int lt_forward_cfg(lt_engine* e, const void* x_dev, const float* t_dev, void* out_dev,
                   const lt_step_args* a, void* stream);

/* ode.sample (integrators.py:104-116) with torchdiffeq's fixed-grid euler / midpoint / rk4:
 * tgrid host pointer, n_grid points; z [B,C,H,W]; traj_dev (may be NULL) receives all n_grid
 * states [n_grid,B,C,H,W]; final_dev (may be NULL) receives the last state.  Every model call is
 * forward_with_cfg when use_cfg != 0, else forward.  t_round_to_state_dtype mirrors torchdiffeq's
 * _PerturbFunc cast of t to the state dtype. */
int lt_sample_ode(lt_engine* e, const void* z_dev, void* traj_dev, void* final_dev,
                  const float* tgrid_host, int32_t n_grid, int32_t method, int32_t use_cfg,
                  int32_t t_round_to_state_dtype, const lt_step_args* a, void* stream);

/* number of model evaluations issued by the last lt_sample_ode call */
int64_t lt_last_nfe(lt_engine* e);
/* model evaluations served by replaying a captured HIP graph since lt_create (0 with lt_set_option("graph", 0), and below 1025 rows under the default "graph" 2) */
int64_t lt_graph_replays(lt_engine* e);

/* ---- mixture-of-experts routing hooks (parity tests: hold the discrete top-2 choice equal to a reference run's) ----
 * Tables are host int32 [n_layers][2 branches: 0 time-routed, 1 token-routed][rows][2], rows = batch * tokens of the call, expert
 * ids in ascending order per row; branches a variant does not run are -1.  Both hooks switch HIP-graph replay off while active.
 * record(on): every following forward stores the experts moe_route picked; read() returns the last forward's table.
 * force(sel, rows): the following forwards of exactly `rows` rows use these experts instead of their own top-2 (the softmax
 * weights are still computed from the call's own router logits); force(NULL, 0) ends it. */
int lt_moe_routing_record(lt_engine* e, int32_t on);
int lt_moe_routing_read(lt_engine* e, int32_t* host_out, int32_t rows);
int lt_moe_routing_force(lt_engine* e, const int32_t* host_sel, int32_t rows);

/* ---- profiling hooks (bench.py roofline object) ----------------------------------------------- */
/* class 0 = MFMA GEMM kernel, 1 = attention kernel, 2 = everything else.  When enabled, every
 * launch of that class is bracketed by HIP events on the launch stream. */
int lt_profile_enable(lt_engine* e, int32_t on); /* 0 off, 1 all classes, else bit mask: 1 GEMM | 2 attention | 4 other (1 alone = all) */
/* exact class selection: bit 0 GEMM, bit 1 attention, bit 2 everything else (mask 1 = GEMM launches ONLY; 0 = off) */
int lt_profile_enable_mask(lt_engine* e, int32_t mask);
/* after the caller synchronised the stream: total ms, launches and algorithmic flops per class */
int lt_profile_read(lt_engine* e, int32_t klass, double* ms, int64_t* launches, double* flops);
int lt_profile_reset(lt_engine* e);
/* bracket only the first max_event_launches launches of a class after each reset (the rest are counted, and
 * lt_profile_read scales the measured time to all launches: every NFE has the same launch mix); < 0 = no limit.
 * Event brackets serialise the queue (~1.4 ms per NFE when every launch is bracketed), so bench.py samples. */
int lt_profile_set_budget(lt_engine* e, int32_t klass, int64_t max_event_launches);
/* the same with the bracketed launches taken from the MIDDLE of a run: launches skip_launches .. skip_launches + max_event_launches
 * after a reset (the first launches after an idle queue run in a different power state than the steady stream) */
int lt_profile_set_window(lt_engine* e, int32_t klass, int64_t skip_launches, int64_t max_event_launches);

/* ---- operator-level entry points (parity tests call each kernel through these) ---------------- */
/* C[M,N] = A[M,K] * W[N,K]^T (+bias[N]) ; bf16 in/out, fp32 accumulate.  K % 64 == 0.
 * epilogue 0: plain, 1: SwiGLU on 32-row interleaved W (out has N/2 columns: silu(w1 x) * (w3 x)).
 * variant 0: what the engine uses (kernel picked from the problem size); 1 / 2: 256x256 / 256x288 tiles, classic double-buffered
 * loop; 3: 256x256 8-wave ping-pong; 7 / 8: 128x128 / 64x128 small-M tiles; 15 / 16: ONE persistent 4-wave kernel on 16x16x32 MFMAs,
 * 256x256 / 256x288 tiles, LDS ring and DMA prefetch carried across tile boundaries (K % 64 == 0, K >= 128, no bias) - the engine's
 * kernel for every large dense GEMM.
 * 4-6, 9-14, 17, 18 were study kernels of rounds 1-3 (csrc/experimental/, deleted in round 5) and are refused by name. */
int lt_op_gemm_bf16(const void* A_dev, const void* W_dev, const void* bias_dev, int32_t bias_dtype,
                    void* C_dev, int32_t M, int32_t N, int32_t K, int32_t epilogue, int32_t variant,
                    void* stream);
/* V projection with the V^T epilogue: vt[b][h][d][perm(tok)] = sum_k A[b * tokens + tok][k] * W[h * hd + d][k] - the attention
 * kernels' transposed, key-permuted V image (lt_op_v_transpose layout with Npad = tokens) written straight from the GEMM.
 * M = B * tokens, tokens % 64 == 0, N = kv_heads * hd; variant 0 auto | 1 256x256 | 2 256x288. */
int lt_op_gemm_vt(const void* A_dev, const void* W_dev, void* vt_dev, int32_t M, int32_t N, int32_t K, int32_t tokens,
                  int32_t hd, int32_t variant, void* stream);
/* fused QKV projection (the engine's form at large M): C[M, N] gets columns [0, split) as a plain GEMM (row stride N; columns >= split
 * are left untouched), vt gets the V columns [split, N) as the transposed image of lt_op_gemm_vt.  One launch of the persistent
 * kernel on 256 x 288 tiles (split and N - split multiples of 288) or, round 3, 256 x 256 tiles (multiples of 256: Flag-DiT 5B);
 * tokens % 64 == 0 (a sample may end inside a row tile), M % tokens == 0, K % 64 == 0, ceil(M / 256) * N / tile width >= #CUs. */
int lt_op_gemm_qkv(const void* A_dev, const void* W_dev, void* C_dev, void* vt_dev, int32_t M, int32_t N, int32_t K, int32_t split,
                   int32_t tokens, int32_t hd, void* stream);
/* 1 if the engine runs this QKV projection as ONE launch (lt_op_gemm_qkv's conditions and the options allow it), else 0 */
int lt_op_gemm_qkv_fusable(int32_t M, int32_t N, int32_t K, int32_t split, int32_t tokens, int32_t hd);
/* round 6: the ROW-PAIR-INTERLEAVED operand layout of the persistent dense GEMM kernel.  Element (r, k) of a dense [rows][cols] bf16 matrix sits at
 * (r >> 1) * 2 cols + (k >> 5) * 64 + (r & 1) * 32 + (k & 31): the 64-byte pieces a 32-deep K slab takes from rows 2 i and 2 i + 1 are ONE 128-byte
 * line, so the kernel's LDS-DMA stream asks the L2 for whole lines (half the requests for the same bytes; the engine keeps the dense blocks'
 * GEMM weights and their activation operands in it whenever all four GEMMs of the block run on that kernel - lumina_dit_debug.h names the switch).
 * lt_op_pair_layout converts in place (to_pair 1: row-major -> pair, 0: back; rows even, cols % 32 == 0, cols <= 16384).
 * lt_op_gemm_bf16_pair = lt_op_gemm_bf16 (no bias, variant 0, shapes that run on the persistent kernel: >= one 256-row tile per CU) with A and W
 * in the pair layout; pair_c != 0 (epilogue 1 only): the [M][N / 2] output is written in it too.  Same products in the same order as the
 * row-major call: bit-identical results. */
int lt_op_pair_layout(void* m_dev, int64_t rows, int32_t cols, int32_t to_pair, void* stream);
int lt_op_gemm_bf16_pair(const void* A_dev, const void* W_dev, void* C_dev, int32_t M, int32_t N, int32_t K, int32_t epilogue, int32_t pair_c,
                         void* stream);
/* name of the kernel lt_op_gemm_bf16(..., variant) would launch for a dense problem (bench.py labels its roofline line with it) */
int lt_op_gemm_describe(int32_t M, int32_t N, int32_t K, int32_t epilogue, int32_t variant, char* out, int32_t cap);
/* the routing plan of one mixture-of-experts FFN (what replaces the host loop `for i, expert in enumerate(self.experts): batch_idx, nth =
 * torch.where(selected == i)` of Next-DiT-MoE/models/models2.py:470-476, :499-505): sel_dev int32 [rows][2] = every row's two experts in
 * ascending id (lt_op_attention-style callers write it; with sample_logits_dev != null - bf16 [rows / rows_per_sample][E], the time
 * branch - the kernel routes first and WRITES sel_dev and the bf16 softmax weights wts_dev [rows][2]).  Outputs: pos_dev int32
 * [rows][2] sorted position of each (row, expert) entry; src_dev int32 [max_tiles * 256] row of each sorted position, -1 = padding;
 * tile_expert_dev int32 [max_tiles] expert of each 256-row tile, -1 behind the last segment.  Segments are in expert order, each
 * starts on a tile, entries keep their row order inside a segment.  sel_dev / pos_dev need 4 ints of slack behind the last entry
 * (16-byte accesses); max_tiles * 256 >= 2 rows + E * 255.  Integer work: bit-exact against oracle/moe_plan_oracle.py. */
int lt_op_moe_plan(void* sel_dev, const void* sample_logits_dev, void* wts_dev, int32_t rows, int32_t rows_per_sample, int32_t E, void* pos_dev,
                   void* src_dev, void* tile_expert_dev, int32_t max_tiles, void* stream);
/* lt_op_gemm_bf16 on the 64 x 128 small-M tile with the K range split over two workgroups per tile (round 4: how the engine runs the
 * 512-row O / W2 projections of the 600M models - F.linear of Next-DiT-ImageNet/models/models.py:403, :494).  The caller lends the
 * workspace: part_f32 [tiles][2][64 * 128] floats and counters_u32 [tiles] (zero before the first launch, left zero by every launch);
 * tiles >= ceil(M / 64) * ceil(N / 128), K % 512 == 0, K >= 1024, else the call runs unsplit.  Launches sharing a workspace must be
 * stream-ordered.  Result: bf16(fp32 sum of the two halves' fp32 partial sums) - independent of which half finishes last. */
int lt_op_gemm_splitk(const void* A_dev, const void* W_dev, void* C_dev, int32_t M, int32_t N, int32_t K, void* part_f32_dev,
                      void* counters_u32_dev, int32_t tiles, void* stream);
/* lt_op_gemm_splitk with the tile shape and the split left to the launcher, exactly as the engine calls it (variant 0): slots of
 * [2][64 * 128] floats in part_f32 and one counter each; a dense plain-epilogue small-M problem splits K two ways on 64 x 128 tiles (one slot per
 * tile, K >= 1024) or - round 5, K >= 4096, K % 1024 == 0, 4 x ceil(M / 128) x ceil(N / 128) <= #CUs and <= slots - four ways on 128 x 128 tiles
 * (four slots per tile; the last arriver sums the four fp32 partials in K order), else runs unsplit. */
int lt_op_gemm_splitk_auto(const void* A_dev, const void* W_dev, void* C_dev, int32_t M, int32_t N, int32_t K, void* part_f32_dev,
                           void* counters_u32_dev, int32_t slots, void* stream);
/* grouped (mixture-of-experts) form of lt_op_gemm_bf16 - replaces the per-expert Python loop `for i, expert in
 * enumerate(self.experts): ... expert(x[batch_idx])` of Next-DiT-MoE/models/models2.py:470-476, :499-505 on expert-sorted
 * rows: rows [256 t, 256 t + 256) of A multiply with W_dev + tile_expert[t] * w_expert_stride (elements); tile_expert[t] < 0
 * marks a padding segment whose output rows are left untouched.  M % 256 == 0; tile_expert_dev: int32 [M / 256]. */
int lt_op_gemm_grouped(const void* A_dev, const void* W_dev, const void* tile_expert_dev, int64_t w_expert_stride,
                       void* C_dev, int32_t M, int32_t N, int32_t K, int32_t epilogue, int32_t variant, void* stream);
/* lt_op_gemm_grouped (plain epilogue) on the persistent kernel with its tail split: when the valid tiles are not a whole number of rounds of the
 * CUs, the tiles of the partial last round are cut along K into 2 / 4 parts; parts hand fp32 accumulators through tail_part_f32_dev
 * ([cap_parts][256 x 256] floats) and count in on counters_u32_dev ([number of CUs] words, zero before the first launch and after every
 * launch); the last part of a tile to arrive sums the parts in K order, so the result does not depend on the arrival order. */
int lt_op_gemm_grouped_tail(const void* A_dev, const void* W_dev, const void* tile_expert_dev, int64_t w_expert_stride, void* C_dev,
                            int32_t M, int32_t N, int32_t K, void* tail_part_f32_dev, void* counters_u32_dev, int32_t cap_parts, void* stream);
/* the same with gather-on-load (round 3: how the engine runs the experts' W1 | W3 GEMM - no gather pass, no expert-sorted copy of
 * the FFN input): row m of the problem is row row_map_dev[m] (int32 [M]) of A_dev [a_rows, K]; -1 = a padding row that reads as
 * zero.  Ping-pong tile kernels (explicit 3 / 7 / 8) and the grouped mode of the persistent 16x16x32 kernel (explicit 15; K >= 256, all of A
 * below 2^30 bytes, at most 1024 row segments); variant 0 picks the persistent kernel from 1.5 tiles of 256 x 256 per CU on (option gemm_w4q_grouped: 2 = from two tiles per CU, 0 = never). */
int lt_op_gemm_grouped_gather(const void* A_dev, int32_t a_rows, const void* row_map_dev, const void* W_dev, const void* tile_expert_dev,
                              int64_t w_expert_stride, void* C_dev, int32_t M, int32_t N, int32_t K, int32_t epilogue, int32_t variant,
                              void* stream);
/* interleave w1[F,K], w3[F,K] into the packed [2F,K] layout epilogue 1 expects */
int lt_op_pack_w13(const void* w1_dev, const void* w3_dev, void* out_dev, int32_t F, int32_t K,
                   void* stream);
/* out = RMSNorm(x; w, eps) * (1 + scale[b]) (+ shift[b]);  w/scale/shift may be NULL.
 * x,out bf16 [B*N, d]; scale/shift bf16 [B, ld_mod] (row stride ld_mod elements).
 * scale_pre != 0: `scale` already holds bf16(1 + scale) (lt_op_prep_mod), as the engine prepares it once per NFE. */
int lt_op_rmsnorm_mod(const void* x_dev, const void* w_dev, const void* scale_dev,
                      const void* shift_dev, int32_t ld_mod, void* out_dev, int32_t B, int32_t N,
                      int32_t d, float eps, int32_t scale_pre, void* stream);
/* x += gate' * post(y) ; h = pre_next(x) * (1+scale) (+shift)      (model.py:597-610)
 * post_mode 0: y, 1: RMSNorm(y; post_w).  gate_mode 0: gate, 1: tanh(gate), 2: no gate.
 * next_mode 0: none, 1: RMSNorm(x; next_w)(w may be NULL), 2: LayerNorm no-affine (eps_next). */
int lt_op_gated_residual_norm(void* x_dev, const void* y_dev, const void* post_w_dev,
                              const void* gate_dev, int32_t post_mode, int32_t gate_mode,
                              const void* next_w_dev, const void* next_scale_dev,
                              const void* next_shift_dev, int32_t next_mode, int32_t ld_mod,
                              void* h_dev, int32_t B, int32_t N, int32_t d, float eps,
                              float eps_next, int32_t scale_pre, void* stream);
/* y = A W^T (A bf16 [B*N, K], W bf16 [d, K]; `wo` / `w2`, model.py:436-438, :502) followed by the sandwich-norm step on it with the
 * engine's prepared vectors (gate = tanh(gate), next_scale = 1 + scale, both bf16 [B, ld_mod]):
 *   x += gate * RMSNorm(y; post_w) ; h = RMSNorm(x; next_w) * next_scale          (model.py:597-610)
 * use_ystat 1 (what the engine does by default): the projection's epilogue leaves every row's sum of squares in ystat_ws (fp32 [B*N, ystat_cap],
 * ystat_cap >= 2 * ceil(d / 256)) and the row kernel streams; refused when the problem does not take the persistent GEMM kernel.
 * use_ystat 0: the row kernel reduces y itself.  The two differ in the fp32 summation order of that statistic only. */
int lt_op_proj_gated_residual_norm(const void* A_dev, const void* W_dev, void* y_dev, void* ystat_ws_dev, int32_t ystat_cap, int32_t K,
                                   void* x_dev, const void* post_w_dev, const void* gate_dev, const void* next_w_dev,
                                   const void* next_scale_dev, int32_t ld_mod, void* h_dev, int32_t B, int32_t N, int32_t d, float eps,
                                   int32_t use_ystat, void* stream);
/* adaLN vectors, in place (mod bf16 [B, ld_mod] = L layers x `chunks` chunks of d, then the final layer's chunks): chunk c of
 * every layer -> bf16(tanh(.)) if bit c of tanh_mask, bf16(1 + .) if bit c of scale_mask; final_scale_chunk >= 0: that chunk
 * of the final layer -> bf16(1 + .).  The row kernels then run with gate_mode 0 and scale_pre 1. */
int lt_op_prep_mod(void* mod_dev, int32_t B, int32_t ld_mod, int32_t L, int32_t chunks, int32_t d,
                   uint32_t tanh_mask, uint32_t scale_mask, int32_t final_scale_chunk, void* stream);
/* q/k post-processing (model.py:361-371): affine LayerNorm over the full width (optional), 2-D or
 * 1-D RoPE, cast bf16, write head-major [B,heads,N,hd].  src bf16 [B*N, ld_src] at column col0.
 * rope_mode 0 none, 1 2-D interleaved (Next-DiT, model.py:959-961), 2 1-D (Flag-DiT).
 * cs_table_dev float2 [pos][hd/4 or hd/2] (cos,sin); grid_w = latent tokens per row.  out_scale multiplies the
 * result before that one rounding (1.0 = reference layout; the engine folds softmax_scale * log2(e) into K). */
int lt_op_qk_norm_rope(const void* src_dev, int32_t ld_src, int32_t col0, const void* ln_w_dev,
                       const void* ln_b_dev, float ln_eps, void* dst_dev, int32_t B, int32_t N,
                       int32_t heads, int32_t hd, int32_t rope_mode, const void* cs_table_dev,
                       int32_t grid_w, float out_scale, void* stream);
/* V -> transposed, key-permuted layout the attention kernel consumes: [B,kvh,hd,Npad] */
int lt_op_v_transpose(const void* src_dev, int32_t ld_src, int32_t col0, void* dst_dev, int32_t B,
                      int32_t N, int32_t Npad, int32_t kv_heads, int32_t hd, void* stream);
/* non-causal softmax(q k^T * scale + bias) v  (model.py:392-405 / 427-432).
 * q [B,H,N,hd], k [B,Hkv,Nk,hd], vt [B,Hkv,hd,Nkpad] (lt_op_v_transpose layout), bias float
 * [B,Nkpad] or NULL (0 / -inf per key), out bf16 [B,N,H*hd].
 * accumulate != 0: out = out + tanh(gate[h]) * result (model.py:433-434), gate bf16 [H].
 * k_prescaled != 0: k already carries scale * log2(e) (lt_op_qk_norm_rope out_scale) and `scale` is not applied again. */
int lt_op_attention(const void* q_dev, const void* k_dev, const void* vt_dev, const float* bias_dev,
                    void* out_dev, const void* gate_dev, int32_t accumulate, int32_t B, int32_t H,
                    int32_t Hkv, int32_t N, int32_t Nk, int32_t Nkpad, int32_t hd, float scale,
                    int32_t k_prescaled, void* stream);
/* The two launches of a layer of the class-conditional 600M models at <= 512 tokens (round 5, option attn_small_fused): the QKV
 * projection on the small-M GEMM tiles, whose epilogue also leaves per-row (sum, sum of squares) of every 128-column tile in
 * rowstat_ws ([M][ceil(3 widths / 128)] float2), then ONE kernel that does q_norm / k_norm (affine LayerNorm over the full width, fp32),
 * 2-D RoPE, the softmax scale fold (k_scale = scale * log2 e) and bf16 rounding of q and k, the V transpose and the attention itself
 * (Next-DiT-ImageNet/models/models.py:358-404).  A [M, K], W [H hd + 2 Hkv hd, K] (q | k | v rows), qkv [M, H hd + 2 Hkv hd] (written),
 * LayerNorm weights / biases bf16 [H hd] / [Hkv hd], cs_table = the (cos, sin) table of lt_op_qk_norm_rope in rope_mode 1 (branch 1 is
 * used), out [M / tokens, tokens, H hd].  head_dim 48, 64 <= tokens <= 512 and tokens % 64 == 0, widths % 128 == 0. */
int lt_op_qkv_attention_small(const void* A_dev, const void* W_dev, void* qkv_dev, int32_t M, int32_t K, int32_t H, int32_t Hkv,
                              int32_t tokens, int32_t hd, const void* q_ln_w, const void* q_ln_b, const void* k_ln_w,
                              const void* k_ln_b, const void* cs_table, int32_t table_len, int32_t grid_w, float k_scale,
                              void* rowstat_ws, void* out_dev, void* stream);
/* self-attention + zero-init gated text cross-attention in ONE launch (hd 72 / 96, attention_variant 3 or 4; model.py:392-434):
 *   out = bf16(softmax(q k^T) v) + bf16(bf16(softmax(q tk^T + tbias) tv) * tanh(tgate[h]))
 * k and tk must already carry their softmax scale * log2(e) (lt_op_qk_norm_rope out_scale); layouts as lt_op_attention,
 * tk [B,Hkv,Tk,hd], tvt [B,Hkv,hd,Tkpad], tbias float [B,Tkpad] (0 / -inf, -inf in the padding), tgate bf16 [H]. */
int lt_op_attention_fused(const void* q_dev, const void* k_dev, const void* vt_dev, const void* tk_dev,
                          const void* tvt_dev, const float* tbias_dev, const void* tgate_dev, void* out_dev, int32_t B,
                          int32_t H, int32_t Hkv, int32_t N, int32_t Nk, int32_t Nkpad, int32_t Tk, int32_t Tkpad,
                          int32_t hd, void* stream);
/* y[m,n] = sum_k act(a[m,k]) w[n,k] + b[n], m < M <= 8 (GEMV-style; adaLN / embedders).
 * act_in 0 none, 1 SiLU.  a bf16 [M,K], w bf16 [N,K], b bf16 [N] or NULL, y bf16 [M,N]. */
int lt_op_linear_small_m(const void* a_dev, const void* w_dev, const void* b_dev, void* y_dev,
                         int32_t M, int32_t N, int32_t K, int32_t act_in, void* stream);
/* 2-D RoPE (cos,sin) table builder (model.py:915-963): out float2 [2 branches][len][hd/4];
 * branch 0 = linear-interpolation (t < watershed), branch 1 = NTK. */
int lt_op_rope_table_2d(void* out_dev, int32_t len, int32_t hd, float theta, float scale_factor,
                        void* stream);
/* the same table in both layouts the engine keeps (round 4): out float2 [2][len][hd/4] and out_t float2 [2][hd/4][len] (the column
 * factors of the attention prologue: 32 consecutive positions per load) */
int lt_op_rope_table_2d_pair(void* out_dev, void* out_t_dev, int32_t len, int32_t hd, float theta, float scale_factor, void* stream);
/* Round 4, the attn_q_fused path of the engine as two stand-alone steps (model.py:355-371: wq | wk | wv, q_norm, k_norm, RoPE).
 * lt_op_qkv_qstat = lt_op_gemm_qkv (same layouts and conditions; q_cols a multiple of the launch's tile width) whose epilogue also leaves
 * per-row partial (sum, sum of squares) of the bf16-rounded Q columns in qstat_ws (float2 [M][32], scratch), followed by the K pass of
 * lt_op_qk_norm_rope (LayerNorm over the K columns [q_cols, split), 2-D RoPE from cs_table = ONE branch's [len][hd/4] table, out_scale folded in,
 * head-major k_out [B, (split - q_cols) / hd, tokens, hd]) which reduces the partials to q_mean_rstd float2 [M] = (mean, rsqrt(var + 1e-5))
 * of each row's Q columns.
 * lt_op_attention_qraw = lt_op_attention (k prescaled, no bias, head_dim 72, N % 64 == 0) with q == NULL: the kernel builds its query
 * fragments from qkv [B * N, ld] (this head's columns at q_col0 + h * hd) as RoPE(LayerNorm(x; q_mean_rstd, q_ln_w, q_ln_b)) with ONE
 * bf16 rounding - cs_table / cs_table_t from lt_op_rope_table_2d_pair (branch 1), token n at grid position (n / grid_w, n % grid_w). */
int lt_op_qkv_qstat(const void* A_dev, const void* W_dev, void* C_dev, void* vt_dev, int32_t M, int32_t N, int32_t K, int32_t split,
                    int32_t tokens, int32_t hd, int32_t q_cols, const void* k_ln_w_dev, const void* k_ln_b_dev, const void* cs_table_dev,
                    int32_t grid_w, float k_out_scale, void* k_out_dev, void* qstat_ws_dev, void* q_mean_rstd_dev, void* stream);
int lt_op_attention_qraw(const void* qkv_dev, int32_t ld, int32_t q_col0, const void* q_mean_rstd_dev, const void* q_ln_w_dev,
                         const void* q_ln_b_dev, const void* cs_table_dev, const void* cs_table_t_dev, int32_t table_len, int32_t grid_w,
                         const void* k_dev, const void* vt_dev, void* out_dev, int32_t B, int32_t H, int32_t Hkv, int32_t N, int32_t Nkpad,
                         int32_t hd, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LUMINA_DIT_H */
