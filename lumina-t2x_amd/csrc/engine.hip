// Host side of the C ABI (include/lumina_dit.h): engine object, weight arena, workspace, the per-NFE
// launch sequence of NextDiT.forward / forward_with_cfg (lumina_next_t2i/models/model.py:836-913) and the
// fixed-grid ODE loop of transport/integrators.py:104-116 (torchdiffeq euler / midpoint / rk4).
//
// Everything here is asynchronous on the caller's stream: no host<->device sync inside a step (the
// reference syncs >= 25 times per NFE: t[0].item() model.py:888, nonzero/.item() per layer :288-289).
// Step-invariant work is hoisted: text K/V of all layers and the caption embedding are computed once per
// prompt (lt_prepare_prompt); the adaLN vectors of all layers are one GEMV per NFE; the RoPE table is
// a 2 x 384 x hd/4 (cos,sin) table rebuilt only when scale_factor changes.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/lumina_dit.h"
#include "common.h"
#include "kernels.h"
#include "options.h"

// ------------------------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";
void lt_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* lt_last_error(void) { return g_err; }
extern "C" const char* lt_version(void) { return "lumina_dit gfx950 r5"; }

namespace {

constexpr float LOG2E = 1.44269504088896340736f;
// Kernel-selection options (options.h; lt_set_option = process default, lt_engine_set_option = per-engine override):
//   qkv_post_fused   1 = one launch for q / k post-processing + V transpose, 0 = three launches, 2 (default) = one launch where the
//                    problem is launch-bound (fewer than 2048 rows: the three passes are 5 us each at 512 rows, i.e. pure launch latency -
//                    profiles/r02/rocprofv3_kernel_stats_cfg1_r02.csv), three where it is bandwidth-bound (in-situ A/B at cfg 2 with the
//                    LDS-staged row kernels, profiles/r01/bench_ab_qkv_post_fused.log: three launches 0.1-0.2 ms / NFE faster)
//   qkv_vt_epilogue  1 = the V projection is its own GEMM launch whose epilogue writes the attention kernels' V^T image (no v_transpose
//                    pass: 15.6 us per layer at cfg 2, and the 37.7 MB V slice is never written row-major / re-read)
//   qkv_fused_gemm   Q | K | V in one launch of the persistent kernel where the shapes allow it
//   attn_q_fused     q_norm + RoPE of the queries inside the attention prologue (hd 72 one-wave kernel, fused QKV GEMM)
//   qk_post_pair     1 = q and k post-processing share one persistent launch (large problems)
//   graph            1 = a model evaluation (~250 launches) is captured into a HIP graph per (arguments, shapes) and replayed; 2 (default) = above
//                    1024 rows only (forward_graphed).  Every option
//                    change moves lt_opt_generation(), which is part of the graph key (kernel selection is baked into a captured graph).

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct LayerW {
    u16 *wqkv = nullptr, *wo = nullptr, *w13 = nullptr, *w2 = nullptr, *wkvy = nullptr;
    u16 *q_norm_w = nullptr, *q_norm_b = nullptr, *k_norm_w = nullptr, *k_norm_b = nullptr;
    u16 *ky_norm_w = nullptr, *ky_norm_b = nullptr, *gate = nullptr;
    u16 *attn_norm1 = nullptr, *attn_norm2 = nullptr, *ffn_norm1 = nullptr, *ffn_norm2 = nullptr, *y_norm = nullptr;
    u16 *ky = nullptr, *vty = nullptr;  // hoisted text K / V^T of the current prompt
    // MoE family (models2.py:731-745): E experts per branch, w13 packed per expert [E][2F, d], w2 [E][d, F]
    u16 *w13_t = nullptr, *w2_t = nullptr, *w13_s = nullptr, *w2_s = nullptr, *gate_t = nullptr, *gate_s = nullptr;
    u16 *norm_time = nullptr, *norm_space = nullptr;
};

struct ProfClass {
    double flops = 0;
    long long launches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    size_t used = 0;
    size_t budget = (size_t)-1;  // bracket at most this many launches with events, count the rest (same launch mix every step)
    long long skip = 0;          // ... starting with launch number `skip` after a reset (a window in the middle of a timed region)
};

}  // namespace

// What differs between the reference's model families on this path (everything else is shared code):
//                    NEXT_T2I (model.py:573-662)   NEXT_IMAGENET (models.py:759-833)   FLAG_T2I (lumina_t2i model.py:572-658)
//  adaLN chunks      scale,gate | scale,gate       scale,gate | scale,gate             shift,scale,gate | shift,scale,gate
//  pre-norm weight   attention_norm1 / ffn_norm1   none (PFRMSNorm)                    attention_norm / ffn_norm
//  post-norm         attention_norm2 / ffn_norm2   attention_norm / ffn_norm           none
//  gate              tanh                          tanh                                plain
//  conditioning      text (cross-attn + pooled)    class label embedding               text (cross-attn + pooled)
//  RoPE              2-D, watershed branches       2-D, (rope_scaling, ntk) at once    1-D over the flattened rows, eol tokens
//  final layer       scale                         shift, scale                        shift, scale
struct VariantDesc {
    int chunks;                  // adaLN chunks per layer
    int i_shift[2], i_scale[2], i_gate[2];  // chunk index of {attention, ffn} branch; -1 = absent
    bool pre_w, post, gate_tanh, text, labels, rope_1d, eol;
    int final_chunks;            // 1: scale;  2: shift, scale
};

static VariantDesc variant_desc(int variant) {
    VariantDesc v{};
    if (variant == LT_VARIANT_NEXT_T2I) {
        v = {4, {-1, -1}, {0, 2}, {1, 3}, true, true, true, true, false, false, false, 1};
    } else if (variant == LT_VARIANT_NEXT_IMAGENET || variant == LT_VARIANT_NEXT_MOE_TIME || variant == LT_VARIANT_NEXT_MOE_SPACE) {
        // (the single-MoE models of Next-DiT-MoE/models/models.py / models1.py are the ImageNet block with its FFN replaced
        //  by one MoeLayer: four adaLN chunks, same norms - models.py:749-758)
        v = {4, {-1, -1}, {0, 2}, {1, 3}, false, true, true, false, true, false, false, 2};
    } else if (variant == LT_VARIANT_NEXT_MOE) {
        // models2.py:783-784: scale_msa, gate_msa, scale_mlp_time, gate_mlp_time, scale_mlp_space, gate_mlp_space;
        // index [1] is the time branch, the space branch uses chunks 4 / 5 (run_forward)
        v = {6, {-1, -1}, {0, 2}, {1, 3}, false, true, true, false, true, false, false, 2};
    } else {  // LT_VARIANT_FLAG_T2I
        v = {6, {0, 3}, {1, 4}, {2, 5}, true, false, false, true, false, true, true, 2};
    }
    return v;
}

struct lt_engine {
    lt_config cfg;
    LtEngineOptions opts;  // per-engine option overrides (lt_engine_set_option); LT_OPT_INHERIT slots follow the process defaults
    VariantDesc v;
    int d, L, H, Hkv, hd, F, dkv, qkvn, A, cap, nfinal, kpad, chunks, ld_mod;
    std::vector<DevBuf> allocs;
    std::vector<LayerW> lw;
    // globals
    u16 *xemb_w = nullptr, *xemb_b = nullptr, *t0_w = nullptr, *t0_b = nullptr, *t2_w = nullptr, *t2_b = nullptr;
    u16 *capln_w = nullptr, *capln_b = nullptr, *cape_w = nullptr, *cape_b = nullptr, *pad_token = nullptr;
    u16 *adaln_w = nullptr, *adaln_b = nullptr;  // [L*chunks*d + d, A], [L*chunks*d + d]
    u16 *final_w = nullptr, *final_b = nullptr;
    u16 *label_table = nullptr, *eol_token = nullptr;
    int label_rows = 0;
    std::map<std::string, bool> need;
    bool weights_ok = false;
    // round 6: the dense blocks' four GEMM weights (wqkv, wo, w13, w2 of every layer) are held either row-major or in the row-pair-interleaved
    // layout the persistent GEMM reads with whole-line requests (GemmArgs::pair_ab); ensure_weight_layout converts all of them in place when an
    // evaluation needs the other one (a change of regime: >= one tile per CU <-> the small-M kernels).  last_pair: what the last run_forward used.
    bool w_pair = false, last_pair = false;
    // workspace
    u16 *x = nullptr, *h = nullptr, *qkv = nullptr, *q = nullptr, *k = nullptr, *vt = nullptr, *attn = nullptr;
    u16 *o = nullptr, *u = nullptr, *patches = nullptr, *frows = nullptr, *mod = nullptr;
    u16 *tfeat = nullptr, *t1 = nullptr, *temb = nullptr, *cap_ln = nullptr, *cap_emb = nullptr, *adaln_in = nullptr;
    // MoE workspace: expert-sorted rows (moe.hip)
    int E = 0, moe_tiles = 0;
    int moe_mode = 0;  // 0: time + space MoE per block (models2.py), 1: time-routed MoE only (models.py), 2: token-routed only (models1.py)
    u16 *moe_us = nullptr, *moe_ys = nullptr, *moe_logits = nullptr, *moe_wts = nullptr;
    u16* gate_t_all = nullptr;  // [L * E, A]: every layer's time-router weight, contiguous (LayerW::gate_t point into it)
    int *moe_sel = nullptr, *moe_pos = nullptr, *moe_tile_expert = nullptr, *moe_src = nullptr;
    // round 5 (option moe_time_plan_hoist): one plan per layer for the time router, all written by ONE launch at the top of the evaluation
    int *moe_tp_sel = nullptr, *moe_tp_pos = nullptr, *moe_tp_tile_expert = nullptr, *moe_tp_src = nullptr;
    u16* moe_tp_wts = nullptr;
    size_t moe_tp_stride_rows = 0, moe_tp_stride_src = 0;
    bool moe_tp_live = false;  // this evaluation's time plans were hoisted (set by run_forward, read by moe_ffn / moe_y)
    // parity hooks (lt_moe_routing_*): [L][2 branches][max rows][2] expert ids, recorded from / forced onto moe_route_kernel
    int *moe_rec = nullptr, *moe_force = nullptr;
    int moe_rec_on = 0, moe_force_rows = 0, moe_rec_rows = 0;
    int qstat_slots = 32;
    float* qstat = nullptr;  // [rows][qstat_slots] float2: LayerNorm partial sums of the Q columns, written by the fused QKV GEMM (GemmArgs::qstat)
    float* ystat = nullptr;  // [rows][ystat_cap] floats: per-row sum-of-squares partials of the O / W2 projection's output (GemmArgs::ystat -> GatedResArgs::ystat)
    int ystat_cap = 0;
    float* attn_tail_ws = nullptr;  // hd 96 only: partials of the attention launch's split last query block (AttnArgs::tail_ws)
    size_t attn_tail_ws_bytes = 0;
    float* qmr = nullptr;    // [rows] float2 (mean, rstd) of the Q rows, reduced from qstat by the K pass of qk_norm_rope (AttnArgs::q_stat)
    float* rope_tr = nullptr;  // the 2-D rotary table once more as [branch][freq][pos] (AttnArgs::rope_cs_t)
    // split-K workspace of the 512-row-class GEMMs (GemmArgs::splitk_*): 128 tiles = one round of half the CUs
    float* splitk_part = nullptr;
    unsigned* splitk_cnt = nullptr;
    // tail split of the grouped persistent GEMM (GemmArgs::tail_*; MoE engines with >= 4096 rows): fp32 parts + arrival counters
    float* tail_part = nullptr;
    unsigned* tail_cnt = nullptr;
    long long tail_cap_parts = 0;
    int splitk_tiles = 0;
    u16 *capb = nullptr, *capn = nullptr, *kvy = nullptr;
    float* txt_bias = nullptr;
    float* rope = nullptr;
    float rope_scale = -1.f, rope_ntk = -1.f;
    int rope_len = 0;
    int prompt_B = 0, prompt_T = 0, prompt_Tpad = 0;
    // ode
    void *ys[2] = {nullptr, nullptr}, *ymid = nullptr, *kbuf[4] = {nullptr, nullptr, nullptr, nullptr};
    float* t_dev = nullptr;
    float* t_pinned = nullptr;       // page-locked staging of the stage times (an async copy from pageable memory synchronises)
    hipEvent_t t_copied = nullptr;   // the previous call's copy out of t_pinned has executed
    int t_cap = 0;
    // compositional (regional) text conditioning (lt_prepare_prompt_regional): Y captions, the first Y-1 belong to regions of
    // the cond row, the last to the uncond row; 0 = off
    int reg_Y = 0, reg_h = 1, reg_w = 1;
    u16* reg_txt = nullptr;    // [Y, max_tokens, d] per-caption text attention outputs
    size_t reg_txt_elems = 0;
    int* reg_qmap = nullptr;   // [max_batch] query batch of each caption
    int* pk_dev = nullptr;     // packed batches: [0,64) token counts, [64,128) grid widths
    int pk_host[128] = {0};
    long long last_nfe = 0;
    // HIP graphs of one model evaluation (forward_graphed): fixed staging buffers the captured kernels read / write, a private
    // stream to capture on (the caller's stream may be the legacy null stream, which cannot capture), cached executables
    struct GraphTally { double flops[3] = {0, 0, 0}; long long launches[3] = {0, 0, 0}; };  // what one replay stands for, per kernel class
    struct GraphEntry { std::vector<char> key; hipGraphExec_t exec = nullptr; int uses = 0; bool failed = false; bool pair = false; GraphTally tally; };
    GraphTally* tally = nullptr;  // set while a graph is being captured: ProfScope counts into it instead of timing
    std::vector<GraphEntry> graphs;
    void *g_x = nullptr, *g_out = nullptr;
    float* g_t = nullptr;
    hipStream_t cap_stream = nullptr;
    long long graph_replays = 0;
    // profiling
    int prof_mask = 0;  // bit k: class k launches are bracketed by HIP events
    bool prof_on = false;
    ProfClass prof[3];
};

namespace {

int dev_alloc(lt_engine* e, void** out, size_t bytes, bool zero = true) {
    void* p = nullptr;
    if (bytes == 0) bytes = 16;
    LT_CHECK_HIP(hipMalloc(&p, bytes));
    if (zero) LT_CHECK_HIP(hipMemset(p, 0, bytes));
    e->allocs.push_back({p, bytes});
    *out = p;
    return 0;
}
struct ProfScope {
    lt_engine* e;
    int k;
    hipStream_t s;
    bool on = false, attach = false;  // attach: the launch carries the event pair itself (no records here)
    size_t slot = 0;
    hipEvent_t ev0() const { return on ? e->prof[k].ev[slot].first : nullptr; }
    hipEvent_t ev1() const { return on ? e->prof[k].ev[slot].second : nullptr; }
    ProfScope(lt_engine* e_, int klass, double flops, hipStream_t s_, bool attach_ = false) : e(e_), k(klass), s(s_), attach(attach_) {
        if (e->tally) {  // graph capture: no events inside a graph, only the bookkeeping a replay will add
            e->tally->flops[k] += flops;
            e->tally->launches[k] += 1;
            return;
        }
        if (!e->prof_on || !((e->prof_mask >> k) & 1)) return;
        ProfClass& pc = e->prof[k];
        pc.flops += flops;
        pc.launches += 1;
        if (pc.launches > pc.skip && pc.used < pc.ev.size() && pc.used < pc.budget) {
            on = true;
            slot = pc.used++;
            if (!attach) (void)hipEventRecord(pc.ev[slot].first, s);
        }
    }
    ~ProfScope() {
        if (on && !attach) (void)hipEventRecord(e->prof[k].ev[slot].second, s);
    }
};

// "gemm_prefetch" 3: the weight panels of the next 512-row-class GEMM are read by rider workgroups inside the row kernel that precedes it.
// (Option 2 - the same reads on a side stream beside the preceding kernel - lost 33 % to +37 us per layer of cross-stream dependencies
//  and was removed in round 5 together with its fork / join state)
void prefetch_rider(lt_engine* e, PrefetchRider* r, const u16* A, int lda, const u16* W, int ldw, u16* C, int ldc, int M, int N, int K, int epi) {
    if (lt_opt(OPT_GEMM_PREFETCH) != 3 || M > 1024) return;
    GemmArgs g;
    g.A = A; g.W = W; g.C = C; g.bias = nullptr; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldw = ldw; g.ldc = ldc;
    g.bias_dtype = -1;
    g.splitk_part = e->splitk_part; g.splitk_cnt = e->splitk_cnt; g.splitk_tiles = e->splitk_tiles;
    if (!gemm_prefetch_rider(g, epi, r)) *r = PrefetchRider();
}

// ystat_slots (optional, out): > 0 when the launch left the rows' sum-of-squares partials in e->ystat (option grn_ystat; GemmArgs::ystat)
int gemm(lt_engine* e, const u16* A, int lda, const u16* W, int ldw, u16* C, int ldc, int M, int N, int K,
         const u16* bias, int epi, hipStream_t s, int* ystat_slots = nullptr, int pair_ab = 0, int pair_c = 0) {
    GemmArgs g;
    g.A = A; g.W = W; g.C = C; g.bias = bias; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldw = ldw; g.ldc = ldc;
    g.bias_dtype = bias ? 1 : -1;
    g.pair_ab = pair_ab; g.pair_c = pair_c;
    g.splitk_part = e->splitk_part; g.splitk_cnt = e->splitk_cnt; g.splitk_tiles = e->splitk_tiles;  // (the launcher decides)
    if (ystat_slots) {
        *ystat_slots = 0;
        const int ys = lt_opt(OPT_GRN_YSTAT) && !bias ? gemm_ystat_slots(g, epi) : 0;
        if (ys > 0 && ys <= e->ystat_cap) { g.ystat = e->ystat; g.ystat_slots = ys; *ystat_slots = ys; }
    }
    if (lt_opt(OPT_GEMM_PREFETCH) == 1 && M <= 1024 && launch_gemm_prefetch_w(g, epi, s)) return 1;
    ProfScope ps(e, 0, 2.0 * M * (double)N * K, s, true);
    return launch_gemm_bf16(g, epi, 0, s, ps.ev0(), ps.ev1());
}

int attention(lt_engine* e, const AttnArgs& a, hipStream_t s) {
    ProfScope ps(e, 1, 4.0 * a.B * a.H * (double)a.N * (a.Nk + (a.tk ? a.Tk : 0)) * a.hd, s);
    return launch_attention(a, s);
}

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// ---- weight table -----------------------------------------------------------------------------------
struct Slot {
    u16* dst;
    int rows, cols, dst_ld, r0, row_map;
};

bool parse_layer_key(const char* key, int* layer, std::string* rest) {
    if (strncmp(key, "layers.", 7) != 0) return false;
    char* end = nullptr;
    long l = strtol(key + 7, &end, 10);
    if (end == key + 7 || *end != '.') return false;
    *layer = (int)l;
    *rest = std::string(end + 1);
    return true;
}

int find_slot(lt_engine* e, const std::string& key, Slot* s) {
    const int d = e->d, A = e->A, cap = e->cap, F = e->F, dkv = e->dkv;
    int l;
    std::string r;
    auto set = [&](u16* dst, int rows, int cols, int ld, int r0 = 0, int map = 0) {
        s->dst = dst; s->rows = rows; s->cols = cols; s->dst_ld = ld; s->r0 = r0; s->row_map = map;
        return 0;
    };
    if (parse_layer_key(key.c_str(), &l, &r)) {
        LT_REQUIRE(l >= 0 && l < e->L, "weight key %s: layer out of range", key.c_str());
        LayerW& w = e->lw[l];
        const int cd = e->chunks * d;
        if (r == "attention.wq.weight") return set(w.wqkv, d, d, d, 0);
        if (r == "attention.wk.weight") return set(w.wqkv, dkv, d, d, d);
        if (r == "attention.wv.weight") return set(w.wqkv, dkv, d, d, d + dkv);
        if (r == "attention.wo.weight") return set(w.wo, d, d, d);
        if (r == "attention.wk_y.weight") return set(w.wkvy, dkv, cap, cap, 0);
        if (r == "attention.wv_y.weight") return set(w.wkvy, dkv, cap, cap, dkv);
        if (r == "attention.q_norm.weight") return set(w.q_norm_w, 1, d, d);
        if (r == "attention.q_norm.bias") return set(w.q_norm_b, 1, d, d);
        if (r == "attention.k_norm.weight") return set(w.k_norm_w, 1, dkv, dkv);
        if (r == "attention.k_norm.bias") return set(w.k_norm_b, 1, dkv, dkv);
        if (r == "attention.ky_norm.weight") return set(w.ky_norm_w, 1, dkv, dkv);
        if (r == "attention.ky_norm.bias") return set(w.ky_norm_b, 1, dkv, dkv);
        if (r == "attention.gate") return set(w.gate, 1, e->H, e->H);
        if (e->E == 0) {
            if (r == "feed_forward.w1.weight") return set(w.w13, F, d, d, 0, 1);
            if (r == "feed_forward.w3.weight") return set(w.w13, F, d, d, 0, 2);
            if (r == "feed_forward.w2.weight") return set(w.w2, d, F, F);
        }
        if (e->E > 0 && e->moe_mode != 0) {  // one MoeLayer as `feed_forward` (models.py:700-707 / models1.py:700-707)
            const bool time = e->moe_mode == 1;
            if (r == "attention_norm.weight") return set(w.attn_norm2, 1, d, d);
            if (r == "ffn_norm.weight") return set(w.ffn_norm2, 1, d, d);
            if (r == "feed_forward.gate.weight") return time ? set(w.gate_t, e->E, A, A) : set(w.gate_s, e->E, d, d);
            const char* pre = "feed_forward.experts.";
            const size_t pl = strlen(pre);
            if (r.compare(0, pl, pre) == 0) {
                char* end = nullptr;
                const long ex = strtol(r.c_str() + pl, &end, 10);
                LT_REQUIRE(end != r.c_str() + pl && ex >= 0 && ex < e->E, "weight key %s: expert out of range", key.c_str());
                const std::string tail(end);
                u16* w13 = (time ? w.w13_t : w.w13_s) + (size_t)ex * 2 * F * d;
                u16* w2 = (time ? w.w2_t : w.w2_s) + (size_t)ex * d * F;
                if (tail == ".w1.weight") return set(w13, F, d, d, 0, 1);
                if (tail == ".w3.weight") return set(w13, F, d, d, 0, 2);
                if (tail == ".w2.weight") return set(w2, d, F, F);
            }
        } else if (e->cfg.variant == LT_VARIANT_NEXT_MOE) {
            if (r == "attention_norm.weight") return set(w.attn_norm2, 1, d, d);
            if (r == "ffn_norm_time.weight") return set(w.norm_time, 1, d, d);
            if (r == "ffn_norm_space.weight") return set(w.norm_space, 1, d, d);
            if (r == "feed_forward_time.gate.weight") return set(w.gate_t, e->E, A, A);
            if (r == "feed_forward_space.gate.weight") return set(w.gate_s, e->E, d, d);
            for (int br = 0; br < 2; ++br) {
                const char* pre = br == 0 ? "feed_forward_time.experts." : "feed_forward_space.experts.";
                const size_t pl = strlen(pre);
                if (r.compare(0, pl, pre) != 0) continue;
                char* end = nullptr;
                const long ex = strtol(r.c_str() + pl, &end, 10);
                LT_REQUIRE(end != r.c_str() + pl && ex >= 0 && ex < e->E, "weight key %s: expert out of range", key.c_str());
                const std::string tail(end);
                u16* w13 = (br == 0 ? w.w13_t : w.w13_s) + (size_t)ex * 2 * F * d;
                u16* w2 = (br == 0 ? w.w2_t : w.w2_s) + (size_t)ex * d * F;
                if (tail == ".w1.weight") return set(w13, F, d, d, 0, 1);
                if (tail == ".w3.weight") return set(w13, F, d, d, 0, 2);
                if (tail == ".w2.weight") return set(w2, d, F, F);
            }
        } else if (e->cfg.variant == LT_VARIANT_NEXT_T2I) {
            if (r == "attention_norm1.weight") return set(w.attn_norm1, 1, d, d);
            if (r == "attention_norm2.weight") return set(w.attn_norm2, 1, d, d);
            if (r == "ffn_norm1.weight") return set(w.ffn_norm1, 1, d, d);
            if (r == "ffn_norm2.weight") return set(w.ffn_norm2, 1, d, d);
        } else if (e->cfg.variant == LT_VARIANT_NEXT_IMAGENET) {  // post-norms carry the weight (models.py:735-738)
            if (r == "attention_norm.weight") return set(w.attn_norm2, 1, d, d);
            if (r == "ffn_norm.weight") return set(w.ffn_norm2, 1, d, d);
        } else {  // Flag-DiT: pre-norms only (lumina_t2i model.py:556-557)
            if (r == "attention_norm.weight") return set(w.attn_norm1, 1, d, d);
            if (r == "ffn_norm.weight") return set(w.ffn_norm1, 1, d, d);
        }
        if (r == "attention_y_norm.weight") return set(w.y_norm, 1, cap, cap);
        if (r == "adaLN_modulation.1.weight") return set(e->adaln_w, cd, A, A, l * cd);
        if (r == "adaLN_modulation.1.bias") return set(e->adaln_b + (size_t)l * cd, 1, cd, cd);
    } else {
        const int cd = e->chunks * d;
        if (key == "x_embedder.weight") return set(e->xemb_w, d, e->cfg.in_channels * e->cfg.patch_size * e->cfg.patch_size, e->kpad);
        if (key == "x_embedder.bias") return set(e->xemb_b, 1, d, d);
        if (key == "t_embedder.mlp.0.weight") return set(e->t0_w, A, 256, 256);
        if (key == "t_embedder.mlp.0.bias") return set(e->t0_b, 1, A, A);
        if (key == "t_embedder.mlp.2.weight") return set(e->t2_w, A, A, A);
        if (key == "t_embedder.mlp.2.bias") return set(e->t2_b, 1, A, A);
        if (key == "cap_embedder.0.weight") return set(e->capln_w, 1, cap, cap);
        if (key == "cap_embedder.0.bias") return set(e->capln_b, 1, cap, cap);
        if (key == "cap_embedder.1.weight") return set(e->cape_w, A, cap, cap);
        if (key == "cap_embedder.1.bias") return set(e->cape_b, 1, A, A);
        if (key == "pad_token") return set(e->pad_token, 1, d, d);
        if (key == "eol_token" && e->v.eol) return set(e->eol_token, 1, d, d);
        if (key == "y_embedder.embedding_table.weight" && e->v.labels) return set(e->label_table, e->label_rows, A, A);
        const int fd = e->v.final_chunks * d;
        if (key == "final_layer.linear.weight") return set(e->final_w, e->nfinal, d, d);
        if (key == "final_layer.linear.bias") return set(e->final_b, 1, e->nfinal, e->nfinal);
        if (key == "final_layer.adaLN_modulation.1.weight") return set(e->adaln_w, fd, A, A, e->L * cd);
        if (key == "final_layer.adaLN_modulation.1.bias") return set(e->adaln_b + (size_t)e->L * cd, 1, fd, fd);
    }
    lt_set_error("unknown weight key '%s' for this variant", key.c_str());
    return 2;
}

// (cos,sin) factor table(s) for this call's RoPE arguments; rebuilt only when they change
int ensure_rope(lt_engine* e, const lt_step_args* a, hipStream_t s) {
    const float ntk = a->ntk_factor > 0.f ? a->ntk_factor : 1.0f;
    const float sf = a->scale_factor > 0.f ? a->scale_factor : 1.0f;
    if (e->rope_scale == sf && e->rope_ntk == ntk) return 0;
    if (e->cfg.variant == LT_VARIANT_NEXT_T2I) {
        if (launch_rope_table_2d(e->rope, e->rope_len, e->hd, 10000.0f, sf, s, e->rope_tr)) return 1;
    } else {
        // both branches identical: theta * ntk_factor, positions / rope_scaling_factor (models.py:1001-1005, model.py:948-955)
        const int step = e->v.rope_1d ? 2 : 4;
        if (launch_rope_table(e->rope, e->rope_len, e->hd, step, 10000.0f * ntk, sf, 10000.0f * ntk, sf, 1, s, e->rope_tr)) return 1;
    }
    e->rope_scale = sf;
    e->rope_ntk = ntk;
    return 0;
}

// one MoE feed-forward (branch 0 = TimeMoeLayer on the timestep embedding, 1 = SpaceMoeLayer on the tokens;
// models2.py:451-506): e->h -> e->o
// routed: sel / wts of this (token-routed) branch were already written by the row kernel that produced its input (GatedResArgs::route_*)
// pair: the experts' weights are in the pair layout (run_forward decided; ensure_weight_layout), and so is the SwiGLU output e->moe_us
int moe_ffn(lt_engine* e, LayerW& w, int layer, int branch, int M, int N, int B, hipStream_t s, bool routed = false, bool pair = false) {
    const int d = e->d, F = e->F;
    MoeArgs m;
    m.x = e->h; m.rows = M; m.rows_per_sample = N; m.d = d; m.E = e->E;
    // tile bound of THIS call's row count, not of the engine's capacity: a capacity-sized launch would be mostly padding
    // tiles, and the XCD-contiguous tile order would put every real tile on XCD 0 (measured: 255 us instead of 60 us per
    // expert GEMM at 512 rows in an engine sized for 8192)
    const int tiles = (int)((2 * (size_t)M + (size_t)e->E * 255 + 255) / 256);
    const bool hoisted = branch == 0 && e->moe_tp_live;  // this layer's time plan was written at the top of the evaluation (run_forward)
    m.sel = e->moe_sel; m.wts = e->moe_wts; m.pos = e->moe_pos; m.tile_expert = e->moe_tile_expert; m.max_tiles = tiles;
    m.src = e->moe_src;
    if (hoisted) {
        m.sel = e->moe_tp_sel + (size_t)layer * e->moe_tp_stride_rows; m.pos = e->moe_tp_pos + (size_t)layer * e->moe_tp_stride_rows;
        m.wts = e->moe_tp_wts + (size_t)layer * e->moe_tp_stride_rows; m.src = e->moe_tp_src + (size_t)layer * e->moe_tp_stride_src;
        m.tile_expert = e->moe_tp_tile_expert + (size_t)layer * e->moe_tiles;
    }
    m.gate_w = nullptr; m.sample_logits = nullptr; m.forced = nullptr;
    {
        ProfScope ps(e, 2, 0, s);
        if (branch == 0) {  // gate(cond) with cond = t_embedder(t) (models2.py:462, :950-951): computed for all layers at once, run_forward
            m.sample_logits = e->moe_logits + (size_t)layer * e->E;
            m.sample_ld = e->L * e->E;
        } else {
            m.gate_w = w.gate_s;
        }
        const size_t slot = ((size_t)layer * 2 + branch) * (size_t)e->cfg.max_batch * e->cfg.max_tokens * 2;
        if (e->moe_force_rows) {
            LT_REQUIRE(e->moe_force_rows == M, "forced MoE routing was given for %d rows, this call has %d", e->moe_force_rows, M);
            m.forced = e->moe_force + slot;
        }
        if (branch != 0 && !routed && launch_moe_route(m, s)) return 1;  // (the time branch routes inside the plan kernel)
        if (!hoisted && launch_moe_plan(m, s)) return 1;
        if (e->moe_rec_on) {
            LT_CHECK_HIP(hipMemcpyAsync(e->moe_rec + slot, m.sel, (size_t)M * 2 * sizeof(int), hipMemcpyDeviceToDevice, s));
            e->moe_rec_rows = M;
        }
    }
    const int P = tiles * 256;
    GemmArgs g;
    g.bias = nullptr; g.bias_dtype = -1; g.tile_expert = m.tile_expert;
    {   // grouped SwiGLU GEMM: each 256-row tile multiplies with its expert's packed w1|w3
        // (A = the un-sorted FFN input: the GEMM gathers its rows through the plan's inverse map, no expert-sorted copy)
        g.A = e->h; g.a_row_map = m.src; g.a_map_rows = M; g.W = branch == 0 ? w.w13_t : w.w13_s; g.C = e->moe_us; g.M = P; g.N = 2 * F; g.K = d;
        g.lda = d; g.ldw = d; g.ldc = F; g.w_expert_stride = (long long)2 * F * d;
        g.pair_ab = pair ? 2 : 0; g.pair_c = pair;  // (A is gathered row by row from e->h: row-major)
        ProfScope ps(e, 0, 2.0 * (2.0 * M) * (2.0 * F) * d, s, true);  // algorithmic: every token visits two experts
        if (launch_gemm_bf16(g, 1, 0, s, ps.ev0(), ps.ev1())) return 1;
    }
    {
        g.A = e->moe_us; g.a_row_map = nullptr; g.a_map_rows = 0; g.W = branch == 0 ? w.w2_t : w.w2_s; g.C = e->moe_ys; g.M = P; g.N = d; g.K = F;
        if (lt_opt(OPT_GEMM_TAIL_SPLIT) && e->tail_part) { g.tail_part = e->tail_part; g.tail_cnt = e->tail_cnt; g.tail_cap_parts = e->tail_cap_parts; g.tail_max_parts = lt_opt(OPT_GEMM_TAIL_SPLIT) == 2 ? 2 : 4; }
        g.lda = F; g.ldw = F; g.ldc = d; g.w_expert_stride = (long long)d * F;
        g.pair_ab = pair ? 3 : 0; g.pair_c = 0;
        ProfScope ps(e, 0, 2.0 * (2.0 * M) * (double)d * F, s, true);
        if (launch_gemm_bf16(g, 0, 0, s, ps.ev0(), ps.ev1())) return 1;
    }
    // (the top-2 combine of e->moe_ys is formed by the gated_residual_norm launch that consumes this branch: moe_y() below)
    return 0;
}

// the branch output of a MoE FFN as gated_residual_norm takes it: experts' outputs + routing, combined on the way in
// layer >= 0: the branch is the time router's and its plan may be the hoisted one of that layer
void moe_y(const lt_engine* e, GatedResArgs& g, int time_layer = -1) {
    g.y = nullptr; g.moe_ys = e->moe_ys; g.moe_pos = e->moe_pos; g.moe_wts = e->moe_wts;
    if (time_layer >= 0 && e->moe_tp_live) {
        g.moe_pos = e->moe_tp_pos + (size_t)time_layer * e->moe_tp_stride_rows;
        g.moe_wts = e->moe_tp_wts + (size_t)time_layer * e->moe_tp_stride_rows;
    }
}

// one forward pass of the configured family [+ CFG combine]:
//   NextDiT.forward / forward_with_cfg   lumina_next_t2i/models/model.py:836-913
//   DiT_Llama (ImageNet)                 Next-DiT-ImageNet/models/models.py:920-974
//   DiT_Llama (Flag-DiT)                 lumina_t2i/models/model.py:829-922
//
// Packed variable-resolution batch (NextDiT.patchify_and_embed list branch, model.py:789-834): sample b is its own
// [C, H_b, W_b] tensor; sequences are padded to the longest one with `pad_token`, padded positions rotate like the sample's
// last token, are masked as KEYS (x_mask, :407-415) and dropped again by unpatchify (:757-768).  Only the plain forward of
// the text-conditional Next-DiT has this path in the reference.
struct PackedDesc {
    const void* const* x_ptrs;  // [B] device pointers
    void* const* out_ptrs;      // [B] device pointers, each [in_channels, H_b, W_b]
    const int32_t* hw;          // [B][2] latent (H_b, W_b), host
};

// all four GEMM weights of every dense block -> the row-pair-interleaved layout (want) or back to row-major, in place, on stream s
int ensure_weight_layout(lt_engine* e, bool want, hipStream_t s) {
    if (e->w_pair == want) return 0;
    for (int l = 0; l < e->L; ++l) {
        LayerW& w = e->lw[l];
        if (e->E == 0) {
            if (launch_pair_layout(w.wqkv, e->qkvn, e->d, want, s)) return 1;
            if (launch_pair_layout(w.wo, e->d, e->d, want, s)) return 1;
            if (launch_pair_layout(w.w13, 2LL * e->F, e->d, want, s)) return 1;
            if (launch_pair_layout(w.w2, e->d, e->F, want, s)) return 1;
        } else {  // MoE families: the experts' weights only ([E][2 F][d] / [E][d][F]: an expert's rows are an even count, pairs stay inside it)
            for (u16* w13 : {w.w13_t, w.w13_s}) if (w13 && launch_pair_layout(w13, 2LL * e->E * e->F, e->d, want, s)) return 1;
            for (u16* w2 : {w.w2_t, w.w2_s}) if (w2 && launch_pair_layout(w2, (long long)e->E * e->d, e->F, want, s)) return 1;
        }
    }
    e->w_pair = want;
    return 0;
}

int run_forward(lt_engine* e, const void* x_in, const float* t_dev, void* out, const lt_step_args* a, int use_cfg,
                hipStream_t s, const PackedDesc* pk = nullptr) {
    const lt_config& c = e->cfg;
    const VariantDesc& v = e->v;
    const int B = a->batch, p = c.patch_size;
    if (pk) {
        LT_REQUIRE(c.variant == LT_VARIANT_NEXT_T2I && !use_cfg, "packed batches: plain forward of the text-conditional Next-DiT only");
        LT_REQUIRE(B >= 1 && B <= c.max_batch && B <= 64, "packed batch of %d exceeds max_batch %d (or 64)", B, c.max_batch);
    }
    LT_REQUIRE(B >= 1 && B <= c.max_batch, "batch %d exceeds max_batch %d", B, c.max_batch);
    LT_REQUIRE(!use_cfg || B % 2 == 0, "forward_with_cfg needs an even batch (cond+uncond)");
    LT_REQUIRE(a->latent_h % p == 0 && a->latent_w % p == 0, "latent %dx%d not divisible by patch", a->latent_h, a->latent_w);
    int Hp = a->latent_h / p, Wp = a->latent_w / p;
    const int Wrow = v.eol ? Wp + 1 : Wp;  // tokens per latent row (Flag-DiT appends one eol token, model.py:779-786)
    int N = Hp * Wrow;
    int pk_ntok[64], pk_gw[64];
    if (pk) {  // N = the longest sequence; a->latent_h / latent_w are ignored
        N = 0; Hp = 0; Wp = 0;
        for (int b = 0; b < B; ++b) {
            const int hb = pk->hw[2 * b], wb = pk->hw[2 * b + 1];
            LT_REQUIRE(hb > 0 && wb > 0 && hb % p == 0 && wb % p == 0, "packed sample %d: latent %dx%d not divisible by patch", b, hb, wb);
            pk_ntok[b] = (hb / p) * (wb / p);
            pk_gw[b] = wb / p;
            N = std::max(N, pk_ntok[b]); Hp = std::max(Hp, hb / p); Wp = std::max(Wp, wb / p);
        }
    }
    const int M = B * N;
    LT_REQUIRE(N <= c.max_tokens, "%d latent tokens exceed max_tokens %d", N, c.max_tokens);
    if (v.rope_1d) LT_REQUIRE(N <= e->rope_len, "sequence of %d tokens exceeds the 1-D RoPE table (%d)", N, e->rope_len);
    else LT_REQUIRE(Hp <= e->rope_len && Wp <= e->rope_len, "latent grid exceeds the RoPE table (%d)", e->rope_len);
    LT_REQUIRE(a->io_dtype == LT_BF16 || a->io_dtype == LT_F32, "io_dtype must be bf16 or f32");
    LT_REQUIRE(e->prompt_B == B, "%s was called for batch %d, step has batch %d", v.labels ? "lt_prepare_labels" : "lt_prepare_prompt",
               e->prompt_B, B);
    LT_REQUIRE(e->reg_Y == 0 || (B == 2 && !pk), "regional captions: one image per call (batch 2 = cond + uncond row), tensor input");
    if (!e->weights_ok && lt_weights_ready(e)) return 2;
    const int d = e->d, L = e->L, H = e->H, Hkv = e->Hkv, hd = e->hd, F = e->F, dkv = e->dkv, A = e->A;
    const int Npad = round_up(N, 64);
    const int cd = e->chunks * d;

    if (ensure_rope(e, a, s)) return 1;
    float sm_scale;
    if (a->proportional_attn && !v.labels) {
        LT_REQUIRE(a->base_seqlen > 1, "proportional_attn needs base_seqlen");
        // math.sqrt(math.log(seqlen, base_seqlen) / head_dim)  (model.py:374; seqlen counts the eol tokens for Flag-DiT)
        sm_scale = (float)std::sqrt(std::log((double)N) / std::log((double)a->base_seqlen) / (double)hd);
    } else {
        sm_scale = (float)std::sqrt(1.0 / (double)hd);  // model.py:376; flash_attn_func default (models.py:389)
    }

    // patchify + x_embedder (model.py:777-779) [+ eol token per row]
    const int *ntok_dev = nullptr, *gw_dev = nullptr;
    if (!pk) {
        ProfScope ps(e, 2, 0, s);
        if (launch_patchify(x_in, a->io_dtype, e->patches, B, c.in_channels, a->latent_h, a->latent_w, p, e->kpad, use_cfg, Wrow, s)) return 1;
    } else {
        ProfScope ps(e, 2, 0, s);
        for (int b = 0; b < B; ++b)
            if (launch_patchify(pk->x_ptrs[b], a->io_dtype, e->patches + (size_t)b * N * e->kpad, 1, c.in_channels, pk->hw[2 * b],
                                pk->hw[2 * b + 1], p, e->kpad, 0, 0, s)) return 1;
        // per-sample token count / grid width for the rotary positions and the key mask
        memcpy(e->pk_host, pk_ntok, B * sizeof(int));
        memcpy(e->pk_host + 64, pk_gw, B * sizeof(int));
        LT_CHECK_HIP(hipMemcpyAsync(e->pk_dev, e->pk_host, 128 * sizeof(int), hipMemcpyHostToDevice, s));
        ntok_dev = e->pk_dev; gw_dev = e->pk_dev + 64;
    }
    if (gemm(e, e->patches, e->kpad, e->xemb_w, e->kpad, e->x, d, M, d, e->kpad, e->xemb_b, 0, s)) return 1;
    if (pk) {  // padded positions hold the learned pad_token, not an embedded patch (model.py:811-817)
        ProfScope ps(e, 2, 0, s);
        for (int b = 0; b < B; ++b)
            if (pk_ntok[b] < N && launch_fill_rows_bf16(e->x + ((size_t)b * N + pk_ntok[b]) * d, e->pad_token, N - pk_ntok[b], d, s)) return 1;
    }
    if (v.eol) {
        ProfScope ps(e, 2, 0, s);
        if (launch_eol_fill(e->x, e->eol_token, B * Hp, Wp, d, s)) return 1;
    }
    // conditioning: t_embedder (model.py:84-87) + caption / label embedding (hoisted) -> adaLN vectors of every layer + final
    {
        ProfScope ps(e, 2, 0, s);
        const int fused_pro = lt_opt(OPT_PROLOGUE_FUSED);  // round 6, bit mask: 1 timestep features, 2 temb + embedding, 4 prep_mod - each one launch less, bit-identical (LinearSmallMExtra)
        if (fused_pro & 1) {
            LinearSmallMExtra x;
            x.t = t_dev;
            if (launch_linear_small_m_ext(nullptr, e->t0_w, e->t0_b, e->t1, B, A, 256, 0, x, s)) return 1;
        } else {
            if (launch_timestep_features(t_dev, 0, e->tfeat, B, 256, s)) return 1;
            if (launch_linear_small_m(e->tfeat, e->t0_w, e->t0_b, e->t1, B, A, 256, 0, s)) return 1;
        }
        if (launch_linear_small_m(e->t1, e->t2_w, e->t2_b, e->temb, B, A, A, 1, s)) return 1;
        if (e->gate_t_all && launch_linear_small_m(e->temb, e->gate_t_all, nullptr, e->moe_logits, B, L * e->E, A, 0, s)) return 1;  // every layer's time-router logits
        e->moe_tp_live = e->gate_t_all && e->moe_tp_sel && lt_opt(OPT_MOE_TIME_PLAN_HOIST) && !e->moe_force_rows && B <= LT_MOE_PLAN_TIME_MAX_SAMPLES;
        if (e->moe_tp_live) {  // ... and every layer's time plan: they depend on nothing else (one launch instead of L)
            MoeArgs m;
            m.x = nullptr; m.gate_w = nullptr; m.forced = nullptr; m.sample_logits = e->moe_logits; m.sample_ld = L * e->E;
            m.rows = M; m.rows_per_sample = N; m.d = d; m.E = e->E;
            m.sel = e->moe_tp_sel; m.pos = e->moe_tp_pos; m.wts = e->moe_tp_wts; m.src = e->moe_tp_src; m.tile_expert = e->moe_tp_tile_expert;
            m.max_tiles = (int)((2 * (size_t)M + (size_t)e->E * 255 + 255) / 256);  // = moe_ffn's bound for this call's rows
            m.layers = L; m.layer_stride_rows = (long long)e->moe_tp_stride_rows; m.layer_stride_src = (long long)e->moe_tp_stride_src;
            m.layer_stride_tiles = e->moe_tiles;
            if (launch_moe_plan(m, s)) return 1;
        }
        // once per (sample, channel) instead of once per token inside the row kernels: tanh of the gate chunks (where the
        // family has it) and bf16(1 + scale) of every scale chunk
        unsigned tanh_mask = 0, scale_mask = 0;
        for (int i = 0; i < 2; ++i) {
            if (v.gate_tanh && v.i_gate[i] >= 0) tanh_mask |= 1u << v.i_gate[i];
            if (v.i_scale[i] >= 0) scale_mask |= 1u << v.i_scale[i];
        }
        if (c.variant == LT_VARIANT_NEXT_MOE) { tanh_mask |= 1u << 5; scale_mask |= 1u << 4; }  // the space branch's chunks
        if (fused_pro & 6) {
            LinearSmallMExtra x;
            const u16* in = e->temb;
            if (fused_pro & 2) x.a2 = e->cap_emb;
            else { if (launch_add_bf16(e->temb, e->cap_emb, e->adaln_in, (long long)B * A, s)) return 1; in = e->adaln_in; }
            if (fused_pro & 4) { x.pm_L = L; x.pm_chunks = e->chunks; x.pm_d = e->d; x.pm_final = v.final_chunks == 2 ? 1 : 0; x.pm_tanh = tanh_mask; x.pm_scale = scale_mask; }
            if (launch_linear_small_m_ext(in, e->adaln_w, e->adaln_b, e->mod, B, e->ld_mod, A, 1, x, s)) return 1;
            if (!(fused_pro & 4) && launch_prep_mod(e->mod, B, e->ld_mod, L, e->chunks, e->d, tanh_mask, scale_mask, v.final_chunks == 2 ? 1 : 0, s)) return 1;
        } else {
            if (launch_add_bf16(e->temb, e->cap_emb, e->adaln_in, (long long)B * A, s)) return 1;
            if (launch_linear_small_m(e->adaln_in, e->adaln_w, e->adaln_b, e->mod, B, e->ld_mod, A, 1, s)) return 1;
            if (launch_prep_mod(e->mod, B, e->ld_mod, L, e->chunks, e->d, tanh_mask, scale_mask, v.final_chunks == 2 ? 1 : 0, s)) return 1;
        }
    }
    auto chunk = [&](int layer, int idx) -> const u16* { return idx < 0 ? nullptr : e->mod + (size_t)layer * cd + (size_t)idx * d; };
    // round 6, option pair_layout: when every GEMM of the dense block runs on the persistent kernel (>= one tile per CU: the fused QKV launch,
    // O, W1 | W3, W2) and the attention on a one-wave kernel, their A operands (h, the attention output, the SwiGLU output) and the four
    // weights are kept in the row-pair-interleaved layout (GemmArgs::pair_ab): the GEMM's LDS-DMA stream then asks the L2 for whole 128-byte
    // lines - half the requests.  Same products in the same order: bit-identical to the row-major path.
    bool pair = false;
    if (lt_opt(OPT_PAIR_LAYOUT) && e->E == 0 && !pk && M % 2 == 0 && d % 32 == 0 && F % 32 == 0 && F <= 16384 && d <= 16384 && !(v.text && e->reg_Y > 0) &&
        (!v.text || attention_fuses_text(hd))) {
        GemmArgs gq;
        gq.A = e->h; gq.W = e->lw[0].wqkv; gq.C = e->qkv; gq.bias = nullptr; gq.bias_dtype = -1; gq.M = M; gq.N = d + 2 * dkv; gq.K = d;
        gq.lda = d; gq.ldw = d; gq.ldc = e->qkvn; gq.VT = e->vt; gq.vt_split = d + dkv; gq.vt_tokens = N; gq.vt_hd = hd; gq.vt_npad = Npad;
        const bool vt_epi0 = lt_opt(OPT_QKV_VT_EPILOGUE) && N % 64 == 0 && (long long)((M + 255) / 256) * ((dkv + 255) / 256) >= 128;
        GemmArgs go, g13, g2;
        go.A = e->attn; go.W = e->lw[0].wo; go.C = e->o; go.bias = nullptr; go.bias_dtype = -1; go.M = M; go.N = d; go.K = d; go.lda = d; go.ldw = d; go.ldc = d;
        g13 = go; g13.A = e->h; g13.W = e->lw[0].w13; g13.C = e->u; g13.N = 2 * F; g13.ldc = F;
        g2 = go; g2.A = e->u; g2.W = e->lw[0].w2; g2.K = F; g2.lda = F; g2.ldw = F;
        AttnArgs at0;
        at0.q = e->q; at0.k = e->k; at0.vt = e->vt; at0.bias = nullptr; at0.out = e->attn; at0.gate = nullptr; at0.accumulate = 0;
        at0.B = B; at0.H = H; at0.Hkv = Hkv; at0.N = N; at0.Nk = N; at0.Nkpad = Npad; at0.hd = hd; at0.scale = sm_scale; at0.k_prescaled = 1;
        at0.nk_batch = ntok_dev;
        if (v.text) { at0.tk = e->lw[0].ky; at0.tvt = e->lw[0].vty; at0.tbias = e->txt_bias; at0.tgate = e->lw[0].gate; at0.Tk = e->prompt_T; at0.Tkpad = e->prompt_Tpad; }
        pair = vt_epi0 && lt_opt(OPT_QKV_FUSED_GEMM) && gemm_qkv_fusable(gq) && gemm_runs_w4q_dense(go, 0) && gemm_runs_w4q_dense(g13, 1) &&
               gemm_runs_w4q_dense(g2, 0) && attention_is_one_wave(at0);
    }
    // the MoE families: the experts' grouped GEMMs on the persistent kernel (>= 1.5 tiles per CU: the 600M MoE at 1024^2) read their weights in the
    // pair layout, and the SwiGLU output between the two (e->moe_us) is written / read in it; the W1 | W3 launch gathers its A rows one by one
    // from e->h, which stays row-major (a gathered row's line mate is not its neighbour in the tile)
    bool pair_moe = false;
    if (lt_opt(OPT_PAIR_LAYOUT) && e->E > 0 && d % 32 == 0 && F % 32 == 0 && F <= 16384 && d <= 16384) {
        const int tiles = (int)((2 * (size_t)M + (size_t)e->E * 255 + 255) / 256);
        GemmArgs g13, g2;
        g13.bias = nullptr; g13.bias_dtype = -1; g13.tile_expert = e->moe_tile_expert;
        g13.A = e->h; g13.a_row_map = e->moe_src; g13.a_map_rows = M; g13.W = e->lw[0].w13_t ? e->lw[0].w13_t : e->lw[0].w13_s; g13.C = e->moe_us;
        g13.M = tiles * 256; g13.N = 2 * F; g13.K = d; g13.lda = d; g13.ldw = d; g13.ldc = F; g13.w_expert_stride = (long long)2 * F * d;
        g2 = g13;
        g2.A = e->moe_us; g2.a_row_map = nullptr; g2.a_map_rows = 0; g2.W = e->lw[0].w2_t ? e->lw[0].w2_t : e->lw[0].w2_s; g2.C = e->moe_ys;
        g2.N = d; g2.K = F; g2.lda = F; g2.ldw = F; g2.ldc = d; g2.w_expert_stride = (long long)d * F;
        pair_moe = gemm_runs_w4q_grouped(g13, 1) && gemm_runs_w4q_grouped(g2, 0);
    }
    const bool pair_any = pair || pair_moe;
    if (e->w_pair != pair_any) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
        if (cs == hipStreamCaptureStatusNone) {
            if (ensure_weight_layout(e, pair_any, s)) return 1;
        } else {
            // inside a capture (the caller's own graph) a layout change would be baked into every replay: run on the layout the weights are in
            LT_REQUIRE(!e->w_pair, "this evaluation runs on the small-M kernels, which read row-major weights, but the engine's weights are in the "
                                   "pair layout of the last large evaluation and the stream is capturing: run one eager evaluation of this shape first");
            pair = false; pair_moe = false;
        }
    }
    e->last_pair = pair || pair_moe;
    // first pre-norm: modulate(attention_norm(x), [shift,] scale) (model.py:599 / models.py:785 / lumina_t2i model.py:600)
    {
        ProfScope ps(e, 2, 0, s);
        NormModArgs n;
        n.x = e->x; n.w = v.pre_w ? e->lw[0].attn_norm1 : nullptr; n.scale = chunk(0, v.i_scale[0]); n.shift = chunk(0, v.i_shift[0]);
        n.out = e->h; n.rows = M; n.rows_per_batch = N; n.d = d; n.ld_mod = e->ld_mod; n.eps = c.norm_eps; n.scale_pre = 1;
        n.out_pair = pair;
        if (launch_rmsnorm_mod(n, s)) return 1;
    }
    const int post_mode = v.post ? 1 : 0, gate_mode = 0;  // gates arrive ready (tanh applied above where the family has it)
    for (int l = 0; l < L; ++l) {
        LayerW& w = e->lw[l];
        // V^T epilogue path: q | k columns in one GEMM, the V columns in a second one that writes e->vt directly.  Large problems
        // only (a second launch of a latency-bound 512-row GEMM costs more than the transpose), whole 64-key tiles per sample
        // (no key padding to zero), and not for packed batches (their padded rows must read as zero keys)
        const bool vt_epi = lt_opt(OPT_QKV_VT_EPILOGUE) && !pk && N % 64 == 0 && (long long)((M + 255) / 256) * ((dkv + 255) / 256) >= 128;
        // ... and ONE launch for all three when the shapes are whole tiles of the persistent 256 x 288 kernel (lt_set_option
        // "qkv_fused_gemm"): Q | K tiles with the plain epilogue, V tiles with swapped MFMA operands and the V^T epilogue
        GemmArgs gq;
        gq.A = e->h; gq.W = w.wqkv; gq.C = e->qkv; gq.bias = nullptr; gq.bias_dtype = -1; gq.M = M; gq.N = d + 2 * dkv; gq.K = d;
        gq.lda = d; gq.ldw = d; gq.ldc = e->qkvn; gq.VT = e->vt; gq.vt_split = d + dkv; gq.vt_tokens = N; gq.vt_hd = hd; gq.vt_npad = Npad;
        // round 4: with the fused launch the Q columns' LayerNorm partial sums leave the GEMM epilogue and the attention kernel's prologue
        // does q_norm + RoPE itself (AttnArgs::q_raw): qk_norm_rope then runs for K only.  Conditions: the hd-72 one-wave kernel takes
        // the call (whole tiles, no packed batch), 2-D RoPE, qk_norm on, text keys fused into the launch or absent, not regional.
        const bool regional = v.text && e->reg_Y > 0;
        const bool fuse_text = v.text && !regional && attention_fuses_text(hd);
        AttnArgs at;
        at.q = e->q; at.k = e->k; at.vt = e->vt; at.bias = nullptr; at.out = e->attn; at.gate = nullptr; at.accumulate = 0;
        at.B = B; at.H = H; at.Hkv = Hkv; at.N = N; at.Nk = N; at.Nkpad = Npad; at.hd = hd; at.scale = sm_scale;
        at.k_prescaled = 1;
        at.nk_batch = ntok_dev;
        at.tail_ws = e->attn_tail_ws; at.tail_ws_bytes = e->attn_tail_ws_bytes;
        if (fuse_text) {  // zero-init gated text cross-attention (model.py:420-434) inside the same launch
            at.tk = w.ky; at.tvt = w.vty; at.tbias = e->txt_bias; at.tgate = w.gate; at.Tk = e->prompt_T; at.Tkpad = e->prompt_Tpad;
        }
        bool raw_q = false;
        // round 5, the class-conditional 600M models at <= 512 tokens: q_norm, k_norm, RoPE, the V transpose and the attention are ONE
        // launch behind the small-M QKV GEMM, whose epilogue leaves the LayerNorm partials (AttnSmallArgs / GemmArgs::rowstat)
        gq.VT = nullptr;
        const bool small_fused = lt_opt(OPT_ATTN_SMALL_FUSED) && !v.text && c.qk_norm && !v.rope_1d && !pk && !vt_epi &&
                                 attention_small_fusable(hd, N, H, Hkv, d, dkv) && gemm_is_small_m(gq, 0);
        gq.VT = e->vt;
        if (small_fused) {
            gq.VT = nullptr; gq.rowstat = e->qstat; gq.rowstat_slots = e->qstat_slots;
            {
                ProfScope ps(e, 0, 2.0 * M * (double)(d + 2 * dkv) * d, s, true);
                if (launch_gemm_bf16(gq, 0, 0, s, ps.ev0(), ps.ev1())) return 1;
            }
            AttnSmallArgs as;
            as.qkv = e->qkv; as.ld = e->qkvn; as.q_col0 = 0; as.k_col0 = d; as.v_col0 = d + dkv;
            as.rowstat = e->qstat; as.slots = e->qstat_slots; as.q_slot0 = 0; as.q_nslot = d / 128; as.k_slot0 = d / 128; as.k_nslot = dkv / 128;
            as.q_ln_w = w.q_norm_w; as.q_ln_b = w.q_norm_b; as.k_ln_w = w.k_norm_w; as.k_ln_b = w.k_norm_b; as.ln_eps = 1e-5f;
            as.cs = e->rope; as.t = t_dev; as.watershed = 0.f; as.cs_len = e->rope_len; as.grid_w = Wp;
            as.k_scale = sm_scale * LOG2E; as.out = e->attn; as.B = B; as.H = H; as.Hkv = Hkv; as.N = N; as.hd = hd;
            prefetch_rider(e, &as.pf, e->attn, d, w.wo, d, e->o, d, M, d, d, 0);
            ProfScope ps(e, 1, 4.0 * B * H * (double)N * N * hd, s);
            if (launch_attention_small(as, s)) return 1;
        } else
        if (vt_epi && lt_opt(OPT_QKV_FUSED_GEMM) && gemm_qkv_fusable(gq)) {
            const int bn = gemm_qkv_tile_width(gq);
            raw_q = lt_opt(OPT_ATTN_Q_FUSED) && c.qk_norm && !v.rope_1d && !pk && !regional && (fuse_text || !v.text) && attention_takes_raw_q(at) &&
                    bn > 0 && d % bn == 0 && 2 * d / bn <= 32;
            if (raw_q) { gq.qstat = e->qstat; gq.qstat_cols = d; gq.qstat_slots = 2 * d / bn; }
            gq.pair_ab = pair ? 3 : 0;
            ProfScope ps(e, 0, 2.0 * M * (double)(d + 2 * dkv) * d, s, true);
            if (launch_gemm_bf16(gq, 3, 0, s, ps.ev0(), ps.ev1())) return 1;
        } else if (vt_epi) {
            if (gemm(e, e->h, d, w.wqkv, d, e->qkv, e->qkvn, M, d + dkv, d, nullptr, 0, s)) return 1;
            GemmArgs g;
            g.A = e->h; g.W = w.wqkv + (size_t)(d + dkv) * d; g.C = e->vt; g.bias = nullptr; g.bias_dtype = -1;
            g.M = M; g.N = dkv; g.K = d; g.lda = d; g.ldw = d; g.ldc = 0; g.vt_tokens = N; g.vt_hd = hd; g.vt_npad = Npad;
            ProfScope ps(e, 0, 2.0 * M * (double)dkv * d, s, true);
            if (launch_gemm_bf16(g, 2, 0, s, ps.ev0(), ps.ev1())) return 1;
        } else if (gemm(e, e->h, d, w.wqkv, d, e->qkv, e->qkvn, M, e->qkvn, d, nullptr, 0, s)) return 1;
        if (!small_fused) {
            ProfScope ps(e, 2, 0, s);
            QkPostArgs qa;
            qa.src = e->qkv; qa.ld_src = e->qkvn; qa.B = B; qa.N = N; qa.hd = hd; qa.rope_mode = v.rope_1d ? 2 : 1;
            qa.cs = e->rope; qa.t = t_dev; qa.grid_w = Wp; qa.cs_len = e->rope_len; qa.ln_eps = 1e-5f;
            qa.n_tok_b = ntok_dev; qa.grid_w_b = gw_dev;
            qa.watershed = c.variant == LT_VARIANT_NEXT_T2I ? a->scale_watershed : 0.f;  // other families: one table (branch 1)
            QkvPostArgs pa;
            qa.col0 = 0; qa.heads = H; qa.dst = e->q;
            qa.ln_w = c.qk_norm ? w.q_norm_w : nullptr; qa.ln_b = c.qk_norm ? w.q_norm_b : nullptr;
            pa.q = qa;
            qa.col0 = d; qa.heads = Hkv; qa.dst = e->k;
            qa.ln_w = c.qk_norm ? w.k_norm_w : nullptr; qa.ln_b = c.qk_norm ? w.k_norm_b : nullptr;
            qa.out_scale = sm_scale * LOG2E;  // softmax scale folded into K's one bf16 rounding (scores in log2 units)
            pa.k = qa;
            pa.v_src = e->qkv; pa.v_dst = e->vt; pa.v_ld_src = e->qkvn; pa.v_col0 = d + dkv; pa.v_B = B; pa.v_N = N;
            pa.v_Npad = Npad; pa.v_kv_heads = Hkv; pa.v_hd = hd;
            if (raw_q) {
                // K only: the queries are normalised and rotated by the attention prologue; the K pass reduces their LayerNorm partials
                pa.k.qstat_in = e->qstat; pa.k.qstat_out = e->qmr; pa.k.qstat_slots = gq.qstat_slots; pa.k.qstat_width = d;
                if (launch_qk_norm_rope(pa.k, s)) return 1;
                at.q = nullptr; at.q_raw = e->qkv; at.q_ld = e->qkvn; at.q_col0 = 0; at.q_stat = e->qmr;
                at.q_ln_w = w.q_norm_w; at.q_ln_b = w.q_norm_b;
                at.rope_cs = e->rope; at.rope_cs_t = e->rope_tr; at.rope_t = t_dev; at.rope_watershed = pa.q.watershed;
                at.rope_cs_len = e->rope_len; at.rope_grid_w = Wp;
            } else if (!vt_epi && (lt_opt(OPT_QKV_POST_FUSED) == 1 || (lt_opt(OPT_QKV_POST_FUSED) == 2 && M < 2048))) {
                if (!regional && (fuse_text || !v.text)) prefetch_rider(e, &pa.pf, e->attn, d, w.wo, d, e->o, d, M, d, d, 0);
                if (launch_qkv_post(pa, s)) return 1;  // q, k post-processing and the V transpose in one launch
            } else if (lt_opt(OPT_QK_POST_PAIR) && M >= 2048) {
                if (launch_qk_norm_rope_pair(pa.q, pa.k, s)) return 1;  // q and k in one persistent launch
                if (!vt_epi && launch_v_transpose(e->qkv, e->qkvn, d + dkv, e->vt, B, N, Npad, Hkv, hd, s)) return 1;
            } else {
                if (launch_qk_norm_rope(pa.q, s)) return 1;
                if (launch_qk_norm_rope(pa.k, s)) return 1;
                if (!vt_epi && launch_v_transpose(e->qkv, e->qkvn, d + dkv, e->vt, B, N, Npad, Hkv, hd, s)) return 1;
            }
        }
        LT_REQUIRE(!pair || (!small_fused && vt_epi && gq.pair_ab && !regional && (fuse_text || !v.text)), "pair layout: the block left the path it was chosen for");
        at.out_pair = pair;
        if (!small_fused && attention(e, at, s)) return 1;
        if (regional) {
            // compositional Next-DiT (lumina_next_compositional_generation/models/model.py:422-446): every caption attends the
            // queries of its row (regional captions -> cond row 0, last caption -> uncond row 1) into its own buffer, then one
            // pass applies region masks, tanh(gate), the caption sum and the residual add with the reference's rounding points
            AttnArgs rt = at;
            rt.B = e->reg_Y; rt.q_batch_map = e->reg_qmap; rt.k = w.ky; rt.vt = w.vty; rt.bias = e->txt_bias; rt.gate = nullptr;
            rt.accumulate = 0; rt.out = e->reg_txt; rt.nk_batch = nullptr;
            rt.Nk = e->prompt_T; rt.Nkpad = e->prompt_Tpad; rt.scale = (float)(1.0 / std::sqrt((double)hd));
            if (attention(e, rt, s)) return 1;
            ProfScope ps(e, 2, 0, s);
            if (launch_region_text_combine(e->attn, e->reg_txt, w.gate, e->reg_Y, N, H, hd, Hp, Wp, e->reg_h, e->reg_w, s)) return 1;
        } else if (v.text && !fuse_text) {
            at.k = w.ky; at.vt = w.vty; at.bias = e->txt_bias; at.gate = w.gate; at.accumulate = 1; at.nk_batch = nullptr;
            at.Nk = e->prompt_T; at.Nkpad = e->prompt_Tpad; at.scale = (float)(1.0 / std::sqrt((double)hd));
            if (attention(e, at, s)) return 1;
        }
        int ys_o = 0, ys_f = 0;
        if (gemm(e, e->attn, d, w.wo, d, e->o, d, M, d, d, nullptr, 0, s, &ys_o, pair ? 3 : 0)) return 1;
        {   // x += gate' * post(attn) ; h = pre_ffn(x) * (1 + scale) [+ shift]
            ProfScope ps(e, 2, 0, s);
            GatedResArgs g;
            if (ys_o) { g.ystat = e->ystat; g.ystat_slots = ys_o; }
            g.x = e->x; g.y = e->o; g.post_w = v.post ? w.attn_norm2 : nullptr; g.gate = chunk(l, v.i_gate[0]);
            g.post_mode = post_mode; g.gate_mode = gate_mode;
            g.next_w = v.pre_w ? w.ffn_norm1 : nullptr; g.next_scale = chunk(l, v.i_scale[1]); g.next_shift = chunk(l, v.i_shift[1]);
            g.next_mode = 1; g.h = e->h; g.h_pair = pair;
            g.rows = M; g.rows_per_batch = N; g.d = d; g.ld_mod = e->ld_mod; g.eps = c.norm_eps; g.eps_next = 1e-6f; g.scale_pre = 1;
            if (e->E == 0) prefetch_rider(e, &g.pf, e->h, d, w.w13, d, e->u, F, M, 2 * F, d, 1);
            if (launch_gated_residual_norm(g, s)) return 1;
        }
        const u16 *last_post_w, *last_gate;
        if (e->E == 0) {
            if (gemm(e, e->h, d, w.w13, d, e->u, F, M, 2 * F, d, nullptr, 1, s, nullptr, pair ? 3 : 0, pair)) return 1;
            if (gemm(e, e->u, F, w.w2, F, e->o, d, M, d, F, nullptr, 0, s, &ys_f, pair ? 3 : 0)) return 1;
            last_post_w = v.post ? w.ffn_norm2 : nullptr;
            last_gate = chunk(l, v.i_gate[1]);
        } else if (e->moe_mode != 0) {  // one MoE FFN in the ImageNet block (models.py:755-758: time-routed; models1.py: per token)
            if (moe_ffn(e, w, l, e->moe_mode == 1 ? 0 : 1, M, N, B, s, false, pair_moe)) return 1;
            last_post_w = w.ffn_norm2;
            last_gate = chunk(l, v.i_gate[1]);
        } else {  // time MoE -> residual -> space MoE (models2.py:793-800)
            if (moe_ffn(e, w, l, 0, M, N, B, s, false, pair_moe)) return 1;
            {
                ProfScope ps(e, 2, 0, s);
                GatedResArgs g;
                g.x = e->x; moe_y(e, g, l); g.post_w = w.norm_time; g.gate = chunk(l, 3); g.post_mode = 1; g.gate_mode = 0;
                g.next_w = nullptr; g.next_scale = chunk(l, 4); g.next_shift = nullptr; g.next_mode = 1; g.h = e->h;
                g.rows = M; g.rows_per_batch = N; g.d = d; g.ld_mod = e->ld_mod; g.eps = c.norm_eps; g.eps_next = 1e-6f; g.scale_pre = 1;
                // round 5 (option moe_route_fused): h is the space router's input and this kernel holds the row - it routes on its way out
                // (one launch less per layer; moe_route_kernel's arithmetic, statement for statement)
                if (lt_opt(OPT_MOE_ROUTE_FUSED)) {
                    g.route_w = w.gate_s; g.route_E = e->E; g.route_sel = e->moe_sel; g.route_wts = e->moe_wts;
                    if (e->moe_force_rows) {
                        LT_REQUIRE(e->moe_force_rows == M, "forced MoE routing was given for %d rows, this call has %d", e->moe_force_rows, M);
                        g.route_forced = e->moe_force + ((size_t)l * 2 + 1) * (size_t)e->cfg.max_batch * e->cfg.max_tokens * 2;
                    }
                }
                if (launch_gated_residual_norm(g, s)) return 1;
            }
            if (moe_ffn(e, w, l, 1, M, N, B, s, lt_opt(OPT_MOE_ROUTE_FUSED) != 0, pair_moe)) return 1;
            last_post_w = w.norm_space;
            last_gate = chunk(l, 5);
        }
        {   // x += gate' * post(ffn) ; h = next layer's pre-norm + modulate, or the final layer's LayerNorm + modulate
            ProfScope ps(e, 2, 0, s);
            GatedResArgs g;
            g.x = e->x; g.y = e->o; g.post_w = last_post_w; g.gate = last_gate;
            if (ys_f) { g.ystat = e->ystat; g.ystat_slots = ys_f; }
            if (e->E != 0) moe_y(e, g, e->moe_mode == 1 ? l : -1);
            g.post_mode = post_mode; g.gate_mode = gate_mode; g.h = e->h;
            g.rows = M; g.rows_per_batch = N; g.d = d; g.ld_mod = e->ld_mod; g.eps = c.norm_eps; g.eps_next = 1e-6f; g.scale_pre = 1;
            if (l + 1 < L) {
                g.next_w = v.pre_w ? e->lw[l + 1].attn_norm1 : nullptr; g.next_scale = chunk(l + 1, v.i_scale[0]);
                g.next_shift = chunk(l + 1, v.i_shift[0]); g.next_mode = 1; g.h_pair = pair;
            } else {  // final layer: LayerNorm(no affine, 1e-6) * (1 + scale) [+ shift] (model.py:657-661 / models.py:829-832)
                const u16* fin = e->mod + (size_t)L * cd;
                g.next_w = nullptr; g.next_mode = 2;
                g.next_shift = v.final_chunks == 2 ? fin : nullptr;
                g.next_scale = v.final_chunks == 2 ? fin + d : fin;
            }
            if (l + 1 < L) prefetch_rider(e, &g.pf, e->h, d, e->lw[l + 1].wqkv, d, e->qkv, e->qkvn, M, e->qkvn, d, 0);
            if (launch_gated_residual_norm(g, s)) return 1;
        }
    }
    if (gemm(e, e->h, d, e->final_w, d, e->frows, e->nfinal, M, e->nfinal, d, e->final_b, 0, s)) return 1;
    {
        ProfScope ps(e, 2, 0, s);
        const int cfg_ch = a->cfg_channels > 0 ? a->cfg_channels : 3;
        if (!pk) {
            if (launch_unpatchify_cfg(e->frows, e->nfinal, out, a->io_dtype, B, c.in_channels, c.out_channels, a->latent_h,
                                      a->latent_w, p, use_cfg, a->cfg_scale, cfg_ch, Wrow, s)) return 1;
        } else {
            for (int b = 0; b < B; ++b)  // x[i][:L] -> [C, H_b, W_b], sigma half dropped (model.py:757-768, :859-864)
                if (launch_unpatchify_cfg(e->frows + (size_t)b * N * e->nfinal, e->nfinal, pk->out_ptrs[b], a->io_dtype, 1, c.in_channels,
                                          c.out_channels, pk->hw[2 * b], pk->hw[2 * b + 1], p, 0, 1.0f, cfg_ch, 0, s)) return 1;
        }
    }
    return 0;
}

// One model evaluation through a cached HIP graph.  The first call for a key runs eagerly (it also performs the one-time
// hipFuncSetAttribute calls of the launchers), the second captures the same launch sequence on the engine's private stream
// with the staging buffers as input / output and instantiates it, later calls replay: copy in, one graph launch, copy out.
// Kernel arguments are baked into a graph, so the key holds everything they depend on: the step arguments (shapes, cfg scale,
// RoPE scaling, softmax scale inputs), the prompt dimensions, cfg on / off and the option generation.  Weights and prompt
// CONTENTS live behind fixed pointers and may change freely.
// profiling brackets launches with HIP events, which a graph cannot carry: while an enabled class still has event budget left the
// evaluation runs eagerly; once the budgets are used up replays resume and only add their launch / flop counts (lt_profile_read
// scales the measured time by launches / bracketed launches - every evaluation has the same launch mix)
bool profiling_wants_events(const lt_engine* e) {
    if (!e->prof_on) return false;
    for (int k = 0; k < 3; ++k) {
        const ProfClass& pc = e->prof[k];
        if (((e->prof_mask >> k) & 1) && pc.used < std::min(pc.ev.size(), pc.budget)) return true;  // (also before the window opens)
    }
    return false;
}

int forward_graphed(lt_engine* e, const void* x_in, const float* t_dev, void* out, const lt_step_args* a, int use_cfg, hipStream_t s) {
    if (!lt_opt(OPT_GRAPH) || profiling_wants_events(e) || e->moe_rec_on || e->moe_force_rows) return run_forward(e, x_in, t_dev, out, a, use_cfg, s);
    const int B = a->batch;
    // "graph" 2 (default): replay above 1024 rows only.  At 512 rows a replay (three staging copies + the graph launch) costs more than it
    // saves - plain launches are 1.5-4.6 % faster for the 600M ImageNet model, same box (profiles/r05/bench_ab_hip_graph_on_off_cfg1_cfg5.log:
    // 1.652 against 1.68-1.73 ms per NFE; the MoE model is neutral) - and the host issues its ~125 launches per evaluation in a quarter of
    // the time the GPU needs for them.  Above that the two are equal and the graph keeps a busy host out of the picture.
    if (lt_opt(OPT_GRAPH) == 2 && a->latent_h > 0 && a->latent_w > 0 && e->cfg.patch_size > 0 &&
        (long long)B * (a->latent_h / e->cfg.patch_size) * (a->latent_w / e->cfg.patch_size) <= 1024)
        return run_forward(e, x_in, t_dev, out, a, use_cfg, s);
    if (B < 1 || B > e->cfg.max_batch || a->latent_h <= 0 || a->latent_w <= 0 || (a->io_dtype != LT_BF16 && a->io_dtype != LT_F32))
        return run_forward(e, x_in, t_dev, out, a, use_cfg, s);  // let the eager path produce the error message
    const size_t sbytes = (size_t)B * e->cfg.in_channels * a->latent_h * a->latent_w * (a->io_dtype == LT_BF16 ? 2 : 4);
    const size_t cap_bytes = (size_t)e->cfg.max_batch * e->cfg.in_channels * e->cfg.max_tokens * e->cfg.patch_size * e->cfg.patch_size * 4;
    if (sbytes > cap_bytes) return run_forward(e, x_in, t_dev, out, a, use_cfg, s);
    // (both option generations: the process defaults' and this engine's overrides' - kernel selection is baked into a captured graph)
    const int extra[9] = {use_cfg, e->prompt_B, e->prompt_T, e->prompt_Tpad, e->reg_Y, e->reg_h, e->reg_w, lt_opt_generation(), lt_opt_engine_generation()};
    std::vector<char> key(sizeof(lt_step_args) + sizeof(extra));
    memcpy(key.data(), a, sizeof(lt_step_args));
    memcpy(key.data() + sizeof(lt_step_args), extra, sizeof(extra));
    lt_engine::GraphEntry* ge = nullptr;
    for (auto& g : e->graphs) if (g.key == key) { ge = &g; break; }
    if (!ge) {
        if (e->graphs.size() >= 16) {  // bound the cache: drop the oldest entry
            if (e->graphs.front().exec) (void)hipGraphExecDestroy(e->graphs.front().exec);
            e->graphs.erase(e->graphs.begin());
        }
        e->graphs.emplace_back();
        ge = &e->graphs.back();
        ge->key = key;
    }
    if (ge->failed || ge->uses++ == 0) {
        const int rc = run_forward(e, x_in, t_dev, out, a, use_cfg, s);
        ge->pair = e->last_pair;  // (the weight layout this key's evaluations run on; a function of the key)
        return rc;
    }
    // a caller that is capturing ITS stream (torch.cuda.graph around the sampler) cannot launch a graph or start a second capture
    // from inside: hand it plain launches, which its own capture records
    hipStreamCaptureStatus caller_cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &caller_cap) != hipSuccess) { (void)hipGetLastError(); caller_cap = hipStreamCaptureStatusNone; }
    if (caller_cap != hipStreamCaptureStatusNone) return run_forward(e, x_in, t_dev, out, a, use_cfg, s);
    // the RoPE table is ONE shared buffer outside every graph (it is rebuilt only when scale / ntk change): a replay of key A after
    // key B changed the table must rebuild it first - before every replay, not only before the capture
    if (ensure_rope(e, a, s)) return 1;
    // ... and so are the GEMM weights' layouts (row-major / pair): evaluations of another regime may have converted them since
    if (ensure_weight_layout(e, ge->pair, s)) return 1;
    if (!ge->exec) {
        if (!e->cap_stream) LT_CHECK_HIP(hipStreamCreateWithFlags(&e->cap_stream, hipStreamNonBlocking));
        hipGraph_t graph = nullptr;
        bool ok = hipStreamBeginCapture(e->cap_stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
        int rc = 0;
        if (ok) {
            ge->tally = lt_engine::GraphTally();
            e->tally = &ge->tally;
            rc = run_forward(e, e->g_x, e->g_t, e->g_out, a, use_cfg, e->cap_stream);
            e->tally = nullptr;
            if (rc == 0 && e->last_pair != ge->pair) { lt_set_error("graph capture: the evaluation's operand layout changed between the eager run and the capture"); rc = 1; }
            ok = hipStreamEndCapture(e->cap_stream, &graph) == hipSuccess && rc == 0 && graph != nullptr;
        }
        if (ok) ok = hipGraphInstantiate(&ge->exec, graph, nullptr, nullptr, 0) == hipSuccess;
        if (graph) (void)hipGraphDestroy(graph);
        if (!ok) {  // capture is an optimisation: fall back to eager launches for this key
            (void)hipGetLastError();
            ge->exec = nullptr;
            ge->failed = true;
            if (rc) return rc;
            return run_forward(e, x_in, t_dev, out, a, use_cfg, s);
        }
    }
    LT_CHECK_HIP(hipMemcpyAsync(e->g_x, x_in, sbytes, hipMemcpyDeviceToDevice, s));
    LT_CHECK_HIP(hipMemcpyAsync(e->g_t, t_dev, (size_t)B * sizeof(float), hipMemcpyDeviceToDevice, s));
    LT_CHECK_HIP(hipGraphLaunch(ge->exec, s));
    LT_CHECK_HIP(hipMemcpyAsync(out, e->g_out, sbytes, hipMemcpyDeviceToDevice, s));
    ++e->graph_replays;
    if (e->prof_on)
        for (int k = 0; k < 3; ++k)
            if ((e->prof_mask >> k) & 1) { e->prof[k].flops += ge->tally.flops[k]; e->prof[k].launches += ge->tally.launches[k]; }
    return 0;
}

float bf16_round_host(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    u &= 0xffff0000u;
    memcpy(&f, &u, 4);
    return f;
}

}  // namespace

// =====================================================================================================
extern "C" int lt_create(const lt_config* cfg, lt_engine** out) {
    LT_REQUIRE(cfg && out, "lt_create: null argument");
    LT_REQUIRE(cfg->variant >= LT_VARIANT_NEXT_T2I && cfg->variant <= LT_VARIANT_NEXT_MOE_SPACE, "lt_create: unknown variant %d", cfg->variant);
    const bool is_moe = cfg->variant == LT_VARIANT_NEXT_MOE || cfg->variant == LT_VARIANT_NEXT_MOE_TIME || cfg->variant == LT_VARIANT_NEXT_MOE_SPACE;
    LT_REQUIRE(!is_moe || (cfg->num_experts >= 2 && cfg->num_experts <= 8),
               "lt_create: the MoE variant needs 2..8 experts (top-2 routing), got %d", cfg->num_experts);
    const VariantDesc vd = variant_desc(cfg->variant);
    LT_REQUIRE(cfg->dim > 0 && cfg->n_layers >= 1 && cfg->n_heads >= 1 && cfg->n_kv_heads >= 1 && cfg->ffn_hidden > 0 && cfg->adaln_dim > 0,
               "lt_create: dim, n_layers, n_heads, n_kv_heads, ffn_hidden and adaln_dim must be positive");
    LT_REQUIRE(cfg->patch_size >= 1 && cfg->in_channels >= 1 && cfg->out_channels >= 1 && cfg->max_tokens >= 1 && cfg->max_text >= 0,
               "lt_create: patch_size, in_channels, out_channels, max_tokens must be positive (max_text >= 0)");
    LT_REQUIRE(cfg->dim % cfg->n_heads == 0, "dim %% n_heads != 0");
    LT_REQUIRE(cfg->n_heads % cfg->n_kv_heads == 0, "n_heads %% n_kv_heads != 0");
    const int hd = cfg->dim / cfg->n_heads;
    LT_REQUIRE(hd == 48 || hd == 72 || hd == 96, "head_dim %d not built (48, 72, 96)", hd);
    LT_REQUIRE(cfg->dim % 64 == 0 && cfg->ffn_hidden % 64 == 0, "dim and ffn_hidden must be multiples of 64");
    LT_REQUIRE(!vd.text || (cfg->cap_feat_dim % 64 == 0 && cfg->cap_feat_dim > 0), "cap_feat_dim must be a positive multiple of 64");
    LT_REQUIRE(!vd.labels || cfg->num_classes > 0, "class-conditional variant needs num_classes > 0");
    LT_REQUIRE(cfg->dim <= 4096 && cfg->cap_feat_dim <= 4096, "dim / cap_feat_dim above 4096 not supported by the row kernels");
    LT_REQUIRE(cfg->adaln_dim % 8 == 0 && cfg->max_batch >= 1 && cfg->max_batch <= 8, "adaln_dim %% 8 and 1 <= max_batch <= 8 required");
    LT_REQUIRE(cfg->in_channels * cfg->patch_size * cfg->patch_size <= 64, "patch vector longer than 64");
    lt_engine* e = new lt_engine();
    e->cfg = *cfg;
    e->v = vd;
    e->d = cfg->dim; e->L = cfg->n_layers; e->H = cfg->n_heads; e->Hkv = cfg->n_kv_heads; e->hd = hd;
    e->F = cfg->ffn_hidden; e->dkv = e->Hkv * hd; e->qkvn = e->d + 2 * e->dkv; e->A = cfg->adaln_dim;
    e->cap = vd.text ? cfg->cap_feat_dim : 0; e->nfinal = cfg->patch_size * cfg->patch_size * cfg->out_channels;
    e->kpad = 64; e->chunks = vd.chunks; e->ld_mod = e->L * e->chunks * e->d + vd.final_chunks * e->d;
    e->label_rows = vd.labels ? cfg->num_classes + 1 : 0;
    e->E = is_moe ? cfg->num_experts : 0;
    e->moe_mode = cfg->variant == LT_VARIANT_NEXT_MOE_TIME ? 1 : (cfg->variant == LT_VARIANT_NEXT_MOE_SPACE ? 2 : 0);
    // 2-D RoPE: positions per axis (384, model.py:734); 1-D: one position per token of the longest sequence
    e->rope_len = vd.rope_1d ? round_up(cfg->max_tokens, 64) : (cfg->rope_table_len > 0 ? cfg->rope_table_len : 384);
    const int d = e->d, L = e->L, F = e->F, dkv = e->dkv, A = e->A, cap = e->cap, H = e->H, Hkv = e->Hkv;
    e->lw.resize(L);
    auto fail = [&]() { lt_destroy(e); return 1; };
#define A16(ptr, n) do { void* _p; if (dev_alloc(e, &_p, (size_t)(n) * 2)) return fail(); (ptr) = (u16*)_p; } while (0)
    const int Tmax = round_up(cfg->max_text > 0 ? cfg->max_text : 64, 64);
    for (int l = 0; l < L; ++l) {
        LayerW& w = e->lw[l];
        A16(w.wqkv, (size_t)e->qkvn * d); A16(w.wo, (size_t)d * d);
        if (e->E == 0) { A16(w.w13, (size_t)2 * F * d); A16(w.w2, (size_t)d * F); }
        else {
            const size_t E_ = e->E;
            if (e->moe_mode != 2) {
                A16(w.w13_t, E_ * 2 * F * d); A16(w.w2_t, E_ * d * F);
                // the time routers of ALL layers are one [L * E, A] matrix: their logits depend on the timestep embedding only, so one GEMV
                // per evaluation computes every layer's (run_forward; 16 launches of 12 us each at the 600M model before round 5)
                if (l == 0) A16(e->gate_t_all, (size_t)L * E_ * A);
                w.gate_t = e->gate_t_all + (size_t)l * E_ * A;
            }
            if (e->moe_mode != 1) { A16(w.w13_s, E_ * 2 * F * d); A16(w.w2_s, E_ * d * F); A16(w.gate_s, E_ * d); }
            A16(w.norm_time, d); A16(w.norm_space, d);
        }
        A16(w.q_norm_w, d); A16(w.q_norm_b, d); A16(w.k_norm_w, dkv); A16(w.k_norm_b, dkv);
        A16(w.attn_norm1, d); A16(w.attn_norm2, d); A16(w.ffn_norm1, d); A16(w.ffn_norm2, d);
        if (vd.text) {
            A16(w.wkvy, (size_t)2 * dkv * cap); A16(w.ky_norm_w, dkv); A16(w.ky_norm_b, dkv); A16(w.gate, H); A16(w.y_norm, cap);
            A16(w.ky, (size_t)cfg->max_batch * Hkv * Tmax * hd); A16(w.vty, (size_t)cfg->max_batch * Hkv * hd * Tmax);
        }
        char key[128];
        std::vector<const char*> names = {"attention.wq.weight", "attention.wk.weight", "attention.wv.weight", "attention.wo.weight",
                                          "adaLN_modulation.1.weight", "adaLN_modulation.1.bias"};
        if (e->E == 0) for (const char* nm : {"feed_forward.w1.weight", "feed_forward.w2.weight", "feed_forward.w3.weight"}) names.push_back(nm);
        if (vd.text) for (const char* nm : {"attention.wk_y.weight", "attention.wv_y.weight", "attention.gate", "attention_y_norm.weight"}) names.push_back(nm);
        if (cfg->variant == LT_VARIANT_NEXT_T2I)
            for (const char* nm : {"attention_norm1.weight", "attention_norm2.weight", "ffn_norm1.weight", "ffn_norm2.weight"}) names.push_back(nm);
        else if (cfg->variant == LT_VARIANT_NEXT_MOE)
            for (const char* nm : {"attention_norm.weight", "ffn_norm_time.weight", "ffn_norm_space.weight",
                                   "feed_forward_time.gate.weight", "feed_forward_space.gate.weight"}) names.push_back(nm);
        else if (e->moe_mode != 0)
            for (const char* nm : {"attention_norm.weight", "ffn_norm.weight", "feed_forward.gate.weight"}) names.push_back(nm);
        else
            for (const char* nm : {"attention_norm.weight", "ffn_norm.weight"}) names.push_back(nm);
        if (cfg->qk_norm) {
            for (const char* nm : {"attention.q_norm.weight", "attention.q_norm.bias", "attention.k_norm.weight", "attention.k_norm.bias"}) names.push_back(nm);
            if (vd.text) for (const char* nm : {"attention.ky_norm.weight", "attention.ky_norm.bias"}) names.push_back(nm);
        }
        for (const char* nm : names) { snprintf(key, sizeof(key), "layers.%d.%s", l, nm); e->need[key] = false; }
        for (int ex = 0; ex < e->E; ++ex)
            for (const char* br : {"feed_forward_time", "feed_forward_space"})
                for (const char* wn : {"w1", "w2", "w3"}) {
                    if (e->moe_mode != 0 && br[13] == 's') continue;  // one MoeLayer: keys "feed_forward.experts..."
                    snprintf(key, sizeof(key), "layers.%d.%s.experts.%d.%s.weight", l, e->moe_mode != 0 ? "feed_forward" : br, ex, wn);
                    e->need[key] = false;
                }
    }
    A16(e->xemb_w, (size_t)d * e->kpad); A16(e->xemb_b, d); A16(e->t0_w, (size_t)A * 256); A16(e->t0_b, A);
    A16(e->t2_w, (size_t)A * A); A16(e->t2_b, A); A16(e->pad_token, d);
    if (vd.text) { A16(e->capln_w, cap); A16(e->capln_b, cap); A16(e->cape_w, (size_t)A * cap); A16(e->cape_b, A); }
    if (vd.labels) A16(e->label_table, (size_t)e->label_rows * A);
    if (vd.eol) A16(e->eol_token, d);
    A16(e->adaln_w, (size_t)e->ld_mod * A); A16(e->adaln_b, e->ld_mod);
    A16(e->final_w, (size_t)e->nfinal * d); A16(e->final_b, e->nfinal);
    for (const char* nm : {"x_embedder.weight", "x_embedder.bias", "t_embedder.mlp.0.weight", "t_embedder.mlp.0.bias",
                           "t_embedder.mlp.2.weight", "t_embedder.mlp.2.bias", "final_layer.linear.weight", "final_layer.linear.bias",
                           "final_layer.adaLN_modulation.1.weight", "final_layer.adaLN_modulation.1.bias"})
        e->need[nm] = false;
    if (vd.text) for (const char* nm : {"cap_embedder.0.weight", "cap_embedder.0.bias", "cap_embedder.1.weight", "cap_embedder.1.bias"}) e->need[nm] = false;
    if (vd.labels) e->need["y_embedder.embedding_table.weight"] = false;
    if (vd.eol) e->need["eol_token"] = false;
    // workspace
    const size_t Bm = cfg->max_batch, Nm = cfg->max_tokens, M = Bm * Nm;
    const size_t Npad = round_up((int)Nm, 64);
    A16(e->x, M * d); A16(e->h, M * d); A16(e->qkv, M * e->qkvn); A16(e->q, M * d); A16(e->k, M * dkv);
    A16(e->vt, Bm * Hkv * hd * Npad); A16(e->attn, M * d); A16(e->o, M * d);
    if (e->E == 0) A16(e->u, M * F);
    else {  // expert-sorted buffers: every row appears twice, each expert segment starts on a 256-row tile
        // the experts' W1 | W3 GEMM gathers its rows through ONE buffer descriptor over the FFN input (launch_gemm_bf16: a_row_map):
        // fail here, with the sizes, not at the first forward
        if ((long long)M * d * 2 >= 0x40000000LL) {
            lt_set_error("lt_create: mixture-of-experts engine with max_batch * max_tokens = %zu rows of d = %d: the FFN input (%.2f GB) exceeds "
                         "the 1 GB the gather-on-load GEMM addresses; lower max_batch / max_tokens", M, d, (double)M * d * 2 / 1e9);
            return fail();
        }
        e->moe_tiles = (int)((2 * M + (size_t)e->E * 255 + 255) / 256);
        const size_t P = (size_t)e->moe_tiles * 256;
        A16(e->moe_us, P * F); A16(e->moe_ys, P * d); A16(e->moe_logits, Bm * e->E * (size_t)L); A16(e->moe_wts, 2 * M);
        void* q;
        if (dev_alloc(e, &q, (2 * M + 4) * sizeof(int))) return fail();  // (+4: moe_plan reads / writes whole 16-byte quads)
        e->moe_sel = (int*)q;
        if (dev_alloc(e, &q, (2 * M + 4) * sizeof(int))) return fail();
        e->moe_pos = (int*)q;
        if (dev_alloc(e, &q, (size_t)e->moe_tiles * sizeof(int))) return fail();
        e->moe_tile_expert = (int*)q;
        if (dev_alloc(e, &q, P * sizeof(int))) return fail();
        e->moe_src = (int*)q;
        if (e->moe_mode != 2) {  // per-layer plans of the time router (a few hundred KB per layer at 8192 rows)
            e->moe_tp_stride_rows = 2 * M + 4;
            e->moe_tp_stride_src = P;
            if (dev_alloc(e, &q, (size_t)L * e->moe_tp_stride_rows * sizeof(int))) return fail();
            e->moe_tp_sel = (int*)q;
            if (dev_alloc(e, &q, (size_t)L * e->moe_tp_stride_rows * sizeof(int))) return fail();
            e->moe_tp_pos = (int*)q;
            if (dev_alloc(e, &q, (size_t)L * e->moe_tp_stride_rows * sizeof(u16))) return fail();
            e->moe_tp_wts = (u16*)q;
            if (dev_alloc(e, &q, (size_t)L * e->moe_tiles * sizeof(int))) return fail();
            e->moe_tp_tile_expert = (int*)q;
            if (dev_alloc(e, &q, (size_t)L * P * sizeof(int))) return fail();
            e->moe_tp_src = (int*)q;
        }
    }
    {
        void* q;
        // [rows][slots] float2: 32 slots for the fused QKV launch's Q partials (GemmArgs::qstat), one per 128-column tile of the whole
        // projection for the small-M form (GemmArgs::rowstat)
        e->qstat_slots = std::max(32, (e->qkvn + 127) / 128);
        if (dev_alloc(e, &q, M * (size_t)e->qstat_slots * 2 * sizeof(float))) return fail();
        e->qstat = (float*)q;
        if (dev_alloc(e, &q, M * 2 * sizeof(float))) return fail();
        e->qmr = (float*)q;
        e->ystat_cap = 2 * ((d + 255) / 256);  // two wave halves per 256- or 288-column tile of a d-wide projection
        if (dev_alloc(e, &q, M * (size_t)e->ystat_cap * sizeof(float))) return fail();
        e->ystat = (float*)q;
    }
    if (hd == 96) {  // tail split of the one-wave attention kernel (launch_attention_v4_hd96): up to 4 parts of <= 128 rows per head
        void* q;
        e->attn_tail_ws_bytes = attention_tail_ws_floats((int)Bm * H, 4, 128) * sizeof(float);
        if (dev_alloc(e, &q, e->attn_tail_ws_bytes)) return fail();
        e->attn_tail_ws = (float*)q;
    }
    {   // split-K workspace (zeroed by dev_alloc: the counters must start at 0; every launch leaves them at 0)
        void* q;
        e->splitk_tiles = 256;  // slots of [2][64 x 128] floats (a 64 x 128 tile split two ways; a 128 x 128 tile split four ways takes four)
        if (dev_alloc(e, &q, (size_t)e->splitk_tiles * 2 * 64 * 128 * sizeof(float))) return fail();
        e->splitk_part = (float*)q;
        if (dev_alloc(e, &q, (size_t)e->splitk_tiles * sizeof(unsigned))) return fail();
        e->splitk_cnt = (unsigned*)q;
    }
    if (e->E > 0 && M >= 8192) {  // tail split of the experts' W2 GEMM (GemmArgs::tail_*): only where the grouped persistent kernel can run (>= 1.5 tiles per CU)
        void* q;
        e->tail_cap_parts = 4LL * num_cus();  // up to 4 parts for each of the < #CUs tiles of a partial round: 256 KB each
        if (dev_alloc(e, &q, (size_t)e->tail_cap_parts * 256 * 256 * sizeof(float))) return fail();
        e->tail_part = (float*)q;
        if (dev_alloc(e, &q, (size_t)num_cus() * sizeof(unsigned))) return fail();
        e->tail_cnt = (unsigned*)q;
    }
    A16(e->patches, M * e->kpad); A16(e->frows, M * e->nfinal); A16(e->mod, Bm * e->ld_mod);
    A16(e->tfeat, Bm * 256); A16(e->t1, Bm * A); A16(e->temb, Bm * A);
    A16(e->cap_emb, Bm * A); A16(e->adaln_in, Bm * A);
    if (vd.text) { A16(e->cap_ln, Bm * cap); A16(e->capb, Bm * Tmax * cap); A16(e->capn, Bm * Tmax * cap); A16(e->kvy, Bm * Tmax * 2 * dkv); }
    {
        void* p;
        if (dev_alloc(e, &p, Bm * Tmax * sizeof(float))) return fail();
        e->txt_bias = (float*)p;
        if (dev_alloc(e, &p, (size_t)2 * e->rope_len * (hd / 2) * 2 * sizeof(float))) return fail();
        e->rope = (float*)p;
        if (dev_alloc(e, &p, (size_t)2 * e->rope_len * (hd / 2) * 2 * sizeof(float))) return fail();
        e->rope_tr = (float*)p;
        const size_t state = Bm * cfg->in_channels * Nm * cfg->patch_size * cfg->patch_size * sizeof(float);
        for (int i = 0; i < 2; ++i) { if (dev_alloc(e, &e->ys[i], state)) return fail(); }
        if (dev_alloc(e, &e->ymid, state)) return fail();
        // stage times of lt_sample_ode: room for a 256-point grid of 4-stage steps without a (synchronising) reallocation
        if (hipMalloc((void**)&e->t_dev, (size_t)1024 * Bm * sizeof(float)) != hipSuccess) return fail();
        if (hipHostMalloc((void**)&e->t_pinned, (size_t)1024 * Bm * sizeof(float), hipHostMallocDefault) != hipSuccess) return fail();
        if (hipEventCreateWithFlags(&e->t_copied, hipEventDisableTiming) != hipSuccess) return fail();
        e->t_cap = (int)(1024 * Bm);
        if (dev_alloc(e, &e->g_x, state)) return fail();
        if (dev_alloc(e, &e->g_out, state)) return fail();
        {
            void* q;
            if (dev_alloc(e, &q, Bm * sizeof(float))) return fail();
            e->g_t = (float*)q;
        }
        for (int i = 0; i < 4; ++i) { if (dev_alloc(e, &e->kbuf[i], state)) return fail(); }
    }
#undef A16
    *out = e;
    return 0;
}

extern "C" void lt_destroy(lt_engine* e) {
    if (!e) return;
    for (auto& ge : e->graphs) if (ge.exec) (void)hipGraphExecDestroy(ge.exec);
    if (e->cap_stream) (void)hipStreamDestroy(e->cap_stream);
    for (auto& b : e->allocs) (void)hipFree(b.p);
    if (e->t_dev) (void)hipFree(e->t_dev);
    if (e->t_pinned) (void)hipHostFree(e->t_pinned);
    if (e->t_copied) (void)hipEventDestroy(e->t_copied);
    if (e->pk_dev) (void)hipFree(e->pk_dev);
    if (e->reg_txt) (void)hipFree(e->reg_txt);
    if (e->reg_qmap) (void)hipFree(e->reg_qmap);
    if (e->moe_rec) (void)hipFree(e->moe_rec);
    if (e->moe_force) (void)hipFree(e->moe_force);
    for (int k = 0; k < 3; ++k)
        for (auto& pr : e->prof[k].ev) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    delete e;
}

extern "C" int lt_set_weight(lt_engine* e, const char* key, const void* src_dev, int32_t dtype, const int64_t* shape,
                             int32_t ndim, void* stream) {
    LT_REQUIRE(e && key && src_dev && shape, "lt_set_weight: null argument");
    LT_REQUIRE(ndim >= 1 && ndim <= 8, "lt_set_weight: ndim %d outside 1..8", ndim);
    for (int i = 0; i < ndim; ++i) LT_REQUIRE(shape[i] > 0, "weight '%s': shape[%d] = %lld", key, i, (long long)shape[i]);
    Slot s;
    if (find_slot(e, key, &s)) return 2;
    if (ensure_weight_layout(e, false, (hipStream_t)stream)) return 1;  // uploads write row-major rows
    long long n = 1;
    for (int i = 0; i < ndim; ++i) n *= shape[i];
    LT_REQUIRE(n == (long long)s.rows * s.cols, "weight '%s': %lld elements given, %lld expected (%d x %d)", key, n,
               (long long)s.rows * s.cols, s.rows, s.cols);
    if (launch_upload_rows(src_dev, dtype, s.dst, s.rows, s.cols, s.dst_ld, s.r0, s.row_map, (hipStream_t)stream)) return 1;
    auto it = e->need.find(key);
    if (it != e->need.end()) it->second = true;
    e->weights_ok = false;
    // the hoisted conditioning (text K / V of every layer, caption / label embedding) was computed from the previous weights:
    // the next step must be preceded by a new lt_prepare_prompt / lt_prepare_labels (run_forward refuses otherwise)
    e->prompt_B = 0;
    return 0;
}

extern "C" int lt_weights_ready(lt_engine* e) {
    LT_REQUIRE(e, "null engine");
    for (auto& kv : e->need) LT_REQUIRE(kv.second, "weight '%s' was never uploaded", kv.first.c_str());
    e->weights_ok = true;
    return 0;
}

// per-layer text K / V of `Bc` captions (model.py:421-422, :602): RMSNorm_y -> wk_y | wv_y -> ky_norm, V^T
static int prepare_caption_kv(lt_engine* e, int Bc, int T, int Tpad, hipStream_t s) {
    const lt_config& c = e->cfg;
    const int cap = e->cap, dkv = e->dkv;
    for (int l = 0; l < e->L; ++l) {
        LayerW& w = e->lw[l];
        NormModArgs n;  // attention_y_norm (model.py:602)
        n.x = e->capb; n.w = w.y_norm; n.scale = nullptr; n.shift = nullptr; n.out = e->capn;
        n.rows = Bc * T; n.rows_per_batch = T; n.d = cap; n.ld_mod = 0; n.eps = c.norm_eps;
        if (launch_rmsnorm_mod(n, s)) return 1;
        GemmArgs g;  // wk_y | wv_y (model.py:421-422)
        g.A = e->capn; g.W = w.wkvy; g.C = e->kvy; g.bias = nullptr; g.M = Bc * T; g.N = 2 * dkv; g.K = cap;
        g.lda = cap; g.ldw = cap; g.ldc = 2 * dkv; g.bias_dtype = -1;
        if (launch_gemm_bf16(g, 0, 0, s)) return 1;
        QkPostArgs qa;  // ky_norm, no rotary (model.py:421)
        qa.src = e->kvy; qa.ld_src = 2 * dkv; qa.col0 = 0; qa.B = Bc; qa.N = T; qa.heads = e->Hkv; qa.hd = e->hd;
        qa.rope_mode = 0; qa.cs = nullptr; qa.t = nullptr; qa.grid_w = 1; qa.cs_len = 0; qa.ln_eps = 1e-5f; qa.watershed = 0.f;
        qa.ln_w = c.qk_norm ? w.ky_norm_w : nullptr; qa.ln_b = c.qk_norm ? w.ky_norm_b : nullptr; qa.dst = w.ky;
        qa.out_scale = (float)(1.0 / std::sqrt((double)e->hd)) * LOG2E;  // SDPA default scale (model.py:427-432), folded like the self-attention K
        if (launch_qk_norm_rope(qa, s)) return 1;
        if (launch_v_transpose(e->kvy, 2 * dkv, dkv, w.vty, Bc, T, Tpad, e->Hkv, e->hd, s)) return 1;
    }
    return 0;
}

extern "C" int lt_prepare_prompt(lt_engine* e, const void* cap_feats_dev, int32_t cap_dtype, const int32_t* cap_mask_dev,
                                 int32_t B, int32_t T, void* stream) {
    LT_REQUIRE(e && cap_feats_dev && cap_mask_dev, "lt_prepare_prompt: null argument");
    LtOptScope opt_scope(&e->opts);
    LT_REQUIRE(e->v.text, "lt_prepare_prompt: this variant is class-conditional (use lt_prepare_labels)");
    hipStream_t s = (hipStream_t)stream;
    const lt_config& c = e->cfg;
    const int Tmax = round_up(c.max_text > 0 ? c.max_text : 64, 64);
    LT_REQUIRE(B >= 1 && B <= c.max_batch, "prompt batch %d exceeds max_batch %d", B, c.max_batch);
    LT_REQUIRE(T >= 1 && T <= Tmax, "text length %d exceeds max_text %d", T, Tmax);
    LT_REQUIRE(cap_dtype == LT_BF16 || cap_dtype == LT_F32, "cap_feats dtype must be bf16 or f32");
    if (lt_weights_ready(e)) return 2;
    const int Tpad = round_up(T, 64), cap = e->cap, A = e->A;
    ProfScope ps(e, 2, 0, s);
    if (launch_cast_to_bf16(cap_feats_dev, cap_dtype, e->capb, (long long)B * T * cap, s)) return 1;
    if (launch_mask_to_bias(cap_mask_dev, e->txt_bias, B, T, Tpad, s)) return 1;
    // adaln_input's caption half (model.py:847-850)
    if (launch_cap_pool_ln(e->capb, LT_BF16, cap_mask_dev, e->capln_w, e->capln_b, e->cap_ln, B, T, cap, s)) return 1;
    if (launch_linear_small_m(e->cap_ln, e->cape_w, e->cape_b, e->cap_emb, B, A, cap, 0, s)) return 1;
    if (prepare_caption_kv(e, B, T, Tpad, s)) return 1;
    e->prompt_B = B; e->prompt_T = T; e->prompt_Tpad = Tpad; e->reg_Y = 0;
    return 0;
}

extern "C" int lt_prepare_prompt_regional(lt_engine* e, const void* cap_feats_dev, int32_t cap_dtype, const int32_t* cap_mask_dev,
                                          int32_t Y, int32_t T, const void* global_feats_dev, const int32_t* global_mask_dev,
                                          int32_t Tg, int32_t h_split, int32_t w_split, void* stream) {
    LT_REQUIRE(e && cap_feats_dev && cap_mask_dev && global_feats_dev && global_mask_dev, "lt_prepare_prompt_regional: null argument");
    LtOptScope opt_scope(&e->opts);
    LT_REQUIRE(e->cfg.variant == LT_VARIANT_NEXT_T2I, "lt_prepare_prompt_regional: text-conditional Next-DiT only");
    hipStream_t s = (hipStream_t)stream;
    const lt_config& c = e->cfg;
    const int Tmax = round_up(c.max_text > 0 ? c.max_text : 64, 64);
    LT_REQUIRE(Y >= 2 && Y <= c.max_batch, "%d captions exceed max_batch %d (the caption buffers are sized by it)", Y, c.max_batch);
    LT_REQUIRE(T >= 1 && T <= Tmax && Tg >= 1 && Tg <= Tmax, "text length %d / %d exceeds max_text %d", T, Tg, Tmax);
    LT_REQUIRE(h_split >= 1 && w_split >= 1, "lt_prepare_prompt_regional: split counts must be >= 1");
    LT_REQUIRE(cap_dtype == LT_BF16 || cap_dtype == LT_F32, "cap_feats dtype must be bf16 or f32");
    if (lt_weights_ready(e)) return 2;
    const int Tpad = round_up(T, 64), cap = e->cap, A = e->A;
    ProfScope ps(e, 2, 0, s);
    // adaLN conditioning from the GLOBAL caption, one row broadcast to the cond and the uncond row (model.py:866-870: the
    // pooled [1, C] embedding is added to t_emb [2, A])
    if (launch_cast_to_bf16(global_feats_dev, cap_dtype, e->capb, (long long)Tg * cap, s)) return 1;
    if (launch_cap_pool_ln(e->capb, LT_BF16, global_mask_dev, e->capln_w, e->capln_b, e->cap_ln, 1, Tg, cap, s)) return 1;
    if (launch_linear_small_m(e->cap_ln, e->cape_w, e->cape_b, e->cap_emb, 1, A, cap, 0, s)) return 1;
    LT_CHECK_HIP(hipMemcpyAsync(e->cap_emb + A, e->cap_emb, (size_t)A * 2, hipMemcpyDeviceToDevice, s));
    // the Y captions' keys / values
    if (launch_cast_to_bf16(cap_feats_dev, cap_dtype, e->capb, (long long)Y * T * cap, s)) return 1;
    if (launch_mask_to_bias(cap_mask_dev, e->txt_bias, Y, T, Tpad, s)) return 1;
    if (prepare_caption_kv(e, Y, T, Tpad, s)) return 1;
    const size_t need = (size_t)Y * c.max_tokens * e->d;
    if (e->reg_txt_elems < need) {
        // captured graphs bake this pointer into their kernels: none may outlive the buffer
        LT_CHECK_HIP(hipStreamSynchronize(s));
        for (auto& ge : e->graphs) if (ge.exec) (void)hipGraphExecDestroy(ge.exec);
        e->graphs.clear();
        if (e->reg_txt) LT_CHECK_HIP(hipFree(e->reg_txt));
        LT_CHECK_HIP(hipMalloc((void**)&e->reg_txt, need * 2));
        e->reg_txt_elems = need;
    }
    if (!e->reg_qmap) LT_CHECK_HIP(hipMalloc((void**)&e->reg_qmap, (size_t)c.max_batch * sizeof(int)));
    std::vector<int> qmap(Y, 0);
    qmap[Y - 1] = 1;
    LT_CHECK_HIP(hipMemcpyAsync(e->reg_qmap, qmap.data(), (size_t)Y * sizeof(int), hipMemcpyHostToDevice, s));
    LT_CHECK_HIP(hipStreamSynchronize(s));  // qmap is a stack temporary
    e->prompt_B = 2; e->prompt_T = T; e->prompt_Tpad = Tpad; e->reg_Y = Y; e->reg_h = h_split; e->reg_w = w_split;
    return 0;
}

extern "C" int lt_prepare_labels(lt_engine* e, const int32_t* labels_dev, int32_t B, void* stream) {
    LT_REQUIRE(e && labels_dev, "lt_prepare_labels: null argument");
    LtOptScope opt_scope(&e->opts);
    LT_REQUIRE(e->v.labels, "lt_prepare_labels: this variant is text-conditional (use lt_prepare_prompt)");
    LT_REQUIRE(B >= 1 && B <= e->cfg.max_batch, "label batch %d exceeds max_batch %d", B, e->cfg.max_batch);
    if (lt_weights_ready(e)) return 2;
    ProfScope ps(e, 2, 0, (hipStream_t)stream);
    // y_embedder(y) in eval mode (models.py:216-221) -> the label half of adaln_input (models.py:937-939)
    if (launch_label_gather(e->label_table, labels_dev, e->cap_emb, B, e->label_rows, e->A, (hipStream_t)stream)) return 1;
    e->prompt_B = B; e->prompt_T = 0; e->prompt_Tpad = 0;
    return 0;
}

extern "C" int lt_forward(lt_engine* e, const void* x_dev, const float* t_dev, void* out_dev, const lt_step_args* a, void* stream) {
    LT_REQUIRE(e && x_dev && t_dev && out_dev && a, "lt_forward: null argument");
    LtOptScope opt_scope(&e->opts);
    return forward_graphed(e, x_dev, t_dev, out_dev, a, 0, (hipStream_t)stream);
}

extern "C" int lt_forward_packed(lt_engine* e, const void* const* x_ptrs, const int32_t* hw_host, const float* t_dev,
                                 void* const* out_ptrs, const lt_step_args* a, void* stream) {
    LT_REQUIRE(e && x_ptrs && hw_host && t_dev && out_ptrs && a, "lt_forward_packed: null argument");
    LtOptScope opt_scope(&e->opts);
    LT_REQUIRE(a->batch >= 1 && a->batch <= e->cfg.max_batch, "lt_forward_packed: batch %d outside 1..max_batch %d", a->batch, e->cfg.max_batch);
    for (int b = 0; b < a->batch; ++b) LT_REQUIRE(x_ptrs[b] && out_ptrs[b], "lt_forward_packed: null sample pointer %d", b);
    if (!e->pk_dev) LT_CHECK_HIP(hipMalloc((void**)&e->pk_dev, 128 * sizeof(int)));
    PackedDesc pk{x_ptrs, out_ptrs, hw_host};
    return run_forward(e, nullptr, t_dev, nullptr, a, 0, (hipStream_t)stream, &pk);
}

extern "C" int lt_forward_cfg(lt_engine* e, const void* x_dev, const float* t_dev, void* out_dev, const lt_step_args* a, void* stream) {
    LT_REQUIRE(e && x_dev && t_dev && out_dev && a, "lt_forward_cfg: null argument");
    LtOptScope opt_scope(&e->opts);
    return forward_graphed(e, x_dev, t_dev, out_dev, a, 1, (hipStream_t)stream);
}

extern "C" int lt_sample_ode(lt_engine* e, const void* z_dev, void* traj_dev, void* final_dev, const float* tgrid_host,
                             int32_t n_grid, int32_t method, int32_t use_cfg, int32_t t_round, const lt_step_args* a,
                             void* stream) {
    LT_REQUIRE(e && z_dev && tgrid_host && a, "lt_sample_ode: null argument");
    LtOptScope opt_scope(&e->opts);
    LT_REQUIRE(n_grid >= 2, "lt_sample_ode: need at least 2 grid points");
    LT_REQUIRE(method >= LT_ODE_EULER && method <= LT_ODE_RK4, "lt_sample_ode: unknown method %d", method);
    hipStream_t s = (hipStream_t)stream;
    const int B = a->batch;
    // the state buffers were sized by lt_create for max_batch x max_tokens; check before the first copy into them (the model
    // calls below validate the same things, but only after z has been copied)
    LT_REQUIRE(B >= 1 && B <= e->cfg.max_batch, "lt_sample_ode: batch %d outside 1..max_batch %d", B, e->cfg.max_batch);
    LT_REQUIRE(a->latent_h > 0 && a->latent_w > 0 && a->latent_h % e->cfg.patch_size == 0 && a->latent_w % e->cfg.patch_size == 0 &&
                   (long long)(a->latent_h / e->cfg.patch_size) * (a->latent_w / e->cfg.patch_size) <= e->cfg.max_tokens,
               "lt_sample_ode: latent %dx%d is not a positive multiple of the patch size or exceeds max_tokens %d", a->latent_h, a->latent_w,
               e->cfg.max_tokens);
    LT_REQUIRE(a->io_dtype == LT_BF16 || a->io_dtype == LT_F32, "io_dtype must be bf16 or f32");
    const int stages = method == LT_ODE_EULER ? 1 : (method == LT_ODE_MIDPOINT ? 2 : 4);
    const int ncalls = (n_grid - 1) * stages;
    const long long n = (long long)B * e->cfg.in_channels * a->latent_h * a->latent_w;
    const size_t esz = a->io_dtype == LT_BF16 ? 2 : 4;
    const size_t sbytes = (size_t)n * esz;
    const bool bf = a->io_dtype == LT_BF16;
    // stage times; torchdiffeq's _PerturbFunc casts t to the state dtype before calling the model, then
    // integrators.py:108 broadcasts it to an fp32 [B] vector
    // (grids beyond the staging buffers' 1024 stage times per batch row - a 257-point rk4 grid - grow them: the one case in which
    //  this call synchronises)
    if (e->t_cap < ncalls * B) {
        LT_CHECK_HIP(hipStreamSynchronize(s));
        if (e->t_dev) LT_CHECK_HIP(hipFree(e->t_dev));
        if (e->t_pinned) LT_CHECK_HIP(hipHostFree(e->t_pinned));
        e->t_dev = nullptr; e->t_pinned = nullptr; e->t_cap = 0;
        LT_CHECK_HIP(hipMalloc((void**)&e->t_dev, (size_t)ncalls * B * sizeof(float)));
        LT_CHECK_HIP(hipHostMalloc((void**)&e->t_pinned, (size_t)ncalls * B * sizeof(float), hipHostMallocDefault));
        e->t_cap = ncalls * B;
    }
    LT_CHECK_HIP(hipEventSynchronize(e->t_copied));  // the previous trajectory's copy has left the staging buffer (normally long ago)
    std::vector<float> dts(n_grid - 1);
    for (int i = 0; i + 1 < n_grid; ++i) {
        const float t0 = tgrid_host[i], t1 = tgrid_host[i + 1];
        const float dt = t1 - t0;
        dts[i] = dt;
        float ts[4];
        if (method == LT_ODE_EULER) ts[0] = t0;
        else if (method == LT_ODE_MIDPOINT) { ts[0] = t0; ts[1] = t0 + 0.5f * dt; }
        else { ts[0] = t0; ts[1] = t0 + dt * (float)(1.0 / 3.0); ts[2] = t0 + dt * (float)(2.0 / 3.0); ts[3] = t1; }
        for (int k = 0; k < stages; ++k) {
            const float tv = (t_round && bf) ? bf16_round_host(ts[k]) : ts[k];
            for (int b = 0; b < B; ++b) e->t_pinned[((size_t)i * stages + k) * B + b] = tv;
        }
    }
    LT_CHECK_HIP(hipMemcpyAsync(e->t_dev, e->t_pinned, (size_t)ncalls * B * sizeof(float), hipMemcpyHostToDevice, s));
    LT_CHECK_HIP(hipEventRecord(e->t_copied, s));
    LT_CHECK_HIP(hipMemcpyAsync(e->ys[0], z_dev, sbytes, hipMemcpyDeviceToDevice, s));
    if (traj_dev) LT_CHECK_HIP(hipMemcpyAsync(traj_dev, z_dev, sbytes, hipMemcpyDeviceToDevice, s));
    int cur = 0;
    long long nfe = 0;
    auto model = [&](const void* y, int call, void* out) {
        ++nfe;
        return forward_graphed(e, y, e->t_dev + (size_t)call * B, out, a, use_cfg, s);
    };
    const int dt_code = bf ? 1 : 0;
    for (int i = 0; i + 1 < n_grid; ++i) {
        void* y0 = e->ys[cur];
        void* y1 = e->ys[cur ^ 1];
        // torchdiffeq multiplies the 0-dim DEVICE tensor dt = t1 - t0 (fp32) with the bf16 state / slopes; PyTorch's type
        // promotion keeps bf16 and casts the 0-dim operand to it first, so with a bf16 state every `dt * k` of the reference
        // sees bf16(dt) (0.5 dt is exact after that).  The stage TIMES above stay fp32 (t0 + dt / 2 is fp32 arithmetic).
        const float dt = bf ? bf16_round_host(dts[i]) : dts[i];
        const int c0 = i * stages;
        if (method == LT_ODE_EULER) {
            if (model(y0, c0, e->kbuf[0])) return 1;
            if (launch_ode_combine(0, y0, e->kbuf[0], nullptr, nullptr, nullptr, y1, dt_code, dt, n, s)) return 1;
        } else if (method == LT_ODE_MIDPOINT) {
            if (model(y0, c0, e->kbuf[0])) return 1;
            if (launch_ode_combine(0, y0, e->kbuf[0], nullptr, nullptr, nullptr, e->ymid, dt_code, 0.5f * dt, n, s)) return 1;
            if (model(e->ymid, c0 + 1, e->kbuf[1])) return 1;
            if (launch_ode_combine(0, y0, e->kbuf[1], nullptr, nullptr, nullptr, y1, dt_code, dt, n, s)) return 1;
        } else {
            if (model(y0, c0, e->kbuf[0])) return 1;
            if (launch_ode_combine(1, y0, e->kbuf[0], nullptr, nullptr, nullptr, e->ymid, dt_code, dt, n, s)) return 1;
            if (model(e->ymid, c0 + 1, e->kbuf[1])) return 1;
            if (launch_ode_combine(2, y0, e->kbuf[0], e->kbuf[1], nullptr, nullptr, e->ymid, dt_code, dt, n, s)) return 1;
            if (model(e->ymid, c0 + 2, e->kbuf[2])) return 1;
            if (launch_ode_combine(3, y0, e->kbuf[0], e->kbuf[1], e->kbuf[2], nullptr, e->ymid, dt_code, dt, n, s)) return 1;
            if (model(e->ymid, c0 + 3, e->kbuf[3])) return 1;
            if (launch_ode_combine(4, y0, e->kbuf[0], e->kbuf[1], e->kbuf[2], e->kbuf[3], y1, dt_code, dt, n, s)) return 1;
        }
        if (traj_dev) LT_CHECK_HIP(hipMemcpyAsync((char*)traj_dev + (size_t)(i + 1) * sbytes, y1, sbytes, hipMemcpyDeviceToDevice, s));
        cur ^= 1;
    }
    if (final_dev) LT_CHECK_HIP(hipMemcpyAsync(final_dev, e->ys[cur], sbytes, hipMemcpyDeviceToDevice, s));
    e->last_nfe = nfe;
    return 0;
}

extern "C" int64_t lt_last_nfe(lt_engine* e) { return e ? e->last_nfe : -1; }
extern "C" int64_t lt_graph_replays(lt_engine* e) { return e ? e->graph_replays : -1; }

// ---- MoE routing parity hooks -------------------------------------------------------------------------
namespace {
size_t moe_table_ints(const lt_engine* e) { return (size_t)e->L * 2 * e->cfg.max_batch * e->cfg.max_tokens * 2; }
}  // namespace

extern "C" int lt_moe_routing_record(lt_engine* e, int32_t on) {
    LT_REQUIRE(e && e->E > 0, "lt_moe_routing_record: not a mixture-of-experts engine");
    if (on && !e->moe_rec) {
        LT_CHECK_HIP(hipMalloc((void**)&e->moe_rec, moe_table_ints(e) * sizeof(int)));
        LT_CHECK_HIP(hipMemset(e->moe_rec, 0xff, moe_table_ints(e) * sizeof(int)));
    }
    e->moe_rec_on = on ? 1 : 0;
    return 0;
}

extern "C" int lt_moe_routing_read(lt_engine* e, int32_t* host_out, int32_t rows) {
    LT_REQUIRE(e && e->E > 0 && host_out, "lt_moe_routing_read: not a mixture-of-experts engine / null buffer");
    LT_REQUIRE(e->moe_rec && e->moe_rec_rows > 0, "lt_moe_routing_read: nothing recorded (lt_moe_routing_record, then a forward)");
    LT_REQUIRE(rows == e->moe_rec_rows, "lt_moe_routing_read: the last recorded call had %d rows, not %d", e->moe_rec_rows, rows);
    LT_CHECK_HIP(hipDeviceSynchronize());
    const size_t cap = (size_t)e->cfg.max_batch * e->cfg.max_tokens * 2;
    for (int lb = 0; lb < e->L * 2; ++lb)
        LT_CHECK_HIP(hipMemcpy(host_out + (size_t)lb * rows * 2, e->moe_rec + (size_t)lb * cap, (size_t)rows * 2 * sizeof(int), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int lt_moe_routing_force(lt_engine* e, const int32_t* host_sel, int32_t rows) {
    LT_REQUIRE(e && e->E > 0, "lt_moe_routing_force: not a mixture-of-experts engine");
    if (!host_sel || rows <= 0) { e->moe_force_rows = 0; return 0; }
    LT_REQUIRE((long long)rows <= (long long)e->cfg.max_batch * e->cfg.max_tokens, "lt_moe_routing_force: %d rows exceed the engine's capacity", rows);
    // per (layer, branch): a branch this variant RUNS needs two distinct expert ids in 0..E-1 for every row (a -1 there reached
    // the plan kernel's packed counters as a negative shift and corrupted the plan silently - ADVICE r3); a branch it does not run
    // (moe_mode 1: space, 2: time) may carry anything, -1 by convention, and is never read
    for (int lb = 0; lb < e->L * 2; ++lb) {
        const int branch = lb & 1;
        const bool runs = e->moe_mode == 0 || (e->moe_mode == 1 && branch == 0) || (e->moe_mode == 2 && branch == 1);
        const int32_t* tb = host_sel + (size_t)lb * rows * 2;
        for (int r = 0; r < rows; ++r) {
            const int a = tb[2 * r], b = tb[2 * r + 1];
            if (!runs) {
                LT_REQUIRE(a >= -1 && a < e->E && b >= -1 && b < e->E, "lt_moe_routing_force: expert id outside -1..%d (layer %d, branch %d, row %d)", e->E - 1, lb >> 1, branch, r);
                continue;
            }
            LT_REQUIRE(a >= 0 && a < e->E && b >= 0 && b < e->E, "lt_moe_routing_force: layer %d branch %d row %d: expert ids (%d, %d) must lie in 0..%d (this branch runs)",
                       lb >> 1, branch, r, a, b, e->E - 1);
            LT_REQUIRE(a != b, "lt_moe_routing_force: layer %d branch %d row %d selects expert %d twice (top-2 picks two different experts)", lb >> 1, branch, r, a);
        }
    }
    if (!e->moe_force) LT_CHECK_HIP(hipMalloc((void**)&e->moe_force, moe_table_ints(e) * sizeof(int)));
    LT_CHECK_HIP(hipDeviceSynchronize());
    const size_t cap = (size_t)e->cfg.max_batch * e->cfg.max_tokens * 2;
    for (int lb = 0; lb < e->L * 2; ++lb)
        LT_CHECK_HIP(hipMemcpy(e->moe_force + (size_t)lb * cap, host_sel + (size_t)lb * rows * 2, (size_t)rows * 2 * sizeof(int), hipMemcpyHostToDevice));
    e->moe_force_rows = rows;
    return 0;
}

// ---- profiling ------------------------------------------------------------------------------------------
extern "C" int lt_profile_enable(lt_engine* e, int32_t on) {
    LT_REQUIRE(e, "null engine");
    // under rocprofv3 the tool intercepts every event / signal; tens of thousands of warm-up records crash it (ROCm 7.2), and
    // its own kernel trace is what such a run is for: LT_NO_EVENT_PROFILE=1 (scripts/gpu_prof.sh) turns the engine's events off
    if (on && getenv("LT_NO_EVENT_PROFILE")) { e->prof_on = false; e->prof_mask = 0; return 0; }
    if (on) {
        const size_t want[3] = {2048, 512, 2048};  // launches bracketed per class before read-out scales up (lt_profile_read)
        bool created = false;
        for (int k = 0; k < 3; ++k) {
            while (e->prof[k].ev.size() < want[k]) {
                hipEvent_t a, b;
                // device-scope release: the default event flags make every record a system-scope release (L2 write-back)
                LT_CHECK_HIP(hipEventCreateWithFlags(&a, hipEventReleaseToDevice));
                LT_CHECK_HIP(hipEventCreateWithFlags(&b, hipEventReleaseToDevice));
                e->prof[k].ev.push_back({a, b});
                created = true;
            }
        }
        if (created) {  // first use of an event allocates its signal (tens of us): pay that here, not in a timed region
            for (int k = 0; k < 3; ++k)
                for (auto& pr : e->prof[k].ev) {
                    LT_CHECK_HIP(hipEventRecord(pr.first, nullptr));
                    LT_CHECK_HIP(hipEventRecord(pr.second, nullptr));
                }
            LT_CHECK_HIP(hipStreamSynchronize(nullptr));
        }
    }
    e->prof_on = on != 0;
    e->prof_mask = on == 1 ? 7 : (on & 7);  // 1 = all classes (historic), otherwise bit mask: 1 GEMM | 2 attention | 4 other
    return 0;
}

extern "C" int lt_profile_enable_mask(lt_engine* e, int32_t mask) {
    LT_REQUIRE(e && mask >= 0 && mask <= 7, "lt_profile_enable_mask: mask 0..7");
    if (lt_profile_enable(e, mask ? 7 : 0)) return 1;  // event pools; honours LT_NO_EVENT_PROFILE
    if (e->prof_on) e->prof_mask = mask;
    return 0;
}

extern "C" int lt_profile_set_budget(lt_engine* e, int32_t klass, int64_t max_event_launches) {
    LT_REQUIRE(e && klass >= 0 && klass < 3, "lt_profile_set_budget: bad class");
    e->prof[klass].budget = max_event_launches < 0 ? (size_t)-1 : (size_t)max_event_launches;
    e->prof[klass].skip = 0;
    return 0;
}

extern "C" int lt_profile_set_window(lt_engine* e, int32_t klass, int64_t skip_launches, int64_t max_event_launches) {
    LT_REQUIRE(e && klass >= 0 && klass < 3 && skip_launches >= 0, "lt_profile_set_window: bad arguments");
    e->prof[klass].skip = skip_launches;
    e->prof[klass].budget = max_event_launches < 0 ? (size_t)-1 : (size_t)max_event_launches;
    return 0;
}

extern "C" int lt_profile_reset(lt_engine* e) {
    LT_REQUIRE(e, "null engine");
    for (int k = 0; k < 3; ++k) { e->prof[k].flops = 0; e->prof[k].launches = 0; e->prof[k].used = 0; }
    return 0;
}

extern "C" int lt_profile_read(lt_engine* e, int32_t klass, double* ms, int64_t* launches, double* flops) {
    LT_REQUIRE(e && klass >= 0 && klass < 3, "lt_profile_read: bad class");
    ProfClass& pc = e->prof[klass];
    double total = 0;
    for (size_t i = 0; i < pc.used; ++i) {
        float t = 0;
        LT_CHECK_HIP(hipEventElapsedTime(&t, pc.ev[i].first, pc.ev[i].second));
        total += t;
    }
    // scale to all launches of the class if the event pool ran out (same launch mix every step)
    if (pc.used > 0 && (long long)pc.used < pc.launches) total *= (double)pc.launches / (double)pc.used;
    if (ms) *ms = total;
    if (launches) *launches = pc.launches;
    if (flops) *flops = pc.flops;
    return 0;
}

// Options (options.h).  lt_set_option: the process default - what every engine without its own override, and every lt_op_* operator
// call, sees.  lt_engine_set_option: an override for one engine (value LT_OPTION_INHERIT drops it again); it applies to every later call
// on that engine, from any thread, and to nothing else.  Both validate against the option table (gemm_prefetch, gemm_splitk
// and gemm_w4q_grouped used to accept any integer).
// returns the option's index, -1 for a retired name given its only accepted value (nothing to do), -2 after lt_set_error
static int option_lookup(const char* name, int32_t* value, const char* who) {
    if (!name) { lt_set_error("%s: null name", who); return -2; }
    if (strcmp(name, "gemm_pipeline") == 0 || strcmp(name, "gemm_pp_tail") == 0 || strcmp(name, "gemm_persist") == 0) {
        // round-1 study knobs: the kernels they selected were deleted with csrc/experimental/ in round 5
        if (*value != 0) { lt_set_error("%s(%s): removed - the round-1 study kernels are gone (git history keeps them)", who, name); return -2; }
        return -1;
    }
    const int id = lt_opt_find(name);
    if (id < 0) {
        lt_set_error("%s: unknown option '%s'", who, name);
        return -2;
    }
    int v = *value;
    if (lt_opt_validate(id, &v)) return -2;
    *value = v;
    return id;
}

extern "C" int lt_set_option(const char* name, int32_t value) {
    const int id = option_lookup(name, &value, "lt_set_option");
    if (id == -2) return 2;
    if (id >= 0) lt_opt_set_process(id, value);
    return 0;
}

extern "C" int lt_engine_set_option(lt_engine* e, const char* name, int32_t value) {
    LT_REQUIRE(e, "lt_engine_set_option: null engine");
    const bool inherit = value == LT_OPTION_INHERIT;
    int32_t v = inherit ? 0 : value;
    int id;
    if (inherit) {
        id = lt_opt_find(name);
        LT_REQUIRE(id >= 0, "lt_engine_set_option: unknown option '%s'", name ? name : "(null)");
    } else {
        id = option_lookup(name, &v, "lt_engine_set_option");
        if (id == -2) return 2;
        if (id < 0) return 0;
    }
    e->opts.v[id].store(inherit ? LT_OPT_INHERIT : v, std::memory_order_relaxed);
    e->opts.gen.fetch_add(1, std::memory_order_release);  // new HIP-graph keys: kernel selection is baked into a captured graph
    return 0;
}

extern "C" int lt_engine_get_option(lt_engine* e, const char* name, int32_t* value) {
    LT_REQUIRE(name && value, "lt_engine_get_option: null argument");
    const int id = lt_opt_find(name);
    LT_REQUIRE(id >= 0, "lt_engine_get_option: unknown option '%s'", name);
    LtOptScope opt_scope(e ? &e->opts : nullptr);  // e == NULL: the process default
    *value = lt_opt(id);
    return 0;
}

// ---- operator-level entry points ---------------------------------------------------------------------------
extern "C" int lt_op_gemm_bf16(const void* A, const void* W, const void* bias, int32_t bias_dtype, void* C, int32_t M,
                               int32_t N, int32_t K, int32_t epilogue, int32_t variant, void* stream) {
    LT_REQUIRE(A && W && C, "lt_op_gemm_bf16: null pointer");
    GemmArgs g;
    g.A = (const u16*)A; g.W = (const u16*)W; g.C = (u16*)C; g.bias = bias; g.M = M; g.N = N; g.K = K;
    g.lda = K; g.ldw = K; g.ldc = epilogue == 1 ? N / 2 : N; g.bias_dtype = bias ? bias_dtype : -1;
    return launch_gemm_bf16(g, epilogue, variant, (hipStream_t)stream);
}

// round 6: the row-pair-interleaved operand layout of the persistent GEMM (GemmArgs::pair_ab) at the op level
extern "C" int lt_op_pair_layout(void* m, int64_t rows, int32_t cols, int32_t to_pair, void* stream) {
    LT_REQUIRE(m, "lt_op_pair_layout: null pointer");
    return launch_pair_layout((u16*)m, rows, cols, to_pair, (hipStream_t)stream);
}
extern "C" int lt_op_gemm_bf16_pair(const void* A, const void* W, void* C, int32_t M, int32_t N, int32_t K, int32_t epilogue, int32_t pair_c,
                                    void* stream) {
    LT_REQUIRE(A && W && C, "lt_op_gemm_bf16_pair: null pointer");
    GemmArgs g;
    g.A = (const u16*)A; g.W = (const u16*)W; g.C = (u16*)C; g.bias = nullptr; g.M = M; g.N = N; g.K = K;
    g.lda = K; g.ldw = K; g.ldc = epilogue == 1 ? N / 2 : N; g.bias_dtype = -1; g.pair_ab = 3; g.pair_c = pair_c;
    return launch_gemm_bf16(g, epilogue, 0, (hipStream_t)stream);
}

extern "C" int lt_op_gemm_vt(const void* A, const void* W, void* vt, int32_t M, int32_t N, int32_t K, int32_t tokens, int32_t hd,
                             int32_t variant, void* stream) {
    LT_REQUIRE(A && W && vt, "lt_op_gemm_vt: null pointer");
    GemmArgs g;
    g.A = (const u16*)A; g.W = (const u16*)W; g.C = (u16*)vt; g.bias = nullptr; g.M = M; g.N = N; g.K = K;
    g.lda = K; g.ldw = K; g.ldc = 0; g.bias_dtype = -1; g.vt_tokens = tokens; g.vt_hd = hd; g.vt_npad = tokens;
    return launch_gemm_bf16(g, 2, variant, (hipStream_t)stream);
}

extern "C" int lt_op_gemm_qkv(const void* A, const void* W, void* C, void* vt, int32_t M, int32_t N, int32_t K, int32_t split,
                              int32_t tokens, int32_t hd, void* stream) {
    LT_REQUIRE(A && W && C && vt, "lt_op_gemm_qkv: null pointer");
    GemmArgs g;
    g.A = (const u16*)A; g.W = (const u16*)W; g.C = (u16*)C; g.bias = nullptr; g.bias_dtype = -1; g.M = M; g.N = N; g.K = K;
    g.lda = K; g.ldw = K; g.ldc = N; g.VT = (u16*)vt; g.vt_split = split; g.vt_tokens = tokens; g.vt_hd = hd; g.vt_npad = tokens;
    return launch_gemm_bf16(g, 3, 0, (hipStream_t)stream);
}

// The fused QKV launch with the Q columns' LayerNorm partials (GemmArgs::qstat) followed by the K pass of qk_norm_rope that reduces
// them (QkPostArgs::qstat_in): exactly the two launches the engine makes per layer on the attn_q_fused path.
extern "C" int lt_op_qkv_qstat(const void* A, const void* W, void* C, void* vt, int32_t M, int32_t N, int32_t K, int32_t split, int32_t tokens,
                               int32_t hd, int32_t q_cols, const void* k_ln_w, const void* k_ln_b, const void* cs_table, int32_t grid_w,
                               float k_out_scale, void* k_out, void* qstat_ws, void* q_mean_rstd, void* stream) {
    LT_REQUIRE(A && W && C && vt && k_ln_w && k_ln_b && cs_table && k_out && qstat_ws && q_mean_rstd, "lt_op_qkv_qstat: null pointer");
    LT_REQUIRE(hd > 0 && q_cols > 0 && q_cols % hd == 0 && split > q_cols && (split - q_cols) % hd == 0 && M % tokens == 0,
               "lt_op_qkv_qstat: bad column split (q %d | k | v at %d, head_dim %d)", q_cols, split, hd);
    GemmArgs g;
    g.A = (const u16*)A; g.W = (const u16*)W; g.C = (u16*)C; g.bias = nullptr; g.bias_dtype = -1; g.M = M; g.N = N; g.K = K;
    g.lda = K; g.ldw = K; g.ldc = N; g.VT = (u16*)vt; g.vt_split = split; g.vt_tokens = tokens; g.vt_hd = hd; g.vt_npad = tokens;
    const int bn = gemm_qkv_tile_width(g);
    LT_REQUIRE(bn > 0 && q_cols % bn == 0 && 2 * q_cols / bn <= 32, "lt_op_qkv_qstat: the problem does not take the fused QKV launch with whole Q tiles");
    g.qstat = (float*)qstat_ws; g.qstat_cols = q_cols; g.qstat_slots = 2 * q_cols / bn;  // qstat_ws: [M][qstat_slots] float2, <= [M][32]
    if (int rc = launch_gemm_bf16(g, 3, 0, (hipStream_t)stream)) return rc;
    QkPostArgs q;
    q.src = (const u16*)C; q.ld_src = N; q.col0 = q_cols; q.ln_w = (const u16*)k_ln_w; q.ln_b = (const u16*)k_ln_b; q.ln_eps = 1e-5f;
    q.dst = (u16*)k_out; q.B = M / tokens; q.N = tokens; q.heads = (split - q_cols) / hd; q.hd = hd; q.rope_mode = 1;
    q.cs = (const float*)cs_table; q.t = nullptr; q.grid_w = grid_w; q.cs_len = 0; q.watershed = 0.f; q.out_scale = k_out_scale;  // (one branch's table, as lt_op_qk_norm_rope)
    q.qstat_in = (const float*)qstat_ws; q.qstat_out = (float*)q_mean_rstd; q.qstat_slots = g.qstat_slots; q.qstat_width = q_cols;
    return launch_qk_norm_rope(q, (hipStream_t)stream);
}

extern "C" int lt_op_gemm_qkv_fusable(int32_t M, int32_t N, int32_t K, int32_t split, int32_t tokens, int32_t hd) {
    GemmArgs g;
    g.A = nullptr; g.W = nullptr; g.C = nullptr; g.bias = nullptr; g.bias_dtype = -1; g.M = M; g.N = N; g.K = K;
    g.lda = K; g.ldw = K; g.ldc = N; g.VT = (u16*)1; g.vt_split = split; g.vt_tokens = tokens; g.vt_hd = hd; g.vt_npad = tokens;
    return lt_opt(OPT_QKV_FUSED_GEMM) && lt_opt(OPT_QKV_VT_EPILOGUE) && gemm_qkv_fusable(g) ? 1 : 0;
}

extern "C" int lt_op_gemm_describe(int32_t M, int32_t N, int32_t K, int32_t epilogue, int32_t variant, char* out, int32_t cap) {
    LT_REQUIRE(out && cap > 0, "lt_op_gemm_describe: null buffer");
    GemmArgs g;
    g.A = nullptr; g.W = nullptr; g.C = nullptr; g.bias = nullptr; g.M = M; g.N = N; g.K = K;
    g.lda = K; g.ldw = K; g.ldc = epilogue == 1 ? N / 2 : N; g.bias_dtype = -1;
    snprintf(out, (size_t)cap, "%s", lt_gemm_describe(g, epilogue, variant));
    return 0;
}

extern "C" int lt_op_moe_plan(void* sel, const void* sample_logits, void* wts, int32_t rows, int32_t rows_per_sample, int32_t E, void* pos,
                              void* src, void* tile_expert, int32_t max_tiles, void* stream) {
    LT_REQUIRE(sel && pos && src && tile_expert, "lt_op_moe_plan: null pointer");
    LT_REQUIRE(sample_logits == nullptr || wts != nullptr, "lt_op_moe_plan: routing from per-sample logits writes the weights too");
    MoeArgs m;
    m.x = nullptr; m.gate_w = nullptr; m.sample_logits = (const u16*)sample_logits; m.forced = nullptr;
    m.rows = rows; m.rows_per_sample = rows_per_sample > 0 ? rows_per_sample : rows; m.d = 8; m.E = E;
    m.sel = (int*)sel; m.wts = (u16*)wts; m.pos = (int*)pos; m.src = (int*)src; m.tile_expert = (int*)tile_expert; m.max_tiles = max_tiles;
    return launch_moe_plan(m, (hipStream_t)stream);
}

extern "C" int lt_op_gemm_splitk(const void* A, const void* W, void* C, int32_t M, int32_t N, int32_t K, void* part_f32, void* counters_u32,
                                 int32_t tiles, void* stream) {
    LT_REQUIRE(A && W && C && part_f32 && counters_u32, "lt_op_gemm_splitk: null pointer");
    LT_REQUIRE(M > 0 && N > 0 && K > 0, "lt_op_gemm_splitk: empty problem");
    GemmArgs g;
    g.A = (const u16*)A; g.W = (const u16*)W; g.C = (u16*)C; g.bias = nullptr; g.bias_dtype = -1; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldc = N;
    g.splitk_part = (float*)part_f32; g.splitk_cnt = (unsigned*)counters_u32; g.splitk_tiles = tiles;
    return launch_gemm_bf16(g, 0, 8, (hipStream_t)stream);  // variant 8 = the 64 x 128 tile, the only one that splits
}

// the same workspace with the kernel and the split left to the launcher, as in the engine (variant 0): two ways on 64 x 128 tiles, or - round 5,
// K >= 4096 - four ways on 128 x 128 tiles, or none
extern "C" int lt_op_gemm_splitk_auto(const void* A, const void* W, void* C, int32_t M, int32_t N, int32_t K, void* part_f32, void* counters_u32,
                                      int32_t slots, void* stream) {
    LT_REQUIRE(A && W && C && part_f32 && counters_u32, "lt_op_gemm_splitk_auto: null pointer");
    LT_REQUIRE(M > 0 && N > 0 && K > 0, "lt_op_gemm_splitk_auto: empty problem");
    GemmArgs g;
    g.A = (const u16*)A; g.W = (const u16*)W; g.C = (u16*)C; g.bias = nullptr; g.bias_dtype = -1; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldc = N;
    g.splitk_part = (float*)part_f32; g.splitk_cnt = (unsigned*)counters_u32; g.splitk_tiles = slots;
    return launch_gemm_bf16(g, 0, 0, (hipStream_t)stream);
}

extern "C" int lt_op_gemm_grouped(const void* A, const void* W, const void* tile_expert, int64_t w_expert_stride, void* C,
                                  int32_t M, int32_t N, int32_t K, int32_t epilogue, int32_t variant, void* stream) {
    LT_REQUIRE(A && W && C && tile_expert, "lt_op_gemm_grouped: null pointer");
    LT_REQUIRE(M > 0 && M % 256 == 0, "lt_op_gemm_grouped: M=%d must be a positive multiple of 256 (expert segments)", M);
    GemmArgs g;
    g.A = (const u16*)A; g.W = (const u16*)W; g.C = (u16*)C; g.bias = nullptr; g.M = M; g.N = N; g.K = K;
    g.lda = K; g.ldw = K; g.ldc = epilogue == 1 ? N / 2 : N; g.bias_dtype = -1;
    g.tile_expert = (const int*)tile_expert; g.w_expert_stride = w_expert_stride;
    return launch_gemm_bf16(g, epilogue, variant, (hipStream_t)stream);
}

// lt_op_gemm_grouped on the persistent kernel (variant 15) with the tail split of round 6: the tiles of a partial last round of the walk are cut
// along K into 2 / 4 parts that hand fp32 accumulators through tail_part_f32 ([cap_parts][256 x 256] floats) and count in on counters_u32
// ([number of CUs] words, zero before the first launch; every launch leaves them zero)
extern "C" int lt_op_gemm_grouped_tail(const void* A, const void* W, const void* tile_expert, int64_t w_expert_stride, void* C, int32_t M, int32_t N,
                                       int32_t K, void* tail_part_f32, void* counters_u32, int32_t cap_parts, void* stream) {
    LT_REQUIRE(A && W && C && tile_expert && tail_part_f32 && counters_u32 && cap_parts > 0, "lt_op_gemm_grouped_tail: null pointer");
    LT_REQUIRE(M > 0 && M % 256 == 0, "lt_op_gemm_grouped_tail: M=%d must be a positive multiple of 256 (expert segments)", M);
    GemmArgs g;
    g.A = (const u16*)A; g.W = (const u16*)W; g.C = (u16*)C; g.bias = nullptr; g.M = M; g.N = N; g.K = K;
    g.lda = K; g.ldw = K; g.ldc = N; g.bias_dtype = -1;
    g.tile_expert = (const int*)tile_expert; g.w_expert_stride = w_expert_stride;
    g.tail_part = (float*)tail_part_f32; g.tail_cnt = (unsigned*)counters_u32; g.tail_cap_parts = cap_parts;
    return launch_gemm_bf16(g, 0, 15, (hipStream_t)stream);
}

extern "C" int lt_op_gemm_grouped_gather(const void* A, int32_t a_rows, const void* row_map, const void* W, const void* tile_expert,
                                         int64_t w_expert_stride, void* C, int32_t M, int32_t N, int32_t K, int32_t epilogue, int32_t variant,
                                         void* stream) {
    LT_REQUIRE(A && W && C && tile_expert && row_map, "lt_op_gemm_grouped_gather: null pointer");
    LT_REQUIRE(M > 0 && M % 256 == 0 && a_rows > 0, "lt_op_gemm_grouped_gather: M=%d must be a positive multiple of 256, a_rows > 0", M);
    GemmArgs g;
    g.A = (const u16*)A; g.W = (const u16*)W; g.C = (u16*)C; g.bias = nullptr; g.M = M; g.N = N; g.K = K;
    g.lda = K; g.ldw = K; g.ldc = epilogue == 1 ? N / 2 : N; g.bias_dtype = -1;
    g.tile_expert = (const int*)tile_expert; g.w_expert_stride = w_expert_stride;
    g.a_row_map = (const int*)row_map; g.a_map_rows = a_rows;
    return launch_gemm_bf16(g, epilogue, variant, (hipStream_t)stream);
}

extern "C" int lt_op_pack_w13(const void* w1, const void* w3, void* out, int32_t F, int32_t K, void* stream) {
    LT_REQUIRE(w1 && w3 && out, "lt_op_pack_w13: null pointer");
    return launch_pack_w13((const u16*)w1, (const u16*)w3, (u16*)out, F, K, (hipStream_t)stream);
}

extern "C" int lt_op_rmsnorm_mod(const void* x, const void* w, const void* scale, const void* shift, int32_t ld_mod,
                                 void* out, int32_t B, int32_t N, int32_t d, float eps, int32_t scale_pre, void* stream) {
    LT_REQUIRE(x && out, "lt_op_rmsnorm_mod: null pointer");
    NormModArgs n;
    n.x = (const u16*)x; n.w = (const u16*)w; n.scale = (const u16*)scale; n.shift = (const u16*)shift; n.out = (u16*)out;
    n.rows = B * N; n.rows_per_batch = N; n.d = d; n.ld_mod = ld_mod; n.eps = eps; n.scale_pre = scale_pre;
    return launch_rmsnorm_mod(n, (hipStream_t)stream);
}

extern "C" int lt_op_gated_residual_norm(void* x, const void* y, const void* post_w, const void* gate, int32_t post_mode,
                                         int32_t gate_mode, const void* next_w, const void* next_scale,
                                         const void* next_shift, int32_t next_mode, int32_t ld_mod, void* h, int32_t B,
                                         int32_t N, int32_t d, float eps, float eps_next, int32_t scale_pre, void* stream) {
    LT_REQUIRE(x && y, "lt_op_gated_residual_norm: null pointer");
    GatedResArgs g;
    g.x = (u16*)x; g.y = (const u16*)y; g.post_w = (const u16*)post_w; g.gate = (const u16*)gate;
    g.next_w = (const u16*)next_w; g.next_scale = (const u16*)next_scale; g.next_shift = (const u16*)next_shift;
    g.h = (u16*)h; g.rows = B * N; g.rows_per_batch = N; g.d = d; g.ld_mod = ld_mod; g.post_mode = post_mode;
    g.gate_mode = gate_mode; g.next_mode = next_mode; g.eps = eps; g.eps_next = eps_next; g.scale_pre = scale_pre;
    return launch_gated_residual_norm(g, (hipStream_t)stream);
}

// The O / W2 projection followed by the sandwich-norm row step - exactly the two launches the engine makes per branch.  use_ystat 1: the
// GEMM's epilogue leaves the rows' sum-of-squares partials in ystat_ws and the row kernel runs its streaming form on them (option grn_ystat's
// path; refused when the problem does not take the persistent kernel's plain dense tiles); 0: the row kernel reduces y itself.
extern "C" int lt_op_proj_gated_residual_norm(const void* A, const void* W, void* y, void* ystat_ws, int32_t ystat_cap, int32_t K, void* x,
                                              const void* post_w, const void* gate, const void* next_w, const void* next_scale, int32_t ld_mod,
                                              void* h, int32_t B, int32_t N, int32_t d, float eps, int32_t use_ystat, void* stream) {
    LT_REQUIRE(A && W && y && x && post_w && gate && next_w && next_scale && h, "lt_op_proj_gated_residual_norm: null pointer");
    GemmArgs g;
    g.A = (const u16*)A; g.W = (const u16*)W; g.C = (u16*)y; g.bias = nullptr; g.M = B * N; g.N = d; g.K = K; g.lda = K; g.ldw = K; g.ldc = d;
    g.bias_dtype = -1;
    GatedResArgs r;
    if (use_ystat) {
        const int ys = gemm_ystat_slots(g, 0);
        LT_REQUIRE(ys > 0, "lt_op_proj_gated_residual_norm: this problem does not run on the persistent kernel's plain dense tiles (no ystat)");
        LT_REQUIRE(ystat_ws && ystat_cap >= ys, "lt_op_proj_gated_residual_norm: ystat workspace of %d floats per row, the launch fills %d", ystat_cap, ys);
        g.ystat = (float*)ystat_ws; g.ystat_slots = ys;
        r.ystat = (const float*)ystat_ws; r.ystat_slots = ys;
    }
    if (int rc = launch_gemm_bf16(g, 0, 0, (hipStream_t)stream)) return rc;
    r.x = (u16*)x; r.y = (const u16*)y; r.post_w = (const u16*)post_w; r.gate = (const u16*)gate; r.next_w = (const u16*)next_w;
    r.next_scale = (const u16*)next_scale; r.next_shift = nullptr; r.h = (u16*)h; r.rows = B * N; r.rows_per_batch = N; r.d = d; r.ld_mod = ld_mod;
    r.post_mode = 1; r.gate_mode = 0; r.next_mode = 1; r.eps = eps; r.eps_next = 1e-6f; r.scale_pre = 1;
    return launch_gated_residual_norm(r, (hipStream_t)stream);
}

extern "C" int lt_op_prep_mod(void* mod, int32_t B, int32_t ld_mod, int32_t L, int32_t chunks, int32_t d, uint32_t tanh_mask,
                              uint32_t scale_mask, int32_t final_scale_chunk, void* stream) {
    LT_REQUIRE(mod, "lt_op_prep_mod: null pointer");
    return launch_prep_mod((u16*)mod, B, ld_mod, L, chunks, d, tanh_mask, scale_mask, final_scale_chunk, (hipStream_t)stream);
}

extern "C" int lt_op_qk_norm_rope(const void* src, int32_t ld_src, int32_t col0, const void* ln_w, const void* ln_b,
                                  float ln_eps, void* dst, int32_t B, int32_t N, int32_t heads, int32_t hd,
                                  int32_t rope_mode, const void* cs_table, int32_t grid_w, float out_scale, void* stream) {
    LT_REQUIRE(src && dst, "lt_op_qk_norm_rope: null pointer");
    QkPostArgs q;
    q.src = (const u16*)src; q.ld_src = ld_src; q.col0 = col0; q.ln_w = (const u16*)ln_w; q.ln_b = (const u16*)ln_b;
    q.ln_eps = ln_eps; q.dst = (u16*)dst; q.B = B; q.N = N; q.heads = heads; q.hd = hd; q.rope_mode = rope_mode;
    q.cs = (const float*)cs_table; q.t = nullptr; q.grid_w = grid_w > 0 ? grid_w : 1; q.watershed = 0.f;
    q.out_scale = out_scale;
    q.cs_len = 0;  // op level: the caller hands over the single branch table it wants (no branch offset)
    return launch_qk_norm_rope(q, (hipStream_t)stream);
}

extern "C" int lt_op_v_transpose(const void* src, int32_t ld_src, int32_t col0, void* dst, int32_t B, int32_t N,
                                 int32_t Npad, int32_t kv_heads, int32_t hd, void* stream) {
    LT_REQUIRE(src && dst, "lt_op_v_transpose: null pointer");
    return launch_v_transpose((const u16*)src, ld_src, col0, (u16*)dst, B, N, Npad, kv_heads, hd, (hipStream_t)stream);
}

extern "C" int lt_op_attention(const void* q, const void* k, const void* vt, const float* bias, void* out, const void* gate,
                               int32_t accumulate, int32_t B, int32_t H, int32_t Hkv, int32_t N, int32_t Nk, int32_t Nkpad,
                               int32_t hd, float scale, int32_t k_prescaled, void* stream) {
    LT_REQUIRE(q && k && vt && out, "lt_op_attention: null pointer");
    AttnArgs a;
    a.q = (const u16*)q; a.k = (const u16*)k; a.vt = (const u16*)vt; a.bias = bias; a.out = (u16*)out;
    a.gate = (const u16*)gate; a.accumulate = accumulate; a.B = B; a.H = H; a.Hkv = Hkv; a.N = N; a.Nk = Nk;
    a.Nkpad = Nkpad; a.hd = hd; a.scale = scale; a.k_prescaled = k_prescaled;
    return launch_attention(a, (hipStream_t)stream);
}

// self-attention whose queries come straight from the QKV projection (AttnArgs::q_raw): q_norm + 2-D RoPE in the kernel's prologue
extern "C" int lt_op_attention_qraw(const void* qkv, int32_t ld, int32_t q_col0, const void* q_mean_rstd, const void* q_ln_w, const void* q_ln_b,
                                    const void* cs_table, const void* cs_table_t, int32_t table_len, int32_t grid_w, const void* k,
                                    const void* vt, void* out, int32_t B, int32_t H, int32_t Hkv, int32_t N, int32_t Nkpad, int32_t hd,
                                    void* stream) {
    LT_REQUIRE(qkv && q_mean_rstd && q_ln_w && q_ln_b && cs_table && cs_table_t && k && vt && out, "lt_op_attention_qraw: null pointer");
    AttnArgs a;
    a.q = nullptr; a.k = (const u16*)k; a.vt = (const u16*)vt; a.bias = nullptr; a.out = (u16*)out; a.gate = nullptr; a.accumulate = 0;
    a.B = B; a.H = H; a.Hkv = Hkv; a.N = N; a.Nk = N; a.Nkpad = Nkpad; a.hd = hd; a.scale = 1.f; a.k_prescaled = 1;
    a.q_raw = (const u16*)qkv; a.q_ld = ld; a.q_col0 = q_col0; a.q_stat = (const float*)q_mean_rstd;
    a.q_ln_w = (const u16*)q_ln_w; a.q_ln_b = (const u16*)q_ln_b;
    a.rope_cs = (const float*)cs_table; a.rope_cs_t = (const float*)cs_table_t; a.rope_t = nullptr; a.rope_watershed = 0.f;  // branch 1
    a.rope_cs_len = table_len; a.rope_grid_w = grid_w;
    return launch_attention(a, (hipStream_t)stream);
}

// The small-M QKV projection with the per-tile LayerNorm partials (GemmArgs::rowstat) followed by the fused q / k post-processing +
// attention launch (AttnSmallArgs): exactly the two launches the engine makes per layer on the attn_small_fused path.
extern "C" int lt_op_qkv_attention_small(const void* A, const void* W, void* qkv, int32_t M, int32_t K, int32_t H, int32_t Hkv, int32_t tokens,
                                         int32_t hd, const void* q_ln_w, const void* q_ln_b, const void* k_ln_w, const void* k_ln_b,
                                         const void* cs_table, int32_t table_len, int32_t grid_w, float k_scale, void* rowstat_ws,
                                         void* out, void* stream) {
    LT_REQUIRE(A && W && qkv && q_ln_w && q_ln_b && k_ln_w && k_ln_b && cs_table && rowstat_ws && out, "lt_op_qkv_attention_small: null pointer");
    LT_REQUIRE(H > 0 && Hkv > 0 && hd > 0 && tokens > 0 && M > 0 && M % tokens == 0, "lt_op_qkv_attention_small: bad shape");
    const int d = H * hd, dkv = Hkv * hd, N = d + 2 * dkv;
    LT_REQUIRE(attention_small_fusable(hd, tokens, H, Hkv, d, dkv), "lt_op_qkv_attention_small: head_dim 48, 64 <= tokens <= 512 in whole tiles, widths %% 128 == 0");
    GemmArgs g;
    g.A = (const u16*)A; g.W = (const u16*)W; g.C = (u16*)qkv; g.bias = nullptr; g.bias_dtype = -1; g.M = M; g.N = N; g.K = K;
    g.lda = K; g.ldw = K; g.ldc = N;
    LT_REQUIRE(gemm_is_small_m(g, 0), "lt_op_qkv_attention_small: %d x %d x %d does not run on the small-M tiles", M, N, K);
    g.rowstat = (float*)rowstat_ws; g.rowstat_slots = (N + 127) / 128;  // rowstat_ws: [M][ceil(N / 128)] float2
    if (launch_gemm_bf16(g, 0, 0, (hipStream_t)stream)) return 1;
    AttnSmallArgs a;
    a.qkv = (const u16*)qkv; a.ld = N; a.q_col0 = 0; a.k_col0 = d; a.v_col0 = d + dkv;
    a.rowstat = (const float*)rowstat_ws; a.slots = g.rowstat_slots; a.q_slot0 = 0; a.q_nslot = d / 128; a.k_slot0 = d / 128; a.k_nslot = dkv / 128;
    a.q_ln_w = (const u16*)q_ln_w; a.q_ln_b = (const u16*)q_ln_b; a.k_ln_w = (const u16*)k_ln_w; a.k_ln_b = (const u16*)k_ln_b; a.ln_eps = 1e-5f;
    a.cs = (const float*)cs_table; a.t = nullptr; a.watershed = 0.f; a.cs_len = table_len; a.grid_w = grid_w;  // branch 1
    a.k_scale = k_scale; a.out = (u16*)out; a.B = M / tokens; a.H = H; a.Hkv = Hkv; a.N = tokens; a.hd = hd;
    return launch_attention_small(a, (hipStream_t)stream);
}

extern "C" int lt_op_attention_fused(const void* q, const void* k, const void* vt, const void* tk, const void* tvt,
                                     const float* tbias, const void* tgate, void* out, int32_t B, int32_t H, int32_t Hkv, int32_t N,
                                     int32_t Nk, int32_t Nkpad, int32_t Tk, int32_t Tkpad, int32_t hd, void* stream) {
    LT_REQUIRE(q && k && vt && tk && tvt && tbias && tgate && out, "lt_op_attention_fused: null pointer");
    LT_REQUIRE(attention_fuses_text(hd), "lt_op_attention_fused: needs head_dim 72 or 96 and attention_variant 3 or 4");
    AttnArgs a;
    a.q = (const u16*)q; a.k = (const u16*)k; a.vt = (const u16*)vt; a.bias = nullptr; a.out = (u16*)out; a.gate = nullptr;
    a.accumulate = 0; a.B = B; a.H = H; a.Hkv = Hkv; a.N = N; a.Nk = Nk; a.Nkpad = Nkpad; a.hd = hd; a.scale = 1.f; a.k_prescaled = 1;
    a.tk = (const u16*)tk; a.tvt = (const u16*)tvt; a.tbias = tbias; a.tgate = (const u16*)tgate; a.Tk = Tk; a.Tkpad = Tkpad;
    return launch_attention(a, (hipStream_t)stream);
}

extern "C" int lt_op_attention_trace(const void* q, const void* k, const void* vt, void* out, int32_t B, int32_t H, int32_t Hkv,
                                     int32_t N, int32_t Nk, int32_t Nkpad, int32_t hd, float scale, void* trace_dev,
                                     void* stream) {
    LT_REQUIRE(q && k && vt && out && trace_dev, "lt_op_attention_trace: null pointer");
    AttnArgs a;
    a.q = (const u16*)q; a.k = (const u16*)k; a.vt = (const u16*)vt; a.bias = nullptr; a.out = (u16*)out;
    a.gate = nullptr; a.accumulate = 0; a.B = B; a.H = H; a.Hkv = Hkv; a.N = N; a.Nk = Nk;
    a.Nkpad = Nkpad; a.hd = hd; a.scale = scale; a.trace = (unsigned long long*)trace_dev;
    return launch_attention(a, (hipStream_t)stream);
}

extern "C" int lt_op_linear_small_m(const void* a, const void* w, const void* b, void* y, int32_t M, int32_t N, int32_t K,
                                    int32_t act_in, void* stream) {
    LT_REQUIRE(a && w && y, "lt_op_linear_small_m: null pointer");
    return launch_linear_small_m((const u16*)a, (const u16*)w, (const u16*)b, (u16*)y, M, N, K, act_in, (hipStream_t)stream);
}

extern "C" int lt_op_rope_table_2d_pair(void* out, void* out_t, int32_t len, int32_t hd, float theta, float scale_factor, void* stream) {
    LT_REQUIRE(out && out_t, "lt_op_rope_table_2d_pair: null pointer");
    return launch_rope_table_2d((float*)out, len, hd, theta, scale_factor, (hipStream_t)stream, (float*)out_t);
}

extern "C" int lt_op_rope_table_2d(void* out, int32_t len, int32_t hd, float theta, float scale_factor, void* stream) {
    LT_REQUIRE(out, "lt_op_rope_table_2d: null pointer");
    return launch_rope_table_2d((float*)out, len, hd, theta, scale_factor, (hipStream_t)stream);
}
