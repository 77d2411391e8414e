// options.h: the option table, the process defaults and the thread-local engine scope (host code only).
#include <atomic>
#include <cstring>

#include "common.h"
#include "kernels.h"
#include "options.h"

// name, lowest, highest, default, boolean.  include/lumina_dit_debug.h documents what each one selects (tests/test_abi.py keeps the two in step).
const LtOptDesc kLtOptDesc[LT_OPT_COUNT] = {
    {"graph", 0, 2, 2, false},
    {"attention_variant", 1, 6, 4, false},
    {"qkv_post_fused", 0, 2, 2, false},
    {"qkv_vt_epilogue", 0, 1, 1, true},
    {"qkv_fused_gemm", 0, 1, 1, true},
    {"qk_post_pair", 0, 1, 1, true},
    {"attn_q_fused", 0, 1, 1, true},
    {"norm_specialize", 0, 1, 1, true},
    {"gemm_w4q", 0, 1, 1, true},
    {"gemm_prefetch", 0, 3, 3, false},
    {"gemm_splitk", 0, 2, 1, false},
    {"gemm_w4q_grouped", 0, 2, 1, false},
    {"gemm_group", 0, 64, 0, false},
    {"gemm_stagger", 0, 256, 0, false},
    {"gemm_variant", 0, 2, 0, false},
    {"rmsnorm_apex", 0, 1, 0, true},
    {"attn_small_fused", 0, 1, 1, true},
    {"moe_route_fused", 0, 1, 1, true},
    {"gemm_splitk4", 0, 1, 1, true},
    {"moe_time_plan_hoist", 0, 1, 1, true},
    {"grn_ystat", 0, 2, 1, false},
    {"qk_wg_per_cu", 1, 8, 3, false},
    {"prologue_fused", 0, 7, 0, false},
    {"gemm_tail_split", 0, 2, 0, false},
    {"attn_text_skip", 0, 1, 1, true},
    {"attn_tail_split", 0, 4, 4, false},
    {"pair_layout", 0, 1, 1, true},
};

namespace {
struct ProcessOptions {
    std::atomic<int> v[LT_OPT_COUNT];
    std::atomic<int> gen{0};
    ProcessOptions() { for (int i = 0; i < LT_OPT_COUNT; ++i) v[i].store(kLtOptDesc[i].def, std::memory_order_relaxed); }
};
ProcessOptions& process() {
    static ProcessOptions p;
    return p;
}
thread_local const LtOptScope* tl_scope = nullptr;
}  // namespace

int lt_opt_find(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < LT_OPT_COUNT; ++i)
        if (strcmp(name, kLtOptDesc[i].name) == 0) return i;
    return -1;
}

int lt_opt(int id) {
    if (tl_scope) return tl_scope->v[id];
    return process().v[id].load(std::memory_order_relaxed);
}

int lt_opt_generation() { return tl_scope ? tl_scope->process_gen : process().gen.load(std::memory_order_relaxed); }
int lt_opt_engine_generation() { return tl_scope ? tl_scope->engine_gen : 0; }

int lt_opt_validate(int id, int* value) {
    const LtOptDesc& d = kLtOptDesc[id];
    if (d.boolean) { *value = *value != 0; return 0; }
    if (id == OPT_ATTENTION_VARIANT && *value == 5) {
        lt_set_error("attention_variant 5 (PV on 16x16x32 MFMAs) was a study kernel of csrc/experimental/, removed in round 5 (it lost on issue slots, NOTEBOOK.md 5.8)");
        return 2;
    }
    if (id == OPT_GEMM_PREFETCH && *value == 2) {
        lt_set_error("gemm_prefetch 2 (weight-panel reads on a side stream) lost 33 %% in its A/B and was removed in round 5; use 0, 1 or 3");
        return 2;
    }
    if (*value < d.lo || *value > d.hi) {
        lt_set_error("option %s must be %d .. %d (got %d)", d.name, d.lo, d.hi, *value);
        return 2;
    }
    return 0;
}

void lt_opt_set_process(int id, int value) {
    process().v[id].store(value, std::memory_order_relaxed);
    process().gen.fetch_add(1, std::memory_order_release);
}

void lt_opt_reset_process() {
    for (int i = 0; i < LT_OPT_COUNT; ++i) process().v[i].store(kLtOptDesc[i].def, std::memory_order_relaxed);
    process().gen.fetch_add(1, std::memory_order_relaxed);
}

LtOptScope::LtOptScope(const LtEngineOptions* o) : prev(tl_scope) {
    // generations first: a setter stores the value, then bumps the generation - a snapshot can carry a newer value under an older
    // generation (the next call then re-captures its graph), never an older value under a newer one
    process_gen = process().gen.load(std::memory_order_acquire);
    engine_gen = o ? o->gen.load(std::memory_order_acquire) : 0;
    for (int i = 0; i < LT_OPT_COUNT; ++i) {
        const int ov = o ? o->v[i].load(std::memory_order_relaxed) : LT_OPT_INHERIT;
        v[i] = ov != LT_OPT_INHERIT ? ov : process().v[i].load(std::memory_order_relaxed);
    }
    tl_scope = this;
}
LtOptScope::~LtOptScope() { tl_scope = prev; }
