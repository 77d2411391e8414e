// Kernel-selection options of the engine (host side).  One table names every option with its range and default; a value is looked
// up in three places, first hit wins:
//   1. the per-engine override of the engine whose C-ABI call is running on this thread (lt_engine_set_option; LtOptScope),
//   2. the process default (lt_set_option),
//   3. the table's built-in default.
// No plain globals: the process defaults are one array behind accessor functions, the "current engine" is a thread-local pointer set
// for the duration of an entry point, so two engines with different settings can be driven from two threads ("no
// hidden global state").
#pragma once
#include <atomic>
#include <climits>

enum LtOpt {
    OPT_GRAPH = 0, OPT_ATTENTION_VARIANT, OPT_QKV_POST_FUSED, OPT_QKV_VT_EPILOGUE, OPT_QKV_FUSED_GEMM, OPT_QK_POST_PAIR, OPT_ATTN_Q_FUSED,
    OPT_NORM_SPECIALIZE, OPT_GEMM_W4Q, OPT_GEMM_PREFETCH, OPT_GEMM_SPLITK, OPT_GEMM_W4Q_GROUPED, OPT_GEMM_GROUP, OPT_GEMM_STAGGER,
    OPT_GEMM_VARIANT, OPT_RMSNORM_APEX, OPT_ATTN_SMALL_FUSED, OPT_MOE_ROUTE_FUSED, OPT_GEMM_SPLITK4, OPT_MOE_TIME_PLAN_HOIST,
    OPT_GRN_YSTAT, OPT_QK_WG_PER_CU, OPT_PROLOGUE_FUSED, OPT_GEMM_TAIL_SPLIT, OPT_ATTN_TEXT_SKIP, OPT_ATTN_TAIL_SPLIT, OPT_PAIR_LAYOUT,
    LT_OPT_COUNT
};
constexpr int LT_OPT_INHERIT = INT_MIN;  // per-engine slot: no override

struct LtOptDesc {
    const char* name;
    int lo, hi, def;
    bool boolean;  // any non-zero value means 1 (the historical behaviour of the on / off knobs)
};
extern const LtOptDesc kLtOptDesc[LT_OPT_COUNT];

int lt_opt_find(const char* name);                 // index into kLtOptDesc, -1 = unknown
int lt_opt(int id);                                // effective value on this thread
int lt_opt_generation();                           // bumped by every change of a process default (HIP-graph cache key); inside an LtOptScope: the value the scope saw at entry
int lt_opt_engine_generation();                    // LtEngineOptions::gen as the running scope saw it at entry (0 outside a scope / without an engine)
int lt_opt_validate(int id, int* value);           // 0 ok (booleans normalised), else lt_set_error was called
void lt_opt_set_process(int id, int value);
void lt_opt_reset_process();

// engine-side storage: v[i] == LT_OPT_INHERIT -> the process default applies.  Atomics: lt_engine_set_option may run on one thread while
// another thread is inside an entry point of the same engine - the running call keeps the snapshot it took at entry.
struct LtEngineOptions {
    std::atomic<int> v[LT_OPT_COUNT];
    std::atomic<int> gen{0};
    LtEngineOptions() { for (int i = 0; i < LT_OPT_COUNT; ++i) v[i].store(LT_OPT_INHERIT, std::memory_order_relaxed); }
};
// RAII: takes ONE snapshot of the effective values (engine override, else process default) at entry; the calling thread's lookups see that
// snapshot until the scope ends, so one evaluation runs on one consistent set whatever other threads set meanwhile (nests; restores the
// previous scope).
struct LtOptScope {
    int v[LT_OPT_COUNT];
    int process_gen, engine_gen;
    const LtOptScope* prev;
    explicit LtOptScope(const LtEngineOptions* o);
    ~LtOptScope();
    LtOptScope(const LtOptScope&) = delete;
    LtOptScope& operator=(const LtOptScope&) = delete;
};
