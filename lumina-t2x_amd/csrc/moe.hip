// Mixture-of-experts routing for the Next-DiT-MoE family (Next-DiT-MoE/models/models2.py:451-506, BASELINE configs[4]).
//
// The reference loops over experts on the host: `batch_idx, nth = torch.where(selected == i)` (a device sync per
// expert) and `results[batch_idx] += w * expert(x[batch_idx])`.  Here routing stays on the device and the expert FFNs
// run as ONE grouped SwiGLU GEMM + ONE grouped W2 GEMM over an expert-sorted copy of the rows:
//   route   : router logits (bf16-rounded, as nn.Linear under autocast) -> top-2 (lowest index wins ties) -> fp32
//             softmax over the two selected logits -> bf16 weights (:464-470 / :493-499)
//   plan    : per-expert counts -> segments aligned to the GEMM's 256-row tiles, a row position for every
//             (token, expert) pair, and the tile -> expert table the grouped GEMM reads (single workgroup scan, no atomics
//             -> bit-reproducible)
//   (gather : round 3 - folded into the grouped SwiGLU GEMM, which reads its A rows through the plan's inverse map src[])
//   combine : out[token] = bf16(bf16(0 + bf16(w_a y_a)) + bf16(w_b y_b)), experts in ascending id = the order of the
//             reference's `for i, expert in enumerate(self.experts)` loop (:472-476) - round 3: formed inside the
//             gated_residual_norm launch that consumes the branch (norm.hip, MOE = true), no kernel of its own
// TimeMoeLayer routes on the timestep embedding, so all tokens of a sample share the two experts; SpaceMoeLayer
// routes every token on its own FFN input.  Both go through the same four kernels.
#include "common.h"
#include "kernels.h"
#include "moe_route.h"

namespace {

constexpr int MAX_E = LT_MOE_MAX_E;
constexpr int TILE = 256;

__global__ __launch_bounds__(256) void moe_route_kernel(MoeArgs p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    float logit[MAX_E];
    if (p.sample_logits) {
        const int b = row / p.rows_per_sample;
#pragma unroll
        for (int e = 0; e < MAX_E; ++e) logit[e] = e < p.E ? bf2f(p.sample_logits[b * p.sample_ld + e]) : -INFINITY;
    } else {
        float acc[MAX_E];
#pragma unroll
        for (int e = 0; e < MAX_E; ++e) acc[e] = 0.f;
        const int nch = p.d >> 3;
        const u16* xr = p.x + (size_t)row * p.d;
        for (int c = lane; c < nch; c += 64) route_accumulate(*(const bf8_t*)(xr + c * 8), p.gate_w, p.E, p.d, c, acc);
#pragma unroll
        for (int e = 0; e < MAX_E; ++e) logit[e] = e < p.E ? bfr(wave_sum(acc[e])) : -INFINITY;  // nn.Linear output in bf16
    }
    if (lane == 0) {
        int s0, s1;
        u16 w0, w1;
        top2_route(logit, p.forced ? p.forced + 2 * row : nullptr, s0, s1, w0, w1);
        p.sel[2 * row] = s0;
        p.sel[2 * row + 1] = s1;
        p.wts[2 * row] = w0;
        p.wts[2 * row + 1] = w1;
    }
}

// single workgroup: entries (row, k) in row-major order keep their order inside each expert segment (bit-reproducible, no atomics).
// Round 3: the per-expert exclusive scan over the 1024 threads' counts is a wave-level shuffle scan on PACKED counters (four 16-bit
// fields per 64-bit word: a thread holds <= 32 entries, a wave <= 2048 per expert) plus a 16-step scan over the wave totals; the
// round-1 form let E threads walk all 1024 counters serially through LDS and took 12.4 us per MoE layer - 0.4 ms per NFE at cfg 5,
// as much as the attention of that model (profiles/r03/rocprofv3_kernel_stats_cfg5_r03.csv).  Also writes the inverse map
// src[sorted position] = token row (-1 in the padding of a segment) that the grouped SwiGLU GEMM gathers its A rows through.
// Round 4: PER > 0 = the thread's (at most PER, a multiple of 4) entries live in REGISTERS, fetched with 16-byte loads that are all
// in flight together, and pos[] leaves as 16-byte stores.  The round-3 form walked its entries with one dependent 4-byte load per
// iteration, twice (count, then place): 2 x 16 serial memory round trips of one CU = 62 us per MoE FFN at 16 384 entries, 2 ms per NFE of
// Next-DiT-MoE at 1024^2 (profiles/r04/rocprofv3_kernel_stats_cfg5_1024_baseline.csv).  PER == 0 keeps that walk for larger problems.
template <int PER>
__global__ __launch_bounds__(1024) void moe_plan_kernel(MoeArgs p) {
    __shared__ int wtot[16][MAX_E];   // per wave and expert: entries in the wave, then the exclusive prefix over waves
    __shared__ int seg_off[MAX_E], seg_cnt[MAX_E];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = p.rows * 2;
    const int per = PER > 0 ? PER : (n + 1023) / 1024;
    const int lo = tid * per, hi = min(n, lo + per);
    unsigned long long c0 = 0, c1 = 0;  // this thread's counts: experts 0..3 / 4..7, 16 bits each
    int ex_r[PER > 0 ? PER : 1];        // PER > 0: the thread's entries (-1 past the end)
    if (p.sample_logits) {  // time router: the logits are per sample - route here, no separate launch (sel / wts written for combine)
        int cached_b = -1, s0 = 0, s1 = 0;
        u16 w0 = 0, w1 = 0;
        auto route_row = [&](int row) __attribute__((always_inline)) {
            const int b = row / p.rows_per_sample;
            if (b != cached_b || p.forced) {  // every token of a sample shares the two experts (unless the parity hook forces per-row choices)
                float logit[MAX_E];
#pragma unroll
                for (int e = 0; e < MAX_E; ++e) logit[e] = e < p.E ? bf2f(p.sample_logits[b * p.sample_ld + e]) : -INFINITY;
                top2_route(logit, p.forced ? p.forced + 2 * row : nullptr, s0, s1, w0, w1);
                cached_b = b;
            }
        };
        if constexpr (PER > 0) {
            // lo is even: the thread's entries are whole rows (entry 2 r = the row's first expert, 2 r + 1 its second) -> one 8-byte
            // store of the pair's experts and one 4-byte store of its two bf16 weights per row; static register indices (a loop over
            // [lo, hi) indexed ex_r dynamically, which put the array into scratch: 50 us per call)
#pragma unroll
            for (int j = 0; j < PER; j += 2) {
                if (lo + j < hi) {
                    route_row((lo + j) >> 1);
                    ex_r[j] = s0; ex_r[j + 1] = s1;
                    *(int2*)(p.sel + lo + j) = int2{s0, s1};
                    *(unsigned*)(p.wts + lo + j) = (unsigned)w0 | ((unsigned)w1 << 16);
                } else {
                    ex_r[j] = -1; ex_r[j + 1] = -1;
                }
            }
        } else {
            for (int i = lo; i < hi; ++i) {
                route_row(i >> 1);
                p.sel[i] = (i & 1) ? s1 : s0;
                p.wts[i] = (i & 1) ? w1 : w0;
            }
            __syncthreads();  // (sel is re-read below by the thread that wrote it)
        }
    } else if constexpr (PER > 0) {
        // n is even and lo a multiple of 4, but n need not be a multiple of 4: the last quad may run 2 entries past the end of sel -
        // the engine's table has room (capacity rows), the op-level entry pads its buffer; values past hi are masked below
#pragma unroll
        for (int j = 0; j < PER; j += 4) {
            int4 v = {-1, -1, -1, -1};
            if (lo + j < hi) v = *(const int4*)(p.sel + lo + j);
            ex_r[j] = v.x; ex_r[j + 1] = v.y; ex_r[j + 2] = v.z; ex_r[j + 3] = v.w;
        }
#pragma unroll
        for (int j = 0; j < PER; ++j) if (lo + j >= hi) ex_r[j] = -1;
    }
    if constexpr (PER > 0) {
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int ex = ex_r[j];
            if (ex >= 0) {
                if (ex < 4) c0 += 1ull << (16 * ex);
                else c1 += 1ull << (16 * (ex - 4));
            }
        }
    } else {
        for (int i = lo; i < hi; ++i) {
            const int ex = p.sel[i];
            if (ex < 4) c0 += 1ull << (16 * ex);
            else c1 += 1ull << (16 * (ex - 4));
        }
    }
    unsigned long long s0 = c0, s1 = c1;  // inclusive scan inside the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long u0 = __shfl_up(s0, d, 64), u1 = __shfl_up(s1, d, 64);
        if (lane >= d) { s0 += u0; s1 += u1; }
    }
    if (lane == 63) {
#pragma unroll
        for (int e = 0; e < MAX_E; ++e) wtot[wave][e] = (int)(((e < 4 ? s0 : s1) >> (16 * (e & 3))) & 0xffff);
    }
    __syncthreads();
    if (tid < p.E) {  // exclusive scan of this expert's 16 wave totals
        int run = 0;
        for (int w = 0; w < 16; ++w) {
            const int c = wtot[w][tid];
            wtot[w][tid] = run;
            run += c;
        }
        seg_cnt[tid] = run;
    }
    __syncthreads();
    if (tid == 0) {
        int off = 0;
        for (int e = 0; e < p.E; ++e) {
            seg_off[e] = off;
            off += (seg_cnt[e] + TILE - 1) / TILE * TILE;
        }
    }
    __syncthreads();
    int next[MAX_E];
#pragma unroll
    for (int e = 0; e < MAX_E; ++e) {
        const int incl = (int)(((e < 4 ? s0 : s1) >> (16 * (e & 3))) & 0xffff), own = (int)(((e < 4 ? c0 : c1) >> (16 * (e & 3))) & 0xffff);
        next[e] = e < p.E ? seg_off[e] + wtot[wave][e] + incl - own : 0;
    }
    if constexpr (PER > 0) {
#pragma unroll
        for (int j = 0; j < PER; j += 4) {
            int q4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ex = ex_r[j + u];
                int q = 0;
#pragma unroll
                for (int e = 0; e < MAX_E; ++e) {
                    if (ex == e) { q = next[e]; next[e] = q + 1; }
                }
                q4[u] = q;
                if (ex >= 0) p.src[q] = (lo + j + u) >> 1;
            }
            if (lo + j + 3 < hi) *(int4*)(p.pos + lo + j) = int4{q4[0], q4[1], q4[2], q4[3]};
            else {
#pragma unroll
                for (int u = 0; u < 4; ++u) if (lo + j + u < hi) p.pos[lo + j + u] = q4[u];
            }
        }
    } else {
        for (int i = lo; i < hi; ++i) {
            const int ex = p.sel[i];
            int q = 0;
#pragma unroll
            for (int e = 0; e < MAX_E; ++e) {
                if (ex == e) { q = next[e]; next[e] = q + 1; }
            }
            p.pos[i] = q;
            p.src[q] = i >> 1;
        }
    }
    for (int t = tid; t < p.max_tiles; t += 1024) {
        int ex = -1;
        for (int e = 0; e < p.E; ++e) {
            const int t0 = seg_off[e] / TILE, t1 = (seg_off[e] + seg_cnt[e] + TILE - 1) / TILE;
            if (t >= t0 && t < t1) ex = e;
        }
        p.tile_expert[t] = ex;
    }
    // padding positions of the map: behind the last entry of every segment up to its tile boundary, and every tile no expert owns
    for (int q = tid; q < p.max_tiles * TILE; q += 1024) {
        bool real = false;
        for (int e = 0; e < p.E; ++e) real |= q >= seg_off[e] && q < seg_off[e] + seg_cnt[e];
        if (!real) p.src[q] = -1;
    }
}

// Time branch without the parity hook: every token of a sample shares its two experts, so the plan is closed form - expert e's segment
// holds, sample by sample, the N rows of every sample that selected e - and needs no scan: any number of workgroups, each thread one
// row (round 4: the single-workgroup kernel took 26 us per TimeMoeLayer at 8192 rows, this one a few).  Every workgroup recomputes the
// B routings and the E segment offsets (B x E bf16 loads), then writes its rows' sel / wts / pos / src and its slice of the padding
// positions and of the tile table.  Same outputs as moe_plan_kernel, bit for bit (tests/test_moe_plan.py).
constexpr int PLAN_T_MAXB = LT_MOE_PLAN_TIME_MAX_SAMPLES;
__global__ __launch_bounds__(256) void moe_plan_time_kernel(MoeArgs p) {
    __shared__ int s_sel[PLAN_T_MAXB][2], s_base[PLAN_T_MAXB][2], s_off[MAX_E], s_cnt[MAX_E];
    __shared__ u16 s_w[PLAN_T_MAXB][2];
    const int tid = threadIdx.x;
    const int N = p.rows_per_sample, B = p.rows / N;
    if (p.layers > 1) {  // blockIdx.y = layer: its E logit columns in, its own plan out
        const long long l = blockIdx.y;
        p.sample_logits += l * p.E;
        p.sel += l * p.layer_stride_rows; p.pos += l * p.layer_stride_rows; p.wts += l * p.layer_stride_rows;
        p.src += l * p.layer_stride_src; p.tile_expert += l * p.layer_stride_tiles;
    }
    if (tid < B) {
        float logit[MAX_E];
#pragma unroll
        for (int e = 0; e < MAX_E; ++e) logit[e] = e < p.E ? bf2f(p.sample_logits[tid * p.sample_ld + e]) : -INFINITY;
        int a, b;
        u16 wa, wb;
        top2_route(logit, nullptr, a, b, wa, wb);
        s_sel[tid][0] = a; s_sel[tid][1] = b; s_w[tid][0] = wa; s_w[tid][1] = wb;
    }
    __syncthreads();
    if (tid < p.E) {  // samples that selected expert `tid`
        int c = 0;
        for (int b = 0; b < B; ++b) c += (s_sel[b][0] == tid) + (s_sel[b][1] == tid);
        s_cnt[tid] = c * N;
    }
    __syncthreads();
    if (tid == 0) {
        int off = 0;
        for (int e = 0; e < p.E; ++e) { s_off[e] = off; off += (s_cnt[e] + TILE - 1) / TILE * TILE; }
    }
    __syncthreads();
    if (tid < B) {  // this sample's block inside each of its two experts' segments: behind the blocks of the samples before it
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int e = s_sel[tid][k];
            int before = 0;
            for (int b = 0; b < tid; ++b) before += (s_sel[b][0] == e) + (s_sel[b][1] == e);
            s_base[tid][k] = s_off[e] + before * N;
        }
    }
    __syncthreads();
    const int row = blockIdx.x * 256 + tid;
    if (row < p.rows) {
        const int b = row / N, j = row - b * N;
        const int q0 = s_base[b][0] + j, q1 = s_base[b][1] + j;
        *(int2*)(p.sel + 2 * row) = int2{s_sel[b][0], s_sel[b][1]};
        *(unsigned*)(p.wts + 2 * row) = (unsigned)s_w[b][0] | ((unsigned)s_w[b][1] << 16);
        *(int2*)(p.pos + 2 * row) = int2{q0, q1};
        p.src[q0] = row;
        p.src[q1] = row;
    }
    // padding positions and the tile table, sliced over the grid
    const int P = p.max_tiles * TILE;
    for (int q = blockIdx.x * 256 + tid; q < P; q += gridDim.x * 256) {
        bool real = false;
        for (int e = 0; e < p.E; ++e) real |= q >= s_off[e] && q < s_off[e] + s_cnt[e];
        if (!real) p.src[q] = -1;
    }
    for (int t = blockIdx.x * 256 + tid; t < p.max_tiles; t += gridDim.x * 256) {
        int ex = -1;
        for (int e = 0; e < p.E; ++e) {
            const int t0 = s_off[e] / TILE, t1 = (s_off[e] + s_cnt[e] + TILE - 1) / TILE;
            if (t >= t0 && t < t1) ex = e;
        }
        p.tile_expert[t] = ex;
    }
}

}  // namespace

static int check(const MoeArgs& a) {
    LT_REQUIRE(a.E >= 2 && a.E <= MAX_E, "moe: %d experts unsupported (2..%d, top-2 routing)", a.E, MAX_E);
    LT_REQUIRE(a.d % 8 == 0 && a.rows > 0 && a.rows_per_sample > 0, "moe: bad shape");
    LT_REQUIRE(a.max_tiles * TILE >= 2 * a.rows + a.E * (TILE - 1), "moe: sorted buffers too small for %d rows", a.rows);
    return 0;
}

int launch_moe_route(const MoeArgs& a, hipStream_t stream) {
    if (check(a)) return 2;
    LT_REQUIRE(a.gate_w != nullptr && a.sample_logits == nullptr, "moe_route: per-token router weights required (per-sample logits are routed by moe_plan)");
    hipLaunchKernelGGL(moe_route_kernel, dim3((a.rows + 3) / 4), dim3(256), 0, stream, a);
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_moe_plan(const MoeArgs& a_in, hipStream_t stream) {
    MoeArgs a = a_in;
    if (a.sample_ld == 0) a.sample_ld = a.E;
    if (check(a)) return 2;
    LT_REQUIRE(2LL * a.rows <= 1024LL * 1023, "moe_plan: %d rows exceed the packed 16-bit counters of the scan (523776 rows)", a.rows);
    const bool closed_form = a.sample_logits && !a.forced && a.rows % a.rows_per_sample == 0 && a.rows / a.rows_per_sample <= PLAN_T_MAXB;
    LT_REQUIRE(a.layers <= 1 || closed_form, "moe_plan: the all-layers form is the time router's closed-form plan (per-sample logits, no forced routing, <= %d samples)", PLAN_T_MAXB);
    if (closed_form) {
        hipLaunchKernelGGL(moe_plan_time_kernel, dim3((a.rows + 255) / 256, a.layers > 1 ? a.layers : 1), dim3(256), 0, stream, a);
        LT_CHECK_HIP(hipGetLastError());
        return 0;
    }
    const int per = (2 * a.rows + 1023) / 1024;  // entries per thread; the register forms hold a multiple of 4
    if (per <= 4) hipLaunchKernelGGL(moe_plan_kernel<4>, dim3(1), dim3(1024), 0, stream, a);
    else if (per <= 8) hipLaunchKernelGGL(moe_plan_kernel<8>, dim3(1), dim3(1024), 0, stream, a);
    else if (per <= 16) hipLaunchKernelGGL(moe_plan_kernel<16>, dim3(1), dim3(1024), 0, stream, a);
    else if (per <= 32) hipLaunchKernelGGL(moe_plan_kernel<32>, dim3(1), dim3(1024), 0, stream, a);
    else hipLaunchKernelGGL(moe_plan_kernel<0>, dim3(1), dim3(1024), 0, stream, a);
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}
