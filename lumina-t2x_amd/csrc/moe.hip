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
//   gather  : xs[pos] = x[token]
//   combine : out[token] = bf16(bf16(0 + bf16(w_a y_a)) + bf16(w_b y_b)), experts in ascending id = the order of the
//             reference's `for i, expert in enumerate(self.experts)` loop (:472-476)
// TimeMoeLayer routes on the timestep embedding, so all tokens of a sample share the two experts; SpaceMoeLayer
// routes every token on its own FFN input.  Both go through the same four kernels.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int MAX_E = 8;
constexpr int TILE = 256;

__global__ __launch_bounds__(256) void moe_route_kernel(MoeArgs p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    float logit[MAX_E];
    if (p.sample_logits) {
        const int b = row / p.rows_per_sample;
#pragma unroll
        for (int e = 0; e < MAX_E; ++e) logit[e] = e < p.E ? bf2f(p.sample_logits[b * p.E + e]) : -INFINITY;
    } else {
        float acc[MAX_E];
#pragma unroll
        for (int e = 0; e < MAX_E; ++e) acc[e] = 0.f;
        const int nch = p.d >> 3;
        const u16* xr = p.x + (size_t)row * p.d;
        for (int c = lane; c < nch; c += 64) {
            float xf[8];
            unpack8(*(const bf8_t*)(xr + c * 8), xf);
#pragma unroll
            for (int e = 0; e < MAX_E; ++e) {
                if (e < p.E) {
                    float wf[8];
                    unpack8(*(const bf8_t*)(p.gate_w + (size_t)e * p.d + c * 8), wf);
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[e] += xf[i] * wf[i];
                }
            }
        }
#pragma unroll
        for (int e = 0; e < MAX_E; ++e) logit[e] = e < p.E ? bfr(wave_sum(acc[e])) : -INFINITY;  // nn.Linear output in bf16
    }
    if (lane == 0) {
        int i1 = 0;
#pragma unroll
        for (int e = 1; e < MAX_E; ++e) if (logit[e] > logit[i1]) i1 = e;
        int i2 = i1 == 0 ? 1 : 0;
#pragma unroll
        for (int e = 0; e < MAX_E; ++e) if (e != i1 && e != i2 && logit[e] > logit[i2]) i2 = e;
        if (p.forced) {  // the discrete choice comes from outside (a reference run's); the weights stay this run's own arithmetic
            i1 = p.forced[2 * row];
            i2 = p.forced[2 * row + 1];
        }
        // softmax over (v1, v2) in fp32, then the cast back to the activation dtype (:466-470)
        const float ex = __expf(logit[i2] - logit[i1]);
        const float w1 = 1.0f / (1.0f + ex), w2 = ex / (1.0f + ex);
        const bool swap = i2 < i1;  // accumulate in ascending expert id
        p.sel[2 * row] = swap ? i2 : i1;
        p.sel[2 * row + 1] = swap ? i1 : i2;
        p.wts[2 * row] = f2bf(swap ? w2 : w1);
        p.wts[2 * row + 1] = f2bf(swap ? w1 : w2);
    }
}

// single workgroup: entries (row, k) in row-major order keep their order inside each expert segment
__global__ __launch_bounds__(1024) void moe_plan_kernel(MoeArgs p) {
    __shared__ int cnt[1024][MAX_E + 1];  // +1: avoid the 8-way bank alias of an 8-int row stride
    __shared__ int seg_off[MAX_E], seg_cnt[MAX_E];
    const int tid = threadIdx.x;
    const int n = p.rows * 2;
    const int per = (n + 1023) / 1024;
    const int lo = tid * per, hi = min(n, lo + per);
    int mine[MAX_E];
#pragma unroll
    for (int e = 0; e < MAX_E; ++e) mine[e] = 0;
    for (int i = lo; i < hi; ++i) {
        const int ex = p.sel[i];
#pragma unroll
        for (int e = 0; e < MAX_E; ++e) mine[e] += (ex == e);
    }
#pragma unroll
    for (int e = 0; e < MAX_E; ++e) cnt[tid][e] = mine[e];
    __syncthreads();
    if (tid < p.E) {  // exclusive scan of this expert's per-thread counts (1024 serial adds: ~1 us, once per MoE layer)
        int run = 0;
        for (int t = 0; t < 1024; ++t) {
            const int c = cnt[t][tid];
            cnt[t][tid] = run;
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
    for (int e = 0; e < MAX_E; ++e) next[e] = e < p.E ? seg_off[e] + cnt[tid][e] : 0;
    for (int i = lo; i < hi; ++i) {
        const int ex = p.sel[i];
        int q = 0;
#pragma unroll
        for (int e = 0; e < MAX_E; ++e) {
            if (ex == e) { q = next[e]; next[e] = q + 1; }
        }
        p.pos[i] = q;
    }
    for (int t = tid; t < p.max_tiles; t += 1024) {
        int ex = -1;
        for (int e = 0; e < p.E; ++e) {
            const int t0 = seg_off[e] / TILE, t1 = (seg_off[e] + seg_cnt[e] + TILE - 1) / TILE;
            if (t >= t0 && t < t1) ex = e;
        }
        p.tile_expert[t] = ex;
    }
}

__global__ __launch_bounds__(256) void moe_gather_kernel(MoeArgs p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const int nch = p.d >> 3;
    const u16* src = p.x + (size_t)row * p.d;
    u16* d0 = p.xs + (size_t)p.pos[2 * row] * p.d;
    u16* d1 = p.xs + (size_t)p.pos[2 * row + 1] * p.d;
    for (int c = lane; c < nch; c += 64) {
        const bf8_t v = *(const bf8_t*)(src + c * 8);
        *(bf8_t*)(d0 + c * 8) = v;
        *(bf8_t*)(d1 + c * 8) = v;
    }
}

__global__ __launch_bounds__(256) void moe_combine_kernel(MoeArgs p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const int nch = p.d >> 3;
    const u16* y0 = p.ys + (size_t)p.pos[2 * row] * p.d;
    const u16* y1 = p.ys + (size_t)p.pos[2 * row + 1] * p.d;
    const float w0 = bf2f(p.wts[2 * row]), w1 = bf2f(p.wts[2 * row + 1]);
    u16* dst = p.out + (size_t)row * p.d;
    for (int c = lane; c < nch; c += 64) {
        float a[8], b[8], o[8];
        unpack8(*(const bf8_t*)(y0 + c * 8), a);
        unpack8(*(const bf8_t*)(y1 + c * 8), b);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = bfr(bfr(0.0f + bfr(w0 * a[i])) + bfr(w1 * b[i]));
        *(bf8_t*)(dst + c * 8) = pack8(o);
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
    LT_REQUIRE((a.gate_w != nullptr) != (a.sample_logits != nullptr), "moe_route: exactly one of gate_w / sample_logits");
    hipLaunchKernelGGL(moe_route_kernel, dim3((a.rows + 3) / 4), dim3(256), 0, stream, a);
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_moe_plan(const MoeArgs& a, hipStream_t stream) {
    if (check(a)) return 2;
    hipLaunchKernelGGL(moe_plan_kernel, dim3(1), dim3(1024), 0, stream, a);
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_moe_gather(const MoeArgs& a, hipStream_t stream) {
    if (check(a)) return 2;
    hipLaunchKernelGGL(moe_gather_kernel, dim3((a.rows + 3) / 4), dim3(256), 0, stream, a);
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_moe_combine(const MoeArgs& a, hipStream_t stream) {
    if (check(a)) return 2;
    hipLaunchKernelGGL(moe_combine_kernel, dim3((a.rows + 3) / 4), dim3(256), 0, stream, a);
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}
