// q/k/v post-processing between the fused QKV GEMM and the attention kernel.
//
//  qk_norm_rope : full-width affine LayerNorm (the reference's "qk_norm", model.py:211-215, :361-362;
//                 fp32 under autocast) -> rotary embedding in fp32 (model.py:254-282; table layout
//                 model.py:915-963: complex slot 2i rotates with the ROW position, slot 2i+1 with the
//                 COLUMN position, both at frequency i) -> one bf16 rounding (model.py:371) -> head-major
//                 [B, heads, N, hd] so an attention K tile is one contiguous 64*hd*2-byte run.
//                 The 42 MB complex table the reference rebuilds on every call (model.py:883-889) is
//                 replaced by a [pos][hd/4] (cos,sin) table per branch; the watershed branch is picked
//                 on the device from t[0] (no .item() sync).
//  v_transpose  : V -> [B, kvh, hd, Npad] with keys permuted inside each group of 16 (quads 1 and 2
//                 swapped) - exactly the order in which a lane of the swapped QK^T MFMA holds its P
//                 values, so the PV MFMA needs no cross-lane exchange (see attention.hip).
#include "common.h"
#include "kernels.h"
#include "options.h"
#include "tile_order.h"
#include <algorithm>

namespace {

// QK_ROWS consecutive token rows per workgroup (4 waves x 2 rows).  Per workgroup the LayerNorm weight / bias and the
// rotary factors of its rows are staged in LDS once: read per row from L1 they were 2 x the row's own bytes (weights) plus
// one 8-byte table load per bf16 pair, which made the kernel TA-bound (2.7 TB/s) instead of HBM-bound.
// In the head-major destination one head's 8+ consecutive rows are whole 128-byte lines (16 * hd bytes, hd % 8 == 0), so no
// line is shared between workgroups (= between XCD L2s).
// (4 waves per workgroup: the persistent kernels hold the current and the next row in registers - ~150 VGPRs, 3 waves per SIMD - so
//  three 4-wave workgroups fit a CU where only one 8-wave workgroup did: 12 instead of 8 streaming waves per CU)
constexpr int QK_ROWS = 8, QK_WAVES = 4, QK_RPW = QK_ROWS / QK_WAVES;
constexpr int QK_WG_PER_CU = 3;

// LDS image: ln_w | ln_b (bf16, `width` each) | (cos, sin)[row][complex slot] fp32
template <int MAXCH>
__device__ __forceinline__ void qk_norm_rope_block(const QkPostArgs& p, int bid, char* smem) {
    const int rows = p.B * p.N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int width = p.heads * p.hd;
    const int nch = width >> 3;
    const int cph = p.hd >> 3;   // chunks per head
    const int nslot = p.hd >> 1;  // complex slots per head
    const int row0 = bid * QK_ROWS;
    u16* sw = (u16*)smem;
    u16* sb = sw + width;
    float2* st = (float2*)(smem + (size_t)width * 4);

    // this wave's rows: issue the global loads first, they are the long-latency part
    bf8_t raw[QK_RPW][MAXCH];
#pragma unroll
    for (int r = 0; r < QK_RPW; ++r) {
        const int row = row0 + wave * QK_RPW + r;
        const u16* src = p.src + (size_t)(row < rows ? row : rows - 1) * p.ld_src + p.col0;
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nch) raw[r][i] = *(const bf8_t*)(src + c * 8);
            else raw[r][i].w[0] = raw[r][i].w[1] = raw[r][i].w[2] = raw[r][i].w[3] = 0u;
        }
    }
    if (p.ln_w) {
        for (int c = tid; c < nch; c += 64 * QK_WAVES) {
            *(bf8_t*)(sw + c * 8) = *(const bf8_t*)(p.ln_w + c * 8);
            *(bf8_t*)(sb + c * 8) = *(const bf8_t*)(p.ln_b + c * 8);
        }
    }
    if (p.rope_mode != 0) {
        // rotary table: branch 0 = linear interpolation (t < watershed), branch 1 = NTK (model.py:944-949)
        int branch = 1;
        if (p.t) branch = (p.t[0] < p.watershed) ? 0 : 1;
        const int nfreq = (p.rope_mode == 1) ? (p.hd >> 2) : (p.hd >> 1);
        const float* cs = p.cs + (size_t)branch * p.cs_len * nfreq * 2;
        for (int i = tid; i < QK_ROWS * nslot; i += 64 * QK_WAVES) {
            const int r = i / nslot, pr = i - r * nslot;
            const int row = row0 + r;
            if (row < rows) {
                const int b = row / p.N, n = row - b * p.N;
                const int n_rot = p.n_tok_b ? min(n, p.n_tok_b[b] - 1) : n;
                int pos, fi;
                if (p.rope_mode == 1) {  // complex slot 2i rotates with the ROW position, 2i+1 with the COLUMN position
                    const int gw = p.grid_w_b ? p.grid_w_b[b] : p.grid_w;
                    const int gr = n_rot / gw, gc = n_rot - gr * gw;
                    fi = pr >> 1; pos = (pr & 1) ? gc : gr;
                } else { fi = pr; pos = n_rot; }
                st[i] = *(const float2*)(cs + ((size_t)pos * nfreq + fi) * 2);
            }
        }
    }
    __syncthreads();

#pragma unroll
    for (int r = 0; r < QK_RPW; ++r) {
        const int row = row0 + wave * QK_RPW + r;
        if (row >= rows) continue;  // wave-uniform
        const int b = row / p.N, n = row - b * p.N;
        float mean = 0.f, rstd = 1.f;
        if (p.ln_w) {  // nn.LayerNorm over the full projection width, fp32 (model.py:211-215, :361-362)
            f32x2 s2 = {0.f, 0.f};
#pragma unroll
            for (int i = 0; i < MAXCH; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) s2 += unpk_bf(raw[r][i].w[k]);
            mean = wave_sum(s2[0] + s2[1]) / (float)width;
            const f32x2 mv = {mean, mean};
            f32x2 q2 = {0.f, 0.f};
#pragma unroll
            for (int i = 0; i < MAXCH; ++i) {
                const int c = lane + 64 * i;
                if (c < nch) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const f32x2 dl = unpk_bf(raw[r][i].w[k]) - mv;
                        q2 = dl * dl + q2;
                    }
                }
            }
            rstd = rsqrtf(wave_sum(q2[0] + q2[1]) / (float)width + p.ln_eps);
        }
        const f32x2 mv = {mean, mean}, rv = {rstd, rstd}, osc = {p.out_scale, p.out_scale};
        const float2* strow = st + (wave * QK_RPW + r) * nslot;
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nch) {
                const int head = c / cph, ci = c - head * cph;
                bf8_t wv, bv, o;
                if (p.ln_w) {
                    wv = *(const bf8_t*)(sw + c * 8);
                    bv = *(const bf8_t*)(sb + c * 8);
                }
                float4 t01 = {1.f, 0.f, 1.f, 0.f}, t23 = {1.f, 0.f, 1.f, 0.f};
                if (p.rope_mode != 0) {
                    t01 = *(const float4*)(strow + 4 * ci);
                    t23 = *(const float4*)(strow + 4 * ci + 2);
                }
                const float tc[4] = {t01.x, t01.z, t23.x, t23.z}, ts[4] = {t01.y, t01.w, t23.y, t23.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {  // one complex slot = one bf16 pair
                    f32x2 y = unpk_bf(raw[r][i].w[j]);
                    if (p.ln_w) y = (y - mv) * rv * unpk_bf(wv.w[j]) + unpk_bf(bv.w[j]);
                    if (p.rope_mode != 0) y = f32x2{y[0] * tc[j] - y[1] * ts[j], y[0] * ts[j] + y[1] * tc[j]};
                    if (p.out_scale != 1.0f) y = y * osc;
                    o.w[j] = pk_bf(y);
                }
                *(bf8_t*)(p.dst + (((size_t)b * p.heads + head) * p.N + n) * p.hd + ci * 8) = o;
            }
        }
    }
}

template <int MAXCH>
__global__ __launch_bounds__(64 * QK_WAVES) void qk_norm_rope_kernel(QkPostArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    qk_norm_rope_block<MAXCH>(p, blockIdx.x, smem_raw);
}

// Persistent form (stand-alone launches): about two workgroups per CU, each wave walks rows with a grid stride and keeps the
// NEXT row's loads in flight while it works on the current one.  With one workgroup per 16 rows the whole grid was resident at
// once and ran in lock step - everybody loads, then everybody computes, then everybody stores - which left the memory system
// idle between the phases (2.8 TB/s); staggered waves stream continuously.  LayerNorm weights are staged once per workgroup;
// each wave stages its own row's rotary factors (36 x 8 bytes) in its private LDS strip.
template <int MAXCH>
__device__ __forceinline__ void qk_norm_rope_persistent_body(const QkPostArgs& p, int block, int nblocks, char* smem) {
    const int rows = p.B * p.N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int width = p.heads * p.hd;
    const int nch = width >> 3;
    const int cph = p.hd >> 3;
    const int nslot = p.hd >> 1;
    u16* sw = (u16*)smem;
    u16* sb = sw + width;
    float2* st = (float2*)(smem + (size_t)width * 4) + wave * nslot;  // this wave's strip
    const int stride = nblocks * QK_WAVES;
    int row = block * QK_WAVES + wave;

    auto load_row = [&](int r, bf8_t (&raw)[MAXCH]) __attribute__((always_inline)) {
        const u16* src = p.src + (size_t)(r < rows ? r : rows - 1) * p.ld_src + p.col0;
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nch) raw[i] = *(const bf8_t*)(src + c * 8);
            else raw[i].w[0] = raw[i].w[1] = raw[i].w[2] = raw[i].w[3] = 0u;
        }
    };
    bf8_t cur[MAXCH], nxt[MAXCH];
    float2 pv_cur = (p.qstat_in && lane < p.qstat_slots) ? ((const float2*)p.qstat_in)[(size_t)(row < rows ? row : rows - 1) * p.qstat_slots + lane] : float2{0.f, 0.f};
    load_row(row, cur);
    if (p.ln_w) {
        for (int c = tid; c < nch; c += 64 * QK_WAVES) {
            *(bf8_t*)(sw + c * 8) = *(const bf8_t*)(p.ln_w + c * 8);
            *(bf8_t*)(sb + c * 8) = *(const bf8_t*)(p.ln_b + c * 8);
        }
    }
    int branch = 1;
    if (p.rope_mode != 0 && p.t) branch = (p.t[0] < p.watershed) ? 0 : 1;
    const int nfreq = (p.rope_mode == 1) ? (p.hd >> 2) : (p.hd >> 1);
    const float* cs = p.rope_mode != 0 ? p.cs + (size_t)branch * p.cs_len * nfreq * 2 : nullptr;
    __syncthreads();  // the only workgroup barrier: weights staged (every wave passes it exactly once)

    // (the Q partials of a row travel like the row itself: fetched one iteration ahead, in FRONT of the row's own loads - vmcnt retires
    //  in order, a load issued behind them could only be waited for together with them)
    auto load_stat = [&](int r) __attribute__((always_inline)) {
        return (p.qstat_in && lane < p.qstat_slots) ? ((const float2*)p.qstat_in)[(size_t)(r < rows ? r : rows - 1) * p.qstat_slots + lane] : float2{0.f, 0.f};
    };
    for (; row < rows; row += stride) {
        const float2 pv_nxt = load_stat(row + stride);
        load_row(row + stride, nxt);  // clamped inside; the extra load of the last iteration is discarded
        const int b = row / p.N, n = row - b * p.N;
        if (p.rope_mode != 0 && lane < nslot) {  // this row's rotary factors -> the wave's LDS strip
            const int pr = lane;
            const int n_rot = p.n_tok_b ? min(n, p.n_tok_b[b] - 1) : n;
            int pos, fi;
            if (p.rope_mode == 1) {
                const int gw = p.grid_w_b ? p.grid_w_b[b] : p.grid_w;
                const int gr = n_rot / gw, gc = n_rot - gr * gw;
                fi = pr >> 1; pos = (pr & 1) ? gc : gr;
            } else { fi = pr; pos = n_rot; }
            st[pr] = *(const float2*)(cs + ((size_t)pos * nfreq + fi) * 2);
        }
        if (p.qstat_in) {
            // the same row of Q: its LayerNorm partials (GemmArgs::qstat) -> (mean, rstd) for the attention prologue (AttnArgs::q_stat).
            // var = E[x^2] - mean^2 in fp32 over bf16 values of O(1): within an fp32 ulp or two of the two-pass form below
            const float inv_w = 1.0f / (float)p.qstat_width;
            const float qm = wave_sum(pv_cur.x) * inv_w;
            const float qr = rsqrtf(fmaxf(wave_sum(pv_cur.y) * inv_w - qm * qm, 0.f) + p.ln_eps);
            if (lane == 0) ((float2*)p.qstat_out)[row] = float2{qm, qr};
            pv_cur = pv_nxt;
        }
        float mean = 0.f, rstd = 1.f;
        if (p.ln_w) {
            f32x2 s2 = {0.f, 0.f};
#pragma unroll
            for (int i = 0; i < MAXCH; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) s2 += unpk_bf(cur[i].w[k]);
            mean = wave_sum(s2[0] + s2[1]) / (float)width;
            const f32x2 mv = {mean, mean};
            f32x2 q2 = {0.f, 0.f};
#pragma unroll
            for (int i = 0; i < MAXCH; ++i) {
                const int c = lane + 64 * i;
                if (c < nch) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const f32x2 dl = unpk_bf(cur[i].w[k]) - mv;
                        q2 = dl * dl + q2;
                    }
                }
            }
            rstd = rsqrtf(wave_sum(q2[0] + q2[1]) / (float)width + p.ln_eps);
        }
        const f32x2 mv = {mean, mean}, rv = {rstd, rstd}, osc = {p.out_scale, p.out_scale};
        __builtin_amdgcn_wave_barrier();  // the strip is written and read by this wave only: LDS ops of one wave stay in order
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nch) {
                const int head = c / cph, ci = c - head * cph;
                bf8_t wv, bv, o;
                if (p.ln_w) {
                    wv = *(const bf8_t*)(sw + c * 8);
                    bv = *(const bf8_t*)(sb + c * 8);
                }
                float4 t01 = {1.f, 0.f, 1.f, 0.f}, t23 = {1.f, 0.f, 1.f, 0.f};
                if (p.rope_mode != 0) {
                    t01 = *(const float4*)(st + 4 * ci);
                    t23 = *(const float4*)(st + 4 * ci + 2);
                }
                const float tc[4] = {t01.x, t01.z, t23.x, t23.z}, ts[4] = {t01.y, t01.w, t23.y, t23.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x2 y = unpk_bf(cur[i].w[j]);
                    if (p.ln_w) y = (y - mv) * rv * unpk_bf(wv.w[j]) + unpk_bf(bv.w[j]);
                    if (p.rope_mode != 0) y = f32x2{y[0] * tc[j] - y[1] * ts[j], y[0] * ts[j] + y[1] * tc[j]};
                    if (p.out_scale != 1.0f) y = y * osc;
                    o.w[j] = pk_bf(y);
                }
                *(bf8_t*)(p.dst + (((size_t)b * p.heads + head) * p.N + n) * p.hd + ci * 8) = o;
            }
        }
        __builtin_amdgcn_wave_barrier();  // all reads of the strip issued before the next row overwrites it
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) cur[i] = nxt[i];
    }
}

template <int MAXCH>
__global__ __launch_bounds__(64 * QK_WAVES) void qk_norm_rope_persistent_kernel(QkPostArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    qk_norm_rope_persistent_body<MAXCH>(p, blockIdx.x, gridDim.x, smem);
}

// q AND k post-processing of one layer in ONE persistent launch: blocks [0, q_blocks) stream the q rows, the rest the k rows
// (two independent HBM-bound passes over the same GEMM output; one launch fills the chip once instead of twice - each launch
// spends its first and last few microseconds with the memory system half empty)
struct QkPost2Args {
    QkPostArgs q, k;
    int q_blocks;
};
template <int MAXCH>
__global__ __launch_bounds__(64 * QK_WAVES) void qk_norm_rope_pair_kernel(QkPost2Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.x < p.q_blocks) qk_norm_rope_persistent_body<MAXCH>(p.q, blockIdx.x, p.q_blocks, smem);
    else qk_norm_rope_persistent_body<MAXCH>(p.k, blockIdx.x - p.q_blocks, gridDim.x - p.q_blocks, smem);
}

// one block per (64-key tile, kv head, batch): V rows -> LDS (transposed, permuted) -> 128-byte rows
__device__ __forceinline__ void v_transpose_tile(const u16* __restrict__ src, int ld_src, int col0, u16* __restrict__ dst,
                                                 int N, int Npad, int kv_heads, int hd, int bx, int kvh, int b,
                                                 char* smem_raw) {
    u16* T = (u16*)smem_raw;  // [hd][72] (row stride 144 B: 16-B aligned, spreads banks)
    constexpr int LDT = 72;
    const int n0 = bx * 64;
    const int cph = hd >> 3;
    const int nin = 64 * cph;
    for (int id = threadIdx.x; id < nin; id += blockDim.x) {
        const int tok = id / cph, ci = id - tok * cph;
        const int n = n0 + tok;
        bf8_t t;
        if (n < N) t = *(const bf8_t*)(src + ((size_t)b * N + n) * ld_src + col0 + kvh * hd + ci * 8);
        else { t.w[0] = t.w[1] = t.w[2] = t.w[3] = 0u; }
        // key position inside its group of 16: swap bit 2 and bit 3
        const int tp = (tok & ~12) | ((tok & 4) << 1) | ((tok & 8) >> 1);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            T[(ci * 8 + 2 * e) * LDT + tp] = (u16)(t.w[e] & 0xffffu);
            T[(ci * 8 + 2 * e + 1) * LDT + tp] = (u16)(t.w[e] >> 16);
        }
    }
    __syncthreads();
    const int nout = hd * 8;
    for (int id = threadIdx.x; id < nout; id += blockDim.x) {
        const int d = id >> 3, c = id & 7;
        const bf8_t t = *(const bf8_t*)(T + d * LDT + c * 8);
        *(bf8_t*)(dst + (((size_t)b * kv_heads + kvh) * hd + d) * Npad + n0 + c * 8) = t;
    }
}

__global__ __launch_bounds__(256) void v_transpose_kernel(const u16* __restrict__ src, int ld_src, int col0,
                                                          u16* __restrict__ dst, int N, int Npad, int kv_heads,
                                                          int hd) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    v_transpose_tile(src, ld_src, col0, dst, N, Npad, kv_heads, hd, blockIdx.x, blockIdx.y, blockIdx.z, smem_raw);
}

// q post-processing, k post-processing and the V transpose of one layer in ONE launch (three independent, HBM-bound
// passes over the QKV GEMM output): workgroups [0, nq) take q rows, [nq, nq + nk) k rows, the rest V tiles, so the three
// streams overlap instead of running back to back with two launch boundaries in between.
template <int MAXCH>
__global__ __launch_bounds__(64 * QK_WAVES) void qkv_post_kernel(QkvPostArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    int bid = blockIdx.x;
    if (bid >= p.pf.first) {  // rider workgroups (QkvPostArgs::pf)
        prefetch_w_block(p.pf, bid - p.pf.first);
        return;
    }
    if (bid < p.nq_blocks) {
        qk_norm_rope_block<MAXCH>(p.q, bid, smem_raw);
        return;
    }
    bid -= p.nq_blocks;
    if (bid < p.nk_blocks) {
        qk_norm_rope_block<MAXCH>(p.k, bid, smem_raw);
        return;
    }
    bid -= p.nk_blocks;
    const int nx = p.v_Npad / 64;
    if (bid >= nx * p.v_kv_heads * p.v_B) return;  // (padding blocks in front of rider workgroups)
    v_transpose_tile(p.v_src, p.v_ld_src, p.v_col0, p.v_dst, p.v_N, p.v_Npad, p.v_kv_heads, p.v_hd, bid % nx,
                     (bid / nx) % p.v_kv_heads, bid / (nx * p.v_kv_heads), smem_raw);
}

}  // namespace

// argument checks shared by the single and the pair launcher
static int validate_qk_post(const QkPostArgs& a, const char* who) {
    const int width = a.heads * a.hd;
    LT_REQUIRE(a.hd % 8 == 0 && width <= 64 * 8 * 8, "%s: hd %% 8 == 0 and heads*hd <= 4096 required (got %d x %d)", who, a.heads, a.hd);
    LT_REQUIRE(a.ld_src % 8 == 0 && a.col0 % 8 == 0, "%s: ld_src/col0 must be multiples of 8", who);
    LT_REQUIRE(a.rope_mode == 0 || a.cs != nullptr, "%s: rotary table missing", who);
    LT_REQUIRE(a.rope_mode != 1 || (a.hd % 4 == 0 && a.grid_w > 0), "%s: 2-D rope needs hd %% 4 == 0 and grid_w > 0 (got hd %d, grid_w %d)", who, a.hd, a.grid_w);
    LT_REQUIRE((a.ln_w == nullptr) == (a.ln_b == nullptr), "%s: LayerNorm weight and bias must come together", who);
    LT_REQUIRE(a.qstat_in == nullptr || (a.qstat_out && a.qstat_slots > 0 && a.qstat_slots <= 64 && a.qstat_width > 0),
               "%s: q-stat reduction needs an output, 1..64 slots and the Q width", who);
    return 0;
}

int launch_qk_norm_rope(const QkPostArgs& a, hipStream_t stream) {
    if (int rc = validate_qk_post(a, "qk_norm_rope")) return rc;
    const int width = a.heads * a.hd;
    const int rows = a.B * a.N;
    // persistent grid: ~2 workgroups (16 waves) per CU, never more workgroups than rows / 8
    const int cus = num_cus();
    const int max_blocks = (rows + QK_WAVES - 1) / QK_WAVES;
    const dim3 grid(std::min(max_blocks, lt_opt(OPT_QK_WG_PER_CU) * cus));
    const size_t smem = (size_t)width * 4 + (size_t)QK_WAVES * (a.hd >> 1) * 8;
    switch (((width >> 3) + 63) / 64) {
        case 1: hipLaunchKernelGGL(qk_norm_rope_persistent_kernel<1>, grid, dim3(64 * QK_WAVES), smem, stream, a); break;
        case 2: hipLaunchKernelGGL(qk_norm_rope_persistent_kernel<2>, grid, dim3(64 * QK_WAVES), smem, stream, a); break;
        case 3: hipLaunchKernelGGL(qk_norm_rope_persistent_kernel<3>, grid, dim3(64 * QK_WAVES), smem, stream, a); break;
        case 4: hipLaunchKernelGGL(qk_norm_rope_persistent_kernel<4>, grid, dim3(64 * QK_WAVES), smem, stream, a); break;
        case 5: hipLaunchKernelGGL(qk_norm_rope_persistent_kernel<5>, grid, dim3(64 * QK_WAVES), smem, stream, a); break;
        case 6: hipLaunchKernelGGL(qk_norm_rope_persistent_kernel<6>, grid, dim3(64 * QK_WAVES), smem, stream, a); break;
        default: hipLaunchKernelGGL(qk_norm_rope_persistent_kernel<8>, grid, dim3(64 * QK_WAVES), smem, stream, a); break;
    }
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_qk_norm_rope_pair(const QkPostArgs& q, const QkPostArgs& k, hipStream_t stream) {
    if (int rc = validate_qk_post(q, "qk_norm_rope_pair (q)")) return rc;
    if (int rc = validate_qk_post(k, "qk_norm_rope_pair (k)")) return rc;
    const int wq = q.heads * q.hd, wk = k.heads * k.hd;
    LT_REQUIRE(q.hd == k.hd && wk <= wq, "qk_norm_rope_pair: widths %d / %d unsupported", wq, wk);
    LT_REQUIRE(!q.qstat_in && !k.qstat_in, "qk_norm_rope_pair: the q-stat reduction rides on the single-stream launch only");
    const int cus = num_cus();
    // QK_WG_PER_CU workgroups per CU in total, split by the bytes of the two streams (GQA: the k stream is a quarter of the q stream)
    const int rows = q.B * q.N;
    const int total = std::min((2 * rows + QK_WAVES - 1) / QK_WAVES, QK_WG_PER_CU * cus);
    int qb = (int)((long long)total * wq / (wq + wk));
    qb = std::max(1, std::min(qb, total - 1));
    QkPost2Args a{q, k, qb};
    const dim3 grid(total);
    const size_t smem = (size_t)wq * 4 + (size_t)QK_WAVES * (q.hd >> 1) * 8;
    switch (((wq >> 3) + 63) / 64) {
        case 1: hipLaunchKernelGGL(qk_norm_rope_pair_kernel<1>, grid, dim3(64 * QK_WAVES), smem, stream, a); break;
        case 2: hipLaunchKernelGGL(qk_norm_rope_pair_kernel<2>, grid, dim3(64 * QK_WAVES), smem, stream, a); break;
        case 3: hipLaunchKernelGGL(qk_norm_rope_pair_kernel<3>, grid, dim3(64 * QK_WAVES), smem, stream, a); break;
        case 4: hipLaunchKernelGGL(qk_norm_rope_pair_kernel<4>, grid, dim3(64 * QK_WAVES), smem, stream, a); break;
        case 5: hipLaunchKernelGGL(qk_norm_rope_pair_kernel<5>, grid, dim3(64 * QK_WAVES), smem, stream, a); break;
        case 6: hipLaunchKernelGGL(qk_norm_rope_pair_kernel<6>, grid, dim3(64 * QK_WAVES), smem, stream, a); break;
        default: hipLaunchKernelGGL(qk_norm_rope_pair_kernel<8>, grid, dim3(64 * QK_WAVES), smem, stream, a); break;
    }
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_v_transpose(const u16* src, int ld_src, int col0, u16* dst, int B, int N, int Npad, int kv_heads, int hd,
                       hipStream_t stream) {
    LT_REQUIRE(hd % 8 == 0 && hd <= 128, "v_transpose: hd=%d unsupported", hd);
    LT_REQUIRE(Npad % 64 == 0 && Npad >= N, "v_transpose: Npad=%d must be a multiple of 64 and >= N=%d", Npad, N);
    LT_REQUIRE(ld_src % 8 == 0 && col0 % 8 == 0, "v_transpose: ld_src/col0 must be multiples of 8");
    dim3 grid(Npad / 64, kv_heads, B);
    hipLaunchKernelGGL(v_transpose_kernel, grid, dim3(256), hd * 72 * 2, stream, src, ld_src, col0, dst, N, Npad,
                       kv_heads, hd);
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_qkv_post(const QkvPostArgs& a0, hipStream_t stream) {
    QkvPostArgs a = a0;
    const int wq = a.q.heads * a.q.hd, wk = a.k.heads * a.k.hd;
    LT_REQUIRE(a.q.hd % 8 == 0 && wq <= 4096 && wk <= wq, "qkv_post: widths %d / %d unsupported", wq, wk);
    LT_REQUIRE(a.v_Npad % 64 == 0 && a.v_Npad >= a.v_N && a.v_hd <= 128, "qkv_post: bad V shape");
    LT_REQUIRE(!a.q.qstat_in && !a.k.qstat_in, "qkv_post: the q-stat reduction rides on the single-stream launch only");
    a.nq_blocks = (a.q.B * a.q.N + QK_ROWS - 1) / QK_ROWS;
    a.nk_blocks = (a.k.B * a.k.N + QK_ROWS - 1) / QK_ROWS;
    const int nv = (a.v_Npad / 64) * a.v_kv_heads * a.v_B;
    int nblk = a.nq_blocks + a.nk_blocks + nv;
    if (a.pf.blocks > 0) {  // riders behind the working blocks, from a multiple of 8 on (block index mod 8 = XCD); the blocks in between exit
        a.pf.first = (nblk + 7) / 8 * 8;
        nblk = a.pf.first + a.pf.blocks;
    } else {
        a.pf.first = 0x7fffffff;
    }
    const dim3 grid(nblk);
    const size_t smem = std::max<size_t>((size_t)a.v_hd * 72 * 2, (size_t)wq * 4 + (size_t)QK_ROWS * (a.q.hd >> 1) * 8);
    switch (((wq >> 3) + 63) / 64) {
        case 1: hipLaunchKernelGGL(qkv_post_kernel<1>, grid, dim3(64 * QK_WAVES), smem, stream, a); break;
        case 2: hipLaunchKernelGGL(qkv_post_kernel<2>, grid, dim3(64 * QK_WAVES), smem, stream, a); break;
        case 3: hipLaunchKernelGGL(qkv_post_kernel<3>, grid, dim3(64 * QK_WAVES), smem, stream, a); break;
        case 4: hipLaunchKernelGGL(qkv_post_kernel<4>, grid, dim3(64 * QK_WAVES), smem, stream, a); break;
        case 5: hipLaunchKernelGGL(qkv_post_kernel<5>, grid, dim3(64 * QK_WAVES), smem, stream, a); break;
        case 6: hipLaunchKernelGGL(qkv_post_kernel<6>, grid, dim3(64 * QK_WAVES), smem, stream, a); break;
        default: hipLaunchKernelGGL(qkv_post_kernel<8>, grid, dim3(64 * QK_WAVES), smem, stream, a); break;
    }
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}
