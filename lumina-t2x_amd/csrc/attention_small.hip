// Fused q / k post-processing + attention for SHORT sequences (round 5): the class-conditional 600M models at 256 tokens
// (Next-DiT-ImageNet/models/models.py:358-404, Next-DiT-MoE/models/models2.py:358-404: LayerNorm(q), LayerNorm(k), 2-D RoPE on q and k
// jointly, flash_attn_func with the default 1/sqrt(hd) scale).
//
// At 512 rows every kernel of a block is a latency-bound launch (profiles/r04/rocprofv3_kernel_stats_cfg1_head_slow_box.csv: q / k / v
// post-processing 13.7 us + attention 11.0 us for ~1 us of work each), and a launch boundary - or a grid barrier inside a persistent
// kernel, which costs the same L2 write-back / invalidate (MI355X_MICROARCH.md price list: boundary 1.5-1.9 us + ~4.5 us kernel floor vs
// barrier-xcd 4.8-7.2 us) - is the unit of cost.  So this kernel removes one: a workgroup (sample b, head h, 128 query rows) builds what
// it needs itself, straight from the QKV projection's row-major output:
//   * K of its (b, kv head): k_norm (affine LayerNorm over the FULL projection width in fp32; the row's sum / sum of squares arrive as
//     per-column-tile partials from the GEMM epilogue, GemmArgs::rowstat) -> RoPE in fp32 -> softmax scale * log2(e) folded in -> ONE
//     bf16 rounding, into an LDS image of all keys of the sample (N <= 512: 57 KB at hd 48, N = 512);
//   * V^T of the same head: transposed and key-permuted into the LDS tile image the PV MFMA reads (what v_transpose + the staging DMA
//     produce in two steps elsewhere);
//   * Q of its rows: q_norm + RoPE in registers, directly as MFMA fragments.
// The redundancy is small: every (b, kv head) is prepared by N / 128 workgroups (2 at 256 tokens), 96 bytes per key and operand.
// The attention loop itself is the round-1 structure (attention.hip: swapped QK^T so that softmax statistics are lane-local, P never
// leaves its lane, O^T = V^T P^T), reading its tiles from the resident images instead of a DMA ring.
// The arithmetic per element is that of qk_norm_rope (qkv_post.hip) and attn_fwd_kernel; the LayerNorm statistics come from (sum, sum
// of squares) instead of the two-pass form, as on the attn_q_fused path of the large models.
// Measured (profiles/r05): 14.4 us per launch at 2 x 32 heads x 256 tokens against 13.7 + 11.0 us for the two launches it replaces; cfg 1
// -3 ... -8 %, cfg 5 -6 ... -10 % same box, another -3 % on cfg 1 from requesting every prologue load before the first use.
#include "common.h"
#include "kernels.h"
#include "tile_order.h"

namespace {

template <int HD, int NW>
__global__ __launch_bounds__(64 * NW) void attn_small_fused_kernel(AttnSmallArgs p) {
    constexpr int KS = (HD + 15) / 16;  // QK^T k-steps
    constexpr int DT = (HD + 31) / 32;  // O^T row tiles
    constexpr int CPR = HD / 8;         // 16-byte chunks per head row
    constexpr int KROW = HD * 2 + 16;   // LDS bytes per K row (+16: consecutive rows start 28 / 40 / 52 dwords apart instead of 24 / 36 / 48)
    constexpr int VTILE = HD * 128;     // LDS bytes of a 64-key V^T tile
    constexpr int NT = 64 * NW;
    static_assert(HD % 16 == 0, "whole QK^T k-steps (hd 48 / 96)");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    if ((int)blockIdx.x >= p.pf.first) {  // rider workgroups (AttnSmallArgs::pf): the O projection's weight panels -> their XCDs' L2
        prefetch_w_block(p.pf, (int)blockIdx.x - p.pf.first);
        return;
    }
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    const int N = p.N;
    char* kimg = smem;
    char* vimg = smem + (size_t)N * KROW;
    float2* kst = (float2*)(vimg + (size_t)(N >> 6) * VTILE);

    constexpr int QB = 32 * NW;
    const int nqb = (N + QB - 1) / QB;
    const int BH = p.B * p.H;
    int bh, qb;
    if ((BH & 7) == 0) {  // head bh lives on XCD bh % 8, its q-blocks run back to back (they share K / V rows in that L2)
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        bh = xcd + 8 * (idx / nqb);
        qb = idx % nqb;
    } else {
        bh = blockIdx.x / nqb;
        qb = blockIdx.x % nqb;
    }
    if (bh >= BH) return;  // (grid padded to a multiple of 8 in front of the riders)
    const int b = bh / p.H, h = bh - b * p.H;
    const int hk = h / (p.H / p.Hkv);
    const size_t row_b0 = (size_t)b * N;

    int branch = 1;  // rotary table: branch 0 = linear interpolation (t < watershed), branch 1 = NTK (model.py:944-949); one table elsewhere
    if (p.t) branch = (p.t[0] < p.watershed) ? 0 : 1;
    constexpr int NFREQ = HD >> 2;
    const float* cs = p.cs + (size_t)branch * p.cs_len * NFREQ * 2;

    // ---- 0. the first batch of K / V rows of (b, hk) is requested before anything else: its latency hides behind steps 1 and 2 ----------
    // Items (key n, 8-channel chunk ci) in batches of BATCH per thread: every global load of a batch is issued before the first use.
    constexpr int BATCH = 6;
    const int items = N * CPR;
    bf8_t rk[BATCH], rvv[BATCH], kwv[BATCH], kbv[BATCH];
    float4 krf[BATCH], kcf[BATCH];
    int nn[BATCH], cc[BATCH];
    auto load_batch = [&](int base) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int id = min(base + u * NT, items - 1);  // (a surplus slot repeats the last item: same bytes to the same place)
            const int n = id / CPR, ci = id - n * CPR;
            nn[u] = n; cc[u] = ci;
            const u16* rowp = p.qkv + (row_b0 + n) * p.ld + hk * HD + ci * 8;
            rk[u] = *(const bf8_t*)(rowp + p.k_col0);
            rvv[u] = *(const bf8_t*)(rowp + p.v_col0);
            kwv[u] = *(const bf8_t*)(p.k_ln_w + hk * HD + ci * 8);
            kbv[u] = *(const bf8_t*)(p.k_ln_b + hk * HD + ci * 8);
            const int gr = n / p.grid_w, gc = n - gr * p.grid_w;
            krf[u] = *(const float4*)(cs + ((size_t)gr * NFREQ + 2 * ci) * 2);
            kcf[u] = *(const float4*)(cs + ((size_t)gc * NFREQ + 2 * ci) * 2);
        }
    };
    load_batch(tid);

    // ---- 0b. ... and the query row's own loads (B operand of S^T = K Q^T: lane = (q row l31, d = 16 s + 8 hi .. +8)) ----------------------
    int qrow = qb * QB + wave * 32 + l31;
    const bool q_ok = qrow < N;
    if (!q_ok) qrow = N - 1;
    bf16x8 qf[KS];
    // (every load of the query row is requested before any statistic is reduced: one memory round trip for the whole prologue)
    const int gr = qrow / p.grid_w, gc = qrow - gr * p.grid_w;
    const u16* src = p.qkv + (row_b0 + qrow) * p.ld + p.q_col0 + h * HD;
    bf8_t qraw[KS], qwv[KS], qbv[KS];
    float4 qrf[KS], qcf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int d0 = 16 * s + 8 * hi;  // < HD (HD % 16 == 0)
        qraw[s] = *(const bf8_t*)(src + d0);
        qwv[s] = *(const bf8_t*)(p.q_ln_w + h * HD + d0);
        qbv[s] = *(const bf8_t*)(p.q_ln_b + h * HD + d0);
        // complex slot pr = d0 / 2 + j rotates at frequency pr >> 1 with the ROW position (pr even) or the COLUMN position (pr odd)
        const int fi0 = d0 >> 2;
        qrf[s] = *(const float4*)(cs + ((size_t)gr * NFREQ + fi0) * 2);  // (cos, sin) at fi0, fi0 + 1 for the row
        qcf[s] = *(const float4*)(cs + ((size_t)gc * NFREQ + fi0) * 2);
    }

    // ---- 1. LayerNorm statistics of this sample's K rows -> LDS ------------------------------------------------------------
    // (loads in groups of four with the sums behind them: a loop that loads and adds one slot per iteration serialises on the memory
    //  latency - the first form of this kernel spent 10 of its 17 us in such loops, profiles/r05/rocprofv3_kernel_stats_cfg1_r05_before_*.csv)
    auto row_stat = [&](const float2* ps, int ns, float inv_w) __attribute__((always_inline)) {
        float s1 = 0.f, s2 = 0.f;
        for (int i = 0; i < ns; i += 4) {
            float2 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = i + u < ns ? ps[i + u] : float2{0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 4; ++u) { s1 += v[u].x; s2 += v[u].y; }
        }
        const float mean = s1 * inv_w;
        return float2{mean, rsqrtf(fmaxf(s2 * inv_w - mean * mean, 0.f) + p.ln_eps)};
    };
    for (int n = tid; n < N; n += NT)
        kst[n] = row_stat((const float2*)p.rowstat + (row_b0 + n) * p.slots + p.k_slot0, p.k_nslot, 1.0f / (float)(p.Hkv * HD));

    // ---- 2. q_norm + RoPE of the query fragments in registers (their loads were requested in step 0b) -----------------------------------
    {
        const float2 qst = row_stat((const float2*)p.rowstat + (row_b0 + qrow) * p.slots + p.q_slot0, p.q_nslot, 1.0f / (float)(p.H * HD));
        const f32x2 mv = {qst.x, qst.x}, rv = {qst.y, qst.y};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float tc[4] = {qrf[s].x, qcf[s].x, qrf[s].z, qcf[s].z}, ts[4] = {qrf[s].y, qcf[s].y, qrf[s].w, qcf[s].w};
            bf8_t o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x2 y = (unpk_bf(qraw[s].w[j]) - mv) * rv * unpk_bf(qwv[s].w[j]) + unpk_bf(qbv[s].w[j]);
                y = f32x2{y[0] * tc[j] - y[1] * ts[j], y[0] * ts[j] + y[1] * tc[j]};
                o.w[j] = pk_bf(y);
            }
            qf[s] = __builtin_bit_cast(bf16x8, o);
        }
    }
    __syncthreads();  // kst complete

    // ---- 3. K image (k_norm + RoPE + scale, one rounding) and V^T tile image of (b, hk) -------------------------------------------
    {
        const f32x2 osc = {p.k_scale, p.k_scale};
        for (int base = tid; base < items; base += NT * BATCH) {
            if (base != tid) load_batch(base);
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int n = nn[u], ci = cc[u];
                const float2 st = kst[n];
                const f32x2 mv = {st.x, st.x}, rv = {st.y, st.y};
                const float tc[4] = {krf[u].x, kcf[u].x, krf[u].z, kcf[u].z}, ts[4] = {krf[u].y, kcf[u].y, krf[u].w, kcf[u].w};
                bf8_t o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x2 y = (unpk_bf(rk[u].w[j]) - mv) * rv * unpk_bf(kwv[u].w[j]) + unpk_bf(kbv[u].w[j]);
                    y = f32x2{y[0] * tc[j] - y[1] * ts[j], y[0] * ts[j] + y[1] * tc[j]};
                    o.w[j] = pk_bf(y * osc);
                }
                *(bf8_t*)(kimg + (size_t)n * KROW + ci * 16) = o;
                // V^T: key position inside its group of 16 with bits 2 and 3 swapped (the order in which a lane of the swapped QK^T MFMA holds
                // its P values, qkv_post.hip v_transpose); chunk c of row d sits in slot c ^ ((d >> 1) & 7) (the staging swizzle of attention.hip)
                const int tok = n & 63;
                const int tp = (tok & ~12) | ((tok & 4) << 1) | ((tok & 8) >> 1);
                char* vt = vimg + (size_t)(n >> 6) * VTILE + (tp & 7) * 2;
                const int c = tp >> 3;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int d = ci * 8 + 2 * e;  // d and d + 1 share (d >> 1)
                    const int slot = (c ^ ((d >> 1) & 7)) << 4;
                    *(u16*)(vt + d * 128 + slot) = (u16)(rvv[u].w[e] & 0xffffu);
                    *(u16*)(vt + (d + 1) * 128 + slot) = (u16)(rvv[u].w[e] >> 16);
                }
            }
        }
    }
    __syncthreads();

    // ---- 4. attention over the resident tiles (attention.hip attn_fwd_kernel, K pre-scaled: scores in the log2 domain) ---------------
    int koff[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) koff[s] = l31 * KROW + (2 * s + hi) * 16;
    int voff[DT][4];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        int d = dt * 32 + l31;
        if (d > HD - 1) d = HD - 1;  // rows past hd are never stored
#pragma unroll
        for (int g = 0; g < 4; ++g) voff[dt][g] = d * 128 + (((2 * g + hi) ^ ((d >> 1) & 7)) << 4);
    }
    f32x16 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -1.0e30f, l_run = 0.f;
    const int ntile = N >> 6;
    for (int t = 0; t < ntile; ++t) {
        const char* kb = kimg + (size_t)t * 64 * KROW;
        const char* vb = vimg + (size_t)t * VTILE;
        // S^T sub-tiles: lane holds keys 32 kt2 + (r & 3) + 8 (r >> 2) + 4 hi for its query row
        f32x16 sc[2];
#pragma unroll
        for (int kt2 = 0; kt2 < 2; ++kt2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[kt2][r] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8 kf = *(const bf16x8*)(kb + kt2 * 32 * KROW + koff[s]);
                sc[kt2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], sc[kt2], 0, 0, 0);
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[kt2][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(sc[kt2][r] - m_new);
                sc[kt2][r] = e;
                psum += e;
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        // O^T += V^T P^T: group g = keys 16 g .. 16 g + 15, the P fragment is this lane's own 8 values
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bf16x8 pf;
#pragma unroll
            for (int e = 0; e < 8; ++e) pf[e] = (__bf16)sc[g >> 1][8 * (g & 1) + e];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const bf16x8 vf = *(const bf16x8*)(vb + voff[dt][g]);
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[dt], 0, 0, 0);
            }
        }
    }

    // ---- 5. epilogue: lane holds out[q = l31][d = 32 dt + 8 q4 + 4 hi + j] ---------------------------------------------------
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q_ok) {
        u16* orow = p.out + (row_b0 + qrow) * ((size_t)p.H * HD) + (size_t)h * HD;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int d0 = 32 * dt + 8 * q4 + 4 * hi;
                if (d0 < HD) {
                    const u32x2 v = {pack2bf(o[dt][4 * q4] * inv, o[dt][4 * q4 + 1] * inv), pack2bf(o[dt][4 * q4 + 2] * inv, o[dt][4 * q4 + 3] * inv)};
                    *(u32x2*)(orow + d0) = v;
                }
            }
    }
}

}  // namespace

// a call the fused kernel takes (the engine and lt_op_qkv_attention_small ask before they route a layer here)
bool attention_small_fusable(int hd, int N, int H, int Hkv, int q_width, int k_width) {
    return hd == 48 && N >= 64 && N <= 512 && N % 64 == 0 && Hkv > 0 && H % Hkv == 0 && q_width % 128 == 0 && k_width % 128 == 0;
}

int launch_attention_small(const AttnSmallArgs& a_in, hipStream_t stream) {
    AttnSmallArgs a = a_in;
    LT_REQUIRE(attention_small_fusable(a.hd, a.N, a.H, a.Hkv, a.H * a.hd, a.Hkv * a.hd),
               "attention_small: head_dim 48, 64 <= tokens <= 512 in whole 64-key tiles, projection widths in whole 128-column tiles (hd %d, N %d, H %d / %d)",
               a.hd, a.N, a.H, a.Hkv);
    LT_REQUIRE(a.qkv && a.rowstat && a.q_ln_w && a.q_ln_b && a.k_ln_w && a.k_ln_b && a.cs && a.out, "attention_small: null pointer");
    LT_REQUIRE(a.grid_w > 0 && a.cs_len > 0 && a.ld % 8 == 0 && a.q_col0 % 8 == 0 && a.k_col0 % 8 == 0 && a.v_col0 % 8 == 0,
               "attention_small: bad layout arguments");
    LT_REQUIRE(a.q_nslot == a.H * a.hd / 128 && a.k_nslot == a.Hkv * a.hd / 128 && a.q_slot0 + a.q_nslot <= a.slots && a.k_slot0 + a.k_nslot <= a.slots,
               "attention_small: the row-statistics slots do not cover the q / k widths");
    LT_REQUIRE((a.N - 1) / a.grid_w < a.cs_len && a.grid_w <= a.cs_len, "attention_small: token grid exceeds the RoPE table (%d)", a.cs_len);
    constexpr int NW = 4;  // 4 waves x 32 query rows (64-row workgroups - twice the workgroups, the K / V^T images built twice as often - measured 4 % slower,
                           // profiles/r05/bench_ab_moe_time_tiles_and_attn_small_64row_workgroups_both_lose.log)
    const int nqb = (a.N + 32 * NW - 1) / (32 * NW);
    int nblk = a.B * a.H * nqb;
    if (a.pf.blocks > 0) {  // riders behind the attention blocks, from a multiple of 8 on (block index mod 8 = XCD)
        a.pf.first = (nblk + 7) / 8 * 8;
        nblk = a.pf.first + a.pf.blocks;
    } else {
        a.pf.first = 0x7fffffff;
    }
    const int smem = a.N * (48 * 2 + 16) + (a.N / 64) * 48 * 128 + a.N * 8;
    if (ensure_dynamic_lds((const void*)attn_small_fused_kernel<48, NW>, smem)) return 1;
    hipLaunchKernelGGL((attn_small_fused_kernel<48, NW>), dim3(nblk), dim3(64 * NW), smem, stream, a);
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}
