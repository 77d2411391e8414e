// Row-wise normalisation kernels of the Next-DiT / Flag-DiT block (HBM-bound, one wave64 per token row).
//
//  rmsnorm_mod          : modulate(RMSNorm(x), scale[, shift])          (model.py:28-29, :599, :608;
//                         components.py:40-54 vanilla RMSNorm rounding: fp32 normalise -> bf16 -> * w)
//  gated_residual_norm  : x += tanh(gate) * RMSNorm(y) fused with the NEXT branch's pre-norm+modulate
//                         (model.py:597-610), or with the final layer's affine-free LayerNorm
//                         (model.py:634-638, :660).  One pass over x and y instead of ~10 eager kernels.
//
// Rows are held in registers between the statistics pass and the apply pass (16-byte loads, 8 bf16 per
// lane per chunk; wave shuffles for the reductions; no LDS, no barrier).
#include "common.h"
#include "kernels.h"
#include "moe_route.h"
#include "options.h"
#include "tile_order.h"

namespace {

constexpr int MAXCH_LIMIT = 8;  // 16-byte chunks per lane -> d <= 64 * 8 * 8 = 4096; kernels are compiled per chunk count

// A row lives in registers as raw bf16 pairs (4 dwords per 16-byte chunk); all arithmetic is pairwise
// (v_pk_*_f32 + one v_cvt_pk_bf16_f32 per rounded pair, common.h) - these kernels do a bf16 rounding after every
// reference op and were VALU-bound with scalar math.
template <int MAXCH>
struct RowRaw {
    bf8_t c[MAXCH];
};

typedef __attribute__((ext_vector_type(4))) unsigned nt_u32x4;
template <int MAXCH>
__device__ __forceinline__ void load_row(const u16* row, int nch, int lane, RowRaw<MAXCH>& r) {
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) r.c[i] = *(const bf8_t*)(row + ch * 8);
        else r.c[i].w[0] = r.c[i].w[1] = r.c[i].w[2] = r.c[i].w[3] = 0u;
    }
}

// sum of squares of a row held as bf16 pairs: v_dot2_f32_bf16 (acc + a.lo * a.lo + a.hi * a.hi, products of bf16 are exact in fp32)
// - one instruction per pair instead of two unpacks and a packed fma; four accumulators keep the dependent chains short
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
template <int MAXCH>
__device__ __forceinline__ float row_sumsq(const RowRaw<MAXCH>& r) {
    float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < MAXCH; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bf16x2_t v = __builtin_bit_cast(bf16x2_t, r.c[i].w[k]);
            s[k] = __builtin_amdgcn_fdot2_f32_bf16(v, v, s[k], false);
        }
    return wave_sum((s[0] + s[1]) + (s[2] + s[3]));
}

// h = bfr(bfr(bfr(x * r) * w) * bfr(1 + scale)) (+ shift)   -- each step optional as in the reference;
// scale_pre: `scale` already holds bfr(1 + scale).  apex (option rmsnorm_apex; a literal 0 from the specialised instantiations, so the
// test folds away there): bfr(x * r * w) - the weight multiplies in fp32 before the one rounding (SURVEY.md 8c's description of
// apex.FusedRMSNorm; DESIGN.md 6 on why the default order is expected to match an apex box as well)
template <int MAXCH>
__device__ __forceinline__ void apply_rms_mod_store(const RowRaw<MAXCH>& r, float rinv, const u16* w, const u16* scale,
                                                    const u16* shift, int scale_pre, u16* out, int nch, int lane, int apex = 0,
                                                    const u16* route_w = nullptr, int route_E = 0, int d = 0, float* route_acc = nullptr, int pair = 0) {
    const f32x2 rv = {rinv, rinv};
    const f32x2 one = {1.f, 1.f};
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) {
            bf8_t wv, sv, hv, o;
            if (w) wv = *(const bf8_t*)(w + ch * 8);
            if (scale) sv = *(const bf8_t*)(scale + ch * 8);
            if (shift) hv = *(const bf8_t*)(shift + ch * 8);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                f32x2 n = unpk_bf(r.c[i].w[k]) * rv;
                if (w) n = (apex ? n : bfr2(n)) * unpk_bf(wv.w[k]);
                if (scale) n = bfr2(n) * (scale_pre ? unpk_bf(sv.w[k]) : bfr2(one + unpk_bf(sv.w[k])));
                if (shift) n = bfr2(n) + unpk_bf(hv.w[k]);
                o.w[k] = pk_bf(n);
            }
            *(bf8_t*)(out + ch * 8 + (pair ? (ch >> 2) * 32 : 0)) = o;  // (pair: `out` = the row's base in the row-pair-interleaved layout)
            if (route_w) {  // (a literal nullptr from every instantiation but the MoE ones: folds away)
                float (&ra)[LT_MOE_MAX_E] = *reinterpret_cast<float (*)[LT_MOE_MAX_E]>(route_acc);
                route_accumulate(o, route_w, route_E, d, ch, ra);
            }
        }
    }
}

template <int MAXCH>
__global__ __launch_bounds__(256) void rmsnorm_mod_kernel(NormModArgs p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const int b = row / p.rows_per_batch;
    const int nch = p.d >> 3;
    RowRaw<MAXCH> r;
    load_row(p.x + (size_t)row * p.d, nch, lane, r);
    const float rinv = rsqrtf(row_sumsq(r) / (float)p.d + p.eps);
    apply_rms_mod_store(r, rinv, p.w, p.scale ? p.scale + (size_t)b * p.ld_mod : nullptr,
                        p.shift ? p.shift + (size_t)b * p.ld_mod : nullptr, p.scale_pre,
                        p.out_pair ? p.out + (size_t)(row >> 1) * (2 * p.d) + (row & 1) * 32 : p.out + (size_t)row * p.d, nch, lane, p.apex, nullptr, 0, 0, nullptr,
                        p.out_pair);
}

// PM / GM / NM >= 0: post_mode / gate_mode / next_mode fixed at compile time (the engine's combinations; selected by
// lt_set_option("norm_specialize", 1), OFF by default until measured) - same statements in the same order, the mode tests inside
// the per-chunk loops fold away.  -1: the mode is read from the arguments (the kernel as it has always been).
// MOE (round 3): the branch output y is not a buffer but the top-2 combine of the experts' outputs, done on the way in:
//   y[row] = bfr(bfr(0 + bfr(w_a ys[pos_a])) + bfr(w_b ys[pos_b]))   (ascending expert id = the reference's accumulation order,
// Next-DiT-MoE/models/models2.py:472-476) - what moe_combine_kernel wrote to `o` before, one launch and one [rows, d] round trip less
// per MoE layer.
template <int MAXCH, int PM = -1, int GM = -1, int NM = -1, bool MOE = false>
__global__ __launch_bounds__(256) void gated_residual_norm_kernel(GatedResArgs p) {
    const int post_mode = PM >= 0 ? PM : p.post_mode, gate_mode = GM >= 0 ? GM : p.gate_mode, next_mode = NM >= 0 ? NM : p.next_mode;
    const int apex = (PM >= 0 || MOE) ? 0 : p.apex;  // generic dense instantiations only (the launcher routes apex calls there)
    if ((int)blockIdx.x >= p.pf.first) {  // rider workgroups (GatedResArgs::pf): the next GEMM's weight panels -> their XCDs' L2
        prefetch_w_block(p.pf, (int)blockIdx.x - p.pf.first);
        return;
    }
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const int b = row / p.rows_per_batch;
    const int nch = p.d >> 3;
    u16* xrow = p.x + (size_t)row * p.d;
    RowRaw<MAXCH> r, xr;
    if constexpr (MOE) {
        RowRaw<MAXCH> ya, yb;
        load_row(p.moe_ys + (size_t)p.moe_pos[2 * row] * p.d, nch, lane, ya);
        load_row(p.moe_ys + (size_t)p.moe_pos[2 * row + 1] * p.d, nch, lane, yb);
        load_row(xrow, nch, lane, xr);
        const float w0 = bf2f(p.moe_wts[2 * row]), w1 = bf2f(p.moe_wts[2 * row + 1]);
        const f32x2 w0v = {w0, w0}, w1v = {w1, w1}, zero = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < MAXCH; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                r.c[i].w[k] = pk_bf(bfr2(zero + bfr2(w0v * unpk_bf(ya.c[i].w[k]))) + bfr2(w1v * unpk_bf(yb.c[i].w[k])));
    } else {
        load_row(p.y + (size_t)row * p.d, nch, lane, r);
        load_row(xrow, nch, lane, xr);  // issued with y: one memory round trip for both streams
    }
    float rinv = 1.f;
    if (post_mode == 1) rinv = rsqrtf(row_sumsq(r) / (float)p.d + p.eps);
    const u16* gate = p.gate ? p.gate + (size_t)b * p.ld_mod : nullptr;
    const f32x2 rv = {rinv, rinv};
    // x' = bfr(x + bfr(g * yn));  r <- x'
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) {
            bf8_t wv, gv;
            if (post_mode == 1) wv = *(const bf8_t*)(p.post_w + ch * 8);
            if (gate_mode != 2) gv = *(const bf8_t*)(gate + ch * 8);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                f32x2 yn = unpk_bf(r.c[i].w[k]);
                if (post_mode == 1) yn = bfr2((apex ? yn * rv : bfr2(yn * rv)) * unpk_bf(wv.w[k]));
                if (gate_mode == 1) {
                    const f32x2 g = unpk_bf(gv.w[k]);
                    yn = bfr2(bfr2(f32x2{tanhf(g[0]), tanhf(g[1])}) * yn);
                } else if (gate_mode == 0) {
                    yn = bfr2(unpk_bf(gv.w[k]) * yn);
                }
                r.c[i].w[k] = pk_bf(unpk_bf(xr.c[i].w[k]) + yn);
            }
            // the residual stream is not read again before the next row kernel (a GEMM or two and the attention later): stored
            // non-temporal, so that it does not push the pre-norm output - the next GEMM's A operand - out of the L2 / MALL
            // (step -0.2 %, four of four same-box pairs: profiles/r03/bench_ab_residual_stream_nontemporal_store.log)
            { const nt_u32x4 v = {r.c[i].w[0], r.c[i].w[1], r.c[i].w[2], r.c[i].w[3]}; __builtin_nontemporal_store(v, (nt_u32x4*)(xrow + ch * 8)); }
        }
    }
    if (next_mode == 0) return;
    const u16* nscale = p.next_scale ? p.next_scale + (size_t)b * p.ld_mod : nullptr;
    const u16* nshift = p.next_shift ? p.next_shift + (size_t)b * p.ld_mod : nullptr;
    const int hp = next_mode == 1 ? p.h_pair : 0;
    u16* hrow = hp ? p.h + (size_t)(row >> 1) * (2 * p.d) + (row & 1) * 32 : p.h + (size_t)row * p.d;
    if (next_mode == 1) {
        const float r2 = rsqrtf(row_sumsq(r) / (float)p.d + p.eps);
        if constexpr (MOE) {
            if (p.route_w) {  // the row is the input of a token-routed MoE layer: route it here (GatedResArgs::route_*), wave-uniform
                float racc[LT_MOE_MAX_E];
#pragma unroll
                for (int e = 0; e < LT_MOE_MAX_E; ++e) racc[e] = 0.f;
                apply_rms_mod_store(r, r2, p.next_w, nscale, nshift, p.scale_pre, hrow, nch, lane, apex, p.route_w, p.route_E, p.d, racc);
                float logit[LT_MOE_MAX_E];
#pragma unroll
                for (int e = 0; e < LT_MOE_MAX_E; ++e) logit[e] = e < p.route_E ? bfr(wave_sum(racc[e])) : -INFINITY;  // nn.Linear output in bf16
                if (lane == 0) {
                    int s0, s1;
                    u16 w0, w1;
                    top2_route(logit, p.route_forced ? p.route_forced + 2 * row : nullptr, s0, s1, w0, w1);
                    p.route_sel[2 * row] = s0;
                    p.route_sel[2 * row + 1] = s1;
                    // (the wave read moe_wts[2 row ..] of the branch it consumed at its start; nobody else reads this row's pair)
                    p.route_wts[2 * row] = w0;
                    p.route_wts[2 * row + 1] = w1;
                }
                return;
            }
        }
        apply_rms_mod_store(r, r2, p.next_w, nscale, nshift, p.scale_pre, hrow, nch, lane, apex, nullptr, 0, 0, nullptr, hp);
    } else {
        // affine-free LayerNorm in fp32, modulate in fp32, one rounding (the cast autocast applies at the
        // final Linear) -- model.py:634-638, :660-661
        f32x2 s2 = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < MAXCH; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) s2 += unpk_bf(r.c[i].w[k]);
        const float mean = wave_sum(s2[0] + s2[1]) / (float)p.d;
        const f32x2 mv = {mean, mean};
        f32x2 q2 = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            const int ch = lane + 64 * i;
            if (ch < nch) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f32x2 dlt = unpk_bf(r.c[i].w[k]) - mv;
                    q2 = dlt * dlt + q2;
                }
            }
        }
        const float rstd = rsqrtf(wave_sum(q2[0] + q2[1]) / (float)p.d + p.eps_next);
        const f32x2 rs = {rstd, rstd};
        const f32x2 one = {1.f, 1.f};
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            const int ch = lane + 64 * i;
            if (ch < nch) {
                bf8_t sv, hv, o;
                if (nscale) sv = *(const bf8_t*)(nscale + ch * 8);
                if (nshift) hv = *(const bf8_t*)(nshift + ch * 8);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    f32x2 n = (unpk_bf(r.c[i].w[k]) - mv) * rs;
                    if (nscale) n = n * (p.scale_pre ? unpk_bf(sv.w[k]) : bfr2(one + unpk_bf(sv.w[k])));
                    if (nshift) n = n + unpk_bf(hv.w[k]);
                    o.w[k] = pk_bf(n);
                }
                *(bf8_t*)(hrow + ch * 8) = o;
            }
        }
    }
}

// ---- round 6: the streaming form of the dense sandwich-norm step (post_mode 1, gate_mode 0, next_mode 1) ---------------------------
// rocprofv3 had the kernel above at 33.9 us for 151 MB at cfg 2 (4.45 TB/s) with ~600 VALU instructions per row on the hot path: not
// issue-bound any more (round 2's 1257 were), but 80 VGPRs = six waves per SIMD for eight rows per SIMD - the launch runs as 1.33
// rounds of latency-bound waves, each of which loads its whole row, reduces it and only then starts to work.  With the row's sum of
// squares known beforehand (GatedResArgs::ystat: the O / W2 GEMM's epilogue leaves it behind) nothing in the first half needs more than
// the element at hand: every lane walks its NH 8-byte steps (d = 256 NH: 4 bf16 per lane and step, 64 lanes x NH steps cover the row
// exactly - the 16-byte layout above idles half the wave in its fifth chunk at d = 2304), keeps only the packed new residual for the
// second norm, and the kernel fits 64 registers.  Same statements in the same order as the kernel above; the one difference is the
// summation order of sum(y^2) (per GEMM tile and wave half, then over the slots).
typedef __attribute__((ext_vector_type(2))) unsigned nt_u32x2;
// (launcher: next_w and next_scale present, scale_pre 1, no next_shift - Next-DiT's block; everything else takes the kernel above)
// RPW rows per wave (1 or 2).  With one row per wave and eight waves per SIMD the whole 8192-row launch is resident at once and runs in
// lock step - every wave loads, then every wave computes, then every wave stores - so reads and writes never overlap (32.6 us = 4.6 TB/s,
// profiles/r06).  RPW 2: half as many waves, each with its SECOND row's loads in flight while it computes and stores the first.
template <int NH, int RPW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(RPW == 1 ? 8 : 4, RPW == 1 ? 8 : 5))) void gated_residual_norm_ys_kernel(GatedResArgs p) {
    const int lane = threadIdx.x & 63;
    const int row0 = __builtin_amdgcn_readfirstlane((blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW);
    if (row0 >= p.rows) return;
    const u16* postw = p.post_w + lane * 4;
    const u16* nw = p.next_w + lane * 4;
    // the whole row of y and x in flight at once (2 NH eight-byte loads per lane; x' overwrites x's registers), the per-sample vectors
    // one step ahead of their use (they come from the L1 / L2)
    nt_u32x2 yv[RPW][NH], xv[RPW][NH];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = row0 + r < p.rows ? row0 + r : p.rows - 1;
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            yv[r][i] = *(const nt_u32x2*)(p.y + (size_t)row * p.d + lane * 4 + 256 * i);
            xv[r][i] = *(const nt_u32x2*)(p.x + (size_t)row * p.d + lane * 4 + 256 * i);
        }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = row0 + r;
        if (row >= p.rows) break;  // wave-uniform
        const int b = row / p.rows_per_batch;
        u16* xrow = p.x + (size_t)row * p.d + lane * 4;
        const u16* gate = p.gate + (size_t)b * p.ld_mod + lane * 4;
        nt_u32x2 wv = *(const nt_u32x2*)postw, gv = *(const nt_u32x2*)gate;
        const float* ys = p.ystat + (size_t)row * p.ystat_slots;
        float ss = 0.f;
        for (int s = 0; s < p.ystat_slots; s += 2) {  // wave-uniform address: scalar loads (two wave halves per column tile: an even count)
            const f32x2 v = *(const f32x2*)(ys + s);
            ss += v[0] + v[1];
        }
        const float rinv = rsqrtf(ss / (float)p.d + p.eps);
        const f32x2 rv = {rinv, rinv};
        // x' = bfr(x + bfr(g * bfr(bfr(y * rinv) * w)))
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            nt_u32x2 wn = wv, gn = gv;
            if (i + 1 < NH) { wn = *(const nt_u32x2*)(postw + 256 * (i + 1)); gn = *(const nt_u32x2*)(gate + 256 * (i + 1)); }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                f32x2 yn = bfr2(bfr2(unpk_bf(yv[r][i][k]) * rv) * unpk_bf(wv[k]));
                yn = bfr2(unpk_bf(gv[k]) * yn);
                xv[r][i][k] = pk_bf(unpk_bf(xv[r][i][k]) + yn);
            }
            __builtin_nontemporal_store(xv[r][i], (nt_u32x2*)(xrow + 256 * i));  // (see the kernel above: the residual stream is not read again soon)
            wv = wn; gv = gn;
        }
        const u16* nscale = p.next_scale + (size_t)b * p.ld_mod + lane * 4;
        nt_u32x2 w2 = *(const nt_u32x2*)nw, s2 = *(const nt_u32x2*)nscale;
        float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NH; ++i)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                // (through a scalar: __builtin_bit_cast applied to an ext-vector ELEMENT drops the element index with hipcc 7.2 - every k read
                //  element 0, caught by tests/test_gpu_ops.py::test_proj_gated_residual_norm_ystat)
                const unsigned u = xv[r][i][k];
                const bf16x2_t v = __builtin_bit_cast(bf16x2_t, u);
                s4[(2 * i + k) & 3] = __builtin_amdgcn_fdot2_f32_bf16(v, v, s4[(2 * i + k) & 3], false);
            }
        const float r2 = rsqrtf(wave_sum((s4[0] + s4[1]) + (s4[2] + s4[3])) / (float)p.d + p.eps);
        const f32x2 r2v = {r2, r2};
        // (h_pair: column c = 4 lane + 256 i of row `row` sits at (row >> 1) * 2 d + (row & 1) * 32 + c + (c >> 5) * 32)
        u16* hrow = p.h_pair ? p.h + (size_t)(row >> 1) * (2 * p.d) + (row & 1) * 32 + lane * 4 + (lane >> 3) * 32 : p.h + (size_t)row * p.d + lane * 4;
        const int hstep = p.h_pair ? 512 : 256;
        // h = bfr(bfr(bfr(x' * r2) * w) * (1 + scale)) with (1 + scale) prepared in bf16: apply_rms_mod_store, 8 bytes at a time
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            nt_u32x2 wn = w2, sn = s2, o;
            if (i + 1 < NH) { wn = *(const nt_u32x2*)(nw + 256 * (i + 1)); sn = *(const nt_u32x2*)(nscale + 256 * (i + 1)); }
#pragma unroll
            for (int k = 0; k < 2; ++k) o[k] = pk_bf(bfr2(bfr2(unpk_bf(xv[r][i][k]) * r2v) * unpk_bf(w2[k])) * unpk_bf(s2[k]));
            *(nt_u32x2*)(hrow + hstep * i) = o;
            w2 = wn; s2 = sn;
        }
    }
}

}  // namespace

#define LT_DISPATCH_CHUNKS(kernel, grid, args)                                                          \
    switch ((((args).d >> 3) + 63) / 64) {                                                                \
        case 1: hipLaunchKernelGGL(kernel<1>, grid, dim3(256), 0, stream, args); break;                   \
        case 2: hipLaunchKernelGGL(kernel<2>, grid, dim3(256), 0, stream, args); break;                   \
        case 3: hipLaunchKernelGGL(kernel<3>, grid, dim3(256), 0, stream, args); break;                   \
        case 4: hipLaunchKernelGGL(kernel<4>, grid, dim3(256), 0, stream, args); break;                   \
        case 5: hipLaunchKernelGGL(kernel<5>, grid, dim3(256), 0, stream, args); break;                   \
        case 6: hipLaunchKernelGGL(kernel<6>, grid, dim3(256), 0, stream, args); break;                   \
        default: hipLaunchKernelGGL(kernel<8>, grid, dim3(256), 0, stream, args); break;                  \
    }

// option norm_specialize (options.h): 1 (default): mode-specialised instantiations (bit-identical; 34.4 -> 31.6 us at cfg 2, profiles/r02); 0: generic kernel

int launch_rmsnorm_mod(const NormModArgs& a_in, hipStream_t stream) {
    NormModArgs a = a_in;
    a.apex = lt_opt(OPT_RMSNORM_APEX);
    LT_REQUIRE(a.d % 8 == 0 && a.d <= 64 * 8 * MAXCH_LIMIT, "rmsnorm_mod: d=%d must be a multiple of 8 and <= 4096", a.d);
    LT_REQUIRE(a.rows_per_batch > 0 && a.rows > 0, "rmsnorm_mod: empty input");
    LT_REQUIRE(!a.out_pair || (a.rows % 2 == 0 && a.d % 32 == 0), "rmsnorm_mod: the pair layout needs an even row count and d %% 32 == 0");
    LT_DISPATCH_CHUNKS(rmsnorm_mod_kernel, dim3((a.rows + 3) / 4), a);
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_gated_residual_norm(const GatedResArgs& a_in, hipStream_t stream) {
    GatedResArgs a = a_in;
    a.apex = lt_opt(OPT_RMSNORM_APEX);
    LT_REQUIRE(a.d % 8 == 0 && a.d <= 64 * 8 * MAXCH_LIMIT, "gated_residual_norm: d=%d must be a multiple of 8 and <= 4096", a.d);
    LT_REQUIRE(a.rows_per_batch > 0 && a.rows > 0, "gated_residual_norm: empty input");
    LT_REQUIRE(a.gate_mode == 2 || a.gate != nullptr, "gated_residual_norm: gate pointer missing");
    LT_REQUIRE(a.y != nullptr || a.moe_pos != nullptr, "gated_residual_norm: branch output missing");
    LT_REQUIRE(a.post_mode == 0 || a.post_w != nullptr, "gated_residual_norm: post-norm weight missing");
    LT_REQUIRE(a.next_mode == 0 || a.h != nullptr, "gated_residual_norm: h output missing");
    LT_REQUIRE(!a.h_pair || (a.next_mode == 1 && !a.moe_pos && a.rows % 2 == 0 && a.d % 32 == 0), "gated_residual_norm: h in the pair layout: dense next_mode 1, even row count, d %% 32 == 0");
    LT_REQUIRE(a.route_w == nullptr || (a.moe_pos && a.next_mode == 1 && a.route_sel && a.route_wts && a.route_E >= 2 && a.route_E <= LT_MOE_MAX_E),
               "gated_residual_norm: routing on the way out needs the MoE kernel, next_mode 1, 2 <= E <= %d and the sel / wts outputs", LT_MOE_MAX_E);
    int nblk = (a.rows + 3) / 4;
    if (a.pf.blocks > 0) {  // riders behind the row blocks, from a multiple of 8 on (block index mod 8 = XCD)
        a.pf.first = (nblk + 7) / 8 * 8;
        nblk = a.pf.first + a.pf.blocks;
    } else {
        a.pf.first = 0x7fffffff;
    }
    const dim3 grid(nblk);
    if (a.moe_pos) {  // y = top-2 combine of the experts' outputs (MoE families: d = 1536 ... 4096)
        LT_REQUIRE(a.moe_ys && a.moe_wts, "gated_residual_norm: incomplete MoE combine arguments");
        // (the MoE instantiations compile the vanilla rounding order in; with the option on, the model's first pre-norm - rmsnorm_mod -
        //  would take the apex order and every later norm the vanilla one, silently)
        LT_REQUIRE(!a.apex, "gated_residual_norm: option rmsnorm_apex is not implemented for the MoE families' combine-on-load row kernel");
        // the 600M MoE's combination (d = 1536, weighted post-norm, prepared gate, next pre-norm or the final LayerNorm) with its mode switches
        // fixed at compile time, like the dense instantiations below (round 5; same statements in the same order: bit-identical)
        if (lt_opt(OPT_NORM_SPECIALIZE) && !a.apex && a.gate_mode == 0 && a.post_mode == 1 && (a.next_mode == 1 || a.next_mode == 2) && ((a.d >> 3) + 63) / 64 == 3) {
            if (a.next_mode == 1) hipLaunchKernelGGL((gated_residual_norm_kernel<3, 1, 0, 1, true>), grid, dim3(256), 0, stream, a);
            else hipLaunchKernelGGL((gated_residual_norm_kernel<3, 1, 0, 2, true>), grid, dim3(256), 0, stream, a);
            LT_CHECK_HIP(hipGetLastError());
            return 0;
        }
        switch (((a.d >> 3) + 63) / 64) {
            case 1: hipLaunchKernelGGL((gated_residual_norm_kernel<1, -1, -1, -1, true>), grid, dim3(256), 0, stream, a); break;
            case 2: hipLaunchKernelGGL((gated_residual_norm_kernel<2, -1, -1, -1, true>), grid, dim3(256), 0, stream, a); break;
            case 3: hipLaunchKernelGGL((gated_residual_norm_kernel<3, -1, -1, -1, true>), grid, dim3(256), 0, stream, a); break;
            case 4: hipLaunchKernelGGL((gated_residual_norm_kernel<4, -1, -1, -1, true>), grid, dim3(256), 0, stream, a); break;
            case 5: hipLaunchKernelGGL((gated_residual_norm_kernel<5, -1, -1, -1, true>), grid, dim3(256), 0, stream, a); break;
            case 6: hipLaunchKernelGGL((gated_residual_norm_kernel<6, -1, -1, -1, true>), grid, dim3(256), 0, stream, a); break;
            default: hipLaunchKernelGGL((gated_residual_norm_kernel<8, -1, -1, -1, true>), grid, dim3(256), 0, stream, a); break;
        }
        LT_CHECK_HIP(hipGetLastError());
        return 0;
    }
    // the streaming kernel (GatedResArgs::ystat): dense, weighted post-norm, prepared gate, next pre-norm; d = 256 NH; no riders (large-M launches)
    if (a.ystat && lt_opt(OPT_NORM_SPECIALIZE) && !a.apex && a.gate_mode == 0 && a.post_mode == 1 && a.next_mode == 1 && a.pf.blocks == 0 &&
        a.next_w && a.next_scale && a.scale_pre && !a.next_shift && a.y && a.d % 256 == 0 && a.ystat_slots > 0 && a.ystat_slots % 2 == 0) {
        const int nh = a.d / 256;
        if (nh == 6 || nh == 9) {  // (d = 3072 / 4096 do not fit the 64 registers of eight waves per SIMD: the kernel above)
            if (lt_opt(OPT_GRN_YSTAT) == 2) {  // two rows per wave: the second row's loads fly under the first row's arithmetic and stores
                const dim3 grid2((a.rows + 7) / 8);
                if (nh == 6) hipLaunchKernelGGL((gated_residual_norm_ys_kernel<6, 2>), grid2, dim3(256), 0, stream, a);
                else hipLaunchKernelGGL((gated_residual_norm_ys_kernel<9, 2>), grid2, dim3(256), 0, stream, a);
            } else if (nh == 6) hipLaunchKernelGGL((gated_residual_norm_ys_kernel<6, 1>), grid, dim3(256), 0, stream, a);
            else hipLaunchKernelGGL((gated_residual_norm_ys_kernel<9, 1>), grid, dim3(256), 0, stream, a);
            LT_CHECK_HIP(hipGetLastError());
            return 0;
        }
    }
    if (lt_opt(OPT_NORM_SPECIALIZE) && !a.apex && a.gate_mode == 0 && (a.post_mode == 0 || a.post_mode == 1) && (a.next_mode == 1 || a.next_mode == 2)) {
        const int nch64 = ((a.d >> 3) + 63) / 64;
#define LT_GRN_SPEC(MC, PMV, NMV) \
    hipLaunchKernelGGL((gated_residual_norm_kernel<MC, PMV, 0, NMV>), grid, dim3(256), 0, stream, a)
#define LT_GRN_MODES(MC)                                                                      \
    do {                                                                                      \
        if (a.post_mode == 1 && a.next_mode == 1) LT_GRN_SPEC(MC, 1, 1);                      \
        else if (a.post_mode == 1) LT_GRN_SPEC(MC, 1, 2);                                     \
        else if (a.next_mode == 1) LT_GRN_SPEC(MC, 0, 1);                                     \
        else LT_GRN_SPEC(MC, 0, 2);                                                           \
    } while (0)
        if (nch64 == 3) { LT_GRN_MODES(3); LT_CHECK_HIP(hipGetLastError()); return 0; }  // d = 1536 (cfg 1, 5)
        if (nch64 == 5) { LT_GRN_MODES(5); LT_CHECK_HIP(hipGetLastError()); return 0; }  // d = 2304 (cfg 2, 4)
        if (nch64 == 6) { LT_GRN_MODES(6); LT_CHECK_HIP(hipGetLastError()); return 0; }  // d = 3072 (cfg 3)
#undef LT_GRN_MODES
#undef LT_GRN_SPEC
    }
    LT_DISPATCH_CHUNKS(gated_residual_norm_kernel, grid, a);
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}

