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

namespace {

constexpr int MAXCH = 8;  // 16-byte chunks per lane -> d <= 64 * 8 * 8 = 4096

struct RowRegs {
    float v[MAXCH][8];
};

__device__ __forceinline__ void load_row(const u16* row, int nch, int lane, RowRegs& r) {
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            const bf8_t t = *(const bf8_t*)(row + c * 8);
            unpack8(t, r.v[i]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) r.v[i][e] = 0.f;
        }
    }
}

__device__ __forceinline__ float row_sumsq(const RowRegs& r) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) s += r.v[i][e] * r.v[i][e];
    return wave_sum(s);
}

// h = bfr(bfr(bfr(x * r) * w) * bfr(1 + scale)) (+ shift)   -- each step optional as in the reference
__device__ __forceinline__ void apply_rms_mod_store(const RowRegs& r, float rinv, const u16* w, const u16* scale,
                                                    const u16* shift, u16* out, int nch, int lane) {
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            float o[8], wf[8], sf[8], hf[8];
            if (w) unpack8(*(const bf8_t*)(w + c * 8), wf);
            if (scale) unpack8(*(const bf8_t*)(scale + c * 8), sf);
            if (shift) unpack8(*(const bf8_t*)(shift + c * 8), hf);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float n = bfr(r.v[i][e] * rinv);
                if (w) n = bfr(n * wf[e]);
                if (scale) n = bfr(n * bfr(1.0f + sf[e]));
                if (shift) n = bfr(n + hf[e]);
                o[e] = n;
            }
            *(bf8_t*)(out + c * 8) = pack8(o);
        }
    }
}

__global__ __launch_bounds__(256) void rmsnorm_mod_kernel(NormModArgs p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const int b = row / p.rows_per_batch;
    const int nch = p.d >> 3;
    RowRegs r;
    load_row(p.x + (size_t)row * p.d, nch, lane, r);
    const float rinv = rsqrtf(row_sumsq(r) / (float)p.d + p.eps);
    apply_rms_mod_store(r, rinv, p.w, p.scale ? p.scale + (size_t)b * p.ld_mod : nullptr,
                        p.shift ? p.shift + (size_t)b * p.ld_mod : nullptr, p.out + (size_t)row * p.d, nch, lane);
}

__global__ __launch_bounds__(256) void gated_residual_norm_kernel(GatedResArgs p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const int b = row / p.rows_per_batch;
    const int nch = p.d >> 3;
    RowRegs r;
    load_row(p.y + (size_t)row * p.d, nch, lane, r);
    float rinv = 1.f;
    if (p.post_mode == 1) rinv = rsqrtf(row_sumsq(r) / (float)p.d + p.eps);
    const u16* gate = p.gate ? p.gate + (size_t)b * p.ld_mod : nullptr;
    u16* xrow = p.x + (size_t)row * p.d;
    // x' = bfr(x + bfr(g * yn));  r <- x'
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            float xf[8], wf[8], gf[8], o[8];
            unpack8(*(const bf8_t*)(xrow + c * 8), xf);
            if (p.post_mode == 1) unpack8(*(const bf8_t*)(p.post_w + c * 8), wf);
            if (p.gate_mode != 2) unpack8(*(const bf8_t*)(gate + c * 8), gf);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float yn = r.v[i][e];
                if (p.post_mode == 1) yn = bfr(bfr(yn * rinv) * wf[e]);
                if (p.gate_mode == 1) yn = bfr(bfr(tanhf(gf[e])) * yn);
                else if (p.gate_mode == 0) yn = bfr(gf[e] * yn);
                o[e] = bfr(xf[e] + yn);
                r.v[i][e] = o[e];
            }
            *(bf8_t*)(xrow + c * 8) = pack8(o);
        }
    }
    if (p.next_mode == 0) return;
    const u16* nscale = p.next_scale ? p.next_scale + (size_t)b * p.ld_mod : nullptr;
    const u16* nshift = p.next_shift ? p.next_shift + (size_t)b * p.ld_mod : nullptr;
    u16* hrow = p.h + (size_t)row * p.d;
    if (p.next_mode == 1) {
        const float r2 = rsqrtf(row_sumsq(r) / (float)p.d + p.eps);
        apply_rms_mod_store(r, r2, p.next_w, nscale, nshift, hrow, nch, lane);
    } else {
        // affine-free LayerNorm in fp32, modulate in fp32, one rounding (the cast autocast applies at the
        // final Linear) -- model.py:634-638, :660-661
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXCH; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) s += r.v[i][e];
        const float mean = wave_sum(s) / (float)p.d;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nch) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float dlt = r.v[i][e] - mean;
                    q += dlt * dlt;
                }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) / (float)p.d + p.eps_next);
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nch) {
                float sf[8], hf[8], o[8];
                if (nscale) unpack8(*(const bf8_t*)(nscale + c * 8), sf);
                if (nshift) unpack8(*(const bf8_t*)(nshift + c * 8), hf);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float n = (r.v[i][e] - mean) * rstd;
                    if (nscale) n = n * bfr(1.0f + sf[e]);
                    if (nshift) n = n + hf[e];
                    o[e] = n;
                }
                *(bf8_t*)(hrow + c * 8) = pack8(o);
            }
        }
    }
}

}  // namespace

int launch_rmsnorm_mod(const NormModArgs& a, hipStream_t stream) {
    LT_REQUIRE(a.d % 8 == 0 && a.d <= 64 * 8 * MAXCH, "rmsnorm_mod: d=%d must be a multiple of 8 and <= 4096", a.d);
    LT_REQUIRE(a.rows_per_batch > 0 && a.rows > 0, "rmsnorm_mod: empty input");
    hipLaunchKernelGGL(rmsnorm_mod_kernel, dim3((a.rows + 3) / 4), dim3(256), 0, stream, a);
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_gated_residual_norm(const GatedResArgs& a, hipStream_t stream) {
    LT_REQUIRE(a.d % 8 == 0 && a.d <= 64 * 8 * MAXCH, "gated_residual_norm: d=%d must be a multiple of 8 and <= 4096", a.d);
    LT_REQUIRE(a.rows_per_batch > 0 && a.rows > 0, "gated_residual_norm: empty input");
    LT_REQUIRE(a.gate_mode == 2 || a.gate != nullptr, "gated_residual_norm: gate pointer missing");
    LT_REQUIRE(a.post_mode == 0 || a.post_w != nullptr, "gated_residual_norm: post-norm weight missing");
    LT_REQUIRE(a.next_mode == 0 || a.h != nullptr, "gated_residual_norm: h output missing");
    hipLaunchKernelGGL(gated_residual_norm_kernel, dim3((a.rows + 3) / 4), dim3(256), 0, stream, a);
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}
