// Small / HBM-bound kernels around the transformer stack: patchify, timestep features, caption pooling,
// GEMV-style linears for the conditioning path (M <= 8), unpatchify + classifier-free guidance, the
// fixed-grid ODE update and weight upload casts.  Each cites the reference lines it restates.
#include "common.h"
#include "kernels.h"

namespace {

// ---- y[m,n] = sum_k act(a[m,k]) w[n,k] + b[n]  (adaLN_modulation model.py:560-569, t_embedder :44-60,
//      cap_embedder :702-711, final adaLN :646-655).  Weight-bandwidth bound: one wave per output column.
constexpr int SM_MAXM = 8;
// Round 3: a wave owns SM_NC consecutive output columns instead of one.  The one-column form re-read and re-activated (SiLU + bf16
// rounding: ~10 VALU per element) the M input rows for every column - 100 k columns at the adaLN GEMV of cfg 1 - and had two 16-byte
// weight loads in flight per lane: VALU-bound at 2.5 TB/s of weights (61 us x 3 per NFE at cfg 2, 80 us at cfg 1).  Now the
// activated inputs of a chunk are formed once per SM_NC columns and SM_NC independent weight loads are in flight per chunk.  Every
// output is still the same lane-strided partial sums followed by the same wave reduction: bit-identical results.
constexpr int SM_NC = 8;
// EXT: LinearSmallMExtra (kernels.h) - inputs formed on load, launch_prep_mod's transform on the way out
template <int MM, bool EXT = false>  // rows held in registers: 2 (a CFG pair), 4 or 8
__global__ __launch_bounds__(256) void linear_small_m_kernel(const u16* __restrict__ a, const u16* __restrict__ w,
                                                             const u16* __restrict__ bias, u16* __restrict__ y, int M,
                                                             int N, int K, int act_in, LinearSmallMExtra x = LinearSmallMExtra()) {
    const int lane = threadIdx.x & 63;
    const int n0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * SM_NC;
    if (n0 >= N) return;
    const int nch = K >> 3;
    float acc[SM_NC][MM];
#pragma unroll
    for (int j = 0; j < SM_NC; ++j)
#pragma unroll
        for (int m = 0; m < MM; ++m) acc[j][m] = 0.f;
    for (int c = lane; c < nch; c += 64) {
        bf8_t wv[SM_NC];
#pragma unroll
        for (int j = 0; j < SM_NC; ++j) {  // columns past N re-read the last one (results discarded)
            const int n = n0 + j < N ? n0 + j : N - 1;
            wv[j] = *(const bf8_t*)(w + (size_t)n * K + c * 8);
        }
        float af[MM][8];
#pragma unroll
        for (int m = 0; m < MM; ++m) {
            if (m < M) {
                if (EXT && x.t) {  // timestep_features_kernel's statements (below), feature k = 8 c + e of t[m]
                    const int half = K / 2;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int k = c * 8 + e, kk = k < half ? k : k - half;
                        const float freq = expf(-9.210340371976184f * (float)kk / (float)half);
                        const float arg = x.t[m] * freq;
                        af[m][e] = bfr(k < half ? cosf(arg) : sinf(arg));
                    }
                } else {
                    unpack8(*(const bf8_t*)(a + (size_t)m * K + c * 8), af[m]);
                }
                if (EXT && x.a2) {  // add_bf16_kernel: bf16(a + a2)
                    float bf_[8];
                    unpack8(*(const bf8_t*)(x.a2 + (size_t)m * K + c * 8), bf_);
#pragma unroll
                    for (int e = 0; e < 8; ++e) af[m][e] = bfr(af[m][e] + bf_[e]);
                }
                if (act_in == 1) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) af[m][e] = bfr(silu_f(af[m][e]));
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) af[m][e] = 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < SM_NC; ++j) {
            float wf[8];
            unpack8(wv[j], wf);
#pragma unroll
            for (int m = 0; m < MM; ++m)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[j][m] += af[m][e] * wf[e];
        }
    }
#pragma unroll
    for (int j = 0; j < SM_NC; ++j) {
        if (n0 + j < N) {
#pragma unroll
            for (int m = 0; m < MM; ++m) {
                if (m < M) {
                    float s = wave_sum(acc[j][m]);
                    if (lane == 0) {
                        if (bias) s += bf2f(bias[n0 + j]);
                        u16 o = f2bf(s);
                        if (EXT && x.pm_d > 0) {  // prep_mod_kernel's transform of this column (on the ROUNDED value, as the separate pass reads it)
                            const int n = n0 + j, layers = x.pm_L * x.pm_chunks * x.pm_d;
                            int mode = 0;  // 1 tanh, 2 one-plus
                            if (n < layers) {
                                const int ch = (n / x.pm_d) % x.pm_chunks;
                                mode = ((x.pm_tanh >> ch) & 1u) ? 1 : (((x.pm_scale >> ch) & 1u) ? 2 : 0);
                            } else if (x.pm_final >= 0 && n >= layers + x.pm_final * x.pm_d && n < layers + (x.pm_final + 1) * x.pm_d) {
                                mode = 2;
                            }
                            if (mode == 1) o = f2bf(tanhf(bf2f(o)));
                            else if (mode == 2) o = f2bf(1.0f + bf2f(o));
                        }
                        y[(size_t)m * N + n0 + j] = o;
                    }
                }
            }
        }
    }
}

__global__ void cast_to_bf16_kernel(const void* __restrict__ src, int dtype, u16* __restrict__ dst, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        if (dtype == 0) dst[i] = f2bf(((const float*)src)[i]);
        else if (dtype == 1) dst[i] = ((const u16*)src)[i];
        else dst[i] = f2bf((float)((const _Float16*)src)[i]);
    }
}

// patchify (model.py:776-777): rows (b, i, j), columns (c, ph, pw); zero padded to kpad
// wp_stride = tokens per latent row in the output (Wp, or Wp + 1 when every row carries an eol token: Flag-DiT,
// lumina_t2i/models/model.py:779-786 - the eol rows are left untouched here and filled by eol_fill_kernel)
__global__ void patchify_kernel(const void* __restrict__ x, int x_dtype, u16* __restrict__ out, int B, int C, int H,
                                int W, int patch, int kpad, int dup_first_half, int wp_stride) {
    const int Hp = H / patch, Wp = W / patch;
    const long long total = (long long)B * Hp * Wp * kpad;
    const int kreal = C * patch * patch;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i % kpad);
        const long long row = i / kpad;
        u16 v = 0;
        if (k < kreal) {
            const int j = (int)(row % Wp);
            const int ii = (int)((row / Wp) % Hp);
            int b = (int)(row / ((long long)Wp * Hp));
            if (dup_first_half) b = b % (B / 2);  // combined = cat([half, half])  (model.py:901-902)
            const int c = k / (patch * patch), ph = (k / patch) % patch, pw = k % patch;
            const size_t idx = (((size_t)b * C + c) * H + (ii * patch + ph)) * W + (j * patch + pw);
            v = x_dtype == 0 ? f2bf(((const float*)x)[idx]) : ((const u16*)x)[idx];
        }
        const long long orow = (row / Wp) * wp_stride + (row % Wp);
        out[orow * kpad + k] = v;
    }
}

// rows[(b * Hp + r) * (Wp + 1) + Wp][:] = eol_token  (lumina_t2i/models/model.py:779-786)
__global__ void eol_fill_kernel(u16* __restrict__ x, const u16* __restrict__ eol, int rows_total, int Wp, int d) {
    const int chunks = d >> 3;
    const long long total = (long long)rows_total * chunks;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % chunks);
        const long long r = i / chunks;
        *(bf8_t*)(x + ((r * (Wp + 1) + Wp) * (long long)d) + c * 8) = *(const bf8_t*)(eol + c * 8);
    }
}

// class-conditional embedding lookup: out[b][:] = table[labels[b]][:]  (ParallelLabelEmbedder.forward in eval,
// Next-DiT-ImageNet/models/models.py:216-221; the null class is row num_classes)
__global__ void label_gather_kernel(const u16* __restrict__ table, const int32_t* __restrict__ labels, u16* __restrict__ out,
                                    int B, int rows, int d) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * d) return;
    const int b = i / d, c = i % d;
    int l = labels[b];
    l = l < 0 ? 0 : (l >= rows ? rows - 1 : l);
    out[i] = table[(size_t)l * d + c];
}

// timestep_embedding (model.py:63-82): [cos(t f_i) | sin(t f_i)], f_i = exp(-ln(1e4) i / half), cast bf16 (:86)
__global__ void timestep_features_kernel(const float* __restrict__ t, u16* __restrict__ out, int B, int dim) {
    const int half = dim / 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * half) return;
    const int b = i / half, k = i % half;
    const float freq = expf(-9.210340371976184f * (float)k / (float)half);
    const float arg = t[b] * freq;
    out[(size_t)b * dim + k] = f2bf(cosf(arg));
    out[(size_t)b * dim + half + k] = f2bf(sinf(arg));
}

// masked mean over tokens in fp32, cast back to the feature dtype (model.py:847-849), then the affine
// LayerNorm of cap_embedder[0] (model.py:703; fp32 under autocast), stored bf16 (cast at the Linear).
__global__ __launch_bounds__(256) void cap_pool_ln_kernel(const void* __restrict__ cap, int cap_dtype,
                                                          const int32_t* __restrict__ mask, const u16* __restrict__ ln_w,
                                                          const u16* __restrict__ ln_b, u16* __restrict__ out, int T, int C) {
    extern __shared__ float sh[];  // pooled[C] + 8 reduction slots
    float* pooled = sh;
    float* red = sh + C;
    const int b = blockIdx.x;
    float cnt = 0.f;
    for (int t = 0; t < T; ++t) cnt += (float)mask[b * T + t];
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s = 0.f;
        for (int t = 0; t < T; ++t) {
            const size_t idx = ((size_t)b * T + t) * C + c;
            const float v = cap_dtype == 0 ? ((const float*)cap)[idx] : bf2f(((const u16*)cap)[idx]);
            s += v * (float)mask[b * T + t];
        }
        s = s / cnt;
        pooled[c] = cap_dtype == 0 ? s : bfr(s);
    }
    __syncthreads();
    float s = 0.f;
    for (int c = threadIdx.x; c < C; c += blockDim.x) s += pooled[c];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)C;
    __syncthreads();
    float q = 0.f;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float d = pooled[c] - mean;
        q += d * d;
    }
    q = wave_sum(q);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = q;
    __syncthreads();
    const float rstd = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)C + 1e-5f);
    for (int c = threadIdx.x; c < C; c += blockDim.x)
        out[(size_t)b * C + c] = f2bf((pooled[c] - mean) * rstd * bf2f(ln_w[c]) + bf2f(ln_b[c]));
}

// adaLN vectors, in place, once per NFE: gate chunks -> bf16(tanh(gate)) (`gate_msa.unsqueeze(1).tanh()` is a bf16 tensor
// op per (sample, channel) in the reference, model.py:597, :606) and scale chunks -> bf16(1 + scale) (modulate, :28-29).
// Doing either per token inside the row kernels made those HBM-bound kernels VALU-bound.
__global__ void prep_mod_kernel(u16* __restrict__ mod, int B, int ld_mod, int L, int chunks, int d, unsigned tanh_mask,
                                unsigned scale_mask, int final_scale_chunk) {
    const long long per_b = (long long)L * chunks * d + (final_scale_chunk >= 0 ? d : 0);
    const long long total = (long long)B * per_b;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / per_b);
        const long long r = i % per_b;
        u16* p;
        int mode;  // 1 tanh, 2 one-plus
        if (r < (long long)L * chunks * d) {
            const int ch = (int)((r / d) % chunks);
            mode = ((tanh_mask >> ch) & 1u) ? 1 : (((scale_mask >> ch) & 1u) ? 2 : 0);
            p = mod + (size_t)b * ld_mod + r;
        } else {
            mode = 2;
            p = mod + (size_t)b * ld_mod + (size_t)L * chunks * d + (size_t)final_scale_chunk * d + (r - (long long)L * chunks * d);
        }
        if (mode == 1) *p = f2bf(tanhf(bf2f(*p)));
        else if (mode == 2) *p = f2bf(1.0f + bf2f(*p));
    }
}

__global__ void add_bf16_kernel(const u16* a, const u16* b, u16* c, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) c[i] = f2bf(bf2f(a[i]) + bf2f(b[i]));
}

__global__ void mask_to_bias_kernel(const int32_t* mask, float* bias, int B, int T, int Tpad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * Tpad) return;
    const int b = i / Tpad, t = i % Tpad;
    bias[i] = (t < T && mask[b * T + t] != 0) ? 0.f : -INFINITY;
}

// unpatchify (model.py:749-755: row layout (pH, pW, C_out)), keep the first C channels (:859-861), then
// CFG on the first cfg_channels channels only (model.py:908-913) with the bf16 rounding of each step.
__global__ void unpatchify_cfg_kernel(const u16* __restrict__ rows, int ld, void* __restrict__ out, int out_dtype, int B,
                                      int C, int out_ch, int H, int W, int patch, int use_cfg, float cfg_scale,
                                      int cfg_channels, int wp_stride) {
    const long long total = (long long)B * C * H * W;
    const int Hp = H / patch;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int w = (int)(i % W);
        const int hh = (int)((i / W) % H);
        const int c = (int)((i / ((long long)W * H)) % C);
        const int b = (int)(i / ((long long)W * H * C));
        const int e = ((hh % patch) * patch + (w % patch)) * out_ch + c;
        const long long tok = (long long)(hh / patch) * wp_stride + (w / patch);  // eol column (if any) is skipped
        auto rd = [&](int bb) { return bf2f(rows[((long long)bb * Hp * wp_stride + tok) * ld + e]); };
        float v;
        if (use_cfg && c < cfg_channels) {
            const int half = B / 2;
            const int bc = b % half;
            const float cond = rd(bc), unc = rd(bc + half);
            v = bfr(unc + bfr(cfg_scale * bfr(cond - unc)));
        } else {
            v = rd(b);
        }
        if (out_dtype == 0) ((float*)out)[i] = v;
        else ((u16*)out)[i] = f2bf(v);
    }
}

// torchdiffeq fixed-grid solver arithmetic on the ODE state (euler / midpoint / rk4 "3/8 rule",
// torchdiffeq rk_common.rk4_alt_step_func).  With a bf16 state every tensor op of the Python expression
// rounds to bf16 (a 0-dim fp32 dt times a bf16 tensor stays bf16 AND sees dt cast to bf16 first - the caller passes
// bf16(dt) for a bf16 state); R() marks those points.
//   mode 0: y0 + R(dt k1)                               euler step, midpoint half step / full step
//   mode 1: y0 + R(R(dt k1) / 3)                        rk4 stage-2 input
//   mode 2: y0 + R(dt R(k2 - R(k1 / 3)))                rk4 stage-3 input
//   mode 3: y0 + R(dt R(R(k1 - k2) + k3))               rk4 stage-4 input
//   mode 4: y0 + R(R(R(R(k1 + R(3 R(k2 + k3))) + k4) dt) 0.125)
template <bool BF>
__global__ void ode_combine_kernel(int mode, const void* y0, const void* k1, const void* k2, const void* k3,
                                   const void* k4, void* out, float dt, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    auto ld = [&](const void* p) { return BF ? bf2f(((const u16*)p)[i]) : ((const float*)p)[i]; };
    auto R = [](float x) { return BF ? bfr(x) : x; };
    const float y = ld(y0);
    float r;
    if (mode == 0) r = y + R(dt * ld(k1));
    else if (mode == 1) r = y + R(R(dt * ld(k1)) * (1.0f / 3.0f));
    else if (mode == 2) r = y + R(dt * R(ld(k2) - R(ld(k1) * (1.0f / 3.0f))));
    else if (mode == 3) r = y + R(dt * R(R(ld(k1) - ld(k2)) + ld(k3)));
    else r = y + R(R(R(R(ld(k1) + R(3.0f * R(ld(k2) + ld(k3)))) + ld(k4)) * dt) * 0.125f);
    if (BF) ((u16*)out)[i] = f2bf(r);
    else ((float*)out)[i] = r;
}

// precompute_freqs_cis of every sub-project reduced to 1-D factor tables: out[branch][pos][fi] = cis(angle), with
//   f_fi = theta_b^(-step fi / hd), step = 4 (2-D RoPE: hd/4 frequencies per axis; lumina_next_t2i/models/model.py:915-963,
//   Next-DiT-ImageNet/models/models.py:977-1012) or 2 (1-D: hd/2 frequencies; lumina_t2i/models/model.py:924-960);
//   angle = pos * (f / lin_b)   (Next-DiT T2I: the frequency is divided, model.py:952-953)  or
//           (pos / lin_b) * f   (ImageNet / Flag-DiT: the position is divided, models.py:1003-1005)
// out_t (optional): the same factors as [branch][fi][pos] - positions contiguous, for readers whose LANES differ in position
// (the attention prologue's column lookups, AttnArgs::rope_cs_t)
__global__ void rope_table_kernel(float* out, float* out_t, int len, int nf, int step, int hd, float theta0, float lin0, float theta1,
                                  float lin1, int lin_on_pos) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * len * nf) return;
    const int fi = i % nf, pos = (i / nf) % len, branch = i / (nf * len);
    const float lin = branch == 0 ? lin0 : lin1;
    const float th = branch == 0 ? theta0 : theta1;
    const float freq = 1.0f / powf(th, (float)(step * fi) / (float)hd);
    const float ang = lin_on_pos ? ((float)pos / lin) * freq : (float)pos * (freq / lin);
    const float c = cosf(ang), sn = sinf(ang);
    out[2 * (size_t)i] = c;
    out[2 * (size_t)i + 1] = sn;
    if (out_t) {
        const size_t j = ((size_t)branch * nf + fi) * len + pos;
        out_t[2 * j] = c;
        out_t[2 * j + 1] = sn;
    }
}

__global__ void fill_rows_bf16_kernel(u16* dst, const u16* row, long long rows, int d) {
    const long long total = rows * d;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
        dst[i] = row[i % d];
}

// Compositional Next-DiT text branch (lumina_next_compositional_generation/models/model.py:422-446).  txt[r] = SDPA of the
// image's queries against caption r (r < Y-1: regional captions of the COND row, r = Y-1: the caption of the UNCOND row),
// already rounded to bf16 as SDPA returns it under autocast.  Reference order of operations:
//   output_y = nan_to_num(SDPA(..., y_mask & region_mask))      tokens outside region r have every key masked -> NaN -> 0
//   output_y = output_y * tanh(gate)                            bf16 * bf16 -> bf16
//   cond = sum(output_y[:-1], dim 0) ; uncond = output_y[-1]    fp32 accumulation inside torch.sum, one rounding
//   output = output + output_y                                  bf16
// region_mask (model.py:872-887): the latent grid is cut into h_split x w_split cells of (Hp / h_split) x (Wp / w_split)
// tokens; cell (i, j) switches on caption (i + 1) * (j + 1) - 1 (the reference's formula, kept as is: cells can share a
// caption and some captions stay empty); tokens beyond the last full cell belong to no region; the last caption covers all.
__global__ void region_text_combine_kernel(u16* __restrict__ out, const u16* __restrict__ txt, const u16* __restrict__ gate,
                                           int Y, int N, int H, int hd, int Hp, int Wp, int h_split, int w_split) {
    const int d = H * hd, chunks = d >> 3;
    const long long total = (long long)2 * N * chunks;
    const int hps = Hp / h_split, wps = Wp / w_split;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % chunks);
        const long long rown = i / chunks;
        const int b = (int)(rown / N), n = (int)(rown % N);
        const int head = (c * 8) / hd;
        const float g = bfr(tanhf(bf2f(gate[head])));
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
        if (b == 0) {
            const int gr = n / Wp, gc = n - gr * Wp;
            const int ci = hps > 0 ? gr / hps : h_split, cj = wps > 0 ? gc / wps : w_split;
            const int region = (ci < h_split && cj < w_split) ? (ci + 1) * (cj + 1) - 1 : -1;
            if (region >= 0 && region < Y - 1) {  // exactly one regional caption can be on for a token
                const bf8_t t = *(const bf8_t*)(txt + ((size_t)region * N + n) * d + c * 8);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f32x2 v = unpk_bf(t.w[k]);
                    acc[2 * k] = bfr(v[0] * g);
                    acc[2 * k + 1] = bfr(v[1] * g);
                }
            }
        } else {
            const bf8_t t = *(const bf8_t*)(txt + ((size_t)(Y - 1) * N + n) * d + c * 8);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x2 v = unpk_bf(t.w[k]);
                acc[2 * k] = bfr(v[0] * g);
                acc[2 * k + 1] = bfr(v[1] * g);
            }
        }
        u16* o = out + ((size_t)b * N + n) * d + c * 8;
        bf8_t cur = *(const bf8_t*)o, res;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x2 v = unpk_bf(cur.w[k]);
            res.w[k] = pk_bf(f32x2{v[0] + acc[2 * k], v[1] + acc[2 * k + 1]});
        }
        *(bf8_t*)o = res;
    }
}

inline int nblk(long long n, int bs) { return (int)((n + bs - 1) / bs); }

}  // namespace

int launch_linear_small_m(const u16* a, const u16* w, const u16* b, u16* y, int M, int N, int K, int act_in,
                          hipStream_t stream) {
    LT_REQUIRE(M >= 1 && M <= SM_MAXM, "linear_small_m: M=%d out of range 1..%d", M, SM_MAXM);
    LT_REQUIRE(K % 8 == 0, "linear_small_m: K=%d must be a multiple of 8", K);
    const dim3 grid(((N + SM_NC - 1) / SM_NC + 3) / 4);
    if (M <= 2) hipLaunchKernelGGL((linear_small_m_kernel<2, false>), grid, dim3(256), 0, stream, a, w, b, y, M, N, K, act_in, LinearSmallMExtra());
    else if (M <= 4) hipLaunchKernelGGL((linear_small_m_kernel<4, false>), grid, dim3(256), 0, stream, a, w, b, y, M, N, K, act_in, LinearSmallMExtra());
    else hipLaunchKernelGGL((linear_small_m_kernel<8, false>), grid, dim3(256), 0, stream, a, w, b, y, M, N, K, act_in, LinearSmallMExtra());
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_linear_small_m_ext(const u16* a, const u16* w, const u16* b, u16* y, int M, int N, int K, int act_in, const LinearSmallMExtra& x,
                              hipStream_t stream) {
    LT_REQUIRE(M >= 1 && M <= SM_MAXM, "linear_small_m: M=%d out of range 1..%d", M, SM_MAXM);
    LT_REQUIRE(K % 8 == 0 && (!x.t || K % 16 == 0), "linear_small_m: K=%d must be a multiple of 8 (16 with timestep features)", K);
    LT_REQUIRE(a || x.t, "linear_small_m: no input");
    LT_REQUIRE(x.pm_d == 0 || (x.pm_L > 0 && x.pm_chunks > 0 && x.pm_chunks <= 32), "linear_small_m: bad prep_mod geometry");
    const dim3 grid(((N + SM_NC - 1) / SM_NC + 3) / 4);
    if (M <= 2) hipLaunchKernelGGL((linear_small_m_kernel<2, true>), grid, dim3(256), 0, stream, a, w, b, y, M, N, K, act_in, x);
    else if (M <= 4) hipLaunchKernelGGL((linear_small_m_kernel<4, true>), grid, dim3(256), 0, stream, a, w, b, y, M, N, K, act_in, x);
    else hipLaunchKernelGGL((linear_small_m_kernel<8, true>), grid, dim3(256), 0, stream, a, w, b, y, M, N, K, act_in, x);
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_cast_to_bf16(const void* src, int dtype, u16* dst, long long n, hipStream_t stream) {
    LT_REQUIRE(dtype >= 0 && dtype <= 2, "cast_to_bf16: bad dtype %d", dtype);
    if (n == 0) return 0;
    int g = nblk(n, 256);
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(cast_to_bf16_kernel, dim3(g), dim3(256), 0, stream, src, dtype, dst, n);
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_patchify(const void* x, int x_dtype, u16* out, int B, int C, int H, int W, int patch, int kpad,
                    int dup_first_half, int wp_stride, hipStream_t stream) {
    LT_REQUIRE(H % patch == 0 && W % patch == 0, "patchify: %dx%d not divisible by patch %d", H, W, patch);
    LT_REQUIRE(C * patch * patch <= kpad, "patchify: kpad too small");
    LT_REQUIRE(!dup_first_half || B % 2 == 0, "patchify: CFG needs an even batch");
    const long long total = (long long)B * (H / patch) * (W / patch) * kpad;
    int g = nblk(total, 256);
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(patchify_kernel, dim3(g), dim3(256), 0, stream, x, x_dtype, out, B, C, H, W, patch, kpad,
                       dup_first_half, wp_stride > 0 ? wp_stride : W / patch);
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_timestep_features(const float* t, int t_index, u16* out, int B, int dim, hipStream_t stream) {
    LT_REQUIRE(dim % 2 == 0, "timestep_features: odd dim unsupported");
    hipLaunchKernelGGL(timestep_features_kernel, dim3(nblk((long long)B * dim / 2, 128)), dim3(128), 0, stream,
                       t + (size_t)t_index * B, out, B, dim);
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_cap_pool_ln(const void* cap, int cap_dtype, const int32_t* mask, const u16* ln_w, const u16* ln_b, u16* out,
                       int B, int T, int C, hipStream_t stream) {
    hipLaunchKernelGGL(cap_pool_ln_kernel, dim3(B), dim3(256), (C + 8) * sizeof(float), stream, cap, cap_dtype, mask,
                       ln_w, ln_b, out, T, C);
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_prep_mod(u16* mod, int B, int ld_mod, int L, int chunks, int d, unsigned tanh_mask, unsigned scale_mask,
                    int final_scale_chunk, hipStream_t stream) {
    const long long total = (long long)B * ((long long)L * chunks * d + d);
    int g = nblk(total, 256);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(prep_mod_kernel, dim3(g), dim3(256), 0, stream, mod, B, ld_mod, L, chunks, d, tanh_mask, scale_mask,
                       final_scale_chunk);
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_add_bf16(const u16* a, const u16* b, u16* c, long long n, hipStream_t stream) {
    hipLaunchKernelGGL(add_bf16_kernel, dim3(nblk(n, 256)), dim3(256), 0, stream, a, b, c, n);
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_mask_to_bias(const int32_t* mask, float* bias, int B, int T, int Tpad, hipStream_t stream) {
    hipLaunchKernelGGL(mask_to_bias_kernel, dim3(nblk((long long)B * Tpad, 256)), dim3(256), 0, stream, mask, bias, B,
                       T, Tpad);
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_unpatchify_cfg(const u16* rows, int ld, void* out, int out_dtype, int B, int C, int out_ch, int H, int W,
                          int patch, int use_cfg, float cfg_scale, int cfg_channels, int wp_stride, hipStream_t stream) {
    LT_REQUIRE(!use_cfg || B % 2 == 0, "unpatchify_cfg: CFG needs an even batch");
    const long long total = (long long)B * C * H * W;
    int g = nblk(total, 256);
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(unpatchify_cfg_kernel, dim3(g), dim3(256), 0, stream, rows, ld, out, out_dtype, B, C, out_ch, H,
                       W, patch, use_cfg, cfg_scale, cfg_channels, wp_stride > 0 ? wp_stride : W / patch);
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_region_text_combine(u16* out, const u16* txt, const u16* gate, int Y, int N, int H, int hd, int Hp, int Wp,
                               int h_split, int w_split, hipStream_t stream) {
    LT_REQUIRE(Y >= 2 && h_split >= 1 && w_split >= 1 && hd % 8 == 0, "region_text_combine: bad arguments");
    int g = nblk((long long)2 * N * (H * hd / 8), 256);
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(region_text_combine_kernel, dim3(g), dim3(256), 0, stream, out, txt, gate, Y, N, H, hd, Hp, Wp, h_split, w_split);
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_ode_combine(int mode, const void* y0, const void* k1, const void* k2, const void* k3, const void* k4,
                       void* out, int dtype, float dt, long long n, hipStream_t stream) {
    LT_REQUIRE(mode >= 0 && mode <= 4, "ode_combine: bad mode %d", mode);
    LT_REQUIRE(dtype == 0 || dtype == 1, "ode_combine: state dtype must be f32 or bf16");
    if (dtype == 1)
        hipLaunchKernelGGL(ode_combine_kernel<true>, dim3(nblk(n, 256)), dim3(256), 0, stream, mode, y0, k1, k2, k3, k4, out, dt, n);
    else
        hipLaunchKernelGGL(ode_combine_kernel<false>, dim3(nblk(n, 256)), dim3(256), 0, stream, mode, y0, k1, k2, k3, k4, out, dt, n);
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_rope_table_2d(float* out, int len, int hd, float theta, float scale_factor, hipStream_t stream, float* out_t) {
    // Next-DiT T2I: branch 0 = linear interpolation (t < watershed), branch 1 = NTK (model.py:944-949)
    return launch_rope_table(out, len, hd, 4, theta, scale_factor, theta * scale_factor, 1.0f, 0, stream, out_t);
}

int launch_rope_table(float* out, int len, int hd, int step, float theta0, float lin0, float theta1, float lin1,
                      int lin_on_pos, hipStream_t stream, float* out_t) {
    LT_REQUIRE(step == 2 || step == 4, "rope_table: step must be 2 (1-D) or 4 (2-D)");
    LT_REQUIRE(hd % step == 0 && len > 0, "rope_table: hd %% %d != 0", step);
    LT_REQUIRE(lin0 > 0.f && lin1 > 0.f && theta0 > 0.f && theta1 > 0.f, "rope_table: factors must be positive");
    const int nf = hd / step;
    hipLaunchKernelGGL(rope_table_kernel, dim3(nblk(2LL * len * nf, 256)), dim3(256), 0, stream, out, out_t, len, nf, step, hd,
                       theta0, lin0, theta1, lin1, lin_on_pos);
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_eol_fill(u16* x, const u16* eol, int rows_total, int Wp, int d, hipStream_t stream) {
    LT_REQUIRE(d % 8 == 0, "eol_fill: d %% 8 != 0");
    int g = nblk((long long)rows_total * (d / 8), 256);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(eol_fill_kernel, dim3(g), dim3(256), 0, stream, x, eol, rows_total, Wp, d);
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_label_gather(const u16* table, const int32_t* labels, u16* out, int B, int rows, int d, hipStream_t stream) {
    hipLaunchKernelGGL(label_gather_kernel, dim3(nblk((long long)B * d, 256)), dim3(256), 0, stream, table, labels, out, B, rows, d);
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---- row-pair-interleaved layout (GemmArgs::pair_ab, round 6) -------------------------------------------------------------------------
// In place, one workgroup per pair of rows of a dense [rows][cols] matrix: element (r, k) <-> (r >> 1) * 2 cols + (k >> 5) * 64 + (r & 1) * 32 +
// (k & 31).  A 16-byte chunk c of row rr sits at chunk rr * (cols / 8) + c of the pair's row-major image and at chunk (c >> 2) * 8 + rr * 4 +
// (c & 3) of its interleaved one.  The whole pair is read into registers before the first store (cols <= 16384: 16 chunks per thread).
namespace {
__global__ __launch_bounds__(256) void pair_layout_kernel(u16* __restrict__ m, int cols, int to_pair) {
    u16* base = m + (size_t)blockIdx.x * 2 * cols;
    const int cpr = cols >> 3, n = 2 * cpr;
    bf8_t v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = threadIdx.x + 256 * i;
        if (c < n) v[i] = *(const bf8_t*)(base + (size_t)c * 8);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = threadIdx.x + 256 * i;  // chunk index in the SOURCE image
        if (c < n) {
            int dst;
            if (to_pair) { const int rr = c >= cpr ? 1 : 0, cc = c - rr * cpr; dst = (cc >> 2) * 8 + rr * 4 + (cc & 3); }
            else { const int rr = (c >> 2) & 1, cc = (c >> 3) * 4 + (c & 3); dst = rr * cpr + cc; }
            *(bf8_t*)(base + (size_t)dst * 8) = v[i];
        }
    }
}
}  // namespace

int launch_pair_layout(u16* m, long long rows, int cols, int to_pair, hipStream_t stream) {
    LT_REQUIRE(m && rows > 0 && rows % 2 == 0 && cols > 0 && cols % 32 == 0 && cols <= 16384 && rows / 2 < 0x7fffffffLL,
               "pair_layout: an even number of rows of 32 k <= 16384 columns (got %lld x %d)", rows, cols);
    hipLaunchKernelGGL(pair_layout_kernel, dim3((unsigned)(rows / 2)), dim3(256), 0, stream, m, cols, to_pair);
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_fill_rows_bf16(u16* dst, const u16* row, long long rows, int d, hipStream_t stream) {
    int g = nblk(rows * d, 256);
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(fill_rows_bf16_kernel, dim3(g), dim3(256), 0, stream, dst, row, rows, d);
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}

namespace {
__global__ void upload_rows_kernel(const void* __restrict__ src, int dtype, u16* __restrict__ dst, int rows, int cols,
                                   int dst_ld, int r0, int row_map) {
    const long long total = (long long)rows * cols;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / cols), c = (int)(i % cols);
        int dr;
        if (row_map == 0) dr = r0 + r;
        else dr = (r >> 5) * 64 + (r & 31) + (row_map == 2 ? 32 : 0);
        u16 v;
        if (dtype == 0) v = f2bf(((const float*)src)[i]);
        else if (dtype == 1) v = ((const u16*)src)[i];
        else v = f2bf((float)((const _Float16*)src)[i]);
        dst[(size_t)dr * dst_ld + c] = v;
    }
}
}  // namespace

int launch_upload_rows(const void* src, int dtype, u16* dst, int rows, int cols, int dst_ld, int r0, int row_map,
                       hipStream_t stream) {
    LT_REQUIRE(dtype >= 0 && dtype <= 2, "upload_rows: bad dtype %d", dtype);
    LT_REQUIRE(row_map == 0 || rows % 32 == 0, "upload_rows: interleaved layout needs rows %% 32 == 0");
    const long long total = (long long)rows * cols;
    if (total == 0) return 0;
    long long g = (total + 255) / 256;
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(upload_rows_kernel, dim3((int)g), dim3(256), 0, stream, src, dtype, dst, rows, cols, dst_ld, r0, row_map);
    LT_CHECK_HIP(hipGetLastError());
    return 0;
}
