// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of the Next-DiT denoising engine.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t u16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// bf16 <-> f32.  f2bf is round-to-nearest-even (v_cvt_pk_bf16_f32 on gfx950).
__device__ __forceinline__ float bf2f(u16 v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ u16 f2bf(float f) { return __builtin_bit_cast(u16, (__bf16)f); }
// round an fp32 value to the nearest bf16 and keep it as fp32 (reference rounding points)
__device__ __forceinline__ float bfr(float f) { return bf2f(f2bf(f)); }
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
    return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);
}
// The same value from ONE v_cvt_pk_bf16_f32 (the scalar form above compiles to two single-operand converts, a shift and an SDWA
// or: four instructions per pair).  Used by the GEMM epilogues, where the main loops were checked to stay opcode-identical; the
// attention kernels keep pack2bf until the change can be measured (it re-schedules three of attn v3's seven MFMA blocks).
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__device__ __forceinline__ unsigned pack2bf_pk(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// full-wave (64 lanes) reductions, result uniform in every lane.  Inside a row of 16 lanes: four DPP steps (quad_perm xor 1,
// xor 2, row_half_mirror, row_mirror - VALU only); across the four rows: v_readlane of one lane per row + three adds.
// (__shfl_xor compiles to ds_bpermute_b32 on this toolchain: six dependent LDS round trips per reduction, which put ~1.2 k
// cycles of pure latency into every row of the row kernels.)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float readlane_f(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);  // row_half_mirror
    v += dpp_mov<0x140>(v);  // row_mirror: every lane holds its row's sum
    return (readlane_f(v, 0) + readlane_f(v, 16)) + (readlane_f(v, 32) + readlane_f(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0x140>(v));
    return fmaxf(fmaxf(readlane_f(v, 0), readlane_f(v, 16)), fmaxf(readlane_f(v, 32), readlane_f(v, 48)));
}

// x * sigmoid(x) with the hardware reciprocal (1 ulp): an IEEE fp32 division expands to ~10 instructions and the result is
// rounded to bf16 right away (reference: F.silu in bf16)
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// 16-byte vector of 8 bf16 as raw words
struct __attribute__((aligned(16))) bf8_t { unsigned w[4]; };

__device__ __forceinline__ void unpack8(const bf8_t& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = bf_lo(v.w[i]); f[2 * i + 1] = bf_hi(v.w[i]); }
}
__device__ __forceinline__ bf8_t pack8(const float* f) {
    bf8_t v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v.w[i] = pack2bf(f[2 * i], f[2 * i + 1]);
    return v;
}

// ---- pairwise (packed) helpers: the row kernels are VALU-heavy (a bf16 rounding after every reference op), so they
// work on two elements per instruction: v_pk_mul/add/fma_f32 and one v_cvt_pk_bf16_f32 per rounded pair ------------------
__device__ __forceinline__ unsigned pk_bf(f32x2 v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)); }
__device__ __forceinline__ f32x2 unpk_bf(unsigned u) { return f32x2{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)}; }
__device__ __forceinline__ f32x2 bfr2(f32x2 v) { return unpk_bf(pk_bf(v)); }  // round both to bf16, keep as fp32
// two SwiGLU outputs: per element bfr(silu_f(bfr(x))) * bfr(y) - the reference's rounding points (model.py:497-502 under bf16: w1 x,
// w3 x, silu, the product is rounded by the caller) - with the multiplies / the add as packed fp32 instructions and one
// v_cvt_pk_bf16_f32 per rounded pair (the scalar form left 2 of its 5 fp32 ops per element unpacked and a quarter of the converts
// single: 13.5 -> 12 VALU instructions per output of the SwiGLU epilogue).  Same operations, same order: bit-identical.
__device__ __forceinline__ f32x2 swiglu2(f32x2 x, f32x2 y) {
    const f32x2 xr = bfr2(x), yr = bfr2(y);
    const float nl2e = __uint_as_float(0xbfb8aa3bu);  // -log2(e), the constant __expf(-x) multiplies by
    const f32x2 t = xr * f32x2{nl2e, nl2e};
    const f32x2 e = f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + f32x2{1.0f, 1.0f};
    const f32x2 r = {__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
    return bfr2(xr * r) * yr;
}

// Output rows of the one-wave-per-SIMD attention kernels: a lane holds, per (dt, q4), 8 bytes of ITS query row (d = 32 dt + 8 q4 +
// 4 hi ..+3).  Stored from there, every instruction touched 64 different lines (rows of the [B, N, H * hd] output lie H * hd * 2 bytes
// apart): 18 instructions x 64 partial-line writes per wave at head_dim 72, ~2 us of the texture addresser per workgroup.  Instead the
// wave parks its 64 rows x HD in a private LDS strip (`tb`, 64 * HD * 2 bytes, touched by this wave only: LDS operations of one wave
// stay in order) and stores them as HD / 8 instructions of 64 consecutive 16-byte chunks - ~7 row segments = ~16 lines each.
// obase = the output address of the wave's first row (this head's columns); rows_valid = rows of the 64 that exist.
template <int HD, int DT>
__device__ __forceinline__ void store_rows_via_lds(char* tb, const u32x2 (&res)[2][DT][4], int lane, u16* obase, size_t row_stride, int rows_valid,
                                                   int pair = 0, size_t grow0 = 0, int col0 = 0) {
    const int hi = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int d0 = 32 * dt + 8 * q4 + 4 * hi;
                if (d0 < HD) *(u32x2*)(tb + (blk * 32 + l31) * (HD * 2) + d0 * 2) = res[blk][dt][q4];
            }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    constexpr int CPR = HD / 8;  // 16-byte chunks per row
#pragma unroll
    for (int i = 0; i < CPR; ++i) {
        const int c = lane + 64 * i;
        const int r = c / CPR, ch = c - r * CPR;
        if (r < rows_valid) {
            // pair != 0: obase is the [rows][row_stride] matrix in the row-pair-interleaved layout (GemmArgs::pair_ab), the strip's first row is
            // its row grow0 and the strip's first column its column col0
            const size_t gr = grow0 + r;
            const int c = col0 + ch * 8;
            u16* dst = pair ? obase + (gr >> 1) * (2 * row_stride) + (gr & 1) * 32 + c + (c >> 5) * 32 : obase + (size_t)r * row_stride + ch * 8;
            *(u32x4*)dst = *(const u32x4*)(tb + r * (HD * 2) + ch * 16);
        }
    }
}

// host-side error plumbing shared by the launchers
void lt_set_error(const char* fmt, ...);
// compute units of the CURRENT device (cached per device id; defined in gemm_bf16.hip)
int num_cus();
#define LT_CHECK_HIP(expr)                                                              \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            lt_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return 1;                                                                   \
        }                                                                               \
    } while (0)
#define LT_REQUIRE(cond, ...)                \
    do {                                     \
        if (!(cond)) {                       \
            lt_set_error(__VA_ARGS__);       \
            return 2;                        \
        }                                    \
    } while (0)
