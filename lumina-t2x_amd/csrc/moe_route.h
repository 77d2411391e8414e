// The router arithmetic shared by moe_route_kernel (moe.hip) and the row kernel that can route on its way out (norm.hip, round 5):
// top-2 of E logits, fp32 softmax over the two, bf16 weights (Next-DiT-MoE/models/models2.py:464-470 / :493-499).
#pragma once
#include "common.h"

constexpr int LT_MOE_MAX_E = 8;

// top-2 of E logits (lowest index wins ties), fp32 softmax over the two, bf16 weights; (sel, wts) in ascending expert id
__device__ __forceinline__ void top2_route(const float (&logit)[LT_MOE_MAX_E], const int* forced2, int& s0, int& s1, u16& w0, u16& w1) {
    int i1 = 0;
#pragma unroll
    for (int e = 1; e < LT_MOE_MAX_E; ++e) if (logit[e] > logit[i1]) i1 = e;
    int i2 = i1 == 0 ? 1 : 0;
#pragma unroll
    for (int e = 0; e < LT_MOE_MAX_E; ++e) if (e != i1 && e != i2 && logit[e] > logit[i2]) i2 = e;
    if (forced2) {  // the discrete choice comes from outside (a reference run's); the weights stay this run's own arithmetic
        i1 = forced2[0];
        i2 = forced2[1];
    }
    float l1 = 0.f, l2 = 0.f;
#pragma unroll
    for (int e = 0; e < LT_MOE_MAX_E; ++e) { l1 = e == i1 ? logit[e] : l1; l2 = e == i2 ? logit[e] : l2; }
    // softmax over (v1, v2) in fp32, then the cast back to the activation dtype (:466-470)
    const float ex = __expf(l2 - l1);
    const float wa = 1.0f / (1.0f + ex), wb = ex / (1.0f + ex);
    const bool swap = i2 < i1;  // accumulate in ascending expert id
    s0 = swap ? i2 : i1;
    s1 = swap ? i1 : i2;
    w0 = f2bf(swap ? wb : wa);
    w1 = f2bf(swap ? wa : wb);
}

// one 8-channel chunk of a token row against the same chunk of every router row: acc[e] += sum_i x[i] * w[e][i], i ascending - the
// statement order moe_route_kernel has always had, so that both routers form the same fp32 sums
__device__ __forceinline__ void route_accumulate(const bf8_t& x, const u16* gate_w, int E, int d, int chunk, float (&acc)[LT_MOE_MAX_E]) {
    float xf[8];
    unpack8(x, xf);
#pragma unroll
    for (int e = 0; e < LT_MOE_MAX_E; ++e) {
        if (e < E) {
            float wf[8];
            unpack8(*(const bf8_t*)(gate_w + (size_t)e * d + chunk * 8), wf);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[e] += xf[i] * wf[i];
        }
    }
}
