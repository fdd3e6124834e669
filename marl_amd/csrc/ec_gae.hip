// ec_gae.hip — generalized advantage estimation as a reverse scan, gfx950.
//
// Replaces utils/advantage_utils.py:4-9 (get_gae): for every (env b, agent n) chain
//     delta_t = r_t + gamma * V'_t * m_t - V_t
//     adv_t   = delta_t + gamma*lambda * adv_{t+1} * m_t,      adv_T = 0
// evaluated in fp32 with one rounding per torch op and in the reference's operand order (no FMA
// contraction), so the result is bit-identical to the reference's Python loop over T.
//
// One lane owns one chain and walks t = T-1 .. 0.  Arrays are [B][T][N]; a wave's 64 chains are
// contiguous in (b, n), so a load at a given t touches ceil(64/N) runs of N floats; the U-step
// unrolled prefetch keeps 4*U independent loads in flight per lane, and consecutive t of one run share
// cache lines, so HBM traffic stays at the algorithmic 20 B per element.
#include "ec_common.h"

namespace {

constexpr int GAE_U = 8;

__global__ void __launch_bounds__(64) k_gae(const float* __restrict__ r, const float* __restrict__ v,
                                           const float* __restrict__ nv, const float* __restrict__ m,
                                           float* __restrict__ adv, int64_t B, int64_t T, int64_t N, float gamma,
                                           float gl) {
    const int64_t chain = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (chain >= B * N) return;
    const int64_t b = chain / N, n = chain - b * N;
    const int64_t base = b * T * N + n;
    float next = 0.0f;
    int64_t t = T - 1;
    for (; t >= GAE_U - 1; t -= GAE_U) {
        float rr[GAE_U], vv[GAE_U], nn[GAE_U], mm[GAE_U];
#pragma unroll
        for (int u = 0; u < GAE_U; u++) {
            const int64_t i = base + (t - u) * N;
            rr[u] = r[i];
            vv[u] = v[i];
            nn[u] = nv[i];
            mm[u] = m[i];
        }
#pragma unroll
        for (int u = 0; u < GAE_U; u++) {
            const float delta = __fsub_rn(__fadd_rn(rr[u], __fmul_rn(__fmul_rn(gamma, nn[u]), mm[u])), vv[u]);
            next = __fadd_rn(delta, __fmul_rn(__fmul_rn(gl, next), mm[u]));
            adv[base + (t - u) * N] = next;
        }
    }
    for (; t >= 0; t--) {
        const int64_t i = base + t * N;
        const float mi = m[i];
        const float delta = __fsub_rn(__fadd_rn(r[i], __fmul_rn(__fmul_rn(gamma, nv[i]), mi)), v[i]);
        next = __fadd_rn(delta, __fmul_rn(__fmul_rn(gl, next), mi));
        adv[i] = next;
    }
}

}  // namespace

extern "C" int ec_gae(const float* r, const float* v, const float* nv, const float* m, float* adv, int64_t B,
                      int64_t T, int64_t N, float gamma, float gamma_lambda, void* stream) {
    if (B < 0 || T < 0 || N < 0) return ec::fail(EC_EINVAL, "ec_gae: negative shape");
    const int64_t chains = B * N;
    if (chains == 0 || T == 0) return EC_OK;  // empty batch: nothing to do (pointers may be null)
    if (!r || !v || !nv || !m || !adv) return ec::fail(EC_EINVAL, "ec_gae: null pointer");
    hipLaunchKernelGGL(k_gae, dim3((unsigned)((chains + 63) / 64)), dim3(64), 0, (hipStream_t)stream, r, v, nv, m, adv,
                       B, T, N, gamma, gamma_lambda);
    EC_HIP(hipGetLastError());
    return EC_OK;
}
