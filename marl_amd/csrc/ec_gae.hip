// ec_gae.hip — generalized advantage estimation as a reverse scan, gfx950.
//
// Replaces utils/advantage_utils.py:4-9 (get_gae): for every (env b, agent n) chain
//     delta_t = r_t + gamma * V'_t * m_t - V_t
//     adv_t   = delta_t + gamma*lambda * adv_{t+1} * m_t,      adv_T = 0
// evaluated in fp32 with one rounding per torch op and in the reference's operand order (no FMA
// contraction), so the result is bit-identical to the reference's Python loop over T.
//
// The recurrence is sequential in t by its rounding (a cross-lane affine scan would re-associate it and lose the
// bit-identity), so the wave's work is split by what CAN be wide:
//   k_gae_tile (N <= 64, the path's shapes): one wave per env walks that env's [T][N] slab, which is contiguous in all
//     five arrays, in chunks of TC steps from the end.  (1) all 64 lanes load r, V, V', m of the chunk as whole 16-byte
//     vectors, fully coalesced, take delta (elementwise) and park delta and m in the wave's LDS tile; (2) lanes n < N
//     walk the chunk's TC steps out of LDS — the only serial part, 3 dependent flops a step, its LDS reads batched
//     ahead of the chain — writing adv over delta; (3) all lanes store the tile as 16-byte vectors.  No block barrier:
//     a wave's LDS operations complete in order, so a wavefront fence orders the phases.
//   k_gae (any other shape): one lane per chain, 4-byte strided accesses, batches of U steps software-pipelined.
// HBM traffic is the algorithmic 20 B per element in both.
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
    // The recurrence itself is a dependent chain per lane; what the wave can hide is the memory round trip of its
    // inputs.  The batches are software-pipelined: the 4 U loads of batch k + 1 are in flight while batch k's U steps
    // are taken (one batch at a time, a 100-step chain paid 13 round trips back to back: 25 us for 82 MB).
    float rr[GAE_U], vv[GAE_U], nn[GAE_U], mm[GAE_U];
    const auto fetch = [&](int64_t tt, float (&a)[GAE_U], float (&bq)[GAE_U], float (&c)[GAE_U], float (&e)[GAE_U]) {
#pragma unroll
        for (int u = 0; u < GAE_U; u++) {
            const int64_t i = base + (tt - u) * N;
            a[u] = r[i];
            bq[u] = v[i];
            c[u] = nv[i];
            e[u] = m[i];
        }
    };
    if (t >= GAE_U - 1) fetch(t, rr, vv, nn, mm);
    for (; t >= GAE_U - 1; t -= GAE_U) {
        float r2[GAE_U], v2[GAE_U], n2[GAE_U], m2[GAE_U];
        const bool more = t - GAE_U >= GAE_U - 1;
        if (more) fetch(t - GAE_U, r2, v2, n2, m2);
#pragma unroll
        for (int u = 0; u < GAE_U; u++) {
            const float delta = __fsub_rn(__fadd_rn(rr[u], __fmul_rn(__fmul_rn(gamma, nn[u]), mm[u])), vv[u]);
            next = __fadd_rn(delta, __fmul_rn(__fmul_rn(gl, next), mm[u]));
            adv[base + (t - u) * N] = next;
        }
        if (more) {
#pragma unroll
            for (int u = 0; u < GAE_U; u++) {
                rr[u] = r2[u];
                vv[u] = v2[u];
                nn[u] = n2[u];
                mm[u] = m2[u];
            }
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

constexpr int GAE_TILE = 1024;   // floats per LDS array of a wave (delta, m): 8 KB per wave, 32 KB per block
constexpr int GAE_WAVES = 4;
constexpr int GAE_K = 4;         // 16-byte vectors per lane per pass of phase 1 (4 arrays x 4 in flight)

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void gae_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ float gae_delta(float r, float v, float nv, float m, float gamma) {
    return __fsub_rn(__fadd_rn(r, __fmul_rn(__fmul_rn(gamma, nv), m)), v);
}

// VEC: T*N is a multiple of 4 and the five bases are 16-byte aligned, so is every chunk (TC*N a multiple of 4 whenever
// T is cut).  Otherwise the same phases with 4-byte coalesced accesses.
template <bool VEC>
__global__ void __launch_bounds__(64 * GAE_WAVES) k_gae_tile(const float* __restrict__ r, const float* __restrict__ v,
                                                            const float* __restrict__ nv, const float* __restrict__ m,
                                                            float* __restrict__ adv, int64_t B, int T, int N, int TC,
                                                            float gamma, float gl) {
    __shared__ __attribute__((aligned(16))) float tile[GAE_WAVES][2][GAE_TILE];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* const D = tile[wave][0];
    float* const M = tile[wave][1];
    for (int64_t b = (int64_t)blockIdx.x * GAE_WAVES + wave; b < B; b += (int64_t)gridDim.x * GAE_WAVES) {
        float next = 0.0f;
        for (int t1 = T; t1 > 0; t1 -= TC) {
            const int t0 = t1 > TC ? t1 - TC : 0;
            const int cnt = (t1 - t0) * N;
            const int64_t off = (b * T + t0) * (int64_t)N;
            // (1) delta and m of the chunk into LDS
            if (VEC) {
                const int nv4 = cnt >> 2;
                for (int i0 = 0; i0 < nv4; i0 += 64 * GAE_K) {
                    f32x4 a[GAE_K], bq[GAE_K], c[GAE_K], e[GAE_K];
#pragma unroll
                    for (int k = 0; k < GAE_K; k++) {
                        const int i = i0 + k * 64 + lane;
                        if (i < nv4) {
                            a[k] = *reinterpret_cast<const f32x4*>(r + off + 4 * i);
                            bq[k] = *reinterpret_cast<const f32x4*>(v + off + 4 * i);
                            c[k] = *reinterpret_cast<const f32x4*>(nv + off + 4 * i);
                            e[k] = *reinterpret_cast<const f32x4*>(m + off + 4 * i);
                        }
                    }
#pragma unroll
                    for (int k = 0; k < GAE_K; k++) {
                        const int i = i0 + k * 64 + lane;
                        if (i < nv4) {
                            f32x4 dl;
#pragma unroll
                            for (int q = 0; q < 4; q++) dl[q] = gae_delta(a[k][q], bq[k][q], c[k][q], e[k][q], gamma);
                            *reinterpret_cast<f32x4*>(D + 4 * i) = dl;
                            *reinterpret_cast<f32x4*>(M + 4 * i) = e[k];
                        }
                    }
                }
            } else {
                for (int i0 = 0; i0 < cnt; i0 += 64 * GAE_K) {
                    float a[GAE_K], bq[GAE_K], c[GAE_K], e[GAE_K];
#pragma unroll
                    for (int k = 0; k < GAE_K; k++) {
                        const int i = i0 + k * 64 + lane;
                        if (i < cnt) {
                            a[k] = r[off + i];
                            bq[k] = v[off + i];
                            c[k] = nv[off + i];
                            e[k] = m[off + i];
                        }
                    }
#pragma unroll
                    for (int k = 0; k < GAE_K; k++) {
                        const int i = i0 + k * 64 + lane;
                        if (i < cnt) {
                            D[i] = gae_delta(a[k], bq[k], c[k], e[k], gamma);
                            M[i] = e[k];
                        }
                    }
                }
            }
            gae_wave_sync();
            // (2) the chains of this env, one lane each, over the chunk's steps (last step first)
            if (lane < N) {
                int t = t1 - t0 - 1;
                for (; t >= GAE_U - 1; t -= GAE_U) {
                    float dd[GAE_U], mm[GAE_U];
#pragma unroll
                    for (int u = 0; u < GAE_U; u++) {
                        dd[u] = D[(t - u) * N + lane];
                        mm[u] = M[(t - u) * N + lane];
                    }
#pragma unroll
                    for (int u = 0; u < GAE_U; u++) {
                        next = __fadd_rn(dd[u], __fmul_rn(__fmul_rn(gl, next), mm[u]));
                        D[(t - u) * N + lane] = next;
                    }
                }
                for (; t >= 0; t--) {
                    next = __fadd_rn(D[t * N + lane], __fmul_rn(__fmul_rn(gl, next), M[t * N + lane]));
                    D[t * N + lane] = next;
                }
            }
            gae_wave_sync();
            // (3) the chunk's advantages out
            if (VEC) {
                for (int i = lane; i < (cnt >> 2); i += 64)
                    *reinterpret_cast<f32x4*>(adv + off + 4 * i) = *reinterpret_cast<const f32x4*>(D + 4 * i);
            } else {
                for (int i = lane; i < cnt; i += 64) adv[off + i] = D[i];
            }
            gae_wave_sync();  // (the next chunk's phase 1 overwrites the tile)
        }
    }
}

}  // namespace

extern "C" int ec_gae(const float* r, const float* v, const float* nv, const float* m, float* adv, int64_t B,
                      int64_t T, int64_t N, float gamma, float gamma_lambda, void* stream) {
    if (B < 0 || T < 0 || N < 0) return ec::fail(EC_EINVAL, "ec_gae: negative shape");
    const int64_t chains = B * N;
    if (chains == 0 || T == 0) return EC_OK;  // empty batch: nothing to do (pointers may be null)
    if (!r || !v || !nv || !m || !adv) return ec::fail(EC_EINVAL, "ec_gae: null pointer");
    if (N <= 64 && T <= INT32_MAX / 64) {
        int TC = (int)T;
        if ((int64_t)TC * N > GAE_TILE) TC = (GAE_TILE / (int)N) & ~3;  // (>= 16; a multiple of 4 keeps chunks aligned)
        const auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
        const bool vec = (T * N) % 4 == 0 && al(r) && al(v) && al(nv) && al(m) && al(adv);
        const int64_t blocks = (B + GAE_WAVES - 1) / GAE_WAVES;
        const dim3 grid((unsigned)(blocks < 256 * 5 ? blocks : 256 * 5));  // 5 blocks of 32 KB of LDS fit a CU
        if (vec)
            hipLaunchKernelGGL(k_gae_tile<true>, grid, dim3(64 * GAE_WAVES), 0, (hipStream_t)stream, r, v, nv, m, adv, B,
                               (int)T, (int)N, TC, gamma, gamma_lambda);
        else
            hipLaunchKernelGGL(k_gae_tile<false>, grid, dim3(64 * GAE_WAVES), 0, (hipStream_t)stream, r, v, nv, m, adv, B,
                               (int)T, (int)N, TC, gamma, gamma_lambda);
    } else {
        hipLaunchKernelGGL(k_gae, dim3((unsigned)((chains + 63) / 64)), dim3(64), 0, (hipStream_t)stream, r, v, nv, m,
                           adv, B, T, N, gamma, gamma_lambda);
    }
    EC_HIP(hipGetLastError());
    return EC_OK;
}
