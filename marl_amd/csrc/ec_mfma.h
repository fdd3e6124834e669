// ec_mfma.h — shared device helpers of the fp32 MFMA kernels (ec_mlp.hip, ec_ppo.hip): fragment-ordered weight staging
// and the register-chained 32x32x2 layer.  See the header of ec_mlp.hip for the scheme.
#pragma once
#include "ec_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int OBS = 9;      // edge_node.py:74
constexpr int H = 64;       // hidden_dim this build supports (ppo_config.py:6, dgail_config.py hidden_dim)
constexpr int NT = 256;     // threads per block = 4 waves
constexpr int NW = NT / 64; // waves per block
constexpr int G = 2;        // row groups of 32 per wave
constexpr int OBS_TILE = 64 * OBS + 4;

__device__ __forceinline__ float relu(float x) { return x > 0.0f ? x : 0.0f; }

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// input walked at k-step s by the lanes of one half when the input is a previous layer's accumulator
__device__ __forceinline__ int kperm(int s, int half) {
    const int T = s >> 4, r = s & 15;
    return 32 * T + 8 * (r >> 2) + 4 * half + (r & 3);
}

// Fragment-ordered A operand: dst[((s/4)*TO + to)*64 + l][s%4] = M[out = 32to + (l&31)][k(s, l>>5)],
// M[out][k] = w[out*so + k*si] (so/si pick the matrix or its transpose), zero outside n_out x K.
__device__ void stage_frag(float* __restrict__ dst, const float* __restrict__ w, int so, int si, int n_out, int K, int steps,
                           int TO, bool hidden_in) {
    const int total = steps * TO * 64;
    // eight gathers in flight per thread before the first LDS store: one element per iteration costs one L2 round
    // trip each (the whole staging of a 9-64-64-11 net was 28 of them, most of a 40 960-row launch)
    constexpr int U = 8;
    for (int base = threadIdx.x; base < total; base += U * blockDim.x) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int idx = base + u * (int)blockDim.x;
            const int c = idx & 3, l = (idx >> 2) & 63, rest = idx >> 8;
            const int to = rest % TO, s = 4 * (rest / TO) + c;
            const int half = l >> 5, out = 32 * to + (l & 31);
            const int k = hidden_in ? kperm(s, half) : 2 * s + half;
            v[u] = (idx < total && out < n_out && k < K) ? w[(size_t)out * so + (size_t)k * si] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int idx = base + u * (int)blockDim.x;
            if (idx < total) dst[idx] = v[u];
        }
    }
}

// accumulator tile <- per-output vector (bias) in the lane's output order
template <int TO>
__device__ __forceinline__ void acc_init(f32x16 (&acc)[TO], const float* __restrict__ vec, int half) {
#pragma unroll
    for (int to = 0; to < TO; to++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float4 v = *reinterpret_cast<const float4*>(vec + 32 * to + 8 * q + 4 * half);
            acc[to][4 * q + 0] = v.x;
            acc[to][4 * q + 1] = v.y;
            acc[to][4 * q + 2] = v.z;
            acc[to][4 * q + 3] = v.w;
        }
}
template <int TO>
__device__ __forceinline__ void acc_add(f32x16 (&acc)[TO], const float* __restrict__ vec, int half) {
#pragma unroll
    for (int to = 0; to < TO; to++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float4 v = *reinterpret_cast<const float4*>(vec + 32 * to + 8 * q + 4 * half);
            acc[to][4 * q + 0] += v.x;
            acc[to][4 * q + 1] += v.y;
            acc[to][4 * q + 2] += v.z;
            acc[to][4 * q + 3] += v.w;
        }
}
template <int TO>
__device__ __forceinline__ void acc_zero(f32x16 (&acc)[TO]) {
#pragma unroll
    for (int to = 0; to < TO; to++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[to][r] = 0.0f;
}

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// hidden layer: out[g] += frag * f(in[g]), input = both 32-output tiles of the previous layer (64 inputs, 32 steps).
// F(g, T, r, x) maps accumulator register r of input tile T to the B operand (ReLU, or a mask select).
template <int TO, typename F>
__device__ __forceinline__ void layer_hidden(f32x16 (&out)[G][TO], const float* __restrict__ frag, const f32x16 (&in)[G][2],
                                             int lane, F f) {
#pragma unroll
    for (int s4 = 0; s4 < 8; s4++) {
#pragma unroll
        for (int to = 0; to < TO; to++) {
            const float4 a = *reinterpret_cast<const float4*>(frag + ((s4 * TO + to) * 64 + lane) * 4);
            const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int g = 0; g < G; g++) {
                    const int T = s4 >> 2, r = (s4 & 3) * 4 + c;
                    out[g][to] = MFMA(av[c], f(g, T, r, in[g][T][r]), out[g][to]);
                }
        }
    }
}

// first layer of the policy / GAIL nets: K = 9 observation inputs (5 steps, the 10th input is a zero column) from the
// wave's observation tile in LDS
__device__ __forceinline__ void layer_obs(f32x16 (&out)[G][2], const float* __restrict__ frag /*8 steps staged*/,
                                          const float* __restrict__ s_obs, int lane) {
    const int jrow = lane & 31, half = lane >> 5;
#pragma unroll
    for (int s4 = 0; s4 < 2; s4++) {
#pragma unroll
        for (int to = 0; to < 2; to++) {
            const float4 a = *reinterpret_cast<const float4*>(frag + ((s4 * 2 + to) * 64 + lane) * 4);
            const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int s = 4 * s4 + c;
                if (s < 5) {
#pragma unroll
                    for (int g = 0; g < G; g++) {
                        const int k = 2 * s + half;
                        const float b = k < OBS ? s_obs[(32 * g + jrow) * OBS + k] : 0.0f;
                        out[g][to] = MFMA(av[c], b, out[g][to]);
                    }
                }
            }
        }
    }
}

// coalesced copy of the wave's 64 observation rows into its LDS tile (zero beyond R)
__device__ __forceinline__ void load_obs_tile(float* __restrict__ s_obs, const float* __restrict__ obs, int64_t rowbase, int64_t R,
                                              int lane) {
    const int64_t nvalid = (R - rowbase < 64 ? R - rowbase : 64) * OBS;
    for (int i = lane; i < 64 * OBS; i += 64) s_obs[i] = i < nvalid ? obs[rowbase * OBS + i] : 0.0f;
}


}  // namespace
