// ec_policy_inline.h — the Actor forward + mask + softmax + sampling of ec_mlp.hip's k_policy for ONE tile of 32 rows
// by ONE wave, callable from inside another kernel (the looping env kernel evaluates the policy between two steps
// without leaving the launch).  Same arithmetic as k_policy, operation for operation (ppo_network.py:21-25,
// ppo_agent.py:61-74, utils/action_selectors.py:11-20): every accumulator sees the same v_mfma_f32_32x32x2_f32 chain
// in the same order, the epilogue the same max / expf / running sums — so the actions equal k_policy's bit for bit.
// What differs is where operands live: there is no LDS to spare next to the env step's queues, so the weight
// fragments (staged once per launch in k_policy's LDS order by ec::stage_policy) are read from global memory / L2
// as 16-byte loads, the observation rows come straight from the env's output, and the logits are exchanged between
// the two half-waves with a cross-lane swap instead of an LDS tile.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace polinl {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int OBS = 9, H = 64;
// float offsets inside the staged block (ec::stage_policy writes it)
constexpr int F1 = 0;                 // layer 1 fragments: 8 steps x 2 tiles x 64 lanes
constexpr int F2 = F1 + 8 * 2 * 64;   // layer 2: 32 steps x 2 tiles
constexpr int F3 = F2 + 32 * 2 * 64;  // layer 3: 32 steps x 1 tile (n_actions <= 32)
constexpr int B1 = F3 + 32 * 1 * 64;
constexpr int B2 = B1 + H;
constexpr int B3 = B2 + H;            // [32], zero beyond n_actions
constexpr int ID = B3 + 32;           // [n_agents][64]: the one-hot agent-id columns of W1
__host__ __device__ constexpr int staged_floats(int n_agents) { return ID + n_agents * H; }

__device__ __forceinline__ float relu(float x) { return x > 0.0f ? x : 0.0f; }
__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

template <int TO>
__device__ __forceinline__ void acc_load(f32x16 (&acc)[TO], const float* __restrict__ vec, int half, bool add) {
#pragma unroll
    for (int to = 0; to < TO; to++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float4 v = *reinterpret_cast<const float4*>(vec + 32 * to + 8 * q + 4 * half);
            if (add) {
                acc[to][4 * q + 0] += v.x;
                acc[to][4 * q + 1] += v.y;
                acc[to][4 * q + 2] += v.z;
                acc[to][4 * q + 3] += v.w;
            } else {
                acc[to][4 * q + 0] = v.x;
                acc[to][4 * q + 1] = v.y;
                acc[to][4 * q + 2] = v.z;
                acc[to][4 * q + 3] = v.w;
            }
        }
}

// Weight fragments of eight k-steps (two s4 groups) x TO tiles: 2*TO 16-byte loads per lane, issued together.
template <int N4>
struct Frag {
    float4 v[N4];
    __device__ __forceinline__ void load(const float* __restrict__ frag, int first, int lane) {
#pragma unroll
        for (int i = 0; i < N4; i++) v[i] = *reinterpret_cast<const float4*>(frag + ((first + i) * 64 + lane) * 4);
    }
};

// s4 groups [S0, S0 + NS) of a hidden layer from fragments already in registers (w.v[(s4 - S0) * TO + to])
template <int TO, int S0, int NS>
__device__ __forceinline__ void hidden_part(f32x16 (&out)[TO], const Frag<NS * TO>& w, const f32x16 (&in)[2]) {
#pragma unroll
    for (int s4 = S0; s4 < S0 + NS; s4++) {
#pragma unroll
        for (int to = 0; to < TO; to++) {
            const float4 a = w.v[(s4 - S0) * TO + to];
            const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int T = s4 >> 2, r = (s4 & 3) * 4 + c;
                out[to] = mfma(av[c], relu(in[T][r]), out[to]);
            }
        }
    }
}

// rows row0 .. row0+nvalid-1 (nvalid <= 32) of obs [rows, 9] / avail [rows, A] -> actions[row]; one full wave.
// uniforms: one per row for the inverse-CDF draw, or null for the greedy choice.  A <= 8 * NQ <= 32.
// The weight loads are staged by hand, a layer (or half of one) ahead of the MFMAs that use them, between
// scheduling barriers: left alone, the compiler hoists all 28 of them to the top and spills 112 registers per lane
// to scratch, which at 2048 resident waves is more memory traffic than the env step itself.
template <int NQ>
__device__ __forceinline__ void policy_tile(const float* __restrict__ st, const float* __restrict__ obs,
                                            const int32_t* __restrict__ avail, const float* __restrict__ uniforms,
                                            int64_t* __restrict__ actions, int* __restrict__ actions_lds, int64_t row0,
                                            int nvalid, int n_agents, int A, int lane) {
    const int jrow = lane & 31, half = lane >> 5;
    const bool live = jrow < nvalid;
    const int64_t row = row0 + (live ? jrow : 0);
    const int agent = live ? (int)((uint32_t)(row % n_agents)) : 0;
    f32x16 h1[2], h2[2], lg[1];
    Frag<4> w1;
    Frag<8> wa, wb;
    w1.load(st + F1, 0, lane);
    // the epilogue's inputs ride along with the first round of loads (the row's lane: half 0)
    int32_t av[8 * NQ];
    float u01 = 0.0f;
#pragma unroll
    for (int a = 0; a < 8 * NQ; a++) av[a] = (live && !half && a < A) ? avail[row * A + a] : 1;
    if (uniforms && live && !half) u01 = uniforms[row];
    float b[5];  // B operand of step s: input k = 2s + half of the lane's row (the 10th input is a zero column)
#pragma unroll
    for (int s = 0; s < 5; s++) {
        const int k = 2 * s + half;
        b[s] = (live && k < OBS) ? obs[row * OBS + k] : 0.0f;
    }
    acc_load<2>(h1, st + B1, half, false);
    acc_load<2>(h1, st + ID + agent * H, half, true);
    wa.load(st + F2, 0, lane);  // layer 2, s4 0..3
    wb.load(st + F2, 8, lane);  // layer 2, s4 4..7
    acc_load<2>(h2, st + B2, half, false);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s4 = 0; s4 < 2; s4++)
#pragma unroll
        for (int to = 0; to < 2; to++) {
            const float4 a = w1.v[s4 * 2 + to];
            const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int s = 4 * s4 + c;
                if (s < 5) h1[to] = mfma(av[c], b[s], h1[to]);
            }
        }
    __builtin_amdgcn_sched_barrier(0);
    hidden_part<2, 0, 4>(h2, wa, h1);
    __builtin_amdgcn_sched_barrier(0);
    wa.load(st + F3, 0, lane);  // layer 3, all of it
    acc_load<1>(lg, st + B3, half, false);
    __builtin_amdgcn_sched_barrier(0);
    hidden_part<2, 4, 4>(h2, wb, h1);
    __builtin_amdgcn_sched_barrier(0);
    hidden_part<1, 0, 8>(lg, wa, h2);

    // logits of the row: output 8q + 4*half' + c sits in register 4q + c of the lane of half'
    float x[8 * NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++)
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const float mine = lg[0][4 * q + c];
            const float other = __shfl_xor(mine, 32);
            x[8 * q + c] = half ? other : mine;
            x[8 * q + 4 + c] = half ? mine : other;
        }
    if (half || !live) return;
    // actor_outs[avail == 0] = -1e10 ; softmax ; sample (ppo_agent.py:69-73)
    float mx = -3.0e38f;
#pragma unroll
    for (int a = 0; a < 8 * NQ; a++) {
        if (a < A) {
            if (av[a] == 0) x[a] = -1e10f;
            mx = fmaxf(mx, x[a]);
        }
    }
    float sum = 0.0f;
#pragma unroll
    for (int a = 0; a < 8 * NQ; a++) {
        if (a < A) {
            x[a] = expf(x[a] - mx);
            sum += x[a];
        }
    }
    int pick = 0;
    if (uniforms) {
        // inverse CDF on the caller's uniform; zero-probability (masked) actions are never returned
        const float u = u01 * sum;
        float c = 0.0f;
        int last = 0;
        bool found = false;
#pragma unroll
        for (int a = 0; a < 8 * NQ; a++) {
            if (a < A) {
                const float e = x[a];
                if (e > 0.0f) {
                    c += e;
                    last = a;
                    if (!found && u < c) {
                        pick = a;
                        found = true;
                    }
                }
            }
        }
        if (!found) pick = last;
    } else {
        float best = -1.0f;  // greedy_selector: first maximum (action_selectors.py:18-20); exp is monotone
#pragma unroll
        for (int a = 0; a < 8 * NQ; a++) {
            if (a < A && x[a] > best) {
                best = x[a];
                pick = a;
            }
        }
    }
    actions[row] = pick;
    if (actions_lds) actions_lds[row - row0] = pick;
}

}  // namespace polinl
