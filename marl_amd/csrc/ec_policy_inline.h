// ec_policy_inline.h — the Actor forward + mask + softmax + sampling of ec_mlp.hip's k_policy for up to 64 rows by ONE
// wave, callable from inside another kernel (the looping env kernel evaluates the policy between two steps
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

// diagnostic (EC_STAMPS_POLICY builds, tools/stamps_mlp.py): wall-clock per section of policy_rows, lane 0 of each wave
#ifdef EC_STAMPS_POLICY
#define POL_ST(i)                                                  \
    do {                                                           \
        if (pst) {                                                 \
            const unsigned long long now_ = wall_clock64();        \
            pst[i] += now_ - pst[7];                               \
            pst[7] = now_;                                         \
        }                                                          \
    } while (0)
#define POL_ST_ARG , unsigned long long* pst = nullptr
#else
#define POL_ST(i)
#define POL_ST_ARG
#endif

namespace polinl {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int OBS = 9, H = 64;
// float offsets inside the staged block (ec::stage_policy writes it)
constexpr int F1 = 0;                 // layer 1 fragments: 8 steps x 2 tiles x 64 lanes
constexpr int F2 = F1 + 8 * 2 * 64;   // layer 2: 32 steps x 2 tiles
constexpr int F3 = F2 + 32 * 2 * 64;  // layer 3: 32 steps x TO3 tiles (TO3 = 1: n_actions <= 32, TO3 = 2: <= 64)
__host__ __device__ constexpr int off_B1(int to3) { return F3 + 32 * to3 * 64; }
__host__ __device__ constexpr int off_B2(int to3) { return off_B1(to3) + H; }
__host__ __device__ constexpr int off_B3(int to3) { return off_B2(to3) + H; }        // [32 TO3], zero beyond n_actions
__host__ __device__ constexpr int off_BI(int to3) { return off_B3(to3) + 32 * to3; }  // [n_agents][64]: b1 + the agent's one-hot column of W1 (the bias an agent's rows start from)
constexpr int B1 = off_B1(1), B2 = off_B2(1), B3 = off_B3(1), BI = off_BI(1);
__host__ __device__ constexpr int staged_floats(int n_agents, int to3 = 1) { return off_BI(to3) + n_agents * H; }

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

// N4 consecutive 16-byte fragment words per lane, issued together
template <int N4>
struct Frag {
    float4 v[N4];
    __device__ __forceinline__ void load(const float* __restrict__ frag, int first, int lane) {
#pragma unroll
        for (int i = 0; i < N4; i++) v[i] = *reinterpret_cast<const float4*>(frag + ((first + i) * 64 + lane) * 4);
    }
};

// NS consecutive s4 groups, starting at S0, of a hidden layer with TO output tiles from fragment words
// w.v[(s4 - W0) * TO + to]
template <int TO, int S0, int NS, int W0, int NW>
__device__ __forceinline__ void hidden_part(f32x16 (&out)[TO], const Frag<NW>& w, const f32x16 (&in)[2]) {
#pragma unroll
    for (int s4 = S0; s4 < S0 + NS; s4++) {
#pragma unroll
        for (int to = 0; to < TO; to++) {
            const float4 a = w.v[(s4 - W0) * TO + to];
            const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int T = s4 >> 2, r = (s4 & 3) * 4 + c;
                out[to] = mfma(av[c], relu(in[T][r]), out[to]);
            }
        }
    }
}

// The Actor on rows row0 .. row0+cnt-1 (cnt <= 32 G) of obs [rows, 9] / avail [rows, A] by one full wave; lane l
// returns the action of row row0 + l (and writes it to actions[row0 + l]); -1 beyond cnt.  uniforms: one per row for
// the inverse-CDF draw, or null for the greedy choice.  A <= 8 NQ <= 32 TO3.  s_b23: the biases of layers 2 and 3
// ([64] + [32 TO3]) in LDS.  TO3 = 2 (33..64 actions): layer 3 has two output tiles; its fragments are fetched per group,
// in two halves, and the mask / uniform after the matrix work — the registers of the wide epilogue (x[8 NQ], av[8 NQ])
// leave no room to hold them across it, and a round trip is small next to a 40-node env step.
// A memory round trip costs ~2 us next to the env step's traffic, several times the 106 MFMAs of a 32-row group, so
// the order of things is about round trips: the fragments of layers 1 and 2, the observation inputs and the first
// layer's bias (+ agent-id column, one precombined vector per agent) of BOTH groups are requested up front in one
// round; layer 3's fragments, the mask and the uniform follow while layer 2 runs; nothing else touches global
// memory.  Scheduling barriers pin that order (left alone, the compiler hoists every load to the top and spills a
// hundred registers per lane to scratch).  G = 2 runs its two 32-row groups one after the other on the same
// fragments.
template <int NQ, int G, int TO3 = 1>
__device__ __forceinline__ int policy_rows(const float* __restrict__ st, const float* s_b23,
                                           const float* __restrict__ obs, const int32_t* __restrict__ avail,
                                           const float* __restrict__ uniforms, int64_t* __restrict__ actions,
                                           int64_t row0, int cnt, int n_agents, int A, int lane POL_ST_ARG) {
    const int jrow = lane & 31, half = lane >> 5;
    POL_ST(6);  // (sets the running stamp)
    f32x16 h1[G][2];
    float b[G][5];  // B operand of step s: input k = 2s + half of the group's row (the 10th input is a zero column)
    Frag<4> w1;
    Frag<16> w2;
    Frag<8> w3;
    w1.load(st + F1, 0, lane);
#pragma unroll
    for (int g = 0; g < G; g++) {
        const bool live = 32 * g + jrow < cnt;
        const int64_t row = row0 + (live ? 32 * g + jrow : 0);
        const int agent = live ? (int)((uint32_t)row % (uint32_t)n_agents) : 0;  // rows < 2^31 (ec_create)
#pragma unroll
        for (int s = 0; s < 5; s++) {
            const int k = 2 * s + half;
            b[g][s] = (live && k < OBS) ? obs[row * OBS + k] : 0.0f;
        }
        acc_load<2>(h1[g], st + off_BI(TO3) + agent * H, half, false);
    }
    w2.load(st + F2, 0, lane);
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
                if (s < 5) {
#pragma unroll
                    for (int g = 0; g < G; g++) h1[g][to] = mfma(av[c], b[g][s], h1[g][to]);
                }
            }
        }
    __builtin_amdgcn_sched_barrier(0);
    POL_ST(0);  // first round trip + layer 1
    // second round, behind layer 2 of the first group: layer 3's fragments and the epilogue's inputs.  Lane l is the
    // lane of row row0 + l (G = 2); lanes 0..31 those of the only group (G = 1)
    if constexpr (TO3 == 1) w3.load(st + F3, 0, lane);
    const bool mine = G == 2 ? lane < cnt : (!half && jrow < cnt);
    const int64_t myrow = row0 + (mine ? lane : 0);
    int32_t av[8 * NQ];
    float u01 = 0.0f;
    if constexpr (TO3 == 1) {
#pragma unroll
        for (int a = 0; a < 8 * NQ; a++) av[a] = (mine && a < A) ? avail[myrow * A + a] : 1;
        if (uniforms && mine) u01 = uniforms[myrow];
    }
    __builtin_amdgcn_sched_barrier(0);
    float x[8 * NQ];
#pragma unroll
    for (int g = 0; g < G; g++) {
        f32x16 h2[2], lg[TO3];
        acc_load<2>(h2, s_b23, half, false);
        hidden_part<2, 0, 8, 0, 16>(h2, w2, h1[g]);
        __builtin_amdgcn_sched_barrier(0);
        POL_ST(1);  // layer 2
        acc_load<TO3>(lg, s_b23 + H, half, false);
        if constexpr (TO3 == 1) {
            hidden_part<1, 0, 8, 0, 8>(lg, w3, h2);
        } else {
            Frag<8> wa, wb;
            wa.load(st + F3, 0, lane);
            wb.load(st + F3, 8, lane);
            hidden_part<2, 0, 4, 0, 8>(lg, wa, h2);
            hidden_part<2, 4, 4, 4, 8>(lg, wb, h2);
        }
        POL_ST(2);  // layer 3 (+ its fragments' arrival)
        // Output 8q + 4h + c of the group's row j sits in register 4q + c of lane (j, h); the row's lane is (j, 0)
        // for G = 1 and (j, g) for G = 2: it keeps the four outputs of its own half and receives the other four
        // from the lane 32 away.
#pragma unroll
        for (int q = 0; q < NQ; q++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const float own = lg[q >> 2][4 * (q & 3) + c];
                const float recv = __shfl_xor(own, 32);
                if (G == 1 || half == g) {
                    x[8 * q + c] = half ? recv : own;
                    x[8 * q + 4 + c] = half ? own : recv;
                }
            }
        __builtin_amdgcn_sched_barrier(0);
        POL_ST(3);  // logits exchange
    }
    if (!mine) return -1;
    if constexpr (TO3 == 2) {
#pragma unroll
        for (int a = 0; a < 8 * NQ; a++) av[a] = a < A ? avail[myrow * A + a] : 1;
        if (uniforms) u01 = uniforms[myrow];
    }
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
    actions[myrow] = pick;
    POL_ST(4);  // mask + softmax + draw
    return pick;
}

}  // namespace polinl
