// ec_mlp.hip — the three small-MLP evaluations of the rollout / reward path, gfx950.
//
//   ec_policy_forward    Actor / PolicyNet forward + availability mask + softmax + sampling
//                        (rl/policy_gradient_rl/ppo/ppo_network.py:21-25, ppo_agent.py:61-74,
//                         dgail_network.py:6-19, dgail_agent.py:114-126, utils/action_selectors.py:11-20)
//   ec_gail_reward       GAILDiscriminator forward + reward (dgail_network.py:37-50, dgail_agent.py:482-485,501-518)
//   ec_diffusion_reward  DiffusionDiscriminator.compute_loss for both conditions with the analytic input
//                        gradient for the R1 term, D_phi and the reward (dgail_network.py:71-108,
//                        dgail_agent.py:487-499)
//
// All three are chains of 64-wide layers over millions of independent rows, in fp32 (the nets are trained and
// checked in fp32).  They run on the fp32 matrix instruction v_mfma_f32_32x32x2_f32 with a wave owning 64 rows
// (two groups of 32):
//     D[out 0..31][row 0..31] += A[out][k 0..1] * B[k][row]
//   * A (weights) sits in LDS in FRAGMENT ORDER: the word lane l needs for k-step s is at [s/4][tile][l][s%4], so
//     four k-steps cost one conflict-free ds_read_b128 and no weight ever occupies a long-lived register;
//   * B (activations) never leaves registers between layers.  The accumulator of the 32x32 tile leaves lane
//     (row = l&31, half = l>>5) holding outputs 8q + 4*half + c (q, c in 0..3) of its row, and B for step s wants
//     from that same lane ONE input of that same row for k = half.  So the next layer simply walks its inputs in
//     the order k(s, half) = 32T + 8q + 4*half + c with s = 16T + 4q + c — i.e. B for step s is accumulator
//     register s of the previous layer — and the weights are staged in LDS in that same order.  No transposes,
//     no shuffles, no LDS round trip between layers.
//   * one-hot inputs (agent id, action) are a column select added to the accumulator's bias; the diffusion net's
//     time/condition embeddings collapse into four precomputed bias vectors; its R1 term uses the closed form
//         d(sum eps)/dx_t = W1x^T (m1 .* (W2^T (m2 .* colsum W3)))          (m = ReLU masks)
//     which is two more MFMA layers on transposed fragments instead of a second autograd pass.
// The fp32 MFMA is an exact fmaf chain (MI355X_MICROARCH.md, "Matrix cores"), at the fp32 vector peak, without the
// vector unit's LDS-broadcast traffic for weights (1 ds_read_b128 per 8 MFMAs here).
#include "ec_mfma.h"
#include "ec_policy_inline.h"

namespace {

// ------------------------------------------------------------------------------------------ policy
template <int TO3>
__global__ void __launch_bounds__(NT)
k_policy(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
         const float* __restrict__ b2, const float* __restrict__ w3, const float* __restrict__ b3,
         const float* __restrict__ obs, const int32_t* __restrict__ avail, const float* __restrict__ uniforms,
         int64_t R, int n_agents, int A, float* __restrict__ logits_out, float* __restrict__ probs_out,
         int64_t* __restrict__ actions_out) {
    extern __shared__ __align__(16) float lds[];
    constexpr int LP = 32 * TO3 + 1;  // logits tile row stride (odd: conflict-free row-per-lane access)
    float* s_f1 = lds;                        // 8 steps x 2 tiles
    float* s_f2 = s_f1 + 8 * 2 * 64;          // 32 steps x 2 tiles
    float* s_f3 = s_f2 + 32 * 2 * 64;         // 32 steps x TO3 tiles
    float* s_b1 = s_f3 + 32 * TO3 * 64;       // [64]
    float* s_b2 = s_b1 + H;                   // [64]
    float* s_b3 = s_b2 + H;                   // [32*TO3]
    float* s_id = s_b3 + 32 * TO3;            // [n_agents][64] one-hot agent-id columns of W1
    float* s_wave = s_id + (size_t)n_agents * H;
    const int in1 = OBS + n_agents;
    stage_frag(s_f1, w1, in1, 1, H, OBS, 8, 2, false);
    stage_frag(s_f2, w2, H, 1, H, H, 32, 2, true);
    stage_frag(s_f3, w3, H, 1, A, H, 32, TO3, true);
    for (int i = threadIdx.x; i < H; i += blockDim.x) {
        s_b1[i] = b1[i];
        s_b2[i] = b2[i];
    }
    for (int i = threadIdx.x; i < 32 * TO3; i += blockDim.x) s_b3[i] = i < A ? b3[i] : 0.0f;
    for (int i = threadIdx.x; i < n_agents * H; i += blockDim.x) {
        const int ag = i / H, o = i - ag * H;
        s_id[i] = w1[(size_t)o * in1 + OBS + ag];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int jrow = lane & 31, half = lane >> 5;
    float* s_obs = s_wave + (size_t)wave * (OBS_TILE + 64 * LP);
    float* s_lg = s_obs + OBS_TILE;
    const int64_t n_tiles = (R + 63) / 64;
    for (int64_t tile = (int64_t)blockIdx.x * NW + wave; tile < n_tiles; tile += (int64_t)gridDim.x * NW) {
        const int64_t rowbase = tile * 64;
        wave_sync();
        load_obs_tile(s_obs, obs, rowbase, R, lane);
        wave_sync();
        f32x16 h1[G][2], h2[G][2];
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int64_t row = rowbase + 32 * g + jrow;
            const int agent = row < R ? (int)(row % n_agents) : 0;  // one-hot agent id (ppo_agent.py:64-65)
            acc_init<2>(h1[g], s_b1, half);
            acc_add<2>(h1[g], s_id + agent * H, half);
            acc_init<2>(h2[g], s_b2, half);
        }
        layer_obs(h1, s_f1, s_obs, lane);
        layer_hidden<2>(h2, s_f2, h1, lane, [](int, int, int, float x) { return relu(x); });
        {
            f32x16 lg[G][TO3];
#pragma unroll
            for (int g = 0; g < G; g++) acc_init<TO3>(lg[g], s_b3, half);
            layer_hidden<TO3>(lg, s_f3, h2, lane, [](int, int, int, float x) { return relu(x); });
#pragma unroll
            for (int g = 0; g < G; g++)
#pragma unroll
                for (int to = 0; to < TO3; to++)
#pragma unroll
                    for (int r = 0; r < 16; r++)
                        s_lg[(32 * g + jrow) * LP + 32 * to + 8 * (r >> 2) + 4 * half + (r & 3)] = lg[g][to][r];
        }
        wave_sync();
        // epilogue: lane = row
        const int64_t row = rowbase + lane;
        if (row < R) {
            float* my = s_lg + lane * LP;
            // actor_outs[avail == 0] = -1e10 ; softmax ; sample (ppo_agent.py:69-73)
            float mx = -3.0e38f;
            for (int a = 0; a < A; a++) {
                float x = my[a];
                if (avail && avail[row * A + a] == 0) x = -1e10f;
                my[a] = x;
                mx = fmaxf(mx, x);
                if (logits_out) logits_out[row * A + a] = x;
            }
            float sum = 0.0f;
            for (int a = 0; a < A; a++) {
                const float e = expf(my[a] - mx);
                my[a] = e;
                sum += e;
            }
            int pick = 0;
            if (uniforms) {
                // inverse CDF on the caller's uniform; zero-probability (masked) actions are never returned
                const float u = uniforms[row] * sum;
                float c = 0.0f;
                int last = 0;
                bool found = false;
                for (int a = 0; a < A; a++) {
                    const float e = my[a];
                    if (e > 0.0f) {
                        c += e;
                        last = a;
                        if (!found && u < c) {
                            pick = a;
                            found = true;
                        }
                    }
                }
                if (!found) pick = last;
            } else {
                float best = -1.0f;  // greedy_selector: first maximum (action_selectors.py:18-20); exp is monotone
                for (int a = 0; a < A; a++)
                    if (my[a] > best) {
                        best = my[a];
                        pick = a;
                    }
            }
            if (probs_out) {
                const float inv = 1.0f / sum;
                for (int a = 0; a < A; a++) probs_out[row * A + a] = my[a] * inv;
            }
            if (actions_out) actions_out[row] = pick;
        }
    }
}

// The same fragments / biases / agent-id columns k_policy stages into LDS, written once to global memory for the
// policy evaluation inside the looping env kernel (ec_policy_inline.h reads them in place).
__global__ void __launch_bounds__(NT)
k_stage_policy(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
               const float* __restrict__ b2, const float* __restrict__ w3, const float* __restrict__ b3, int n_agents, int A,
               int to3, float* __restrict__ dst) {
    static_assert(polinl::OBS == OBS && polinl::H == H, "ec_policy_inline.h is for the same net");
    const int in1 = OBS + n_agents;
    stage_frag(dst + polinl::F1, w1, in1, 1, H, OBS, 8, 2, false);
    stage_frag(dst + polinl::F2, w2, H, 1, H, H, 32, 2, true);
    stage_frag(dst + polinl::F3, w3, H, 1, A, H, 32, to3, true);
    for (int i = threadIdx.x; i < H; i += blockDim.x) {
        dst[polinl::off_B1(to3) + i] = b1[i];
        dst[polinl::off_B2(to3) + i] = b2[i];
    }
    for (int i = threadIdx.x; i < 32 * to3; i += blockDim.x) dst[polinl::off_B3(to3) + i] = i < A ? b3[i] : 0.0f;
    for (int i = threadIdx.x; i < n_agents * H; i += blockDim.x) {
        const int ag = i / H, o = i - ag * H;
        dst[polinl::off_BI(to3) + i] = b1[o] + w1[(size_t)o * in1 + OBS + ag];  // k_policy: acc = b1; acc += column
    }
}

// ------------------------------------------------------------------------------------------ GAIL discriminator reward
__global__ void __launch_bounds__(NT)
k_gail_reward(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
              const float* __restrict__ b2, const float* __restrict__ w3, const float* __restrict__ b3,
              const float* __restrict__ obs, const int64_t* __restrict__ actions,
              const float* __restrict__ env_reward, int64_t R, int A, float weight, float env_coef,
              float* __restrict__ prob_out, float* __restrict__ out) {
    extern __shared__ __align__(16) float lds[];
    float* s_f1 = lds;                   // 8 steps x 2 tiles
    float* s_f2 = s_f1 + 8 * 2 * 64;     // 32 steps x 2 tiles
    float* s_b1 = s_f2 + 32 * 2 * 64;    // [64]
    float* s_b2 = s_b1 + H;              // [64]
    float* s_w3 = s_b2 + H;              // [64]
    float* s_act = s_w3 + H;             // [A][64] one-hot action columns of W1
    float* s_wave = s_act + (size_t)A * H;
    const int in1 = OBS + A;
    stage_frag(s_f1, w1, in1, 1, H, OBS, 8, 2, false);
    stage_frag(s_f2, w2, H, 1, H, H, 32, 2, true);
    for (int i = threadIdx.x; i < H; i += blockDim.x) {
        s_b1[i] = b1[i];
        s_b2[i] = b2[i];
        s_w3[i] = w3[i];
    }
    for (int i = threadIdx.x; i < A * H; i += blockDim.x) {
        const int a = i / H, o = i - a * H;
        s_act[i] = w1[(size_t)o * in1 + OBS + a];
    }
    __syncthreads();
    const float bias3 = b3[0];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int jrow = lane & 31, half = lane >> 5;
    float* s_obs = s_wave + (size_t)wave * OBS_TILE;
    const int64_t n_tiles = (R + 63) / 64;
    for (int64_t tile = (int64_t)blockIdx.x * NW + wave; tile < n_tiles; tile += (int64_t)gridDim.x * NW) {
        const int64_t rowbase = tile * 64;
        wave_sync();
        load_obs_tile(s_obs, obs, rowbase, R, lane);
        wave_sync();
        f32x16 h1[G][2], h2[G][2];
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int64_t row = rowbase + 32 * g + jrow;
            int a = row < R ? (int)actions[row] : 0;
            a = a < 0 ? 0 : (a >= A ? A - 1 : a);  // one-hot action (dgail_agent.py:199-201)
            acc_init<2>(h1[g], s_b1, half);
            acc_add<2>(h1[g], s_act + a * H, half);
            acc_init<2>(h2[g], s_b2, half);
        }
        layer_obs(h1, s_f1, s_obs, lane);
        layer_hidden<2>(h2, s_f2, h1, lane, [](int, int, int, float x) { return relu(x); });
        float z[G];
#pragma unroll
        for (int g = 0; g < G; g++) {
            float part = 0.0f;
#pragma unroll
            for (int to = 0; to < 2; to++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const float4 w = *reinterpret_cast<const float4*>(s_w3 + 32 * to + 8 * q + 4 * half);
                    part = fmaf(w.x, relu(h2[g][to][4 * q + 0]), part);
                    part = fmaf(w.y, relu(h2[g][to][4 * q + 1]), part);
                    part = fmaf(w.z, relu(h2[g][to][4 * q + 2]), part);
                    part = fmaf(w.w, relu(h2[g][to][4 * q + 3]), part);
                }
            z[g] = part + __shfl_xor(part, 32) + bias3;
        }
        const int64_t row = rowbase + lane;  // lane (jrow, half) finishes group `half`
        if (row < R) {
            const float zz = half ? z[1] : z[0];
            float p = 1.0f / (1.0f + expf(-zz));
            if (prob_out) prob_out[row] = p;
            p = fminf(fmaxf(p, 1e-8f), 1.0f - 1e-8f);               // .clamp(1e-8, 1 - 1e-8), dgail_agent.py:484
            const float base_r = -logf(1.0f - p);                    // :485
            float v = base_r * weight;                               // :502-504
            if (env_reward) v = v + env_coef * env_reward[row];      // :515-516
            out[row] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------ diffusion discriminator reward
// One condition c of compute_loss (dgail_network.py:82-108) for the wave's 64 rows:
//   x_t = sa + t * noise ; eps = MLP([x_t | E_time[t] | E_cond[c]]) ; L = mean_d (eps-noise)^2 + lambda * |d(sum eps)/dx_t|^2
struct DiffLds {
    const float* f1;     // x-part of W1, raw-input order, steps1 x 2 tiles
    const float* f2;     // W2, 32 x 2
    const float* f3;     // W3, 32 x TS
    const float* f2t;    // W2^T, 32 x 2
    const float* f1t;    // W1x^T, 32 x TS
    const float* bias;   // [2(t)][2(c)][64]  b1 + W1t E_time[t] + W1c E_cond[c]
    const float* b2;     // [64]
    const float* b3;     // [32*TS]
    const float* w3sum;  // [64] column sums of W3 = d(sum eps)/d(relu h2)
};

template <int TS>
__device__ __forceinline__ void diffusion_loss(float (&loss)[G], const DiffLds& W, const float* __restrict__ s_obs,
                                               float* __restrict__ s_noise, const float* __restrict__ noise,
                                               const uint8_t* __restrict__ tt, const int (&act)[G], int64_t rowbase, int64_t R,
                                               int c, int SA, int SAp, int steps1, float lambda, int lane) {
    const int jrow = lane & 31, half = lane >> 5;
    // the wave's noise tile [64][SAp] (SAp odd) and its rows' t
    wave_sync();
    {
        const int64_t nvalid = (R - rowbase < 64 ? R - rowbase : 64) * (int64_t)SA;
        int r = 0, k = lane;
        while (k >= SA) {
            k -= SA;
            r++;
        }
        for (int i = lane; i < 64 * SA; i += 64) {
            s_noise[r * SAp + k] = i < nvalid ? noise[rowbase * SA + i] : 0.0f;
            k += 64;
            while (k >= SA) {
                k -= SA;
                r++;
            }
        }
    }
    wave_sync();
    int t[G];
    f32x16 h1[G][2], h2[G][2];
#pragma unroll
    for (int g = 0; g < G; g++) {
        const int64_t row = rowbase + 32 * g + jrow;
        t[g] = (row < R && tt[row]) ? 1 : 0;
        acc_init<2>(h1[g], W.bias + (t[g] * 2 + c) * H, half);
        acc_init<2>(h2[g], W.b2, half);
    }
    // layer 1 over x_t = (obs | one-hot action) + t * noise
    for (int s4 = 0; s4 < steps1 / 4; s4++) {
#pragma unroll
        for (int to = 0; to < 2; to++) {
            const float4 a = *reinterpret_cast<const float4*>(W.f1 + ((s4 * 2 + to) * 64 + lane) * 4);
            const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int cc = 0; cc < 4; cc++)
#pragma unroll
                for (int g = 0; g < G; g++) {
                    const int k = 2 * (4 * s4 + cc) + half, r = 32 * g + jrow;
                    float b = 0.0f;
                    if (k < SA) {
                        const float sa = k < OBS ? s_obs[r * OBS + k] : (k - OBS == act[g] ? 1.0f : 0.0f);
                        b = sa + (float)t[g] * s_noise[r * SAp + k];
                    }
                    h1[g][to] = MFMA(av[cc], b, h1[g][to]);
                }
        }
    }
    unsigned m1[G], m2[G];
#pragma unroll
    for (int g = 0; g < G; g++) {
        m1[g] = 0;
#pragma unroll
        for (int T = 0; T < 2; T++)
#pragma unroll
            for (int r = 0; r < 16; r++) m1[g] |= h1[g][T][r] > 0.0f ? (1u << (16 * T + r)) : 0u;
    }
    layer_hidden<2>(h2, W.f2, h1, lane, [](int, int, int, float x) { return relu(x); });
#pragma unroll
    for (int g = 0; g < G; g++) {
        m2[g] = 0;
#pragma unroll
        for (int T = 0; T < 2; T++)
#pragma unroll
            for (int r = 0; r < 16; r++) m2[g] |= h2[g][T][r] > 0.0f ? (1u << (16 * T + r)) : 0u;
    }
    float mse[G];
    {
        f32x16 eps[G][TS];
#pragma unroll
        for (int g = 0; g < G; g++) acc_init<TS>(eps[g], W.b3, half);
        layer_hidden<TS>(eps, W.f3, h2, lane, [](int, int, int, float x) { return relu(x); });
#pragma unroll
        for (int g = 0; g < G; g++) {
            float part = 0.0f;
#pragma unroll
            for (int to = 0; to < TS; to++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int k = 32 * to + 8 * (r >> 2) + 4 * half + (r & 3);
                    if (k < SA) {
                        const float dlt = eps[g][to][r] - s_noise[(32 * g + jrow) * SAp + k];
                        part = fmaf(dlt, dlt, part);
                    }
                }
            mse[g] = (part + __shfl_xor(part, 32)) / (float)SA;
        }
    }
    // backward of sum(eps) to x_t: u = W2^T (m2 .* colsum W3), grad = W1x^T (m1 .* u)
    {
        // h2 now carries colsum(W3) at the lane's output positions so that the mask select is the B operand
#pragma unroll
        for (int g = 0; g < G; g++) acc_init<2>(h2[g], W.w3sum, half);
        acc_zero<2>(h1[0]);
        acc_zero<2>(h1[1]);
        layer_hidden<2>(h1, W.f2t, h2, lane,
                        [&](int g, int T, int r, float x) { return ((m2[g] >> (16 * T + r)) & 1u) ? x : 0.0f; });
    }
    {
        f32x16 gr[G][TS];
        acc_zero<TS>(gr[0]);
        acc_zero<TS>(gr[1]);
        layer_hidden<TS>(gr, W.f1t, h1, lane,
                         [&](int g, int T, int r, float x) { return ((m1[g] >> (16 * T + r)) & 1u) ? x : 0.0f; });
#pragma unroll
        for (int g = 0; g < G; g++) {
            float part = 0.0f;
#pragma unroll
            for (int to = 0; to < TS; to++)
#pragma unroll
                for (int r = 0; r < 16; r++) part = fmaf(gr[g][to][r], gr[g][to][r], part);  // rows >= SA are exact zeros
            const float r1 = part + __shfl_xor(part, 32);
            loss[g] = mse[g] + r1 * lambda;
        }
    }
}

template <int TS>
__global__ void __launch_bounds__(NT)
k_diffusion_reward(const float* __restrict__ w1 /*[H][SA+2H]*/, const float* __restrict__ b1,
                   const float* __restrict__ w2, const float* __restrict__ b2, const float* __restrict__ w3 /*[SA][H]*/,
                   const float* __restrict__ b3, const float* __restrict__ time_embed /*[2][H]*/,
                   const float* __restrict__ cond_embed /*[2][H]*/, const float* __restrict__ obs,
                   const int64_t* __restrict__ actions, const uint8_t* __restrict__ t_exp,
                   const float* __restrict__ noise_exp, const uint8_t* __restrict__ t_agent,
                   const float* __restrict__ noise_agent, int64_t R, int A, float lambda,
                   float* __restrict__ loss_exp_out, float* __restrict__ loss_agent_out, float* __restrict__ d_out,
                   float* __restrict__ reward_out) {
    extern __shared__ __align__(16) float lds[];
    const int SA = OBS + A;
    const int in1 = SA + 2 * H;
    const int steps1 = (((SA + 1) / 2) + 3) & ~3;
    const int SAp = SA | 1;
    float* s_f1 = lds;
    float* s_f2 = s_f1 + steps1 * 2 * 64;
    float* s_f3 = s_f2 + 32 * 2 * 64;
    float* s_f2t = s_f3 + 32 * TS * 64;
    float* s_f1t = s_f2t + 32 * 2 * 64;
    float* s_bias = s_f1t + 32 * TS * 64;  // [4][64]
    float* s_b2 = s_bias + 4 * H;
    float* s_b3 = s_b2 + H;                // [32*TS]
    float* s_w3sum = s_b3 + 32 * TS;       // [64]
    float* s_wave = s_w3sum + H;
    stage_frag(s_f1, w1, in1, 1, H, SA, steps1, 2, false);
    stage_frag(s_f2, w2, H, 1, H, H, 32, 2, true);
    stage_frag(s_f3, w3, H, 1, SA, H, 32, TS, true);
    stage_frag(s_f2t, w2, 1, H, H, H, 32, 2, true);      // M[out=j][k=i] = W2[i][j]
    stage_frag(s_f1t, w1, 1, in1, SA, H, 32, TS, true);  // M[out=k_in][k=j] = W1[j][k_in]
    for (int idx = threadIdx.x; idx < 4 * H; idx += blockDim.x) {
        const int tc = idx / H, i = idx - tc * H, t = tc >> 1, c = tc & 1;
        float acc = b1[i];
        for (int j = 0; j < H; j++) acc = fmaf(w1[(size_t)i * in1 + SA + j], time_embed[t * H + j], acc);
        for (int j = 0; j < H; j++) acc = fmaf(w1[(size_t)i * in1 + SA + H + j], cond_embed[c * H + j], acc);
        s_bias[idx] = acc;
    }
    for (int i = threadIdx.x; i < H; i += blockDim.x) {
        s_b2[i] = b2[i];
        float cs = 0.0f;
        for (int k = 0; k < SA; k++) cs += w3[(size_t)k * H + i];
        s_w3sum[i] = cs;
    }
    for (int k = threadIdx.x; k < 32 * TS; k += blockDim.x) s_b3[k] = k < SA ? b3[k] : 0.0f;
    __syncthreads();
    const DiffLds W{s_f1, s_f2, s_f3, s_f2t, s_f1t, s_bias, s_b2, s_b3, s_w3sum};

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int jrow = lane & 31, half = lane >> 5;
    float* s_obs = s_wave + (size_t)wave * (OBS_TILE + 64 * SAp);
    float* s_noise = s_obs + OBS_TILE;
    const int64_t n_tiles = (R + 63) / 64;
    for (int64_t tile = (int64_t)blockIdx.x * NW + wave; tile < n_tiles; tile += (int64_t)gridDim.x * NW) {
        const int64_t rowbase = tile * 64;
        wave_sync();
        load_obs_tile(s_obs, obs, rowbase, R, lane);
        int act[G];
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int64_t row = rowbase + 32 * g + jrow;
            int a = row < R ? (int)actions[row] : 0;
            act[g] = a < 0 ? 0 : (a >= A ? A - 1 : a);  // (obs | one-hot action), dgail_agent.py:199-201
        }
        // c = 1: expert condition, c = 0: agent condition (dgail_agent.py:488-492); each has its own (t, noise)
        float Le[G], La[G];
        diffusion_loss<TS>(Le, W, s_obs, s_noise, noise_exp, t_exp, act, rowbase, R, 1, SA, SAp, steps1, lambda, lane);
        diffusion_loss<TS>(La, W, s_obs, s_noise, noise_agent, t_agent, act, rowbase, R, 0, SA, SAp, steps1, lambda, lane);
        const int64_t row = rowbase + lane;  // lane (jrow, half) finishes group `half`
        if (row < R) {
            const float le = half ? Le[1] : Le[0], la = half ? La[1] : La[0];
            const float pe = expf(-le), pa = expf(-la);
            const float dphi = pe / (pe + pa + 1e-8f);  // dgail_agent.py:495-497
            if (loss_exp_out) loss_exp_out[row] = le;
            if (loss_agent_out) loss_agent_out[row] = la;
            if (d_out) d_out[row] = dphi;
            if (reward_out) reward_out[row] = -logf(1.0f - dphi + 1e-8f);  // :499
        }
    }
}

int grid_for(int64_t R, int blocks_per_cu) {
    const int64_t blocks = (R + 64 * NW - 1) / (64 * NW);
    const int64_t cap = 256 * (int64_t)blocks_per_cu;  // beyond that waves stride over tiles; weights are staged once per block
    return (int)(blocks < cap ? (blocks < 1 ? 1 : blocks) : cap);
}

constexpr size_t LDS_MAX = 160 * 1024;

template <typename K>
int set_lds(K kernel, size_t lds) {
    EC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return EC_OK;
}

int blocks_per_cu(size_t lds) {
    const int b = (int)(LDS_MAX / (lds + 1024));
    return b < 1 ? 1 : (b > 2 ? 2 : b);
}

}  // namespace

int ec::stage_policy(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                     int n_agents, int n_actions, float* dst, hipStream_t stream) {
    if (!w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !dst) return ec::fail(EC_EINVAL, "stage_policy: null weights");
    if (n_actions < 1 || n_actions > 64 || n_agents < 1)
        return ec::fail(EC_EINVAL, "in-kernel policy: n_actions = %d (supported: 1..64)", n_actions);
    hipLaunchKernelGGL(k_stage_policy, dim3(1), dim3(NT), 0, stream, w1, b1, w2, b2, w3, b3, n_agents, n_actions,
                       n_actions <= 32 ? 1 : 2, dst);
    EC_HIP(hipGetLastError());
    return EC_OK;
}

extern "C" {

int ec_policy_forward(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                      const float* b3, const float* obs, const int32_t* avail, const float* uniforms, int64_t rows,
                      int32_t n_agents, int32_t n_actions, int32_t hidden, float* logits, float* probs,
                      int64_t* actions, void* stream) {
    if (rows == 0) return EC_OK;
    if (rows < 0) return ec::fail(EC_EINVAL, "ec_policy_forward: rows < 0");
    if (!w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !obs) return ec::fail(EC_EINVAL, "ec_policy_forward: null pointer");
    if (hidden != H) return ec::fail(EC_EINVAL, "ec_policy_forward: hidden_dim %d unsupported (this build: %d)", hidden, H);
    if (n_agents < 1 || n_actions < 1 || n_actions > 64)
        return ec::fail(EC_EINVAL, "ec_policy_forward: n_actions %d out of range [1,64]", n_actions);
    const int TO3 = n_actions <= 32 ? 1 : 2;
    const int LP = 32 * TO3 + 1;
    const size_t lds = sizeof(float) * ((size_t)8 * 2 * 64 + 32 * 2 * 64 + (size_t)32 * TO3 * 64 + 2 * H + 32 * TO3 +
                                        (size_t)n_agents * H + (size_t)NW * (OBS_TILE + 64 * LP));
    if (lds > LDS_MAX) return ec::fail(EC_EINVAL, "ec_policy_forward: %d agents need %zu B of LDS (max %zu)", n_agents, lds, LDS_MAX);
    const int grid = grid_for(rows, blocks_per_cu(lds));
    if (TO3 == 1) {
        if (set_lds(k_policy<1>, lds)) return EC_EHIP;
        hipLaunchKernelGGL(k_policy<1>, dim3(grid), dim3(NT), lds, (hipStream_t)stream, w1, b1, w2, b2, w3, b3, obs, avail,
                           uniforms, rows, (int)n_agents, (int)n_actions, logits, probs, actions);
    } else {
        if (set_lds(k_policy<2>, lds)) return EC_EHIP;
        hipLaunchKernelGGL(k_policy<2>, dim3(grid), dim3(NT), lds, (hipStream_t)stream, w1, b1, w2, b2, w3, b3, obs, avail,
                           uniforms, rows, (int)n_agents, (int)n_actions, logits, probs, actions);
    }
    EC_HIP(hipGetLastError());
    return EC_OK;
}

int ec_gail_reward(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                   const float* obs, const int64_t* actions, const float* env_reward, int64_t rows, int32_t n_actions,
                   int32_t hidden, float weight, float env_coef, float* prob, float* out, void* stream) {
    if (rows == 0) return EC_OK;
    if (rows < 0) return ec::fail(EC_EINVAL, "ec_gail_reward: rows < 0");
    if (!w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !obs || !actions || !out) return ec::fail(EC_EINVAL, "ec_gail_reward: null pointer");
    if (hidden != H) return ec::fail(EC_EINVAL, "ec_gail_reward: hidden_dim %d unsupported (this build: %d)", hidden, H);
    if (n_actions < 1) return ec::fail(EC_EINVAL, "ec_gail_reward: n_actions out of range");
    const size_t lds = sizeof(float) * ((size_t)8 * 2 * 64 + 32 * 2 * 64 + 3 * H + (size_t)n_actions * H + (size_t)NW * OBS_TILE);
    if (lds > LDS_MAX) return ec::fail(EC_EINVAL, "ec_gail_reward: %d actions need %zu B of LDS (max %zu)", n_actions, lds, LDS_MAX);
    const int grid = grid_for(rows, blocks_per_cu(lds));
    if (set_lds(k_gail_reward, lds)) return EC_EHIP;
    hipLaunchKernelGGL(k_gail_reward, dim3(grid), dim3(NT), lds, (hipStream_t)stream, w1, b1, w2, b2, w3, b3, obs, actions,
                       env_reward, rows, (int)n_actions, weight, env_coef, prob, out);
    EC_HIP(hipGetLastError());
    return EC_OK;
}

int ec_diffusion_reward(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                        const float* b3, const float* time_embed, const float* cond_embed, const float* obs,
                        const int64_t* actions, const uint8_t* t_expert, const float* noise_expert,
                        const uint8_t* t_agent, const float* noise_agent, int64_t rows, int32_t n_actions, int32_t hidden,
                        float r1_lambda, float* loss_expert, float* loss_agent, float* d_phi, float* reward, void* stream) {
    if (rows == 0) return EC_OK;
    if (rows < 0) return ec::fail(EC_EINVAL, "ec_diffusion_reward: rows < 0");
    if (!w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !time_embed || !cond_embed || !obs || !actions || !t_expert ||
        !noise_expert || !t_agent || !noise_agent)
        return ec::fail(EC_EINVAL, "ec_diffusion_reward: null pointer");
    if (hidden != H) return ec::fail(EC_EINVAL, "ec_diffusion_reward: hidden_dim %d unsupported (this build: %d)", hidden, H);
    const int SA = OBS + n_actions;
    if (n_actions < 1 || SA > 64) return ec::fail(EC_EINVAL, "ec_diffusion_reward: obs+actions = %d exceeds 64", SA);
    const int TS = SA <= 32 ? 1 : 2;
    const int steps1 = (((SA + 1) / 2) + 3) & ~3;
    const int SAp = SA | 1;
    const size_t lds = sizeof(float) * ((size_t)steps1 * 2 * 64 + 2 * (32 * 2 * 64) + 2 * ((size_t)32 * TS * 64) + 4 * H + H +
                                        32 * TS + H + (size_t)NW * (OBS_TILE + 64 * SAp));
    if (lds > LDS_MAX) return ec::fail(EC_EINVAL, "ec_diffusion_reward: LDS need %zu too large", lds);
    const int grid = grid_for(rows, blocks_per_cu(lds));
    if (TS == 1) {
        if (set_lds(k_diffusion_reward<1>, lds)) return EC_EHIP;
        hipLaunchKernelGGL(k_diffusion_reward<1>, dim3(grid), dim3(NT), lds, (hipStream_t)stream, w1, b1, w2, b2, w3, b3,
                           time_embed, cond_embed, obs, actions, t_expert, noise_expert, t_agent, noise_agent, rows,
                           (int)n_actions, r1_lambda, loss_expert, loss_agent, d_phi, reward);
    } else {
        if (set_lds(k_diffusion_reward<2>, lds)) return EC_EHIP;
        hipLaunchKernelGGL(k_diffusion_reward<2>, dim3(grid), dim3(NT), lds, (hipStream_t)stream, w1, b1, w2, b2, w3, b3,
                           time_embed, cond_embed, obs, actions, t_expert, noise_expert, t_agent, noise_agent, rows,
                           (int)n_actions, r1_lambda, loss_expert, loss_agent, d_phi, reward);
    }
    EC_HIP(hipGetLastError());
    return EC_OK;
}

}  // extern "C"
