// ec_ppo.hip — fused forward + loss + backward of the PPO actor and critic (obs | one-hot agent id) -> 64 -> 64 -> {A, 1}
// MLPs over the whole rollout batch, gfx950.
//
// Replaces, per optimiser step of PPOAgent.train (rl/policy_gradient_rl/ppo/ppo_agent.py:104-134; same losses in
// gail_agent.py:198-226): evaluate_actions (:76-83 — masked softmax, Categorical log-prob and entropy), the clipped
// surrogate + entropy objective, the masked critic MSE, and the autograd backward of both nets.  One launch per net
// reads 76 B (actor) / 48 B (critic) per row and writes the parameter gradients; nothing per-row is materialised in HBM
// (the torch path streams ~60 [rows,64] fp32 tensors per epoch and spends most of its time in split-K-less weight-
// gradient GEMMs and bias reductions over 4 M rows).
//
// Scheme (fp32 MFMA 32x32x2, a wave owns 32 rows per pass; see ec_mlp.hip for the forward chaining):
//   forward      h1 = W1x.obs + b1 + W1id[:, agent];  h2 = W2.relu(h1) + b2;  out = W3.relu(h2) + b3
//   head         per row in registers (the row's outputs live in lanes l and l^32): softmax / log-prob / entropy /
//                ratio / clip (actor) or squared error (critic) and d loss / d out
//   backward     d2 = (W3^T d3) .* [h2>0],  d1 = (W2^T d2) .* [h1>0]   — the same register chaining on transposed
//                fragments
//   weight grads dW = sum_rows d (x) act: the row index is the MFMA's K dimension, so d and act are transposed through
//                a per-wave LDS tile ([index][row], stride 33) and accumulated in persistent MFMA accumulators; bias
//                gradients ride along as a per-lane running sum of the A operand.  One atomic add per accumulator
//                register at the end of the wave's row loop.
#include <cstdlib>
#include <mutex>

#include "ec_mfma_grad.h"

namespace {

constexpr float EPS32 = 1.1920929e-07f;  // torch.finfo(float32).eps: Categorical clamps probs to [eps, 1-eps]
#ifdef ABL_PPO_STAMPS  // diagnostic build (make abl_ppo): shader-clock time per section of the tile loop, wave 0 of block 0
#define PST(i)                                              \
    do {                                                    \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        pst[i] += now_ - pst_prev;                          \
        pst_prev = now_;                                    \
    } while (0)
#else
#define PST(i)
#endif

struct PpoArgs {
    const float *w1, *b1, *w2, *b2, *w3, *b3;
    const float* obs;        // [R,9]
    const int32_t* avail;    // [R,A] (actor)
    const int64_t* actions;  // [R]   (actor)
    const float* old_logp;   // [R]   (actor; NULL: forward only)
    const float* adv;        // [R]   (actor)
    const float* target;     // [R]   (critic)
    const float* mask;       // [R]
    int64_t R;
    int n_agents, A;
    float eps_clip, entropy_coef, inv_denom;
    float* grads;     // flat [dW1 | db1 | dW2 | db2 | dW3 | db3] (torch parameter order), accumulated into; NULL: no backward
    float* loss;      // [1] accumulated into
    float* out_a;     // actor: log-prob of the taken action [R]; critic: value [R]
    float* out_b;     // actor: entropy [R]
    // ec_ppo_*_grad_ex: row r of the pass is row row_index[r] of the per-row tables (a minibatch of a shuffled epoch);
    // NULL: the identity.  Outputs (out_a / out_b) are indexed by r.
    const int64_t* row_index;
    int flags;        // EC_PPO_LOGITS: Categorical(logits=...) semantics (exact log-softmax, no [eps, 1-eps] clamp)
                      // EC_PPO_NLL: the actor objective is the log-prob of `actions` (behaviour cloning)
    // ec_gail_disc_grad (critic head): the one-hot column of row r is onehot[r] (an action) instead of (row % n_agents),
    // and the head is the GAIL discriminator's BCE (bce = +1: expert rows, -log(p + 1e-8); -1: agent rows,
    // -log(1 - p + 1e-8)); out_a receives p = sigmoid(z)
    const int64_t* onehot;
    int bce;
    // backward passes: block b writes its sums to ws[b * ws_stride + i] (i: index into grads; loss at n_params) and
    // k_ppo_sum adds them to grads / loss in block order -- no atomics, the same bits every time.  NULL: atomics.
    float* ws;
    int ws_stride;
};

// HEAD 0: actor (TO3 output tiles), HEAD 1: critic.  IN1T: tiles of the first layer's dense input (9 + n_agents).
// BWD false: the forward-only passes (old log-probs, old values): the same forward arithmetic without the persistent
// gradient accumulators, so that two blocks fit a CU instead of one.
// NQ (actor): 8-column groups of the last output tile that hold actions (A <= 32 (TO3 - 1) + 8 NQ); the head's per-action
// arithmetic and the k-steps of d2 = W3^T d3 beyond them are compiled out (11 actions: half of the head).
template <int HEAD, int TO3, int IN1T, bool BWD, int NQ = 4>
__global__ void __launch_bounds__(256) k_ppo(PpoArgs p) {
    extern __shared__ __align__(16) float lds[];
    const int nw = blockDim.x >> 6;
    const int N = p.n_agents, A = p.A, in1 = OBS + N;
    constexpr int T3 = HEAD == 0 ? TO3 : 1;
#define LIVE3(t, r) ((t) < T3 - 1 || (r) < 4 * NQ)  // register r of logits tile t can hold an action
    float* s_f1 = lds;                          // W1 obs part: 8 steps x 2
    float* s_f2 = s_f1 + 8 * 2 * 64;            // W2: 32 x 2
    float* s_f2t = s_f2 + 32 * 2 * 64;          // W2^T: 32 x 2
    float* s_f3 = s_f2t + 32 * 2 * 64;          // actor W3: 32 x TO3;   critic: w3 [64]
    float* s_f3t = s_f3 + (HEAD == 0 ? 32 * TO3 * 64 : 64);  // actor W3^T: 16*TO3 x 2
    float* s_b1 = s_f3t + (HEAD == 0 ? 16 * TO3 * 2 * 64 : 0);
    float* s_b2 = s_b1 + H;
    float* s_b3 = s_b2 + H;                     // [32*T3]
    float* s_id = s_b3 + 32 * T3;               // [N][64]
    float* s_wave = s_id + (size_t)N * H;
    stage_frag(s_f1, p.w1, in1, 1, H, OBS, 8, 2, false);
    stage_frag(s_f2, p.w2, H, 1, H, H, 32, 2, true);
    stage_frag(s_f2t, p.w2, 1, H, H, H, 32, 2, true);  // M[out=j][k=i] = W2[i][j]
    if (HEAD == 0) {
        stage_frag(s_f3, p.w3, H, 1, A, H, 32, TO3, true);
        stage_frag(s_f3t, p.w3, 1, H, H, A, 16 * TO3, 2, true);  // M[out=j][k=a] = W3[a][j]
    } else {
        for (int i = threadIdx.x; i < H; i += blockDim.x) s_f3[i] = p.w3[i];
    }
    for (int i = threadIdx.x; i < H; i += blockDim.x) {
        s_b1[i] = p.b1[i];
        s_b2[i] = p.b2[i];
    }
    for (int i = threadIdx.x; i < 32 * T3; i += blockDim.x) s_b3[i] = i < (HEAD == 0 ? A : 1) ? p.b3[i] : 0.0f;
    for (int i = threadIdx.x; i < N * H; i += blockDim.x) {
        const int ag = i / H, o = i - ag * H;
        s_id[i] = p.w1[(size_t)o * in1 + OBS + ag];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int jrow = lane & 31, half = lane >> 5;
    const int availw = HEAD == 0 ? 32 * A : 0;
    float* s_obs = s_wave + (size_t)wave * (32 * OBS + 4 + availw + (BWD ? 2 * TILE : 0));
    int32_t* s_av = reinterpret_cast<int32_t*>(s_obs + 32 * OBS + 4);
    float* Td = s_obs + 32 * OBS + 4 + availw;
    float* Ta = Td + TILE;
    constexpr bool backward = BWD;  // the launcher picks the instantiation by p.grads

    // persistent gradient accumulators of this wave
    f32x16 gW1[2][IN1T], gW2[2][2], gW3[T3][2];
    float gb1[2] = {0.0f, 0.0f}, gb2[2] = {0.0f, 0.0f}, gb3[T3];
    float gb3c = 0.0f, loss_acc = 0.0f;
#pragma unroll
    for (int a = 0; a < 2; a++) {
#pragma unroll
        for (int b = 0; b < IN1T; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) gW1[a][b][r] = 0.0f;
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) gW2[a][b][r] = 0.0f;
    }
#pragma unroll
    for (int a = 0; a < T3; a++) {
        gb3[a] = 0.0f;
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) gW3[a][b][r] = 0.0f;
    }

#ifdef ABL_PPO_STAMPS
    unsigned long long pst[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pst_prev = __builtin_readcyclecounter();
#endif
    // BWD passes fetch their per-row inputs ONE TILE AHEAD into registers (and the minibatch index two tiles ahead): a wave
    // is alone on its SIMD there, so a load issued where its value is needed stalls the whole pipe for a memory round
    // trip -- measured 20 % of a pass (observation tile, then target / advantage / old log-prob / mask / action each
    // behind its own wait).  The forward-only passes run two waves per SIMD and keep their registers: they fetch in place.
    constexpr bool PF = BWD && (HEAD == 1 || (TO3 == 1 && IN1T == 1));  // (the wider actor instantiations have no registers to spare)
    constexpr bool WA = PF;  // weight-gradient operands one step ahead: the same instantiations have the registers for it
    constexpr int AVP = (HEAD == 0 && PF) ? 6 : 0;  // prefetched mask words per lane (32 x A <= 384); the rest in place
    struct Pre {
        float obs[5];  // 32 x 9 floats over 64 lanes
        int32_t av[AVP > 0 ? AVP : 1];
        int64_t irow, trow, rowbase;
        int act, onehot, nvalid;
        float olp, adv, m, tgt;
        bool ok;
    };
    const int64_t n_tiles = (p.R + 31) / 32;
    const bool gathered = (p.flags & EC_PPO_GATHERED) != 0;  // tables already in pass order: the index only names the agent
    const bool indexed = p.row_index && !gathered;
    const auto load_index = [&](int64_t tile) -> int64_t {
        const int64_t row = tile * 32 + jrow;
        return (p.row_index && tile < n_tiles && row < p.R) ? p.row_index[row] : 0;
    };
    const auto fetch = [&](int64_t tile, int64_t index_of_row) -> Pre {
        Pre f;
        const int64_t rowbase = tile * 32, row = rowbase + jrow;
        const bool live = tile < n_tiles;
        f.rowbase = rowbase;
        f.ok = live && row < p.R;
        f.irow = f.ok ? (p.row_index ? index_of_row : row) : 0;
        f.trow = gathered ? (f.ok ? row : 0) : f.irow;  // row of the per-row tables
        f.nvalid = !live ? 0 : (int)(p.R - rowbase < 32 ? p.R - rowbase : 32);
#pragma unroll
        for (int u = 0; u < 5; u++) {
            const int i = lane + 64 * u, r = i / OBS;  // (rows r <= 31: their lanes hold a valid trow)
            const int64_t src = indexed ? (int64_t)__shfl((long long)f.trow, r & 31) : rowbase + r;
            f.obs[u] = (i < 32 * OBS && r < f.nvalid) ? p.obs[src * OBS + (i - r * OBS)] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < AVP; u++) {
            f.av[u] = 0;
            if (HEAD == 0 && p.avail) {
                const int i = lane + 64 * u, r = i < 32 * A ? i / A : 0;
                const int64_t src = indexed ? (int64_t)__shfl((long long)f.trow, r) : rowbase + r;
                if (i < 32 * A && r < f.nvalid) f.av[u] = p.avail[src * A + (i - r * A)];
            }
        }
        f.act = (HEAD == 0 && f.ok) ? (int)p.actions[f.trow] : 0;
        f.onehot = (HEAD == 1 && p.onehot && f.ok) ? (int)p.onehot[f.trow] : 0;
        f.olp = (HEAD == 0 && p.old_logp && f.ok) ? p.old_logp[f.trow] : 0.0f;
        f.adv = (HEAD == 0 && p.adv && f.ok) ? p.adv[f.trow] : 0.0f;
        f.tgt = (HEAD == 1 && p.target && f.ok) ? p.target[f.trow] : 0.0f;
        f.m = f.ok ? (p.mask ? p.mask[f.trow] : 1.0f) : 0.0f;
        return f;
    };
    // The first layer's weight-gradient MFMAs of a tile are issued inside the NEXT tile's head: they depend on nothing
    // there, and the head is the one long stretch of vector arithmetic during which the matrix pipe would idle (a wave
    // alone on its SIMD overlaps the two only where the instruction stream interleaves them).  The exchange tiles hold
    // layer 1's operands until the next tile's first put; they start as zeros, and the last tile's product follows the loop.
    if (BWD) {
        for (int i = lane; i < 2 * TILE; i += 64) Td[i] = 0.0f;
        wave_sync();
    }
    const int64_t tile0 = (int64_t)blockIdx.x * nw + wave, tstride = (int64_t)gridDim.x * nw;
    int64_t idx_next = 0, idx_next2 = 0;
    Pre cur, nxt;
    if (PF) {
        idx_next = load_index(tile0 + tstride);
        cur = fetch(tile0, load_index(tile0));
        nxt = cur;
    }
    for (int64_t tile = tile0; tile < n_tiles; tile += tstride) {
        if (PF) {
            if (tile != tile0) {
                cur = nxt;
                idx_next = idx_next2;
            }
            nxt = fetch(tile + tstride, idx_next);  // in flight while this tile computes
            idx_next2 = load_index(tile + 2 * tstride);
        } else {
            cur = fetch(tile, load_index(tile));
        }
        const int64_t row = tile * 32 + jrow;
        const bool ok = cur.ok;
        const int64_t irow = cur.irow;
        wave_sync();
#pragma unroll
        for (int u = 0; u < 5; u++)
            if (lane + 64 * u < 32 * OBS) s_obs[lane + 64 * u] = cur.obs[u];
        if (HEAD == 0 && p.avail) {
#pragma unroll
            for (int u = 0; u < AVP; u++)
                if (lane + 64 * u < 32 * A) s_av[lane + 64 * u] = cur.av[u];
            for (int i0 = 64 * AVP; i0 < 32 * A; i0 += 64) {  // mask words beyond the prefetched ones
                const int i = i0 + lane, r = i < 32 * A ? i / A : 0;
                const int64_t src = indexed ? (int64_t)__shfl((long long)cur.trow, r) : cur.rowbase + r;
                if (i < 32 * A) s_av[i] = r < cur.nvalid ? p.avail[src * A + (i - r * A)] : 0;
            }
        }
        wave_sync();
        PST(0);
        // one-hot agent id (ppo_agent.py:64-65); N = 0: none.  (32-bit remainder where the row number allows it)
        int agent = (ok && N > 0) ? (irow < 0x7fffffffll ? (int)((unsigned)irow % (unsigned)N) : (int)(irow % N)) : 0;
        if (HEAD == 1 && p.onehot)  // (obs | one-hot action), dgail_agent.py:199-201
            agent = cur.onehot < 0 ? 0 : (cur.onehot >= N ? N - 1 : cur.onehot);
        const float m = cur.m;

        // ------------------------------------------------------------------ forward
        f32x16 h1[2], h2[2];
        acc_init<2>(h1, s_b1, half);
        if (N > 0) acc_add<2>(h1, s_id + agent * H, half);
        acc_init<2>(h2, s_b2, half);
        chain_obs(h1, s_f1, s_obs, lane);
        chain<2, 2>(h2, s_f2, h1, lane, [](int, int, float x) { return relu(x); });
        PST(1);

        f32x16 d2[2];  // d loss / d h2 (pre-activation), accumulator layout
        if (HEAD == 0) {
            f32x16 z[T3];
            acc_init<T3>(z, s_b3, half);
            chain<T3, 2>(z, s_f3, h2, lane, [](int, int, float x) { return relu(x); });
            PST(2);
            if (BWD) wgrad<2, IN1T, 16, WA>(gW1, gb1, Td, Ta, lane);  // the previous tile's dW1 (see above)
            // ---- head: masked softmax, Categorical log-prob / entropy, clipped surrogate (ppo_agent.py:76-83,135-141)
            const int act = cur.act;
            const bool clampp = !(p.flags & EC_PPO_LOGITS);
            float mx = -3.0e38f;
#pragma unroll
            for (int t = 0; t < T3; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    if (!LIVE3(t, r)) continue;
                    const int j = 32 * t + 8 * (r >> 2) + 4 * half + (r & 3);
                    float x = z[t][r];
                    if (j < A) {
                        if (p.avail && s_av[jrow * A + j] == 0) x = -1e10f;
                        mx = fmaxf(mx, x);
                    }
                    z[t][r] = x;
                }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            float sum = 0.0f;
#pragma unroll
            for (int t = 0; t < T3; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    if (!LIVE3(t, r)) continue;
                    const int j = 32 * t + 8 * (r >> 2) + 4 * half + (r & 3);
                    const float e = j < A ? expf(z[t][r] - mx) : 0.0f;
                    z[t][r] = e;
                    sum += e;
                }
            sum += __shfl_xor(sum, 32);
            const float inv = 1.0f / sum;
            float Hent = 0.0f, gbar = 0.0f, lp_a = 0.0f, inr_a = 0.0f;
            f32x16 gk[T3];  // dH/dp_k
#pragma unroll
            for (int t = 0; t < T3; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    if (!LIVE3(t, r)) continue;
                    const int j = 32 * t + 8 * (r >> 2) + 4 * half + (r & 3);
                    const float pj = z[t][r] * inv;
                    // Categorical(probs): log of the clamped probability, zero gradient outside the clamp range;
                    // Categorical(logits): the log-softmax itself (kept finite where a probability underflows)
                    const bool inr = !clampp || (pj >= EPS32 && pj <= 1.0f - EPS32);
                    const float lp = clampp ? logf(fminf(fmaxf(pj, EPS32), 1.0f - EPS32)) : fmaxf(logf(pj), -3.0e38f);
                    z[t][r] = pj;
                    const float g = j < A ? -(lp + (inr ? 1.0f : 0.0f)) : 0.0f;
                    gk[t][r] = g;
                    Hent = j < A ? fmaf(-pj, lp, Hent) : Hent;
                    gbar = fmaf(g, pj, gbar);
                    if (j == act) {
                        lp_a = lp;
                        inr_a = inr ? 1.0f : 0.0f;
                    }
                }
            Hent += __shfl_xor(Hent, 32);
            gbar += __shfl_xor(gbar, 32);
            lp_a += __shfl_xor(lp_a, 32);
            inr_a += __shfl_xor(inr_a, 32);
            if (ok && half == 0) {
                if (p.out_a) p.out_a[row] = lp_a;
                if (p.out_b) p.out_b[row] = Hent;
            }
            const bool nll = (p.flags & EC_PPO_NLL) != 0;
            // EC_PPO_FIRST_EPOCH: the parameters are the ones the rollout's log-probs belong to (epoch 0 of
            // ppo_agent.py:96-141), so old_logp IS this pass's log-prob: ratio = exp(0) = 1, and the pass leaves the
            // log-probs in out_a for the later epochs instead of a forward-only launch producing them first
            const bool first_epoch = (p.flags & EC_PPO_FIRST_EPOCH) != 0;
            if (!p.old_logp && !nll && !first_epoch) continue;  // forward only
            float obj, c1;
            if (nll) {  // F.cross_entropy(logits, actions) = -mean log-prob (dgail_agent.py:255-262)
                obj = lp_a;
                c1 = inr_a;
            } else {
                const float advr = cur.adv;
                const float ratio = expf(lp_a - (first_epoch ? lp_a : cur.olp));
                const float lo = 1.0f - p.eps_clip, hi = 1.0f + p.eps_clip;
                const float surr1 = ratio * advr, surr2 = fminf(fmaxf(ratio, lo), hi) * advr;
                const float inrange = (ratio >= lo && ratio <= hi) ? 1.0f : 0.0f;
                obj = fminf(surr1, surr2) + p.entropy_coef * Hent;
                // torch.min backward: the smaller input takes the gradient, a tie splits it (ratio inside the clip range)
                const float gr = surr1 < surr2 ? advr : (surr1 > surr2 ? advr * inrange : advr * 0.5f * (1.0f + inrange));
                c1 = gr * ratio * inr_a;
            }
            if (half == 0) loss_acc = fmaf(-obj * m, p.inv_denom, loss_acc);
            if (!backward) continue;
            const float scale = -m * p.inv_denom;
#pragma unroll
            for (int t = 0; t < T3; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    if (!LIVE3(t, r)) {
                        z[t][r] = 0.0f;
                        continue;
                    }
                    const int j = 32 * t + 8 * (r >> 2) + 4 * half + (r & 3);
                    const float pj = z[t][r];
                    const float dz = c1 * ((j == act ? 1.0f : 0.0f) - pj) + (nll ? 0.0f : p.entropy_coef * pj * (gk[t][r] - gbar));
                    z[t][r] = j < A ? scale * dz : 0.0f;  // d3
                }
            // dW3 += d3 (x) relu(h2)
            PST(3);
            wave_sync();
            put_tile<T3>(Td, z, lane, [](int, int, float x) { return x; });
            put_tile<2>(Ta, h2, lane, [](int, int, float x) { return relu(x); });
            wave_sync();
            PST(4);
            wgrad<T3, 2, 4, WA>(gW3, gb3, Td, Ta, lane);
            PST(5);
            // d2 = (W3^T d3) .* [h2 > 0]
            acc_zero<2>(d2);
            chain<2, T3, NQ>(d2, s_f3t, z, lane, [](int, int, float x) { return x; });
        } else {
            // ---- critic head: v = w3 . relu(h2) + b3 ; loss = ((v - target) * mask)^2 / denom (ppo_agent.py:124-125)
            if (BWD) wgrad<2, IN1T, 16, WA>(gW1, gb1, Td, Ta, lane);  // the previous tile's dW1 (see above)
            float part = 0.0f;
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const float4 w = *reinterpret_cast<const float4*>(s_f3 + 32 * t + 8 * q + 4 * half);
                    part = fmaf(w.x, relu(h2[t][4 * q + 0]), part);
                    part = fmaf(w.y, relu(h2[t][4 * q + 1]), part);
                    part = fmaf(w.z, relu(h2[t][4 * q + 2]), part);
                    part = fmaf(w.w, relu(h2[t][4 * q + 3]), part);
                }
            const float v = part + __shfl_xor(part, 32) + s_b3[0];
            float dv;
            if (p.bce) {
                // GAIL discriminator (dgail_network.py:37-50, dgail_agent.py:425-431): p = sigmoid(v);
                // loss = -log(p + 1e-8) on expert rows, -log(1 - p + 1e-8) on agent rows
                const float pr = 1.0f / (1.0f + expf(-v));
                if (ok && half == 0 && p.out_a) p.out_a[row] = pr;
                const float arg = p.bce > 0 ? pr + 1e-8f : 1.0f - pr + 1e-8f;
                if (half == 0) loss_acc = fmaf(-logf(arg) * m, p.inv_denom, loss_acc);
                if (!backward) continue;
                dv = (p.bce > 0 ? -1.0f : 1.0f) / arg * (pr * (1.0f - pr)) * m * p.inv_denom;
            } else {
                if (ok && half == 0 && p.out_a) p.out_a[row] = v;
                if (!p.target) continue;  // forward only
                const float err = (v - cur.tgt) * m;
                if (half == 0) loss_acc = fmaf(err * err, p.inv_denom, loss_acc);
                if (!backward) continue;
                dv = 2.0f * err * m * p.inv_denom;
            }
            if (half == 0) gb3c += dv;
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const float4 w = *reinterpret_cast<const float4*>(s_f3 + 32 * t + 8 * q + 4 * half);
                    const float wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const float a2 = h2[t][4 * q + c];
                        gW3[0][t][4 * q + c] = fmaf(dv, relu(a2), gW3[0][t][4 * q + c]);  // per-lane partial of dW3[j]
                        d2[t][4 * q + c] = a2 > 0.0f ? dv * wv[c] : 0.0f;
                    }
                }
        }
        if (HEAD == 0) {
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) d2[t][r] = h2[t][r] > 0.0f ? d2[t][r] : 0.0f;
        }
        // dW2 += d2 (x) relu(h1)
        PST(6);
        wave_sync();
        put_tile<2>(Td, d2, lane, [](int, int, float x) { return x; });
        put_tile<2>(Ta, h1, lane, [](int, int, float x) { return relu(x); });
        wave_sync();
        PST(4);
        wgrad<2, 2, 4, WA>(gW2, gb2, Td, Ta, lane);
        PST(5);
        // d1 = (W2^T d2) .* [h1 > 0]
        f32x16 d1[2];
        acc_zero<2>(d1);
        chain<2, 2>(d1, s_f2t, d2, lane, [](int, int, float x) { return x; });
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) d1[t][r] = h1[t][r] > 0.0f ? d1[t][r] : 0.0f;
        // dW1 += d1 (x) (obs | one-hot agent id)
        PST(7);
        wave_sync();
        put_tile<2>(Td, d1, lane, [](int, int, float x) { return x; });
        for (int in = half; in < 32 * IN1T; in += 2)
            Ta[in * TSTR + jrow] = in < OBS ? s_obs[jrow * OBS + in] : ((ok && N > 0 && in - OBS == agent) ? 1.0f : 0.0f);
        wave_sync();
        PST(4);
    }
    if (BWD) wgrad<2, IN1T, 4, WA>(gW1, gb1, Td, Ta, lane);
#ifdef ABL_PPO_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0 && BWD && n_tiles > 10000)
        printf("k_ppo<head %d> cycles per section, wave 0 of block 0: load %llu | fwd L1+L2 %llu | L3 %llu | head %llu | put %llu | "
               "wgrad %llu | chain d2 %llu | chain d1 %llu\n", HEAD, pst[0], pst[1], pst[2], pst[3], pst[4], pst[5], pst[6], pst[7]);
#endif

    // ------------------------------------------------------------------ flush the partial sums
    if (backward && nw > 1) {  // wave 0 collects the block's sums and is the only one to flush them
        block_sum(gW2, lds, wave, nw, lane);
        block_sum(gW1, lds, wave, nw, lane);
        block_sum(gW3, lds, wave, nw, lane);
        float sc[8] = {gb1[0], gb1[1], gb2[0], gb2[1], gb3[0], T3 > 1 ? gb3[T3 - 1] : 0.0f, gb3c, loss_acc};
        block_sum_scalars(sc, lds, wave, nw, lane);
        gb1[0] = sc[0];
        gb1[1] = sc[1];
        gb2[0] = sc[2];
        gb2[1] = sc[3];
        gb3[0] = sc[4];
        if (T3 > 1) gb3[T3 - 1] = sc[5];
        gb3c = sc[6];
        loss_acc = sc[7];
        if (wave > 0) return;
    }
    const bool to_ws = backward && p.ws != nullptr;
    const int n3 = HEAD == 0 ? A : 1;
    const int n_params = H * in1 + H + H * H + H + n3 * H + n3;
    if (p.loss || to_ws) {  // (atomics: one per wave -- 32 lanes adding to the same word serialise at the memory side)
        float v = loss_acc;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) {
            if (to_ws) p.ws[(size_t)blockIdx.x * p.ws_stride + n_params] = v;
            else if (v != 0.0f) atomicAdd(p.loss, v);
        }
    }
#ifdef ABL_PPO_NOFLUSH  // diagnostic build (make abl_ppo): what the gradient flush costs
    if (backward && p.R < 0) {
#else
    if (backward) {
#endif
        const bool atomic = !to_ws;
        float* g = to_ws ? p.ws + (size_t)blockIdx.x * p.ws_stride : p.grads;
        float* gW1p = g;
        float* gb1p = gW1p + (size_t)H * in1;
        float* gW2p = gb1p + H;
        float* gb2p = gW2p + H * H;
        float* gW3p = gb2p + H;
        float* gb3p = gW3p + (size_t)n3 * H;
        const auto put = [&](float* dst, float v) {
            if (atomic) atomicAdd(dst, v);
            else *dst = v;
        };
        flush<2, IN1T>(gW1p, in1, H, in1, gW1, lane, atomic);
        flush<2, 2>(gW2p, H, H, H, gW2, lane, atomic);
        // bias sums: the two halves of the wave hold the two row parities of the same entries
#pragma unroll
        for (int to = 0; to < 2; to++) {
            const float v1 = gb1[to] + __shfl_xor(gb1[to], 32), v2 = gb2[to] + __shfl_xor(gb2[to], 32);
            if (half == 0) {
                put(gb1p + 32 * to + jrow, v1);
                put(gb2p + 32 * to + jrow, v2);
            }
        }
        if (HEAD == 0) {
            flush<T3, 2>(gW3p, H, A, H, gW3, lane, atomic);
#pragma unroll
            for (int to = 0; to < T3; to++) {
                const float v3 = gb3[to] + __shfl_xor(gb3[to], 32);
                if (half == 0 && 32 * to + jrow < A) put(gb3p + 32 * to + jrow, v3);
            }
        } else {
            // the 32 lanes of a half hold partial sums of the SAME 32 entries of dW3: reduce them in the wave first
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    float v = gW3[0][t][r];
#pragma unroll
                    for (int o = 16; o >= 1; o >>= 1) v += __shfl_xor(v, o);
                    if (jrow == 0) put(gW3p + 32 * t + 8 * (r >> 2) + 4 * half + (r & 3), v);
                }
            float vb = gb3c;
#pragma unroll
            for (int o = 16; o >= 1; o >>= 1) vb += __shfl_xor(vb, o);
            if (lane == 0) put(gb3p, vb);
        }
    }
}

#undef LIVE3

// second launch of a backward pass: grads[i] += sum over blocks of their private sums, in a fixed order (8 segments of
// the block range per entry, then the segments in order); loss likewise
__global__ void __launch_bounds__(256) k_ppo_sum(const float* __restrict__ ws, int ws_stride, int n_blocks, int n_params,
                                                 float* __restrict__ grads, float* __restrict__ loss) {
    __shared__ float s_part[8][32];
    const int pi = threadIdx.x & 31, seg = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + pi;
    const int per = (n_blocks + 7) / 8, b0 = seg * per, b1 = b0 + per < n_blocks ? b0 + per : n_blocks;
    float s = 0.0f;
    if (i <= n_params) {
#pragma unroll 8
        for (int b = b0; b < b1; b++) s += ws[(size_t)b * ws_stride + i];
    }
    s_part[seg][pi] = s;
    __syncthreads();
    if (seg == 0 && i <= n_params) {
        float t = 0.0f;
#pragma unroll
        for (int q = 0; q < 8; q++) t += s_part[q][pi];
        if (i < n_params) grads[i] += t;
        else if (loss) loss[0] += t;
    }
}

constexpr int PPO_WS_STRIDE = 12800;  // >= the largest parameter count (64 x 64 first layer, 64 actions) + 1
constexpr int PPO_WS_BLOCKS = 256;
std::mutex g_ws_mu;
float* g_ws[64] = {nullptr};

// the per-device workspace of the backward passes (allocated at the first one; calls on one device share it, so passes
// that overlap on one GPU from different streams must not: MARL_AMD_PPO_ATOMICS=1 restores the atomic flush)
int ppo_workspace(float** out, hipStream_t stream) {
    static const bool atomics = getenv("MARL_AMD_PPO_ATOMICS") && atoi(getenv("MARL_AMD_PPO_ATOMICS")) != 0;
    *out = nullptr;
    if (atomics) return EC_OK;
    int dev = 0;
    EC_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return EC_OK;
    std::lock_guard<std::mutex> lock(g_ws_mu);
    if (!g_ws[dev]) {
        // a first backward pass under stream capture must not allocate (hipMalloc would break the capture): that pass
        // flushes with atomics, and a later pass outside a capture allocates
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) return EC_OK;
        EC_HIP(hipMalloc(reinterpret_cast<void**>(&g_ws[dev]), sizeof(float) * PPO_WS_STRIDE * PPO_WS_BLOCKS));
    }
    *out = g_ws[dev];
    return EC_OK;
}

constexpr size_t PPO_LDS_MAX = 160 * 1024;

size_t ppo_lds_bytes(int head, int TO3, int N, int A, int nw, bool bwd) {
    size_t f = 8 * 2 * 64 + 2 * (32 * 2 * 64);
    f += head == 0 ? (size_t)32 * TO3 * 64 + (size_t)16 * TO3 * 2 * 64 : 64;
    f += 2 * H + 32 * (head == 0 ? TO3 : 1) + (size_t)N * H;
    const size_t per_wave = 32 * OBS + 4 + (head == 0 ? 32 * A : 0) + (bwd ? 2 * TILE : 0);  // transposition tiles
    return sizeof(float) * (f + nw * per_wave);
}

template <int HEAD, int TO3, int IN1T, int NQ = 4>
int launch_ppo(const PpoArgs& a, hipStream_t stream) {
    if (HEAD == 0 && NQ == 4 && a.A <= 32 * (TO3 - 1) + 16) return launch_ppo<HEAD, TO3, IN1T, HEAD == 0 ? 2 : 4>(a, stream);
    int nw = 4;
    const bool bwd = a.grads != nullptr;
    size_t lds = ppo_lds_bytes(HEAD, TO3, a.n_agents, a.A, nw, bwd);
    for (int w = 3; w >= 2 && lds > PPO_LDS_MAX; w--) {  // fewer waves per block until the per-wave tiles fit
        nw = w;
        lds = ppo_lds_bytes(HEAD, TO3, a.n_agents, a.A, nw, bwd);
    }
    if (lds > PPO_LDS_MAX) return ec::fail(EC_EINVAL, "ec_ppo: %d agents need %zu B of LDS", a.n_agents, lds);
    const int64_t tiles = (a.R + 31) / 32;
    int64_t blocks = (tiles + nw - 1) / nw;
    if (a.grads) {
        EC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ppo<HEAD, TO3, IN1T, true, NQ>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (blocks > 256) blocks = 256;  // one block per CU: the row loop amortises weight staging and the final atomics
        // MARL_AMD_PPO_MIN_TILES (diagnostic): fewer, longer-running blocks.  Measured at DGAIL's 262 k-row minibatch
        // passes: 1, 8 and 16 tiles per wave take the same time, more is slower; small passes (the BC term's 640 rows)
        // want every tile on a wave of its own.
        static const int min_tiles = getenv("MARL_AMD_PPO_MIN_TILES") ? atoi(getenv("MARL_AMD_PPO_MIN_TILES")) : 1;
        const int64_t want = (tiles + (int64_t)nw * min_tiles - 1) / ((int64_t)nw * min_tiles);
        if (blocks > want) blocks = want < 1 ? 1 : want;
        PpoArgs b = a;
        const int n3 = HEAD == 0 ? a.A : 1;
        const int n_params = H * (OBS + a.n_agents) + H + H * H + H + n3 * H + n3;
        if (int rc = ppo_workspace(&b.ws, stream)) return rc;
        b.ws_stride = PPO_WS_STRIDE;
        if (n_params + 1 > PPO_WS_STRIDE || blocks > PPO_WS_BLOCKS) b.ws = nullptr;
        hipLaunchKernelGGL((k_ppo<HEAD, TO3, IN1T, true, NQ>), dim3((unsigned)blocks), dim3(64 * nw), lds, stream, b);
        if (b.ws)
            hipLaunchKernelGGL(k_ppo_sum, dim3((n_params + 1 + 31) / 32), dim3(256), 0, stream, b.ws, b.ws_stride, (int)blocks,
                               n_params, a.grads, a.loss);
    } else {
        EC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ppo<HEAD, TO3, IN1T, false, NQ>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (blocks > 512) blocks = 512;  // two blocks per CU: no transposition tiles in LDS, a third of the registers
        hipLaunchKernelGGL((k_ppo<HEAD, TO3, IN1T, false, NQ>), dim3((unsigned)blocks), dim3(64 * nw), lds, stream, a);
    }
    EC_HIP(hipGetLastError());
    return EC_OK;
}

int check_common(const char* what, const void* const* ptrs, int n, int64_t rows, int hidden, int n_agents) {
    for (int i = 0; i < n; i++)
        if (!ptrs[i]) return ec::fail(EC_EINVAL, "%s: null pointer", what);
    if (rows < 0) return ec::fail(EC_EINVAL, "%s: rows < 0", what);
    if (hidden != H) return ec::fail(EC_EINVAL, "%s: hidden_dim %d unsupported (this build: %d)", what, hidden, H);
    if (n_agents < 0 || OBS + n_agents > 64) return ec::fail(EC_EINVAL, "%s: 9 + n_agents = %d outside [9, 64]", what, OBS + n_agents);
    return EC_OK;
}

}  // namespace

extern "C" {

int ec_ppo_actor_grad_ex(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                         const float* b3, const float* obs, const int32_t* avail, const int64_t* actions,
                         const float* old_logp, const float* adv, const float* mask, int64_t rows, int32_t n_agents,
                         int32_t n_actions, int32_t hidden, float eps_clip, float entropy_coef, float inv_denom,
                         float* grads, float* loss, float* logp, float* entropy, const int64_t* row_index, int32_t flags,
                         void* stream) {
    if (rows == 0) return EC_OK;
    const void* need[] = {w1, b1, w2, b2, w3, b3, obs, actions};
    if (int rc = check_common("ec_ppo_actor_grad", need, 8, rows, hidden, n_agents)) return rc;
    if (n_agents < 1) return ec::fail(EC_EINVAL, "ec_ppo_actor_grad: n_agents %d", n_agents);
    if (n_actions < 1 || n_actions > 64) return ec::fail(EC_EINVAL, "ec_ppo_actor_grad: n_actions %d out of range [1,64]", n_actions);
    if ((old_logp || (flags & EC_PPO_FIRST_EPOCH)) && !adv) return ec::fail(EC_EINVAL, "ec_ppo_actor_grad: old_logp without adv");
    if ((flags & EC_PPO_FIRST_EPOCH) && (old_logp || !logp))
        return ec::fail(EC_EINVAL, "ec_ppo_actor_grad: EC_PPO_FIRST_EPOCH takes no old_logp and needs the logp output");
    if (grads && !old_logp && !(flags & (EC_PPO_NLL | EC_PPO_FIRST_EPOCH)))
        return ec::fail(EC_EINVAL, "ec_ppo_actor_grad: grads without old_logp");
    PpoArgs a{w1, b1, w2, b2, w3, b3, obs, avail, actions, old_logp, adv, nullptr, mask, rows, n_agents, n_actions,
              eps_clip, entropy_coef, inv_denom, grads, loss, logp, entropy, row_index, flags, nullptr, 0, nullptr, 0};
    const int TO3 = n_actions <= 32 ? 1 : 2, IN1T = OBS + n_agents <= 32 ? 1 : 2;
    if (TO3 == 1 && IN1T == 1) return launch_ppo<0, 1, 1>(a, (hipStream_t)stream);
    if (TO3 == 1) return launch_ppo<0, 1, 2>(a, (hipStream_t)stream);
    if (IN1T == 1) return launch_ppo<0, 2, 1>(a, (hipStream_t)stream);
    return launch_ppo<0, 2, 2>(a, (hipStream_t)stream);
}

int ec_ppo_actor_grad(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                      const float* obs, const int32_t* avail, const int64_t* actions, const float* old_logp,
                      const float* adv, const float* mask, int64_t rows, int32_t n_agents, int32_t n_actions,
                      int32_t hidden, float eps_clip, float entropy_coef, float inv_denom, float* grads, float* loss,
                      float* logp, float* entropy, void* stream) {
    if (rows != 0 && (!avail || !mask)) return ec::fail(EC_EINVAL, "ec_ppo_actor_grad: null pointer");
    return ec_ppo_actor_grad_ex(w1, b1, w2, b2, w3, b3, obs, avail, actions, old_logp, adv, mask, rows, n_agents, n_actions,
                                hidden, eps_clip, entropy_coef, inv_denom, grads, loss, logp, entropy, nullptr, 0, stream);
}

int ec_ppo_critic_grad_ex(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                          const float* b3, const float* obs, const float* target, const float* mask, int64_t rows,
                          int32_t n_agents, int32_t hidden, float inv_denom, float* grads, float* loss, float* values,
                          const int64_t* row_index, void* stream) {
    if (rows == 0) return EC_OK;
    const void* need[] = {w1, b1, w2, b2, w3, b3, obs};
    if (int rc = check_common("ec_ppo_critic_grad", need, 7, rows, hidden, n_agents)) return rc;
    if (grads && !target) return ec::fail(EC_EINVAL, "ec_ppo_critic_grad: grads without target");
    PpoArgs a{w1, b1, w2, b2, w3, b3, obs, nullptr, nullptr, nullptr, nullptr, target, mask, rows, n_agents, 1,
              0.0f, 0.0f, inv_denom, grads, loss, values, nullptr, row_index, 0, nullptr, 0, nullptr, 0};
    if (OBS + n_agents <= 32) return launch_ppo<1, 1, 1>(a, (hipStream_t)stream);
    return launch_ppo<1, 1, 2>(a, (hipStream_t)stream);
}

int ec_gail_disc_grad(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                      const float* obs, const int64_t* actions, int64_t rows, int32_t n_actions, int32_t hidden,
                      int32_t expert, float inv_denom, float* grads, float* loss, float* prob, void* stream) {
    if (rows == 0) return EC_OK;
    const void* need[] = {w1, b1, w2, b2, w3, b3, obs, actions};
    if (int rc = check_common("ec_gail_disc_grad", need, 8, rows, hidden, n_actions)) return rc;
    if (n_actions < 1) return ec::fail(EC_EINVAL, "ec_gail_disc_grad: n_actions %d", n_actions);
    // the critic instantiation with the action as the one-hot column and the BCE head
    PpoArgs a{w1, b1, w2, b2, w3, b3, obs, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, rows, n_actions, 1,
              0.0f, 0.0f, inv_denom, grads, loss, prob, nullptr, nullptr, 0, actions, expert ? 1 : -1, nullptr, 0};
    if (OBS + n_actions <= 32) return launch_ppo<1, 1, 1>(a, (hipStream_t)stream);
    return launch_ppo<1, 1, 2>(a, (hipStream_t)stream);
}

int ec_ppo_critic_grad(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                       const float* obs, const float* target, const float* mask, int64_t rows, int32_t n_agents,
                       int32_t hidden, float inv_denom, float* grads, float* loss, float* values, void* stream) {
    if (rows != 0 && (!mask || n_agents < 1)) return ec::fail(EC_EINVAL, "ec_ppo_critic_grad: null mask or n_agents < 1");
    return ec_ppo_critic_grad_ex(w1, b1, w2, b2, w3, b3, obs, target, mask, rows, n_agents, hidden, inv_denom, grads, loss,
                                 values, nullptr, stream);
}

}  // extern "C"
