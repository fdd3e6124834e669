// ec_diff.hip — fused forward + loss + backward of the diffusion discriminator's compute_loss for ONE condition
// (rl/policy_gradient_rl/dgail/dgail_network.py:82-108), gfx950.
//
// Replaces, per compute_loss call of DGAILAgent._update_discriminator (dgail_agent.py:203-243): the denoiser forward, the
// per-row MSE, the R1 penalty |d sum(eps) / d x_t|^2 (an autograd.grad with create_graph=True in the reference) and the
// double backward of all of it.  The caller supplies d loss / d L per row (`coef`, from the BCE of
// D = e^-Lp / (e^-Lp + e^-Ln + 1e-8)); the kernel returns L per row and the gradients of its direct inputs.  Nothing
// per-row is materialised in HBM: a row costs 36 + 8 + 1 + 4 SA + 4 bytes of reads (obs, action, t, noise, coef).
//
//   x_t  = (obs | one-hot action) + t * noise
//   a1   = W1x x_t + bias[t]          bias[t] = b1 + W1t E_time[t] + W1c E_cond[c]   (built by the caller, [2][64])
//   a2   = W2 relu(a1) + b2 ;  eps = W3 relu(a2) + b3
//   L    = mean_d (eps - noise)^2 + lambda * |g|^2,   g = W1x^T (m1 .* W2^T (m2 .* w3sum)),  m = [a > 0], w3sum = colsum W3
// backward with cf = coef[row]:
//   d3 = cf * 2 (eps - noise) / SA ;        dW3 += d3 (x) relu(a2) ;  d2 = m2 .* W3^T d3
//   dg = cf * 2 lambda g ;                  dW1x += u1 (x) dg  (u1 = m1 .* v2) ;  dv2 = m1 .* (W1x dg)
//   dW2 += u2 (x) dv2  (u2 = m2 .* w3sum) ; dw3sum += m2 .* (W2 dv2)
//   dW2 += d2 (x) relu(a1) ;  d1 = m1 .* W2^T d2 ;  dW1x += d1 (x) x_t ;  dbias[t] += d1
// Same fp32 MFMA 32x32x2 scheme as ec_ppo.hip: a wave owns 32 rows per pass, layers are chained through registers, weight
// gradients use the row index as the MFMA's K dimension through per-wave transposition tiles in LDS.
#include "ec_mfma_grad.h"

namespace {

struct DiffArgs {
    const float* w1;      // [H][ld1], the x part is columns [0, SA)
    int ld1;
    const float* bias;    // [2][H]
    const float *w2, *b2; // [H][H], [H]
    const float *w3, *b3; // [SA][H], [SA]
    const float* w3sum;   // [H]
    const float* obs;        // [R][9]
    const int64_t* actions;  // [R]
    const uint8_t* t;        // [R]
    const float* noise;      // [R][SA]
    const float* coef;       // [R] (backward)
    int64_t R;
    int A;
    float lambda;
    float* loss_out;  // [R] or NULL
    // ec_diffusion_grad_bce: the coefficient is the derivative of D_phi's BCE (dgail_agent.py:209-221) with respect to this
    // call's per-row loss L, computed in the kernel from L and the OTHER condition's loss of the same row:
    //   p = exp(-lp) / (exp(-lp) + exp(-ln) + 1e-8);  BCE(p, label) mean over rows;  side 0: this call is lp, 1: ln
    const float* other;  // [R] or NULL (coef is used)
    int side, label;
    float bce_scale;     // 1 / rows (x the data-parallel weight)
    float* grads;     // [dW1x H*SA | dbias(t=0)+dbias(t=1) H | dbias(t=1) H | dW2 H*H | db2 H | dW3 SA*H | db3 SA | dw3sum H]
    // ec_diffusion_reward_from_loss (forward only): this call is the agent-condition loss ln of dgail_agent.py:487-499 and
    // rw_lp [R] the expert-condition loss of the same rows; rw_out[r] = -log(1 - D + 1e-8) * rw_weight + rw_env_coef * rw_env[r]
    // with D = e^-lp / (e^-lp + e^-ln + 1e-8)
    const float* rw_lp;
    const float* rw_env;
    float rw_weight, rw_env_coef;
    float* rw_out;
};

template <int TX, bool BWD>
__global__ void __launch_bounds__(256) k_diff(DiffArgs p) {
    extern __shared__ __align__(16) float lds[];
    const int nw = blockDim.x >> 6;
    const int A = p.A, SA = OBS + A, SAp = SA | 1;
    float* s_f1 = lds;                               // W1x, inputs in accumulator order: 16 TX steps x 2
    float* s_f2 = s_f1 + 16 * TX * 2 * 64;           // W2: 32 x 2
    float* s_f3 = s_f2 + 32 * 2 * 64;                // W3: 32 x TX
    float* s_f2t = s_f3 + 32 * TX * 64;              // W2^T: 32 x 2
    float* s_f1t = s_f2t + 32 * 2 * 64;              // W1x^T: 32 x TX
    float* s_f3t = s_f1t + 32 * TX * 64;             // W3^T: 16 TX x 2 (backward)
    float* s_bias = s_f3t + (BWD ? 16 * TX * 2 * 64 : 0);
    float* s_b2 = s_bias + 2 * H;
    float* s_b3 = s_b2 + H;                          // [32 TX]
    float* s_w3s = s_b3 + 32 * TX;                   // [H]
    float* s_wave = s_w3s + H;
    stage_frag(s_f1, p.w1, p.ld1, 1, H, SA, 16 * TX, 2, true);
    stage_frag(s_f2, p.w2, H, 1, H, H, 32, 2, true);
    stage_frag(s_f3, p.w3, H, 1, SA, H, 32, TX, true);
    stage_frag(s_f2t, p.w2, 1, H, H, H, 32, 2, true);        // M[out=j][k=i] = W2[i][j]
    stage_frag(s_f1t, p.w1, 1, p.ld1, SA, H, 32, TX, true);  // M[out=x][k=j] = W1[j][x]
    if (BWD) stage_frag(s_f3t, p.w3, 1, H, H, SA, 16 * TX, 2, true);  // M[out=j][k=d] = W3[d][j]
    for (int i = threadIdx.x; i < 2 * H; i += blockDim.x) s_bias[i] = p.bias[i];
    for (int i = threadIdx.x; i < H; i += blockDim.x) {
        s_b2[i] = p.b2[i];
        s_w3s[i] = p.w3sum[i];
    }
    for (int i = threadIdx.x; i < 32 * TX; i += blockDim.x) s_b3[i] = i < SA ? p.b3[i] : 0.0f;
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int jrow = lane & 31, half = lane >> 5;
    float* s_x = s_wave + (size_t)wave * (2 * 32 * SAp + (BWD ? 2 * TILE : 0));
    float* s_noise = s_x + 32 * SAp;
    float* Td = s_noise + 32 * SAp;
    float* Ta = Td + TILE;

    // persistent gradient accumulators of this wave
    f32x16 gW1[2][TX], gW2[2][2], gW3[TX][2], gw3s[2];
    float gb1[2] = {0.0f, 0.0f}, gb2[2] = {0.0f, 0.0f}, gb3[TX];
    float gbt[2] = {0.0f, 0.0f};  // dbias(t=1) when the first layer's gradient tile has no spare column (SA == 32 TX)
    if (BWD) {
#pragma unroll
        for (int a = 0; a < 2; a++) {
            acc_zero<TX>(gW1[a]);
            acc_zero<2>(gW2[a]);
        }
#pragma unroll
        for (int a = 0; a < TX; a++) {
            acc_zero<2>(gW3[a]);
            gb3[a] = 0.0f;
        }
        acc_zero<2>(gw3s);
    }
    const auto ident = [](int, int, float x) { return x; };
    const auto relu_f = [](int, int, float x) { return relu(x); };

    const int64_t n_tiles = (p.R + 31) / 32;
    for (int64_t tile = (int64_t)blockIdx.x * nw + wave; tile < n_tiles; tile += (int64_t)gridDim.x * nw) {
        const int64_t rowbase = tile * 32;
        const int64_t row = rowbase + jrow;
        const bool ok = row < p.R;
        int act = ok ? (int)p.actions[row] : 0;
        act = act < 0 ? 0 : (act >= A ? A - 1 : act);  // (obs | one-hot action), dgail_agent.py:199-201
        const int tr = (ok && p.t[row]) ? 1 : 0;
        float cf = (BWD && ok && p.coef) ? p.coef[row] : 0.0f;
        const float other = (BWD && ok && p.other) ? p.other[row] : 0.0f;
        // the wave's x_t and noise tiles [32][SAp] (zero rows beyond R)
        wave_sync();
        for (int i0 = 0; i0 < 32 * SA; i0 += 64) {
            const int i = i0 + lane;
            const int r = i < 32 * SA ? i / SA : 0, k = i - r * SA;
            const int a_r = __shfl(act, r), t_r = __shfl(tr, r);
            if (i < 32 * SA) {
                const bool valid = rowbase + r < p.R;
                const float nz = valid ? p.noise[rowbase * SA + i] : 0.0f;
                const float sa = !valid ? 0.0f : (k < OBS ? p.obs[(rowbase + r) * OBS + k] : (k - OBS == a_r ? 1.0f : 0.0f));
                s_noise[r * SAp + k] = nz;
                s_x[r * SAp + k] = sa + (float)t_r * nz;  // dgail_network.py:92
            }
        }
        wave_sync();

        // ------------------------------------------------------------------ forward
        f32x16 h1[2], h2[2];
        acc_init<2>(h1, s_bias + tr * H, half);
#pragma unroll
        for (int s4 = 0; s4 < 4 * TX; s4++) {
#pragma unroll
            for (int to = 0; to < 2; to++) {
                const float4 a = *reinterpret_cast<const float4*>(s_f1 + ((s4 * 2 + to) * 64 + lane) * 4);
                const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int k = kperm(4 * s4 + c, half);
                    h1[to] = MFMA(av[c], k < SA ? s_x[jrow * SAp + k] : 0.0f, h1[to]);
                }
            }
        }
        unsigned m1 = 0, m2 = 0;
#pragma unroll
        for (int T = 0; T < 2; T++)
#pragma unroll
            for (int r = 0; r < 16; r++) m1 |= h1[T][r] > 0.0f ? (1u << (16 * T + r)) : 0u;
        acc_init<2>(h2, s_b2, half);
        chain<2, 2>(h2, s_f2, h1, lane, relu_f);
#pragma unroll
        for (int T = 0; T < 2; T++)
#pragma unroll
            for (int r = 0; r < 16; r++) m2 |= h2[T][r] > 0.0f ? (1u << (16 * T + r)) : 0u;
        const auto sel1 = [&](int T, int r, float x) { return ((m1 >> (16 * T + r)) & 1u) ? x : 0.0f; };
        const auto sel2 = [&](int T, int r, float x) { return ((m2 >> (16 * T + r)) & 1u) ? x : 0.0f; };

        f32x16 e[TX];  // eps, then d3
        acc_init<TX>(e, s_b3, half);
        chain<TX, 2>(e, s_f3, h2, lane, relu_f);
        float mse = 0.0f;
#pragma unroll
        for (int t = 0; t < TX; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int k = 32 * t + 8 * (r >> 2) + 4 * half + (r & 3);
                const float dlt = k < SA ? e[t][r] - s_noise[jrow * SAp + k] : 0.0f;
                mse = fmaf(dlt, dlt, mse);
                e[t][r] = dlt;  // (scaled by the row's coefficient below, once the row's loss is known)
            }
        mse = (mse + __shfl_xor(mse, 32)) / (float)SA;  // F.mse_loss(reduction="none").mean(dim=1), :97

        // R1: u2 = m2 .* w3sum, v2 = W2^T u2, u1 = m1 .* v2, g = W1x^T u1   (dgail_network.py:99-106)
        f32x16 w3s[2], v2[2], g[TX];
        acc_init<2>(w3s, s_w3s, half);
        acc_zero<2>(v2);
        chain<2, 2>(v2, s_f2t, w3s, lane, sel2);
        acc_zero<TX>(g);
        chain<TX, 2>(g, s_f1t, v2, lane, sel1);
        float r1 = 0.0f;
#pragma unroll
        for (int t = 0; t < TX; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) r1 = fmaf(g[t][r], g[t][r], r1);  // entries >= SA are exact zeros
        r1 += __shfl_xor(r1, 32);
        const float L = mse + r1 * p.lambda;
        if (ok && half == 0 && p.loss_out) p.loss_out[row] = L;
        if (!BWD) {
            if (p.rw_out && ok && half == 0) {
                const float pe = expf(-p.rw_lp[row]), pa = expf(-L);
                const float dphi = pe / (pe + pa + 1e-8f);
                const float base = -logf(1.0f - dphi + 1e-8f);  // dgail_agent.py:497-499
                p.rw_out[row] = base * p.rw_weight + (p.rw_env ? p.rw_env_coef * p.rw_env[row] : 0.0f);
            }
            continue;
        }
        if (p.other) {
            const float lp = p.side == 0 ? L : other, ln = p.side == 0 ? other : L;
            const float ea = expf(-lp), eb = expf(-ln), s = ea + eb + 1e-8f, pr = ea / s;
            // F.binary_cross_entropy backward: (p - y) / max((1 - p) p, 1e-12) / rows;  d p / d lp = -p (1 - p),
            // d p / d ln = p exp(-ln) / s
            const float dLdp = (pr - (float)p.label) / fmaxf((1.0f - pr) * pr, 1e-12f);
            const float dpdl = p.side == 0 ? -pr * (1.0f - pr) : pr * eb / s;
            cf = ok ? dLdp * dpdl * p.bce_scale : 0.0f;
        }
        const float c3 = cf * 2.0f / (float)SA;
#pragma unroll
        for (int t = 0; t < TX; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) e[t][r] *= c3;

        float nob[2] = {0.0f, 0.0f};  // bias sums of the products that have no bias
        // (1) dW3 += d3 (x) relu(a2);  d2 = m2 .* W3^T d3
        wave_sync();
        put_tile<TX>(Td, e, lane, ident);
        put_tile<2>(Ta, h2, lane, relu_f);
        wave_sync();
        wgrad<TX, 2, 4, TX == 1>(gW3, gb3, Td, Ta, lane);
        f32x16 d2[2];
        acc_zero<2>(d2);
        chain<2, TX>(d2, s_f3t, e, lane, ident);
#pragma unroll
        for (int T = 0; T < 2; T++)
#pragma unroll
            for (int r = 0; r < 16; r++) d2[T][r] = sel2(T, r, d2[T][r]);
        // (2) the R1 term: dg = cf * 2 lambda g;  dW1x += u1 (x) dg;  dv2 = m1 .* (W1x dg)
        const float cg = cf * 2.0f * p.lambda;
#pragma unroll
        for (int t = 0; t < TX; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) g[t][r] *= cg;
        wave_sync();
        put_tile<2>(Td, v2, lane, sel1);
        put_tile<TX>(Ta, g, lane, ident);
        wave_sync();
        wgrad<2, TX, 4, TX == 1>(gW1, nob, Td, Ta, lane);
        f32x16 dv2[2];
        acc_zero<2>(dv2);
        chain<2, TX>(dv2, s_f1, g, lane, ident);
#pragma unroll
        for (int T = 0; T < 2; T++)
#pragma unroll
            for (int r = 0; r < 16; r++) dv2[T][r] = sel1(T, r, dv2[T][r]);
        // (3) dW2 += u2 (x) dv2;  dw3sum += m2 .* (W2 dv2)
        wave_sync();
        put_tile<2>(Td, w3s, lane, sel2);
        put_tile<2>(Ta, dv2, lane, ident);
        wave_sync();
        wgrad<2, 2, 4, TX == 1>(gW2, nob, Td, Ta, lane);
        {
            f32x16 du2[2];
            acc_zero<2>(du2);
            chain<2, 2>(du2, s_f2, dv2, lane, ident);
#pragma unroll
            for (int T = 0; T < 2; T++)
#pragma unroll
                for (int r = 0; r < 16; r++) gw3s[T][r] += sel2(T, r, du2[T][r]);
        }
        // (4) dW2 += d2 (x) relu(a1);  d1 = m1 .* W2^T d2
        wave_sync();
        put_tile<2>(Td, d2, lane, ident);
        put_tile<2>(Ta, h1, lane, relu_f);
        wave_sync();
        wgrad<2, 2, 4, TX == 1>(gW2, gb2, Td, Ta, lane);
        f32x16 d1[2];
        acc_zero<2>(d1);
        chain<2, 2>(d1, s_f2t, d2, lane, ident);
#pragma unroll
        for (int T = 0; T < 2; T++)
#pragma unroll
            for (int r = 0; r < 16; r++) d1[T][r] = sel1(T, r, d1[T][r]);
        // (5) dW1x += d1 (x) [x_t | t]: the spare input column SA carries t, so its gradient column is dbias(t=1)
        wave_sync();
        put_tile<2>(Td, d1, lane, ident);
        for (int in = half; in < 32 * TX; in += 2)
            Ta[in * TSTR + jrow] = in < SA ? s_x[jrow * SAp + in] : (in == SA ? (float)tr : 0.0f);
        wave_sync();
        wgrad<2, TX, 4, TX == 1>(gW1, gb1, Td, Ta, lane);
        if (SA == 32 * TX) {  // no spare column: the t = 1 rows' share of the bias sum straight from the d1 tile
            const unsigned tmask = (unsigned)__ballot(tr != 0);  // (both halves hold the same 32 rows)
#pragma unroll 4
            for (int s = 0; s < 16; s++) {
                const int r_ = 2 * s + half;
                if ((tmask >> r_) & 1u) {
                    gbt[0] += Td[jrow * TSTR + r_];
                    gbt[1] += Td[(32 + jrow) * TSTR + r_];
                }
            }
        }
    }

    if (BWD && nw > 1) {  // wave 0 collects the block's sums and is the only one to flush them (see block_sum)
        block_sum(gW2, lds, wave, nw, lane);
        block_sum(gW1, lds, wave, nw, lane);
        block_sum(gW3, lds, wave, nw, lane);
        f32x16 ws[1][2] = {{gw3s[0], gw3s[1]}};
        block_sum(ws, lds, wave, nw, lane);
        gw3s[0] = ws[0][0];
        gw3s[1] = ws[0][1];
        float sc[8] = {gb1[0], gb1[1], gb2[0], gb2[1], gb3[0], TX > 1 ? gb3[TX - 1] : 0.0f, gbt[0], gbt[1]};
        block_sum_scalars(sc, lds, wave, nw, lane);
        gbt[0] = sc[6];
        gbt[1] = sc[7];
        gb1[0] = sc[0];
        gb1[1] = sc[1];
        gb2[0] = sc[2];
        gb2[1] = sc[3];
        gb3[0] = sc[4];
        if (TX > 1) gb3[TX - 1] = sc[5];
        if (wave > 0) return;
    }
    if (BWD) {
        float* gW1p = p.grads;
        float* gbp = gW1p + (size_t)H * SA;
        float* gb1p = gbp + H;
        float* gW2p = gb1p + H;
        float* gb2p = gW2p + H * H;
        float* gW3p = gb2p + H;
        float* gb3p = gW3p + (size_t)SA * H;
        float* gwsp = gb3p + SA;
#pragma unroll
        for (int to = 0; to < 2; to++)
#pragma unroll
            for (int ti = 0; ti < TX; ti++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int out = 32 * to + 8 * (r >> 2) + 4 * half + (r & 3), in = 32 * ti + jrow;
                    if (in < SA) atomicAdd(gW1p + (size_t)out * SA + in, gW1[to][ti][r]);
                    else if (in == SA) atomicAdd(gb1p + out, gW1[to][ti][r]);
                }
        flush<2, 2>(gW2p, H, H, H, gW2, lane);
        flush<TX, 2>(gW3p, H, SA, H, gW3, lane);
#pragma unroll
        for (int to = 0; to < 2; to++) {
            atomicAdd(gbp + 32 * to + jrow, gb1[to]);
            atomicAdd(gb2p + 32 * to + jrow, gb2[to]);
            if (SA == 32 * TX) atomicAdd(gb1p + 32 * to + jrow, gbt[to]);
        }
#pragma unroll
        for (int to = 0; to < TX; to++)
            if (32 * to + jrow < SA) atomicAdd(gb3p + 32 * to + jrow, gb3[to]);
        // the 32 lanes of a half hold per-row partial sums of the same entries of dw3sum
#pragma unroll
        for (int T = 0; T < 2; T++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                float v = gw3s[T][r];
#pragma unroll
                for (int o = 16; o >= 1; o >>= 1) v += __shfl_xor(v, o);
                if (jrow == 0) atomicAdd(gwsp + 32 * T + 8 * (r >> 2) + 4 * half + (r & 3), v);
            }
    }
}

constexpr size_t DIFF_LDS_MAX = 160 * 1024;

size_t diff_lds_bytes(int TX, int SA, int nw, bool bwd) {
    size_t f = (size_t)16 * TX * 2 * 64 + 2 * (32 * 2 * 64) + 2 * ((size_t)32 * TX * 64) + (bwd ? (size_t)16 * TX * 2 * 64 : 0);
    f += 2 * H + H + 32 * TX + H;
    const size_t per_wave = (size_t)2 * 32 * (SA | 1) + (bwd ? 2 * TILE : 0);
    return sizeof(float) * (f + nw * per_wave);
}

template <int TX, bool BWD>
int launch_diff(const DiffArgs& a, hipStream_t stream) {
    const int SA = OBS + a.A;
    int nw = 4;
    size_t lds = diff_lds_bytes(TX, SA, nw, BWD);
    for (int w = 3; w >= 1 && lds > DIFF_LDS_MAX; w--) {  // fewer waves per block until the per-wave tiles fit
        nw = w;
        lds = diff_lds_bytes(TX, SA, nw, BWD);
    }
    if (lds > DIFF_LDS_MAX) return ec::fail(EC_EINVAL, "ec_diffusion_grad: LDS need %zu too large", lds);
    EC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_diff<TX, BWD>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds));
    const int64_t tiles = (a.R + 31) / 32;
    int64_t blocks = (tiles + nw - 1) / nw;
    const int64_t cap = BWD ? 256 : 256 * (int64_t)(DIFF_LDS_MAX / (lds + 1024) >= 2 ? 2 : 1);  // blocks that fit a CU
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL((k_diff<TX, BWD>), dim3((unsigned)blocks), dim3(64 * nw), lds, stream, a);
    EC_HIP(hipGetLastError());
    return EC_OK;
}

}  // namespace

extern "C" {

static int diffusion_grad_impl(const char* what, const float* w1, int32_t ld1, const float* bias_t, const float* w2,
                               const float* b2, const float* w3, const float* b3, const float* w3sum, const float* obs,
                               const int64_t* actions, const uint8_t* t, const float* noise, const float* coef,
                               const float* other, int side, int label, float bce_scale, int64_t rows, int32_t n_actions,
                               int32_t hidden, float r1_lambda, float* loss, float* grads, void* stream,
                               const float* rw_lp = nullptr, const float* rw_env = nullptr, float rw_weight = 0.0f,
                               float rw_env_coef = 0.0f, float* rw_out = nullptr) {
    if (rows == 0) return EC_OK;
    if (rows < 0) return ec::fail(EC_EINVAL, "%s: rows < 0", what);
    if (!w1 || !bias_t || !w2 || !b2 || !w3 || !b3 || !w3sum || !obs || !actions || !t || !noise)
        return ec::fail(EC_EINVAL, "%s: null pointer", what);
    if (hidden != H) return ec::fail(EC_EINVAL, "%s: hidden_dim %d unsupported (this build: %d)", what, hidden, H);
    const int SA = OBS + n_actions;
    if (n_actions < 1 || SA > 64) return ec::fail(EC_EINVAL, "%s: obs+actions = %d exceeds 64", what, SA);
    if (ld1 < SA) return ec::fail(EC_EINVAL, "%s: ld1 %d < obs+actions %d", what, ld1, SA);
    if (grads && !coef && !other) return ec::fail(EC_EINVAL, "%s: grads without a coefficient", what);
    if (!grads && !loss && !rw_out) return ec::fail(EC_EINVAL, "%s: neither loss nor grads requested", what);
    // (the backward borrows input column SA of the first layer's gradient tile for the per-t bias sums; with
    // obs+actions = 32 or 64 there is none and the t = 1 share is summed from the d1 tile instead)
    DiffArgs a{w1, ld1, bias_t, w2, b2, w3, b3, w3sum, obs, actions, t, noise, coef, rows, n_actions, r1_lambda, loss,
               other, side, label, bce_scale, grads, rw_lp, rw_env, rw_weight, rw_env_coef, rw_out};
    const int TX = SA <= 32 ? 1 : 2;
    if (grads) return TX == 1 ? launch_diff<1, true>(a, (hipStream_t)stream) : launch_diff<2, true>(a, (hipStream_t)stream);
    return TX == 1 ? launch_diff<1, false>(a, (hipStream_t)stream) : launch_diff<2, false>(a, (hipStream_t)stream);
}

int ec_diffusion_grad(const float* w1, int32_t ld1, const float* bias_t, const float* w2, const float* b2, const float* w3,
                      const float* b3, const float* w3sum, const float* obs, const int64_t* actions, const uint8_t* t,
                      const float* noise, const float* coef, int64_t rows, int32_t n_actions, int32_t hidden,
                      float r1_lambda, float* loss, float* grads, void* stream) {
    return diffusion_grad_impl("ec_diffusion_grad", w1, ld1, bias_t, w2, b2, w3, b3, w3sum, obs, actions, t, noise, coef, nullptr,
                               0, 0, 0.0f, rows, n_actions, hidden, r1_lambda, loss, grads, stream);
}

int ec_diffusion_grad_bce(const float* w1, int32_t ld1, const float* bias_t, const float* w2, const float* b2,
                          const float* w3, const float* b3, const float* w3sum, const float* obs, const int64_t* actions,
                          const uint8_t* t, const float* noise, const float* other_loss, int32_t side, int32_t label,
                          float scale, int64_t rows, int32_t n_actions, int32_t hidden, float r1_lambda, float* loss,
                          float* grads, void* stream) {
    if (!other_loss || !grads) return ec::fail(EC_EINVAL, "ec_diffusion_grad_bce: other_loss and grads are required");
    if ((side != 0 && side != 1) || (label != 0 && label != 1))
        return ec::fail(EC_EINVAL, "ec_diffusion_grad_bce: side %d / label %d must be 0 or 1", side, label);
    return diffusion_grad_impl("ec_diffusion_grad_bce", w1, ld1, bias_t, w2, b2, w3, b3, w3sum, obs, actions, t, noise, nullptr,
                               other_loss, side, label, scale, rows, n_actions, hidden, r1_lambda, loss, grads, stream);
}

int ec_diffusion_reward_from_loss(const float* w1, int32_t ld1, const float* bias_t, const float* w2, const float* b2,
                                  const float* w3, const float* b3, const float* w3sum, const float* obs,
                                  const int64_t* actions, const uint8_t* t, const float* noise, const float* loss_expert,
                                  const float* env_reward, float weight, float env_coef, int64_t rows, int32_t n_actions,
                                  int32_t hidden, float r1_lambda, float* loss_agent, float* reward, void* stream) {
    if (rows != 0 && (!loss_expert || !reward))
        return ec::fail(EC_EINVAL, "ec_diffusion_reward_from_loss: loss_expert and reward are required");
    return diffusion_grad_impl("ec_diffusion_reward_from_loss", w1, ld1, bias_t, w2, b2, w3, b3, w3sum, obs, actions, t, noise,
                               nullptr, nullptr, 0, 0, 0.0f, rows, n_actions, hidden, r1_lambda, loss_agent, nullptr, stream,
                               loss_expert, env_reward, weight, env_coef, reward);
}

}  // extern "C"
