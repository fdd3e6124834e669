// ec_optim.hip — clip_grad_norm_ + Adam.step() for a handful of small parameter tensors in ONE launch.
//
// The reference ends every optimiser step with torch.nn.utils.clip_grad_norm_(params, max_norm) and Adam.step()
// (dgail_agent.py:319-325, ppo_agent.py:128-141, gail_agent.py:222-231).  On the device torch runs those as ~20 small
// launches per network (foreach norms, stack, norm, clamp, foreach mul; foreach lerp / mul / addcmul / sqrt / div / addcdiv
// and the bias-correction scalars) — for DGAIL's 64 minibatch steps per update more launches than everything else
// together.  A 64-64 MLP has ~6 k parameters: one block does the whole step of one network, and one launch takes up to
// EC_ADAM_MAX_NETS networks.
//   total = sqrt(sum g^2);  c = min(max_norm / (total + 1e-6), 1);  g *= c                       (clip_grad_norm_)
//   t += 1;  m += (g - m) (1 - b1);  v = v b2 + (1 - b2) g g;
//   p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)                                  (Adam, no weight decay)
// State (exp_avg, exp_avg_sq, step) stays in the torch optimiser's own tensors, so checkpoints do not change.
#include <cmath>

#include "ec_common.h"

namespace {

constexpr int ADAM_MAX_TENSORS = 16, ADAM_MAX_NETS = 4;  // (the argument block stays under the 4 KB a launch can carry)

struct AdamNet {
    float* p[ADAM_MAX_TENSORS];
    float* g[ADAM_MAX_TENSORS];
    float* m[ADAM_MAX_TENSORS];
    float* v[ADAM_MAX_TENSORS];
    float* step[ADAM_MAX_TENSORS];  // device scalars (capturable optimisers), or all NULL: step_host is the new step count
    int n[ADAM_MAX_TENSORS];
    int n_tensors;
    float step_host;
    float step_size_host, bc2_sqrt_host;  // host step counts: lr / (1 - b1^t) and sqrt(1 - b2^t) evaluated in double, as torch does
    float lr, b1, b2, omb1, omb2, eps, max_norm;  // omb = 1 - beta rounded from double, as torch passes it; max_norm <= 0: no clipping
};
struct AdamArgs {
    AdamNet net[ADAM_MAX_NETS];
};

__global__ void __launch_bounds__(1024) k_adam_clip(AdamArgs args) {
    const AdamNet& a = args.net[blockIdx.x];
    __shared__ float s_red[16];
    __shared__ float s_coef;
    float ss = 0.0f;
    for (int k = 0; k < a.n_tensors; k++)
        for (int i = threadIdx.x; i < a.n[k]; i += blockDim.x) {
            const float g = a.g[k][i];
            ss = fmaf(g, g, ss);
        }
    for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = ss;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.0f;
        for (int w = 0; w < (int)(blockDim.x >> 6); w++) t += s_red[w];
        const float c = a.max_norm > 0.0f ? a.max_norm / (sqrtf(t) + 1e-6f) : 1.0f;
        s_coef = (c < 1.0f || c != c) ? c : 1.0f;  // (a NaN norm poisons every gradient, as torch's clamp lets it)
    }
    __syncthreads();
    const float coef = s_coef;
    for (int k = 0; k < a.n_tensors; k++) {
        // device step counters (capturable Adam): torch evaluates the bias corrections on the fp32 step tensor too
        float step_size = a.step_size_host, bc2_sqrt = a.bc2_sqrt_host;
        if (a.step[k]) {
            const float t = *a.step[k] + 1.0f;
            step_size = a.lr / (1.0f - powf(a.b1, t));
            bc2_sqrt = sqrtf(1.0f - powf(a.b2, t));
        }
        for (int i = threadIdx.x; i < a.n[k]; i += blockDim.x) {
            const float g = a.g[k][i] * coef;
            float m = a.m[k][i], v = a.v[k][i];
            m = m + (g - m) * a.omb1;
            v = v * a.b2 + a.omb2 * g * g;
            a.g[k][i] = g;
            a.m[k][i] = m;
            a.v[k][i] = v;
            a.p[k][i] -= step_size * m / (sqrtf(v) / bc2_sqrt + a.eps);
        }
    }
    __syncthreads();  // every thread has read the step counts
    if (threadIdx.x < a.n_tensors && a.step[threadIdx.x]) *a.step[threadIdx.x] += 1.0f;
}

}  // namespace

extern "C" int ec_adam_clip_step(int32_t n_nets, const int32_t* n_tensors, float* const* params, float* const* grads,
                                 float* const* exp_avg, float* const* exp_avg_sq, float* const* steps, const int64_t* numel,
                                 const double* step_host, const double* lr, const double* beta1, const double* beta2,
                                 const double* eps, const double* max_norm, void* stream) {
    if (n_nets == 0) return EC_OK;
    if (n_nets < 0 || n_nets > ADAM_MAX_NETS) return ec::fail(EC_EINVAL, "ec_adam_clip_step: %d networks (1..%d)", n_nets, ADAM_MAX_NETS);
    if (!n_tensors || !params || !grads || !exp_avg || !exp_avg_sq || !numel || !lr || !beta1 || !beta2 || !eps || !max_norm)
        return ec::fail(EC_EINVAL, "ec_adam_clip_step: null argument");
    AdamArgs a;
    int at = 0;
    for (int net = 0; net < n_nets; net++) {
        AdamNet& x = a.net[net];
        if (n_tensors[net] < 1 || n_tensors[net] > ADAM_MAX_TENSORS)
            return ec::fail(EC_EINVAL, "ec_adam_clip_step: network %d has %d tensors (1..%d)", net, n_tensors[net], ADAM_MAX_TENSORS);
        x.n_tensors = n_tensors[net];
        bool dev_steps = steps && steps[at];
        for (int k = 0; k < ADAM_MAX_TENSORS; k++) {
            const bool on = k < x.n_tensors;
            x.p[k] = on ? params[at + k] : nullptr;
            x.g[k] = on ? grads[at + k] : nullptr;
            x.m[k] = on ? exp_avg[at + k] : nullptr;
            x.v[k] = on ? exp_avg_sq[at + k] : nullptr;
            x.step[k] = (on && dev_steps) ? steps[at + k] : nullptr;
            x.n[k] = on ? (int)numel[at + k] : 0;
            if (on && (!x.p[k] || !x.g[k] || !x.m[k] || !x.v[k] || numel[at + k] < 0 || numel[at + k] > (1 << 30) ||
                       (dev_steps && !steps[at + k])))
                return ec::fail(EC_EINVAL, "ec_adam_clip_step: network %d tensor %d: null pointer or bad size", net, k);
        }
        if (!dev_steps && (!step_host || !(step_host[net] >= 1.0)))
            return ec::fail(EC_EINVAL, "ec_adam_clip_step: network %d has neither device step counters nor a host step count", net);
        x.step_host = dev_steps ? 0.0f : (float)step_host[net];
        x.step_size_host = x.bc2_sqrt_host = 1.0f;
        if (!dev_steps) {
            x.step_size_host = (float)(lr[net] / (1.0 - pow(beta1[net], step_host[net])));
            x.bc2_sqrt_host = (float)sqrt(1.0 - pow(beta2[net], step_host[net]));
        }
        x.lr = (float)lr[net];
        x.b1 = (float)beta1[net];
        x.b2 = (float)beta2[net];
        x.omb1 = (float)(1.0 - beta1[net]);
        x.omb2 = (float)(1.0 - beta2[net]);
        x.eps = (float)eps[net];
        x.max_norm = (float)max_norm[net];
        at += x.n_tensors;
    }
    hipLaunchKernelGGL(k_adam_clip, dim3((unsigned)n_nets), dim3(1024), 0, (hipStream_t)stream, a);
    EC_HIP(hipGetLastError());
    return EC_OK;
}
