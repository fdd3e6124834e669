// ec_rollout.hip — the per-step bookkeeping of the on-policy runners as ONE launch.
//
// Replaces, per environment step of ParallelRunner.run (runners/parallel_episode_runner.py:60-104; same pattern in
// gail_runner.py:59-95 and dgail_runner.py:66-91): the six EpisodeBuffer.insert slice writes
// (buffer/episode_buffer.py:21-28) with the team-reward broadcast and mask = 1 - terminated, and the running sums of
// reward / completion time / task counters.  In torch these are ~20 launches of 4-5 us each around a 54 us step
// kernel; here it is one.  The observation written as next_obs[t] is also obs[t+1] / avail[t+1] of the next step, so
// the runner never copies the env's observation tensors.
#include "ec_common.h"

namespace {

struct StoreArgs {
    const float* obs;          // [B,N,O]   observation AFTER the step (or the initial one when t < 0)
    const int32_t* avail;      // [B,N,A]
    const int64_t* actions;    // [B,N]     what the buffer stores for step t
    const float* reward;       // [B]
    const uint8_t* terminated; // [B]
    const int32_t* counters;   // [B,4]
    const float* ctime;        // [B]
    int64_t B;
    int N, O, A;
    int64_t t, T;
    float* b_obs;        // [B,T,N,O]
    int32_t* b_avail;    // [B,T,N,A]
    int64_t* b_actions;  // [B,T,N,1]
    float* b_rewards;    // [B,T,N,1]
    float* b_masks;      // [B,T,N,1]
    float* b_next_obs;   // [B,T,N,O]
    double* totals;      // [6] reward, completion time, finish, success, drop, failure (accumulated)
};

__global__ void __launch_bounds__(256) k_store(StoreArgs a) {
    const int64_t P = a.B * a.N;
    if (blockIdx.x + 1 < gridDim.x) {
        const int64_t lane = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (lane >= P) return;
        const int64_t b = lane / a.N;
        const int n = (int)(lane - b * a.N);
        const float* o = a.obs + lane * a.O;
        const int64_t t = a.t;
        if (t >= 0 && a.b_obs) {
            const int64_t at = (b * a.T + t) * a.N + n;  // (b, t, n)
            float* no = a.b_next_obs + at * a.O;
            for (int k = 0; k < a.O; k++) no[k] = o[k];
            a.b_actions[at] = a.actions[lane];
            a.b_rewards[at] = a.reward[b];                         // team reward, parallel_episode_runner.py:101
            a.b_masks[at] = a.terminated[b] ? 0.0f : 1.0f;         // :102
        }
        if (t + 1 < a.T && a.b_obs) {
            const int64_t at = (b * a.T + t + 1) * a.N + n;  // (b, t+1, n): the observation the next step acts on
            float* oo = a.b_obs + at * a.O;
            for (int k = 0; k < a.O; k++) oo[k] = o[k];
            const int32_t* av = a.avail + lane * a.A;
            int32_t* ao = a.b_avail + at * a.A;
            for (int k = 0; k < a.A; k++) ao[k] = av[k];
        }
        return;
    }
    // last block: episode statistics in a fixed order (deterministic), parallel_episode_runner.py:94-99
    if (a.t < 0 || !a.totals) return;
    __shared__ double s[6][256];
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int64_t b = threadIdx.x; b < a.B; b += blockDim.x) {
        acc[0] += (double)a.reward[b];
        acc[1] += (double)a.ctime[b];
        for (int q = 0; q < 4; q++) acc[2 + q] += (double)a.counters[4 * b + q];
    }
    for (int q = 0; q < 6; q++) s[q][threadIdx.x] = acc[q];
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w)
            for (int q = 0; q < 6; q++) s[q][threadIdx.x] += s[q][threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x < 6) a.totals[threadIdx.x] += s[threadIdx.x][0];
}

}  // namespace

extern "C" int ec_store_transition(const float* obs, const int32_t* avail, const int64_t* actions, const float* reward,
                                   const uint8_t* terminated, const int32_t* counters, const float* completion_time,
                                   int64_t n_envs, int32_t n_agents, int32_t obs_dim, int32_t n_actions, int64_t t,
                                   int64_t episode_limit, float* buf_obs, int32_t* buf_avail, int64_t* buf_actions,
                                   float* buf_rewards, float* buf_masks, float* buf_next_obs, double* totals,
                                   void* stream) {
    if (n_envs <= 0) return EC_OK;
    if (!obs || !avail) return ec::fail(EC_EINVAL, "ec_store_transition: null observation");
    if (t >= episode_limit || t < -1) return ec::fail(EC_EINVAL, "ec_store_transition: t %lld outside [-1, %lld)", (long long)t, (long long)episode_limit);
    if (t >= 0 && (!reward || !terminated || !counters || !completion_time || (buf_obs && !actions)))
        return ec::fail(EC_EINVAL, "ec_store_transition: null step output");
    if (buf_obs && (!buf_avail || !buf_actions || !buf_rewards || !buf_masks || !buf_next_obs))
        return ec::fail(EC_EINVAL, "ec_store_transition: incomplete buffer");
    StoreArgs a{obs, avail, actions, reward, terminated, counters, completion_time, n_envs, n_agents, obs_dim, n_actions,
                t, episode_limit, buf_obs, buf_avail, buf_actions, buf_rewards, buf_masks, buf_next_obs, totals};
    const int64_t P = n_envs * n_agents;
    const unsigned blocks = (unsigned)((P + 255) / 256) + 1;
    hipLaunchKernelGGL(k_store, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    EC_HIP(hipGetLastError());
    return EC_OK;
}
