// ec_rollout.hip — the per-step bookkeeping of the on-policy runners as ONE launch.
//
// Replaces, per environment step of ParallelRunner.run (runners/parallel_episode_runner.py:60-104; same pattern in
// gail_runner.py:59-95 and dgail_runner.py:66-91): the six EpisodeBuffer.insert slice writes
// (buffer/episode_buffer.py:21-28) with the team-reward broadcast and mask = 1 - terminated, and the running sums of
// reward / completion time / task counters.  In torch these are ~20 launches of 4-5 us each around a 54 us step
// kernel; here it is one.  The observation written as next_obs[t] is also obs[t+1] / avail[t+1] of the next step, so
// the runner never copies the env's observation tensors.
#include <mutex>

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

// The same bookkeeping for a whole episode that ec_rollout_mlp_policy / ec_rollout_random_policy left as time-major
// histories: one launch files every transition into the [B,T,N,.] buffer and adds the episode statistics.  The
// statistics of step t are reduced by block t in k_store's order and added to the totals in step order by whichever
// block finishes last, so the totals equal those of T ec_store_transition calls bit for bit.
struct EpisodeArgs {
    const float* obs0;          // [B,N,O]     observation before the first step
    const int32_t* avail0;      // [B,N,A]
    const float* h_obs;         // [T,B,N,O]   observation after step t
    const int32_t* h_avail;     // [T,B,N,A]
    const int64_t* h_actions;   // [T,B,N]     action taken at step t
    const float* h_reward;      // [T,B]
    const uint8_t* h_term;      // [T,B]
    const int32_t* h_counters;  // [T,B,4]
    const float* h_ctime;       // [T,B]
    int64_t B;
    int N, O, A;
    int64_t T;
    float* b_obs;
    int32_t* b_avail;
    int64_t* b_actions;
    float* b_rewards;
    float* b_masks;
    float* b_next_obs;
    double* totals;    // [6]
    double* partial;   // [T,6] scratch
    uint32_t* done;    // zero on entry, zero again on exit
    int64_t copy_blocks;
    int vec2;          // the copy part moves 8-byte pieces (even N*O and N*A, 8-byte aligned bases)
};

constexpr int EP_TC = 20;  // steps per copy block of k_store_episode
constexpr int EP_U = 5;    // steps per wave of a 4-wave block in its vector path

__global__ void __launch_bounds__(256) k_store_episode(EpisodeArgs a) {
    if ((int64_t)blockIdx.x < a.copy_blocks) {
        // one env, TC consecutive steps per block: the destination range of each tensor is contiguous over the whole
        // chunk, the source one N*dim-element segment per step
        const int64_t chunks = (a.T + EP_TC - 1) / EP_TC;
        const int64_t b = (int64_t)blockIdx.x / chunks;
        const int64_t t0 = ((int64_t)blockIdx.x - b * chunks) * EP_TC;
        const int nt = (int)(a.T - t0 < EP_TC ? a.T - t0 : EP_TC);
        const int NO = a.N * a.O, NA = a.N * a.A;
        if (a.vec2) {
            // N*O and N*A even, every base 8-byte aligned (the launcher checked): a wave takes a step at a time and copies
            // each of its segments as 8-byte pieces -- no index division per element, half the memory instructions
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
            const int NO2 = NO >> 1, NA2 = NA >> 1;
            if (NO2 <= 64 && NA2 <= 64 && a.N <= 64 && nw * EP_U >= EP_TC) {
                // (the path's shapes: a segment is at most one piece per lane; the wave's up to EP_U steps are loaded
                // before any is stored, so their round trips overlap)
                float2 vo[EP_U], vp[EP_U];
                int2 va[EP_U];
                int64_t ac[EP_U];
                float rw[EP_U], mk[EP_U];
#pragma unroll
                for (int u = 0; u < EP_U; u++) {
                    const int tt = wave + u * nw;
                    vo[u] = vp[u] = make_float2(0.0f, 0.0f);
                    va[u] = make_int2(0, 0);
                    ac[u] = 0;
                    rw[u] = mk[u] = 0.0f;
                    if (tt < nt) {
                        const int64_t t = t0 + tt;
                        if (lane < NO2) {
                            vo[u] = reinterpret_cast<const float2*>(a.h_obs + (t * a.B + b) * NO)[lane];
                            vp[u] = reinterpret_cast<const float2*>(t == 0 ? a.obs0 + b * NO : a.h_obs + ((t - 1) * a.B + b) * NO)[lane];
                        }
                        if (lane < NA2)
                            va[u] = reinterpret_cast<const int2*>(t == 0 ? a.avail0 + b * NA : a.h_avail + ((t - 1) * a.B + b) * NA)[lane];
                        if (lane < a.N) ac[u] = a.h_actions[(t * a.B + b) * a.N + lane];
                        rw[u] = a.h_reward[t * a.B + b];
                        mk[u] = a.h_term[t * a.B + b] ? 0.0f : 1.0f;
                    }
                }
#pragma unroll
                for (int u = 0; u < EP_U; u++) {
                    const int tt = wave + u * nw;
                    if (tt < nt) {
                        const int64_t row = b * a.T + t0 + tt;
                        if (lane < NO2) {
                            reinterpret_cast<float2*>(a.b_next_obs + row * NO)[lane] = vo[u];
                            reinterpret_cast<float2*>(a.b_obs + row * NO)[lane] = vp[u];
                        }
                        if (lane < NA2) reinterpret_cast<int2*>(a.b_avail + row * NA)[lane] = va[u];
                        if (lane < a.N) {
                            a.b_actions[row * a.N + lane] = ac[u];
                            a.b_rewards[row * a.N + lane] = rw[u];  // team reward, parallel_episode_runner.py:101
                            a.b_masks[row * a.N + lane] = mk[u];    // :102
                        }
                    }
                }
                return;
            }
            for (int tt = wave; tt < nt; tt += nw) {
                const int64_t t = t0 + tt, row = b * a.T + t;
                const float2* so = reinterpret_cast<const float2*>(a.h_obs + (t * a.B + b) * NO);
                const float2* sp = reinterpret_cast<const float2*>(t == 0 ? a.obs0 + b * NO : a.h_obs + ((t - 1) * a.B + b) * NO);
                const int2* sa = reinterpret_cast<const int2*>(t == 0 ? a.avail0 + b * NA : a.h_avail + ((t - 1) * a.B + b) * NA);
                float2* dn = reinterpret_cast<float2*>(a.b_next_obs + row * NO);
                float2* dp = reinterpret_cast<float2*>(a.b_obs + row * NO);
                int2* da = reinterpret_cast<int2*>(a.b_avail + row * NA);
                for (int k = lane; k < NO2; k += 64) {
                    dn[k] = so[k];
                    dp[k] = sp[k];
                }
                for (int k = lane; k < NA2; k += 64) da[k] = sa[k];
                const float rw = a.h_reward[t * a.B + b];                       // team reward, parallel_episode_runner.py:101
                const float mk = a.h_term[t * a.B + b] ? 0.0f : 1.0f;           // :102
                for (int n = lane; n < a.N; n += 64) {
                    a.b_actions[row * a.N + n] = a.h_actions[(t * a.B + b) * a.N + n];
                    a.b_rewards[row * a.N + n] = rw;
                    a.b_masks[row * a.N + n] = mk;
                }
            }
            return;
        }
        for (int i = threadIdx.x; i < nt * NO; i += blockDim.x) {
            const int tt = i / NO, k = i - tt * NO;
            const int64_t t = t0 + tt, dst = (b * a.T + t) * NO + k;
            a.b_next_obs[dst] = a.h_obs[(t * a.B + b) * NO + k];
            a.b_obs[dst] = t == 0 ? a.obs0[b * NO + k] : a.h_obs[((t - 1) * a.B + b) * NO + k];
        }
        for (int i = threadIdx.x; i < nt * NA; i += blockDim.x) {
            const int tt = i / NA, k = i - tt * NA;
            const int64_t t = t0 + tt;
            a.b_avail[(b * a.T + t) * NA + k] = t == 0 ? a.avail0[b * NA + k] : a.h_avail[((t - 1) * a.B + b) * NA + k];
        }
        for (int i = threadIdx.x; i < nt * a.N; i += blockDim.x) {
            const int tt = i / a.N, n = i - tt * a.N;
            const int64_t t = t0 + tt, dst = (b * a.T + t) * a.N + n;
            a.b_actions[dst] = a.h_actions[(t * a.B + b) * a.N + n];
            a.b_rewards[dst] = a.h_reward[t * a.B + b];                  // team reward, parallel_episode_runner.py:101
            a.b_masks[dst] = a.h_term[t * a.B + b] ? 0.0f : 1.0f;        // :102
        }
        return;
    }
    if (!a.totals) return;
    const int64_t t = (int64_t)blockIdx.x - a.copy_blocks;
    __shared__ double s[6][256];
    __shared__ bool last;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int64_t b = threadIdx.x; b < a.B; b += blockDim.x) {
        acc[0] += (double)a.h_reward[t * a.B + b];
        acc[1] += (double)a.h_ctime[t * a.B + b];
        for (int q = 0; q < 4; q++) acc[2 + q] += (double)a.h_counters[4 * (t * a.B + b) + q];
    }
    for (int q = 0; q < 6; q++) s[q][threadIdx.x] = acc[q];
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w)
            for (int q = 0; q < 6; q++) s[q][threadIdx.x] += s[q][threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x < 6) a.partial[t * 6 + threadIdx.x] = s[threadIdx.x][0];
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(a.done, 1u) == (uint32_t)(a.T - 1);
    __syncthreads();
    if (!last) return;
    __threadfence();
    if (threadIdx.x < 6) {
        double tot = a.totals[threadIdx.x];
        for (int64_t i = 0; i < a.T; i++)
            tot += __hip_atomic_load(a.partial + i * 6 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.totals[threadIdx.x] = tot;
    }
    if (threadIdx.x == 0) *a.done = 0;
}

}  // namespace

namespace {

int store_episode_launch(const char* who, const float* obs0, const int32_t* avail0, const float* hist_obs,
                         const int32_t* hist_avail, const int64_t* hist_actions, const float* hist_reward,
                         const uint8_t* hist_terminated, const int32_t* hist_counters, const float* hist_completion_time,
                         int64_t n_envs, int32_t n_agents, int32_t obs_dim, int32_t n_actions, int64_t episode_limit,
                         float* buf_obs, int32_t* buf_avail, int64_t* buf_actions, float* buf_rewards, float* buf_masks,
                         float* buf_next_obs, double* totals, double* partial, uint32_t* done, hipStream_t stream) {
    if (!hist_reward || !hist_terminated || !hist_counters || !hist_completion_time)
        return ec::fail(EC_EINVAL, "%s: null step history", who);
    if (buf_obs && (!buf_avail || !buf_actions || !buf_rewards || !buf_masks || !buf_next_obs || !obs0 || !avail0 ||
                    !hist_obs || !hist_avail || !hist_actions))
        return ec::fail(EC_EINVAL, "%s: incomplete buffer or observation history", who);
    if (totals && (!partial || !done)) return ec::fail(EC_EINVAL, "%s: totals need the partial / done scratch", who);
    const int64_t copy_blocks = buf_obs ? n_envs * ((episode_limit + EP_TC - 1) / EP_TC) : 0;
    const int64_t blocks = copy_blocks + (totals ? episode_limit : 0);
    if (blocks == 0) return EC_OK;
    if (blocks > 0x7fffffffll) return ec::fail(EC_EINVAL, "%s: %lld blocks", who, (long long)blocks);
    const auto al8 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; };
    const int vec2 = buf_obs && (n_agents * obs_dim) % 2 == 0 && (n_agents * n_actions) % 2 == 0 && al8(obs0) && al8(avail0) &&
                     al8(hist_obs) && al8(hist_avail) && al8(buf_obs) && al8(buf_avail) && al8(buf_next_obs);
    EpisodeArgs a{obs0, avail0, hist_obs, hist_avail, hist_actions, hist_reward, hist_terminated, hist_counters,
                  hist_completion_time, n_envs, n_agents, obs_dim, n_actions, episode_limit, buf_obs, buf_avail,
                  buf_actions, buf_rewards, buf_masks, buf_next_obs, totals, partial, done, copy_blocks, vec2};
    hipLaunchKernelGGL(k_store_episode, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    EC_HIP(hipGetLastError());
    return EC_OK;
}

}  // namespace

extern "C" int ec_store_episode(const float* obs0, const int32_t* avail0, const float* hist_obs, const int32_t* hist_avail,
                                const int64_t* hist_actions, const float* hist_reward, const uint8_t* hist_terminated,
                                const int32_t* hist_counters, const float* hist_completion_time, int64_t n_envs,
                                int32_t n_agents, int32_t obs_dim, int32_t n_actions, int64_t episode_limit,
                                float* buf_obs, int32_t* buf_avail, int64_t* buf_actions, float* buf_rewards,
                                float* buf_masks, float* buf_next_obs, double* totals, double* partial, uint32_t* done,
                                void* stream) {
    if (n_envs <= 0 || episode_limit <= 0) return EC_OK;
    return store_episode_launch("ec_store_episode", obs0, avail0, hist_obs, hist_avail, hist_actions, hist_reward,
                                hist_terminated, hist_counters, hist_completion_time, n_envs, n_agents, obs_dim,
                                n_actions, episode_limit, buf_obs, buf_avail, buf_actions, buf_rewards, buf_masks,
                                buf_next_obs, totals, partial, done, (hipStream_t)stream);
}

// ---- ec_store_episode_static: the same launch for callers that cannot allocate ------------------------------------
// The per-step partial sums and the arrival counter of k_store_episode live in a file-scope device array instead of a
// caller-provided workspace.  Layout: bytes [0, 256) = the arrival counter `done` (first word; the rest of the line
// is padding so that the counter has a 256-byte block to itself), then EC_STORE_EPISODE_STATIC_MAX_T rows of 6 f64
// partial sums.  Every GPU the library runs on holds its own copy (197 KB) until the process ends.  In host code the
// array's NAME is only a handle for hipGetSymbolAddress: its device address is looked up once per device and kept in
// g_scratch_ptr; kernels, memsets and copies take that pointer, never the name.
constexpr int64_t EP_STATIC_HEAD = 256;  // bytes in front of the partial sums
static __device__ __align__(256) unsigned char scratch[EP_STATIC_HEAD + EC_STORE_EPISODE_STATIC_MAX_T * 6 * sizeof(double)];

namespace {

constexpr int EP_MAX_DEVICES = 64;
void* g_scratch_ptr[EP_MAX_DEVICES] = {};  // per device: address of that GPU's copy of `scratch`
std::mutex g_scratch_mutex;

// the current device's copy of `scratch` (looked up on the first call on that device, with that device current)
int scratch_of_current_device(unsigned char** out) {
    int dev = -1;
    EC_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= EP_MAX_DEVICES)
        return ec::fail(EC_EINVAL, "ec_store_episode_static: device %d outside [0, %d)", dev, EP_MAX_DEVICES);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        EC_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return ec::fail(EC_EHIP, "ec_store_episode_static: hipGetSymbolAddress returned a null address");
        g_scratch_ptr[dev] = p;
    }
    *out = (unsigned char*)g_scratch_ptr[dev];
    return EC_OK;
}

}  // namespace

extern "C" int ec_store_episode_static(const float* obs0, const int32_t* avail0, const float* hist_obs,
                                       const int32_t* hist_avail, const int64_t* hist_actions, const float* hist_reward,
                                       const uint8_t* hist_terminated, const int32_t* hist_counters,
                                       const float* hist_completion_time, int64_t n_envs, int32_t n_agents,
                                       int32_t obs_dim, int32_t n_actions, int64_t episode_limit, float* buf_obs,
                                       int32_t* buf_avail, int64_t* buf_actions, float* buf_rewards, float* buf_masks,
                                       float* buf_next_obs, double* totals, void* stream) {
    if (n_envs <= 0 || episode_limit <= 0) return EC_OK;
    if (episode_limit > EC_STORE_EPISODE_STATIC_MAX_T)
        return ec::fail(EC_EINVAL, "ec_store_episode_static: episode_limit %lld exceeds the static scratch's %d steps "
                        "(use ec_store_episode with a caller-provided workspace)", (long long)episode_limit,
                        (int)EC_STORE_EPISODE_STATIC_MAX_T);
    unsigned char* base = nullptr;
    if (int rc = scratch_of_current_device(&base)) return rc;
    // the array keeps the last call's contents: the arrival counter must start at zero (k_store_episode leaves it at
    // zero, but a launch that was aborted half-way would not), the partial sums are all rewritten before they are read
    if (totals) EC_HIP(hipMemsetAsync(base, 0, EP_STATIC_HEAD, (hipStream_t)stream));
    return store_episode_launch("ec_store_episode_static", obs0, avail0, hist_obs, hist_avail, hist_actions, hist_reward,
                                hist_terminated, hist_counters, hist_completion_time, n_envs, n_agents, obs_dim,
                                n_actions, episode_limit, buf_obs, buf_avail, buf_actions, buf_rewards, buf_masks,
                                buf_next_obs, totals, (double*)(base + EP_STATIC_HEAD), (uint32_t*)base,
                                (hipStream_t)stream);
}

// test hook: the address (and size) of the current device's copy of the static scratch, optionally filled with a byte first
// (stream-ordered): lets tests/test_zz_no_alloc.py start a call from a poisoned arrival counter and stale partial sums
extern "C" int ec_debug_store_episode_static_scratch(int32_t fill_byte, void** ptr_out, int64_t* bytes_out, void* stream) {
    unsigned char* base = nullptr;
    if (int rc = scratch_of_current_device(&base)) return rc;
    if (fill_byte >= 0) EC_HIP(hipMemsetAsync(base, fill_byte & 0xff, sizeof(scratch), (hipStream_t)stream));
    if (ptr_out) *ptr_out = base;
    if (bytes_out) *bytes_out = (int64_t)sizeof(scratch);
    return EC_OK;
}

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

// ------------------------------------------------------------------------------------------ expert mixing
// Per step of the expert-guided runners (gail_runner.py:72-98, dgail_runner.py:46-94): the per-env coin decides whether an
// env executes the expert's or the agent's action, and expert-driven envs append (obs, expert action) to the agent's
// FIFO of expert transitions (gail_agent.py:77-83).  In torch: compare, where, cumsum, sum, index arithmetic and two
// index_copy_ per step (~17 launches); here two.
namespace {

struct MixArgs {
    const float* coin;    // [B] uniform draws
    const float* ratio;   // device scalar: expert iff coin < ratio
    const int64_t* expert_actions;  // [B,N]
    const int64_t* agent_actions;   // [B,N]
    const float* obs;     // [B,N,O]
    int64_t B;
    int N, O;
    int64_t capacity;
    int64_t* exec_actions;  // [B,N]
    float* ring_obs;        // [>= capacity, N, O]
    int64_t* ring_actions;  // [>= capacity, N]
    int64_t* written;       // device scalar: transitions ever appended
    int64_t* n_expert;      // device scalar (optional): expert-driven env steps so far
};

constexpr int MIX_EPB = 16;  // envs per block

__device__ __forceinline__ int64_t block_sum(int64_t v, int64_t* s_red) {
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    int64_t t = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); w++) t += s_red[w];
    return t;
}

// Envs [b0, b0 + MIX_EPB) of the block: the rank of a selected env among all selected ones is the count of selected
// envs before it, which every block counts for itself from the B coins (L2-resident).  The FIFO keeps the newest
// `capacity` of one call's rows, as appending them one by one would (ExpertTransitions.append).
__global__ void __launch_bounds__(256) k_expert_mix(MixArgs a) {
    __shared__ int64_t s_red[4];
    __shared__ int64_t s_slot[MIX_EPB];
    const float ratio = *a.ratio;
    const int64_t b0 = (int64_t)blockIdx.x * MIX_EPB;
    int64_t before = 0, total = 0;
    for (int64_t b = threadIdx.x; b < a.B; b += blockDim.x) {
        const int64_t u = a.coin[b] < ratio ? 1 : 0;
        total += u;
        before += b < b0 ? u : 0;
    }
    total = block_sum(total, s_red);
    before = block_sum(before, s_red);
    const int64_t written = *a.written;
    __shared__ int s_use[MIX_EPB];
    if (threadIdx.x < MIX_EPB) {  // (the block's own coins in one round trip, not one per iteration of the scan below)
        const int64_t b = b0 + threadIdx.x;
        s_use[threadIdx.x] = (b < a.B && a.coin[b] < ratio) ? 1 : 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t rank = before;
        for (int e = 0; e < MIX_EPB; e++) {
            const bool use = s_use[e] != 0;
            s_slot[e] = (use && rank >= total - a.capacity) ? (written + rank) % a.capacity : (use ? -2 : -1);
            rank += use ? 1 : 0;
        }
    }
    __syncthreads();
    const int64_t nb = a.B - b0 < MIX_EPB ? a.B - b0 : MIX_EPB;
    for (int i = threadIdx.x; i < (int)nb * a.N; i += blockDim.x) {
        const int e = i / a.N, n = i - e * a.N;
        const int64_t src = (b0 + e) * a.N + n, slot = s_slot[e];
        const int64_t ex = a.expert_actions[src];
        a.exec_actions[src] = slot != -1 ? ex : a.agent_actions[src];
        if (slot >= 0) a.ring_actions[slot * a.N + n] = ex;
    }
    const int row = a.N * a.O;
    for (int i = threadIdx.x; i < (int)nb * row; i += blockDim.x) {
        const int e = i / row, k = i - e * row;
        const int64_t slot = s_slot[e];
        if (slot >= 0) a.ring_obs[slot * row + k] = a.obs[(b0 + e) * row + k];
    }
}

// after every block of k_expert_mix has read `written`: advance the counters
__global__ void __launch_bounds__(256) k_expert_mix_count(MixArgs a) {
    __shared__ int64_t s_red[4];
    const float ratio = *a.ratio;
    int64_t total = 0;
    for (int64_t b = threadIdx.x; b < a.B; b += blockDim.x) total += a.coin[b] < ratio ? 1 : 0;
    total = block_sum(total, s_red);
    if (threadIdx.x == 0) {
        *a.written += total;
        if (a.n_expert) *a.n_expert += total;
    }
}

}  // namespace

extern "C" int ec_expert_mix(const float* coin, const float* ratio, const int64_t* expert_actions, const int64_t* agent_actions,
                             const float* obs, int64_t n_envs, int32_t n_agents, int32_t obs_dim, int64_t capacity,
                             int64_t* exec_actions, float* ring_obs, int64_t* ring_actions, int64_t* written,
                             int64_t* n_expert, void* stream) {
    if (n_envs == 0) return EC_OK;
    if (!coin || !ratio || !expert_actions || !agent_actions || !obs || !exec_actions || !ring_obs || !ring_actions || !written)
        return ec::fail(EC_EINVAL, "ec_expert_mix: null pointer");
    if (n_envs < 0 || n_agents < 1 || obs_dim < 1 || capacity < 1)
        return ec::fail(EC_EINVAL, "ec_expert_mix: n_envs %lld, n_agents %d, obs_dim %d, capacity %lld", (long long)n_envs,
                        n_agents, obs_dim, (long long)capacity);
    MixArgs a{coin, ratio, expert_actions, agent_actions, obs, n_envs, n_agents, obs_dim, capacity, exec_actions, ring_obs,
              ring_actions, written, n_expert};
    const int64_t blocks = (n_envs + MIX_EPB - 1) / MIX_EPB;
    hipLaunchKernelGGL(k_expert_mix, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(k_expert_mix_count, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
    EC_HIP(hipGetLastError());
    return EC_OK;
}

// The same FIFO append for a whole episode of ec_rollout_mlp_expert_policy: step t's expert-driven envs (coin[t][b] <
// ratios[t]) append (the observation step t acted on, the expert's decision) in (t, b) order.  Appending step by step keeps,
// per slot, the newest entry that maps to it; so does writing only the entries among the newest `capacity` of the whole
// episode, which is what happens here, in three launches: per-(step, 256-env chunk) counts, the copy, the counters.
namespace {

constexpr int EPI_EPB = 256;

struct EpisodeAppendArgs {
    const float* coin;           // [T][B]
    const float* ratios;         // [T]
    const int64_t* expert_hist;  // [T][B,N]
    const float* obs0;           // [B,N,O] observation step 0 acted on
    const float* obs_hist;       // [T][B,N,O] observation after step t
    int T;
    int64_t B;
    int N, O;
    int64_t capacity;
    float* ring_obs;
    int64_t* ring_actions;
    int64_t* written;
    int64_t* n_expert;  // optional
    int32_t* counts;    // [T][chunks] workspace
    int chunks;
};

__global__ void __launch_bounds__(EPI_EPB) k_expert_episode_count(EpisodeAppendArgs a) {
    __shared__ int64_t s_red[EPI_EPB / 64];
    const int t = blockIdx.y;
    const int64_t b = (int64_t)blockIdx.x * EPI_EPB + threadIdx.x;
    const int64_t use = (b < a.B && a.coin[(int64_t)t * a.B + b] < a.ratios[t]) ? 1 : 0;
    const int64_t total = block_sum(use, s_red);
    if (threadIdx.x == 0) a.counts[t * a.chunks + blockIdx.x] = (int32_t)total;
}

__global__ void __launch_bounds__(EPI_EPB) k_expert_episode_append(EpisodeAppendArgs a) {
    __shared__ int64_t s_red[EPI_EPB / 64];
    __shared__ int64_t s_slot[EPI_EPB];
    __shared__ int s_wave[EPI_EPB / 64];
    const int t = blockIdx.y, mine = t * a.chunks + (int)blockIdx.x, n_all = a.T * a.chunks;
    int64_t before = 0, total = 0;
    for (int i = threadIdx.x; i < n_all; i += blockDim.x) {
        const int64_t c = a.counts[i];
        total += c;
        before += i < mine ? c : 0;
    }
    total = block_sum(total, s_red);
    before = block_sum(before, s_red);
    const int64_t own = a.counts[mine];
    if (own == 0 || before + own <= total - a.capacity) return;  // (block-uniform) nothing selected, or all of it superseded
    const int64_t written = *a.written;
    const int64_t b0 = (int64_t)blockIdx.x * EPI_EPB, b = b0 + threadIdx.x;
    const bool use = b < a.B && a.coin[(int64_t)t * a.B + b] < a.ratios[t];
    const unsigned long long bal = __ballot(use);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_wave[wave] = __popcll(bal);
    __syncthreads();
    int64_t rank = before + __popcll(bal & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; w++) rank += s_wave[w];
    s_slot[threadIdx.x] = (use && rank >= total - a.capacity) ? (written + rank) % a.capacity : -1;
    __syncthreads();
    const int nb = (int)(a.B - b0 < EPI_EPB ? a.B - b0 : EPI_EPB);
    const int64_t* ex = a.expert_hist + ((int64_t)t * a.B + b0) * a.N;
    for (int i = threadIdx.x; i < nb * a.N; i += blockDim.x) {
        const int e = i / a.N, n = i - e * a.N;
        const int64_t slot = s_slot[e];
        if (slot >= 0) a.ring_actions[slot * a.N + n] = ex[i];
    }
    const int row = a.N * a.O;
    const float* ob = (t == 0 ? a.obs0 : a.obs_hist + (int64_t)(t - 1) * a.B * row) + b0 * row;
    for (int i = threadIdx.x; i < nb * row; i += blockDim.x) {
        const int e = i / row, k = i - e * row;
        const int64_t slot = s_slot[e];
        if (slot >= 0) a.ring_obs[slot * row + k] = ob[i];
    }
}

// after every block of k_expert_episode_append has read `written`
__global__ void __launch_bounds__(256) k_expert_episode_commit(EpisodeAppendArgs a) {
    __shared__ int64_t s_red[4];
    int64_t total = 0;
    for (int i = threadIdx.x; i < a.T * a.chunks; i += blockDim.x) total += a.counts[i];
    total = block_sum(total, s_red);
    if (threadIdx.x == 0) {
        *a.written += total;
        if (a.n_expert) *a.n_expert += total;
    }
}

}  // namespace

extern "C" int64_t ec_expert_append_episode_workspace(int32_t n_steps, int64_t n_envs) {
    return n_steps < 0 || n_envs < 0 ? 0 : (int64_t)n_steps * ((n_envs + EPI_EPB - 1) / EPI_EPB) * (int64_t)sizeof(int32_t);
}

extern "C" int ec_expert_append_episode(const float* coin, const float* ratios, const int64_t* expert_actions, const float* obs0,
                                        const float* obs_hist, int32_t n_steps, int64_t n_envs, int32_t n_agents,
                                        int32_t obs_dim, int64_t capacity, float* ring_obs, int64_t* ring_actions,
                                        int64_t* written, int64_t* n_expert, void* workspace, void* stream) {
    if (n_envs == 0 || n_steps == 0) return EC_OK;
    if (!coin || !ratios || !expert_actions || !obs0 || !ring_obs || !ring_actions || !written || !workspace ||
        (n_steps > 1 && !obs_hist))
        return ec::fail(EC_EINVAL, "ec_expert_append_episode: null pointer");
    const int64_t chunks = (n_envs + EPI_EPB - 1) / EPI_EPB;
    if (n_envs < 0 || n_steps < 0 || n_steps > 65535 || n_agents < 1 || obs_dim < 1 || capacity < 1 ||
        chunks * n_steps > (int64_t)1 << 24)
        return ec::fail(EC_EINVAL, "ec_expert_append_episode: n_steps %d, n_envs %lld, n_agents %d, obs_dim %d, capacity %lld",
                        n_steps, (long long)n_envs, n_agents, obs_dim, (long long)capacity);
    EpisodeAppendArgs a{coin, ratios, expert_actions, obs0, obs_hist, n_steps, n_envs, n_agents, obs_dim, capacity, ring_obs,
                        ring_actions, written, n_expert, static_cast<int32_t*>(workspace), (int)chunks};
    const dim3 grid((unsigned)chunks, (unsigned)n_steps);
    hipLaunchKernelGGL(k_expert_episode_count, grid, dim3(EPI_EPB), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(k_expert_episode_append, grid, dim3(EPI_EPB), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(k_expert_episode_commit, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
    EC_HIP(hipGetLastError());
    return EC_OK;
}
