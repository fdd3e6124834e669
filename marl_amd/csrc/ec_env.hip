// ec_env.hip — batched edge_computing environment for gfx950 (MI355X).
//
// Replaces, for B environments at once, the reference's
//   EdgeComputingEnv.step / get_obs / get_avail_actions / reset   envs/edge_computing/multihop/edge_computing_env.py:32-148
//   EdgeNode.offload_task / execute_task / receive_task / generate_task / get_observation / get_avail_actions
//                                                                  envs/edge_computing/multihop/edge_node.py:109-333
//   Task.__init__                                                  envs/edge_computing/multihop/task.py:6-33
//   np.random.RandomState streams                                  edge_node.py:28-29
//
// Formulation: one lane per (env, node).  The reference walks every queue every
// mini-slot and adds 0.05 to Python floats; here every timer is an integer tick count, a timer's fp64
// value is acc[n] (acc[n] = acc[n-1] + mini_time_slot, the same rounding sequence the reference
// produces), and waiting is accounted lazily from enqueue ticks, so a mini-slot touches memory only
// when something happens to a task.  Deadline / completion thresholds are turned into integer tick
// thresholds once per task by evaluating the reference's left-to-right fp64 sum on the LUT.
//
// Layout: every per-node array is [slot][P] with P = B*N lanes minor, so that a wave's access to
// "slot s of my queue" is one contiguous run; task records are 32-byte (two dwordx4) elements.
// MT19937 state is per lane [P][624] and advanced incrementally (one word regenerated per word drawn).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ec_common.h"

namespace {

typedef uint32_t u32;
typedef uint64_t u64;
typedef uint16_t u16;

constexpr int MT_N = 624;
constexpr int MT_M = 397;
constexpr u32 NEVER = 0xFFFFFFFFu;
constexpr u32 T16_NEVER = 0xFFFFu;

// task record while in buffer_queue / new_task / a receiving queue (32 B)
struct __attribute__((aligned(16))) Rec {
    double size;    // task.py:11
    double cycles;  // task.py:12
    u16 twt;        // transmission_waiting_time ticks (as of enqueue when in a receiving queue)
    u16 ctt;        // current_transmission_time ticks (never reset between hops, edge_node.py:227,316)
    u16 hop;
    u16 ntx;        // ticks needed on the current link: min n with acc[n] >= size / rate
    int32_t bwt;    // buffer_waiting_time ticks; inside buffer_queue: bias = bwt - n_off at entry
    u32 enq;        // receiving queue: index of the first receive_task call after the push
};

// task record while in execution_queue (waiting, 32 B)
struct __attribute__((aligned(16))) WRec {
    double et;    // execute_time = cycles / cpu_capacity (edge_node.py:220)
    u32 enq;      // index of the first execute_task call after the push
    u32 drop_at;  // execute_task call index at which the deadline sweep drops it (edge_node.py:299-309)
    u16 nexec;    // min n with acc[n] >= et
    u16 twt, ctt, bwt;
    u16 hop;
    u16 pad[3];
};

static_assert(sizeof(Rec) == 32 && sizeof(WRec) == 32, "record size");

struct EcDev {
    // geometry
    int N, A, k, D, S, T, H;
    int capB, capR, capW, maxTicks;
    int64_t B, P;
    // scalars
    double dt, deadline;
    double size_lo, size_range, cx_lo, cx_range;
    double pen_drop, pen_fail, rew_succ, pen_hop;
    double obs_size_norm, obs_cyc_norm;
    // constant tables
    const double* acc;        // [maxTicks+1]
    const double* cap;        // [N]
    const double* thr_exec;   // [N] exp(-efr*dt)
    const double* qlen;       // [N]
    const double* qthr;       // [N] qlen * rl_queue_coeff
    const float* obs_const;   // [N][3]
    const double* bin_qn;     // [N]
    const double* bin_px2;    // [N]
    const uint8_t* bin_flip;  // [N] p > 0.5
    const int* deg;           // [N] in-degree
    const int* in_src;        // [N][D] ascending source ids
    const double* thr_link;   // [N][D] exp(-tfr[src][dst]*dt)
    const int* slot_of;       // [N(src)][N(dst)] slot of src in dst's in-list, -1 if not adjacent
    const double* rate;       // [N][N]
    const uint8_t* adj;       // [N][A]
    // per-lane state
    u32* mt;       // [2][P][624]
    u32* rng_pos;  // [2][P]
    u64* rng_cnt;  // [2][P]
    u32* has_new;  // [P]
    Rec* newt;     // [P]
    u32* buf_head; // [P]
    u32* buf_len;  // [P]
    Rec* buf;      // [capB][P]
    u32* wq_head;  // [P]
    u32* wq_len;   // [P]
    u32* wq_min;   // [P] lower bound of min drop_at over the waiting queue
    WRec* wq;      // [capW][P]
    u32* ex_n;     // [P]
    u64* exA;      // [k][P] {cet, drop_cet, nexec, hop}
    u64* exB;      // [k][P] {ewt, twt, ctt, bwt}
    u32* rq_head;  // [D][P]
    u32* rq_len;   // [D][P]
    u32* rq_arrive;// [D][P] receive_task call index at which the head arrives
    u32* rq_hstart;// [D][P] first call index with the current head as head
    Rec* rq;       // [D][capR][P]
    u32* n_off;    // [P] offload events (lazy buffer_waiting_time)
    // per-env state
    u32* cur_step; // [B]
    u32* tick;     // [B] execute/receive call index
    int32_t* hop_hist;  // [B][H]
    u32* status;   // [1]
};

// ------------------------------------------------------------------------------------------ helpers
__device__ __forceinline__ double dadd(double a, double b) { return __dadd_rn(a, b); }
__device__ __forceinline__ double dmul(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ double ddiv(double a, double b) { return __ddiv_rn(a, b); }

__device__ __forceinline__ void flag(const EcDev& d, u32 bit) { atomicOr(d.status, bit); }

__device__ __forceinline__ double lut(const EcDev& d, u32 n) {
    if (n > (u32)d.maxTicks) {
        flag(d, EC_ST_TICK_OVERFLOW);
        n = d.maxTicks;
    }
    return d.acc[n];
}

template <typename R>
__device__ __forceinline__ R load_rec(const R* p) {
    union {
        R r;
        uint4 v[2];
    } u;
    const uint4* q = reinterpret_cast<const uint4*>(p);
    u.v[0] = q[0];
    u.v[1] = q[1];
    return u.r;
}
template <typename R>
__device__ __forceinline__ void store_rec(R* p, const R& r) {
    union {
        R r;
        uint4 v[2];
    } u;
    u.r = r;
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = u.v[0];
    q[1] = u.v[1];
}

// Incremental MT19937: position p is regenerated at the moment it is consumed, which yields the same
// word sequence as the block twist of numpy's mt19937_gen (positions are regenerated in the same order
// from the same neighbours).  Only 53-bit doubles are drawn, so pos stays even.
struct Rng {
    u32* mt;
    u32 pos;
    u32 used;
};
__device__ __forceinline__ u32 mt_temper(u32 y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}
__device__ __forceinline__ u32 mt_next(u32 cur, u32 nxt, u32 far) {
    u32 y = (cur & 0x80000000u) | (nxt & 0x7fffffffu);
    return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __forceinline__ double rng_double(Rng& r) {
    const u32 p = r.pos;  // even, <= 622
    u32 p2 = p + 2;
    if (p2 == MT_N) p2 = 0;
    u32 q0 = p + MT_M;
    if (q0 >= MT_N) q0 -= MT_N;
    u32 q1 = q0 + 1;
    if (q1 == MT_N) q1 = 0;
    const uint2 a01 = *reinterpret_cast<const uint2*>(r.mt + p);
    const u32 a2 = r.mt[p2];
    const u32 b0 = r.mt[q0];
    const u32 b1 = r.mt[q1];
    const u32 n0 = mt_next(a01.x, a01.y, b0);
    const u32 n1 = mt_next(a01.y, a2, b1);
    *reinterpret_cast<uint2*>(r.mt + p) = make_uint2(n0, n1);
    r.pos = p2;
    r.used += 2;
    const u32 a = mt_temper(n0) >> 5, b = mt_temper(n1) >> 6;
    // (a * 2^26 + b) / 2^53, exact in fp64 (numpy random_sample)
    return (double)(((u64)a << 26) | (u64)b) * (1.0 / 9007199254740992.0);
}

// smallest n >= lo with acc[n] >= x, T16_NEVER if none in the LUT
__device__ __forceinline__ u32 ticks_to_reach(const EcDev& d, double x) {
    double est = x / d.dt;
    u32 n = 0;
    if (est > 2.0) n = est >= (double)d.maxTicks ? (u32)d.maxTicks : (u32)est - 2u;
    while (n <= (u32)d.maxTicks && d.acc[n] < x) n++;
    return n > (u32)d.maxTicks ? T16_NEVER : n;
}

// The reference's task_time (edge_node.py:255,301): ((((ewt + cet) + twt) + ctt) + bwt) in fp64.
__device__ __forceinline__ double task_time(const EcDev& d, u32 ewt, u32 cet, u32 twt, u32 ctt, u32 bwt) {
    return dadd(dadd(dadd(dadd(lut(d, ewt), lut(d, cet)), lut(d, twt)), lut(d, ctt)), lut(d, bwt));
}

// smallest n >= 1 such that task_time with the free timer at n ticks exceeds the deadline.
// which = 0: free timer is ewt (cet = 0);  which = 1: free timer is cet (ewt fixed).
__device__ __forceinline__ u32 deadline_ticks(const EcDev& d, int which, u32 ewt, u32 twt, u32 ctt, u32 bwt) {
    const u32 others = twt + ctt + bwt + (which ? ewt : 0u);
    const long nominal = (long)(d.deadline / d.dt) - (long)others - 1;
    u32 n = nominal < 1 ? 1u : (u32)nominal;
    while (n <= (u32)d.maxTicks) {
        const double tt = which ? task_time(d, ewt, n, twt, ctt, bwt) : task_time(d, n, 0, twt, ctt, bwt);
        if (tt > d.deadline) return n;
        n++;
    }
    return T16_NEVER;
}

__device__ __forceinline__ u64 pack4(u32 a, u32 b, u32 c, u32 e) {
    return (u64)(a & 0xffffu) | ((u64)(b & 0xffffu) << 16) | ((u64)(c & 0xffffu) << 32) | ((u64)(e & 0xffffu) << 48);
}
__device__ __forceinline__ u32 f16(u64 w, int i) { return (u32)((w >> (16 * i)) & 0xffffu); }

__device__ __forceinline__ u32 clamp16(const EcDev& d, u32 v) {
    if (v > 0xfffeu) {
        flag(d, EC_ST_TICK_OVERFLOW);
        v = 0xfffeu;
    }
    return v;
}

// append to this lane's buffer_queue (edge_node.py:163,333); rec.bwt holds the true tick count
__device__ __forceinline__ void buffer_push(const EcDev& d, int64_t lane, u32& head, u32& len, u32 n_off, Rec rec) {
    if (len >= (u32)d.capB) {
        flag(d, EC_ST_BUFFER_OVERFLOW);
        return;
    }
    rec.bwt = rec.bwt - (int32_t)n_off;
    u32 pos = head + len;
    if (pos >= (u32)d.capB) pos -= d.capB;
    store_rec(d.buf + (size_t)pos * d.P + lane, rec);
    len++;
}

// edge_node.py:159-166 + task.py:6-33
__device__ __forceinline__ void generate_task(const EcDev& d, int64_t lane, int node, Rng& trng, u32& buf_head,
                                              u32& buf_len, u32 n_off, u32& has_new, Rec& newt) {
    has_new = 0;
    // numpy legacy binomial(1, p): inversion with bound = 1 (see oracle/ec_oracle.c binomial_inversion)
    const double qn = d.bin_qn[node], px2 = d.bin_px2[node];
    u32 X;
    for (;;) {
        double U = rng_double(trng);
        if (!(U > qn)) {
            X = 0;
            break;
        }
        U = U - qn;
        if (!(U > px2)) {
            X = 1;
            break;
        }
        // X would exceed the bound: redraw (probability ~2^-53)
    }
    const u32 arrival = d.bin_flip[node] ? 1u - X : X;
    bool direct = false;
    if (arrival) {
        (void)rng_double(trng);  // task.id
        Rec t;
        t.size = dmul(dmul(dadd(d.size_lo, dmul(d.size_range, rng_double(trng))), 1024.0), 8.0);
        t.cycles = dmul(dadd(d.cx_lo, dmul(d.cx_range, rng_double(trng))), t.size);
        t.twt = t.ctt = t.hop = t.ntx = 0;
        t.bwt = 0;
        t.enq = 0;
        if (buf_len == 0) {
            newt = t;  // push then pop of an empty queue
            direct = true;
        } else {
            buffer_push(d, lane, buf_head, buf_len, n_off, t);
        }
    }
    if (direct) {
        has_new = 1;
    } else if (buf_len > 0) {
        newt = load_rec(d.buf + (size_t)buf_head * d.P + lane);
        newt.bwt = newt.bwt + (int32_t)n_off;
        buf_head = buf_head + 1 == (u32)d.capB ? 0u : buf_head + 1;
        buf_len--;
        has_new = 1;
    }
}

struct LaneId {
    int64_t env, lane;
    int node, env_local;
    bool valid;
};
__device__ __forceinline__ LaneId lane_id(const EcDev& d) {
    LaneId L;
    const int epb = blockDim.x / d.N;
    L.env_local = threadIdx.x / d.N;
    L.node = threadIdx.x - L.env_local * d.N;
    L.env = (int64_t)blockIdx.x * epb + L.env_local;
    L.valid = L.env_local < epb && L.env < d.B;
    L.lane = L.env * d.N + L.node;
    return L;
}

// ------------------------------------------------------------------------------------------ kernels
__global__ void k_seed(EcDev d, u32 task_seed0, u32 fail_seed0) {
    const int64_t lane = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= d.P) return;
    const int64_t env = lane / d.N;
    const int node = (int)(lane - env * d.N);
    for (int s = 0; s < 2; s++) {
        u32 seed = (s == 0 ? task_seed0 : fail_seed0) + (u32)env + (u32)node;
        u32* mt = d.mt + ((size_t)s * d.P + lane) * MT_N;
        for (int i = 0; i < MT_N; i++) {
            mt[i] = seed;
            seed = 1812433253u * (seed ^ (seed >> 30)) + (u32)(i + 1);
        }
        d.rng_pos[(size_t)s * d.P + lane] = 0;
        d.rng_cnt[(size_t)s * d.P + lane] = 0;
    }
    d.has_new[lane] = 0;
    d.buf_head[lane] = d.buf_len[lane] = 0;
    d.wq_head[lane] = d.wq_len[lane] = 0;
    d.wq_min[lane] = NEVER;
    d.ex_n[lane] = 0;
    d.n_off[lane] = 0;
    for (int s = 0; s < d.D; s++) {
        d.rq_head[(size_t)s * d.P + lane] = 0;
        d.rq_len[(size_t)s * d.P + lane] = 0;
        d.rq_arrive[(size_t)s * d.P + lane] = NEVER;
        d.rq_hstart[(size_t)s * d.P + lane] = 0;
    }
    if (node == 0) {
        d.cur_step[env] = 0;
        d.tick[env] = 0;
        for (int h = 0; h < d.H; h++) d.hop_hist[env * d.H + h] = 0;
    }
}

// the constructor's generate_task pass (edge_computing_env.py:17-18)
__global__ void k_initial_generate(EcDev d) {
    const int64_t lane = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= d.P) return;
    const int node = (int)(lane % d.N);
    Rng trng{d.mt + (size_t)lane * MT_N, d.rng_pos[lane], 0};
    u32 bh = d.buf_head[lane], bl = d.buf_len[lane], hn;
    Rec nt;
    generate_task(d, lane, node, trng, bh, bl, d.n_off[lane], hn, nt);
    d.rng_pos[lane] = trng.pos;
    d.rng_cnt[lane] += trng.used;
    d.buf_head[lane] = bh;
    d.buf_len[lane] = bl;
    d.has_new[lane] = hn;
    if (hn) store_rec(d.newt + lane, nt);
}

// edge_computing_env.py:145-148 / edge_node.py:171-177
__global__ void k_reset(EcDev d) {
    const int64_t lane = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= d.P) return;
    d.has_new[lane] = 0;
    d.buf_head[lane] = d.buf_len[lane] = 0;
    d.wq_head[lane] = d.wq_len[lane] = 0;
    d.wq_min[lane] = NEVER;
    d.ex_n[lane] = 0;
    for (int s = 0; s < d.D; s++) {
        d.rq_head[(size_t)s * d.P + lane] = 0;
        d.rq_len[(size_t)s * d.P + lane] = 0;
        d.rq_arrive[(size_t)s * d.P + lane] = NEVER;
    }
    if (lane % d.N == 0) {
        d.cur_step[lane / d.N] = 0;
        d.tick[lane / d.N] = 0;
    }
}

// Shared tail: observation + availability mask of the current state for the envs of this block
// (edge_node.py:109-136,179-197).  s_wql/s_hasnew/s_any are [blockDim] LDS scratch, s_obs [blockDim*9].
__device__ void observe_block(const EcDev& d, const LaneId& L, u32 wq_head, u32 wq_len, u32 has_new, const Rec& newt,
                              float* __restrict__ obs, int32_t* __restrict__ avail, u32* s_wql, u32* s_hasnew,
                              u32* s_any, float* s_obs) {
    const int N = d.N, A = d.A;
    const int tid = threadIdx.x;
    const int epb = blockDim.x / N;
    if (L.valid) {
        s_wql[tid] = ((double)wq_len >= d.qthr[L.node]) ? 1u : 0u;  // "full" flag, edge_node.py:186-188
        s_hasnew[tid] = has_new;
        if (obs) {
            float* o = s_obs + tid * 9;
            o[0] = d.obs_const[L.node * 3 + 0];
            o[1] = d.obs_const[L.node * 3 + 1];
            o[2] = d.obs_const[L.node * 3 + 2];
            o[3] = (float)ddiv((double)wq_len, d.qlen[L.node]);
            double w = 0.0;  // get_waiting_time, edge_node.py:99-104 (cet == 0 for waiting tasks)
            u32 pos = wq_head;
            for (u32 j = 0; j < wq_len; j++) {
                w = dadd(w, d.wq[(size_t)pos * d.P + L.lane].et);
                pos = pos + 1 == (u32)d.capW ? 0u : pos + 1;
            }
            o[4] = (float)ddiv(w, d.deadline);
            if (has_new) {
                o[5] = (float)ddiv(newt.size, d.obs_size_norm);
                o[6] = (float)ddiv(newt.cycles, d.obs_cyc_norm);
                o[7] = (float)ddiv(d.deadline, d.deadline);
                o[8] = (float)newt.hop;
            } else {
                o[5] = o[6] = o[7] = -1.0f;
                o[8] = 0.0f;
            }
        }
    }
    __syncthreads();
    if (avail) {
        if (L.valid) {
            u32 any = 0;
            const u32* full = s_wql + L.env_local * N;
            for (int j = 0; j < N; j++) any |= (d.adj[L.node * A + j] && !full[j]) ? 1u : 0u;
            s_any[tid] = any;
        }
        __syncthreads();
        const int64_t env0 = (int64_t)blockIdx.x * epb;
        const int64_t nenv = (d.B - env0) < epb ? (d.B - env0) : epb;
        const int total = (int)nenv * N * A;
        int32_t* out = avail + env0 * N * A;
        for (int idx = tid; idx < total; idx += blockDim.x) {
            const int t = idx / A, j = idx - t * A;  // t = env_local*N + node
            const int el = t / N, i = t - el * N;
            int v;
            if (s_hasnew[t]) {
                v = (j < N) ? ((d.adj[i * A + j] && !s_wql[el * N + j]) || (!s_any[t] && j == i)) : 0;
            } else {
                v = (j == N);
            }
            out[idx] = v;
        }
    }
    if (obs) {
        const int64_t env0 = (int64_t)blockIdx.x * epb;
        const int64_t nenv = (d.B - env0) < epb ? (d.B - env0) : epb;
        const int total = (int)nenv * N * 9;
        float* out = obs + env0 * N * 9;
        for (int idx = tid; idx < total; idx += blockDim.x) out[idx] = s_obs[idx];
    }
}

__global__ void k_observe(EcDev d, float* __restrict__ obs, int32_t* __restrict__ avail) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int NT = blockDim.x;
    u32* s_wql = reinterpret_cast<u32*>(smem);
    u32* s_hasnew = s_wql + NT;
    u32* s_any = s_hasnew + NT;
    float* s_obs = reinterpret_cast<float*>(s_any + NT);
    const LaneId L = lane_id(d);
    u32 wq_head = 0, wq_len = 0, has_new = 0;
    Rec newt;
    newt.size = newt.cycles = 0;
    newt.hop = 0;
    if (L.valid) {
        wq_head = d.wq_head[L.lane];
        wq_len = d.wq_len[L.lane];
        has_new = d.has_new[L.lane];
        if (has_new) newt = load_rec(d.newt + L.lane);
    }
    observe_block(d, L, wq_head, wq_len, has_new, newt, obs, avail, s_wql, s_hasnew, s_any, s_obs);
}

// EdgeComputingEnv.step (edge_computing_env.py:84-143) for the envs of this block.
__global__ void k_step(EcDev d, const int64_t* __restrict__ actions, float* __restrict__ reward,
                       uint8_t* __restrict__ terminated, int32_t* __restrict__ counters,
                       float* __restrict__ completion_time, float* __restrict__ next_obs,
                       int32_t* __restrict__ next_avail) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int NT = blockDim.x;
    const int tid = threadIdx.x;
    // LDS carve-up (all [slot][thread])
    u64* s_exA = reinterpret_cast<u64*>(smem);            // [k][NT]
    u64* s_exB = s_exA + (size_t)d.k * NT;                 // [k][NT]
    double* s_rew = reinterpret_cast<double*>(s_exB + (size_t)d.k * NT);  // [NT]
    double* s_ct = s_rew + NT;                             // [NT]
    u32* s_rqlen = reinterpret_cast<u32*>(s_ct + NT);      // [D][NT]
    u32* s_rqarr = s_rqlen + (size_t)d.D * NT;             // [D][NT]
    u32* s_cnt = s_rqarr + (size_t)d.D * NT;               // [4][NT]
    u32* s_wql = s_cnt + 4 * NT;                           // [NT]
    u32* s_hasnew = s_wql + NT;
    u32* s_any = s_hasnew + NT;
    float* s_obs = reinterpret_cast<float*>(s_any + NT);   // [NT*9]

    const LaneId L = lane_id(d);
    const int N = d.N;
    const int64_t P = d.P;
    const int64_t lane = L.lane;
    const int node = L.node;

    u32 c0 = 0;  // index of the first execute/receive call of this step
    u32 has_new = 0, n_off = 0;
    Rec newt;
    newt.size = newt.cycles = 0;
    newt.twt = newt.ctt = newt.hop = newt.ntx = 0;
    newt.bwt = 0;
    newt.enq = 0;

    // ---------------------------------------------------------------- phase A: offload_task, edge_node.py:199-234
    if (L.valid) {
        c0 = d.tick[L.env];
        has_new = d.has_new[lane];
        n_off = d.n_off[lane];
        const int64_t action = actions[lane];
        if (has_new && action != N) {
            newt = load_rec(d.newt + lane);
            if (action == node) {
                // local: execution_queue.append(task)
                n_off++;
                WRec w;
                w.et = ddiv(newt.cycles, d.cap[node]);
                w.enq = c0;
                w.nexec = (u16)ticks_to_reach(d, w.et);
                w.twt = newt.twt;
                w.ctt = newt.ctt;
                w.bwt = (u16)clamp16(d, (u32)newt.bwt);
                w.hop = newt.hop;
                w.pad[0] = w.pad[1] = w.pad[2] = 0;
                const u32 nd = deadline_ticks(d, 0, 0, w.twt, w.ctt, w.bwt);
                w.drop_at = nd == T16_NEVER ? NEVER : c0 + nd - 1;
                u32 wl = d.wq_len[lane], wh = d.wq_head[lane];
                if (wl >= (u32)d.capW) {
                    flag(d, EC_ST_WAIT_OVERFLOW);
                } else {
                    u32 pos = wh + wl;
                    if (pos >= (u32)d.capW) pos -= d.capW;
                    store_rec(d.wq + (size_t)pos * P + lane, w);
                    d.wq_len[lane] = wl + 1;
                    const u32 m = d.wq_min[lane];
                    if (wl == 0 || w.drop_at < m) d.wq_min[lane] = w.drop_at;
                }
            } else {
                const int slot = (action >= 0 && action < N) ? d.slot_of[node * N + (int)action] : -1;
                if (slot < 0) {
                    flag(d, EC_ST_BAD_ACTION);  // reference: ZeroDivisionError at edge_node.py:227
                } else {
                    n_off++;
                    const int64_t dst = L.env * N + action;
                    Rec r = newt;
                    const double tt = ddiv(r.size, d.rate[node * N + (int)action]);
                    r.ntx = (u16)ticks_to_reach(d, tt);
                    r.hop = (u16)(r.hop + 1);
                    r.enq = c0;
                    const size_t so = (size_t)slot * P + dst;
                    const u32 ql = d.rq_len[so], qh = d.rq_head[so];
                    if (ql >= (u32)d.capR) {
                        flag(d, EC_ST_RECV_OVERFLOW);
                    } else {
                        u32 pos = qh + ql;
                        if (pos >= (u32)d.capR) pos -= d.capR;
                        store_rec(d.rq + ((size_t)slot * d.capR + pos) * P + dst, r);
                        d.rq_len[so] = ql + 1;
                        if (ql == 0) {
                            d.rq_hstart[so] = c0;
                            const u32 need = r.ntx == T16_NEVER ? NEVER : (r.ntx > (u32)r.ctt + 1 ? r.ntx - r.ctt - 1 : 0u);
                            d.rq_arrive[so] = need == NEVER ? NEVER : c0 + need;
                        }
                    }
                }
            }
        }
    }
    __syncthreads();  // receiving-queue pushes of this env are visible to their destination lanes

    // ---------------------------------------------------------------- load the per-lane working set
    u32 ex_n = 0, wq_head = 0, wq_len = 0, wq_min = NEVER, buf_head = 0, buf_len = 0;
    int deg = 0;
    double rew = 0.0, ct = 0.0;
    u32 n_fail = 0, n_drop = 0, n_fin = 0, n_succ = 0;
    Rng frng{nullptr, 0, 0}, trng{nullptr, 0, 0};
    double thr_exec = 0.0;
    if (L.valid) {
        ex_n = d.ex_n[lane];
        wq_head = d.wq_head[lane];
        wq_len = d.wq_len[lane];
        wq_min = d.wq_min[lane];
        buf_head = d.buf_head[lane];
        buf_len = d.buf_len[lane];
        deg = d.deg[node];
        thr_exec = d.thr_exec[node];
        trng = Rng{d.mt + (size_t)lane * MT_N, d.rng_pos[lane], 0};
        frng = Rng{d.mt + ((size_t)P + lane) * MT_N, d.rng_pos[P + lane], 0};
        for (u32 s = 0; s < ex_n; s++) {
            s_exA[s * NT + tid] = d.exA[(size_t)s * P + lane];
            s_exB[s * NT + tid] = d.exB[(size_t)s * P + lane];
        }
        for (int s = 0; s < deg; s++) {
            s_rqlen[s * NT + tid] = d.rq_len[(size_t)s * P + lane];
            s_rqarr[s * NT + tid] = d.rq_arrive[(size_t)s * P + lane];
        }
    }

    // ---------------------------------------------------------------- phase B: mini-slots, edge_computing_env.py:101-105
    if (L.valid) {
        for (int ms = 0; ms < d.S; ms++) {
            const u32 c = c0 + (u32)ms;
            // ---- execute_task, edge_node.py:236-309
            // (i) admit heads of the waiting queue while a slot is free (:238-241)
            while (ex_n < (u32)d.k && wq_len > 0) {
                const WRec w = load_rec(d.wq + (size_t)wq_head * P + lane);
                wq_head = wq_head + 1 == (u32)d.capW ? 0u : wq_head + 1;
                wq_len--;
                const u32 ewt = clamp16(d, c - w.enq);
                const u32 dc = deadline_ticks(d, 1, ewt, w.twt, w.ctt, w.bwt);
                s_exA[ex_n * NT + tid] = pack4(0, dc, w.nexec, w.hop);
                s_exB[ex_n * NT + tid] = pack4(ewt, w.twt, w.ctt, w.bwt);
                ex_n++;
            }
            if (wq_len == 0) wq_min = NEVER;
            // (ii) is implicit: waiting time is c - enq.  (iii) the executing tasks, in order (:252-296)
            {
                u32 w = 0;
                for (u32 s = 0; s < ex_n; s++) {
                    const u64 a = s_exA[s * NT + tid];
                    const u32 cet = f16(a, 0) + 1;
                    const double u = rng_double(frng);
                    bool done = true;
                    if (u > thr_exec) {
                        rew = dadd(rew, d.pen_fail);
                        n_fail++;
                        n_fin++;
                    } else if (cet >= f16(a, 1)) {
                        const u64 b = s_exB[s * NT + tid];
                        rew = dadd(rew, d.pen_drop);
                        n_drop++;
                        n_fin++;
                        ct = dadd(ct, task_time(d, f16(b, 0), cet, f16(b, 1), f16(b, 2), f16(b, 3)));
                    } else if (cet >= f16(a, 2)) {
                        const u64 b = s_exB[s * NT + tid];
                        const u32 hop = f16(a, 3);
                        rew = dadd(rew, d.rew_succ);
                        rew = dadd(rew, hop > 1 ? dmul(d.pen_hop, (double)(hop - 1)) : 0.0);
                        n_succ++;
                        n_fin++;
                        ct = dadd(ct, task_time(d, f16(b, 0), cet, f16(b, 1), f16(b, 2), f16(b, 3)));
                        u32 bin = hop;
                        if (bin >= (u32)d.H) {
                            bin = d.H - 1;
                            flag(d, EC_ST_HOP_OVERFLOW);
                        }
                        atomicAdd(d.hop_hist + L.env * d.H + bin, 1);
                    } else {
                        done = false;
                    }
                    if (!done) {
                        s_exA[w * NT + tid] = (a & ~0xffffull) | (u64)clamp16(d, cet);
                        if (w != s) s_exB[w * NT + tid] = s_exB[s * NT + tid];
                        w++;
                    }
                }
                ex_n = w;
            }
            // (iv) deadline sweep of the waiting queue (:299-309), only when some drop_at has come due
            if (wq_len > 0 && wq_min <= c) {
                u32 rd = wq_head, wr = wq_head, kept = 0, newmin = NEVER;
                for (u32 j = 0; j < wq_len; j++) {
                    const WRec w = load_rec(d.wq + (size_t)rd * P + lane);
                    if (w.drop_at <= c) {
                        n_drop++;
                        n_fin++;
                        rew = dadd(rew, d.pen_drop);
                        ct = dadd(ct, task_time(d, c - w.enq + 1, 0, w.twt, w.ctt, w.bwt));
                    } else {
                        if (wr != rd) store_rec(d.wq + (size_t)wr * P + lane, w);
                        wr = wr + 1 == (u32)d.capW ? 0u : wr + 1;
                        kept++;
                        newmin = w.drop_at < newmin ? w.drop_at : newmin;
                    }
                    rd = rd + 1 == (u32)d.capW ? 0u : rd + 1;
                }
                wq_len = kept;
                wq_min = newmin;
            }
            // ---- receive_task, edge_node.py:311-333: in-neighbours in ascending source order
            for (int s = 0; s < deg; s++) {
                u32 ql = s_rqlen[s * NT + tid];
                if (ql == 0) continue;
                const double u = rng_double(frng);
                const bool failed = u > d.thr_link[node * d.D + s];
                const bool arrived = !failed && c >= s_rqarr[s * NT + tid];
                if (!failed && !arrived) continue;
                const size_t so = (size_t)s * P + lane;
                u32 qh = d.rq_head[so];
                if (failed) {
                    n_fail++;
                    n_fin++;
                    rew = dadd(rew, d.pen_fail);
                } else {
                    Rec r = load_rec(d.rq + ((size_t)s * d.capR + qh) * P + lane);
                    const u32 hs = d.rq_hstart[so];
                    r.twt = (u16)clamp16(d, (u32)r.twt + (hs - r.enq));
                    r.ctt = (u16)clamp16(d, (u32)r.ctt + (c - hs + 1));
                    rew = dadd(rew, r.hop > 1 ? dmul(d.pen_hop, (double)(r.hop - 1)) : 0.0);
                    buffer_push(d, lane, buf_head, buf_len, n_off, r);
                }
                // popleft, then promote the next record to head for call c+1
                qh = qh + 1 == (u32)d.capR ? 0u : qh + 1;
                ql--;
                d.rq_head[so] = qh;
                u32 arr = NEVER;
                if (ql > 0) {
                    const Rec nx = load_rec(d.rq + ((size_t)s * d.capR + qh) * P + lane);
                    d.rq_hstart[so] = c + 1;
                    if (nx.ntx != T16_NEVER) arr = c + 1 + (nx.ntx > (u32)nx.ctt + 1 ? nx.ntx - nx.ctt - 1 : 0u);
                }
                s_rqlen[s * NT + tid] = ql;
                s_rqarr[s * NT + tid] = arr;
            }
        }
    }

    // ---------------------------------------------------------------- phase C: per-env statistics, edge_computing_env.py:108-123
    if (L.valid) {
        s_rew[tid] = rew;
        s_ct[tid] = ct;
        s_cnt[0 * NT + tid] = n_fin;
        s_cnt[1 * NT + tid] = n_succ;
        s_cnt[2 * NT + tid] = n_drop;
        s_cnt[3 * NT + tid] = n_fail;
    }
    __syncthreads();
    u32 step_now = 0;
    if (L.valid && node == 0) {
        double r = 0.0, t = 0.0;
        u32 c4[4] = {0, 0, 0, 0};
        for (int i = 0; i < N; i++) {  // node order, as the reference sums
            r = dadd(r, s_rew[tid + i]);
            t = dadd(t, s_ct[tid + i]);
            for (int q = 0; q < 4; q++) c4[q] += s_cnt[q * NT + tid + i];
        }
        step_now = d.cur_step[L.env] + 1;  // :125
        d.cur_step[L.env] = step_now;
        d.tick[L.env] = c0 + (u32)d.S;
        if (reward) reward[L.env] = (float)r;
        if (completion_time) completion_time[L.env] = (float)t;
        if (counters) {
            int4 v = make_int4((int)c4[0], (int)c4[1], (int)c4[2], (int)c4[3]);
            *reinterpret_cast<int4*>(counters + 4 * L.env) = v;
        }
        if (terminated) terminated[L.env] = step_now >= (u32)d.T ? 1 : 0;  // :139-142
    }

    // ---------------------------------------------------------------- phase D: generate_task, :127-128
    if (L.valid) {
        generate_task(d, lane, node, trng, buf_head, buf_len, n_off, has_new, newt);
        // write the working set back
        d.has_new[lane] = has_new;
        if (has_new) store_rec(d.newt + lane, newt);
        d.n_off[lane] = n_off;
        d.buf_head[lane] = buf_head;
        d.buf_len[lane] = buf_len;
        d.wq_head[lane] = wq_head;
        d.wq_len[lane] = wq_len;
        d.wq_min[lane] = wq_min;
        d.ex_n[lane] = ex_n;
        for (u32 s = 0; s < ex_n; s++) {
            d.exA[(size_t)s * P + lane] = s_exA[s * NT + tid];
            d.exB[(size_t)s * P + lane] = s_exB[s * NT + tid];
        }
        for (int s = 0; s < deg; s++) {
            d.rq_len[(size_t)s * P + lane] = s_rqlen[s * NT + tid];
            d.rq_arrive[(size_t)s * P + lane] = s_rqarr[s * NT + tid];
        }
        d.rng_pos[lane] = trng.pos;
        d.rng_pos[P + lane] = frng.pos;
        d.rng_cnt[lane] += trng.used;
        d.rng_cnt[P + lane] += frng.used;
    }

    // ---------------------------------------------------------------- phase E: fused observation of the new state
    if (next_obs || next_avail) {
        __syncthreads();  // the sweep's compaction stores are ordered before the waiting-time loads
        observe_block(d, L, wq_head, wq_len, has_new, newt, next_obs, next_avail, s_wql, s_hasnew, s_any, s_obs);
    }
}

__global__ void k_copy_i32(const int32_t* __restrict__ src, int32_t* __restrict__ dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// Uniform choice over the availability mask with a counter-based generator (splitmix64 finaliser).
// counter_dev (optional): word 0 = launch counter, word 1 = block ticket; the last block to finish bumps the
// counter, so a captured hipGraph draws fresh numbers on every replay.
__global__ void k_random_policy(const int32_t* __restrict__ avail, int64_t rows, int A, u64 seed, u64 counter,
                                u64* counter_dev, int64_t* __restrict__ actions) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (counter_dev) {
        const u64 base = counter_dev[0];
        counter += base;
        __syncthreads();  // every thread of this block has read the counter
        if (threadIdx.x == 0) {
            const u64 t = atomicAdd((unsigned long long*)(counter_dev + 1), 1ull);
            if (t == (u64)gridDim.x - 1) {  // all other blocks have read it too
                counter_dev[1] = 0;
                counter_dev[0] = base + 1;
            }
        }
    }
    if (r >= rows) return;
    const int32_t* a = avail + r * A;
    int cnt = 0;
    for (int j = 0; j < A; j++) cnt += a[j] != 0;
    u64 z = seed + 0x9E3779B97F4A7C15ull * (counter * (u64)rows + (u64)r + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    int pick = cnt > 0 ? (int)(((z >> 32) * (u64)cnt) >> 32) : 0;
    int act = A - 1;
    for (int j = 0; j < A; j++) {
        if (a[j] != 0) {
            if (pick == 0) {
                act = j;
                break;
            }
            pick--;
        }
    }
    actions[r] = act;
}

// parity-test helper: one thread serialises env `env`
__global__ void k_dump(EcDev d, int64_t env, double* rows, int64_t max_rows, int64_t* n_rows) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    int64_t r = 0;
    const u32 now = d.tick[env];
    auto put = [&](int node, int qt, int pos, double size, double cycles, u32 hop, u32 ewt, u32 cet, u32 twt,
                   u32 ctt, u32 bwt) {
        if (r < max_rows) {
            double* o = rows + 11 * r;
            o[0] = node; o[1] = qt; o[2] = pos; o[3] = size; o[4] = cycles; o[5] = hop;
            o[6] = ewt; o[7] = cet; o[8] = twt; o[9] = ctt; o[10] = bwt;
        }
        r++;
    };
    for (int i = 0; i < d.N; i++) {
        const int64_t lane = env * d.N + i;
        const u32 n_off = d.n_off[lane];
        if (d.has_new[lane]) {
            const Rec t = d.newt[lane];
            put(i, 0, 0, t.size, t.cycles, t.hop, 0, 0, t.twt, t.ctt, (u32)t.bwt);
        }
        u32 pos = d.buf_head[lane];
        for (u32 j = 0; j < d.buf_len[lane]; j++) {
            const Rec t = d.buf[(size_t)pos * d.P + lane];
            put(i, 1, j, t.size, t.cycles, t.hop, 0, 0, t.twt, t.ctt, (u32)(t.bwt + (int32_t)n_off));
            pos = pos + 1 == (u32)d.capB ? 0u : pos + 1;
        }
        pos = d.wq_head[lane];
        for (u32 j = 0; j < d.wq_len[lane]; j++) {
            const WRec w = d.wq[(size_t)pos * d.P + lane];
            // size/cycles are not kept once a task is queued for execution: report et and 0
            put(i, 2, j, w.et, 0.0, w.hop, now - w.enq, 0, w.twt, w.ctt, w.bwt);
            pos = pos + 1 == (u32)d.capW ? 0u : pos + 1;
        }
        for (u32 j = 0; j < d.ex_n[lane]; j++) {
            const u64 a = d.exA[(size_t)j * d.P + lane], b = d.exB[(size_t)j * d.P + lane];
            put(i, 3, j, (double)f16(a, 2), (double)f16(a, 1), f16(a, 3), f16(b, 0), f16(a, 0), f16(b, 1), f16(b, 2),
                f16(b, 3));
        }
        for (int s = 0; s < d.deg[i]; s++) {
            const size_t so = (size_t)s * d.P + lane;
            u32 p2 = d.rq_head[so];
            const u32 hs = d.rq_hstart[so];
            for (u32 j = 0; j < d.rq_len[so]; j++) {
                const Rec t = d.rq[((size_t)s * d.capR + p2) * d.P + lane];
                u32 twt, ctt;
                if (j == 0) {
                    twt = t.twt + (hs - t.enq);
                    ctt = t.ctt + (now - hs);
                } else {
                    twt = t.twt + (now - t.enq);
                    ctt = t.ctt;
                }
                put(i, 4 + d.in_src[i * d.D + s], j, t.size, t.cycles, t.hop, 0, 0, twt, ctt, (u32)t.bwt);
                p2 = p2 + 1 == (u32)d.capR ? 0u : p2 + 1;
            }
        }
    }
    *n_rows = r;
}

}  // namespace

// ------------------------------------------------------------------------------------------ host side
struct ec_env {
    EcDev d;
    int device;
    int threads;       // per block
    int envs_per_block;
    int64_t blocks;
    size_t lds_step, lds_obs;
    std::vector<void*> allocs;
    size_t state_bytes;
};

namespace {

int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

template <typename T>
int dev_alloc(ec_env* h, T** out, size_t count, bool zero = false) {
    void* p = nullptr;
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = sizeof(T);
    if (hipMalloc(&p, bytes) != hipSuccess) return ec::fail(EC_ENOMEM, "hipMalloc of %zu bytes failed", bytes);
    h->allocs.push_back(p);
    h->state_bytes += bytes;
    if (zero && hipMemset(p, 0, bytes) != hipSuccess) return ec::fail(EC_EHIP, "hipMemset failed");
    *out = static_cast<T*>(p);
    return EC_OK;
}

template <typename T>
int dev_upload(ec_env* h, const T** out, const std::vector<T>& v) {
    T* p = nullptr;
    int rc = dev_alloc(h, &p, v.size());
    if (rc) return rc;
    if (!v.empty()) EC_HIP(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = p;
    return EC_OK;
}

void free_all(ec_env* h) {
    for (void* p : h->allocs) (void)hipFree(p);
    h->allocs.clear();
}

#define EC_TRY(expr)          \
    do {                      \
        int rc_ = (expr);     \
        if (rc_) return rc_;  \
    } while (0)

int build(ec_env* h, const ec_config* cfg, const ec_topology* topo, int64_t B, int64_t env_id_base) {
    EcDev& d = h->d;
    const int N = cfg->n_nodes;
    d.N = N;
    d.A = N + 1;
    d.k = cfg->k;
    d.S = cfg->mini_slots;
    d.T = cfg->episode_limit;
    d.H = cfg->hop_hist_len > 0 ? cfg->hop_hist_len : 32;
    d.B = B;
    d.P = B * N;
    d.dt = cfg->mini_time_slot;
    d.deadline = cfg->deadline;
    d.size_lo = cfg->task_size_min;
    d.size_range = cfg->task_size_max - cfg->task_size_min;  // numpy uniform: range = high - low
    d.cx_lo = cfg->task_complexity_min;
    d.cx_range = cfg->task_complexity_max - cfg->task_complexity_min;
    d.pen_drop = cfg->task_drop_penalty;
    d.pen_fail = cfg->task_failure_penalty;
    d.rew_succ = cfg->task_success_reward;
    d.pen_hop = cfg->task_hop_penalty;
    d.obs_size_norm = cfg->task_size_max * 1024 * 8;                                  // edge_node.py:126
    d.obs_cyc_norm = cfg->task_complexity_max * cfg->task_size_max * 1024 * 8;       // edge_node.py:128

    // in-neighbour lists (ascending source id == the order receive_task visits its queues)
    std::vector<int> deg(N, 0);
    int D = 1;
    for (int i = 0; i < N; i++) {
        for (int s = 0; s < N; s++)
            if (s != i && topo->adj[s * (N + 1) + i]) deg[i]++;
        if (deg[i] > D) D = deg[i];
    }
    d.D = D;
    std::vector<int> in_src((size_t)N * D, -1), slot_of((size_t)N * N, -1);
    std::vector<double> thr_link((size_t)N * D, 1.0);
    for (int i = 0; i < N; i++) {
        int c = 0;
        for (int s = 0; s < N; s++) {
            if (s == i || !topo->adj[s * (N + 1) + i]) continue;
            if (topo->rate[s * N + i] == 0.0) return ec::fail(EC_EINVAL, "adjacent pair (%d,%d) has zero rate", s, i);
            in_src[i * D + c] = s;
            slot_of[s * N + i] = c;
            // edge_node.py:321: exp(-transmission_failure_rate * mini_time_slot), rate taken from the sender's table
            thr_link[i * D + c] = exp(-topo->tfr[s * N + i] * cfg->mini_time_slot);
            c++;
        }
    }

    // ring capacities: bounds that hold for any policy within one episode
    const int steps_per_deadline = (int)ceil(cfg->deadline / (cfg->mini_time_slot * cfg->mini_slots));
    d.capW = cfg->cap_wait > 0 ? cfg->cap_wait : next_pow2(steps_per_deadline + 3);
    d.capR = cfg->cap_recv > 0 ? cfg->cap_recv : next_pow2(cfg->episode_limit + 2);
    d.capB = cfg->cap_buffer > 0 ? cfg->cap_buffer : next_pow2((D + 1) * (cfg->episode_limit + 2));
    d.maxTicks = cfg->max_ticks > 0 ? cfg->max_ticks : cfg->mini_slots * (cfg->episode_limit + 2);
    if (d.maxTicks > 0xfff0) return ec::fail(EC_EINVAL, "max_ticks %d exceeds the 16-bit tick range", d.maxTicks);
    if (d.capB > 0xffff || d.capR > 0xffff) return ec::fail(EC_EINVAL, "ring capacity too large");

    std::vector<double> acc(d.maxTicks + 1);
    acc[0] = 0.0;
    for (int n = 1; n <= d.maxTicks; n++) acc[n] = acc[n - 1] + cfg->mini_time_slot;  // edge_node.py:216,250,254,316,319

    std::vector<double> cap(N), thr_exec(N), qlen(N), qthr(N), bqn(N), bpx2(N), rate((size_t)N * N);
    std::vector<float> obs_const((size_t)N * 3);
    std::vector<uint8_t> bflip(N), adj((size_t)N * (N + 1));
    for (int i = 0; i < N; i++) {
        cap[i] = topo->cpu_capacity[i];
        thr_exec[i] = exp(-topo->efr[i] * cfg->mini_time_slot);  // edge_node.py:257
        qlen[i] = topo->queue_len[i];
        qthr[i] = topo->queue_len[i] * cfg->rl_queue_coeff;
        obs_const[3 * i + 0] = (float)(topo->efr[i] / cfg->execution_failure_rate_max);
        obs_const[3 * i + 1] = (float)(topo->cores[i] / cfg->core_norm);
        obs_const[3 * i + 2] = (float)(topo->task_p[i] / cfg->task_probability_max);
        // numpy legacy binomial(1, p) set-up (random_binomial_inversion): q = 1-p, qn = exp(n*log(q)),
        // and the X=1 continuation threshold px = ((n-X+1)*p*px)/(X*q)
        double p = topo->task_p[i];
        bflip[i] = p > 0.5;
        if (bflip[i]) p = 1.0 - p;
        const double q = 1.0 - p;
        const double qn = exp(1 * log(q));
        bqn[i] = qn;
        bpx2[i] = ((1 - 1 + 1) * p * qn) / (1 * q);
    }
    memcpy(rate.data(), topo->rate, sizeof(double) * N * N);
    memcpy(adj.data(), topo->adj, (size_t)N * (N + 1));

    EC_TRY(dev_upload(h, &d.acc, acc));
    EC_TRY(dev_upload(h, &d.cap, cap));
    EC_TRY(dev_upload(h, &d.thr_exec, thr_exec));
    EC_TRY(dev_upload(h, &d.qlen, qlen));
    EC_TRY(dev_upload(h, &d.qthr, qthr));
    EC_TRY(dev_upload(h, &d.obs_const, obs_const));
    EC_TRY(dev_upload(h, &d.bin_qn, bqn));
    EC_TRY(dev_upload(h, &d.bin_px2, bpx2));
    EC_TRY(dev_upload(h, &d.bin_flip, bflip));
    EC_TRY(dev_upload(h, &d.deg, deg));
    EC_TRY(dev_upload(h, &d.in_src, in_src));
    EC_TRY(dev_upload(h, &d.thr_link, thr_link));
    EC_TRY(dev_upload(h, &d.slot_of, slot_of));
    EC_TRY(dev_upload(h, &d.rate, rate));
    EC_TRY(dev_upload(h, &d.adj, adj));

    const size_t P = (size_t)d.P;
    EC_TRY(dev_alloc(h, &d.mt, 2 * P * MT_N));
    EC_TRY(dev_alloc(h, &d.rng_pos, 2 * P));
    EC_TRY(dev_alloc(h, &d.rng_cnt, 2 * P));
    EC_TRY(dev_alloc(h, &d.has_new, P));
    EC_TRY(dev_alloc(h, &d.newt, P, true));
    EC_TRY(dev_alloc(h, &d.buf_head, P));
    EC_TRY(dev_alloc(h, &d.buf_len, P));
    EC_TRY(dev_alloc(h, &d.buf, (size_t)d.capB * P));
    EC_TRY(dev_alloc(h, &d.wq_head, P));
    EC_TRY(dev_alloc(h, &d.wq_len, P));
    EC_TRY(dev_alloc(h, &d.wq_min, P));
    EC_TRY(dev_alloc(h, &d.wq, (size_t)d.capW * P));
    EC_TRY(dev_alloc(h, &d.ex_n, P));
    EC_TRY(dev_alloc(h, &d.exA, (size_t)d.k * P));
    EC_TRY(dev_alloc(h, &d.exB, (size_t)d.k * P));
    EC_TRY(dev_alloc(h, &d.rq_head, (size_t)D * P));
    EC_TRY(dev_alloc(h, &d.rq_len, (size_t)D * P));
    EC_TRY(dev_alloc(h, &d.rq_arrive, (size_t)D * P));
    EC_TRY(dev_alloc(h, &d.rq_hstart, (size_t)D * P));
    EC_TRY(dev_alloc(h, &d.rq, (size_t)D * d.capR * P));
    EC_TRY(dev_alloc(h, &d.n_off, P));
    EC_TRY(dev_alloc(h, &d.cur_step, (size_t)B));
    EC_TRY(dev_alloc(h, &d.tick, (size_t)B));
    EC_TRY(dev_alloc(h, &d.hop_hist, (size_t)B * d.H));
    EC_TRY(dev_alloc(h, &d.status, 1, true));

    // launch geometry: whole envs per block; small batches use one wave per block to spread over the CUs
    int threads = 256;
    if (const char* e = getenv("MARL_AMD_EC_THREADS")) threads = atoi(e);
    else if (d.P <= 256 * 256) threads = 64;
    if (threads < N) threads = ((N + 63) / 64) * 64;
    if (threads > 1024 || threads % 64 != 0) return ec::fail(EC_EINVAL, "threads per block %d unsupported (n_nodes=%d)", threads, N);
    h->threads = threads;
    h->envs_per_block = threads / N;
    h->blocks = (B + h->envs_per_block - 1) / h->envs_per_block;
    h->lds_obs = (size_t)threads * (3 * 4 + 9 * 4);
    h->lds_step = (size_t)threads * ((size_t)d.k * 16 + 16 + (size_t)D * 8 + 16 + 12 + 36);
    if (h->lds_step > 160 * 1024) return ec::fail(EC_EINVAL, "LDS need %zu exceeds 160 KiB", h->lds_step);
    EC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_step), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)h->lds_step));

    const u32 ts = (u32)(cfg->task_seed + env_id_base), fs = (u32)(cfg->failure_seed + env_id_base);
    const int64_t gb = (d.P + 255) / 256;
    hipLaunchKernelGGL(k_seed, dim3((unsigned)gb), dim3(256), 0, 0, d, ts, fs);
    hipLaunchKernelGGL(k_initial_generate, dim3((unsigned)gb), dim3(256), 0, 0, d);
    EC_HIP(hipGetLastError());
    EC_HIP(hipDeviceSynchronize());
    return EC_OK;
}

}  // namespace

extern "C" {

int ec_create(const ec_config* cfg, const ec_topology* topo, int64_t n_envs, int64_t env_id_base, int device,
              ec_handle* out) {
    if (!cfg || !topo || !out) return ec::fail(EC_EINVAL, "null argument");
    if (cfg->n_nodes < 2 || cfg->n_nodes > 1024) return ec::fail(EC_EINVAL, "n_nodes %d out of range [2,1024]", cfg->n_nodes);
    if (n_envs < 1) return ec::fail(EC_EINVAL, "n_envs must be positive");
    if (cfg->k < 1 || cfg->k > 64) return ec::fail(EC_EINVAL, "k %d out of range [1,64]", cfg->k);
    if (cfg->mini_slots < 1 || cfg->mini_time_slot <= 0) return ec::fail(EC_EINVAL, "bad mini slot configuration");
    const int64_t top = n_envs + env_id_base + cfg->n_nodes;
    // numpy: RandomState(seed) needs 0 <= seed < 2**32
    if (cfg->task_seed < 0 || cfg->failure_seed < 0 || env_id_base < 0 || cfg->task_seed + top > 0xFFFFFFFFll ||
        cfg->failure_seed + top > 0xFFFFFFFFll)
        return ec::fail(EC_EINVAL, "seed out of RandomState range");
    EC_HIP(hipSetDevice(device));
    ec_env* h = new ec_env();
    memset(&h->d, 0, sizeof(h->d));
    h->device = device;
    h->state_bytes = 0;
    int rc = build(h, cfg, topo, n_envs, env_id_base);
    if (rc) {
        free_all(h);
        delete h;
        return rc;
    }
    *out = h;
    return EC_OK;
}

int ec_destroy(ec_handle h) {
    if (!h) return EC_OK;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    free_all(h);
    delete h;
    return EC_OK;
}

int ec_reset(ec_handle h, void* stream) {
    if (!h) return ec::fail(EC_EINVAL, "null handle");
    const int64_t gb = (h->d.P + 255) / 256;
    hipLaunchKernelGGL(k_reset, dim3((unsigned)gb), dim3(256), 0, (hipStream_t)stream, h->d);
    EC_HIP(hipGetLastError());
    return EC_OK;
}

int ec_observe(ec_handle h, float* obs, int32_t* avail, void* stream) {
    if (!h) return ec::fail(EC_EINVAL, "null handle");
    if (!obs && !avail) return EC_OK;
    hipLaunchKernelGGL(k_observe, dim3((unsigned)h->blocks), dim3(h->threads), h->lds_obs, (hipStream_t)stream, h->d,
                       obs, avail);
    EC_HIP(hipGetLastError());
    return EC_OK;
}

int ec_step(ec_handle h, const int64_t* actions, float* reward, uint8_t* terminated, int32_t* counters,
            float* completion_time, int32_t* hop_hist, float* next_obs, int32_t* next_avail, void* stream) {
    if (!h || !actions) return ec::fail(EC_EINVAL, "null handle or actions");
    hipLaunchKernelGGL(k_step, dim3((unsigned)h->blocks), dim3(h->threads), h->lds_step, (hipStream_t)stream, h->d,
                       actions, reward, terminated, counters, completion_time, next_obs, next_avail);
    if (hop_hist) {
        const int64_t n = h->d.B * h->d.H;
        hipLaunchKernelGGL(k_copy_i32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           h->d.hop_hist, hop_hist, n);
    }
    EC_HIP(hipGetLastError());
    return EC_OK;
}

int ec_random_policy(const int32_t* avail, int64_t rows, int32_t n_actions, uint64_t seed, uint64_t counter,
                     uint64_t* counter_dev, int64_t* actions, void* stream) {
    if (!avail || !actions || rows < 0 || n_actions < 1) return ec::fail(EC_EINVAL, "bad argument");
    if (rows == 0) return EC_OK;
    hipLaunchKernelGGL(k_random_policy, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, avail,
                       rows, (int)n_actions, seed, counter, counter_dev, actions);
    EC_HIP(hipGetLastError());
    return EC_OK;
}

int ec_status(ec_handle h, uint32_t* status_host, int clear, void* stream) {
    if (!h || !status_host) return ec::fail(EC_EINVAL, "null argument");
    EC_HIP(hipMemcpyAsync(status_host, h->d.status, sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    if (clear) EC_HIP(hipMemsetAsync(h->d.status, 0, sizeof(uint32_t), (hipStream_t)stream));
    EC_HIP(hipStreamSynchronize((hipStream_t)stream));
    return EC_OK;
}

int ec_debug_dump(ec_handle h, int64_t env, double* rows_host, int64_t max_rows, int64_t* n_rows_host) {
    if (!h || !rows_host || !n_rows_host || env < 0 || env >= h->d.B) return ec::fail(EC_EINVAL, "bad argument");
    double* rows = nullptr;
    int64_t* n = nullptr;
    EC_HIP(hipDeviceSynchronize());
    EC_HIP(hipMalloc(&rows, sizeof(double) * 11 * (size_t)(max_rows > 0 ? max_rows : 1)));
    EC_HIP(hipMalloc(&n, sizeof(int64_t)));
    hipLaunchKernelGGL(k_dump, dim3(1), dim3(64), 0, 0, h->d, env, rows, max_rows, n);
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(n_rows_host, n, sizeof(int64_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess) {
        const int64_t cnt = *n_rows_host < max_rows ? *n_rows_host : max_rows;
        if (cnt > 0) e = hipMemcpy(rows_host, rows, sizeof(double) * 11 * (size_t)cnt, hipMemcpyDeviceToHost);
    }
    (void)hipFree(rows);
    (void)hipFree(n);
    if (e != hipSuccess) return ec::fail(EC_EHIP, "ec_debug_dump: %s", hipGetErrorString(e));
    return EC_OK;
}

int ec_debug_rng_cursors(ec_handle h, uint64_t* cursors_host) {
    if (!h || !cursors_host) return ec::fail(EC_EINVAL, "null argument");
    const size_t P = (size_t)h->d.P;
    std::vector<u64> tmp(2 * P);
    EC_HIP(hipDeviceSynchronize());
    EC_HIP(hipMemcpy(tmp.data(), h->d.rng_cnt, sizeof(u64) * 2 * P, hipMemcpyDeviceToHost));
    for (size_t l = 0; l < P; l++) {
        cursors_host[2 * l + 0] = tmp[l];
        cursors_host[2 * l + 1] = tmp[P + l];
    }
    return EC_OK;
}

int ec_debug_live_tasks(ec_handle h, int64_t* total_host) {
    if (!h || !total_host) return ec::fail(EC_EINVAL, "null argument");
    const size_t P = (size_t)h->d.P;
    std::vector<u32> tmp(P * (size_t)(4 + h->d.D));
    EC_HIP(hipDeviceSynchronize());
    const u32* srcs[4] = {h->d.has_new, h->d.buf_len, h->d.wq_len, h->d.ex_n};
    for (int i = 0; i < 4; i++) EC_HIP(hipMemcpy(tmp.data() + i * P, srcs[i], sizeof(u32) * P, hipMemcpyDeviceToHost));
    EC_HIP(hipMemcpy(tmp.data() + 4 * P, h->d.rq_len, sizeof(u32) * P * h->d.D, hipMemcpyDeviceToHost));
    int64_t t = 0;
    for (u32 v : tmp) t += v;
    *total_host = t;
    return EC_OK;
}

int ec_info(ec_handle h, int64_t* info_host, int32_t n) {
    if (!h || !info_host) return ec::fail(EC_EINVAL, "null argument");
    const int64_t v[12] = {h->d.B, h->d.N, h->d.k, h->d.D, h->d.capB, h->d.capR, h->d.capW, h->d.maxTicks,
                           h->threads, h->envs_per_block, (int64_t)(h->state_bytes & 0xffffffffull),
                           (int64_t)(h->state_bytes >> 32)};
    for (int i = 0; i < n && i < 12; i++) info_host[i] = v[i];
    return EC_OK;
}

const char* ec_last_error(void) { return ec::err_buf(); }
const char* ec_version(void) { return "marl_amd 0.1 (gfx950)"; }

}  // extern "C"
