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
// produces), and waiting is accounted lazily from enqueue ticks, so a mini-slot touches a task only
// when something happens to it.  Deadline / completion thresholds are turned into integer tick
// thresholds once per task by evaluating the reference's left-to-right fp64 sum on the LUT.
//
// Memory plan of k_step: the working set of a lane (its executing slots, its waiting queue, the head
// of each receiving queue, the next doubles of both MT19937 streams, the low part of the tick LUT) is
// staged into LDS/registers with independent, lane-contiguous loads at kernel entry, the ten
// mini-slots then run out of LDS, and the set is written back at exit; global memory is touched
// inside the mini-slot loop only for the rare deep-queue cases.
//
// Layout: every per-node array is [slot][P] with P = B*N lanes minor, so a wave's access to "slot s of
// my queue" is one contiguous run; task records are 32-byte (two dwordx4) elements.  MT19937 state is
// per lane [P][624], regenerated in place ahead of consumption in 32-word batches.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "ec_common.h"
#include "ec_policy_inline.h"

namespace {

typedef uint32_t u32;
typedef uint64_t u64;
typedef uint16_t u16;

constexpr int MT_N = 624;
constexpr int MT_M = 397;
constexpr int MT_MIRROR = 96;              // words [0, MT_MIRROR) are repeated after the state (wrap-free reads)
constexpr int MT_STRIDE = MT_N + MT_MIRROR; // words per stream in memory
typedef uint32_t u32x4 __attribute__((ext_vector_type(4), aligned(8)));
typedef int32_t i32x4u __attribute__((ext_vector_type(4), aligned(4)));  // rows of 4-byte elements: 16-byte stores at any word
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
constexpr u32 NEVER = 0xFFFFFFFFu;
constexpr u32 T16_NEVER = 0xFFFFu;
constexpr int RB_FAIL = 40;  // draws of the failure stream held in the LDS window (>= KMAX + DMAX; 2*RB_FAIL <= MT_MIRROR)
constexpr int AHEAD_FAIL = 2 * RB_FAIL;  // words regenerated ahead of the failure cursor at kernel entry
constexpr int RB_TASK = 4;   // doubles of the task stream staged per refill (binomial + id + size + cycles)
constexpr int AHEAD_TASK = 2 * RB_TASK;
constexpr int RB_EARLY = 16;  // draws of the window staged before the step's first barrier; the rest follows during the mini-slots
constexpr u32 STAGED_FINAL = 0x80000000u;  // flag on the window's staged count: the producer will add no more
constexpr u32 STAGED_COUNT = 0xffu;        // RC_STAGED word: bits 0-7 draws staged, bits 8-23 "hot" flags of the first
constexpr int STAGED_HOT_SHIFT = 8;        // RB_EARLY draws (see StepLds::hot), bit 31 STAGED_FINAL
constexpr int AHEAD_MAX = 512;  // cap of the words kept regenerated ahead (a 64-word batch must not lap the cursor)
constexpr int ACC_LDS = 256; // 8-byte words of LDS for the tick LUT: its first 4 * ACC_LDS entries (16-bit corrections)
constexpr int ACC_LDS_N = 4 * ACC_LDS;

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
static_assert(sizeof(Rec) == 32, "record size");

// waiting-queue (execution_queue) entry: 16-byte meta + fp64 execute_time
//   meta.x = enq   index of the first execute_task call after the push
//   meta.y = nd | nexec << 16     nd: the deadline sweep drops it at call enq + nd - 1 (edge_node.py:299-309)
//                                 nexec: min n with acc[n] >= execute_time
//   meta.z = twt | ctt << 16
//   meta.w = bwt | hop << 16
struct WMeta {
    u32 enq, nd, nexec, twt, ctt, bwt, hop;
};
__device__ __forceinline__ uint4 wmeta_pack(const WMeta& m) {
    return make_uint4(m.enq, (m.nd & 0xffffu) | (m.nexec << 16), (m.twt & 0xffffu) | (m.ctt << 16),
                      (m.bwt & 0xffffu) | (m.hop << 16));
}
__device__ __forceinline__ WMeta wmeta_unpack(const uint4 v) {
    WMeta m;
    m.enq = v.x;
    m.nd = v.y & 0xffffu;
    m.nexec = v.y >> 16;
    m.twt = v.z & 0xffffu;
    m.ctt = v.z >> 16;
    m.bwt = v.w & 0xffffu;
    m.hop = v.w >> 16;
    return m;
}
__device__ __forceinline__ u32 wmeta_drop_at(const uint4 v) {
    const u32 nd = v.y & 0xffffu;
    return nd == T16_NEVER ? NEVER : v.x + nd - 1;
}

struct FastDiv {
    u32 d, m;
};

struct alignas(16) EcDev {
    // geometry
    int N, A, k, D, S, T, H;
    int capB, capR, capW, maxTicks;
    int epb;       // environments per block (launch geometry)
    int epw;       // environments per main wave of k_step (0: an env spans waves, lanes are contiguous)
    int nmain;     // main threads per k_step block; threads beyond them are the RNG producer lanes
    int dl_ticks;  // (int)(deadline / mini_time_slot): starting point of the deadline searches
    FastDiv divN, divA;
    int fast_dl;   // the host has checked the bound behind Lut::deadline_ticks' closed form (see build())
    int acc_words; // 8-byte words of LDS holding the tick LUT: all of it (>= ACC_LDS, 4 entries per word)
    int link_lds;  // the looping kernels keep the out-link tables (slot, rate) in LDS (small topologies: N <= LINK_LDS_N)
    int64_t B, P;
    // scalars
    double dt, inv_dt, deadline;
    double size_lo, size_range, cx_lo, cx_range;
    double pen_drop, pen_fail, rew_succ, pen_hop;
    double obs_size_norm, obs_cyc_norm;
    // constant tables
    const int16_t* acc;       // [>= max(maxTicks + 1, ACC_LDS_N), multiple of 4] tick LUT as corrections: the reference's
                              //     timer after n additions of dt is the double whose bit pattern is
                              //     bits(fl(n * dt)) + acc[n] (computed on the host from the same additions)
    const double* cap;        // [N]
    const u64* thr_exec_m;    // [N] floor(exp(-efr*dt) * 2^53)
    const u32* hot_thr;       // [N] lowest 27-bit first-word threshold of the node's failure tests (execution and every
                              //     in-link): a draw whose first word is below it fails no test (event skipping)
    const double* qlen;       // [N]
    const double* qthr;       // [N] qlen * rl_queue_coeff
    const float* obs_const;   // [N][3]
    const double* bin_qn;     // [N]
    const double* bin_px2;    // [N]
    const uint8_t* bin_flip;  // [N] p > 0.5
    const int* deg;           // [N] in-degree
    const int* in_src;        // [N][D] ascending source ids
    const u64* thr_link_m;    // [N][D] floor(exp(-tfr[src][dst]*dt) * 2^53)
    const int* slot_of;       // [N(src)][N(dst)] slot of src in dst's in-list, -1 if not adjacent
    const double* rate;       // [N][N]
    const uint8_t* adj;       // [N][A]
    const u64* adj_bits;      // [N] adj rows as bit masks (N <= 64), else null
    const double* efr;        // [N] execution failure rates (expert policy)
    const double* tfr;        // [N][N] link failure rates (expert policy)
    const int* cores;         // [N]
    // per-lane state
    u32* mt;        // [2][P][MT_STRIDE]
    u32* rng_pos;   // [2][P] next word to consume
    u32* rng_ahead; // [2][P] words already regenerated beyond rng_pos
    u64* rng_cnt;   // [2][P] words consumed so far
    u32* has_new;   // [P]
    Rec* newt;      // [P]
    u32* buf_head;  // [P]
    u32* buf_len;   // [P]
    Rec* buf;       // [capB][P] ring
    u32* wq_len;    // [P]
    uint4* wqm;     // [capW][P] linear (index 0 = head)
    double* wqe;    // [capW][P]
    u32* ex_n;      // [P]
    u64* exA;       // [k][P] cet | fin << 16 | hop << 32 | success << 48
    double* exB;    // [k][P] task_time at cet == fin
    u64* exC;       // [k][P] {ewt, twt, ctt, bwt}
    u32* rq_head;   // [D][P]
    u32* rq_len;    // [D][P]
    u32* rq_arrive; // [D][P] receive_task call index at which the head arrives
    u32* rq_hstart; // [D][P] first call index with the current head as head
    Rec* rq;        // [D][capR][P] rings
    u32* n_off;     // [P] offload events (lazy buffer_waiting_time)
    // per-env state
    u32* cur_step;  // [B]
    u32* tick;      // [B] execute/receive call index
    u32* total_steps;  // [B] steps taken since creation (never reset; feeds the fused random policy's counter)
    int32_t* hop_hist;  // [B][H]
    u32* status;    // [1]
    u64* stamps;    // [blocks][16] phase time stamps (EC_STAMPS diagnostic build only), else null
};

// The looping kernels' view of the geometry block.  A step needs a few dozen of its ~150 words and the launch's first and
// last steps most of the rest; read into scalar registers as a whole at the top of every step (the only way to keep the
// compiler from holding all of it live across the loop) it cost 38 LDS reads, 152 v_readfirstlane, ~130 spill writes and
// 110 scalar adds per step: 1.8 of a step's 18.8 us, by the stamps.  Here the block sits in the lanes of DEV_VREGS vector
// registers for the whole launch (lane i of register j = word 64 j + i, three loads per wave and launch) and a member is
// read where it is used: one v_readlane per word -- what a scalar spill reload costs, without the spill.  Members
// convert implicitly, so `d.N`, `d.cap[node]`, `d.rq + i` read as they do on the plain struct; pointers come out as
// global-address-space pointers (built from the integer, so there is no generic -> global cast to fold away).
constexpr int DEV_WORDS = (int)(sizeof(EcDev) / 4);
constexpr int DEV_VREGS = (DEV_WORDS + 63) / 64;
template <typename T, size_t OFF>
struct DevField {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "geometry members are one or two words");
    static_assert(OFF % 4 == 0, "word-aligned member");
    const u32 (&v)[DEV_VREGS];
    __device__ __forceinline__ operator T() const {
        constexpr int w = (int)(OFF / 4);
        const u32 lo = (u32)__builtin_amdgcn_readlane((int)v[w / 64], w % 64);
        if constexpr (sizeof(T) == 4) {
            T r;
            __builtin_memcpy(&r, &lo, 4);
            return r;
        } else {
            const u32 hi = (u32)__builtin_amdgcn_readlane((int)v[(w + 1) / 64], (w + 1) % 64);
            const u64 x = (u64)lo | ((u64)hi << 32);
            if constexpr (std::is_pointer<T>::value) {
                typedef typename std::remove_pointer<T>::type E;
                typedef E __attribute__((address_space(1))) GE;
                return (T) reinterpret_cast<GE*>(x);
            } else {
                T r;
                __builtin_memcpy(&r, &x, 8);
                return r;
            }
        }
    }
};
struct EcDevL {
    DevField<decltype(EcDev::N), offsetof(EcDev, N)> N;
    DevField<decltype(EcDev::A), offsetof(EcDev, A)> A;
    DevField<decltype(EcDev::k), offsetof(EcDev, k)> k;
    DevField<decltype(EcDev::D), offsetof(EcDev, D)> D;
    DevField<decltype(EcDev::S), offsetof(EcDev, S)> S;
    DevField<decltype(EcDev::T), offsetof(EcDev, T)> T;
    DevField<decltype(EcDev::H), offsetof(EcDev, H)> H;
    DevField<decltype(EcDev::capB), offsetof(EcDev, capB)> capB;
    DevField<decltype(EcDev::capR), offsetof(EcDev, capR)> capR;
    DevField<decltype(EcDev::capW), offsetof(EcDev, capW)> capW;
    DevField<decltype(EcDev::maxTicks), offsetof(EcDev, maxTicks)> maxTicks;
    DevField<decltype(EcDev::epb), offsetof(EcDev, epb)> epb;
    DevField<decltype(EcDev::epw), offsetof(EcDev, epw)> epw;
    DevField<decltype(EcDev::nmain), offsetof(EcDev, nmain)> nmain;
    DevField<decltype(EcDev::dl_ticks), offsetof(EcDev, dl_ticks)> dl_ticks;
    DevField<decltype(EcDev::divN), offsetof(EcDev, divN)> divN;
    DevField<decltype(EcDev::divA), offsetof(EcDev, divA)> divA;
    DevField<decltype(EcDev::fast_dl), offsetof(EcDev, fast_dl)> fast_dl;
    DevField<decltype(EcDev::acc_words), offsetof(EcDev, acc_words)> acc_words;
    DevField<decltype(EcDev::link_lds), offsetof(EcDev, link_lds)> link_lds;
    DevField<decltype(EcDev::B), offsetof(EcDev, B)> B;
    DevField<decltype(EcDev::P), offsetof(EcDev, P)> P;
    DevField<decltype(EcDev::dt), offsetof(EcDev, dt)> dt;
    DevField<decltype(EcDev::inv_dt), offsetof(EcDev, inv_dt)> inv_dt;
    DevField<decltype(EcDev::deadline), offsetof(EcDev, deadline)> deadline;
    DevField<decltype(EcDev::size_lo), offsetof(EcDev, size_lo)> size_lo;
    DevField<decltype(EcDev::size_range), offsetof(EcDev, size_range)> size_range;
    DevField<decltype(EcDev::cx_lo), offsetof(EcDev, cx_lo)> cx_lo;
    DevField<decltype(EcDev::cx_range), offsetof(EcDev, cx_range)> cx_range;
    DevField<decltype(EcDev::pen_drop), offsetof(EcDev, pen_drop)> pen_drop;
    DevField<decltype(EcDev::pen_fail), offsetof(EcDev, pen_fail)> pen_fail;
    DevField<decltype(EcDev::rew_succ), offsetof(EcDev, rew_succ)> rew_succ;
    DevField<decltype(EcDev::pen_hop), offsetof(EcDev, pen_hop)> pen_hop;
    DevField<decltype(EcDev::obs_size_norm), offsetof(EcDev, obs_size_norm)> obs_size_norm;
    DevField<decltype(EcDev::obs_cyc_norm), offsetof(EcDev, obs_cyc_norm)> obs_cyc_norm;
    DevField<decltype(EcDev::acc), offsetof(EcDev, acc)> acc;
    DevField<decltype(EcDev::cap), offsetof(EcDev, cap)> cap;
    DevField<decltype(EcDev::thr_exec_m), offsetof(EcDev, thr_exec_m)> thr_exec_m;
    DevField<decltype(EcDev::hot_thr), offsetof(EcDev, hot_thr)> hot_thr;
    DevField<decltype(EcDev::qlen), offsetof(EcDev, qlen)> qlen;
    DevField<decltype(EcDev::qthr), offsetof(EcDev, qthr)> qthr;
    DevField<decltype(EcDev::obs_const), offsetof(EcDev, obs_const)> obs_const;
    DevField<decltype(EcDev::bin_qn), offsetof(EcDev, bin_qn)> bin_qn;
    DevField<decltype(EcDev::bin_px2), offsetof(EcDev, bin_px2)> bin_px2;
    DevField<decltype(EcDev::bin_flip), offsetof(EcDev, bin_flip)> bin_flip;
    DevField<decltype(EcDev::deg), offsetof(EcDev, deg)> deg;
    DevField<decltype(EcDev::in_src), offsetof(EcDev, in_src)> in_src;
    DevField<decltype(EcDev::thr_link_m), offsetof(EcDev, thr_link_m)> thr_link_m;
    DevField<decltype(EcDev::slot_of), offsetof(EcDev, slot_of)> slot_of;
    DevField<decltype(EcDev::rate), offsetof(EcDev, rate)> rate;
    DevField<decltype(EcDev::adj), offsetof(EcDev, adj)> adj;
    DevField<decltype(EcDev::adj_bits), offsetof(EcDev, adj_bits)> adj_bits;
    DevField<decltype(EcDev::efr), offsetof(EcDev, efr)> efr;
    DevField<decltype(EcDev::tfr), offsetof(EcDev, tfr)> tfr;
    DevField<decltype(EcDev::cores), offsetof(EcDev, cores)> cores;
    DevField<decltype(EcDev::mt), offsetof(EcDev, mt)> mt;
    DevField<decltype(EcDev::rng_pos), offsetof(EcDev, rng_pos)> rng_pos;
    DevField<decltype(EcDev::rng_ahead), offsetof(EcDev, rng_ahead)> rng_ahead;
    DevField<decltype(EcDev::rng_cnt), offsetof(EcDev, rng_cnt)> rng_cnt;
    DevField<decltype(EcDev::has_new), offsetof(EcDev, has_new)> has_new;
    DevField<decltype(EcDev::newt), offsetof(EcDev, newt)> newt;
    DevField<decltype(EcDev::buf_head), offsetof(EcDev, buf_head)> buf_head;
    DevField<decltype(EcDev::buf_len), offsetof(EcDev, buf_len)> buf_len;
    DevField<decltype(EcDev::buf), offsetof(EcDev, buf)> buf;
    DevField<decltype(EcDev::wq_len), offsetof(EcDev, wq_len)> wq_len;
    DevField<decltype(EcDev::wqm), offsetof(EcDev, wqm)> wqm;
    DevField<decltype(EcDev::wqe), offsetof(EcDev, wqe)> wqe;
    DevField<decltype(EcDev::ex_n), offsetof(EcDev, ex_n)> ex_n;
    DevField<decltype(EcDev::exA), offsetof(EcDev, exA)> exA;
    DevField<decltype(EcDev::exB), offsetof(EcDev, exB)> exB;
    DevField<decltype(EcDev::exC), offsetof(EcDev, exC)> exC;
    DevField<decltype(EcDev::rq_head), offsetof(EcDev, rq_head)> rq_head;
    DevField<decltype(EcDev::rq_len), offsetof(EcDev, rq_len)> rq_len;
    DevField<decltype(EcDev::rq_arrive), offsetof(EcDev, rq_arrive)> rq_arrive;
    DevField<decltype(EcDev::rq_hstart), offsetof(EcDev, rq_hstart)> rq_hstart;
    DevField<decltype(EcDev::rq), offsetof(EcDev, rq)> rq;
    DevField<decltype(EcDev::n_off), offsetof(EcDev, n_off)> n_off;
    DevField<decltype(EcDev::cur_step), offsetof(EcDev, cur_step)> cur_step;
    DevField<decltype(EcDev::tick), offsetof(EcDev, tick)> tick;
    DevField<decltype(EcDev::total_steps), offsetof(EcDev, total_steps)> total_steps;
    DevField<decltype(EcDev::hop_hist), offsetof(EcDev, hop_hist)> hop_hist;
    DevField<decltype(EcDev::status), offsetof(EcDev, status)> status;
    DevField<decltype(EcDev::stamps), offsetof(EcDev, stamps)> stamps;
    __device__ __forceinline__ explicit EcDevL(const u32 (&v_)[DEV_VREGS])
        : N{v_}, A{v_}, k{v_}, D{v_}, S{v_}, T{v_}, H{v_}, capB{v_},
          capR{v_}, capW{v_}, maxTicks{v_}, epb{v_}, epw{v_}, nmain{v_}, dl_ticks{v_}, divN{v_},
          divA{v_}, fast_dl{v_}, acc_words{v_}, link_lds{v_}, B{v_}, P{v_}, dt{v_}, inv_dt{v_}, deadline{v_},
          size_lo{v_}, size_range{v_}, cx_lo{v_}, cx_range{v_}, pen_drop{v_}, pen_fail{v_}, rew_succ{v_}, pen_hop{v_},
          obs_size_norm{v_}, obs_cyc_norm{v_}, acc{v_}, cap{v_}, thr_exec_m{v_}, hot_thr{v_}, qlen{v_}, qthr{v_},
          obs_const{v_}, bin_qn{v_}, bin_px2{v_}, bin_flip{v_}, deg{v_}, in_src{v_}, thr_link_m{v_}, slot_of{v_},
          rate{v_}, adj{v_}, adj_bits{v_}, efr{v_}, tfr{v_}, cores{v_}, mt{v_}, rng_pos{v_},
          rng_ahead{v_}, rng_cnt{v_}, has_new{v_}, newt{v_}, buf_head{v_}, buf_len{v_}, buf{v_}, wq_len{v_},
          wqm{v_}, wqe{v_}, ex_n{v_}, exA{v_}, exB{v_}, exC{v_}, rq_head{v_}, rq_len{v_},
          rq_arrive{v_}, rq_hstart{v_}, rq{v_}, n_off{v_}, cur_step{v_}, tick{v_}, total_steps{v_}, hop_hist{v_},
          status{v_}, stamps{v_} {}
};
// lane image of *mem for this wave (every wave of the block holds its own)
__device__ __forceinline__ void dev_lanes_load(u32 (&v)[DEV_VREGS], const EcDev* mem) {
    const int l = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < DEV_VREGS; j++) v[j] = 64 * j + l < DEV_WORDS ? reinterpret_cast<const u32*>(mem)[64 * j + l] : 0u;
}

// ------------------------------------------------------------------------------------------ helpers
__device__ __forceinline__ double dadd(double a, double b) { return __dadd_rn(a, b); }
__device__ __forceinline__ double dmul(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ double ddiv(double a, double b) { return __ddiv_rn(a, b); }

template <class DEV>
__device__ __forceinline__ void flag(const DEV& d, u32 bit) { atomicOr(d.status, bit); }

// tick LUT.  A reference timer is dt added n times in fp64 (edge_node.py:216,250,254,316,319), which is not n * dt:
// the roundings accumulate.  The LUT stores, per n, the (small) distance in units in the last place between the two:
// 2 bytes per entry, so the whole table of an episode sits in LDS (2 KB for 1024 entries) and a lookup is one LDS
// read, one fp64 multiply and one integer add -- no global memory and no LDS-or-global divergence inside the step.
// Longer episodes take more LDS (d.acc_words; every lookup site used to carry an LDS-or-global diamond, ~200
// instructions of the mini-slot loop).  An index beyond the LUT is clamped and remembered in `ovf` (reported through
// the status word by the caller).
typedef const int16_t __attribute__((address_space(3))) * lds_ci16_ptr;
// (maxTicks is copied in: of the geometry members the lookups use it is the one that pays for a register of its own
// inside the mini-slot loop; dt and the rest are read where they are used, see EcDevL)
template <class DEV>
struct Lut {
    const DEV& d;
    u32 maxTicks;
#define dt ((double)d.dt)
#define inv_dt ((double)d.inv_dt)
#define deadline ((double)d.deadline)
#define dl_ticks ((int)d.dl_ticks)
#define fast_dl ((int)d.fast_dl)
    lds_ci16_ptr s_acc;  // explicitly LDS
    mutable u32 ovf;
    __device__ __forceinline__ double make(u32 n, int corr) const {
        return __longlong_as_double(__double_as_longlong(dmul((double)n, dt)) + (long long)corr);
    }
    __device__ __forceinline__ double at(u32 n) const {
        ovf |= n > maxTicks ? 1u : 0u;
        n = n > maxTicks ? maxTicks : n;
        return make(n, s_acc[n]);  // (the whole table is in LDS: d.acc_words)
    }
    // acc[n .. n+3] (indices clamped to the LUT) with the four loads in flight together: the searches below
    // advance four candidates per round trip instead of one
    __device__ __forceinline__ void peek4(u32 n, double (&a)[4]) const {
        const u32 mx = maxTicks;
        u32 i[4];
        int c[4];
#pragma unroll
        for (int j = 0; j < 4; j++) i[j] = n + j > mx ? mx : n + j;
#pragma unroll
        for (int j = 0; j < 4; j++) c[j] = s_acc[i[j]];
#pragma unroll
        for (int j = 0; j < 4; j++) a[j] = make(i[j], c[j]);
    }
    // smallest n with acc[n] >= x, T16_NEVER if none in the LUT
    __device__ __forceinline__ u32 ticks_to_reach(double x) const {
        // starting point only.  acc[n] is n * dt up to accumulated rounding, so the answer is floor(est) or
        // floor(est) + 1 and any start at or below it gives the same result (acc is non-decreasing)
        const double est = x * inv_dt;
        u32 n = 0;
        if (est > 2.0) n = est >= (double)maxTicks ? maxTicks : (u32)est - 2u;
        while (n <= maxTicks) {
            double a[4];
            peek4(n, a);  // acc is non-decreasing: the first candidate to reach x is the count of those below it
            const u32 below = (a[0] < x ? 1u : 0u) + (a[1] < x ? 1u : 0u) + (a[2] < x ? 1u : 0u) + (a[3] < x ? 1u : 0u);
            n += below;
            if (below < 4u) break;
        }
        return n > maxTicks ? T16_NEVER : n;
    }
    // The reference's task_time (edge_node.py:255,301): ((((ewt + cet) + twt) + ctt) + bwt) in fp64.
    __device__ __forceinline__ double task_time(u32 ewt, u32 cet, u32 twt, u32 ctt, u32 bwt) const {
        return dadd(dadd(dadd(dadd(at(ewt), at(cet)), at(twt)), at(ctt)), at(bwt));
    }
    // smallest n >= 1 such that task_time with the free timer at n ticks exceeds the deadline, and that
    // task_time.  which = 0: free timer is ewt (cet = 0);  which = 1: free timer is cet (ewt fixed).
    //
    // Closed form (fast_dl): every timer is its tick count times dt up to rounding, so a task_time of K ticks in all
    // is K * dt up to a bound E the host has evaluated for this LUT (build()): it cannot exceed the deadline for
    // K <= dl_ticks - 1 and must exceed it for K >= dl_ticks + 2, whatever the split over the five timers.  Only the
    // free-timer values with K = dl_ticks or dl_ticks + 1 need the fp64 sum: no search loop.
    __device__ __forceinline__ u32 deadline_ticks(int which, u32 ewt, u32 twt, u32 ctt, u32 bwt, double* tt_out) const {
        const u32 others = twt + ctt + bwt + (which ? ewt : 0u);
        const double tail0 = at(twt), tail1 = at(ctt), tail2 = at(bwt), fixed = which ? at(ewt) : 0.0;
        if (fast_dl) {
            const int nlo = dl_ticks - (int)others;
            const u32 c0 = nlo < 1 ? 1u : (u32)nlo;  // (c0 + 2 <= dl_ticks + 2 <= maxTicks: checked by the host)
            double tt[3];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const double a = at(c0 + j);
                const double head = which ? dadd(fixed, a) : dadd(a, 0.0);
                tt[j] = dadd(dadd(dadd(head, tail0), tail1), tail2);
            }
            const bool h0 = tt[0] > deadline, h1 = tt[1] > deadline;
            *tt_out = h0 ? tt[0] : (h1 ? tt[1] : tt[2]);
            return h0 ? c0 : (h1 ? c0 + 1 : c0 + 2);
        }
        const int nominal = dl_ticks - (int)others - 1;
        u32 n = nominal < 1 ? 1u : (u32)nominal;
        while (n <= maxTicks) {
            double a[4], tt[4];
            peek4(n, a);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const double head = which ? dadd(fixed, a[j]) : dadd(a[j], 0.0);
                tt[j] = dadd(dadd(dadd(head, tail0), tail1), tail2);
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (n + j <= maxTicks && tt[j] > deadline) {
                    *tt_out = tt[j];
                    return n + j;
                }
            }
            n += 4;
        }
        *tt_out = 0.0;
        return T16_NEVER;
    }
};
#undef dt
#undef inv_dt
#undef deadline
#undef dl_ticks
#undef fast_dl
template <class DEV>
__device__ __forceinline__ Lut<DEV> make_lut(const DEV& d, lds_ci16_ptr s_acc) {
    return Lut<DEV>{d, (u32)d.maxTicks, s_acc, 0u};
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

__device__ __forceinline__ u32 mt_temper(u32 y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}
__device__ __forceinline__ u32 mt_next(u32 cur, u32 nxt, u32 far) {
    const u32 y = (cur & 0x80000000u) | (nxt & 0x7fffffffu);
    return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __forceinline__ u32 mt_wrap(u32 p) { return p >= (u32)MT_N ? p - MT_N : p; }
// numpy random_sample: (a * 2^26 + b) / 2^53 with a = w0 >> 5, b = w1 >> 6 — every step exact in fp64
__device__ __forceinline__ double mt_to_double(u32 w0, u32 w1) {
    const u32 a = mt_temper(w0) >> 5, b = mt_temper(w1) >> 6;
    return __fma_rn((double)a, 1.0 / 134217728.0, (double)b * (1.0 / 9007199254740992.0));
}

// MT19937 stream of one lane (numpy RandomState / mt19937_gen).  Positions are regenerated in place and in
// order, `ahead` words beyond the consumption cursor `pos`: a word at position q needs the not-yet-regenerated
// word q+1 and the current word (q+397) mod 624, which in-order regeneration provides for any batch shorter
// than 227 words.  Only 53-bit doubles (two words) are drawn, so pos stays even.
//
// Memory form: MT_STRIDE words per stream = the 624 state words + a mirror of the first MT_MIRROR words, so that
// a window of already regenerated words can be read with wide loads without wrapping.  The step kernel keeps
// `ahead` >= a whole window at kernel entry: regeneration happens at the END of a step, 64 words at a time, by
// the whole wave for one stream (coalesced), and the entry path only reads and tempers.
__device__ __forceinline__ void mt_store_pair(u32* mt, u32 q, uint2 n) {
    *reinterpret_cast<uint2*>(mt + q) = n;
    if (q < (u32)MT_MIRROR) *reinterpret_cast<uint2*>(mt + q + MT_N) = n;
}

// One pair (per-lane slow path): returns the (regenerated) words at positions q, q+1.  `fresh`: the pair was
// regenerated by an earlier batch and only needs to be read.
struct MtPairIn {
    uint2 cur;
    u32 nxt, f0, f1;
};
__device__ __forceinline__ MtPairIn mt_pair_load(const u32* mt, u32 q, bool fresh) {
    MtPairIn in;
    in.cur = *reinterpret_cast<const uint2*>(mt + q);
    in.nxt = in.f0 = in.f1 = 0;
    if (!fresh) {
        in.nxt = mt[mt_wrap(q + 2)];
        in.f0 = mt[mt_wrap(q + MT_M)];
        in.f1 = mt[mt_wrap(q + MT_M + 1)];
    }
    return in;
}
__device__ __forceinline__ uint2 mt_pair_finish(u32* mt, u32 q, bool fresh, const MtPairIn& in) {
    if (fresh) return in.cur;
    const uint2 n = make_uint2(mt_next(in.cur.x, in.cur.y, in.f0), mt_next(in.cur.y, in.nxt, in.f1));
    mt_store_pair(mt, q, n);
    return n;
}

// Wave-cooperative regeneration: while some lane of the wave has fewer than `want` words regenerated ahead of
// its cursor, the 64 lanes regenerate the next 64 words of that lane's stream (one word each: three coalesced
// loads, one coalesced store).  Must be called by all 64 lanes of the wave.  mt_of_lane: this lane's stream.
__device__ __forceinline__ void mt_coop_refill(u32* __restrict__ mt_all, u64 stream_of_lane, u32 pos, u32& ahead,
                                               u32 want, bool active) {
    const u32 lid = __lane_id();
    unsigned long long need = __ballot(active && ahead < want);
    // Up to MT_COOP streams (of different lanes) per round: their loads are in flight together, so a round costs one
    // memory round trip whatever it serves -- the producer wave's regeneration is a chain of such rounds (eleven a step
    // at one stream per round: the 6.9 us it took from the end of the offload phase were nearly all round trips).
    constexpr int MT_COOP = 4;
    while (need) {
        int o[MT_COOP];
        unsigned long long rem = need;
#pragma unroll
        for (int u = 0; u < MT_COOP; u++) {
            o[u] = rem ? __ffsll((long long)rem) - 1 : -1;
            rem &= rem - 1;
        }
        u32 q[MT_COOP], cur[MT_COOP], nxt[MT_COOP], far[MT_COOP];
        u32* omt[MT_COOP];
#pragma unroll
        for (int u = 0; u < MT_COOP; u++) {
            q[u] = cur[u] = nxt[u] = far[u] = 0;
            omt[u] = mt_all;
            if (o[u] >= 0) {  // (wave-uniform)
                const u64 ostream = __shfl((unsigned long long)stream_of_lane, o[u]);
                const u32 opos = __shfl(pos, o[u]), oahead = __shfl(ahead, o[u]);
                omt[u] = mt_all + ostream * MT_STRIDE;
                q[u] = mt_wrap(mt_wrap(opos + oahead) + lid);
                cur[u] = omt[u][q[u]];
                nxt[u] = omt[u][mt_wrap(q[u] + 1)];
                far[u] = omt[u][mt_wrap(q[u] + MT_M)];
            }
        }
#pragma unroll
        for (int u = 0; u < MT_COOP; u++) {
            if (o[u] >= 0) {
                const u32 n = mt_next(cur[u], nxt[u], far[u]);
                omt[u][q[u]] = n;
                if (q[u] < (u32)MT_MIRROR) omt[u][q[u] + MT_N] = n;
                if ((int)lid == o[u]) ahead += 64;
            }
        }
        need = __ballot(active && ahead < want);
    }
}

// Failure stream (edge_node.py:258,322): the tests are `random() > threshold` with random() =
// (a*2^26 + b) / 2^53, i.e. the integer test  a*2^26 + b > M  with M = floor(threshold * 2^53)  (exact: see
// DESIGN.md).  With A = M >> 26 that is  a > A || (a == A && b > (M & (2^26-1))), so the window keeps only the
// 27-bit `a` of each draw (LDS, [draw][stride]); the second word is read back from the regenerated state in
// the 2^-27 tie case.
struct RngFail {
    u32* mt;
    u32 pos, ahead;  // cursor of the window start; words regenerated beyond it
    u32 idx;         // draws consumed from the window
    u32 staged;      // draws in the window
    u32 used;        // draws consumed in windows already retired
    u32* buf;
    int stride;

    // entry path: the window's words are already regenerated (ahead >= 2n): wide reads + temper only
    __device__ __forceinline__ void stage_fresh(u32 n) {
        const u32* w = mt + pos;  // no wrap thanks to the mirror
        for (u32 i0 = 0; i0 < n; i0 += 8) {
            u32x4 v[4];
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (i0 + 2 * j < n) v[j] = *reinterpret_cast<const u32x4*>(w + 2 * (i0 + 2 * j));
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const u32 i = i0 + 2 * j;
                if (i < n) {
                    buf[i * stride] = mt_temper(v[j].x) >> 5;
                    buf[(i + 1) * stride] = mt_temper(v[j].z) >> 5;  // slot i+1 < RB_FAIL + 1 (window has a spare slot)
                }
            }
        }
        staged = n;
        idx = 0;
    }
    // slow path: stage draws [0, n), regenerating per lane whatever is not yet regenerated
    __device__ __forceinline__ void stage(u32 n) {
        for (u32 i0 = 0; i0 < n; i0 += 4) {
            MtPairIn in[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const u32 i = i0 + j;
                if (i < n) in[j] = mt_pair_load(mt, mt_wrap(pos + 2 * i), 2 * i < ahead);
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const u32 i = i0 + j;
                if (i < n) {
                    const uint2 w = mt_pair_finish(mt, mt_wrap(pos + 2 * i), 2 * i < ahead, in[j]);
                    buf[i * stride] = mt_temper(w.x) >> 5;
                }
            }
        }
        ahead = ahead > 2 * n ? ahead : 2 * n;
        staged = n;
        idx = 0;
    }
    __device__ __forceinline__ void open(u32* mt_, u32 pos_, u32 ahead_, u32* buf_, int stride_, u32 n) {
        mt = mt_;
        pos = pos_;
        ahead = ahead_;
        buf = buf_;
        stride = stride_;
        used = 0;
        if (ahead >= 2 * n) stage_fresh(n);
        else stage(n);
    }
    // consumer side of a window another lane has staged
    __device__ __forceinline__ void attach(u32* mt_, u32 pos_, u32 ahead_, u32* buf_, int stride_, u32 n) {
        mt = mt_;
        pos = pos_;
        ahead = ahead_;
        buf = buf_;
        stride = stride_;
        used = 0;
        staged = n;
        idx = 0;
    }
    // slide the window so that it starts at the next unconsumed draw (only when a lane's per-step bound exceeds
    // the window capacity)
    __device__ __forceinline__ void restage(u32 n) {
        pos = mt_wrap(pos + 2 * idx);
        ahead -= 2 * idx;
        used += idx;
        stage(n);
    }
    // bit i set: window draw i (i < n <= 64) reaches the first-word threshold `thr` (read back from the window)
    __device__ __forceinline__ u64 hot_mask(u32 n, u32 thr) const {
        u64 m = 0;
        for (u32 i = 0; i < n; i++) m |= (u64)(buf[i * stride] >= thr ? 1u : 0u) << i;
        return m;
    }
    // second word of window draw i, as numpy's b = word >> 6
    __device__ __forceinline__ u32 b_of(u32 i) const { return mt_temper(mt[mt_wrap(pos + 2 * i + 1)]) >> 6; }
    __device__ __forceinline__ void close(u32& pos_out, u32& ahead_out, u32& draws) const {
        pos_out = mt_wrap(pos + 2 * idx);
        ahead_out = ahead - 2 * idx;
        draws = used + idx;
    }
};

// Task stream (edge_node.py:161, task.py:10-12): at most 4 doubles per step (binomial, id, size, cycles) except
// for the 2^-53 binomial redraw; the four are produced up front into registers.
struct RngTask {
    u32* mt;
    u32 pos, ahead;
    u32 idx, used;
    double u0, u1, u2, u3;
    __device__ __forceinline__ void stage_fresh() {
        static_assert(RB_TASK == 4, "two wide loads");
        const u32x4 v0 = *reinterpret_cast<const u32x4*>(mt + pos);
        const u32x4 v1 = *reinterpret_cast<const u32x4*>(mt + pos + 4);
        u0 = mt_to_double(v0.x, v0.y);
        u1 = mt_to_double(v0.z, v0.w);
        u2 = mt_to_double(v1.x, v1.y);
        u3 = mt_to_double(v1.z, v1.w);
        idx = 0;
    }
    __device__ __forceinline__ void stage() {
        MtPairIn i0 = mt_pair_load(mt, mt_wrap(pos + 0), 0u < ahead);
        MtPairIn i1 = mt_pair_load(mt, mt_wrap(pos + 2), 2u < ahead);
        MtPairIn i2 = mt_pair_load(mt, mt_wrap(pos + 4), 4u < ahead);
        MtPairIn i3 = mt_pair_load(mt, mt_wrap(pos + 6), 6u < ahead);
        const uint2 w0 = mt_pair_finish(mt, mt_wrap(pos + 0), 0u < ahead, i0);
        const uint2 w1 = mt_pair_finish(mt, mt_wrap(pos + 2), 2u < ahead, i1);
        const uint2 w2 = mt_pair_finish(mt, mt_wrap(pos + 4), 4u < ahead, i2);
        const uint2 w3 = mt_pair_finish(mt, mt_wrap(pos + 6), 6u < ahead, i3);
        u0 = mt_to_double(w0.x, w0.y);
        u1 = mt_to_double(w1.x, w1.y);
        u2 = mt_to_double(w2.x, w2.y);
        u3 = mt_to_double(w3.x, w3.y);
        ahead = ahead > 2u * RB_TASK ? ahead : 2u * RB_TASK;
        idx = 0;
    }
    __device__ __forceinline__ void open(u32* mt_, u32 pos_, u32 ahead_) {
        mt = mt_;
        pos = pos_;
        ahead = ahead_;
        used = 0;
        if (ahead >= 2u * RB_TASK) stage_fresh();
        else stage();
    }
    __device__ __forceinline__ void attach(u32* mt_, u32 pos_, u32 ahead_, double a, double b, double c, double e) {
        mt = mt_;
        pos = pos_;
        ahead = ahead_;
        used = 0;
        idx = 0;
        u0 = a;
        u1 = b;
        u2 = c;
        u3 = e;
    }
    __device__ __forceinline__ double next() {
        if (__builtin_expect(idx == (u32)RB_TASK, 0)) {  // only after a binomial redraw
            pos = mt_wrap(pos + 2 * idx);
            ahead -= 2 * idx;
            used += idx;
            stage();
        }
        const double r = idx == 0 ? u0 : idx == 1 ? u1 : idx == 2 ? u2 : u3;
        idx++;
        return r;
    }
    __device__ __forceinline__ void close(u32& pos_out, u32& ahead_out, u32& draws) const {
        pos_out = mt_wrap(pos + 2 * idx);
        ahead_out = ahead - 2 * idx;
        draws = used + idx;
    }
};

// Draw-at-consumption form of the same generator (ahead must be 0): used by the constructor pass only.
struct RngDirect {
    u32* mt;
    u32 pos;
    u32 used;
    __device__ __forceinline__ double next() {
        const MtPairIn in = mt_pair_load(mt, pos, false);
        const uint2 n = mt_pair_finish(mt, pos, false, in);
        pos = mt_wrap(pos + 2);
        used += 1;
        return mt_to_double(n.x, n.y);
    }
};

__device__ __forceinline__ u64 pack4(u32 a, u32 b, u32 c, u32 e) {
    return (u64)(a & 0xffffu) | ((u64)(b & 0xffffu) << 16) | ((u64)(c & 0xffffu) << 32) | ((u64)(e & 0xffffu) << 48);
}
__device__ __forceinline__ u32 f16(u64 w, int i) { return (u32)((w >> (16 * i)) & 0xffffu); }

__device__ __forceinline__ u32 sat16(u32 v, u32& ovf) {
    ovf |= v > 0xfffeu ? 1u : 0u;
    return v > 0xfffeu ? 0xfffeu : v;
}

// n / dv.d for n below 2^32 / dv.d (m = ceil(2^32 / d)); d == 1 is the identity
__device__ __forceinline__ u32 fdiv(u32 n, FastDiv dv) { return dv.d == 1 ? n : __umulhi(n, dv.m); }

// buffer_queue of one lane.  generate_task pops whenever the queue is non-empty (edge_node.py:165-166), i.e. the
// head is consumed at the end of EVERY step: `head_rec` is that record, prefetched at kernel entry or captured
// when a push finds the queue empty (such a record never needs to reach the global ring).
struct Buffer {
    u32 head, len;
    Rec head_rec;
    template <class DEV>
    __device__ __forceinline__ void push(const DEV& d, int64_t lane, u32 n_off, Rec rec) {
        if (__builtin_expect(len >= (u32)d.capB, 0)) {
            flag(d, EC_ST_BUFFER_OVERFLOW);
            return;
        }
        rec.bwt = rec.bwt - (int32_t)n_off;  // edge_node.py:214-216 applied lazily
        if (len == 0) {
            head_rec = rec;
        } else {
            u32 pos = head + len;
            if (pos >= (u32)d.capB) pos -= d.capB;
            store_rec(d.buf + (size_t)pos * d.P + lane, rec);
        }
        len++;
    }
    template <class DEV>
    __device__ __forceinline__ Rec pop(const DEV& d, u32 n_off) {
        Rec r = head_rec;
        r.bwt = r.bwt + (int32_t)n_off;
        head = head + 1 == (u32)d.capB ? 0u : head + 1;
        len--;
        return r;
    }
};

// edge_node.py:159-166 + task.py:6-33
template <class DEV, typename RNG>
__device__ __forceinline__ void generate_task(const DEV& d, int64_t lane, double qn, double px2, u32 flip,
                                              RNG& trng, Buffer& buf, u32 n_off, u32& has_new, Rec& newt) {
    has_new = 0;
    // numpy legacy binomial(1, p): inversion with bound = 1 (see oracle/ec_oracle.c binomial_inversion)
    u32 X;
    for (;;) {
        double U = trng.next();
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
    const u32 arrival = flip ? 1u - X : X;
    if (arrival) {
        (void)trng.next();  // task.id
        Rec t;
        t.size = dmul(dmul(dadd(d.size_lo, dmul(d.size_range, trng.next())), 1024.0), 8.0);
        t.cycles = dmul(dadd(d.cx_lo, dmul(d.cx_range, trng.next())), t.size);
        t.twt = t.ctt = t.hop = t.ntx = 0;
        t.bwt = 0;
        t.enq = 0;
        buf.push(d, lane, n_off, t);
    }
    if (buf.len > 0) {
        newt = buf.pop(d, n_off);
        has_new = 1;
    }
}

struct LaneId {
    int64_t env, lane;
    int node, env_local;
    int lidx;  // env_local * N + node: index of the lane in the block's per-lane LDS arrays
    bool valid;
};
// lanes packed contiguously over the block's threads, starting at thread t0
template <class DEV>
__device__ __forceinline__ LaneId lane_id(const DEV& d, int t0 = 0) {
    LaneId L;
    const u32 t = threadIdx.x - t0;
    L.env_local = (int)fdiv(t, d.divN);
    L.node = t - L.env_local * d.N;
    L.lidx = (int)t;
    L.env = (int64_t)blockIdx.x * d.epb + L.env_local;
    L.valid = (int)threadIdx.x >= t0 && L.env_local < d.epb && L.env < d.B;
    L.lane = L.env * d.N + L.node;
    return L;
}
// main lanes of k_step: whole envs per wave (d.epw of them) so that a block of several waves keeps the
// per-wave divergence of a one-wave block; contiguous when one env needs more than a wave
template <class DEV>
__device__ __forceinline__ LaneId lane_id_main(const DEV& d) {
    if (d.epw == 0) {
        LaneId L = lane_id(d);
        L.valid = L.valid && (int)threadIdx.x < d.nmain;
        return L;
    }
    LaneId L;
    const u32 w = threadIdx.x >> 6, l = threadIdx.x & 63u;
    const int el = (int)fdiv(l, d.divN);
    L.node = l - el * d.N;
    L.env_local = (int)w * d.epw + el;
    L.lidx = L.env_local * d.N + L.node;
    L.env = (int64_t)blockIdx.x * d.epb + L.env_local;
    L.valid = (int)threadIdx.x < d.nmain && el < d.epw && L.env_local < d.epb && L.env < d.B;
    L.lane = L.env * d.N + L.node;
    return L;
}

__device__ __forceinline__ u32 arrive_at(u32 hstart, u32 ntx, u32 ctt) {
    if (ntx == T16_NEVER) return NEVER;
    return hstart + (ntx > ctt + 1 ? ntx - ctt - 1 : 0u);
}

// ------------------------------------------------------------------------------------------ kernels
__global__ void k_seed(EcDev d, u32 task_seed0, u32 fail_seed0) {
    const int64_t lane = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= d.P) return;
    const int64_t env = lane / d.N;
    const int node = (int)(lane - env * d.N);
    for (int s = 0; s < 2; s++) {
        u32 seed = (s == 0 ? task_seed0 : fail_seed0) + (u32)env + (u32)node;
        u32* mt = d.mt + ((size_t)s * d.P + lane) * MT_STRIDE;
        for (int i = 0; i < MT_N; i++) {
            mt[i] = seed;
            if (i < MT_MIRROR) mt[i + MT_N] = seed;
            seed = 1812433253u * (seed ^ (seed >> 30)) + (u32)(i + 1);
        }
        d.rng_pos[(size_t)s * d.P + lane] = 0;
        d.rng_ahead[(size_t)s * d.P + lane] = 0;
        d.rng_cnt[(size_t)s * d.P + lane] = 0;
    }
    d.has_new[lane] = 0;
    d.buf_head[lane] = d.buf_len[lane] = 0;
    d.wq_len[lane] = 0;
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
        d.total_steps[env] = 0;
        for (int h = 0; h < d.H; h++) d.hop_hist[env * d.H + h] = 0;
    }
}

// the constructor's generate_task pass (edge_computing_env.py:17-18)
__global__ void k_initial_generate(EcDev d) {
    const int64_t lane = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= d.P) return;
    const int node = (int)(lane % d.N);
    RngDirect trng{d.mt + (size_t)lane * MT_STRIDE, d.rng_pos[lane], 0};
    Buffer buf;
    buf.head = d.buf_head[lane];
    buf.len = d.buf_len[lane];  // 0 at construction
    u32 hn;
    Rec nt;
    generate_task(d, lane, d.bin_qn[node], d.bin_px2[node], d.bin_flip[node], trng, buf, d.n_off[lane], hn, nt);
    d.rng_pos[lane] = trng.pos;
    d.rng_cnt[lane] += 2ull * trng.used;
    d.buf_head[lane] = buf.head;
    d.buf_len[lane] = buf.len;
    d.has_new[lane] = hn;
    if (hn) store_rec(d.newt + lane, nt);
}

// establishes the step kernel's entry invariant: a whole window regenerated ahead of both cursors
__global__ void k_prime_rng(EcDev d) {
    const int64_t lane = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = lane < d.P;
    const int64_t l = ok ? lane : 0;
    for (int s = 0; s < 2; s++) {
        u32 pos = d.rng_pos[(size_t)s * d.P + l], ahead = d.rng_ahead[(size_t)s * d.P + l];
        mt_coop_refill(d.mt, (u64)s * d.P + l, pos, ahead, s == 0 ? AHEAD_TASK : AHEAD_FAIL, ok);
        if (ok) d.rng_ahead[(size_t)s * d.P + l] = ahead;
    }
}

// edge_computing_env.py:145-148 / edge_node.py:171-177
__global__ void k_reset(EcDev d) {
    const int64_t lane = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= d.P) return;
    d.has_new[lane] = 0;
    d.buf_head[lane] = d.buf_len[lane] = 0;
    d.wq_len[lane] = 0;
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

// Observation + availability mask for the envs of this block (edge_node.py:109-136,179-197).
// Per lane in: wq_len, `wait_sum` = get_waiting_time() (fp64, queue order), has_new/newt.
// LDS scratch: s_full/s_hasnew/s_any [NT] (NT = active lanes), s_obs [NT*9], s_fullbits [epb] or null.
// With s_fullbits (requires d.adj_bits and every env's lanes consecutive in ONE wave, node 0 first) the "queue
// full" flags of an env travel as one ballot word and the second barrier + s_any pass are not needed.
template <class DEV>
__device__ __forceinline__ void observe_block(const DEV& d, const LaneId& L, u32 wq_len, double wait_sum, u32 has_new,
                                              const Rec& newt, float* __restrict__ obs, int32_t* __restrict__ avail,
                                              u32* s_full, u32* s_hasnew, u32* s_any, float* s_obs, u64* s_fullbits) {
    const int N = d.N, A = d.A;
    const int tid = threadIdx.x;  // cooperative copy loops
    const int li = L.lidx;        // this lane's slot in the per-lane arrays
    const int epb = d.epb;
    bool full = false;
    if (L.valid) {
        full = (double)wq_len >= d.qthr[L.node];  // edge_node.py:186-188
        s_full[li] = full ? 1u : 0u;
        s_hasnew[li] = has_new;
        if (obs) {
            float* o = s_obs + li * 9;
            o[0] = d.obs_const[L.node * 3 + 0];
            o[1] = d.obs_const[L.node * 3 + 1];
            o[2] = d.obs_const[L.node * 3 + 2];
            o[3] = (float)ddiv((double)wq_len, d.qlen[L.node]);
            o[4] = (float)ddiv(wait_sum, d.deadline);
            if (has_new) {
                o[5] = (float)ddiv(newt.size, d.obs_size_norm);
                o[6] = (float)ddiv(newt.cycles, d.obs_cyc_norm);
                o[7] = 1.0f;  // task_deadline / deadline (edge_node.py:130): the same number over itself
                o[8] = (float)newt.hop;
            } else {
                o[5] = o[6] = o[7] = -1.0f;
                o[8] = 0.0f;
            }
        }
    }
    if (s_fullbits) {
        const u64 ball = __ballot(L.valid && full);
        if (L.valid && L.node == 0)
            s_fullbits[L.env_local] = (ball >> (threadIdx.x & 63u)) & (N >= 64 ? ~0ull : (1ull << N) - 1ull);
    }
    __syncthreads();
    const int64_t env0 = (int64_t)blockIdx.x * epb;
    const int64_t nenv = (d.B - env0) < epb ? (d.B - env0) : epb;
    if (avail) {
        if (!s_fullbits) {
            if (L.valid) {
                // any adjacent (or own) queue with room?  adj rows are bit masks when N <= 64
                u32 any = 0;
                const u32* fl = s_full + L.env_local * N;
                if (d.adj_bits) {
                    const u64 row = d.adj_bits[L.node];
                    for (int j = 0; j < N; j++) any |= (((row >> j) & 1ull) && !fl[j]) ? 1u : 0u;
                } else {
                    for (int j = 0; j < N; j++) any |= (d.adj[L.node * A + j] && !fl[j]) ? 1u : 0u;
                }
                s_any[li] = any;
            }
            __syncthreads();
        }
        const int total = (int)nenv * N * A;
        int32_t* out = avail + env0 * N * A;
        for (int idx = tid; idx < total; idx += blockDim.x) {
            const int t = (int)fdiv((u32)idx, d.divA), j = idx - t * A;  // t = env_local*N + node
            const int el = (int)fdiv((u32)t, d.divN), i = t - el * N;
            int v;
            if (s_hasnew[t]) {
                if (s_fullbits) {
                    const u64 room = d.adj_bits[i] & ~s_fullbits[el];  // adjacent queues with room
                    v = (j < N) ? (int)((room >> j) & 1ull) | (int)(!room && j == i) : 0;
                } else {
                    const bool adjacent = d.adj_bits ? ((d.adj_bits[i] >> j) & 1ull) : (j < N && d.adj[i * A + j]);
                    v = (j < N) ? ((adjacent && !s_full[el * N + j]) || (!s_any[t] && j == i)) : 0;
                }
            } else {
                v = (j == N);
            }
            out[idx] = v;
        }
    }
    if (obs) {
        const int total = (int)nenv * N * 9;
        float* out = obs + env0 * N * 9;
        for (int idx = tid; idx < total; idx += blockDim.x) out[idx] = s_obs[idx];
    }
}

__global__ void k_observe(EcDev d, float* __restrict__ obs, int32_t* __restrict__ avail) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int NT = d.epb * d.N;  // active lanes
    u32* s_full = reinterpret_cast<u32*>(smem);
    u32* s_hasnew = s_full + NT;
    u32* s_any = s_hasnew + NT;
    float* s_obs = reinterpret_cast<float*>(s_any + NT);
    const LaneId L = lane_id(d);
    u32 wq_len = 0, has_new = 0;
    double w = 0.0;
    Rec newt;
    newt.size = newt.cycles = 0;
    newt.hop = 0;
    if (L.valid) {
        wq_len = d.wq_len[L.lane];
        has_new = d.has_new[L.lane];
        if (has_new) newt = load_rec(d.newt + L.lane);
        // get_waiting_time, edge_node.py:99-104 (current_execute_time is 0 for every waiting task)
        for (u32 j = 0; j < wq_len; j++) w = dadd(w, d.wqe[(size_t)j * d.P + L.lane]);
    }
    observe_block(d, L, wq_len, w, has_new, newt, obs, avail, s_full, s_hasnew, s_any, s_obs, nullptr);
}

// LDS carve-up of k_step; all per-lane arrays are [slot][NT], NT = active lanes of the block
struct StepLds {
    u64* acc;        // [ACC_LDS] words = the first ACC_LDS_N entries of the tick LUT
    uint4* wqm;      // [capW][NT]
    double* wqe;     // [capW][NT]
    u64* exC;        // [KMAX][NT] {ewt, twt, ctt, bwt} of the executing tasks (kept for ec_debug_dump)
    uint4* rqrec;    // [DMAX][2][NT] head record of each receiving queue
    u32* rqhs;       // [DMAX][NT]
    u32* rqhead;     // [DMAX][NT]
    u32* rqdirty;    // [DMAX][NT] head record exists only in LDS
    u32* rqlen;      // [DMAX][NT] staging / hand-off copy of the queue length
    u32* rqarr;      // [DMAX][NT] staging / hand-off copy of the head's arrival tick
    u32* fa;         // [RB_FAIL][NT] failure-stream window (aliased by the observation scratch afterwards)
    double* rew;     // [NT]
    double* ct;      // [NT]
    u32* cnt;        // [4][NT]
    double* tu;      // [RB_TASK][NT] the step's task-stream doubles (producer -> consumer)
    u64* fullbits;   // [NT] (one word per env used) "queue full" flags of an env's nodes, for the fused observation
    u64* hot;        // [NT] aliases fullbits (dead until phase E): bit i = draw i of the failure window is "hot", i.e.
                     //      its first word reaches d.hot_thr[node] and may fail a test; all other draws fail none
    u32* rc;         // [RC_N][NT] RNG cursors and the producer/consumer hand-off (RC_*)
    u32* nodec;      // [N][NODEC_W] per-node constants of phases D and E, filed by the launch's first step (looping kernels)
    double* linkr;   // [N][DMAX] rate of the link into node n from its s-th in-neighbour   } N <= LINK_LDS_N: the offload
    signed char* slotb;  // [N][N] slot of src in dst's in-list, -1 if not adjacent          } phase's look-ups, as nodec
};
constexpr int LINK_LDS_N = 16;
constexpr int NODEC_W = 16;  // bin_qn, bin_px2, qthr, qlen (doubles), adj_bits (u64), bin_flip (u32), obs_const (3 floats), deg, hot_thr
enum { RC_FPOS, RC_FAHEAD, RC_TPOS, RC_TAHEAD, RC_FDELTA, RC_TDELTA, RC_DONE, RC_NX, RC_STAGED, RC_FNEXT, RC_PFPOS, RC_PTPOS,
       RC_PSTAGED, RC_N };
__host__ __device__ inline size_t link_lds_bytes(int DMAX, int n_nodes) {
    return n_nodes <= LINK_LDS_N ? (size_t)n_nodes * DMAX * 8 + (((size_t)n_nodes * n_nodes + 7) & ~(size_t)7) : 0;
}
// n_nodes_links > 0: with the link tables (k_rollout_mlp's launches)
__host__ __device__ inline size_t step_lds_bytes(int NT, int KMAX, int DMAX, int capW, int acc_words, int n_nodes,
                                                 int n_nodes_links = 0) {
    size_t b = (size_t)n_nodes * NODEC_W * 4;
    b += sizeof(double) * acc_words;
    b += (size_t)NT * capW * (16 + 8);
    b += (size_t)NT * KMAX * 8;
    b += (size_t)NT * DMAX * (32 + 5 * 4);
    b += (size_t)NT * (RB_FAIL + 1) * 4;
    b += (size_t)NT * (8 + 8 + 16);
    b += (size_t)NT * (RB_TASK * 8 + RC_N * 4 + 8);
    b = (b + 7) & ~(size_t)7;
    if (n_nodes_links > 0) b += (link_lds_bytes(DMAX, n_nodes_links) + 15) & ~(size_t)15;
    return b + 64;
}

// executing slot in registers: exf = index of the execute_task call at which the task completes or drops (NEVER: the
// slot is unused), exh = hop | EXH_SUCCESS | fin << 16 (fin: the current_execute_time, in ticks, at that call)
constexpr u32 EXH_HOP = 0x7fffu;
constexpr u32 EXH_SUCCESS = 1u << 15;

// optional tail of the step: the uniform-random policy of ec_random_policy for the NEXT step, drawn from the
// availability mask the fused observation has just produced (same generator, same row numbering)
struct StepPolicy {
    u64 seed, counter;       // the draw uses counter + (steps this handle has taken, this one included)
    int64_t* next_actions;   // [B,N] or null (no policy)
    // ec_rollout_random_policy: the kernel takes n_steps steps, each fed by the actions its predecessor drew.
    // keep = 0: every step overwrites the same [B,...] outputs; keep = 1: outputs are [n_steps,B,...] histories.
    int n_steps, keep;
};

__device__ __forceinline__ u64 policy_hash(u64 seed, u64 counter, u64 rows, u64 r) {
    u64 z = seed + 0x9E3779B97F4A7C15ull * (counter * rows + r + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

#ifdef EC_STAMPS
#define EC_STAMP(i)                                                                           \
    do {  /* the timeline of wave 0; no barrier, which would serialise the producer wave */  \
        if (threadIdx.x == 0 && d.stamps) d.stamps[(size_t)blockIdx.x * 16 + (i)] = wall_clock64(); \
    } while (0)
#define EC_STAMP_P(i)                                                                         \
    do {  /* the producer wave's timeline */                                                  \
        if ((int)threadIdx.x == d.nmain && d.stamps) d.stamps[(size_t)blockIdx.x * 16 + (i)] = wall_clock64(); \
    } while (0)
// time of wave 0 between consecutive EC_SUB points of the mini-slot loop, accumulated per point
// (kept per wave in LDS by whichever lane is the first active one: lanes take different passes)
#define EC_SUB(i)                                                                   \
    do {                                                                            \
        if ((int)__lane_id() == __ffsll((long long)__ballot(1)) - 1) {              \
            const u64 now_ = wall_clock64();                                        \
            dbg_lds[threadIdx.x >> 6][i] += now_ - dbg_lds[threadIdx.x >> 6][11];   \
            dbg_lds[threadIdx.x >> 6][11] = now_;                                   \
        }                                                                           \
    } while (0)
#else
#define EC_SUB(i) \
    do {          \
    } while (0)
#define EC_STAMP(i) \
    do {            \
    } while (0)
#define EC_STAMP_P(i) \
    do {              \
    } while (0)
#endif

// EdgeComputingEnv.step (edge_computing_env.py:84-143) for the envs of this block.
// KMAX >= k and DMAX >= max in-degree are compile-time so that the executing slots and the receiving-queue
// scalars live in registers and the per-mini-slot work is branch-free except for events.
// What a main lane of the looping kernel keeps in registers from one step to the next (its queues stay in LDS):
// with it, the steps between the first and the last of a launch neither stage their state from global memory nor
// write it back.
// ------------------------------------------------------------------------------------------ expert policy
// OptimalAgent._calculate_conservative_score (optimal/optimal_agent.py:211-258) with its four terms (:260-375), in
// the reference's evaluation order (this file is built without FMA contraction).
template <class DEV>
__device__ __forceinline__ double expert_score(const DEV& d, int nwait, int nexec, int s, int t, double size, double cycles,
                                               double cons) {
    const double exec_t = cycles / d.cap[t];
    double rel = exp(-d.efr[t] * exec_t);  // :260-289
    double trans_t = 0.0;
    if (s != t) {
        trans_t = size / d.rate[s * d.N + t];
        rel = exp(-d.tfr[s * d.N + t] * trans_t) * rel;
    }
    if (rel < 0.85) rel = fmax(0.0, rel - 5.0 * (0.85 - rel));
    const double qf = 1.5 * cons;  // :291-328
    const double waiting = nwait * exec_t * qf / (d.k > 1 ? d.k : 1);
    const double total = (s == t) ? exec_t + waiting : trans_t + exec_t + waiting;
    const double margin = d.deadline - total;
    const double required = d.deadline * 0.4 * cons;
    double time_score = 0.0;
    if (!(margin < required)) {
        time_score = fmin(1.0, margin / d.deadline);
        if (time_score < 0.5) time_score *= 3.0;
    }
    const int cur = nwait + nexec;  // :330-353
    const double max_load = (d.qlen[t] + d.k) * 0.7;
    double load = 0.0;
    if (!(cur >= max_load)) {
        load = 1.0 - cur / max_load;
        if (cur > 0) load = fmax(0.0, load - cur * 0.2);
    }
    const double opt = d.dt * 3;  // :355-375
    double eff = exec_t <= opt ? 1.0 : fmax(0.2, opt / exec_t);
    eff += fmin(0.2, d.cores[t] / 32.0);
    eff = fmin(1.0, eff);
    if (rel < 0.85) return -INFINITY;  // :235-239
    if (time_score == 0) return -INFINITY;
    const double bonus = (s == t) ? 0.3 * cons : 0.0;
    return 0.6 * rel + 0.25 * time_score + 0.1 * load + 0.05 * eff + bonus;
}

// OptimalAgent.get_current_optimal_action -> _conservative_action_selection(env, 1.0) (optimal_agent.py:102-120,166-209)
// for node i holding a new task; a: the node's availability mask [A]; cnt(t, &nwait, &nexec): queue lengths of node t of
// the same env (global memory in k_expert, the block's LDS copy inside the rollout kernel)
// bits [64 w, 64 w + 64) of an availability row as a mask; the loads of a chunk are independent, so they overlap (a
// one-element-per-iteration loop pays one memory round trip per action: 41 of them at 40 nodes)
__device__ __forceinline__ u64 avail_word(const int32_t* __restrict__ a, int A, int w) {
    u64 m = 0;
    const int hi = A - 64 * w < 64 ? A - 64 * w : 64;
    for (int j0 = 0; j0 < hi; j0 += 8) {
        int32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = j0 + u < hi ? a[64 * w + j0 + u] : 0;
#pragma unroll
        for (int u = 0; u < 8; u++) m |= (u64)(v[u] != 0) << (j0 + u);
    }
    return m;
}

template <class DEV, typename CNT>
__device__ __forceinline__ int expert_decide(const DEV& d, int i, const int32_t* __restrict__ a, u32 has_new, double size,
                                             double cycles, CNT cnt) {
    const int nwords = (d.A + 63) >> 6;
    int last = -1;
    u64 m0 = 0;
    for (int w = 0; w < nwords; w++) {
        const u64 m = avail_word(a, d.A, w);
        if (w == 0) m0 = m;
        if (m) last = 64 * w + 63 - __clzll((long long)m);
    }
    if (last < 0 || !has_new) return last >= 0 ? last : d.N;
    int best = last;
    double best_score = -INFINITY;
    int nw, nx;
    if (i < 64 ? (m0 >> i) & 1ull : (avail_word(a, d.A, i >> 6) >> (i & 63)) & 1ull) {
        cnt(i, nw, nx);
        const double sc = expert_score(d, nw, nx, i, i, size, cycles, 1.0);
        if (sc > best_score) {
            best_score = sc;
            best = i;
        }
    }
    for (int w = 0; w < nwords; w++) {  // available_indices[:-1]: every available index but the last one, ascending
        u64 m = w == 0 ? m0 : avail_word(a, d.A, w);
        while (m) {
            const int j = 64 * w + __builtin_ctzll(m);
            m &= m - 1;
            if (j >= last) break;
            if (j >= d.N || j == i) continue;
            cnt(j, nw, nx);
            const double sc = expert_score(d, nw, nx, i, j, size, cycles, 1.0);
            if (sc > best_score + 0.1 * 1.0) {
                best_score = sc;
                best = j;
            }
        }
    }
    return best;
}


template <int KMAX, int DMAX>
struct StepCarry {
    u32 c0, steps_before, has_new, n_off, ex_n, wq_h, wq_len, wq_min, exA_thr;
    bool wq_dirty;
    int deg;
    int64_t action;  // drawn by the previous step's policy tail
    double cap;
    u32 lkA[DMAX];
    u32 exf[KMAX], exh[KMAX];
    double exT[KMAX];
    Rec newt;
    u32 buf_head, buf_len;
};

// Expert mixing inside the rollout kernel (ec_rollout_mlp_expert_policy; gail_runner.py:72-98, dgail_runner.py:46-94): per
// env a coin decides whether the step executes the expert's decision or the policy's; the expert's decision of every
// step is recorded (the FIFO append of the expert-driven envs happens after the launch, from the histories).
struct ExpertMix {
    const int32_t* avail;  // [P][A] availability mask of the observation the action is chosen on
    const float* coin;     // [B] this step's uniform draws
    float ratio;           // expert iff coin < ratio
    int64_t* expert_out;   // [P] the expert's decisions of this step
    u32* s_cnt;            // LDS [main lanes]: waiting | executing << 16 queue lengths of every lane of the block, as the
                           // previous step left them (the expert scores its neighbours' queues)
};

// first: the lane's state comes from global memory (else from `cr` and the LDS arrays as the previous step left
// them); last: it is written back.  One-step launches are both.
template <int KMAX, int DMAX, bool EVX, int NTC = 0, class DEV = EcDev>
__device__ __forceinline__ void step_body(const DEV& d, const int64_t* __restrict__ actions, float* __restrict__ reward,
                                          uint8_t* __restrict__ terminated, int32_t* __restrict__ counters,
                                          float* __restrict__ completion_time, float* __restrict__ next_obs,
                                          int32_t* __restrict__ next_avail, StepPolicy pol, StepCarry<KMAX, DMAX>& cr,
                                          const bool first, const bool last, const int* s_pick = nullptr,
                                          const ExpertMix* em = nullptr) {
    static_assert(RB_FAIL >= KMAX + DMAX, "failure window too small");
    static_assert(RB_FAIL * 4 >= 12 + 36, "observation scratch does not fit the failure window");
    extern __shared__ __align__(16) unsigned char smem[];
    // active lanes of the block = stride of the per-lane LDS arrays; NTC > 0: known at compile time (the launch geometry
    // of the reference's default size, 4 envs x 10 nodes per block: the offsets of the LDS accesses become immediates and
    // their loop-invariant scalars leave the SGPR file: 19.45 -> 18.8 us per env step)
    const int NT = NTC > 0 ? NTC : d.epb * d.N;
    const int N = d.N, D = d.D;
    const int64_t P = d.P;
    const u32 wmask = (u32)d.capW - 1;  // capW is a power of two

    StepLds S;
    {
        unsigned char* p = smem;
        S.wqm = reinterpret_cast<uint4*>(p);     p += (size_t)NT * d.capW * 16;
        S.rqrec = reinterpret_cast<uint4*>(p);   p += (size_t)NT * DMAX * 32;
        S.acc = reinterpret_cast<u64*>(p);       p += sizeof(u64) * d.acc_words;
        S.wqe = reinterpret_cast<double*>(p);    p += (size_t)NT * d.capW * 8;
        S.exC = reinterpret_cast<u64*>(p);       p += (size_t)NT * KMAX * 8;
        S.rew = reinterpret_cast<double*>(p);    p += (size_t)NT * 8;
        S.ct = reinterpret_cast<double*>(p);     p += (size_t)NT * 8;
        S.tu = reinterpret_cast<double*>(p);     p += (size_t)NT * RB_TASK * 8;
        S.fullbits = reinterpret_cast<u64*>(p);  p += (size_t)NT * 8;
        S.hot = S.fullbits;
        S.nodec = reinterpret_cast<u32*>(p);     p += (size_t)N * NODEC_W * 4;  // (8-byte aligned here: doubles inside)
        S.fa = reinterpret_cast<u32*>(p);        p += (size_t)NT * (RB_FAIL + 1) * 4;
        S.rqhs = reinterpret_cast<u32*>(p);      p += (size_t)NT * DMAX * 4;
        S.rqhead = reinterpret_cast<u32*>(p);    p += (size_t)NT * DMAX * 4;
        S.rqdirty = reinterpret_cast<u32*>(p);   p += (size_t)NT * DMAX * 4;
        S.rqlen = reinterpret_cast<u32*>(p);     p += (size_t)NT * DMAX * 4;
        S.rqarr = reinterpret_cast<u32*>(p);     p += (size_t)NT * DMAX * 4;
        S.cnt = reinterpret_cast<u32*>(p);       p += (size_t)NT * 16;
        S.rc = reinterpret_cast<u32*>(p);        p += (size_t)NT * RC_N * 4;
        // (the link tables last, 8-byte aligned: only k_rollout_mlp's launches have them)
        p = smem + (((size_t)(p - smem) + 7) & ~(size_t)7);
        S.linkr = reinterpret_cast<double*>(p);
        S.slotb = reinterpret_cast<signed char*>(p + (N <= LINK_LDS_N ? (size_t)N * DMAX * 8 : 0));
    }
    Lut<DEV> lut = make_lut(d, (lds_ci16_ptr)S.acc);

    // Two roles.  The main lanes (threads < d.nmain) run the env logic and only CONSUME random words.  The RNG
    // producer lanes regenerate the Mersenne-Twister states and stage the step's draws: a wave of their own
    // when the block was launched with more than d.nmain threads (its memory latency then overlaps the main
    // waves' dependent chain), else the main lanes themselves at the same program points.
    const bool has_producers = (int)blockDim.x > d.nmain;
    const bool obs_direct = d.epw > 0 && d.adj_bits != nullptr && next_avail != nullptr;  // (see phase E)
    const LaneId L = lane_id_main(d);
    const LaneId R = has_producers ? lane_id(d, d.nmain) : L;
    const bool producer_wave = has_producers ? (int)threadIdx.x >= d.nmain : true;  // wave-uniform
    const int tid = L.lidx;  // slot of this main lane in the per-lane LDS arrays
    u64 lut_lo[ACC_LDS / 64];  // block-shared LDS part of the tick LUT: loaded by the producer lanes with their
    if (producer_wave && first) { // first round of loads, stored to LDS before the first barrier
        const int p0 = has_producers ? (int)threadIdx.x - d.nmain : (int)threadIdx.x;
        const int np = has_producers ? (int)blockDim.x - d.nmain : (int)blockDim.x;
#pragma unroll
        for (int i = 0; i < ACC_LDS / 64; i++) {
            const int a = p0 + i * np;
            lut_lo[i] = a < ACC_LDS ? reinterpret_cast<const u64*>((const int16_t*)d.acc)[a] : 0ull;  // (the table is padded to ACC_LDS_N)
        }
    }
    const int64_t lane = L.lane;
    const int node = L.node;
    // the out-link of an action: slot of this node in the destination's in-list and the link's rate -- in carried steps of
    // the looping kernels from the LDS tables the first step filed (small topologies), else from global memory
    const bool link_lds = EVX && !first && (s_pick != nullptr || em != nullptr) && d.link_lds;
    const auto link_of = [&](int a, int& sl, double& rt) {
        if (link_lds) {
            sl = (int)S.slotb[node * N + a];
            rt = sl >= 0 ? S.linkr[a * DMAX + sl] : 0.0;
        } else {
            sl = d.slot_of[node * N + a];
            rt = d.rate[node * N + a];
        }
    };
    EC_STAMP(0);

    // ---------------------------------------------------------------- stage the lane's working set
    u32 c0 = 0, has_new = 0, n_off = 0, ex_n = 0, wq_h = 0, wq_len = 0, wq_min = NEVER, ovf = 0;
    bool wq_dirty = false;
    int deg = 0;
    int64_t action = N;
    u32 exA_thr = 0;                       // execution failure threshold M = A << 26 | B: A (B is re-read on a tie)
    u32 lkA[DMAX];                         // per in-link transmission failure thresholds, likewise
    double cap = 1.0, rate = 0.0;
    int slot = -1;
    Rec newt;
    newt.size = newt.cycles = 0;
    newt.twt = newt.ctt = newt.hop = newt.ntx = 0;
    newt.bwt = 0;
    newt.enq = 0;
    Buffer buf;
    buf.head = buf.len = 0;
    buf.head_rec = newt;
    RngFail frng;
    // executing slots: exf / exh as described at EXH_HOP, exT = the reference's task_time at the finishing call
    u32 exf[KMAX], exh[KMAX];
    double exT[KMAX];
    u32 rql[DMAX], rqa[DMAX];
#pragma unroll
    for (int s = 0; s < KMAX; s++) {
        exf[s] = NEVER;
        exh[s] = 0;
        exT[s] = 0.0;
    }
#pragma unroll
    for (int s = 0; s < DMAX; s++) {
        rql[s] = 0;
        rqa[s] = NEVER;
        lkA[s] = 0xffffffffu;
    }
    u32 steps_before = 0;
    if (L.valid && !first) {
        // carried over: registers from `cr`, queues in LDS; only the head of the buffer ring is (re)loaded
        c0 = cr.c0;
        steps_before = cr.steps_before;
        has_new = cr.has_new;
        n_off = cr.n_off;
        ex_n = cr.ex_n;
        wq_h = cr.wq_h;
        wq_len = cr.wq_len;
        wq_min = cr.wq_min;
        wq_dirty = cr.wq_dirty;
        exA_thr = cr.exA_thr;
        deg = cr.deg;
        action = cr.action;
        cap = cr.cap;
        newt = cr.newt;
        buf.head = cr.buf_head;
        buf.len = cr.buf_len;
        if (buf.len > 0) buf.head_rec = load_rec(d.buf + (size_t)buf.head * P + lane);
#pragma unroll
        for (int s = 0; s < DMAX; s++) lkA[s] = cr.lkA[s];
#pragma unroll
        for (int s = 0; s < KMAX; s++) {
            exf[s] = cr.exf[s];
            exh[s] = cr.exh[s];
            exT[s] = cr.exT[s];
        }
        if (has_new && action >= 0 && action < N && action != node) link_of((int)action, slot, rate);
    }
    if (L.valid && first) {
        c0 = d.tick[L.env];
        steps_before = d.total_steps[L.env];
        has_new = d.has_new[lane];
        n_off = d.n_off[lane];
        action = actions[lane];
        buf.head = d.buf_head[lane];
        buf.len = d.buf_len[lane];
        wq_len = d.wq_len[lane];
        ex_n = d.ex_n[lane];
        deg = d.deg[node];
        const u64 me = d.thr_exec_m[node];
        exA_thr = (u32)(me >> 26);
        cap = d.cap[node];
        // Second round of loads: every address the scalars above determine, issued before the first is consumed
        // (into registers first, LDS afterwards), so that the lane's chain pays one memory round trip for the
        // lot rather than one per queue.  The receiving-queue headers go first because the head records (third
        // round) hang off them.  Every block of the launch does this at the same time: the burst is bandwidth
        // bound at 4096 envs (3.8 us at 1024 envs, 6 us at 4096), so only live slots are loaded.
        u32 ql[DMAX], qh[DMAX], qa[DMAX], qs[DMAX];
        u64 ml[DMAX];
#pragma unroll
        for (int s = 0; s < DMAX; s++) {
            const size_t so = (size_t)s * P + lane;
            ql[s] = qh[s] = qa[s] = qs[s] = 0;
            ml[s] = 0;
            if (s < deg) {
                ql[s] = d.rq_len[so];
                qh[s] = d.rq_head[so];
                qa[s] = d.rq_arrive[so];
                qs[s] = d.rq_hstart[so];
                ml[s] = d.thr_link_m[node * D + s];
            }
        }
        {
            const bool off = has_new && action >= 0 && action < N && action != node;
            const int ai = node * N + (off ? (int)action : node);
            const int slot_l = d.slot_of[ai];
            const double rate_l = d.rate[ai];
            const Rec newt_l = load_rec(d.newt + lane);
            const Rec head_l = load_rec(d.buf + (size_t)buf.head * P + lane);
            if (off) {
                slot = slot_l;
                rate = rate_l;
            }
            if (has_new) newt = newt_l;
            if (buf.len > 0) buf.head_rec = head_l;
        }
        u64 exa[KMAX], exc[KMAX];
        double exb[KMAX];
#pragma unroll
        for (int s = 0; s < KMAX; s++) {
            const size_t eo = (size_t)s * P + lane;
            exa[s] = exc[s] = 0;
            exb[s] = 0.0;
            if ((u32)s < ex_n) {
                exa[s] = d.exA[eo];
                exb[s] = d.exB[eo];
                exc[s] = d.exC[eo];
            }
        }
        constexpr int WB = 8;  // waiting-queue entries per batch of loads
        uint4 wm[WB];
        double we[WB];
#pragma unroll
        for (int j = 0; j < WB; j++) {
            const size_t wo = (size_t)j * P + lane;
            wm[j] = make_uint4(0, 0, 0, 0);
            we[j] = 0.0;
            if ((u32)j < wq_len) {
                wm[j] = d.wqm[wo];
                we[j] = d.wqe[wo];
            }
        }
        // third round: the head record of each receiving queue
        uint4 q0[DMAX], q1[DMAX];
#pragma unroll
        for (int s = 0; s < DMAX; s++) {
            q0[s] = q1[s] = make_uint4(0, 0, 0, 0);
            if (s < deg && ql[s] > 0) {
                const uint4* q = reinterpret_cast<const uint4*>(d.rq + ((size_t)s * d.capR + qh[s]) * P + lane);
                q0[s] = q[0];
                q1[s] = q[1];
            }
        }
#pragma unroll
        for (int s = 0; s < KMAX; s++) {
            if ((u32)s < ex_n) {
                // memory form: cet | fin << 16 | hop << 32 | success << 48 (cet < fin: the next call is c0)
                const u32 cet = (u32)exa[s] & 0xffffu, fin = ((u32)exa[s] >> 16) & 0xffffu;
                const u32 hop = (u32)(exa[s] >> 32) & 0xffffu;
                ovf |= hop > EXH_HOP ? 1u : 0u;
                exf[s] = c0 + (fin > cet + 1 ? fin - cet - 1 : 0u);
                exh[s] = (hop & EXH_HOP) | (((exa[s] >> 48) & 1ull) ? EXH_SUCCESS : 0u) | (fin << 16);
                exT[s] = exb[s];
            }
        }
        // consume
#pragma unroll
        for (int s = 0; s < DMAX; s++) {
            if (s < deg) {
                S.rqlen[s * NT + tid] = ql[s];
                S.rqarr[s * NT + tid] = qa[s];
                S.rqhead[s * NT + tid] = qh[s];
                S.rqhs[s * NT + tid] = qs[s];
                S.rqdirty[s * NT + tid] = 0;
                lkA[s] = (u32)(ml[s] >> 26);
            }
            // unconditional (the slot exists for every s < DMAX; an unused one is never read): under the condition
            // the compiler sinks each load to its store, one waited-for round trip per queue
            S.rqrec[(s * 2 + 0) * NT + tid] = q0[s];
            S.rqrec[(s * 2 + 1) * NT + tid] = q1[s];
        }
#pragma unroll
        for (int s = 0; s < KMAX; s++)
            if ((u32)s < ex_n) S.exC[s * NT + tid] = exc[s];
#pragma unroll
        for (int j = 0; j < WB; j++) {
            if ((u32)j < wq_len) {
                S.wqm[j * NT + tid] = wm[j];
                S.wqe[j * NT + tid] = we[j];
                const u32 da = wmeta_drop_at(wm[j]);
                wq_min = da < wq_min ? da : wq_min;
            }
        }
        for (u32 j = WB; j < wq_len; j++) {  // a queue deeper than one batch
            const uint4 m = d.wqm[(size_t)j * P + lane];
            S.wqm[j * NT + tid] = m;
            S.wqe[j * NT + tid] = d.wqe[(size_t)j * P + lane];
            const u32 da = wmeta_drop_at(m);
            wq_min = da < wq_min ? da : wq_min;
        }
    }
    // producer, part 1: this step's task doubles and the failure window, straight from the words regenerated
    // ahead of the cursors during the previous step (per-lane regeneration only if that invariant is absent)
    u32 prod_tpos = 0;  // the task cursor at the start of this step (producer lanes)
    if (__builtin_expect(R.valid, 0)) {  // (out of line: the main wave, the step's critical path, falls through)
        const int ri = R.lidx;
        const size_t tl = (size_t)R.lane, fl = (size_t)P + (size_t)R.lane;
        // cursors: from global memory, or -- a carried step -- from where the consumer left them in LDS
        const u32 tpos = first ? d.rng_pos[tl] : S.rc[RC_TPOS * NT + ri], tah = first ? d.rng_ahead[tl] : S.rc[RC_TAHEAD * NT + ri];
        const u32 fpos = first ? d.rng_pos[fl] : S.rc[RC_FPOS * NT + ri], fah = first ? d.rng_ahead[fl] : S.rc[RC_FAHEAD * NT + ri];
        RngTask pt;
        RngFail pf;
        prod_tpos = tpos;
        // a carried step whose early window and task doubles the producer staged during the previous step, behind that
        // step's barrier of phase C (see there): they are in S.fa / S.tu already
        const bool pre = !first && obs_direct && has_producers && S.rc[RC_PSTAGED * NT + ri] != 0 &&
                         S.rc[RC_PFPOS * NT + ri] == fpos && S.rc[RC_PTPOS * NT + ri] == tpos;
        if (pre && tah >= (u32)AHEAD_TASK && fah >= (u32)AHEAD_FAIL) {
            S.rc[RC_STAGED * NT + ri] = S.rc[RC_PSTAGED * NT + ri];
            pt.ahead = tah;
            pf.ahead = fah;
        } else if (tah >= (u32)AHEAD_TASK && fah >= (u32)AHEAD_FAIL) {
            const u32 hthr = d.hot_thr[R.node];
            // the common case: both streams are regenerated far enough; every load is issued before the first
            // word is tempered, so the whole window costs one memory round trip
            const u32* tw = d.mt + tl * MT_STRIDE + tpos;  // wrap-free thanks to the mirror
            const u32* fw = d.mt + fl * MT_STRIDE + fpos;
            // (the first RB_EARLY draws now; the rest of the window after the second barrier, off the burst of
            // loads with which every block of the launch starts)
            u32x4 fv[RB_EARLY / 2];
#pragma unroll
            for (int i = 0; i < RB_EARLY / 2; i++) fv[i] = *reinterpret_cast<const u32x4*>(fw + 4 * i);
            const u32x4 t0 = *reinterpret_cast<const u32x4*>(tw), t1 = *reinterpret_cast<const u32x4*>(tw + 4);
            u32 hot16 = 0;
#pragma unroll
            for (int i = 0; i < RB_EARLY / 2; i++) {
                const u32 a0 = mt_temper(fv[i].x) >> 5, a1 = mt_temper(fv[i].z) >> 5;
                S.fa[(2 * i) * NT + ri] = a0;
                S.fa[(2 * i + 1) * NT + ri] = a1;
                hot16 |= (a0 >= hthr ? 1u : 0u) << (2 * i);
                hot16 |= (a1 >= hthr ? 1u : 0u) << (2 * i + 1);
            }
            static_assert(RB_EARLY <= 16, "the early hot flags share the RC_STAGED word");
            S.rc[RC_STAGED * NT + ri] = (u32)RB_EARLY | (hot16 << STAGED_HOT_SHIFT);
            pt.u0 = mt_to_double(t0.x, t0.y);
            pt.u1 = mt_to_double(t0.z, t0.w);
            pt.u2 = mt_to_double(t1.x, t1.y);
            pt.u3 = mt_to_double(t1.z, t1.w);
            pt.ahead = tah;
            pf.ahead = fah;
            S.tu[0 * NT + ri] = pt.u0;
            S.tu[1 * NT + ri] = pt.u1;
            S.tu[2 * NT + ri] = pt.u2;
            S.tu[3 * NT + ri] = pt.u3;
        } else {
            const u32 hthr = d.hot_thr[R.node];
            pt.open(d.mt + tl * MT_STRIDE, tpos, tah);
            pf.open(d.mt + fl * MT_STRIDE, fpos, fah, S.fa + ri, NT, (u32)RB_FAIL);
            S.hot[ri] = pf.hot_mask((u32)RB_FAIL, hthr);
            S.rc[RC_STAGED * NT + ri] = (u32)RB_FAIL | STAGED_FINAL;
            S.tu[0 * NT + ri] = pt.u0;
            S.tu[1 * NT + ri] = pt.u1;
            S.tu[2 * NT + ri] = pt.u2;
            S.tu[3 * NT + ri] = pt.u3;
        }
        S.rc[RC_FPOS * NT + ri] = fpos;
        S.rc[RC_FAHEAD * NT + ri] = pf.ahead;
        S.rc[RC_TPOS * NT + ri] = tpos;
        S.rc[RC_TAHEAD * NT + ri] = pt.ahead;
        S.rc[RC_DONE * NT + ri] = 0;
    }
    if (producer_wave && first) {
        const int p0 = has_producers ? (int)threadIdx.x - d.nmain : (int)threadIdx.x;
        const int np = has_producers ? (int)blockDim.x - d.nmain : (int)blockDim.x;
#pragma unroll
        for (int i = 0; i < ACC_LDS / 64; i++) {
            const int a = p0 + i * np;
            if (a < ACC_LDS) S.acc[a] = lut_lo[i];
        }
        for (int a = ACC_LDS + p0; a < d.acc_words; a += np) S.acc[a] = reinterpret_cast<const u64*>((const int16_t*)d.acc)[a];  // long episodes
    }
    if (em && first && L.valid) em->s_cnt[tid] = wq_len | (ex_n << 16);
    EC_STAMP_P(11);
    __syncthreads();
    EC_STAMP(1);
    if (__builtin_expect(EVX && has_producers && producer_wave && R.valid, 0)) {
        // Looping kernels: the rest of the failure window NOW, while the main lanes are in the offload phase and this wave
        // would wait at the next barrier -- all of it (the stream is regenerated a whole window ahead of the cursor; the
        // step's draw bound, known only behind that barrier, merely says how much of it can be used).  Part 2 below is
        // then regeneration only, and a busy lane finds its 17th draw staged when the mini-slots begin.
        const int ri = R.lidx;
        const u32 staged1 = S.rc[RC_STAGED * NT + ri];
        if (!(staged1 & STAGED_FINAL)) {
            const u32* fw = d.mt + ((size_t)P + (size_t)R.lane) * MT_STRIDE + S.rc[RC_FPOS * NT + ri];
            u32x4 fv[(RB_FAIL - RB_EARLY) / 2];
#pragma unroll
            for (int i = 0; i < (RB_FAIL - RB_EARLY) / 2; i++) fv[i] = *reinterpret_cast<const u32x4*>(fw + 4 * (i + RB_EARLY / 2));
            const u32 hthr = !first ? S.nodec[R.node * NODEC_W + 15] : d.hot_thr[R.node];
            u64 hotm = (u64)((staged1 >> STAGED_HOT_SHIFT) & 0xffffu);
#pragma unroll
            for (int i = 0; i < (RB_FAIL - RB_EARLY) / 2; i++) {
                const u32 a0 = mt_temper(fv[i].x) >> 5, a1 = mt_temper(fv[i].z) >> 5;
                S.fa[(RB_EARLY + 2 * i) * NT + ri] = a0;
                S.fa[(RB_EARLY + 2 * i + 1) * NT + ri] = a1;
                hotm |= (u64)(a0 >= hthr ? 1u : 0u) << (RB_EARLY + 2 * i);
                hotm |= (u64)(a1 >= hthr ? 1u : 0u) << (RB_EARLY + 2 * i + 1);
            }
            S.hot[ri] = hotm;
            S.rc[RC_STAGED * NT + ri] = (u32)RB_FAIL | STAGED_FINAL;  // (published by the barrier behind the offload phase)
        }
    }
    // ec_rollout_mlp_policy: the Actor rows beyond a main wave's first 32 were evaluated by a producer wave, which left
    // the actions in LDS ahead of this barrier (k_rollout_mlp)
    if (s_pick && L.valid && (threadIdx.x & 63u) >= 32u) {
        action = s_pick[tid];
        slot = -1;
        rate = 0.0;
        if (has_new && action >= 0 && action < N && action != node) link_of((int)action, slot, rate);
    }

    if (em && L.valid) {
        // the expert's decision for this lane (ec_expert_policy's, on the state the previous step left), then the coin
        const u32* cnt = em->s_cnt + L.env_local * N;
        const int ea = expert_decide(d, node, em->avail + lane * d.A, has_new, newt.size, newt.cycles, [&](int t, int& nw, int& nx) {
            const u32 c = cnt[t];
            nw = (int)(c & 0xffffu);
            nx = (int)(c >> 16);
        });
        em->expert_out[lane] = ea;
        if (em->coin[L.env] < em->ratio) {
            action = ea;
            slot = -1;
            rate = 0.0;
            if (has_new && action >= 0 && action < N && action != node) link_of((int)action, slot, rate);
        }
    }

    // ---------------------------------------------------------------- phase A: offload_task, edge_node.py:199-234
    if (L.valid && has_new && action != N) {
        if (action == node) {
            // local: execution_queue.append(task)
            n_off++;
            const double et = ddiv(newt.cycles, cap);
            WMeta m;
            m.enq = c0;
            m.nexec = lut.ticks_to_reach(et);
            m.twt = newt.twt;
            m.ctt = newt.ctt;
            m.bwt = sat16((u32)newt.bwt, ovf);
            m.hop = newt.hop;
            double unused;
            m.nd = lut.deadline_ticks(0, 0, m.twt, m.ctt, m.bwt, &unused);
            if (__builtin_expect(wq_len >= (u32)d.capW, 0)) {
                flag(d, EC_ST_WAIT_OVERFLOW);
            } else {
                const u32 pos = (wq_h + wq_len) & wmask;
                const uint4 pm = wmeta_pack(m);
                S.wqm[pos * NT + tid] = pm;
                S.wqe[pos * NT + tid] = et;
                wq_len++;
                wq_dirty = true;
                const u32 da = wmeta_drop_at(pm);
                wq_min = da < wq_min ? da : wq_min;
            }
        } else if (__builtin_expect(slot < 0, 0)) {
            flag(d, EC_ST_BAD_ACTION);  // reference: ZeroDivisionError at edge_node.py:227
        } else {
            n_off++;
            const int dt_ = L.env_local * N + (int)action;  // destination lane's thread index in this block
            const int64_t dst = L.env * N + action;
            Rec r = newt;
            const double tt = ddiv(r.size, rate);
            r.ntx = (u16)lut.ticks_to_reach(tt);
            r.hop = (u16)(r.hop + 1);
            r.enq = c0;
            // receiving_queues[self.id] of the destination has this lane as its only producer
            const u32 ql = S.rqlen[slot * NT + dt_], qh = S.rqhead[slot * NT + dt_];
            if (__builtin_expect(ql >= (u32)d.capR, 0)) {
                flag(d, EC_ST_RECV_OVERFLOW);
            } else {
                if (ql == 0) {
                    union {
                        Rec r;
                        uint4 v[2];
                    } u;
                    u.r = r;
                    S.rqrec[(slot * 2 + 0) * NT + dt_] = u.v[0];
                    S.rqrec[(slot * 2 + 1) * NT + dt_] = u.v[1];
                    S.rqdirty[slot * NT + dt_] = 1;
                    S.rqhs[slot * NT + dt_] = c0;
                    S.rqarr[slot * NT + dt_] = arrive_at(c0, r.ntx, r.ctt);
                } else {
                    u32 pos = qh + ql;
                    if (pos >= (u32)d.capR) pos -= d.capR;
                    store_rec(d.rq + ((size_t)slot * d.capR + pos) * P + dst, r);
                }
                S.rqlen[slot * NT + dt_] = ql + 1;
            }
        }
    }
    if (L.valid) S.rc[RC_NX * NT + tid] = ex_n + wq_len;
    __syncthreads();  // pushes into the LDS queue heads are visible to their destination lanes
    EC_STAMP(2);

    // producer, part 2 (concurrent with the mini-slots when it has a wave of its own): regenerate both streams
    // far enough ahead that the NEXT step finds its window and task doubles ready whatever this step consumes
    if (__builtin_expect(producer_wave, 0)) {
        u32 tpos = 0, tah = 0, fpos = 0, fah = 0, want_f = 0;
        const int ri = R.lidx;
        if (R.valid) {
            fpos = S.rc[RC_FPOS * NT + ri];
            fah = S.rc[RC_FAHEAD * NT + ri];
            tpos = S.rc[RC_TPOS * NT + ri];
            tah = S.rc[RC_TAHEAD * NT + ri];
            u32 n_ne = 0;
            const int rdeg = (EVX && !first) ? (int)S.nodec[R.node * NODEC_W + 14] : d.deg[R.node];
            for (int s = 0; s < rdeg; s++) n_ne += S.rqlen[s * NT + ri] > 0 ? 1u : 0u;
            const u32 nx = S.rc[RC_NX * NT + ri];
            const u32 bound = (u32)d.S * ((nx < (u32)d.k ? nx : (u32)d.k) + n_ne);  // the step's draw bound, as below
            const u32 staged1 = S.rc[RC_STAGED * NT + ri];
            if (!(staged1 & STAGED_FINAL)) {
                // the rest of the window, as far as this step can draw (fah >= AHEAD_FAIL here)
                const u32 upto = bound < (u32)RB_FAIL ? bound : (u32)RB_FAIL;
                const u32* fw = d.mt + ((size_t)P + (size_t)R.lane) * MT_STRIDE + fpos;
                u32x4 fv[(RB_FAIL - RB_EARLY) / 2];
#pragma unroll
                for (int i = 0; i < (RB_FAIL - RB_EARLY) / 2; i++) {
                    fv[i] = u32x4{0, 0, 0, 0};
                    if ((u32)(RB_EARLY + 2 * i) < upto) fv[i] = *reinterpret_cast<const u32x4*>(fw + 4 * (i + RB_EARLY / 2));
                }
                const u32 hthr = (EVX && !first) ? S.nodec[R.node * NODEC_W + 15] : d.hot_thr[R.node];
                u64 hotm = (u64)((staged1 >> STAGED_HOT_SHIFT) & 0xffffu);
#pragma unroll
                for (int i = 0; i < (RB_FAIL - RB_EARLY) / 2; i++) {
                    const u32 a0 = mt_temper(fv[i].x) >> 5, a1 = mt_temper(fv[i].z) >> 5;
                    S.fa[(RB_EARLY + 2 * i) * NT + ri] = a0;
                    S.fa[(RB_EARLY + 2 * i + 1) * NT + ri] = a1;
                    hotm |= (u64)(a0 >= hthr ? 1u : 0u) << (RB_EARLY + 2 * i);
                    hotm |= (u64)(a1 >= hthr ? 1u : 0u) << (RB_EARLY + 2 * i + 1);
                }
                static_assert(RB_FAIL <= 64, "one hot bit per window draw");
                S.hot[ri] = hotm;  // (draws beyond `staged` were not loaded: their bits are never looked at)
                const u32 staged = upto < (u32)RB_EARLY ? (u32)RB_EARLY : (upto + 1u) & ~1u;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __hip_atomic_store(&S.rc[RC_STAGED * NT + ri], staged | STAGED_FINAL, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            want_f = (u32)AHEAD_FAIL + 2u * bound;
            want_f = want_f < (u32)AHEAD_MAX ? want_f : (u32)AHEAD_MAX;
        }
        const u32 fah0 = fah, tah0 = tah;
        mt_coop_refill(d.mt, (u64)(R.valid ? R.lane : 0), tpos, tah, 2u * AHEAD_TASK, R.valid);
        mt_coop_refill(d.mt, (u64)P + (u64)(R.valid ? R.lane : 0), fpos, fah, want_f, R.valid);
        if (R.valid) {
            S.rc[RC_FDELTA * NT + ri] = fah - fah0;
            S.rc[RC_TDELTA * NT + ri] = tah - tah0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the regenerated words before the flag
            __hip_atomic_store(&S.rc[RC_DONE * NT + ri], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        EC_STAMP_P(12);
    }

    // ---------------------------------------------------------------- failure-stream window
    // Within a step no queue gains a task after phase A, so this bounds the lane's draws: per mini-slot at most
    // min(k, executing + waiting) execution draws and one draw per non-empty receiving queue.
    u32 draw_bound = 0;
    u64 hot = 0;  // "hot" draws of the failure window (StepLds::hot); bits at or beyond frng.staged are undefined
    bool f_final = true;  // the producer has staged all it is going to stage of the first window
    if (L.valid) {
        u32 n_ne = 0;
#pragma unroll
        for (int s = 0; s < DMAX; s++) {
            if (s < deg) {
                rql[s] = S.rqlen[s * NT + tid];
                rqa[s] = S.rqarr[s * NT + tid];
                n_ne += rql[s] > 0 ? 1u : 0u;
            }
        }
        const u32 n_ex = ex_n + wq_len < (u32)d.k ? ex_n + wq_len : (u32)d.k;
        draw_bound = (u32)d.S * (n_ex + n_ne);
        const u32 st0 = S.rc[RC_STAGED * NT + tid];
        frng.attach(d.mt + ((size_t)P + lane) * MT_STRIDE, S.rc[RC_FPOS * NT + tid], S.rc[RC_FAHEAD * NT + tid],
                    S.fa + tid, NT, st0 & STAGED_COUNT);
        f_final = (st0 & STAGED_FINAL) != 0;
        // (a window the producer is still filling: the flags of its first RB_EARLY draws travel in the count word)
        hot = f_final ? S.hot[tid] : (u64)((st0 >> STAGED_HOT_SHIFT) & 0xffffu);
    }
    bool f_synced = false;  // frng.ahead includes the producer's part-2 regeneration

    EC_STAMP(3);
    // ---------------------------------------------------------------- phase B: mini-slots, edge_computing_env.py:101-105
    double rew = 0.0, ct = 0.0;
    u32 n_fail = 0, n_drop = 0, n_fin = 0, n_succ = 0;
    if (L.valid) {
        // Event skipping.  Inside a step the nodes of an env do not interact (offload_task pushed everything in phase
        // A), so a lane need not walk the mini-slots in lock step with its wave.  Most mini-slots of a lane are
        // uneventful: every executing task and every non-empty receiving queue draws one number that fails nothing and
        // one tick is added to the running timers.  Such mini-slots are taken in one jump (timers += n, draw cursor
        // += n * draws per mini-slot); the reference's per-mini-slot code below runs only for mini-slots in which
        // something can happen: an admission, a task reaching its finish tick, a head record arriving, a deadline
        // sweep coming due, a draw whose first word reaches the node's lowest failure threshold ("hot"), or the end of
        // the staged part of the draw window.  A wave iterates as often as its busiest lane has such mini-slots.
        const u32 S_ = (u32)d.S;
        u32 ms = 0;
#ifdef EC_STAMPS
        u32 dbg_iters = 0;
        __shared__ u64 dbg_lds[4][12];
        if ((threadIdx.x & 63u) == 0) {
            for (int i = 0; i < 11; i++) dbg_lds[threadIdx.x >> 6][i] = 0;
            dbg_lds[threadIdx.x >> 6][11] = wall_clock64();
        }
#endif
        for (;;) {
            // ---- execute_task (i), edge_node.py:238-241: the call with index c0 + ms admits heads of the waiting
            // queue while a slot is free.  Done here, ahead of the search, so that a mini-slot whose only event is an
            // admission (the call after a slot was vacated) needs no pass of its own.
            if (ms < S_) {
                const u32 c = c0 + ms;
                while (ex_n < (u32)d.k && wq_len > 0) {
                    const WMeta w = wmeta_unpack(S.wqm[wq_h * NT + tid]);
                    wq_h = (wq_h + 1) & wmask;
                    wq_len--;
                    wq_dirty = true;
                    const u32 ewt = sat16(c - w.enq, ovf);
                    bool succ;
                    u32 fin;
                    double T;
                    // (closed form, see Lut::deadline_ticks: a task that completes with fewer than dl_ticks ticks on its
                    // five timers cannot have been dropped on the way -- most admissions)
                    if (d.fast_dl && (int)(w.nexec + ewt + w.twt + w.ctt + w.bwt) < d.dl_ticks) {
                        succ = true;
                        fin = w.nexec;
                        T = lut.task_time(ewt, fin, w.twt, w.ctt, w.bwt);
                    } else {
                        double Td;
                        const u32 dc = lut.deadline_ticks(1, ewt, w.twt, w.ctt, w.bwt, &Td);
                        // the drop test precedes the completion test (:268,278): a tie is a drop
                        succ = w.nexec < dc;
                        fin = succ ? w.nexec : dc;
                        T = succ ? lut.task_time(ewt, fin, w.twt, w.ctt, w.bwt) : Td;
                    }
                    ovf |= w.hop > EXH_HOP ? 1u : 0u;
                    const u32 nf = c + fin - 1, nh = (w.hop & EXH_HOP) | (succ ? EXH_SUCCESS : 0u) | (fin << 16);
#pragma unroll
                    for (int s = 0; s < KMAX; s++) {
                        if ((u32)s == ex_n) {
                            exf[s] = nf;
                            exh[s] = nh;
                            exT[s] = T;
                        }
                    }
                    S.exC[ex_n * NT + tid] = pack4(ewt, w.twt, w.ctt, w.bwt);
                    ex_n++;
                }
            }
            EC_SUB(1);
            // ---- the next mini-slot m >= ms that needs the full code
            u32 q = ex_n;  // draws per uneventful mini-slot
#pragma unroll
            for (int s = 0; s < DMAX; s++) q += rql[s] > 0 ? 1u : 0u;
            u32 m;
            {
                // absolute call indices: a task completes or drops at exf (NEVER: slot unused), a head record arrives
                // at rqa (NEVER: queue empty), the deadline sweep of the waiting queue comes due at wq_min
                u32 na = wq_len > 0 ? wq_min : NEVER;
#pragma unroll
                for (int s = 0; s < KMAX; s++) na = exf[s] < na ? exf[s] : na;
#pragma unroll
                for (int s = 0; s < DMAX; s++) na = rqa[s] < na ? rqa[s] : na;
                const u32 cnow = c0 + ms;
                m = na > cnow ? na - c0 : ms;
                if (q > 0) {  // the first hot draw, or the end of what is staged of the window
                    const u32 left = frng.staged - frng.idx;  // staged draws not yet consumed (<= RB_FAIL)
                    const u64 h = (hot >> frng.idx) & ((1ull << left) - 1ull);
                    const u32 pd = h ? (u32)__ffsll((unsigned long long)h) - 1u : left;
                    // pd / q for pd <= 64, q <= KMAX + DMAX: (pd + 0.5) / q is at least 0.5 / q away from an integer
                    const u32 at = ms + (u32)(((float)pd + 0.5f) * __builtin_amdgcn_rcpf((float)q));
                    m = at < m ? at : m;
                }
                m = m < S_ ? m : S_;
                frng.idx += (m - ms) * q;
            }
            if (m >= S_) break;
            ms = m + 1;
            const u32 c = c0 + m;
#ifdef EC_STAMPS
            dbg_iters++;  // wave 0's count lands in stamp 9 of one-step launches (lanes differ: lane 0's is stored)
#endif
            EC_SUB(0);
            if (wq_len == 0) wq_min = NEVER;
            EC_SUB(1);

            // the draws of this mini-slot: one per executing task, then one per non-empty receiving queue (q in all)
            if (__builtin_expect(frng.idx + q > frng.staged && !f_final, 0)) {
                // the producer is still filling this window: it has been at it since the second barrier
                u32 st;
                while (!((st = __hip_atomic_load(&S.rc[RC_STAGED * NT + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) &
                         STAGED_FINAL))
                    __builtin_amdgcn_s_sleep(2);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                frng.staged = st & STAGED_COUNT;
                hot = S.hot[tid];
                f_final = true;
            }
            if (__builtin_expect(frng.idx + q > frng.staged, 0)) {
                if (__builtin_expect(!f_synced, 0)) {
                    // the producer may still be regenerating this stream: wait for it (it never waits for us)
                    while (__hip_atomic_load(&S.rc[RC_DONE * NT + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0)
                        __builtin_amdgcn_s_sleep(2);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    frng.ahead += S.rc[RC_FDELTA * NT + tid];
                    f_synced = true;
                }
#ifdef EC_STAMPS
                if (d.stamps) atomicAdd((unsigned long long*)&d.stamps[(size_t)blockIdx.x * 16 + 13], 1ull);
#endif
                const u32 left = draw_bound - (frng.used + frng.idx);  // the bound only shrinks as draws are spent
                frng.restage(left < (u32)RB_FAIL ? left : (u32)RB_FAIL);
                u32 hthr = exA_thr;
#pragma unroll
                for (int s = 0; s < DMAX; s++) hthr = lkA[s] < hthr ? lkA[s] : hthr;
                hot = frng.hot_mask(frng.staged, hthr);
            }
            u32 ae[KMAX], ar[DMAX], ir[DMAX];
            const u32 base = frng.idx;
            {
#pragma unroll
                for (int s = 0; s < KMAX; s++) {
                    const u32 i = base + s;
                    ae[s] = frng.buf[(i < (u32)RB_FAIL ? i : (u32)RB_FAIL - 1) * NT];
                }
                u32 r = base + ex_n;
#pragma unroll
                for (int s = 0; s < DMAX; s++) {
                    ir[s] = r;
                    ar[s] = frng.buf[(r < (u32)RB_FAIL ? r : (u32)RB_FAIL - 1) * NT];
                    r += rql[s] > 0 ? 1u : 0u;
                }
                frng.idx = r;
            }

            EC_SUB(2);
            // (ii) is implicit: waiting time is c - enq.  (iii) the executing tasks, in order (:252-296)
            // Integer bookkeeping in straight-line code for all slots.  EVX (the looping kernel): the fp64 reward /
            // completion-time terms and the hop histogram of the slots that failed or finished are added afterwards,
            // one event per lane per pass in slot order (the order of the reference's loop), as in receive_task
            // below; the one-step kernel measured faster with them predicated in the slot loop.
            u32 keep = 0;
            {
                const u32 n0 = ex_n;
                u32 evx = 0, fbx = 0, sbx = 0;
#pragma unroll
                for (int s = 0; s < KMAX; s++) {
                    const bool act = (u32)s < n0;
                    bool fail = act && ae[s] > exA_thr;
                    if (__builtin_expect(act && ae[s] == exA_thr, 0))  // 2^-27 tie on the first word
                        fail = frng.b_of(base + s) > (u32)(d.thr_exec_m[node] & 0x3ffffffull);
                    const bool done = act && !fail && c >= exf[s];
                    const bool succ = done && (exh[s] & EXH_SUCCESS);
                    const bool drop = done && !succ;
                    n_fail += fail;
                    n_drop += drop;
                    n_succ += succ;
                    n_fin += (fail || done);
                    if (!EVX) {
                        const u32 hop = exh[s] & EXH_HOP;
                        rew = fail ? dadd(rew, d.pen_fail) : rew;
                        rew = drop ? dadd(rew, d.pen_drop) : rew;
                        rew = succ ? dadd(rew, d.rew_succ) : rew;
                        rew = succ ? dadd(rew, hop > 1 ? dmul(d.pen_hop, (double)(hop - 1)) : 0.0) : rew;
                        ct = done ? dadd(ct, exT[s]) : ct;
                        if (succ) {
                            u32 bin = hop;
                            if (__builtin_expect(bin >= (u32)d.H, 0)) {
                                bin = d.H - 1;
                                flag(d, EC_ST_HOP_OVERFLOW);
                            }
                            atomicAdd(d.hop_hist + L.env * d.H + bin, 1);
                        }
                    }
                    evx |= (fail || done) ? (1u << s) : 0u;
                    fbx |= fail ? (1u << s) : 0u;
                    sbx |= succ ? (1u << s) : 0u;
                    const bool kp = act && !fail && !done;
                    keep |= kp ? (1u << s) : 0u;
                }
                EC_SUB(3);
                while (EVX && evx) {
                    const int s = __ffs((int)evx) - 1;
                    evx &= evx - 1;
                    if ((fbx >> s) & 1u) {
                        rew = dadd(rew, d.pen_fail);
                        continue;
                    }
                    u32 eh = 0;
                    double eT = 0.0;
#pragma unroll
                    for (int q = 0; q < KMAX; q++) {
                        eh = s == q ? exh[q] : eh;
                        eT = s == q ? exT[q] : eT;
                    }
                    if ((sbx >> s) & 1u) {
                        const u32 hop = eh & EXH_HOP;
                        rew = dadd(rew, d.rew_succ);
                        rew = dadd(rew, hop > 1 ? dmul(d.pen_hop, (double)(hop - 1)) : 0.0);
                        u32 bin = hop;
                        if (__builtin_expect(bin >= (u32)d.H, 0)) {
                            bin = d.H - 1;
                            flag(d, EC_ST_HOP_OVERFLOW);
                        }
                        atomicAdd(d.hop_hist + L.env * d.H + bin, 1);
                    } else {
                        rew = dadd(rew, d.pen_drop);
                    }
                    ct = dadd(ct, eT);
                }
                EC_SUB(4);
                if (keep != ((1u << n0) - 1)) {
                    // stable compaction of the survivors (executing_queue.remove keeps the order)
                    u32 nw[KMAX], nh[KMAX];
                    double nT[KMAX];
#pragma unroll
                    for (int w = 0; w < KMAX; w++) {
                        nw[w] = NEVER;
                        nh[w] = 0;
                        nT[w] = 0.0;
                    }
                    u32 cnt = 0;
#pragma unroll
                    for (int s = 0; s < KMAX; s++) {
                        const bool kp = (keep >> s) & 1u;
#pragma unroll
                        for (int w = 0; w <= s; w++) {
                            const bool hit = kp && cnt == (u32)w;
                            nw[w] = hit ? exf[s] : nw[w];
                            nh[w] = hit ? exh[s] : nh[w];
                            nT[w] = hit ? exT[s] : nT[w];
                        }
                        if (kp && cnt != (u32)s) S.exC[cnt * NT + tid] = S.exC[s * NT + tid];
                        cnt += kp ? 1u : 0u;
                    }
#pragma unroll
                    for (int w = 0; w < KMAX; w++) {
                        exf[w] = nw[w];
                        exh[w] = nh[w];
                        exT[w] = nT[w];
                    }
                    ex_n = cnt;
                }
            }
            EC_SUB(5);
            // (iv) deadline sweep of the waiting queue (:299-309), only when some drop has come due
            if (wq_len > 0 && wq_min <= c) {
                u32 rd = wq_h, wr = wq_h, kept = 0, newmin = NEVER;
                for (u32 j = 0; j < wq_len; j++) {
                    const uint4 pm = S.wqm[rd * NT + tid];
                    const u32 da = wmeta_drop_at(pm);
                    if (da <= c) {
                        const WMeta w = wmeta_unpack(pm);
                        n_drop++;
                        n_fin++;
                        rew = dadd(rew, d.pen_drop);
                        ct = dadd(ct, lut.task_time(c - w.enq + 1, 0, w.twt, w.ctt, w.bwt));
                    } else {
                        if (wr != rd) {
                            S.wqm[wr * NT + tid] = pm;
                            S.wqe[wr * NT + tid] = S.wqe[rd * NT + tid];
                        }
                        wr = (wr + 1) & wmask;
                        kept++;
                        newmin = da < newmin ? da : newmin;
                    }
                    rd = (rd + 1) & wmask;
                }
                wq_len = kept;
                wq_min = newmin;
                wq_dirty = true;
            }
            EC_SUB(6);
            // ---- receive_task, edge_node.py:311-333: in-neighbours in ascending source order
            // Classify the queues in straight-line code, then handle the events (a failed or arrived head) one per
            // lane per pass, lowest slot first: the passes a wave makes are the most events any of its lanes has
            // (usually one), where a handler per slot ran whenever ANY lane had an event in THAT slot (~2.2 of 4).
            u32 ev = 0, fb = 0;
#pragma unroll
            for (int s = 0; s < DMAX; s++) {
                const bool ne = rql[s] > 0;
                bool failed = ne && ar[s] > lkA[s];
                if (__builtin_expect(ne && ar[s] == lkA[s], 0))  // 2^-27 tie on the first word
                    failed = frng.b_of(ir[s]) > (u32)(d.thr_link_m[node * D + s] & 0x3ffffffull);
                const bool arrived = ne && !failed && c >= rqa[s];
                ev |= (failed || arrived) ? (1u << s) : 0u;
                fb |= failed ? (1u << s) : 0u;
            }
            EC_SUB(7);
            while (ev) {
                const int s = __ffs((int)ev) - 1;
                ev &= ev - 1;
                if ((fb >> s) & 1u) {
                    rew = dadd(rew, d.pen_fail);
                    n_fail++;
                    n_fin++;
                } else {
                    union {
                        Rec r;
                        uint4 v[2];
                    } uu;
                    uu.v[0] = S.rqrec[(s * 2 + 0) * NT + tid];
                    uu.v[1] = S.rqrec[(s * 2 + 1) * NT + tid];
                    Rec r = uu.r;
                    const u32 hs = S.rqhs[s * NT + tid];
                    r.twt = (u16)sat16((u32)r.twt + (hs - r.enq), ovf);
                    r.ctt = (u16)sat16((u32)r.ctt + (c - hs + 1), ovf);
                    rew = dadd(rew, r.hop > 1 ? dmul(d.pen_hop, (double)(r.hop - 1)) : 0.0);
                    buf.push(d, lane, n_off, r);  // buffer_queue.append, :333
                }
                // popleft, then promote the next record to head for call c+1
                u32 qh = S.rqhead[s * NT + tid];
                qh = qh + 1 == (u32)d.capR ? 0u : qh + 1;
                u32 cur = 0;
#pragma unroll
                for (int q = 0; q < DMAX; q++) cur = s == q ? rql[q] : cur;
                const u32 ql = cur - 1;
                u32 arr = NEVER;
                if (__builtin_expect(ql > 0, 0)) {  // deeper than one: the next record lives in the global ring
                    const uint4* q = reinterpret_cast<const uint4*>(d.rq + ((size_t)s * d.capR + qh) * P + lane);
                    union {
                        Rec r;
                        uint4 v[2];
                    } nx;
                    nx.v[0] = q[0];
                    nx.v[1] = q[1];
                    S.rqrec[(s * 2 + 0) * NT + tid] = nx.v[0];
                    S.rqrec[(s * 2 + 1) * NT + tid] = nx.v[1];
                    S.rqhs[s * NT + tid] = c + 1;
                    arr = arrive_at(c + 1, nx.r.ntx, nx.r.ctt);
                }
                S.rqdirty[s * NT + tid] = 0;
                S.rqhead[s * NT + tid] = qh;
#pragma unroll
                for (int q = 0; q < DMAX; q++) {
                    rql[q] = s == q ? ql : rql[q];
                    rqa[q] = s == q ? arr : rqa[q];
                }
            }
            EC_SUB(8);
        }
#ifdef EC_STAMPS
        {   // passes the wave made through the per-mini-slot code = the most any of its lanes needed
            u32 w = dbg_iters;
            for (int o = 32; o >= 1; o >>= 1) {
                const u32 x = __shfl_xor(w, o);
                w = x > w ? x : w;
            }
            if (threadIdx.x == 0 && d.stamps && first && last) {
                d.stamps[(size_t)blockIdx.x * 16 + 9] = w;
                d.stamps[(size_t)blockIdx.x * 16 + 10] = dbg_iters;
                EC_SUB(9);
                for (int i = 0; i < 10; i++) d.stamps[((size_t)gridDim.x + blockIdx.x) * 16 + i] = dbg_lds[0][i];
            }
        }
#endif
    }

    EC_STAMP(4);
    // ---------------------------------------------------------------- phase C: per-env statistics, edge_computing_env.py:108-123
    // loads that phase D needs, issued before the barrier so that their latency hides behind it
    double bin_qn = 0.0, bin_px2 = 0.0;
    u32 bin_flip = 0;
    u64 cnt_t = 0, cnt_f = 0;
    // Phase E in its direct form (every env's lanes inside one wave, adjacency rows as bit masks): each main lane writes
    // its own observation and availability row, the "queue full" flags of its env come from a ballot -- no LDS staging,
    // no barrier, no cooperative copy loops.  Its per-node constants are requested here as well.
    double o_qthr = 0.0, o_qlen = 1.0;
    u64 o_adjb = 0;
    float o_c0 = 0.0f, o_c1 = 0.0f, o_c2 = 0.0f;
    if (L.valid) {
        cnt_t = d.rng_cnt[lane];
        cnt_f = d.rng_cnt[P + lane];
        u32* nc = S.nodec + node * NODEC_W;
        if (EVX && !first) {
            // a carried step of a looping kernel: the node's constants from the table the first step filed (an LDS read
            // where the global loads' latency showed: the barrier below is short when the producer wave is done)
            bin_qn = *reinterpret_cast<const double*>(nc + 0);
            bin_px2 = *reinterpret_cast<const double*>(nc + 2);
            bin_flip = nc[10];
            if (obs_direct) {
                o_qthr = *reinterpret_cast<const double*>(nc + 4);
                o_qlen = *reinterpret_cast<const double*>(nc + 6);
                o_adjb = *reinterpret_cast<const u64*>(nc + 8);
                o_c0 = __uint_as_float(nc[11]);
                o_c1 = __uint_as_float(nc[12]);
                o_c2 = __uint_as_float(nc[13]);
            }
        } else {
            bin_qn = d.bin_qn[node];
            bin_px2 = d.bin_px2[node];
            bin_flip = d.bin_flip[node];
            if (obs_direct) {
                o_qthr = d.qthr[node];
                o_qlen = d.qlen[node];
                o_adjb = d.adj_bits[node];
                o_c0 = d.obs_const[node * 3 + 0];
                o_c1 = d.obs_const[node * 3 + 1];
                o_c2 = d.obs_const[node * 3 + 2];
            }
            if (EVX && !last && L.env_local == 0) {  // one env's lanes file the table for the steps that follow
                *reinterpret_cast<double*>(nc + 0) = bin_qn;
                *reinterpret_cast<double*>(nc + 2) = bin_px2;
                *reinterpret_cast<double*>(nc + 4) = o_qthr;
                *reinterpret_cast<double*>(nc + 6) = o_qlen;
                *reinterpret_cast<u64*>(nc + 8) = o_adjb;
                nc[10] = bin_flip;
                nc[11] = __float_as_uint(o_c0);
                nc[12] = __float_as_uint(o_c1);
                nc[13] = __float_as_uint(o_c2);
                nc[14] = (u32)deg;
                nc[15] = d.hot_thr[node];
                if ((s_pick != nullptr || em != nullptr) && d.link_lds) {
                    for (int a = 0; a < N; a++) S.slotb[node * N + a] = (signed char)d.slot_of[node * N + a];
                    for (int q = 0; q < deg; q++) S.linkr[node * DMAX + q] = d.rate[d.in_src[node * D + q] * N + node];
                }
            }
        }
    }
    double tu0 = 0.0, tu1 = 0.0, tu2 = 0.0, tu3 = 0.0;
    if (L.valid) {
        // (the step's task doubles: read ahead of the barrier, behind which the producer may stage the next step's)
        tu0 = S.tu[0 * NT + tid];
        tu1 = S.tu[1 * NT + tid];
        tu2 = S.tu[2 * NT + tid];
        tu3 = S.tu[3 * NT + tid];
        S.rc[RC_FNEXT * NT + tid] = mt_wrap(frng.pos + 2 * frng.idx);  // the failure cursor is final
        S.rew[tid] = rew;
        S.ct[tid] = ct;
        S.cnt[0 * NT + tid] = n_fin;
        S.cnt[1 * NT + tid] = n_succ;
        S.cnt[2 * NT + tid] = n_drop;
        S.cnt[3 * NT + tid] = n_fail;
    }
    __syncthreads();
    // the per-env sums are the producer lanes' job when they have a wave of their own (it is idle by now)
    if (__builtin_expect(R.valid && R.node == 0, 0)) {
        const int ri = R.lidx;
        const u32 tick0 = has_producers ? d.tick[R.env] : c0;
        const u32 steps0 = has_producers ? d.total_steps[R.env] : steps_before;
        const u32 step_now = d.cur_step[R.env] + 1;  // :125
        double r = 0.0, t = 0.0;
        u32 c4[4] = {0, 0, 0, 0};
        for (int i = 0; i < N; i++) {  // node order, as the reference sums
            r = dadd(r, S.rew[ri + i]);
            t = dadd(t, S.ct[ri + i]);
#pragma unroll
            for (int q = 0; q < 4; q++) c4[q] += S.cnt[q * NT + ri + i];
        }
        d.cur_step[R.env] = step_now;
        d.total_steps[R.env] = steps0 + 1;
        d.tick[R.env] = tick0 + (u32)d.S;
        if (reward) reward[R.env] = (float)r;
        if (completion_time) completion_time[R.env] = (float)t;
        if (counters) *reinterpret_cast<int4*>(counters + 4 * R.env) = make_int4((int)c4[0], (int)c4[1], (int)c4[2], (int)c4[3]);
        if (terminated) terminated[R.env] = step_now >= (u32)d.T ? 1 : 0;  // :139-142
    }

    if (__builtin_expect(has_producers && obs_direct && !last && R.valid && producer_wave, 0)) {
        // Part 1 of the NEXT step, now: the cursors are final, the failure window is dead (phase E in its direct form
        // does not borrow it) and the main lanes have taken this step's task doubles, so the memory round trip of the
        // next early window runs under phases D and E instead of ahead of the next step's first barrier.  The task
        // cursor moves by one double, or by four when a task arrives (generate_task's binomial on this step's first
        // double, redone here; a redraw is left to the next step's own part 1, as is a stream not regenerated far enough).
        const int ri = R.lidx;
        const u32* rnc = S.nodec + R.node * NODEC_W;  // (filed ahead of the barrier above even in the launch's first step)
        const double U = S.tu[0 * NT + ri], qn = EVX ? *reinterpret_cast<const double*>(rnc + 0) : d.bin_qn[R.node];
        int X = -1;
        if (!(U > qn)) X = 0;
        else if (!(U - qn > (EVX ? *reinterpret_cast<const double*>(rnc + 2) : d.bin_px2[R.node]))) X = 1;
        u32 staged_word = 0;
        if (X >= 0) {
            const u32 arrival = (EVX ? rnc[10] : (u32)d.bin_flip[R.node]) ? 1u - (u32)X : (u32)X;
            const u32 fposn = S.rc[RC_FNEXT * NT + ri], tposn = mt_wrap(prod_tpos + (arrival ? 8u : 2u));
            const u32* fw = d.mt + ((size_t)P + (size_t)R.lane) * MT_STRIDE + fposn;  // wrap-free thanks to the mirror
            const u32* tw = d.mt + (size_t)R.lane * MT_STRIDE + tposn;
            const u32 hthr = EVX ? rnc[15] : d.hot_thr[R.node];
            u32x4 fv[RB_EARLY / 2];
#pragma unroll
            for (int i = 0; i < RB_EARLY / 2; i++) fv[i] = *reinterpret_cast<const u32x4*>(fw + 4 * i);
            const u32x4 t0 = *reinterpret_cast<const u32x4*>(tw), t1 = *reinterpret_cast<const u32x4*>(tw + 4);
            u32 hot16 = 0;
#pragma unroll
            for (int i = 0; i < RB_EARLY / 2; i++) {
                const u32 a0 = mt_temper(fv[i].x) >> 5, a1 = mt_temper(fv[i].z) >> 5;
                S.fa[(2 * i) * NT + ri] = a0;
                S.fa[(2 * i + 1) * NT + ri] = a1;
                hot16 |= (a0 >= hthr ? 1u : 0u) << (2 * i);
                hot16 |= (a1 >= hthr ? 1u : 0u) << (2 * i + 1);
            }
            S.tu[0 * NT + ri] = mt_to_double(t0.x, t0.y);
            S.tu[1 * NT + ri] = mt_to_double(t0.z, t0.w);
            S.tu[2 * NT + ri] = mt_to_double(t1.x, t1.y);
            S.tu[3 * NT + ri] = mt_to_double(t1.z, t1.w);
            S.rc[RC_PFPOS * NT + ri] = fposn;
            S.rc[RC_PTPOS * NT + ri] = tposn;
            staged_word = (u32)RB_EARLY | (hot16 << STAGED_HOT_SHIFT);
        }
        S.rc[RC_PSTAGED * NT + ri] = staged_word;
    }
    EC_STAMP(5);
    // ---------------------------------------------------------------- phase D: generate_task (:127-128) + write-back
    double wait_sum = 0.0;
    Rec nnew;  // generate_task always clears new_task first (edge_node.py:160): the old record is dead by now
    nnew.size = nnew.cycles = 0;
    nnew.twt = nnew.ctt = nnew.hop = nnew.ntx = 0;
    nnew.bwt = 0;
    nnew.enq = 0;
    if (L.valid) {
        // past the barrier above the producer has finished: its regeneration counts, and the streams are ours
        if (!f_synced) frng.ahead += S.rc[RC_FDELTA * NT + tid];
        RngTask trng;
        trng.attach(d.mt + (size_t)lane * MT_STRIDE, S.rc[RC_TPOS * NT + tid],
                    S.rc[RC_TAHEAD * NT + tid] + S.rc[RC_TDELTA * NT + tid], tu0, tu1, tu2, tu3);
        generate_task(d, lane, bin_qn, bin_px2, bin_flip, trng, buf, n_off, has_new, nnew);
        for (u32 j0 = 0; j0 < wq_len; j0 += 4) {  // four entries per LDS round trip; the sum stays in queue order
            double et[4];
#pragma unroll
            for (u32 u = 0; u < 4; u++) et[u] = j0 + u < wq_len ? S.wqe[((wq_h + j0 + u) & wmask) * NT + tid] : 0.0;
#pragma unroll
            for (u32 u = 0; u < 4; u++)
                if (j0 + u < wq_len) wait_sum = dadd(wait_sum, et[u]);  // get_waiting_time, edge_node.py:99-104
        }
        if (__builtin_expect(last && wq_dirty, 0)) {
            for (u32 j = 0; j < wq_len; j++) {
                const u32 pos = (wq_h + j) & wmask;
                d.wqm[(size_t)j * P + lane] = S.wqm[pos * NT + tid];
                d.wqe[(size_t)j * P + lane] = S.wqe[pos * NT + tid];
            }
        }
        if (__builtin_expect(last, 0)) {
            d.has_new[lane] = has_new;
            if (has_new) store_rec(d.newt + lane, nnew);
            d.n_off[lane] = n_off;
            d.buf_head[lane] = buf.head;
            d.buf_len[lane] = buf.len;
            d.wq_len[lane] = wq_len;
            d.ex_n[lane] = ex_n;
#pragma unroll
            for (int s = 0; s < KMAX; s++) {
                if ((u32)s < ex_n) {
                    // cet as the next step's first call (index c0 + S) will find it
                    const u32 fin = exh[s] >> 16, cet = c0 + (u32)d.S + fin - 1 - exf[s];
                    d.exA[(size_t)s * P + lane] = (u64)(cet & 0xffffu) | ((u64)fin << 16) | ((u64)(exh[s] & EXH_HOP) << 32) |
                                                  ((exh[s] & EXH_SUCCESS) ? 1ull << 48 : 0ull);
                    d.exB[(size_t)s * P + lane] = exT[s];
                    d.exC[(size_t)s * P + lane] = S.exC[s * NT + tid];
                }
            }
#pragma unroll
            for (int s = 0; s < DMAX; s++) {
                if (s < deg) {
                    const size_t so = (size_t)s * P + lane;
                    const u32 qh = S.rqhead[s * NT + tid];
                    d.rq_len[so] = rql[s];
                    d.rq_head[so] = qh;
                    d.rq_arrive[so] = rqa[s];
                    d.rq_hstart[so] = S.rqhs[s * NT + tid];
                    if (rql[s] > 0 && S.rqdirty[s * NT + tid]) {
                        uint4* q = reinterpret_cast<uint4*>(d.rq + ((size_t)s * d.capR + qh) * P + lane);
                        q[0] = S.rqrec[(s * 2 + 0) * NT + tid];
                        q[1] = S.rqrec[(s * 2 + 1) * NT + tid];
                    }
                }
            }
        } else {
            // the next step of this launch continues from registers and LDS
            cr.c0 = c0 + (u32)d.S;
            cr.steps_before = steps_before + 1;
            cr.has_new = has_new;
            cr.n_off = n_off;
            cr.ex_n = ex_n;
            cr.wq_h = wq_h;
            cr.wq_len = wq_len;
            cr.wq_min = wq_min;
            cr.wq_dirty = wq_dirty;
            if (em) em->s_cnt[tid] = wq_len | (ex_n << 16);  // (read by the next step's expert, past the loop's barrier)
            cr.exA_thr = exA_thr;
            cr.deg = deg;
            cr.cap = cap;
            cr.newt = nnew;
            cr.buf_head = buf.head;
            cr.buf_len = buf.len;
#pragma unroll
            for (int s = 0; s < DMAX; s++) {
                cr.lkA[s] = lkA[s];
                if (s < deg) {  // the pushes of the next step's phase A look at these
                    S.rqlen[s * NT + tid] = rql[s];
                    S.rqarr[s * NT + tid] = rqa[s];
                }
            }
#pragma unroll
            for (int s = 0; s < KMAX; s++) {
                cr.exf[s] = exf[s];
                cr.exh[s] = exh[s];
                cr.exT[s] = exT[s];
            }
        }
        u32 pos, ahead, used;
        trng.close(pos, ahead, used);
        if (__builtin_expect(last, 0)) {
            d.rng_pos[lane] = pos;
            d.rng_ahead[lane] = ahead;
        } else {  // the producer lane of the next step reads them here (after the step's closing barrier)
            S.rc[RC_TPOS * NT + tid] = pos;
            S.rc[RC_TAHEAD * NT + tid] = ahead;
        }
        d.rng_cnt[lane] = cnt_t + 2ull * used;
        frng.close(pos, ahead, used);
        if (__builtin_expect(last, 0)) {
            d.rng_pos[P + lane] = pos;
            d.rng_ahead[P + lane] = ahead;
        } else {
            S.rc[RC_FPOS * NT + tid] = pos;
            S.rc[RC_FAHEAD * NT + tid] = ahead;
        }
        d.rng_cnt[P + lane] = cnt_f + 2ull * used;
        if (__builtin_expect((ovf | lut.ovf) != 0, 0)) flag(d, EC_ST_TICK_OVERFLOW);
    }
    EC_STAMP(6);
    EC_STAMP(7);
    // ---------------------------------------------------------------- phase E: fused observation of the new state
#ifdef ABL_NO_OBS
    if (false) {
#else
    if (__builtin_expect(obs_direct, 1)) {  // (the staged form below goes out of line)
#endif
        // edge_node.py:109-136,179-197 per lane (see observe_block for the staged form of the same arithmetic)
        const bool full = L.valid && (double)wq_len >= o_qthr;  // :186-188
        const u64 ball = __ballot(full);
        const u64 fullbits = (ball >> ((threadIdx.x & 63u) - (u32)node)) & (N >= 64 ? ~0ull : (1ull << N) - 1ull);
        if (L.valid) {
            const u64 room = o_adjb & ~fullbits;  // adjacent (or own) queues with room
            const u64 amask = has_new ? (room ? room : 1ull << node) : 0ull;  // bits of the actions 0..N-1; idle = !has_new
            {
                int32_t* out = next_avail + lane * d.A;
                int j = 0;
                for (; j + 4 <= d.A; j += 4) {
                    i32x4u v;
                    v.x = (int)((amask >> j) & 1ull);
                    v.y = (int)((amask >> (j + 1)) & 1ull);
                    v.z = (int)((amask >> (j + 2)) & 1ull);
                    v.w = (int)((amask >> (j + 3)) & 1ull);
                    if (j + 3 >= N) {  // the chunk holds the idle action (index N)
                        const int idle = has_new ? 0 : 1;
                        v.x = j == N ? idle : v.x;
                        v.y = j + 1 == N ? idle : v.y;
                        v.z = j + 2 == N ? idle : v.z;
                        v.w = j + 3 == N ? idle : v.w;
                    }
                    *reinterpret_cast<i32x4u*>(out + j) = v;
                }
                for (; j < d.A; j++) out[j] = j < N ? (int)((amask >> j) & 1ull) : (has_new ? 0 : 1);
            }
            if (next_obs) {
                f32x4u a, b;
                a.x = o_c0;
                a.y = o_c1;
                a.z = o_c2;
                a.w = (float)ddiv((double)wq_len, o_qlen);
                b.x = (float)ddiv(wait_sum, d.deadline);
                b.y = has_new ? (float)ddiv(nnew.size, d.obs_size_norm) : -1.0f;
                b.z = has_new ? (float)ddiv(nnew.cycles, d.obs_cyc_norm) : -1.0f;
                b.w = has_new ? 1.0f : -1.0f;  // task_deadline / deadline (edge_node.py:130): the same number over itself
                float* out = next_obs + lane * 9;
                *reinterpret_cast<f32x4u*>(out) = a;
                *reinterpret_cast<f32x4u*>(out + 4) = b;
                out[8] = has_new ? (float)nnew.hop : 0.0f;
            }
            if (pol.next_actions) {
                // the uniform-random policy's draw for the next step from the mask just written (same generator and row
                // numbering as ec_random_policy); every env of the handle has taken the same number of steps
                int act = N;  // no task: idle is the only available action
                if (has_new) {
                    const u64 z = policy_hash(pol.seed, pol.counter + (u64)steps_before + 1, (u64)P, (u64)lane);
                    u64 m = amask;
                    int pick = (int)(((z >> 32) * (u64)__popcll(m)) >> 32);
                    while (pick-- > 0) m &= m - 1;  // drop the `pick` lowest set bits
                    act = __ffsll((unsigned long long)m) - 1;
                }
                pol.next_actions[lane] = act;
                cr.action = act;
            }
        }
    } else if (next_obs || next_avail) {
        // the failure-stream window is dead since the barrier of phase C: reuse it as observation scratch
        u32* s_full = S.fa;
        u32* s_hasnew = s_full + NT;
        u32* s_any = s_hasnew + NT;
        float* s_obs = reinterpret_cast<float*>(s_any + NT);
        u64* s_fullbits = (d.epw > 0 && d.adj_bits) ? S.fullbits : nullptr;
        observe_block(d, L, wq_len, wait_sum, has_new, nnew, next_obs, next_avail, s_full, s_hasnew, s_any, s_obs,
                      s_fullbits);
        if (pol.next_actions) {
            // s_full / s_hasnew / s_any still hold the mask's ingredients (edge_node.py:179-197)
            // every env of the handle has taken the same number of steps: a launch counter without atomics
            const u64 counter = pol.counter + (u64)steps_before + 1;
            if (L.valid) {
                int act = N;  // no task: idle is the only available action
                if (has_new) {
                    const u32* full = s_full + L.env_local * N;
                    const u64 z = policy_hash(pol.seed, counter, (u64)P, (u64)lane);
                    if (d.adj_bits) {
                        // the mask as bits: neighbours and self with room, else self (edge_node.py:179-197)
                        u64 fullbits = 0;
                        if (s_fullbits) {
                            fullbits = s_fullbits[L.env_local];
                        } else {
                            for (int j = 0; j < N; j++) fullbits |= (u64)(full[j] != 0) << j;
                        }
                        u64 m = d.adj_bits[node] & ~fullbits;
                        if (!m) m = 1ull << node;
                        const int cnt = __popcll(m);
                        int pick = (int)(((z >> 32) * (u64)cnt) >> 32);
                        while (pick-- > 0) m &= m - 1;  // drop the `pick` lowest set bits
                        act = __ffsll((unsigned long long)m) - 1;
                    } else {
                        const bool any = s_any[tid] != 0;
                        int cnt = 0;
                        for (int j = 0; j < N; j++) cnt += ((d.adj[node * d.A + j] && !full[j]) || (!any && j == node)) ? 1 : 0;
                        int pick = cnt > 0 ? (int)(((z >> 32) * (u64)cnt) >> 32) : 0;
                        act = d.A - 1;
                        for (int j = 0; j < N; j++) {
                            if ((d.adj[node * d.A + j] && !full[j]) || (!any && j == node)) {
                                if (pick == 0) {
                                    act = j;
                                    break;
                                }
                                pick--;
                            }
                        }
                    }
                }
                pol.next_actions[lane] = act;
                cr.action = act;
            }
        }
    }
    EC_STAMP(8);
#ifdef EC_STAMPS
    if (d.stamps && (threadIdx.x == 0 || (int)threadIdx.x == d.nmain)) {  // where the two waves run: HW_ID | XCC_ID << 32
        const u64 hw = (u64)__builtin_amdgcn_s_getreg(63492) | ((u64)__builtin_amdgcn_s_getreg(63508) << 32);
        d.stamps[(size_t)blockIdx.x * 16 + (threadIdx.x == 0 ? 14 : 15)] = hw;
    }
#endif
}

// One env step per launch (ec_step, ec_step_random_policy), or -- MULTI -- pol0.n_steps of them in one launch
// (ec_rollout_random_policy): each block carries its own envs through all the steps, so there is no launch gap and
// the blocks drift apart instead of hitting memory in lock step.  State goes through global memory between
// iterations exactly as between launches; a block only ever re-reads what it wrote itself (workgroup-scope
// ordering: the barrier).  The MULTI loop re-reads the geometry/pointer block from memory every iteration through
// a pointer the compiler cannot see through: held in registers across the whole body instead, its ~150 words
// double the spills of the step code.
template <int KMAX, int DMAX, bool MULTI, int NTC = 0>
__global__ void __launch_bounds__(256, 2) k_step(EcDev d_arg, const EcDev* d_mem, const int64_t* actions0, float* reward0,
                                              uint8_t* terminated0, int32_t* counters0, float* completion_time0,
                                              float* next_obs0, int32_t* next_avail0, StepPolicy pol0) {
    StepCarry<KMAX, DMAX> cr;
    if (!MULTI) {
        step_body<KMAX, DMAX, false, NTC>(d_arg, actions0, reward0, terminated0, counters0, completion_time0, next_obs0, next_avail0,
                              pol0, cr, true, true);
        return;
    }
#ifdef EC_STAMPS
    if (threadIdx.x == 0 && d_arg.stamps) d_arg.stamps[(size_t)blockIdx.x * 16 + 13] = wall_clock64();  // launch entry
#endif
#ifdef EC_STAMPS
    u64 dbg_prev = wall_clock64(), dbg_gap = 0, dbg_body = 0;
#endif
    // the geometry block in the lanes of three vector registers; its members are read where they are used (see EcDevL)
    static_assert(DEV_VREGS == 3, "the asm statement below names the registers");
    u32 dv[DEV_VREGS];
    dev_lanes_load(dv, d_mem);
    // One step.  The launch's first step is a copy of its own ahead of the loop (`first` is a literal at the two call
    // sites): the loop's copy of the step has no staging code in it, and what a lane carries from step to step then
    // stays in registers -- with the staging code inside the loop 20 of the carried registers went through scratch
    // memory at every step boundary (17.26 -> 16.93 us per step, 124 -> 0 B/lane of scratch; peeling the last step as
    // well: nothing more, twice the code again).
    const auto one_step = [&](int it, const bool first, const bool last) __attribute__((always_inline)) {
        // (re-read in every step as far as the compiler can tell: nothing derived from the block is hoisted out of the
        // loop and kept live across the step, which doubles the step code's spills)
        asm volatile("" : "+v"(dv[0]), "+v"(dv[1]), "+v"(dv[2]));
        const EcDevL d(dv);
        const int64_t hs = pol0.keep ? (int64_t)it : 0;  // history slot of this iteration's outputs
        StepPolicy pol = pol0;
        pol.next_actions += hs * d.P;
#ifdef EC_STAMPS
        const u64 dbg_t0 = wall_clock64();
        dbg_gap += dbg_t0 - dbg_prev;
#endif
        step_body<KMAX, DMAX, true, NTC>(d, first ? actions0 : pol0.next_actions + (pol0.keep ? (int64_t)(it - 1) * (int64_t)d.P : 0),
                              reward0 ? reward0 + hs * d.B : nullptr, terminated0 ? terminated0 + hs * d.B : nullptr,
                              counters0 ? counters0 + hs * d.B * 4 : nullptr,
                              completion_time0 ? completion_time0 + hs * d.B : nullptr,
                              next_obs0 ? next_obs0 + hs * d.P * 9 : nullptr, next_avail0 + hs * d.P * d.A, pol, cr,
                              first, last);
#ifdef EC_STAMPS
        dbg_prev = wall_clock64();
        dbg_body += dbg_prev - dbg_t0;
#endif
        __syncthreads();  // the step is complete in LDS and (RNG cursors, MT words) in this block's global memory
    };
    one_step(0, true, pol0.n_steps == 1);
    for (int it = 1; it < pol0.n_steps; it++) one_step(it, false, it + 1 == pol0.n_steps);
#ifdef EC_STAMPS
    if (threadIdx.x == 0 && d_arg.stamps) {  // per block: time between step bodies (barrier + geometry reload) and inside them
        d_arg.stamps[(size_t)blockIdx.x * 16 + 9] = dbg_gap;
        d_arg.stamps[(size_t)blockIdx.x * 16 + 10] = dbg_body;
    }
#endif
}

// ec_rollout_mlp_policy: the looping kernel with the Actor evaluated inside the launch.  Per step: every main wave
// takes the rows of its own lanes through polinl::policy_rows (the observation and mask the previous step wrote, or
// the caller's initial ones), then the step runs.  Outputs are always histories ([n_steps, ...]).
struct MlpRollout {
    const float* staged;    // polinl layout
    const float* uniforms;  // [n_steps, B*N] or null (greedy)
    const float* obs0;      // [B,N,9]   observation the first action is chosen on
    const int32_t* avail0;  // [B,N,A]
    int n_steps;
    unsigned b23_lds;       // byte offset, past the step's LDS, of the layer-2/3 biases ([64] + [32] f32)
};

struct ExpertRollout {
    const float* coin;     // [n_steps][B]
    const float* ratios;   // [n_steps]
    int64_t* expert_hist;  // [n_steps][P]
};

// TO3: output tiles of the Actor's last layer (1: up to 32 actions, 2: up to 64)
template <int TO3, int G>
__device__ __forceinline__ int rollout_policy_rows(const float* __restrict__ st, const float* s_b23, const float* __restrict__ obs,
                                                   const int32_t* __restrict__ avail, const float* __restrict__ uniforms,
                                                   int64_t* __restrict__ actions, int64_t row0, int cnt, int n_agents, int A,
                                                   int lane POL_ST_ARG) {
#ifdef EC_STAMPS_POLICY
    if constexpr (TO3 == 1)
        if (A <= 16) return polinl::policy_rows<2, G>(st, s_b23, obs, avail, uniforms, actions, row0, cnt, n_agents, A, lane, pst);
#endif
    if constexpr (TO3 == 1)
        return __builtin_expect(A <= 16, 1) ? polinl::policy_rows<2, G>(st, s_b23, obs, avail, uniforms, actions, row0, cnt, n_agents, A, lane)
                                            : polinl::policy_rows<4, G>(st, s_b23, obs, avail, uniforms, actions, row0, cnt, n_agents, A, lane);
    else
        return A <= 48 ? polinl::policy_rows<6, G, 2>(st, s_b23, obs, avail, uniforms, actions, row0, cnt, n_agents, A, lane)
                       : polinl::policy_rows<8, G, 2>(st, s_b23, obs, avail, uniforms, actions, row0, cnt, n_agents, A, lane);
}

template <int KMAX, int DMAX, bool EXPERT, int TO3, int NTC = 0>
__global__ void __launch_bounds__(256, 2) k_rollout_mlp(const EcDev* d_mem, MlpRollout m, int64_t* actions0, float* reward0,
                                                     uint8_t* terminated0, int32_t* counters0, float* completion_time0,
                                                     float* next_obs0, int32_t* next_avail0, ExpertRollout er) {
    StepCarry<KMAX, DMAX> cr;
#ifdef EC_STAMPS
    u64 dbg_pol = 0, dbg_wait = 0, dbg_body = 0, dbg_gap = 0, dbg_prev = wall_clock64();
    u64* dbg_out = nullptr;
#endif
#ifdef EC_STAMPS_POLICY
    unsigned long long dbg_ps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define POL_PST , dbg_ps
#else
#define POL_PST
#endif
    extern __shared__ __align__(16) unsigned char smem_all[];
    float* s_b23 = reinterpret_cast<float*>(smem_all + m.b23_lds);
    int* s_pick = reinterpret_cast<int*>(s_b23 + polinl::H + 32 * TO3);  // [main lanes] actions handed over by producer waves
    u32* s_cnt = reinterpret_cast<u32*>(s_pick + d_mem->nmain);    // [main lanes] queue lengths for the expert (EXPERT)
    for (int i = threadIdx.x; i < polinl::H + 32 * TO3; i += blockDim.x) s_b23[i] = m.staged[polinl::off_B2(TO3) + i];
    u32 dv[DEV_VREGS];  // the geometry block, as in k_step
    dev_lanes_load(dv, d_mem);
    __syncthreads();
    const auto one_step = [&](const int it, const bool first) __attribute__((always_inline)) {
        asm volatile("" : "+v"(dv[0]), "+v"(dv[1]), "+v"(dv[2]));
        const EcDevL d(dv);
        int64_t* act = actions0 + (int64_t)it * d.P;
        // (see the Actor evaluation below) rows beyond a main wave's first 32 go to the producer waves
        const bool split = (int)blockDim.x > d.nmain && (d.epw > 0 ? d.epw * d.N : 64) > 32;
#ifdef EC_STAMPS
        dbg_out = d.stamps;
        const u64 dbg_t0 = wall_clock64();
        dbg_gap += dbg_t0 - dbg_prev;
#endif
        {
            // Every main wave evaluates the Actor on the rows of its own lanes (lane l of the wave <-> row r0 + l of
            // the block), so the action lands in the lane that takes it: no hand-off and no barrier, and the RNG
            // producer waves go straight into the step to stage its draws while the policy runs.
            const float* ob = first ? m.obs0 : next_obs0 + (int64_t)(it - 1) * d.P * 9;
            const int32_t* av = first ? m.avail0 : next_avail0 + (int64_t)(it - 1) * d.P * d.A;
            const float* un = m.uniforms ? m.uniforms + (int64_t)it * d.P : nullptr;
            int pick = -1;
            const int64_t row_base = (int64_t)blockIdx.x * d.epb * d.N;  // the block's rows are contiguous (lane_id)
            const int64_t left = d.P - row_base;
            const int nrows = left < (int64_t)d.epb * d.N ? (int)left : d.epb * d.N;
            const int per = d.epw > 0 ? d.epw * d.N : 64;  // rows of a main wave (lane_id_main)
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            const int nmw = d.nmain >> 6, npw = ((int)blockDim.x - d.nmain) >> 6;
            // With RNG producer waves in the block the rows of a main wave beyond its first 32 (one 32-row MFMA group
            // is what a wave evaluates in one go) are the producer waves' job, at the same time on another SIMD; the
            // actions reach their lanes through LDS behind the step's first barrier.  Same arithmetic per row either
            // way (policy_rows<., 1> == one group of policy_rows<., 2>).
            if (wave < nmw) {  // wave-uniform
                const int r0 = wave * per;
                int cnt = nrows - r0 < per ? nrows - r0 : per;
                if (split && cnt > 32) cnt = 32;
                if (__builtin_expect(cnt > 32, 0)) {  // (geometries without producer waves; out of line)
                    pick = rollout_policy_rows<TO3, 2>(m.staged, s_b23, ob, av, un, act, row_base + r0, cnt, d.N, d.A, lane);
                } else if (cnt > 0) {
                    pick = rollout_policy_rows<TO3, 1>(m.staged, s_b23, ob, av, un, act, row_base + r0, cnt, d.N, d.A, lane POL_PST);
                }
            } else if (split) {
                for (int w = wave - nmw; w < nmw; w += npw) {  // the remainders of main waves w, w + npw, ...
                    const int r0 = w * per + 32;
                    const int cnt = nrows - r0 < per - 32 ? nrows - r0 : per - 32;
                    if (cnt <= 0) continue;
                    const int pk = rollout_policy_rows<TO3, 1>(m.staged, s_b23, ob, av, un, act, row_base + r0, cnt, d.N, d.A, lane);
                    if (lane < cnt) s_pick[r0 + lane] = pk;  // index = the taking lane's slot (lane_id_main: w * per + l)
                }
            }
            if (!first) cr.action = pick;  // the first step reads act[] (this lane's own store); split rows: s_pick
        }
#ifdef EC_STAMPS
        const u64 dbg_t1 = wall_clock64();
        const u64 dbg_t2 = dbg_t1;
#endif
        const int64_t hs = it;
        ExpertMix emx;
        const ExpertMix* em = nullptr;
        if (EXPERT) {
            emx.avail = first ? m.avail0 : next_avail0 + (int64_t)(it - 1) * d.P * d.A;
            emx.coin = er.coin + (int64_t)it * d.B;
            emx.ratio = er.ratios[it];
            emx.expert_out = er.expert_hist + (int64_t)it * d.P;
            emx.s_cnt = s_cnt;
            em = &emx;
        }
        step_body<KMAX, DMAX, true, NTC>(d, act, reward0 ? reward0 + hs * d.B : nullptr,
                                    terminated0 ? terminated0 + hs * d.B : nullptr,
                                    counters0 ? counters0 + hs * d.B * 4 : nullptr,
                                    completion_time0 ? completion_time0 + hs * d.B : nullptr, next_obs0 + hs * d.P * 9,
                                    next_avail0 + hs * d.P * d.A, StepPolicy{0, 0, nullptr, m.n_steps, 1}, cr, first,
                                    it + 1 == m.n_steps, split ? s_pick : nullptr, em);
#ifdef EC_STAMPS
        dbg_prev = wall_clock64();
        dbg_pol += dbg_t1 - dbg_t0;
        dbg_wait += dbg_t2 - dbg_t1;
        dbg_body += dbg_prev - dbg_t2;
#endif
        __syncthreads();
    };
    // (one copy of the step here: with the first step peeled as in k_step this kernel measured 0.5 us per step slower)
    for (int it = 0; it < m.n_steps; it++) one_step(it, it == 0);
#ifdef EC_STAMPS
    if (threadIdx.x == 0 && dbg_out) {  // wave 0's totals: policy tile, wait for the other tiles + action, step, loop gap
        dbg_out[(size_t)blockIdx.x * 16 + 9] = dbg_pol;
        dbg_out[(size_t)blockIdx.x * 16 + 10] = dbg_wait;
        dbg_out[(size_t)blockIdx.x * 16 + 11] = dbg_body;
        dbg_out[(size_t)blockIdx.x * 16 + 12] = dbg_gap;
#ifdef EC_STAMPS_POLICY  // (over the last step's phase stamps)
        for (int i = 0; i < 5; i++) dbg_out[(size_t)blockIdx.x * 16 + i] = dbg_ps[i];
#endif
    }
#endif
}

typedef void (*rollout_mlp_fn)(const EcDev*, MlpRollout, int64_t*, float*, uint8_t*, int32_t*, float*, float*, int32_t*,
                               ExpertRollout);
#ifndef EC_NT_DEFAULT
#define EC_NT_DEFAULT 40
#endif
constexpr int NT_DEFAULT = EC_NT_DEFAULT;  // lanes per block of the default size's geometry (4 envs x 10 nodes): specialised instantiations
rollout_mlp_fn pick_rollout_mlp(int k, int D, int A, bool expert, int nt) {  // the common shapes only: every instantiation is a full step kernel
    if (k > 4 || A > 64) return nullptr;
    if (nt == NT_DEFAULT && A <= 32 && D <= 4)
        return expert ? k_rollout_mlp<4, 4, true, 1, NT_DEFAULT> : k_rollout_mlp<4, 4, false, 1, NT_DEFAULT>;
#ifndef EC_DEV_BUILD
    // (the geometries a 10-node job takes at 1 / 2 envs per block: up to 1024 / 2048 envs per GPU, the shards of a
    // multi-GPU run; 5 % per step at 1024 envs)
    if (nt == 10 && A <= 32 && D <= 4) return expert ? k_rollout_mlp<4, 4, true, 1, 10> : k_rollout_mlp<4, 4, false, 1, 10>;
    if (nt == 20 && A <= 32 && D <= 4) return expert ? k_rollout_mlp<4, 4, true, 1, 20> : k_rollout_mlp<4, 4, false, 1, 20>;
#endif
#ifdef EC_DEV_BUILD
    if (A > 32 || D > 4) return nullptr;
    return expert ? k_rollout_mlp<4, 4, true, 1> : k_rollout_mlp<4, 4, false, 1>;
#endif
    if (A > 32)  // 33..64 actions (the 40-node topology of BASELINE configs[4]): in-degree <= 4 only
        return D > 4 ? nullptr : (expert ? k_rollout_mlp<4, 4, true, 2> : k_rollout_mlp<4, 4, false, 2>);
    if (D <= 4) return expert ? k_rollout_mlp<4, 4, true, 1> : k_rollout_mlp<4, 4, false, 1>;
    if (D <= 8) return expert ? k_rollout_mlp<4, 8, true, 1> : k_rollout_mlp<4, 8, false, 1>;
    return nullptr;
}

typedef void (*step_fn)(EcDev, const EcDev*, const int64_t*, float*, uint8_t*, int32_t*, float*, float*, int32_t*, StepPolicy);
template <int KMAX, bool MULTI>
step_fn pick_step_d(int D) {
    if (D <= 4) return k_step<KMAX, 4, MULTI>;
#ifndef EC_DEV_BUILD  // (make dev: only the <4, 4> instantiations, a quarter of the compile time while tuning)
    if (D <= 8) return k_step<KMAX, 8, MULTI>;
#endif
    return nullptr;
}
template <bool MULTI>
step_fn pick_step(int k, int D, int* kmax, int* dmax, int nt = 0) {
    *dmax = D <= 4 ? 4 : 8;
    if (k <= 4 && D <= 4 && nt == NT_DEFAULT) return *kmax = 4, k_step<4, 4, MULTI, NT_DEFAULT>;
#ifndef EC_DEV_BUILD
    if (k <= 4 && D <= 4 && nt == 10) return *kmax = 4, k_step<4, 4, MULTI, 10>;
    if (k <= 4 && D <= 4 && nt == 20) return *kmax = 4, k_step<4, 4, MULTI, 20>;
#endif
    if (k <= 4) return *kmax = 4, pick_step_d<4, MULTI>(D);
#ifndef EC_DEV_BUILD
    if (k <= 8) return *kmax = 8, pick_step_d<8, MULTI>(D);
    if (k <= 12) return *kmax = 12, pick_step_d<12, MULTI>(D);
#endif
    return nullptr;
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
// (expert policy: expert_score / expert_decide are defined ahead of step_body, which evaluates them inside the rollout kernel)
__global__ void k_expert(EcDev d, const int32_t* __restrict__ avail, int64_t* __restrict__ actions) {
    const int64_t lane = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= d.P) return;
    const int64_t env = lane / d.N;
    const int i = (int)(lane - env * d.N);
    const int64_t base = env * d.N;
    const u32 has_new = d.has_new[lane];
    double size = 0.0, cycles = 0.0;
    if (has_new) {
        const Rec nt = load_rec(d.newt + lane);
        size = nt.size;
        cycles = nt.cycles;
    }
    actions[lane] = expert_decide(d, i, avail + lane * d.A, has_new, size, cycles, [&](int t, int& nw, int& nx) {
        nw = (int)d.wq_len[base + t];
        nx = (int)d.ex_n[base + t];
    });
}

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
        for (u32 j = 0; j < d.wq_len[lane]; j++) {
            const WMeta w = wmeta_unpack(d.wqm[(size_t)j * d.P + lane]);
            // size/cycles are not kept once a task is queued for execution: report execute_time and 0
            put(i, 2, j, d.wqe[(size_t)j * d.P + lane], 0.0, w.hop, now - w.enq, 0, w.twt, w.ctt, w.bwt);
        }
        for (u32 j = 0; j < d.ex_n[lane]; j++) {
            const u64 a = d.exA[(size_t)j * d.P + lane], b = d.exC[(size_t)j * d.P + lane];
            // cols 3,4: the precomputed finish tick and the task_time at that tick
            put(i, 3, j, (double)f16(a, 1), d.exB[(size_t)j * d.P + lane], f16(a, 2), f16(b, 0), f16(a, 0), f16(b, 1),
                f16(b, 2), f16(b, 3));
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
    int threads_obs;  // block size of k_observe (the block's lanes, contiguous)
    int64_t blocks;
    size_t lds_step, lds_obs;
    size_t lds_step_links;  // lds_step + the link tables (k_rollout_mlp)
    step_fn step;        // one env step per launch
    step_fn step_multi;  // the same kernel looping over steps (ec_rollout_random_policy)
    EcDev* d_dev;        // device copy of `d` for the looping kernel
    bool multi_resident; // every block of the looping kernel fits on the device at once
    rollout_mlp_fn rollout_mlp;  // looping kernel with the Actor inside (ec_rollout_mlp_policy); null if not compiled for k / D
    rollout_mlp_fn rollout_mlp_expert;  // the same with the expert mixing of the guided runners (ec_rollout_mlp_expert_policy)
    float* d_staged;     // its weight fragments (polinl layout)
    size_t lds_rollout_mlp;
    int kmax, dmax;
    std::vector<void*> allocs;
    size_t state_bytes;
};

namespace {

int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

FastDiv make_div(int dvs) {
    FastDiv f;
    f.d = (u32)dvs;
    f.m = dvs == 1 ? 0u : (u32)(((1ull << 32) + (u64)dvs - 1) / (u64)dvs);
    return f;
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

// Every entry point that takes a handle runs on the handle's device and leaves the caller's current device as it
// found it (a handle built for cuda:1 may be driven from a thread whose current device is cuda:0).
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int device) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) {
            err = hipSetDevice(device);
            switched = err == hipSuccess;
        }
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};
#define EC_ON_DEVICE(h)                                                                         \
    DeviceGuard guard_((h)->device);                                                            \
    if (guard_.err != hipSuccess)                                                               \
        return ec::fail(EC_EHIP, "cannot switch to device %d: %s", (h)->device, hipGetErrorString(guard_.err))

#define EC_TRY(expr)          \
    do {                      \
        int rc_ = (expr);     \
        if (rc_) return rc_;  \
    } while (0)

// `random() > thr` as an integer test on the 53-bit numerator: M = floor(thr * 2^53)
u64 threshold_m(double thr) {
    if (!(thr > 0.0)) return 0;
    if (thr >= 1.0) return 1ull << 53;
    return (u64)floor(ldexp(thr, 53));
}

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
    d.inv_dt = 1.0 / cfg->mini_time_slot;
    d.deadline = cfg->deadline;
    d.dl_ticks = (int)(cfg->deadline / cfg->mini_time_slot);
    d.size_lo = cfg->task_size_min;
    d.size_range = cfg->task_size_max - cfg->task_size_min;  // numpy uniform: range = high - low
    d.cx_lo = cfg->task_complexity_min;
    d.cx_range = cfg->task_complexity_max - cfg->task_complexity_min;
    d.pen_drop = cfg->task_drop_penalty;
    d.pen_fail = cfg->task_failure_penalty;
    d.rew_succ = cfg->task_success_reward;
    d.pen_hop = cfg->task_hop_penalty;
    d.obs_size_norm = cfg->task_size_max * 1024 * 8;                             // edge_node.py:126
    d.obs_cyc_norm = cfg->task_complexity_max * cfg->task_size_max * 1024 * 8;  // edge_node.py:128
    d.divN = make_div(N);
    d.divA = make_div(N + 1);

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
    std::vector<u64> thr_link_m((size_t)N * D, 1ull << 53);
    for (int i = 0; i < N; i++) {
        int c = 0;
        for (int s = 0; s < N; s++) {
            if (s == i || !topo->adj[s * (N + 1) + i]) continue;
            if (topo->rate[s * N + i] == 0.0) return ec::fail(EC_EINVAL, "adjacent pair (%d,%d) has zero rate", s, i);
            in_src[i * D + c] = s;
            slot_of[s * N + i] = c;
            // edge_node.py:321: exp(-transmission_failure_rate * mini_time_slot), rate taken from the sender's table
            thr_link_m[i * D + c] = threshold_m(exp(-topo->tfr[s * N + i] * cfg->mini_time_slot));
            c++;
        }
    }

    // ring capacities: bounds that hold for any policy within one episode
    const int steps_per_deadline = (int)ceil(cfg->deadline / (cfg->mini_time_slot * cfg->mini_slots));
    d.capW = next_pow2(cfg->cap_wait > 0 ? cfg->cap_wait : steps_per_deadline + 3);
    d.capR = cfg->cap_recv > 0 ? cfg->cap_recv : next_pow2(cfg->episode_limit + 2);
    d.capB = cfg->cap_buffer > 0 ? cfg->cap_buffer : next_pow2((D + 1) * (cfg->episode_limit + 2));
    d.maxTicks = cfg->max_ticks > 0 ? cfg->max_ticks : cfg->mini_slots * (cfg->episode_limit + 2);
    if (d.maxTicks > 0xfff0) return ec::fail(EC_EINVAL, "max_ticks %d exceeds the 16-bit tick range", d.maxTicks);
    if (d.capB > 0xffff || d.capR > 0xffff) return ec::fail(EC_EINVAL, "ring capacity too large");

    std::vector<double> acc(d.maxTicks + 1);
    acc[0] = 0.0;
    for (int n = 1; n <= d.maxTicks; n++) acc[n] = acc[n - 1] + cfg->mini_time_slot;  // edge_node.py:216,250,254,316,319

    {   // Lut::deadline_ticks' closed form: |task_time - K dt| <= E for a task_time of K ticks in all, with
        // E = 5 max_n |acc[n] - n dt| (the five LUT entries) + 4 half-ulps of the largest partial sum (the four additions)
        long double emax = 0.0L;
        for (int n = 0; n <= d.maxTicks; n++) {
            const long double e = fabsl((long double)acc[n] - (long double)n * (long double)cfg->mini_time_slot);
            emax = e > emax ? e : emax;
        }
        const long double vmax = 5.0L * (long double)acc[d.maxTicks];
        const long double E = 5.0L * emax + 4.0L * vmax * 1.1102230246251565e-16L;
        const long double dtl = cfg->mini_time_slot, DL = cfg->deadline;
        d.fast_dl = (d.dl_ticks >= 2 && d.dl_ticks + 2 <= d.maxTicks && (d.dl_ticks - 1) * dtl + E <= DL &&
                     (d.dl_ticks + 2) * dtl - E > DL && !getenv("MARL_AMD_EC_SEARCH_DL")) ? 1 : 0;
    }

    std::vector<double> cap(N), qlen(N), qthr(N), bqn(N), bpx2(N), rate((size_t)N * N);
    std::vector<u64> thr_exec_m(N), adj_bits(N, 0);
    std::vector<float> obs_const((size_t)N * 3);
    std::vector<uint8_t> bflip(N), adj((size_t)N * (N + 1));
    for (int i = 0; i < N; i++) {
        cap[i] = topo->cpu_capacity[i];
        thr_exec_m[i] = threshold_m(exp(-topo->efr[i] * cfg->mini_time_slot));  // edge_node.py:257
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
        if (N <= 64)
            for (int j = 0; j < N; j++)
                if (topo->adj[i * (N + 1) + j]) adj_bits[i] |= 1ull << j;
    }
    memcpy(rate.data(), topo->rate, sizeof(double) * N * N);
    memcpy(adj.data(), topo->adj, (size_t)N * (N + 1));
    std::vector<double> efr(topo->efr, topo->efr + N), tfr(topo->tfr, topo->tfr + (size_t)N * N);
    std::vector<int> cores(topo->cores, topo->cores + N);

    {   // the tick LUT as 16-bit corrections to fl(n * dt) (see struct Lut)
        size_t n_ent = (size_t)d.maxTicks + 1;
        if (n_ent < (size_t)ACC_LDS_N) n_ent = ACC_LDS_N;
        n_ent = (n_ent + 3) & ~(size_t)3;
        d.acc_words = (int)(n_ent / 4);
        d.link_lds = N <= LINK_LDS_N ? 1 : 0;
        std::vector<int16_t> corr(n_ent, 0);
        for (int n = 0; n <= d.maxTicks; n++) {
            const double prod = (double)n * cfg->mini_time_slot;
            int64_t ba, bp;
            memcpy(&ba, &acc[n], 8);
            memcpy(&bp, &prod, 8);
            const int64_t c = ba - bp;
            if (c < -32768 || c > 32767)
                return ec::fail(EC_EINVAL, "tick %d: accumulated timer is %lld ulp from n * mini_time_slot (16-bit table)", n, (long long)c);
            corr[n] = (int16_t)c;
        }
        EC_TRY(dev_upload(h, &d.acc, corr));
    }
    EC_TRY(dev_upload(h, &d.cap, cap));
    EC_TRY(dev_upload(h, &d.thr_exec_m, thr_exec_m));
    {   // lowest first-word threshold (M >> 26) among a node's failure tests: below it a draw fails none of them
        std::vector<u32> hot_thr(N);
        for (int i = 0; i < N; i++) {
            u64 lo = thr_exec_m[i];
            for (int c = 0; c < deg[i]; c++) lo = thr_link_m[i * D + c] < lo ? thr_link_m[i * D + c] : lo;
            hot_thr[i] = (u32)(lo >> 26);
        }
        EC_TRY(dev_upload(h, &d.hot_thr, hot_thr));
    }
    EC_TRY(dev_upload(h, &d.qlen, qlen));
    EC_TRY(dev_upload(h, &d.qthr, qthr));
    EC_TRY(dev_upload(h, &d.obs_const, obs_const));
    EC_TRY(dev_upload(h, &d.bin_qn, bqn));
    EC_TRY(dev_upload(h, &d.bin_px2, bpx2));
    EC_TRY(dev_upload(h, &d.bin_flip, bflip));
    EC_TRY(dev_upload(h, &d.deg, deg));
    EC_TRY(dev_upload(h, &d.in_src, in_src));
    EC_TRY(dev_upload(h, &d.thr_link_m, thr_link_m));
    EC_TRY(dev_upload(h, &d.slot_of, slot_of));
    EC_TRY(dev_upload(h, &d.rate, rate));
    EC_TRY(dev_upload(h, &d.adj, adj));
    EC_TRY(dev_upload(h, &d.efr, efr));
    EC_TRY(dev_upload(h, &d.tfr, tfr));
    EC_TRY(dev_upload(h, &d.cores, cores));
    d.adj_bits = nullptr;
    if (N <= 64) EC_TRY(dev_upload(h, &d.adj_bits, adj_bits));

    const size_t P = (size_t)d.P;
    EC_TRY(dev_alloc(h, &d.mt, 2 * P * MT_STRIDE));
    EC_TRY(dev_alloc(h, &d.rng_pos, 2 * P));
    EC_TRY(dev_alloc(h, &d.rng_ahead, 2 * P));
    EC_TRY(dev_alloc(h, &d.rng_cnt, 2 * P));
    EC_TRY(dev_alloc(h, &d.has_new, P));
    EC_TRY(dev_alloc(h, &d.newt, P, true));
    EC_TRY(dev_alloc(h, &d.buf_head, P));
    EC_TRY(dev_alloc(h, &d.buf_len, P));
    EC_TRY(dev_alloc(h, &d.buf, (size_t)d.capB * P));
    EC_TRY(dev_alloc(h, &d.wq_len, P));
    EC_TRY(dev_alloc(h, &d.wqm, (size_t)d.capW * P));
    EC_TRY(dev_alloc(h, &d.wqe, (size_t)d.capW * P));
    EC_TRY(dev_alloc(h, &d.ex_n, P));
    EC_TRY(dev_alloc(h, &d.exA, (size_t)d.k * P));
    EC_TRY(dev_alloc(h, &d.exB, (size_t)d.k * P));
    EC_TRY(dev_alloc(h, &d.exC, (size_t)d.k * P));
    EC_TRY(dev_alloc(h, &d.rq_head, (size_t)D * P));
    EC_TRY(dev_alloc(h, &d.rq_len, (size_t)D * P));
    EC_TRY(dev_alloc(h, &d.rq_arrive, (size_t)D * P));
    EC_TRY(dev_alloc(h, &d.rq_hstart, (size_t)D * P));
    EC_TRY(dev_alloc(h, &d.rq, (size_t)D * d.capR * P));
    EC_TRY(dev_alloc(h, &d.n_off, P));
    EC_TRY(dev_alloc(h, &d.cur_step, (size_t)B));
    EC_TRY(dev_alloc(h, &d.tick, (size_t)B));
    EC_TRY(dev_alloc(h, &d.total_steps, (size_t)B));
    EC_TRY(dev_alloc(h, &d.hop_hist, (size_t)B * d.H));
    EC_TRY(dev_alloc(h, &d.status, 1, true));
    d.stamps = nullptr;

    // Launch geometry of k_step.  The step is bound by the dependent chain of one wave, not by bandwidth, so
    //  * a launch whose blocks are not all co-resident costs a second full round: every block must be resident;
    //  * fewer environments per main wave mean less divergence per wave;
    //  * the RNG producer lanes get a wave of their own (one per 64 lanes of the block) when that still fits.
    // Candidates in order of preference: producers yes/no is tried per (envs per wave, main waves) pair, envs
    // per wave ascending, main waves ascending (more of them share one producer wave and one LUT copy).
    h->step = pick_step<false>(d.k, D, &h->kmax, &h->dmax);
    h->step_multi = pick_step<true>(d.k, D, &h->kmax, &h->dmax);
    if (!h->step) return ec::fail(EC_EINVAL, "k=%d / max in-degree %d exceed the compiled limits (12 / 8)", d.k, D);
    const void* fn = reinterpret_cast<const void*>(h->step);
    hipDeviceProp_t prop;
    EC_HIP(hipGetDeviceProperties(&prop, h->device));
    struct Geo {
        int epw, mw, pw;  // envs per main wave (0: one env over mw waves), main waves, producer waves
        int epb() const { return epw == 0 ? 1 : epw * mw; }
        int threads() const { return 64 * (mw + pw); }
    };
    // rounds: how many times over the launch's blocks exceed what the device holds at once (0: does not fit a block)
    auto geo_fits = [&](const Geo& g, size_t* lds_out, bool* resident, int64_t* rounds) -> int {
        const size_t lds = step_lds_bytes(g.epb() * N, h->kmax, h->dmax, d.capW, d.acc_words, N);
        *lds_out = lds;
        *resident = false;
        *rounds = 0;
        if (lds > 160 * 1024 || g.threads() > 256) return EC_OK;
        EC_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int per_cu = 0;
        EC_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, g.threads(), lds));
        const int64_t blocks = (B + g.epb() - 1) / g.epb(), fit = (int64_t)per_cu * prop.multiProcessorCount;
        if (fit < 1) return EC_OK;
        *rounds = (blocks + fit - 1) / fit;
        *resident = *rounds == 1;
        return EC_OK;
    };
    const bool want_producers = !(getenv("MARL_AMD_EC_PRODUCERS") && atoi(getenv("MARL_AMD_EC_PRODUCERS")) == 0);
    Geo geo{0, 0, 0};
    bool found = false;
    if (N > 64) {
        const int mw = (N + 63) / 64;
        if (mw > 4) return ec::fail(EC_EINVAL, "n_nodes=%d exceeds 256 lanes per env", N);
        geo = Geo{0, mw, (want_producers && 2 * mw <= 4) ? mw : 0};
        found = true;
    } else if (getenv("MARL_AMD_EC_EPW") && getenv("MARL_AMD_EC_MW")) {
        geo.epw = atoi(getenv("MARL_AMD_EC_EPW"));
        geo.mw = atoi(getenv("MARL_AMD_EC_MW"));
        if (geo.epw < 1 || geo.epw * N > 64 || geo.mw < 1 || geo.mw > 4)
            return ec::fail(EC_EINVAL, "envs per wave %d x main waves %d unsupported (n_nodes=%d)", geo.epw, geo.mw, N);
        geo.pw = want_producers ? (geo.epw * geo.mw * N + 63) / 64 : 0;
        if (geo.threads() > 256) return ec::fail(EC_EINVAL, "%d threads per block exceed 256", geo.threads());
        found = true;
    } else {
        // too large for one round: the candidate with the fewest rounds, first in the same order of preference
        // (8192 envs x 10 nodes: 2 rounds of {4 envs per wave + producer} take 102 us, 3 rounds of 6 envs per wave
        // 112 us, the densest packing without producers 155 us)
        const int epw_max = 64 / N;
        Geo best{epw_max, 1, 0};
        int64_t best_rounds = 0;
        // First: a block per CU at most, with the fewest envs per wave that allows it.  Blocks that share a CU slow each
        // other down by more than the extra lanes of a fuller wave cost (512 envs x 10 nodes: 256 blocks of 2 envs
        // 12.5 us per step, 512 blocks of 1 env 13.4; 1024 envs: 256 blocks of 4 envs = 1024 blocks of 1 env).
        for (int epw = 1; want_producers && epw <= epw_max && !found; epw++) {
            Geo g{epw, 1, (epw * N + 63) / 64};
            if ((B + g.epb() - 1) / g.epb() > prop.multiProcessorCount) continue;
            size_t lds;
            bool resident;
            int64_t rounds;
            EC_TRY(geo_fits(g, &lds, &resident, &rounds));
            if (resident) {
                geo = g;
                found = true;
            }
        }
        for (int pass = want_producers ? 0 : 1; pass < 2 && !found; pass++) {  // pass 0: with producer waves
            for (int epw = 1; epw <= epw_max && !found; epw++) {
                for (int mw = 1; mw <= 4 && !found; mw++) {
                    Geo g{epw, mw, pass == 0 ? (epw * mw * N + 63) / 64 : 0};
                    size_t lds;
                    bool resident;
                    int64_t rounds;
                    EC_TRY(geo_fits(g, &lds, &resident, &rounds));
                    if (resident) {
                        geo = g;
                        found = true;
                    } else if (rounds > 0 && (best_rounds == 0 || rounds < best_rounds)) {
                        best = g;
                        best_rounds = rounds;
                    }
                }
            }
        }
        if (!found) geo = best;
    }
    const int epb = geo.epb();
    d.epb = epb;
    d.epw = geo.epw;
    d.nmain = 64 * geo.mw;
    // the instantiations with the block's lane count known at compile time, where they exist (same registers and LDS
    // as the generic ones the geometry search above queried)
    h->step = pick_step<false>(d.k, D, &h->kmax, &h->dmax, epb * N);
    h->step_multi = pick_step<true>(d.k, D, &h->kmax, &h->dmax, epb * N);
    fn = reinterpret_cast<const void*>(h->step);
    h->threads = geo.threads();
    h->threads_obs = ((epb * N + 63) / 64) * 64;
    h->envs_per_block = epb;
    h->blocks = (B + epb - 1) / epb;
    h->lds_obs = (size_t)epb * N * (3 * 4 + 9 * 4);
    h->lds_step = step_lds_bytes(epb * N, h->kmax, h->dmax, d.capW, d.acc_words, N);
    h->lds_step_links = step_lds_bytes(epb * N, h->kmax, h->dmax, d.capW, d.acc_words, N, N);
    if (h->lds_step > 160 * 1024) return ec::fail(EC_EINVAL, "LDS need %zu exceeds 160 KiB", h->lds_step);
    EC_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_step));
    EC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(h->step_multi), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)h->lds_step));
    {   // the looping kernel is correct at any residency (blocks are independent), but a second round doubles its time
        int per_cu = 0;
        const void* fn_multi = reinterpret_cast<const void*>(h->step_multi);
        EC_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn_multi, h->threads, h->lds_step));
        h->multi_resident = h->blocks <= (int64_t)per_cu * prop.multiProcessorCount;
    }
#ifdef EC_STAMPS
    EC_TRY(dev_alloc(h, &d.stamps, (size_t)h->blocks * 32, true));  // [0, 16 blocks): phases; then the EC_SUB sums
#endif

    EC_TRY(dev_alloc(h, &h->d_dev, 1));
    h->rollout_mlp = pick_rollout_mlp(d.k, D, d.A, false, epb * N);
    h->rollout_mlp_expert = pick_rollout_mlp(d.k, D, d.A, true, epb * N);
    h->d_staged = nullptr;
    if (h->rollout_mlp) {
        // + layer-2/3 biases + one action word per main lane (the producer waves' share of the Actor rows)
        // (+ another word per main lane: the queue lengths the in-kernel expert reads)
        h->lds_rollout_mlp = ((h->lds_step_links + 15) & ~(size_t)15) + (polinl::H + 64) * sizeof(float) + 2 * (size_t)d.nmain * sizeof(int);
        if (h->lds_rollout_mlp > 160 * 1024) return ec::fail(EC_EINVAL, "LDS need %zu exceeds 160 KiB", h->lds_rollout_mlp);
        EC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(h->rollout_mlp), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)h->lds_rollout_mlp));
        if (h->rollout_mlp_expert)
            EC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(h->rollout_mlp_expert),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_rollout_mlp));
        EC_TRY(dev_alloc(h, &h->d_staged, (size_t)polinl::staged_floats(N, 2)));
        if (getenv("MARL_AMD_EC_DEBUG")) {
            int per_cu = 0;
            EC_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(h->rollout_mlp),
                                                                h->threads, h->lds_rollout_mlp));
            fprintf(stderr, "rollout_mlp: %d blocks/CU, %lld blocks, %d CUs\n", per_cu, (long long)h->blocks,
                    prop.multiProcessorCount);
        }
    }
    EC_HIP(hipMemcpy(h->d_dev, &d, sizeof(EcDev), hipMemcpyHostToDevice));

    const u32 ts = (u32)(cfg->task_seed + env_id_base), fs = (u32)(cfg->failure_seed + env_id_base);
    const int64_t gb = (d.P + 255) / 256;
    hipLaunchKernelGGL(k_seed, dim3((unsigned)gb), dim3(256), 0, 0, d, ts, fs);
    hipLaunchKernelGGL(k_initial_generate, dim3((unsigned)gb), dim3(256), 0, 0, d);
    hipLaunchKernelGGL(k_prime_rng, dim3((unsigned)gb), dim3(256), 0, 0, d);
    EC_HIP(hipGetLastError());
    EC_HIP(hipDeviceSynchronize());
    return EC_OK;
}

}  // namespace

extern "C" {

int ec_create(const ec_config* cfg, const ec_topology* topo, int64_t n_envs, int64_t env_id_base, int device,
              ec_handle* out) {
    if (!cfg || !topo || !out) return ec::fail(EC_EINVAL, "null argument");
    if (cfg->n_nodes < 2 || cfg->n_nodes > 256) return ec::fail(EC_EINVAL, "n_nodes %d out of range [2,256]", cfg->n_nodes);
    if (n_envs < 1) return ec::fail(EC_EINVAL, "n_envs must be positive");
    if (cfg->k < 1 || cfg->k > 12) return ec::fail(EC_EINVAL, "k %d out of range [1,12]", cfg->k);
    if (cfg->mini_slots < 1 || cfg->mini_time_slot <= 0) return ec::fail(EC_EINVAL, "bad mini slot configuration");
    const int64_t top = n_envs + env_id_base + cfg->n_nodes;
    // numpy: RandomState(seed) needs 0 <= seed < 2**32
    if (cfg->task_seed < 0 || cfg->failure_seed < 0 || env_id_base < 0 || cfg->task_seed + top > 0xFFFFFFFFll ||
        cfg->failure_seed + top > 0xFFFFFFFFll)
        return ec::fail(EC_EINVAL, "seed out of RandomState range");
    ec_env* h = new ec_env();
    memset(&h->d, 0, sizeof(h->d));
    h->device = device;
    h->state_bytes = 0;
    DeviceGuard guard(device);  // the caller's current device is restored on return
    if (guard.err != hipSuccess) {
        delete h;
        return ec::fail(EC_EHIP, "cannot switch to device %d: %s", device, hipGetErrorString(guard.err));
    }
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
    DeviceGuard guard(h->device);
    (void)hipDeviceSynchronize();
    free_all(h);
    delete h;
    return EC_OK;
}

int ec_reset(ec_handle h, void* stream) {
    if (!h) return ec::fail(EC_EINVAL, "null handle");
    EC_ON_DEVICE(h);
    const int64_t gb = (h->d.P + 255) / 256;
    hipLaunchKernelGGL(k_reset, dim3((unsigned)gb), dim3(256), 0, (hipStream_t)stream, h->d);
    EC_HIP(hipGetLastError());
    return EC_OK;
}

int ec_observe(ec_handle h, float* obs, int32_t* avail, void* stream) {
    if (!h) return ec::fail(EC_EINVAL, "null handle");
    if (!obs && !avail) return EC_OK;
    EC_ON_DEVICE(h);
    hipLaunchKernelGGL(k_observe, dim3((unsigned)h->blocks), dim3(h->threads_obs), h->lds_obs, (hipStream_t)stream, h->d,
                       obs, avail);
    EC_HIP(hipGetLastError());
    return EC_OK;
}

static int step_impl(ec_handle h, const int64_t* actions, float* reward, uint8_t* terminated, int32_t* counters,
                     float* completion_time, int32_t* hop_hist, float* next_obs, int32_t* next_avail, StepPolicy pol,
                     void* stream) {
    if (!h || !actions) return ec::fail(EC_EINVAL, "null handle or actions");
    EC_ON_DEVICE(h);
    const bool multi = pol.n_steps > 1;
    hipLaunchKernelGGL(multi ? h->step_multi : h->step, dim3((unsigned)h->blocks), dim3(h->threads),
                       h->lds_step, (hipStream_t)stream, h->d, h->d_dev, actions, reward,
                       terminated, counters, completion_time, next_obs, next_avail, pol);
    if (hop_hist) {
        const int64_t n = h->d.B * h->d.H;
        hipLaunchKernelGGL(k_copy_i32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           h->d.hop_hist, hop_hist, n);
    }
    EC_HIP(hipGetLastError());
    return EC_OK;
}

int ec_step(ec_handle h, const int64_t* actions, float* reward, uint8_t* terminated, int32_t* counters,
            float* completion_time, int32_t* hop_hist, float* next_obs, int32_t* next_avail, void* stream) {
    return step_impl(h, actions, reward, terminated, counters, completion_time, hop_hist, next_obs, next_avail,
                     StepPolicy{0, 0, nullptr, 1, 0}, stream);
}

int ec_step_random_policy(ec_handle h, const int64_t* actions, float* reward, uint8_t* terminated, int32_t* counters,
                          float* completion_time, int32_t* hop_hist, float* next_obs, int32_t* next_avail,
                          uint64_t seed, uint64_t counter, int64_t* next_actions, void* stream) {
    if (!next_actions) return ec::fail(EC_EINVAL, "ec_step_random_policy: null next_actions");
    if (!next_avail) return ec::fail(EC_EINVAL, "ec_step_random_policy needs the fused observation (next_avail)");
    return step_impl(h, actions, reward, terminated, counters, completion_time, hop_hist, next_obs, next_avail,
                     StepPolicy{seed, counter, next_actions, 1, 0}, stream);
}

int ec_rollout_random_policy(ec_handle h, int32_t n_steps, int32_t keep_history, const int64_t* actions, float* reward,
                             uint8_t* terminated, int32_t* counters, float* completion_time, int32_t* hop_hist,
                             float* next_obs, int32_t* next_avail, uint64_t seed, uint64_t counter,
                             int64_t* next_actions, void* stream) {
    if (n_steps < 1) return ec::fail(EC_EINVAL, "ec_rollout_random_policy: n_steps = %d", n_steps);
    if (!next_actions) return ec::fail(EC_EINVAL, "ec_rollout_random_policy: null next_actions");
    if (!next_avail) return ec::fail(EC_EINVAL, "ec_rollout_random_policy needs the fused observation (next_avail)");
    if (keep_history && next_actions == actions)
        return ec::fail(EC_EINVAL, "ec_rollout_random_policy: with keep_history the drawn actions need their own [n_steps,B,N] array");
    return step_impl(h, actions, reward, terminated, counters, completion_time, hop_hist, next_obs, next_avail,
                     StepPolicy{seed, counter, next_actions, n_steps, keep_history ? 1 : 0}, stream);
}

static int rollout_mlp_impl(const char* what, ec_handle h, int32_t n_steps, const float* w1, const float* b1, const float* w2,
                            const float* b2, const float* w3, const float* b3, const float* uniforms, const float* obs0,
                            const int32_t* avail0, int64_t* actions, float* reward, uint8_t* terminated, int32_t* counters,
                            float* completion_time, int32_t* hop_hist, float* next_obs, int32_t* next_avail,
                            const ExpertRollout* er, void* stream);

int ec_rollout_mlp_policy(ec_handle h, int32_t n_steps, const float* w1, const float* b1, const float* w2, const float* b2,
                          const float* w3, const float* b3, const float* uniforms, const float* obs0,
                          const int32_t* avail0, int64_t* actions, float* reward, uint8_t* terminated, int32_t* counters,
                          float* completion_time, int32_t* hop_hist, float* next_obs, int32_t* next_avail, void* stream) {
    return rollout_mlp_impl("ec_rollout_mlp_policy", h, n_steps, w1, b1, w2, b2, w3, b3, uniforms, obs0, avail0, actions, reward,
                            terminated, counters, completion_time, hop_hist, next_obs, next_avail, nullptr, stream);
}

int ec_rollout_mlp_expert_policy(ec_handle h, int32_t n_steps, const float* w1, const float* b1, const float* w2,
                                 const float* b2, const float* w3, const float* b3, const float* uniforms, const float* obs0,
                                 const int32_t* avail0, const float* coin, const float* ratios, int64_t* actions,
                                 int64_t* expert_actions, float* reward, uint8_t* terminated, int32_t* counters,
                                 float* completion_time, int32_t* hop_hist, float* next_obs, int32_t* next_avail,
                                 void* stream) {
    if (!coin || !ratios || !expert_actions)
        return ec::fail(EC_EINVAL, "ec_rollout_mlp_expert_policy: coin, ratios and expert_actions are required");
    const ExpertRollout er{coin, ratios, expert_actions};
    return rollout_mlp_impl("ec_rollout_mlp_expert_policy", h, n_steps, w1, b1, w2, b2, w3, b3, uniforms, obs0, avail0, actions,
                            reward, terminated, counters, completion_time, hop_hist, next_obs, next_avail, &er, stream);
}

static int rollout_mlp_impl(const char* what, ec_handle h, int32_t n_steps, const float* w1, const float* b1, const float* w2,
                            const float* b2, const float* w3, const float* b3, const float* uniforms, const float* obs0,
                            const int32_t* avail0, int64_t* actions, float* reward, uint8_t* terminated, int32_t* counters,
                            float* completion_time, int32_t* hop_hist, float* next_obs, int32_t* next_avail,
                            const ExpertRollout* er, void* stream) {
    if (!h) return ec::fail(EC_EINVAL, "%s: null handle", what);
    if (n_steps < 1) return ec::fail(EC_EINVAL, "%s: n_steps = %d", what, n_steps);
    if (!obs0 || !avail0 || !actions || !next_obs || !next_avail)
        return ec::fail(EC_EINVAL, "%s: obs0, avail0, actions, next_obs and next_avail are required", what);
    const rollout_mlp_fn fn = er ? h->rollout_mlp_expert : h->rollout_mlp;
    if (!h->rollout_mlp || !fn)
        return ec::fail(EC_EINVAL, "%s is compiled for k <= 4 and in-degree <= 8 with n_actions <= 32 or in-degree <= 4 with "
                        "n_actions <= 64 (have k=%d, D=%d, A=%d)", what, h->d.k, h->d.D, h->d.A);
    EC_ON_DEVICE(h);
    const int rc = ec::stage_policy(w1, b1, w2, b2, w3, b3, h->d.N, h->d.A, h->d_staged, (hipStream_t)stream);
    if (rc != EC_OK) return rc;
    hipLaunchKernelGGL(fn, dim3((unsigned)h->blocks), dim3(h->threads), h->lds_rollout_mlp, (hipStream_t)stream, h->d_dev,
                       MlpRollout{h->d_staged, uniforms, obs0, avail0, n_steps, (unsigned)((h->lds_step_links + 15) & ~(size_t)15)},
                       actions, reward, terminated, counters, completion_time, next_obs, next_avail,
                       er ? *er : ExpertRollout{nullptr, nullptr, nullptr});
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

int ec_expert_policy(ec_handle h, const int32_t* avail, int64_t* actions, void* stream) {
    if (!h || !avail || !actions) return ec::fail(EC_EINVAL, "ec_expert_policy: null argument");
    EC_ON_DEVICE(h);
    hipLaunchKernelGGL(k_expert, dim3((unsigned)((h->d.P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, h->d, avail,
                       actions);
    EC_HIP(hipGetLastError());
    return EC_OK;
}

int ec_status(ec_handle h, uint32_t* status_host, int clear, void* stream) {
    if (!h || !status_host) return ec::fail(EC_EINVAL, "null argument");
    EC_ON_DEVICE(h);
    EC_HIP(hipMemcpyAsync(status_host, h->d.status, sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    if (clear) EC_HIP(hipMemsetAsync(h->d.status, 0, sizeof(uint32_t), (hipStream_t)stream));
    EC_HIP(hipStreamSynchronize((hipStream_t)stream));
    return EC_OK;
}

int ec_debug_dump(ec_handle h, int64_t env, double* rows_host, int64_t max_rows, int64_t* n_rows_host) {
    if (!h || !rows_host || !n_rows_host || env < 0 || env >= h->d.B) return ec::fail(EC_EINVAL, "bad argument");
    EC_ON_DEVICE(h);
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
    EC_ON_DEVICE(h);
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
    EC_ON_DEVICE(h);
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
    const int64_t v[19] = {h->d.B, h->d.N, h->d.k, h->d.D, h->d.capB, h->d.capR, h->d.capW, h->d.maxTicks,
                           h->threads, h->envs_per_block, (int64_t)(h->state_bytes & 0xffffffffull),
                           (int64_t)(h->state_bytes >> 32), (int64_t)h->lds_step,
                           h->d.epw, h->d.nmain / 64, (h->threads - h->d.nmain) / 64, h->multi_resident ? 1 : 0,
                           h->rollout_mlp ? 1 : 0, (h->rollout_mlp && h->rollout_mlp_expert) ? 1 : 0};
    for (int i = 0; i < n && i < 19; i++) info_host[i] = v[i];
    return EC_OK;
}

#ifdef EC_STAMPS
/* diagnostic build only: copy the per-block phase stamps of the last ec_step to the host */
int ec_debug_stamps(ec_handle h, uint64_t* out_host, int64_t* n_blocks) {
    EC_HIP(hipDeviceSynchronize());
    EC_HIP(hipMemcpy(out_host, h->d.stamps, sizeof(u64) * 32 * (size_t)h->blocks, hipMemcpyDeviceToHost));
    *n_blocks = h->blocks;
    return EC_OK;
}
#endif

const char* ec_last_error(void) { return ec::err_buf(); }
const char* ec_version(void) { return "marl_amd 0.4 (gfx950)"; }

}  // extern "C"
