// ec_mfma_grad.h — device helpers shared by the fused forward + backward kernels (ec_ppo.hip, ec_diff.hip): the
// register-chained 32x32x2 layer on one group of 32 rows, the per-wave [index][row] exchange tiles, the
// weight-gradient accumulation with the row index as the MFMA's K dimension, and the atomic flush.
#pragma once
#include "ec_mfma.h"

namespace {

// Exchange tiles between the accumulator layout (a lane holds 16 outputs of ONE row) and the weight-gradient MFMA's
// operands (a lane holds ONE output of 16 rows): [index][row], row stride TSTR = 33 floats (conflict-free both ways).
// (A [row][index] layout with one 16-byte store per four registers was measured too: same time, and the 4-register
// alignment it asks for costs the wide instantiations (N = 40) 1.5 KB of spills per lane.)
constexpr int TSTR = 33;
constexpr int TILE = 64 * TSTR;          // floats per tile

// out[to] += frag * f(in): TI input tiles of 16 accumulator registers (k order kperm), one group of 32 rows.
// KQ: live 4-register groups of the LAST input tile (the k-steps of the others multiply zeros and are skipped).
// The fragment word of a group of four MFMAs is requested one group AHEAD: a wave alone on its SIMD issues in order, so
// a ds_read placed between two groups is waited for with the matrix pipe draining (≈100 cycles per 256 of MFMA); the
// sched_group_barriers keep the compiler from sinking the read back to its use.
template <int TO, int TI, int KQ = 4, typename F>
__device__ __forceinline__ void chain(f32x16 (&out)[TO], const float* __restrict__ frag, const f32x16 (&in)[TI], int lane, F f) {
    constexpr int NG = (4 * (TI - 1) + KQ) * TO;  // groups of four MFMAs = fragment words
    float4 nxt = *reinterpret_cast<const float4*>(frag + lane * 4);
#pragma unroll
    for (int g = 0; g < NG; g++) {
        const int s4 = g / TO, to = g - s4 * TO;
        const float4 a = nxt;
        if (g + 1 < NG) nxt = *reinterpret_cast<const float4*>(frag + ((g + 1) * 64 + lane) * 4);
        const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int T = s4 >> 2, r = (s4 & 3) * 4 + c;
            out[to] = MFMA(av[c], f(T, r, in[T][r]), out[to]);
        }
    }
#pragma unroll
    for (int g = 0; g < NG; g++) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read (the next group's word)
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);  // 4 MFMA
    }
}

// first layer over the 9 observation inputs of the wave's 32 rows (s_obs [32][9])
__device__ __forceinline__ void chain_obs(f32x16 (&out)[2], const float* __restrict__ frag, const float* __restrict__ s_obs, int lane) {
    const int jrow = lane & 31, half = lane >> 5;
#pragma unroll
    for (int s4 = 0; s4 < 2; s4++) {
#pragma unroll
        for (int to = 0; to < 2; to++) {
            const float4 a = *reinterpret_cast<const float4*>(frag + ((s4 * 2 + to) * 64 + lane) * 4);
            const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int s = 4 * s4 + c;
                if (s < 5) {
                    const int k = 2 * s + half;
                    out[to] = MFMA(av[c], k < OBS ? s_obs[jrow * OBS + k] : 0.0f, out[to]);
                }
            }
        }
    }
}

// accumulator-layout registers -> exchange tile [index][row]
template <int T, typename F>
__device__ __forceinline__ void put_tile(float* __restrict__ tile, const f32x16 (&v)[T], int lane, F f) {
#ifdef ABL_NO_PUT  // diagnostic builds (make abl_ppo): what the transposition stores / the weight-gradient MFMAs cost
    return;
#endif
    const int jrow = lane & 31, half = lane >> 5;
#pragma unroll
    for (int t = 0; t < T; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) tile[(32 * t + 8 * (r >> 2) + 4 * half + (r & 3)) * TSTR + jrow] = f(t, r, v[t][r]);
}

// acc[to][ti] += Td[out][row] (x) Ta[in][row] over the tile's 32 rows; dbias[to] += sum_rows Td[out][row] (lane's parity)
// UNR = 16: straight-line code, for a call site whose surroundings are to be scheduled between the MFMAs
// AHEAD: request the operands of step s + 1 before the MFMAs of step s (for instantiations with registers to spare)
template <int TO, int TI, int UNR = 4, bool AHEAD = true>
__device__ __forceinline__ void wgrad(f32x16 (&acc)[TO][TI], float (&dbias)[TO], const float* __restrict__ Td,
                                      const float* __restrict__ Ta, int lane) {
#ifdef ABL_NO_WGRAD
    return;
#endif
    const int i = lane & 31, half = lane >> 5;
    if constexpr (!AHEAD) {
#pragma unroll UNR
        for (int s = 0; s < 16; s++) {
            const int row = 2 * s + half;
            float a[TO], b[TI];
#pragma unroll
            for (int to = 0; to < TO; to++) {
                a[to] = Td[(32 * to + i) * TSTR + row];
                dbias[to] += a[to];
            }
#pragma unroll
            for (int ti = 0; ti < TI; ti++) b[ti] = Ta[(32 * ti + i) * TSTR + row];
#pragma unroll
            for (int to = 0; to < TO; to++)
#pragma unroll
                for (int ti = 0; ti < TI; ti++) acc[to][ti] = MFMA(a[to], b[ti], acc[to][ti]);
        }
        return;
    }
    // operands of step s + 1 are requested before the MFMAs of step s (see chain(): an LDS round trip between two
    // steps is otherwise waited for with the matrix pipe idle)
    float an[TO], bn[TI];
#pragma unroll
    for (int to = 0; to < TO; to++) an[to] = Td[(32 * to + i) * TSTR + half];
#pragma unroll
    for (int ti = 0; ti < TI; ti++) bn[ti] = Ta[(32 * ti + i) * TSTR + half];
#pragma unroll UNR
    for (int s = 0; s < 16; s++) {
        float a[TO], b[TI];
#pragma unroll
        for (int to = 0; to < TO; to++) a[to] = an[to];
#pragma unroll
        for (int ti = 0; ti < TI; ti++) b[ti] = bn[ti];
        const int row = 2 * (s < 15 ? s + 1 : 15) + half;  // (the last step re-reads its own row: no branch)
#pragma unroll
        for (int to = 0; to < TO; to++) an[to] = Td[(32 * to + i) * TSTR + row];
#pragma unroll
        for (int ti = 0; ti < TI; ti++) bn[ti] = Ta[(32 * ti + i) * TSTR + row];
#pragma unroll
        for (int to = 0; to < TO; to++) dbias[to] += a[to];
#pragma unroll
        for (int to = 0; to < TO; to++)
#pragma unroll
            for (int ti = 0; ti < TI; ti++) acc[to][ti] = MFMA(a[to], b[ti], acc[to][ti]);
        __builtin_amdgcn_sched_group_barrier(0x100, TO + TI, 0);   // the next step's DS reads
        __builtin_amdgcn_sched_group_barrier(0x008, TO * TI, 0);   // this step's MFMAs
    }
}

// Sum the block's waves' accumulators into wave 0 through LDS (`buf`: at least (nw - 1) * TA * TB * 1024 floats that no
// wave uses any more).  Every wave of a pass ends with the same ~6 k gradient words to add to: one wave per block doing
// the atomics instead of four is a quarter of the memory-side atomic traffic.
template <int TA, int TB>
__device__ __forceinline__ void block_sum(f32x16 (&acc)[TA][TB], float* __restrict__ buf, int wave, int nw, int lane) {
    __syncthreads();
    if (wave > 0) {
#pragma unroll
        for (int a = 0; a < TA; a++)
#pragma unroll
            for (int b = 0; b < TB; b++)
#pragma unroll
                for (int r = 0; r < 16; r++) buf[((((wave - 1) * TA + a) * TB + b) * 16 + r) * 64 + lane] = acc[a][b][r];
    }
    __syncthreads();
    if (wave == 0) {
        for (int w = 1; w < nw; w++) {
#pragma unroll
            for (int a = 0; a < TA; a++)
#pragma unroll
                for (int b = 0; b < TB; b++)
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[a][b][r] += buf[((((w - 1) * TA + a) * TB + b) * 16 + r) * 64 + lane];
        }
    }
}
template <int NS>
__device__ __forceinline__ void block_sum_scalars(float (&v)[NS], float* __restrict__ buf, int wave, int nw, int lane) {
    __syncthreads();
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < NS; i++) buf[((wave - 1) * NS + i) * 64 + lane] = v[i];
    }
    __syncthreads();
    if (wave == 0) {
        for (int w = 1; w < nw; w++) {
#pragma unroll
            for (int i = 0; i < NS; i++) v[i] += buf[((w - 1) * NS + i) * 64 + lane];
        }
    }
}

// flush an accumulator tile pair into a row-major [n_out][ld] gradient with atomics
// (atomic = false: plain stores into a block-private copy of the gradient, summed by a second launch)
template <int TO, int TI>
__device__ __forceinline__ void flush(float* __restrict__ g, int ld, int n_out, int n_in, const f32x16 (&acc)[TO][TI], int lane,
                                      bool atomic = true) {
    const int j = lane & 31, half = lane >> 5;
#pragma unroll
    for (int to = 0; to < TO; to++)
#pragma unroll
        for (int ti = 0; ti < TI; ti++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int out = 32 * to + 8 * (r >> 2) + 4 * half + (r & 3), in = 32 * ti + j;
                if (out < n_out && in < n_in) {
                    if (atomic) atomicAdd(g + (size_t)out * ld + in, acc[to][ti][r]);
                    else g[(size_t)out * ld + in] = acc[to][ti][r];
                }
            }
}

}  // namespace
