// ec_linear.hip — y = x W^T + b for a tall batch: the forward of the GAIL discriminator's Linear layers
// (rl/policy_gradient_rl/gail/gail_network.py:97-125: 30 / 50 -> 128 -> 64 -> 32 on 4.1 M .. 65.5 M rows).
//
// At these shapes a layer is HBM-bound (read [rows, K], write [rows, C]); hipBLASLt runs it at ~3 TB/s.  Here a wave owns
// 32 rows at a time and the ROWS are the MFMA's M dimension: D[row][out] += A[row][k] B[k][out].  A lane fetches its row's
// inputs 16 bytes at a time in the order the MFMA walks them (kperm); the weights sit in LDS in the fragment order of
// ec_mfma.h (the lane index is the output, as there); and the accumulator leaves lane (out j, half h) holding 16 rows of
// output j -- so every store instruction writes 32 consecutive floats of two rows: whole 128-byte lines.  (The same
// kernel with outputs as M, as the fused learner kernels have it, stores 16-byte pieces and runs at hipBLASLt's rate.)
//
// Also here (round 3): k_linear2_fwd_bn, the discriminator's first two layers in one kernel for the reward pass (the
// 128-wide tensor between them stays in registers); k_moments, the moments of the input rows from which the first
// BatchNorm's statistics follow without computing the layer; k_disc_rows, the (obs | id | action) input rows.
#include "ec_mfma.h"

namespace {

struct LinArgs {
    const float* x;  // [rows, K]
    const float* w;  // [C, K]
    const float* b;  // [C] or NULL
    int64_t rows;
    int K, C;
    float* y;        // [rows, C]
    // BN: the input is the previous layer's pre-BatchNorm output; a = relu(x * scale + shift) * keep / (1 - p) is formed
    // in registers on the way to the MFMA (csrc/ec_bn.hip's forward, same arithmetic and the same keep_draw)
    const float* scale;  // [K]
    const float* shift;  // [K]
    uint64_t seed;
    uint32_t thr24;
    float inv_keep;
    // optional: the batch statistics of y for the NEXT BatchNorm, as ec_bn_col_stats leaves them for ec_bn_finalize:
    // partials [gridDim.x][2][C] <- this block's sums over its rows of (y - pivot) and (y - pivot)^2; pivot [C], or NULL:
    // the bias row
    float* partials;
    int n_partials;
    const float* pivot;
};

// (the keep flags are ec_bn.hip's keep_draw(seed, row * K + col), restated in the kernel with the per-row work hoisted: the
// two must stay the same function -- tests compare the fused layer with the two-step one bit for bit)
// STG (K = 32 TI exactly, rows 16-byte aligned): the wave's 32 x K tile comes in as whole 1 KB runs -- lane l fetches the
// tile's 16-byte pieces l, l + 64, .. -- and is handed to the row-per-lane order through a padded LDS tile; the next tile's
// pieces are requested before this tile's MFMAs.  (Row-per-lane reads fetch 16 bytes of each of 32 rows per instruction:
// 32 lines touched for 1 KB, and the 64 -> 32 and 32 -> 1 layers ran at 3.0 / 2.1 TB/s.)  Measured at 4.1 M rows: 64 -> 32
// 0.37 -> 0.30 ms, 128 -> 64 1.08 -> 0.87 ms, 128 -> 30 0.67 -> 0.58 ms, 96 -> 128 1.33 -> 1.15 ms; with the BN prologue
// 64 -> 32 0.52 -> 0.39 ms but 128 -> 64 1.15 -> 1.30 ms (one block of 100 KB per CU against three rolled waves per SIMD),
// so BN instantiations with more than 64 inputs keep the rolled row-per-lane loads.
template <int TO, int TI, bool BN, bool ST, bool STG = false>   // ST: the plain layer with the statistics epilogue (BN takes it at run time)
__global__ void __launch_bounds__(256) k_linear_fwd(LinArgs p) {
    extern __shared__ __align__(16) float lds[];
    float* s_f = lds;                      // fragments: 16 TI steps x TO tiles x 64 lanes
    float* s_b = s_f + 16 * TI * TO * 64;  // [32 TO] bias
    float* s_sc = s_b + 32 * TO;           // BN: [32 TI] scale, [32 TI] shift
    float* s_sh = s_sc + 32 * TI;
    float* s_pv = s_sh + 32 * TI;          // [32 TO] pivot of the statistics
    constexpr int KS = 32 * TI + 4;        // STG: row stride of a wave's staged tile (conflict-free 16-byte reads)
    float* s_x = s_pv + 32 * TO + (threadIdx.x >> 6) * 32 * KS;
    const int K = p.K, C = p.C;
    stage_frag(s_f, p.w, K, 1, C, K, 16 * TI, TO, true);
    for (int i = threadIdx.x; i < 32 * TO; i += blockDim.x) {
        s_b[i] = (p.b && i < C) ? p.b[i] : 0.0f;
        s_pv[i] = i < C ? (p.pivot ? p.pivot[i] : s_b[i]) : 0.0f;
    }
    if (BN)
        for (int i = threadIdx.x; i < 32 * TI; i += blockDim.x) {
            s_sc[i] = i < K ? p.scale[i] : 0.0f;
            s_sh[i] = i < K ? p.shift[i] : 0.0f;
        }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int j = lane & 31, half = lane >> 5;
    const bool vec = (K & 3) == 0;  // rows are 16-byte aligned
    const int64_t n_tiles = (p.rows + 31) / 32;
    const bool stats = (BN || ST) && p.partials != nullptr;   // (in the plain kernels the accumulators cost up to a third)
    float st1[TO], st2[TO];   // lane (out j, half): sums over the rows it held of output 32 to + j
#pragma unroll
    for (int to = 0; to < TO; to++) st1[to] = st2[to] = 0.0f;
    constexpr int NV = STG ? 4 * TI : 1;   // 16-byte pieces of a 32 x 32 TI tile per lane
    float4 g[NV];
    const int64_t total = p.rows * (int64_t)K;
    const auto stage_load = [&](int64_t tile_) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const int64_t e = tile_ * 32 * K + 4 * (i * 64 + lane);
            const bool in = e < total;   // (K a multiple of 4: a piece lies wholly inside or outside)
            const float4 v = *reinterpret_cast<const float4*>(p.x + (in ? e : 0));
            g[i] = make_float4(in ? v.x : 0.0f, in ? v.y : 0.0f, in ? v.z : 0.0f, in ? v.w : 0.0f);
        }
    };
    if (STG) stage_load((int64_t)blockIdx.x * nw + wave);
    for (int64_t tile = (int64_t)blockIdx.x * nw + wave; tile < n_tiles; tile += (int64_t)gridDim.x * nw) {
        if (STG) {
            wave_sync();   // (the previous tile's reads are done)
#pragma unroll
            for (int i = 0; i < NV; i++) {
                const int idx = i * 64 + lane;
                *reinterpret_cast<float4*>(s_x + (idx / (8 * TI)) * KS + 4 * (idx % (8 * TI))) = g[i];
            }
            wave_sync();
            stage_load(tile + (int64_t)gridDim.x * nw);   // (past the last tile: zeros)
        }
        const int64_t rowbase = tile * 32, row = rowbase + j;  // as the A operand, lane (j, half) is row j
        const bool ok = row < p.rows;
        const float* xr = p.x + (ok ? row : 0) * K;
        // BN: keep_draw(seed, row * K + k) with the per-row part of the hash taken out of the element loop.  The index's
        // low word times the first constant is linear in k (mod 2^32), and its high word takes one of two values per row
        // (k < 128 carries at most once): two multiplies per element are left, the finaliser's.
        uint32_t kd_lo = 0, kd_mul = 0, kd_hi0 = 0, kd_hi1 = 0;
        if (BN) {
            const uint64_t base = (uint64_t)(ok ? row : 0) * (uint64_t)K + (uint64_t)(4 * half);
            kd_lo = (uint32_t)base;
            kd_mul = kd_lo * 0x9e3779b1u;
            const uint32_t hi = (uint32_t)(base >> 32) + (uint32_t)(p.seed >> 32);
            kd_hi0 = hi * 0x85ebca77u;
            kd_hi1 = (hi + 1u) * 0x85ebca77u;
        }
        f32x16 acc[TO];
#pragma unroll
        for (int to = 0; to < TO; to++) {
            const float bias = s_b[32 * to + j];
#pragma unroll
            for (int r = 0; r < 16; r++) acc[to][r] = bias;
        }
        // ROLL: the T loop stays rolled, with the next step's row loads issued ahead of this step's MFMAs.  Fully unrolled,
        // a 128-input instantiation holds every load of the tile at once (288 .. 416 registers: one wave per SIMD); rolled
        // it fits three.  Measured at 4.1 M rows: 128 -> 64 with the BN prologue 1.54 -> 1.14 ms, 128 -> 30 0.94 -> 0.67 ms;
        // but 128 -> 64 without the prologue 1.09 -> 1.27 ms and 64 -> 128 0.96 -> 1.16 ms, so those stay unrolled.
        auto load_x = [&](int T, float4 (&xa)[4]) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int k0 = 32 * T + 8 * q + 4 * half;
                if (STG) {
                    xa[q] = *reinterpret_cast<const float4*>(s_x + j * KS + k0);
                } else if (vec) {
                    xa[q] = (ok && k0 < K) ? *reinterpret_cast<const float4*>(xr + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
                } else {
                    xa[q].x = (ok && k0 + 0 < K) ? xr[k0 + 0] : 0.0f;
                    xa[q].y = (ok && k0 + 1 < K) ? xr[k0 + 1] : 0.0f;
                    xa[q].z = (ok && k0 + 2 < K) ? xr[k0 + 2] : 0.0f;
                    xa[q].w = (ok && k0 + 3 < K) ? xr[k0 + 3] : 0.0f;
                }
            }
        };
        constexpr bool ROLL = !STG && TI >= 3 && (BN || TO == 1);   // (staged: the rows wait in LDS, nothing to keep in flight)
        constexpr int UNR_T = ROLL ? 1 : TI;
        float4 xa[4], xn[4];
        if (ROLL) load_x(0, xa);
#pragma unroll UNR_T
        for (int T = 0; T < TI; T++) {
            if (ROLL) {
                if (T + 1 < TI) load_x(T + 1, xn);
            } else {
                load_x(T, xa);
            }
            if (BN) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int k0 = 32 * T + 8 * q + 4 * half;
                    const float4 sc = *reinterpret_cast<const float4*>(s_sc + k0), sh = *reinterpret_cast<const float4*>(s_sh + k0);
                    const float xv[4] = {xa[q].x, xa[q].y, xa[q].z, xa[q].w}, scv[4] = {sc.x, sc.y, sc.z, sc.w},
                                shv[4] = {sh.x, sh.y, sh.z, sh.w};
                    float o[4];
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const float yv = xv[c] * scv[c] + shv[c];
                        const uint32_t off = (uint32_t)(32 * T + 8 * q + c);   // (compile-time; 4 * half is in the base)
                        uint32_t h = (kd_mul + off * 0x9e3779b1u) ^ (uint32_t)p.seed;
                        h ^= (kd_lo + off < kd_lo) ? kd_hi1 : kd_hi0;
                        h ^= h >> 16;
                        h *= 0x85ebca6bu;
                        h ^= h >> 13;
                        h *= 0xc2b2ae35u;
                        h ^= h >> 16;
                        const bool keep = ok && k0 + c < K && (h >> 8) >= p.thr24;
                        o[c] = (keep && yv > 0.0f) ? yv * p.inv_keep : 0.0f;
                    }
                    xa[q] = make_float4(o[0], o[1], o[2], o[3]);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float av[4] = {xa[q].x, xa[q].y, xa[q].z, xa[q].w};
#pragma unroll
                for (int to = 0; to < TO; to++) {
                    const float4 f = *reinterpret_cast<const float4*>(s_f + (((4 * T + q) * TO + to) * 64 + lane) * 4);
                    const float bv[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
                    for (int c = 0; c < 4; c++) acc[to] = MFMA(av[c], bv[c], acc[to]);
                }
            }
            if (ROLL) {
#pragma unroll
                for (int q = 0; q < 4; q++) xa[q] = xn[q];
            }
        }
        if (stats) {
#pragma unroll
            for (int to = 0; to < TO; to++) {
                const float pivot = s_pv[32 * to + j];
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float d = acc[to][r] - pivot;
                    if (rowbase + 8 * (r >> 2) + 4 * half + (r & 3) < p.rows) {
                        st1[to] += d;
                        st2[to] += d * d;
                    }
                }
            }
        }
        if (ST && p.y == nullptr) continue;   // statistics only (ec_linear_fwd_stats without y)
        // register r of lane (out j, half): row 8 (r >> 2) + 4 half + (r & 3) of the tile
#pragma unroll
        for (int to = 0; to < TO; to++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int64_t orow = rowbase + 8 * (r >> 2) + 4 * half + (r & 3);
                if (orow < p.rows && 32 * to + j < C) p.y[orow * C + 32 * to + j] = acc[to][r];
            }
    }
    if (stats) {   // the block's eight (wave, half) sums per output, added in a fixed order; the fragments are done with
        __syncthreads();
        float* s_red = lds;   // [2][8][32 TO]
#pragma unroll
        for (int to = 0; to < TO; to++) {
            s_red[(0 * 8 + wave * 2 + half) * 32 * TO + 32 * to + j] = st1[to];
            s_red[(1 * 8 + wave * 2 + half) * 32 * TO + 32 * to + j] = st2[to];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
            const int which = i / C, c = i - which * C;
            float sum = 0.0f;
            for (int k = 0; k < 2 * nw; k++) sum += s_red[(which * 8 + k) * 32 * TO + c];
            p.partials[((size_t)blockIdx.x * 2 + which) * C + c] = sum;
        }
    }
}

// Two layers in one kernel, for passes that keep nothing for a backward (GAIL's reward pass, gail_agent.py:165-170):
//     y = dropout(relu((x W1^T + b1) * scale + shift), p) W2^T + b2,       [rows, K] -> 128 -> 64
// The 128-wide pre-activations (512 B per row: written once and read once by the layer-by-layer pass, two thirds of its
// traffic) stay in registers.  Both layers run with the OUTPUTS as the MFMA's M dimension, as the fused learner kernels
// do (ec_mfma.h): lane (row j, half) then holds 16 of row j's outputs per 32-output tile in exactly the order in which
// the next layer's k-steps walk its inputs (kperm), so accumulator register r of tile T IS the B operand of step
// 16 T + r.  Per output the products are added in k_linear_fwd's order -- (T, q, c) ascending, k = 32 T + 8 q + c and
// + 4 paired in one MFMA -- and the tail is ec_linear_fwd_bn's arithmetic with its keep_draw, so y equals the two
// kernels' (ec_linear_fwd, then ec_linear_fwd_bn) bit for bit.  scale / shift come from a statistics-only pass of layer 1
// (ec_linear_fwd_stats with y = NULL) through ec_bn_finalize.
struct Lin2Args {
    const float* x;      // [rows, K]
    const float* w1;     // [128, K]
    const float* b1;     // [128] or NULL
    const float* scale;  // [128]
    const float* shift;  // [128]
    const float* w2;     // [64, 128]
    const float* b2;     // [64] or NULL
    int64_t rows;
    int K;
    float* y;            // [rows, 64]
    uint64_t seed;
    uint32_t thr24;
    float inv_keep;
    float* partials;     // optional [gridDim.x][2][64]: this block's column sums of (y - b2) and (y - b2)^2
    int n_partials;
};
constexpr int L2_C1 = 128, L2_C2 = 64, L2_TO1 = L2_C1 / 32, L2_TO2 = L2_C2 / 32;

template <int TI, bool VEC>   // VEC: k_in a multiple of 4 (rows 16-byte aligned)
__global__ void __launch_bounds__(256, 2) k_linear2_fwd_bn(Lin2Args p) {
    extern __shared__ __align__(16) float lds[];
    float* s_f1 = lds;                               // layer 1: 16 TI steps x 4 tiles x 64 lanes
    float* s_f2 = s_f1 + 16 * TI * L2_TO1 * 64;      // layer 2: 64 steps x 2 tiles x 64 lanes
    float* s_b1 = s_f2 + 64 * L2_TO2 * 64;           // [128]
    float* s_sc = s_b1 + L2_C1;                      // [128]
    float* s_sh = s_sc + L2_C1;                      // [128]
    float* s_b2 = s_sh + L2_C1;                      // [64]
    const int K = p.K;
    stage_frag(s_f1, p.w1, K, 1, L2_C1, K, 16 * TI, L2_TO1, true);
    stage_frag(s_f2, p.w2, L2_C1, 1, L2_C2, L2_C1, 64, L2_TO2, true);
    for (int i = threadIdx.x; i < L2_C1; i += blockDim.x) {
        s_b1[i] = p.b1 ? p.b1[i] : 0.0f;
        s_sc[i] = p.scale[i];
        s_sh[i] = p.shift[i];
        if (i < L2_C2) s_b2[i] = p.b2 ? p.b2[i] : 0.0f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int j = lane & 31, half = lane >> 5;
    constexpr bool vec = VEC;
    const int64_t n_tiles = (p.rows + 31) / 32;
    const bool stats = p.partials != nullptr;
    const uint32_t thr32 = p.thr24 << 8;   // (thr24 < 2^24)
    f32x16 st1[L2_TO2], st2[L2_TO2];   // lane (row j, half): sums over its rows of the 16 outputs it holds per tile
    acc_zero<L2_TO2>(st1);
    acc_zero<L2_TO2>(st2);
    // the rows of a tile, as the B operand of layer 1: lane (j, half) holds row j's inputs 32 T + 8 q + 4 half + (0..3).  No
    // branches (an index past the row's end reads its last element and is zeroed by a select); issued one tile ahead
    float4 xa[TI][4];
    const auto load_x = [&](int64_t tile_) __attribute__((always_inline)) {
        const int64_t row_ = tile_ * 32 + j;
        const bool ok_ = row_ < p.rows;
        const float* xr = p.x + (ok_ ? row_ : 0) * K;
#pragma unroll
        for (int T = 0; T < TI; T++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int k0 = 32 * T + 8 * q + 4 * half;
                if (vec) {
                    const float4 v = *reinterpret_cast<const float4*>(xr + (k0 < K ? k0 : 0));
                    const bool in = ok_ & (k0 < K);
                    xa[T][q] = make_float4(in ? v.x : 0.0f, in ? v.y : 0.0f, in ? v.z : 0.0f, in ? v.w : 0.0f);
                } else {
                    float v[4];
#pragma unroll
                    for (int c = 0; c < 4; c++) v[c] = xr[k0 + c < K ? k0 + c : K - 1];
                    xa[T][q] = make_float4((ok_ & (k0 + 0 < K)) ? v[0] : 0.0f, (ok_ & (k0 + 1 < K)) ? v[1] : 0.0f,
                                           (ok_ & (k0 + 2 < K)) ? v[2] : 0.0f, (ok_ & (k0 + 3 < K)) ? v[3] : 0.0f);
                }
            }
    };
    const int64_t tile0 = (int64_t)blockIdx.x * nw + wave, tile_step = (int64_t)gridDim.x * nw;
    load_x(tile0);
    for (int64_t tile = tile0; tile < n_tiles; tile += tile_step) {
        const int64_t row = tile * 32 + j;   // as the B operand, lane (j, half) is row j
        const bool ok = row < p.rows;
        // keep_draw(seed, row * 128 + col) with the per-row part hoisted, as in k_linear_fwd
        const uint64_t base = (uint64_t)(ok ? row : 0) * (uint64_t)L2_C1 + (uint64_t)(4 * half);
        const uint32_t kd_lo = (uint32_t)base, kd_mul = kd_lo * 0x9e3779b1u;
        const uint32_t hi = (uint32_t)(base >> 32) + (uint32_t)(p.seed >> 32);
        // (the seed's low word folded into the two per-row words; a carry out of the low index word iff off > ~kd_lo)
        const uint32_t kd_x0 = (hi * 0x85ebca77u) ^ (uint32_t)p.seed, kd_x1 = ((hi + 1u) * 0x85ebca77u) ^ (uint32_t)p.seed;
        const uint32_t kd_nlo = ~kd_lo;
        f32x16 a1[L2_TO1];
        acc_init<L2_TO1>(a1, s_b1, half);
        // A wave issues in order, so a stretch of vector arithmetic with no MFMA in it leaves the matrix pipe idle (and the
        // two waves of a SIMD run in step: both in their MFMA stretch, then both in their vector stretch -- 20.9 k cycles
        // per tile for 12.3 k of MFMAs).  The work is therefore cut into stages, one output tile of layer 1 (16 TI MFMAs)
        // or one input tile of layer 2 (32 MFMAs) each, and the tail of output tile `to` -- ~15 vector instructions per
        // element, independent of everything in flight -- is issued between the MFMAs of the following stage (pinned with
        // sched_group_barrier).  Per accumulator the order of the products is unchanged.
        const auto l1 = [&](int to) __attribute__((always_inline)) {
#pragma unroll
            for (int T = 0; T < TI; T++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const float bv[4] = {xa[T][q].x, xa[T][q].y, xa[T][q].z, xa[T][q].w};
                    const float4 f = *reinterpret_cast<const float4*>(s_f1 + (((4 * T + q) * L2_TO1 + to) * 64 + lane) * 4);
                    const float av[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
                    for (int c = 0; c < 4; c++) a1[to] = MFMA(av[c], bv[c], a1[to]);
                }
        };
        // the tail of layer 1 in place: register 4 q + c of tile `to` is column 32 to + 8 q + 4 half + c
        const auto tail = [&](int to) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int k0 = 32 * to + 8 * q + 4 * half;
                const float4 sc = *reinterpret_cast<const float4*>(s_sc + k0), sh = *reinterpret_cast<const float4*>(s_sh + k0);
                const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const float yv = a1[to][4 * q + c] * scv[c] + shv[c];
                    const uint32_t off = (uint32_t)(32 * to + 8 * q + c);
                    uint32_t h = (kd_mul + off * 0x9e3779b1u) ^ (off > kd_nlo ? kd_x1 : kd_x0);
                    h ^= h >> 16;
                    h *= 0x85ebca6bu;
                    h ^= h >> 13;
                    h *= 0xc2b2ae35u;
                    h ^= h >> 16;
                    // (no branches: a stage is one block for the scheduler; rows past the end are neither stored nor counted;
                    // (h >> 8) >= thr24 as one compare)
                    const bool keep = (h >= thr32) & (yv > 0.0f);
                    const float kept = yv * p.inv_keep;
                    a1[to][4 * q + c] = keep ? kept : 0.0f;
                }
            }
        };
        f32x16 a2[L2_TO2];
        const auto l2 = [&](int T) __attribute__((always_inline)) {
#pragma unroll
            for (int s4 = 4 * T; s4 < 4 * T + 4; s4++)
#pragma unroll
                for (int to = 0; to < L2_TO2; to++) {
                    const float4 f = *reinterpret_cast<const float4*>(s_f2 + ((s4 * L2_TO2 + to) * 64 + lane) * 4);
                    const float av[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
                    for (int c = 0; c < 4; c++) a2[to] = MFMA(av[c], a1[s4 >> 2][4 * (s4 & 3) + c], a2[to]);
                }
        };
        constexpr int TAIL_VALU = 16 * 15;   // vector instructions of one tail, spread over the stage's MFMAs
        l1(0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int to = 1; to < L2_TO1; to++) {
            l1(to);
            tail(to - 1);
#pragma unroll
            for (int i = 0; i < 16 * TI; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, (TAIL_VALU + 16 * TI - 1) / (16 * TI), 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        load_x(tile + tile_step);   // (layer 1 is done with this tile's rows; past the last tile: zeros, from row 0's addresses)
        acc_init<L2_TO2>(a2, s_b2, half);
        // (stage T = 0 of layer 2 does not read tile 3 of its input: that tile's tail rides along)
        l2(0);
        tail(L2_TO1 - 1);
#pragma unroll
        for (int i = 0; i < 32; i++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, (TAIL_VALU + 31) / 32, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int T = 1; T < L2_TO1; T++) {
            l2(T);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ok) {
#pragma unroll
            for (int to = 0; to < L2_TO2; to++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int o = 32 * to + 8 * q + 4 * half;
                    if (stats) {
                        const float4 pv = *reinterpret_cast<const float4*>(s_b2 + o);
                        const float pvv[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            const float d = a2[to][4 * q + c] - pvv[c];
                            st1[to][4 * q + c] += d;
                            st2[to][4 * q + c] += d * d;
                        }
                    }
                    *reinterpret_cast<float4*>(p.y + row * L2_C2 + o) =
                        make_float4(a2[to][4 * q + 0], a2[to][4 * q + 1], a2[to][4 * q + 2], a2[to][4 * q + 3]);
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
    }
    if (stats) {
        // column 32 to + 8 q + 4 half + c is register 4 q + c of tile `to` in the 32 lanes of `half` of every wave: the
        // block's 128 (wave, row lane) terms per column, added in a fixed order through LDS (the fragments are done with)
        float* s_red = lds;   // [256 threads][33]
        for (int which = 0; which < 2; which++) {
            __syncthreads();
#pragma unroll
            for (int to = 0; to < L2_TO2; to++)
#pragma unroll
                for (int r = 0; r < 16; r++) s_red[threadIdx.x * 33 + 16 * to + r] = which ? st2[to][r] : st1[to][r];
            __syncthreads();
            if (threadIdx.x < L2_C2) {
                const int c = threadIdx.x, to = c >> 5, q = (c >> 3) & 3, hf = (c >> 2) & 1, cc = c & 3;
                float sum = 0.0f;
                for (int w = 0; w < nw; w++)
                    for (int jj = 0; jj < 32; jj++) sum += s_red[(w * 64 + hf * 32 + jj) * 33 + 16 * to + 4 * q + cc];
                p.partials[((size_t)blockIdx.x * 2 + which) * L2_C2 + c] = sum;
            }
        }
    }
}

// First and second moments of the rows of a tall x [rows, K <= 96] about its first row x0, with the row index as the
// MFMA's K dimension: partials [block][KP * KP + KP] <- sum over the block's rows of d d^T (KP x KP, row-major) and of d,
// d = x[row] - x0, KP = 32 ceil(K / 32) (zero beyond K).  The batch statistics of y = x W^T + b follow without computing
// y: sum(y - y0) = W sum(d), sum((y - y0)^2)[o] = w_o^T (sum d d^T) w_o -- what the statistics-only pass of the GAIL
// discriminator's first layer needs (a quarter of that layer's MFMAs at K <= 32 and none of its per-element sums).
// At step s the lanes of half h hold row 2 s + h of the tile, lane j feature j: one coalesced read of two rows, and the
// same register is the A and the B operand.
struct MomArgs {
    const float* x;
    int64_t rows;
    int K;
    float* partials;
};

template <int TK>
__global__ void __launch_bounds__(256) k_moments(MomArgs p) {
    extern __shared__ __align__(16) float lds[];   // [KP * KP + KP]
    constexpr int KP = 32 * TK;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int j = lane & 31, half = lane >> 5;
    const int K = p.K;
    float x0[TK], sum[TK];
    f32x16 acc[TK][TK];
#pragma unroll
    for (int t = 0; t < TK; t++) {
        x0[t] = 32 * t + j < K ? p.x[32 * t + j] : 0.0f;
        sum[t] = 0.0f;
#pragma unroll
        for (int u = 0; u < TK; u++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[t][u][r] = 0.0f;
    }
    const int64_t n_tiles = (p.rows + 31) / 32;
    for (int64_t tile = (int64_t)blockIdx.x * nw + wave; tile < n_tiles; tile += (int64_t)gridDim.x * nw) {
        float v[16][TK];
        // every load of the tile is issued before the first is used (the pins: left to itself the compiler sinks each
        // load behind a branch on `in` and waits for it there -- sixteen serialised round trips per tile)
#pragma unroll
        for (int s = 0; s < 16; s++) {
            const int64_t row = tile * 32 + 2 * s + half;
#pragma unroll
            for (int t = 0; t < TK; t++) {
                const bool in = row < p.rows && 32 * t + j < K;
                v[s][t] = p.x[(in ? row : 0) * K + (32 * t + j < K ? 32 * t + j : 0)];
            }
        }
#pragma unroll
        for (int s = 0; s < 16; s++) {
            const int64_t row = tile * 32 + 2 * s + half;
#pragma unroll
            for (int t = 0; t < TK; t++) {
                const bool in = row < p.rows && 32 * t + j < K;
                asm volatile("" : "+v"(v[s][t]));
                v[s][t] = in ? v[s][t] - x0[t] : 0.0f;
            }
        }
#pragma unroll
        for (int s = 0; s < 16; s++)
#pragma unroll
            for (int t = 0; t < TK; t++) {
                sum[t] += v[s][t];
#pragma unroll
                for (int u = 0; u < TK; u++) acc[t][u] = MFMA(v[s][t], v[s][u], acc[t][u]);
            }
    }
    // the block's sums: the waves add theirs into LDS one after the other (a fixed order)
    for (int i = threadIdx.x; i < KP * KP + KP; i += blockDim.x) lds[i] = 0.0f;
    for (int w = 0; w < nw; w++) {
        __syncthreads();
        if (wave == w) {
            // register r of lane (j, half) of acc[t][u]: entry [32 t + 8 (r >> 2) + 4 half + (r & 3)][32 u + j]
#pragma unroll
            for (int t = 0; t < TK; t++) {
#pragma unroll
                for (int u = 0; u < TK; u++)
#pragma unroll
                    for (int r = 0; r < 16; r++) lds[(32 * t + 8 * (r >> 2) + 4 * half + (r & 3)) * KP + 32 * u + j] += acc[t][u][r];
                const float both = sum[t] + __shfl_xor(sum[t], 32);
                if (half == 0) lds[KP * KP + 32 * t + j] += both;
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < KP * KP + KP; i += blockDim.x) p.partials[(size_t)blockIdx.x * (KP * KP + KP) + i] = lds[i];
}

// The discriminator's input rows (obs | agent one-hot | action one-hot), gail_agent.py:131-141: a block writes 256
// consecutive rows, its threads walking the output elements in order (whole lines per store instruction).
// STAGED (obs_dim <= 63): the block's 256 observation rows and actions come in coalesced and wait in LDS, so that a
// thread's elements do not each pay a dependent global round trip (0.35 ms at 4.1 M rows was thirty of them in a row).
template <bool STAGED>
__global__ void __launch_bounds__(256) k_disc_rows(const float* __restrict__ obs, const int64_t* __restrict__ actions, int64_t rows,
                                                   int O, int N, int A, float* __restrict__ out) {
    extern __shared__ __align__(16) float lds[];
    float* s_obs = lds;                                   // [256][O]
    int* s_act = reinterpret_cast<int*>(lds + 256 * O);   // [256], -1: outside [0, A)
    const int W = O + N + A;
    for (int64_t r0 = (int64_t)blockIdx.x * 256; r0 < rows; r0 += (int64_t)gridDim.x * 256) {
        const int n = (int)(rows - r0 < 256 ? rows - r0 : 256);
        const int first_agent = (int)(r0 % N);
        if (STAGED) {
            __syncthreads();   // (the previous pass's reads)
            for (int i = threadIdx.x; i < n * O; i += 256) s_obs[i] = obs[r0 * O + i];
            if ((int)threadIdx.x < n) {
                const int64_t a = actions[r0 + threadIdx.x];
                s_act[threadIdx.x] = (a >= 0 && a < A) ? (int)a : -1;
            }
            __syncthreads();
        }
        // (row, column) of element j carried along instead of divided out per element
        const int dr = 256 / W, dc = 256 - dr * W;
        int r = (int)threadIdx.x / W, c = (int)threadIdx.x - r * W;
        for (int j = threadIdx.x; j < n * W; j += 256, r += dr, c += dc) {
            if (c >= W) {
                c -= W;
                r++;
            }
            float v;
            if (c < O) v = STAGED ? s_obs[r * O + c] : obs[(r0 + r) * O + c];
            else if (c < O + N) v = (c - O == (first_agent + r) % N) ? 1.0f : 0.0f;
            else if (STAGED) v = (c - O - N == s_act[r]) ? 1.0f : 0.0f;
            else v = ((int64_t)(c - O - N) == actions[r0 + r]) ? 1.0f : 0.0f;
            out[r0 * W + j] = v;
        }
    }
}

template <int TO, int TI, bool BN, bool ST, bool STG = false>
int launch_linear(const LinArgs& a, hipStream_t stream) {
    if constexpr (!STG && (TI <= 2 || !BN)) {   // the staged row load where it applies and wins (header of k_linear_fwd)
        if (a.K == 32 * TI && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0) return launch_linear<TO, TI, BN, ST, true>(a, stream);
    }
    const size_t lds = sizeof(float) * ((size_t)16 * TI * TO * 64 + 64 * TO + 64 * TI + (STG ? 4 * 32 * (32 * TI + 4) : 0));
    EC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_linear_fwd<TO, TI, BN, ST, STG>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t tiles = (a.rows + 31) / 32;
    int64_t blocks = (tiles + 3) / 4;
    const int64_t cap = STG ? 2048 : 1024;   // (staged: one tile in flight per wave, so more waves)
    if (blocks > cap) blocks = cap;
    if (a.partials) blocks = a.n_partials;   // one partial per block: the caller sized the buffer (blocks without rows write zeros)
    hipLaunchKernelGGL((k_linear_fwd<TO, TI, BN, ST, STG>), dim3((unsigned)blocks), dim3(256), lds, stream, a);
    EC_HIP(hipGetLastError());
    return EC_OK;
}

template <int TO, bool BN, bool ST>
int launch_linear_ti(const LinArgs& a, int TI, hipStream_t stream) {
    switch (TI) {
        case 1: return launch_linear<TO, 1, BN, ST>(a, stream);
        case 2: return launch_linear<TO, 2, BN, ST>(a, stream);
        case 3: return launch_linear<TO, 3, BN, ST>(a, stream);
        default: return launch_linear<TO, 4, BN, ST>(a, stream);
    }
}

template <bool BN, bool ST = false>
int launch_linear_to(const LinArgs& a, hipStream_t stream) {
    const int TI = (a.K + 31) / 32;
    switch ((a.C + 31) / 32) {
        case 1: return launch_linear_ti<1, BN, ST>(a, TI, stream);
        case 2: return launch_linear_ti<2, BN, ST>(a, TI, stream);
        case 3: return launch_linear_ti<3, BN, ST>(a, TI, stream);
        default: return launch_linear_ti<4, BN, ST>(a, TI, stream);
    }
}

template <int TI, bool VEC>
int launch_linear2(const Lin2Args& a, hipStream_t stream) {
    size_t lds = sizeof(float) * ((size_t)16 * TI * L2_TO1 * 64 + 64 * L2_TO2 * 64 + 3 * L2_C1 + L2_C2);
    const size_t red = sizeof(float) * 256 * 33;   // the statistics' reduction reuses the fragments' space
    if (lds < red) lds = red;
    EC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_linear2_fwd_bn<TI, VEC>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds));
    const int64_t tiles = (a.rows + 31) / 32;
    int64_t blocks = (tiles + 3) / 4;
    if (blocks > 1024) blocks = 1024;
    if (a.partials) blocks = a.n_partials;   // one partial per block (blocks without rows write zeros)
    hipLaunchKernelGGL((k_linear2_fwd_bn<TI, VEC>), dim3((unsigned)blocks), dim3(256), lds, stream, a);
    EC_HIP(hipGetLastError());
    return EC_OK;
}

template <int TK>
int launch_moments(const MomArgs& a, int n_partials, hipStream_t stream) {
    constexpr int KP = 32 * TK;
    const size_t lds = sizeof(float) * (KP * KP + KP);
    EC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_moments<TK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_moments<TK>), dim3((unsigned)n_partials), dim3(256), lds, stream, a);
    EC_HIP(hipGetLastError());
    return EC_OK;
}

}  // namespace

extern "C" int ec_linear_fwd(const float* x, const float* w, const float* b, int64_t rows, int32_t k_in, int32_t c_out, float* y,
                             void* stream) {
    if (rows == 0) return EC_OK;
    if (!x || !w || !y) return ec::fail(EC_EINVAL, "ec_linear_fwd: null pointer");
    if (rows < 0 || k_in < 1 || k_in > 128 || c_out < 1 || c_out > 128)
        return ec::fail(EC_EINVAL, "ec_linear_fwd: rows %lld, k_in %d, c_out %d (1..128 each)", (long long)rows, k_in, c_out);
    const LinArgs a{x, w, b, rows, k_in, c_out, y, nullptr, nullptr, 0, 0, 1.0f, nullptr, 0, nullptr};
    return launch_linear_to<false>(a, (hipStream_t)stream);
}

// ec_linear_fwd that also leaves the column sums of its output for a BatchNorm over y: partials [n_partials][2][c_out] <-
// per-block sums of (y - pivot) and (y - pivot)^2, the input of ec_bn_finalize with x_row0 = pivot.  pivot [c_out]: any
// row of y's scale (the caller computes y's first row with a one-row call).  y = NULL: the statistics only, nothing written
// (the same partials: for a layer whose output a later kernel recomputes, ec_linear2_fwd_bn).
extern "C" int ec_linear_fwd_stats(const float* x, const float* w, const float* b, int64_t rows, int32_t k_in, int32_t c_out,
                                   float* y, const float* pivot, int32_t n_partials, float* partials, void* stream) {
    if (!x || !w || !pivot || !partials) return ec::fail(EC_EINVAL, "ec_linear_fwd_stats: null pointer");   // (y may be NULL)
    if (rows < 1 || k_in < 1 || k_in > 128 || c_out < 1 || c_out > 128 || n_partials < 1 || n_partials > 1024)
        return ec::fail(EC_EINVAL, "ec_linear_fwd_stats: rows %lld, k_in %d, c_out %d (1..128 each), %d partials (1..1024)",
                        (long long)rows, k_in, c_out, n_partials);
    const LinArgs a{x, w, b, rows, k_in, c_out, y, nullptr, nullptr, 0, 0, 1.0f, partials, n_partials, pivot};
    return launch_linear_to<false, true>(a, (hipStream_t)stream);
}

// y = dropout(relu(x * scale + shift), p) W^T + b: the BatchNorm -> ReLU -> Dropout tail of the previous layer (scale, shift
// from ec_bn_finalize; the keep flags are ec_bn_relu_dropout_fwd's draws for the same seed) applied on the way in, so the
// activated [rows, k_in] tensor is never written.  For passes that keep nothing for a backward (GAIL's reward pass).
// partials (optional, [n_partials][2][c_out], 1 <= n_partials <= 1024 blocks): the column sums of (y - b) and (y - b)^2 per
// block, what ec_bn_col_stats would leave for ec_bn_finalize with the bias row as the pivot -- the statistics pass of the
// next BatchNorm without reading y again.  Needs b.
extern "C" int ec_linear_fwd_bn(const float* x, const float* scale, const float* shift, uint64_t seed, float p, const float* w,
                                const float* b, int64_t rows, int32_t k_in, int32_t c_out, float* y, int32_t n_partials,
                                float* partials, void* stream) {
    if (rows == 0) return EC_OK;
    if (!x || !scale || !shift || !w || !y) return ec::fail(EC_EINVAL, "ec_linear_fwd_bn: null pointer");
    if (rows < 0 || k_in < 1 || k_in > 128 || c_out < 1 || c_out > 128)
        return ec::fail(EC_EINVAL, "ec_linear_fwd_bn: rows %lld, k_in %d, c_out %d (1..128 each)", (long long)rows, k_in, c_out);
    if (!(p >= 0.0f && p < 1.0f)) return ec::fail(EC_EINVAL, "ec_linear_fwd_bn: dropout p %g outside [0, 1)", (double)p);
    if (partials && (!b || n_partials < 1 || n_partials > 1024))
        return ec::fail(EC_EINVAL, "ec_linear_fwd_bn: statistics need the bias and 1..1024 partials (got %d)", n_partials);
    const LinArgs a{x, w, b, rows, k_in, c_out, y, scale, shift, seed, (uint32_t)((double)p * 16777216.0), 1.0f / (1.0f - p),
                    partials, partials ? n_partials : 0, nullptr};
    return launch_linear_to<true>(a, (hipStream_t)stream);
}

// Layers 1 and 2 of the GAIL discriminator in one kernel (k_linear2_fwd_bn): y [rows, 64] =
// dropout(relu((x W1^T + b1) * scale + shift), p) W2^T + b2 with w1 [128, k_in], k_in <= 96, w2 [64, 128]; scale / shift
// [128] of the BatchNorm over layer 1's output (ec_linear_fwd_stats with y = NULL, then ec_bn_finalize), the keep flags
// ec_bn_relu_dropout_fwd's for `seed`.  partials (optional, [n_partials][2][64]): the column sums of (y - b2), (y - b2)^2
// per block for the next BatchNorm (pivot b2; needs b2).  Bit-equal in y to ec_linear_fwd followed by ec_linear_fwd_bn.
extern "C" int ec_linear2_fwd_bn(const float* x, const float* w1, const float* b1, const float* scale, const float* shift,
                                 uint64_t seed, float p, const float* w2, const float* b2, int64_t rows, int32_t k_in,
                                 int32_t c_mid, int32_t c_out, float* y, int32_t n_partials, float* partials, void* stream) {
    if (rows == 0) return EC_OK;
    if (!x || !w1 || !scale || !shift || !w2 || !y) return ec::fail(EC_EINVAL, "ec_linear2_fwd_bn: null pointer");
    if (rows < 0 || k_in < 1 || k_in > 96 || c_mid != L2_C1 || c_out != L2_C2)
        return ec::fail(EC_EINVAL, "ec_linear2_fwd_bn: rows %lld, k_in %d (1..96), widths %d -> %d (this build: 128 -> 64)",
                        (long long)rows, k_in, c_mid, c_out);
    if (!(p >= 0.0f && p < 1.0f)) return ec::fail(EC_EINVAL, "ec_linear2_fwd_bn: dropout p %g outside [0, 1)", (double)p);
    if (partials && (!b2 || n_partials < 1 || n_partials > 1024))
        return ec::fail(EC_EINVAL, "ec_linear2_fwd_bn: statistics need the bias and 1..1024 partials (got %d)", n_partials);
    const Lin2Args a{x, w1, b1, scale, shift, w2, b2, rows, k_in, y, seed, (uint32_t)((double)p * 16777216.0), 1.0f / (1.0f - p),
                     partials, partials ? n_partials : 0};
    const bool vec = (k_in & 3) == 0;
    switch ((k_in + 31) / 32) {
        case 1: return vec ? launch_linear2<1, true>(a, (hipStream_t)stream) : launch_linear2<1, false>(a, (hipStream_t)stream);
        case 2: return vec ? launch_linear2<2, true>(a, (hipStream_t)stream) : launch_linear2<2, false>(a, (hipStream_t)stream);
        default: return vec ? launch_linear2<3, true>(a, (hipStream_t)stream) : launch_linear2<3, false>(a, (hipStream_t)stream);
    }
}

// Moments of the rows of x [rows, k_in <= 96] about its first row (k_moments): partials [n_partials][KP * KP + KP], KP =
// 32 ceil(k_in / 32) <- per block, sum of d d^T (KP x KP row-major, zero beyond k_in) then sum of d, d = x[row] - x[0].
// The caller adds the blocks up (in float64) and derives the batch statistics of a Linear layer over x from them.
extern "C" int ec_row_moments(const float* x, int64_t rows, int32_t k_in, int32_t n_partials, float* partials, void* stream) {
    if (!x || !partials) return ec::fail(EC_EINVAL, "ec_row_moments: null pointer");
    if (rows < 1 || k_in < 1 || k_in > 96 || n_partials < 1 || n_partials > 4096)
        return ec::fail(EC_EINVAL, "ec_row_moments: rows %lld, k_in %d (1..96), %d partials (1..4096)", (long long)rows, k_in,
                        n_partials);
    const MomArgs a{x, rows, k_in, partials};
    switch ((k_in + 31) / 32) {
        case 1: return launch_moments<1>(a, n_partials, (hipStream_t)stream);
        case 2: return launch_moments<2>(a, n_partials, (hipStream_t)stream);
        default: return launch_moments<3>(a, n_partials, (hipStream_t)stream);
    }
}

// out [rows, obs_dim + n_agents + n_actions] = (obs row | one-hot of row % n_agents | one-hot of the row's action): the
// input of the GAIL discriminator for rows in [.., agent] order (an action outside [0, n_actions) sets nothing).
extern "C" int ec_disc_rows(const float* obs, const int64_t* actions, int64_t rows, int32_t obs_dim, int32_t n_agents,
                            int32_t n_actions, float* out, void* stream) {
    if (rows == 0) return EC_OK;
    if (!obs || !actions || !out) return ec::fail(EC_EINVAL, "ec_disc_rows: null pointer");
    if (rows < 0 || obs_dim < 1 || n_agents < 1 || n_actions < 1 || obs_dim + n_agents + n_actions > 4096)
        return ec::fail(EC_EINVAL, "ec_disc_rows: rows %lld, obs_dim %d, n_agents %d, n_actions %d", (long long)rows, obs_dim,
                        n_agents, n_actions);
    int64_t blocks = (rows + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (obs_dim <= 63)
        hipLaunchKernelGGL(k_disc_rows<true>, dim3((unsigned)blocks), dim3(256), 256 * (obs_dim + 1) * sizeof(float),
                           (hipStream_t)stream, obs, actions, rows, obs_dim, n_agents, n_actions, out);
    else
        hipLaunchKernelGGL(k_disc_rows<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, obs, actions, rows, obs_dim,
                           n_agents, n_actions, out);
    EC_HIP(hipGetLastError());
    return EC_OK;
}
