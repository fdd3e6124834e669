// ec_linear.hip — y = x W^T + b for a tall batch: the forward of the GAIL discriminator's Linear layers
// (rl/policy_gradient_rl/gail/gail_network.py:97-125: 30 / 50 -> 128 -> 64 -> 32 on 4.1 M .. 65.5 M rows).
//
// At these shapes a layer is HBM-bound (read [rows, K], write [rows, C]); hipBLASLt runs it at ~3 TB/s.  Here a wave owns
// 32 rows at a time and the ROWS are the MFMA's M dimension: D[row][out] += A[row][k] B[k][out].  A lane fetches its row's
// inputs 16 bytes at a time in the order the MFMA walks them (kperm); the weights sit in LDS in the fragment order of
// ec_mfma.h (the lane index is the output, as there); and the accumulator leaves lane (out j, half h) holding 16 rows of
// output j -- so every store instruction writes 32 consecutive floats of two rows: whole 128-byte lines.  (The same
// kernel with outputs as M, as the fused learner kernels have it, stores 16-byte pieces and runs at hipBLASLt's rate.)
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
template <int TO, int TI, bool BN, bool ST>   // ST: the plain layer with the statistics epilogue (BN takes it at run time)
__global__ void __launch_bounds__(256) k_linear_fwd(LinArgs p) {
    extern __shared__ __align__(16) float lds[];
    float* s_f = lds;                      // fragments: 16 TI steps x TO tiles x 64 lanes
    float* s_b = s_f + 16 * TI * TO * 64;  // [32 TO] bias
    float* s_sc = s_b + 32 * TO;           // BN: [32 TI] scale, [32 TI] shift
    float* s_sh = s_sc + 32 * TI;
    float* s_pv = s_sh + 32 * TI;          // [32 TO] pivot of the statistics
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
    for (int64_t tile = (int64_t)blockIdx.x * nw + wave; tile < n_tiles; tile += (int64_t)gridDim.x * nw) {
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
                if (vec) {
                    xa[q] = (ok && k0 < K) ? *reinterpret_cast<const float4*>(xr + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
                } else {
                    xa[q].x = (ok && k0 + 0 < K) ? xr[k0 + 0] : 0.0f;
                    xa[q].y = (ok && k0 + 1 < K) ? xr[k0 + 1] : 0.0f;
                    xa[q].z = (ok && k0 + 2 < K) ? xr[k0 + 2] : 0.0f;
                    xa[q].w = (ok && k0 + 3 < K) ? xr[k0 + 3] : 0.0f;
                }
            }
        };
        constexpr bool ROLL = TI >= 3 && (BN || TO == 1);
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

// The discriminator's input rows (obs | agent one-hot | action one-hot), gail_agent.py:131-141: a block writes 256
// consecutive rows, its threads walking the output elements in order (whole lines per store instruction).
__global__ void __launch_bounds__(256) k_disc_rows(const float* __restrict__ obs, const int64_t* __restrict__ actions, int64_t rows,
                                                   int O, int N, int A, float* __restrict__ out) {
    const int W = O + N + A;
    for (int64_t r0 = (int64_t)blockIdx.x * 256; r0 < rows; r0 += (int64_t)gridDim.x * 256) {
        const int n = (int)(rows - r0 < 256 ? rows - r0 : 256);
        const int first_agent = (int)(r0 % N);
        for (int j = threadIdx.x; j < n * W; j += 256) {
            const int r = j / W, c = j - r * W;
            float v;
            if (c < O) v = obs[(r0 + r) * O + c];
            else if (c < O + N) v = (c - O == (first_agent + r) % N) ? 1.0f : 0.0f;
            else v = ((int64_t)(c - O - N) == actions[r0 + r]) ? 1.0f : 0.0f;
            out[r0 * W + j] = v;
        }
    }
}

template <int TO, int TI, bool BN, bool ST>
int launch_linear(const LinArgs& a, hipStream_t stream) {
    const size_t lds = sizeof(float) * ((size_t)16 * TI * TO * 64 + 64 * TO + 64 * TI);
    EC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_linear_fwd<TO, TI, BN, ST>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds));
    const int64_t tiles = (a.rows + 31) / 32;
    int64_t blocks = (tiles + 3) / 4;
    if (blocks > 1024) blocks = 1024;
    if (a.partials) blocks = a.n_partials;   // one partial per block: the caller sized the buffer (blocks without rows write zeros)
    hipLaunchKernelGGL((k_linear_fwd<TO, TI, BN, ST>), dim3((unsigned)blocks), dim3(256), lds, stream, a);
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
// row of y's scale (the caller computes y's first row with a one-row call).
extern "C" int ec_linear_fwd_stats(const float* x, const float* w, const float* b, int64_t rows, int32_t k_in, int32_t c_out,
                                   float* y, const float* pivot, int32_t n_partials, float* partials, void* stream) {
    if (!x || !w || !y || !pivot || !partials) return ec::fail(EC_EINVAL, "ec_linear_fwd_stats: null pointer");
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
    hipLaunchKernelGGL(k_disc_rows, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, obs, actions, rows, obs_dim, n_agents,
                       n_actions, out);
    EC_HIP(hipGetLastError());
    return EC_OK;
}
