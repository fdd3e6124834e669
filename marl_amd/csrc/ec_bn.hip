// ec_bn.hip — the tail of a GAIL-discriminator layer, BatchNorm1d (training mode) -> ReLU -> Dropout(p), as streaming
// kernels over [rows, C] fp32 (rl/policy_gradient_rl/gail/gail_network.py:59-132: fc -> bn -> activation -> dropout,
// three times, on 4.1 M .. 65.5 M rows per update).
//
// torch runs the tail as batch_norm (statistics, transform), relu, dropout (+ mask) forward and threshold_backward,
// masked_scale, batch_norm_backward (reduce, elementwise) backward: ~16 passes over a [rows, C] tensor per layer, and it
// keeps four such tensors for the backward.  Here, with the batch statistics given (one var_mean pass by the caller):
//   forward   out = relu(x * scale + shift) * keep / (1 - p)                          1 read + 1 write
//   backward  dy  = dout * keep / (1 - p) * [x * scale + shift > 0]
//             s1 = sum_rows dy,  s2 = sum_rows dy * x        (per-block partials)     2 reads
//             dx  = a * dy + b * x + c                                                2 reads + 1 write
// with a, b, c per column from s1, s2 and the statistics (the exact training-mode backward through mean and variance,
// assembled by the caller).  keep is either an explicit byte mask (tests replay the reference's draws) or a counter-based
// draw from (seed, element index), recomputed in the backward instead of stored.  HBM-bound by construction.
#include "ec_common.h"

namespace {

// Counter-based keep flag of element idx: a 32-bit mix (murmur3's finaliser) of the element index and the seed.  (The
// first version was splitmix64: two 64-bit multiplies per element are ~25 vector instructions on this machine, which
// made the 1-read-1-write forward kernel compute-bound at 2.5 TB/s.)
__device__ __forceinline__ bool keep_draw(uint64_t seed, uint64_t idx, uint32_t thr24) {
    uint32_t h = (uint32_t)idx * 0x9e3779b1u ^ (uint32_t)seed;
    h ^= ((uint32_t)(idx >> 32) + (uint32_t)(seed >> 32)) * 0x85ebca77u;
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return (h >> 8) >= thr24;  // uniform 24-bit draw >= p * 2^24  <=>  keep with probability 1 - p
}

struct BnArgs {
    const float* x;      // [rows, C]
    const float* dout;   // [rows, C] (backward)
    const float* scale;  // [C] invstd * weight
    const float* shift;  // [C] bias - mean * scale
    const uint8_t* mask; // [rows, C] keep flags, or NULL: keep_draw(seed, row * C + col)
    uint64_t seed;
    uint32_t thr24;
    float inv_keep;      // 1 / (1 - p)
    int64_t rows;
    int C;               // a multiple of 4, <= 256
    const float *a, *b, *c;  // [C] (dx)
    float* out;          // forward: out;  dx: dx;  sums: partials [blocks][2][C]
};

__device__ __forceinline__ void keep4(const BnArgs& p, int64_t e, bool (&k)[4]) {
    if (p.mask) {
        const uchar4 m = *reinterpret_cast<const uchar4*>(p.mask + e);
        k[0] = m.x != 0;
        k[1] = m.y != 0;
        k[2] = m.z != 0;
        k[3] = m.w != 0;
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++) k[j] = keep_draw(p.seed, (uint64_t)(e + j), p.thr24);
    }
}

// MODE 0: forward;  MODE 1: dx
template <int MODE>
__global__ void __launch_bounds__(256) k_bn_tail(BnArgs p) {
    const int q4 = p.C >> 2;
    const int64_t n4 = p.rows * q4;
    // C/4 divides the block size and the grid stride: a thread stays on one column quad
    const int col = (int)(threadIdx.x % q4) * 4;
    const float4 sc = *reinterpret_cast<const float4*>(p.scale + col), sh = *reinterpret_cast<const float4*>(p.shift + col);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = i * 4;
        const float4 x = *reinterpret_cast<const float4*>(p.x + e);
        bool k[4];
        keep4(p, e, k);
        const float xv[4] = {x.x, x.y, x.z, x.w}, scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
        float o[4];
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float y = xv[j] * scv[j] + shv[j];
                o[j] = (k[j] && y > 0.0f) ? y * p.inv_keep : 0.0f;
            }
        } else {
            const float4 d = *reinterpret_cast<const float4*>(p.dout + e);
            const float4 a = *reinterpret_cast<const float4*>(p.a + col), b = *reinterpret_cast<const float4*>(p.b + col),
                         c = *reinterpret_cast<const float4*>(p.c + col);
            const float dv[4] = {d.x, d.y, d.z, d.w}, av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w},
                        cv[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float y = xv[j] * scv[j] + shv[j];
                const float dy = (k[j] && y > 0.0f) ? dv[j] * p.inv_keep : 0.0f;
                o[j] = av[j] * dy + bv[j] * xv[j] + cv[j];
            }
        }
        *reinterpret_cast<float4*>(p.out + e) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// partials[blk][0][col] = sum over the block's rows of dy, [blk][1][col] of dy * x.  A thread owns one column quad and
// every (256 / (C/4))-th row of the block's range; the row lanes are summed through LDS in a fixed order.
__global__ void __launch_bounds__(256) k_bn_tail_sums(BnArgs p) {
    __shared__ float s_red[2][4][256];
    const int q4 = p.C >> 2;
    const int lanes = 256 / q4;                       // row lanes per block (C <= 256 and a power of two times 4)
    const int q = threadIdx.x % q4, rl = threadIdx.x / q4;
    const int col = q * 4;
    const int64_t per = (p.rows + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * per, r1 = r0 + per < p.rows ? r0 + per : p.rows;
    float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    if (rl < lanes) {
        const float4 sc = *reinterpret_cast<const float4*>(p.scale + col), sh = *reinterpret_cast<const float4*>(p.shift + col);
        const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
        for (int64_t r = r0 + rl; r < r1; r += lanes) {
            const int64_t e = r * p.C + col;
            const float4 x = *reinterpret_cast<const float4*>(p.x + e), d = *reinterpret_cast<const float4*>(p.dout + e);
            bool k[4];
            keep4(p, e, k);
            const float xv[4] = {x.x, x.y, x.z, x.w}, dv[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float y = xv[j] * scv[j] + shv[j];
                const float dy = (k[j] && y > 0.0f) ? dv[j] * p.inv_keep : 0.0f;
                s1[j] += dy;
                s2[j] += dy * xv[j];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        s_red[0][j][threadIdx.x] = s1[j];
        s_red[1][j][threadIdx.x] = s2[j];
    }
    __syncthreads();
    if (threadIdx.x < q4) {
#pragma unroll
        for (int w = 0; w < 2; w++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float t = 0.0f;
                for (int l = 0; l < lanes; l++) t += s_red[w][j][l * q4 + q];
                p.out[((size_t)blockIdx.x * 2 + w) * p.C + col + j] = t;
            }
    }
}

// Batch statistics in one pass: partials[blk][0][col] = sum over the block's rows of (x - k), [blk][1][col] of (x - k)^2,
// with the pivot k = x[0][col] (a sample of the column, so the sums stay of the size of the spread and
// var = S2/n - (S1/n)^2 does not cancel; the caller adds the partials up in float64).  torch's var_mean on a
// [rows, C] column reduction runs at a third of the HBM rate; this is the same thread mapping as k_bn_tail_sums.
__global__ void __launch_bounds__(256) k_bn_col_stats(BnArgs p) {
    __shared__ float s_red[2][4][256];
    const int q4 = p.C >> 2;
    const int lanes = 256 / q4;
    const int q = threadIdx.x % q4, rl = threadIdx.x / q4;
    const int col = q * 4;
    const int64_t per = (p.rows + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * per, r1 = r0 + per < p.rows ? r0 + per : p.rows;
    float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    if (rl < lanes) {
        const float4 k = *reinterpret_cast<const float4*>(p.x + col);
        const float kv[4] = {k.x, k.y, k.z, k.w};
        for (int64_t r = r0 + rl; r < r1; r += lanes) {
            const float4 x = *reinterpret_cast<const float4*>(p.x + r * p.C + col);
            const float xv[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float d = xv[j] - kv[j];
                s1[j] += d;
                s2[j] = fmaf(d, d, s2[j]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        s_red[0][j][threadIdx.x] = s1[j];
        s_red[1][j][threadIdx.x] = s2[j];
    }
    __syncthreads();
    if (threadIdx.x < q4) {
#pragma unroll
        for (int w = 0; w < 2; w++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float t = 0.0f;
                for (int l = 0; l < lanes; l++) t += s_red[w][j][l * q4 + q];
                p.out[((size_t)blockIdx.x * 2 + w) * p.C + col + j] = t;
            }
    }
}

int bn_check(const char* what, const BnArgs& a, float p) {
    if (!a.x || !a.scale || !a.shift || !a.out) return ec::fail(EC_EINVAL, "%s: null pointer", what);
    if (a.rows < 0 || a.C < 4 || a.C > 256 || (a.C & 3) || (256 % (a.C >> 2)))
        return ec::fail(EC_EINVAL, "%s: rows %lld, C %d (C must be 4..256 with C/4 dividing 256)", what, (long long)a.rows, a.C);
    if (!(p >= 0.0f && p < 1.0f)) return ec::fail(EC_EINVAL, "%s: dropout p = %g outside [0, 1)", what, p);
    return EC_OK;
}

int bn_grid(int64_t n4) {
    const int64_t b = (n4 + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

BnArgs bn_args(const float* x, const float* dout, const float* scale, const float* shift, const uint8_t* mask, uint64_t seed,
               float p, int64_t rows, int C, const float* a, const float* b, const float* c, float* out) {
    BnArgs r{x, dout, scale, shift, mask, seed, (uint32_t)((double)p * 16777216.0), 1.0f / (1.0f - p), rows, C, a, b, c, out};
    return r;
}

}  // namespace

extern "C" {

int ec_bn_relu_dropout_fwd(const float* x, const float* scale, const float* shift, const uint8_t* mask, uint64_t seed, float p,
                           int64_t rows, int32_t C, float* out, void* stream) {
    if (rows == 0) return EC_OK;
    const BnArgs a = bn_args(x, nullptr, scale, shift, mask, seed, p, rows, C, nullptr, nullptr, nullptr, out);
    if (int rc = bn_check("ec_bn_relu_dropout_fwd", a, p)) return rc;
    hipLaunchKernelGGL(k_bn_tail<0>, dim3(bn_grid(rows * (C >> 2))), dim3(256), 0, (hipStream_t)stream, a);
    EC_HIP(hipGetLastError());
    return EC_OK;
}

int ec_bn_col_stats(const float* x, int64_t rows, int32_t C, int32_t n_partials, float* partials, void* stream) {
    const BnArgs a = bn_args(x, nullptr, x, x, nullptr, 0, 0.0f, rows, C, nullptr, nullptr, nullptr, partials);
    if (int rc = bn_check("ec_bn_col_stats", a, 0.0f)) return rc;
    if (rows < 1 || n_partials < 1) return ec::fail(EC_EINVAL, "ec_bn_col_stats: rows %lld, n_partials %d", (long long)rows, n_partials);
    hipLaunchKernelGGL(k_bn_col_stats, dim3(n_partials), dim3(256), 0, (hipStream_t)stream, a);
    EC_HIP(hipGetLastError());
    return EC_OK;
}

int ec_bn_relu_dropout_bwd_sums(const float* dout, const float* x, const float* scale, const float* shift, const uint8_t* mask,
                                uint64_t seed, float p, int64_t rows, int32_t C, int32_t n_partials, float* partials,
                                void* stream) {
    if (rows == 0) return EC_OK;
    const BnArgs a = bn_args(x, dout, scale, shift, mask, seed, p, rows, C, nullptr, nullptr, nullptr, partials);
    if (int rc = bn_check("ec_bn_relu_dropout_bwd_sums", a, p)) return rc;
    if (!dout || n_partials < 1) return ec::fail(EC_EINVAL, "ec_bn_relu_dropout_bwd_sums: null dout or n_partials < 1");
    hipLaunchKernelGGL(k_bn_tail_sums, dim3(n_partials), dim3(256), 0, (hipStream_t)stream, a);
    EC_HIP(hipGetLastError());
    return EC_OK;
}

int ec_bn_relu_dropout_bwd_dx(const float* dout, const float* x, const float* scale, const float* shift, const uint8_t* mask,
                              uint64_t seed, float p, const float* a_, const float* b_, const float* c_, int64_t rows, int32_t C,
                              float* dx, void* stream) {
    if (rows == 0) return EC_OK;
    const BnArgs a = bn_args(x, dout, scale, shift, mask, seed, p, rows, C, a_, b_, c_, dx);
    if (int rc = bn_check("ec_bn_relu_dropout_bwd_dx", a, p)) return rc;
    if (!dout || !a_ || !b_ || !c_) return ec::fail(EC_EINVAL, "ec_bn_relu_dropout_bwd_dx: null pointer");
    hipLaunchKernelGGL(k_bn_tail<1>, dim3(bn_grid(rows * (C >> 2))), dim3(256), 0, (hipStream_t)stream, a);
    EC_HIP(hipGetLastError());
    return EC_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------------------------
// Weight gradient of a Linear layer over a tall batch: dW [Cout, K] = dy^T x, db [Cout] = sum_rows dy, for dy [rows, Cout],
// x [rows, K] fp32 with Cout a multiple of 32 up to 128 and K <= 128 (gail_network.py:97-125: 30/50 -> 128 -> 64 -> 32 on
// 4.1 M .. 65.5 M rows).  torch hands [Cout, rows] x [rows, K] to a GEMM whose contraction length is the row count; even
// chunked into a batched GEMM + sum it runs at a fifth of the HBM rate.  Here the row index is the MFMA's K dimension, as
// in the fused learner kernels, but the operands come straight from global memory: at step s the lanes of half h read 32
// consecutive floats of row 2 s + h of dy and of x (coalesced), and every (output tile, input tile) pair takes one
// 32x32x2 MFMA into a persistent accumulator.  One read of dy and x; per-block partial sums, added up by the caller.
namespace {

typedef float wg_f32x16 __attribute__((ext_vector_type(16)));

struct WgArgs {
    const float* dy;  // [rows, Cout]
    const float* x;   // [rows, K]
    int64_t rows;
    int Cout, K;
    float* partials;  // [blocks][Cout * K + Cout]
};

template <int TO, int TI>
__global__ void __launch_bounds__(256) k_linear_wgrad(WgArgs p) {
    __shared__ float s_red[TO * TI * 16 + TO][64];  // one wave's accumulators at a time (wave 0 adds them up, in wave order)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, half = lane >> 5;
    const int nw = blockDim.x >> 6;
    wg_f32x16 acc[TO][TI];
    float db[TO];
#pragma unroll
    for (int to = 0; to < TO; to++) {
        db[to] = 0.0f;
#pragma unroll
        for (int ti = 0; ti < TI; ti++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[to][ti][r] = 0.0f;
    }
    // rows are dealt out in chunks of 64 (32 steps) to the grid's waves
    const int64_t n_chunks = (p.rows + 63) / 64;
    for (int64_t chunk = (int64_t)blockIdx.x * nw + wave; chunk < n_chunks; chunk += (int64_t)gridDim.x * nw) {
        const int64_t r0 = chunk * 64;
        constexpr int U = 8;  // steps whose loads are issued together
#pragma unroll 1
        for (int s0 = 0; s0 < 32; s0 += U) {
            float a[U][TO], b[U][TI];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int64_t row = r0 + 2 * (s0 + u) + half;
                const bool ok = row < p.rows;
#pragma unroll
                for (int to = 0; to < TO; to++) a[u][to] = ok ? p.dy[row * p.Cout + 32 * to + i] : 0.0f;
#pragma unroll
                for (int ti = 0; ti < TI; ti++) b[u][ti] = (ok && 32 * ti + i < p.K) ? p.x[row * p.K + 32 * ti + i] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
#pragma unroll
                for (int to = 0; to < TO; to++) {
                    db[to] += a[u][to];
#pragma unroll
                    for (int ti = 0; ti < TI; ti++)
                        acc[to][ti] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][to], b[u][ti], acc[to][ti], 0, 0, 0);
                }
            }
        }
    }
    // block reduction in a fixed order (wave 1, 2, 3 into wave 0), then the block's partial sums
    for (int w = 1; w < nw; w++) {
        if (wave == w) {
#pragma unroll
            for (int to = 0; to < TO; to++) {
#pragma unroll
                for (int ti = 0; ti < TI; ti++)
#pragma unroll
                    for (int r = 0; r < 16; r++) s_red[(to * TI + ti) * 16 + r][lane] = acc[to][ti][r];
                s_red[TO * TI * 16 + to][lane] = db[to];
            }
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int to = 0; to < TO; to++) {
#pragma unroll
                for (int ti = 0; ti < TI; ti++)
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[to][ti][r] += s_red[(to * TI + ti) * 16 + r][lane];
                db[to] += s_red[TO * TI * 16 + to][lane];
            }
        }
        __syncthreads();
    }
    if (wave > 0) return;
    float* out = p.partials + (size_t)blockIdx.x * ((size_t)p.Cout * p.K + p.Cout);
#pragma unroll
    for (int to = 0; to < TO; to++) {
#pragma unroll
        for (int ti = 0; ti < TI; ti++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                // accumulator register r of lane (i, half): output 32 to + 8 (r >> 2) + 4 half + (r & 3), input 32 ti + i
                const int o = 32 * to + 8 * (r >> 2) + 4 * half + (r & 3), k = 32 * ti + i;
                if (k < p.K) out[(size_t)o * p.K + k] = acc[to][ti][r];
            }
        const float v = db[to] + __shfl_xor(db[to], 32);  // the two halves hold the two row parities of the same output
        if (half == 0) out[(size_t)p.Cout * p.K + 32 * to + i] = v;
    }
}

template <int TO>
int launch_wgrad_ti(const WgArgs& a, int TI, int blocks, hipStream_t stream) {
    switch (TI) {
        case 1: hipLaunchKernelGGL((k_linear_wgrad<TO, 1>), dim3(blocks), dim3(256), 0, stream, a); break;
        case 2: hipLaunchKernelGGL((k_linear_wgrad<TO, 2>), dim3(blocks), dim3(256), 0, stream, a); break;
        case 3: hipLaunchKernelGGL((k_linear_wgrad<TO, 3>), dim3(blocks), dim3(256), 0, stream, a); break;
        default: hipLaunchKernelGGL((k_linear_wgrad<TO, 4>), dim3(blocks), dim3(256), 0, stream, a); break;
    }
    return EC_OK;
}

}  // namespace

extern "C" int ec_linear_wgrad(const float* dy, const float* x, int64_t rows, int32_t c_out, int32_t k_in, int32_t n_partials,
                               float* partials, void* stream) {
    if (!dy || !x || !partials) return ec::fail(EC_EINVAL, "ec_linear_wgrad: null pointer");
    if (rows < 0 || c_out < 32 || c_out > 128 || (c_out & 31) || k_in < 1 || k_in > 128 || n_partials < 1 || n_partials > 65535)
        return ec::fail(EC_EINVAL, "ec_linear_wgrad: rows %lld, c_out %d (32, 64, 96 or 128), k_in %d (1..128), n_partials %d",
                        (long long)rows, c_out, k_in, n_partials);
    const WgArgs a{dy, x, rows, c_out, k_in, partials};
    const int TI = (k_in + 31) / 32;
    switch (c_out / 32) {
        case 1: launch_wgrad_ti<1>(a, TI, n_partials, (hipStream_t)stream); break;
        case 2: launch_wgrad_ti<2>(a, TI, n_partials, (hipStream_t)stream); break;
        case 3: launch_wgrad_ti<3>(a, TI, n_partials, (hipStream_t)stream); break;
        default: launch_wgrad_ti<4>(a, TI, n_partials, (hipStream_t)stream); break;
    }
    EC_HIP(hipGetLastError());
    return EC_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// The per-column arithmetic between the streaming passes, one launch each instead of ~20 / ~10 tiny torch launches (the
// 64-row expert batch of a GAIL step is otherwise host-launch-bound, and the big passes enqueue behind it):
//   ec_bn_finalize     partial sums of ec_bn_col_stats -> mean, biased var, invstd, scale = invstd * weight,
//                      shift = bias - mean * scale; running statistics updated as nn.BatchNorm1d does in training mode
//   ec_bn_bwd_coeffs   partial sums of ec_bn_relu_dropout_bwd_sums -> the a, b, c of the dx kernel, dweight, dbias
// Sums over the partials in float64, one thread per column.
namespace {

// 8 columns per block, 32 lanes per column: each lane adds up every 32nd partial, the lanes are combined in a fixed
// shuffle order (a single thread walking 1024 partials per column took ~100 us, on the critical path of every layer)
__device__ __forceinline__ void column_sums(const float* __restrict__ partials, int n_part, int C, int c, int lane, double& s1,
                                            double& s2) {
    s1 = s2 = 0.0;
    if (c < C)
        for (int b = lane; b < n_part; b += 32) {
            s1 += (double)partials[((size_t)b * 2 + 0) * C + c];
            s2 += (double)partials[((size_t)b * 2 + 1) * C + c];
        }
    for (int o = 16; o >= 1; o >>= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
    }
}

__global__ void __launch_bounds__(256) k_bn_finalize(const float* __restrict__ partials, int n_part, const float* __restrict__ x0,
                                                     int64_t rows, int C, const float* __restrict__ weight,
                                                     const float* __restrict__ bias, float eps, float momentum,
                                                     float* running_mean, float* running_var, const int64_t* num_batches,
                                                     float* mean, float* var, float* invstd, float* scale, float* shift) {
    const int lane = threadIdx.x & 31, c = blockIdx.x * 8 + (threadIdx.x >> 5);
    double s1, s2;
    column_sums(partials, n_part, C, c, lane, s1, s2);
    if (c >= C || lane != 0) return;
    s1 /= (double)rows;
    s2 /= (double)rows;
    const float m = (float)((double)x0[c] + s1);
    const double vd = s2 - s1 * s1;
    const float v = (float)(vd > 0.0 ? vd : 0.0);
    const float is = rsqrtf(v + eps);
    const float w = weight ? weight[c] : 1.0f, bb = bias ? bias[c] : 0.0f;
    const float sc = is * w;
    mean[c] = m;
    var[c] = v;
    invstd[c] = is;
    scale[c] = sc;
    shift[c] = bb - m * sc;
    if (running_mean && running_var) {
        // momentum < 0: cumulative moving average over the batches seen (momentum=None); the caller has counted this one
        const float mo = momentum >= 0.0f ? momentum : 1.0f / (float)(num_batches ? *num_batches : 1);
        const float unbiased = v * ((float)rows / (float)(rows > 1 ? rows - 1 : 1));
        running_mean[c] = running_mean[c] * (1.0f - mo) + m * mo;
        running_var[c] = running_var[c] * (1.0f - mo) + unbiased * mo;
    }
}

__global__ void __launch_bounds__(256) k_bn_bwd_coeffs(const float* __restrict__ partials, int n_part, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, const float* __restrict__ weight,
                                                       int64_t rows, int C, float* a, float* b, float* c_, float* dweight,
                                                       float* dbias) {
    const int lane = threadIdx.x & 31, c = blockIdx.x * 8 + (threadIdx.x >> 5);
    double t1, t2;
    column_sums(partials, n_part, C, c, lane, t1, t2);
    if (c >= C || lane != 0) return;
    const float s1 = (float)t1;                                      // sum dy
    const float s2 = ((float)t2 - mean[c] * s1) * invstd[c];          // sum dy * xhat
    const float av = (weight ? weight[c] : 1.0f) * invstd[c];
    const float bv = -av * invstd[c] * s2 / (float)rows;
    a[c] = av;
    b[c] = bv;
    c_[c] = -av * s1 / (float)rows - bv * mean[c];
    dweight[c] = s2;
    dbias[c] = s1;
}

}  // namespace

extern "C" int ec_bn_finalize(const float* partials, int32_t n_partials, const float* x_row0, int64_t rows, int32_t C,
                              const float* weight, const float* bias, float eps, float momentum, float* running_mean,
                              float* running_var, const int64_t* num_batches_tracked, float* mean, float* var, float* invstd,
                              float* scale, float* shift, void* stream) {
    if (!partials || !x_row0 || !mean || !var || !invstd || !scale || !shift) return ec::fail(EC_EINVAL, "ec_bn_finalize: null pointer");
    if (n_partials < 1 || rows < 1 || C < 1 || C > 256) return ec::fail(EC_EINVAL, "ec_bn_finalize: n_partials %d, rows %lld, C %d (<= 256)", n_partials, (long long)rows, C);
    hipLaunchKernelGGL(k_bn_finalize, dim3((C + 7) / 8), dim3(256), 0, (hipStream_t)stream, partials, n_partials, x_row0, rows, C, weight, bias,
                       eps, momentum, running_mean, running_var, num_batches_tracked, mean, var, invstd, scale, shift);
    EC_HIP(hipGetLastError());
    return EC_OK;
}

extern "C" int ec_bn_bwd_coeffs(const float* partials, int32_t n_partials, const float* mean, const float* invstd,
                                const float* weight, int64_t rows, int32_t C, float* a, float* b, float* c, float* dweight,
                                float* dbias, void* stream) {
    if (!partials || !mean || !invstd || !a || !b || !c || !dweight || !dbias) return ec::fail(EC_EINVAL, "ec_bn_bwd_coeffs: null pointer");
    if (n_partials < 1 || rows < 1 || C < 1 || C > 256) return ec::fail(EC_EINVAL, "ec_bn_bwd_coeffs: n_partials %d, rows %lld, C %d (<= 256)", n_partials, (long long)rows, C);
    hipLaunchKernelGGL(k_bn_bwd_coeffs, dim3((C + 7) / 8), dim3(256), 0, (hipStream_t)stream, partials, n_partials, mean, invstd, weight, rows, C,
                       a, b, c, dweight, dbias);
    EC_HIP(hipGetLastError());
    return EC_OK;
}
