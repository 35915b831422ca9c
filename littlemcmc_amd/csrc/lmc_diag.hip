// Fourth translation unit of liblmc_hip.so: cross-chain convergence diagnostics on the draws where they live (HBM).
//
// The reference has no diagnostics (ArviZ appears only in a docs recipe, /root/reference/docs/tutorials/
// framework_cookbook.rst:201-213): the definitions are this build's own -- split R-hat and the
// Geyer initial-monotone-sequence ESS of the Stan reference manual -- pinned against oracle/diagnostics_oracle.py.
// Everything reduces to per-dimension SUFFICIENT STATISTICS THAT ADD OVER CHAINS
//     sum_c mean_c, sum_c mean_c^2, sum_c var_c, sum_c acov_c[k]   (k = lag0 .. lag0 + 15)
// which is what makes the multi-GPU version one small all-reduce per block of 16 lags (littlemcmc_amd/diagnostics.py).
//
// chain_stats_kernel: one wavefront walks the sub-series [t0, t0 + n) of one chain for 64 dimensions (lane = dimension:
// a draw row is read as one coalesced 512-byte segment), first for the mean, then for the centred lagged products. The
// 16 lags of a pass are formed from register-resident windows -- the current block of 16 centred draws and the two
// blocks of the delayed stream -- with static indices only: 16 FMAs per draw and dimension, no ring-buffer moves.
// Blocks loop over chains and keep running sums; a second kernel adds the per-block partials in a fixed order, so the
// result is bit-reproducible (no floating-point atomics). Bound: HBM (the trace is read twice per pass; 16 loads of
// 512 B in flight per wave); algorithmic bytes = 2 x 8 x chains x n x dim per pass.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lmc_hip.h"

namespace lmc {

constexpr int kDiagLags = 16;                 // lags per pass
constexpr int kDiagRows = 3 + kDiagLags;      // sum mean, sum mean^2, sum var, sum acov[16]

__global__ __launch_bounds__(64) void chain_stats_kernel(const double* __restrict__ x, long long chains, long long draws_stride,
                                                         int dim, long long t0, long long n, int lag0, int nslab,
                                                         int chain_blocks, double* __restrict__ partial) {
    const int lane = static_cast<int>(threadIdx.x);
    const int slab = static_cast<int>(blockIdx.x) % nslab;
    const int cb = static_cast<int>(blockIdx.x) / nslab;
    const int j = slab * 64 + lane;             // this lane's dimension
    const bool live = j < dim;
    double s_mean = 0.0, s_mean_sq = 0.0, s_var = 0.0;
    double s_acov[kDiagLags];
#pragma unroll
    for (int k = 0; k < kDiagLags; ++k) s_acov[k] = 0.0;
    const double inv_n = 1.0 / static_cast<double>(n);

    for (long long c = cb; c < chains; c += chain_blocks) {
        const double* base = x + (c * draws_stride + t0) * dim + j;
        // ---- pass 1: mean (16 independent loads in flight)
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        long long t = 0;
        for (; t + 16 <= n; t += 16) {
            double v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = live ? base[(t + i) * dim] : 0.0;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i & 3] += v[i];
        }
        for (; t < n; ++t) acc[0] += live ? base[t * dim] : 0.0;
        const double mean = ((acc[0] + acc[1]) + (acc[2] + acc[3])) * inv_n;
        // ---- pass 2: centred lagged products, 16 draws per block
        double acov[kDiagLags];
#pragma unroll
        for (int k = 0; k < kDiagLags; ++k) acov[k] = 0.0;
        double w[kDiagLags];   // delayed stream, previous block: centred x at t = b*16 - lag0 - 16 + i
#pragma unroll
        for (int i = 0; i < kDiagLags; ++i) w[i] = 0.0;
        if (lag0 > 0) {        // the block of the delayed stream that precedes block 0 lies at t < 0: zeros
        }
        for (long long b = 0; b < n; b += kDiagLags) {
            double cur[kDiagLags], e[kDiagLags];   // e: delayed stream, current block: centred x at t = b - lag0 + i
#pragma unroll
            for (int i = 0; i < kDiagLags; ++i) {
                const long long tt = b + i;
                cur[i] = (live && tt < n) ? base[tt * dim] - mean : 0.0;
            }
            if (lag0 == 0) {
#pragma unroll
                for (int i = 0; i < kDiagLags; ++i) e[i] = cur[i];
            } else {
#pragma unroll
                for (int i = 0; i < kDiagLags; ++i) {
                    const long long tt = b - lag0 + i;
                    e[i] = (live && tt >= 0 && tt < n) ? base[tt * dim] - mean : 0.0;
                }
            }
            // acov[lag0 + k] += sum_i cur[i] * x(t_i - lag0 - k), the delayed value being e[i - k] or w[16 + i - k]
#pragma unroll
            for (int k = 0; k < kDiagLags; ++k) {
                double a = acov[k];
#pragma unroll
                for (int i = 0; i < kDiagLags; ++i) a = __builtin_fma(cur[i], (i >= k) ? e[i - k] : w[kDiagLags + i - k], a);
                acov[k] = a;
            }
#pragma unroll
            for (int i = 0; i < kDiagLags; ++i) w[i] = e[i];
        }
        s_mean += mean;
        s_mean_sq += mean * mean;
        if (lag0 == 0) s_var += acov[0] / static_cast<double>(n - 1);   // unbiased within-chain variance
#pragma unroll
        for (int k = 0; k < kDiagLags; ++k) s_acov[k] += acov[k] * inv_n;   // biased (1/n) autocovariance
    }
    double* out = partial + (static_cast<long long>(cb) * nslab + slab) * kDiagRows * 64 + lane;
    out[0] = s_mean;
    out[64] = s_mean_sq;
    out[128] = s_var;
#pragma unroll
    for (int k = 0; k < kDiagLags; ++k) out[(3 + k) * 64] = s_acov[k];
}

// out[row][dim] = sum over chain blocks, in block order (deterministic)
__global__ void chain_stats_reduce_kernel(const double* __restrict__ partial, int nslab, int chain_blocks, int dim,
                                          double* __restrict__ out) {
    const int idx = static_cast<int>(blockIdx.x) * static_cast<int>(blockDim.x) + static_cast<int>(threadIdx.x);
    if (idx >= kDiagRows * dim) return;
    const int row = idx / dim, j = idx % dim;
    const int slab = j / 64, lane = j % 64;
    double acc = 0.0;
    for (int cb = 0; cb < chain_blocks; ++cb)
        acc += partial[((static_cast<long long>(cb) * nslab + slab) * kDiagRows + row) * 64 + lane];
    out[idx] = acc;
}

}  // namespace lmc

extern "C" int lmc_diag_lags_per_pass(void) { return lmc::kDiagLags; }

// See include/lmc_hip.h. x and out are DEVICE pointers on the current device; the work is enqueued on `stream`.
extern "C" int lmc_diag_chain_stats(const double* x, int64_t chains, int64_t draws_stride, int32_t dim, int64_t t0, int64_t n,
                                    int32_t lag0, double* out, void* stream) {
    using namespace lmc;
    if (!x || !out || chains < 1 || dim < 1 || n < 2 || t0 < 0 || t0 + n > draws_stride || lag0 < 0) return LMC_ERR_INVALID;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nslab = (dim + 63) / 64;
    long long cbl = 4096 / nslab;                       // ~16 wavefronts per CU, each looping over chains
    if (cbl > chains) cbl = chains;
    if (cbl < 1) cbl = 1;
    const int chain_blocks = static_cast<int>(cbl);
    double* partial = nullptr;
    const size_t bytes = static_cast<size_t>(chain_blocks) * nslab * kDiagRows * 64 * sizeof(double);
    if (hipMallocAsync(reinterpret_cast<void**>(&partial), bytes, s) != hipSuccess) return LMC_ERR_HIP;
    (void)hipGetLastError();
    hipLaunchKernelGGL(chain_stats_kernel, dim3(chain_blocks * nslab), dim3(64), 0, s, x, static_cast<long long>(chains),
                       static_cast<long long>(draws_stride), dim, static_cast<long long>(t0), static_cast<long long>(n), lag0,
                       nslab, chain_blocks, partial);
    const int total = kDiagRows * dim;
    hipLaunchKernelGGL(chain_stats_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, s, partial, nslab, chain_blocks, dim, out);
    const hipError_t err = hipGetLastError();
    (void)hipFreeAsync(partial, s);
    return err == hipSuccess ? LMC_OK : LMC_ERR_HIP;
}
