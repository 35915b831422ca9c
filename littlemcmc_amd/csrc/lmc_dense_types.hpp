// Types shared by the dense-mass kernels (lmc_dense.hpp, compiled in lmc_dense.hip) and the host side of the
// C ABI (lmc_engine.hip). No kernels here: both translation units include this file.
#pragma once
#include "lmc_sampler.hpp"

namespace lmc {

enum DenseKind : int { kDenseFull = 2, kDenseFullInv = 3, kDenseFullAdapt = 4 };   // == LMC_POT_FULL*

struct DenseArrays {
    int kind;
    void* covT;             // MatT [P][sweep_rows(d)][dpad]: transposed inverse mass matrix (P = chains for FullAdapt, else 1)
    void* fac;              // Full*: float Cholesky factor L of cov, row-major lower [P][d8][dpad] (rows >= d identity);
                            // FullInv: double LT[j][i] = L[i][j] of the mass matrix A = L L^T, [d][dpad]
    const double* fac_inv;  // Full with a shared matrix only (else nullptr): L^-1 of the float32 factor, float64 [sweep_rows(d)][dpad],
                            // formed once on the host in extended precision: solve_triangular(chol.T, z) = rows of L^-1 swept
                            // with z (the coop kernel's momentum draw: no 128-step dependent chain at the meeting point)
    long long mat_stride;   // elements between two chains' matrices (0 = shared)
    long long fac_stride;
    int cache_rows;         // leading rows of covT every wave keeps in LDS during a launch (host-chosen, lmc_engine.hip)
    int lds_slots;          // leading tree slots (trajectory ends, low subtree levels) kept in LDS instead of the HBM row
    // FullAdapt estimators (slot esel[c] = foreground)
    double* rawT;           // [2][C][d][dpad]   rawT[j][i] = raw_cov[i][j]
    double* emean;          // [2][C][dpad]
    double* en;             // [C][2]            n_samples of the two estimators
    int* esel;              // [C]
    int* prev_update;       // [C]
    int* window;            // [C]
    int* chol_failed;       // [C]  number of refreshes whose factorisation failed (old factor kept)
    void* chol_work;        // MatT [C][sweep_rows(d)][dpad] scratch of cholesky_hbm (d > 256 or float64, else nullptr): the factor, transposed
    int force_chol_hbm;     // lmc_config.tuning.chol_hbm: FullAdapt's refresh factorises through HBM whenever chol_work exists
    int mat_f64;            // covT / fac / chol_work hold doubles (FullInv, Full float64, FullAdapt(dtype="float64")), else floats
};

// FullAdapt's refresh factorises in registers up to here, through HBM beyond (lmc_dense.hpp: cholesky_registers / cholesky_hbm)
constexpr int kDenseAdaptRegisterMaxDim = 256;

// rows of a stored (transposed) inverse mass matrix: dim rounded up to two sweep batches, extra rows zero
constexpr int kSweepBatch = 8;
__host__ __device__ constexpr int sweep_rows(int d) { return (d + 2 * kSweepBatch - 1) / (2 * kSweepBatch) * (2 * kSweepBatch); }

// occupancy target of run_dense_kernel (waves per SIMD; register budget 512 / waves) and its fixed LDS carve.
// Measured: forcing more waves than the live state allows costs more in scratch spills than it gains in latency
// hiding (d = 128: 2 / 3 / 4 waves -> 6.6 / 6.2 / 5.7 e7 leapfrog-steps/s; d = 256: 1 / 2 waves -> 2.2 / 1.6 e7).
#ifndef LMC_DENSE_WAVES_NS2
#define LMC_DENSE_WAVES_NS2 2
#endif
#ifndef LMC_DENSE_WAVES_NS4
#define LMC_DENSE_WAVES_NS4 1
#endif
#ifndef LMC_DENSE_WAVES_NS1
#define LMC_DENSE_WAVES_NS1 2
#endif
constexpr int dense_waves_per_simd(int ns) { return ns <= 1 ? LMC_DENSE_WAVES_NS1 : ns == 2 ? LMC_DENSE_WAVES_NS2 : LMC_DENSE_WAVES_NS4; }
constexpr int dense_lds_doubles(int dpad) { return 2 * dpad + kLdsMtDoubles; }   // sweep operands / normals, MT19937 state

// per-chain HBM scratch row of the dense kernels: 2 trajectory ends x {q, p, g, v, w} + 6 vectors per subtree level
constexpr int dense_scratch_vectors(int max_levels) { return 10 + 6 * max_levels; }

}  // namespace lmc
