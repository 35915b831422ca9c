// Host-callable launchers of the dense-mass kernels (lmc_dense.hip); called by the C ABI in lmc_engine.hip.
// Return value: 0 = launched, kDenseUnsupported = no such instantiation in this build, otherwise a hipError_t.
#pragma once
#include <hip/hip_runtime.h>

#include "lmc_dense_types.hpp"

namespace lmc {

constexpr int kDenseUnsupported = -1;

// run / adapt: chains [P.chain_begin, P.chain_begin + n_chains) resp. [chain_begin, chain_begin + n_chains); n_chains <= 0 = all
int dense_launch_run(int family, int ns, bool mat_f64, hipStream_t stream, const ChainArrays& A, const DenseArrays& D,
                     const SamplerParams& P, const double* tparams, int n_chains = 0);
// QuadPotentialFull with a matrix shared by all chains, dim <= 128: eight chains per workgroup, the per-leapfrog product on
// the matrix cores (lmc_dense_coop.hip). dense_coop_supported() says whether that kernel exists for the shape.
int dense_coop_supported(int family, int ns, int d, int dpad);
int dense_coop_lds_slots(int d, int dpad, int max_slots);
int dense_launch_run_coop(int family, int ns, hipStream_t stream, const ChainArrays& A, const DenseArrays& D,
                          const SamplerParams& P, const double* tparams, int n_chains = 0);
int dense_launch_trajectory(int family, int ns, bool mat_f64, hipStream_t stream, const ChainArrays& A,
                            const DenseArrays& D, const double* tparams, const double* q0, const double* p0,
                            int p0_is_f32, int sdot_mode, double eps, int n_fwd, int n_back, double* oq, double* op,
                            double* ov, double* og, double* oe, double* ol);
int dense_launch_momentum(int ns, hipStream_t stream, const ChainArrays& A, const DenseArrays& D, double* out);
int dense_launch_adapt(hipStream_t stream, const ChainArrays& A, const DenseArrays& D, double multiplier,
                       int update_window, int* mask = nullptr, int chain_begin = 0, int n_chains = 0, int expect_iter = -1);
// the tick kernel with a dense mass matrix (lmc_dense.hip: TickDenseMass + tick_step of lmc_tick.hpp)
struct TickArrays;
int tick_dense_launch(int ns, bool mat_f64, hipStream_t stream, const ChainArrays& A, const DenseArrays& D,
                      const TickArrays& K, const SamplerParams& P, const double* logp, const double* grad, int* adapt_mask);
int dense_launch_reset(hipStream_t stream, const ChainArrays& A, const DenseArrays& D, const void* cov1T,   // MatT per D.mat_f64
                       const void* fac1, const double* raw1T, const double* mean1, double weight, int window, int d8);

}  // namespace lmc
