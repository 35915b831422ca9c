// Host-callable launchers of the general ("wide") kernels (lmc_wide.hip); called by the C ABI in lmc_engine.hip.
// Return value: 0 = launched, kWideUnsupported = no such instantiation in this build, otherwise a hipError_t.
#pragma once
#include <hip/hip_runtime.h>

#include "lmc_dense_types.hpp"

namespace lmc {

constexpr int kWideUnsupported = -1;
constexpr int kWideBlock = 1024;        // threads per chain of the large team (16 wavefronts)
constexpr int kWideMaxDim = 16384;      // 1024 threads x 16 elements
constexpr int kWideOneWaveMaxDim = 512;  // one wavefront per chain up to here (8 elements per lane), the 16-wavefront team beyond
constexpr int kWideMaxDenseAdaptDim = 1024;  // FullAdapt: two float64 estimators, covariance, factor and work area per chain (26 MB at 1024)
constexpr int kWideMaxDenseDim = 2048;  // dense mass matrices: the operand vector is staged in LDS, the host factorises in O(d^3)

int wide_scratch_slots(int max_levels);
int wide_lds_bytes(int dpad);
// chains [P.chain_begin, P.chain_begin + n_chains) (n_chains <= 0: all)
int wide_launch_run(int family, int ns, int w, hipStream_t stream, const ChainArrays& A, const DenseArrays& D, const SamplerParams& P,
                    const double* tparams, int n_chains = 0);
int wide_launch_logp(int family, int ns, int w, hipStream_t stream, const ChainArrays& A, const double* tparams, const double* q,
                     double* logp, double* grad);
int wide_launch_trajectory(int family, int ns, int w, hipStream_t stream, const ChainArrays& A, const DenseArrays& D,
                           const double* tparams, const double* q0, const double* p0, int p0_is_f32, int sdot_mode, double eps,
                           int n_fwd, int n_back, double* oq, double* op, double* ov, double* og, double* oe, double* ol);
int wide_launch_momentum(int ns, int w, hipStream_t stream, const ChainArrays& A, const DenseArrays& D, int momentum_f32, double* out);
int wide_launch_mass_update(int ns, int w, hipStream_t stream, const ChainArrays& A, const SamplerParams& P);
struct TickArrays;
int tick_wide_launch(int ns, hipStream_t stream, const ChainArrays& A, const TickArrays& K, const SamplerParams& P,
                     const double* logp, const double* grad);
int tick_wide_launch_begin(int ns, hipStream_t stream, const ChainArrays& A, const TickArrays& K, long long iter_begin);

}  // namespace lmc
