// Second translation unit of liblmc_hip.so: the dense-mass-matrix kernels (lmc_dense.hpp) and their launchers.
// Kept apart from lmc_engine.hip so that the two sets of kernel instantiations compile in parallel.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "../../include/lmc_hip.h"
#include "lmc_dense.hpp"
#include "lmc_tick.hpp"
#include "lmc_dense_launch.hpp"
#ifdef LMC_USER_TARGET_HEADER
#include LMC_USER_TARGET_HEADER
#endif

namespace lmc {

#ifdef LMC_USER_TARGET_HEADER
#define DENSE_USER_CASE(CALL) case LMC_TARGET_USER: { CALL(UserTarget); } break;
#else
#define DENSE_USER_CASE(CALL)
#endif

#if defined(LMC_USER_TARGET_HEADER) && defined(LMC_ONLY_USER)
#define DENSE_FAMILY_SWITCH(family, CALL) \
    switch (family) {                     \
        DENSE_USER_CASE(CALL)             \
        default: return kDenseUnsupported; \
    }
#else
#define DENSE_FAMILY_SWITCH(family, CALL)                                   \
    switch (family) {                                                       \
        case LMC_TARGET_STD_NORMAL: { CALL(StdNormalTarget); } break;       \
        case LMC_TARGET_DIAG_GAUSSIAN: { CALL(DiagGaussianTarget); } break; \
        case LMC_TARGET_AR1: { CALL(AR1Target); } break;                    \
        case LMC_TARGET_FUNNEL: { CALL(FunnelTarget); } break;              \
        case LMC_TARGET_NORMAL1D: { CALL(Normal1DTarget); } break;          \
        DENSE_USER_CASE(CALL)                                               \
        default: return kDenseUnsupported;                                  \
    }
#endif

#define DENSE_SHAPE_SWITCH(ns, mat_f64, BODY)                                              \
    if (mat_f64) {                                                                         \
        typedef double MatT;                                                               \
        switch (ns) {                                                                      \
            case 1: { constexpr int NS = 1; BODY; } break;                                 \
            case 2: { constexpr int NS = 2; BODY; } break;                                 \
            case 4: { constexpr int NS = 4; BODY; } break;                                 \
            default: return kDenseUnsupported;                                             \
        }                                                                                  \
    } else {                                                                               \
        typedef float MatT;                                                                \
        switch (ns) {                                                                      \
            case 1: { constexpr int NS = 1; BODY; } break;                                 \
            case 2: { constexpr int NS = 2; BODY; } break;                                 \
            case 4: { constexpr int NS = 4; BODY; } break;                                 \
            default: return kDenseUnsupported;                                             \
        }                                                                                  \
    }

int dense_launch_run(int family, int ns, bool mat_f64, hipStream_t stream, const ChainArrays& A, const DenseArrays& D,
                     const SamplerParams& P, const double* tparams, int n_chains) {
    const dim3 grid(n_chains > 0 ? n_chains : A.chains), block(64);
    const int lds = dense_lds_doubles(A.dpad) * 8 + D.cache_rows * A.dpad * (mat_f64 ? 8 : 4) + D.lds_slots * A.dpad * 8;
    (void)hipGetLastError();
#define RUN_CALL(T) \
    DENSE_SHAPE_SWITCH(ns, mat_f64, {                                                                               \
        if (lds > 64 * 1024) {                                                                                      \
            hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(&run_dense_kernel<NS, MatT, T>),     \
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, lds);                  \
            if (err != hipSuccess) return static_cast<int>(err);                                                    \
        }                                                                                                           \
        hipLaunchKernelGGL((run_dense_kernel<NS, MatT, T>), grid, block, lds, stream, A, D, P, tparams);            \
    })
    DENSE_FAMILY_SWITCH(family, RUN_CALL)
#undef RUN_CALL
    return static_cast<int>(hipGetLastError());
}

int dense_launch_trajectory(int family, int ns, bool mat_f64, hipStream_t stream, const ChainArrays& A,
                            const DenseArrays& D, const double* tparams, const double* q0, const double* p0,
                            int p0_is_f32, int sdot_mode, double eps, int n_fwd, int n_back, double* oq, double* op,
                            double* ov, double* og, double* oe, double* ol) {
    const dim3 grid(A.chains), block(64);
    const int lds = 2 * A.dpad * 8;
    (void)hipGetLastError();
#define TRAJ_CALL(T)                                                                                                   \
    DENSE_SHAPE_SWITCH(ns, mat_f64, hipLaunchKernelGGL((dense_trajectory_kernel<NS, MatT, T>), grid, block, lds, stream, A, D, \
                                                       tparams, q0, p0, p0_is_f32, sdot_mode, eps, n_fwd, n_back, oq, op, ov,  \
                                                       og, oe, ol))
    DENSE_FAMILY_SWITCH(family, TRAJ_CALL)
#undef TRAJ_CALL
    return static_cast<int>(hipGetLastError());
}

int dense_launch_momentum(int ns, hipStream_t stream, const ChainArrays& A, const DenseArrays& D, double* out) {
    const dim3 grid(A.chains), block(64);
    const int lds = 2 * A.dpad * 8;
    (void)hipGetLastError();
    switch (ns) {
        case 1: hipLaunchKernelGGL((dense_momentum_kernel<1>), grid, block, lds, stream, A, D, out); break;
        case 2: hipLaunchKernelGGL((dense_momentum_kernel<2>), grid, block, lds, stream, A, D, out); break;
        case 4: hipLaunchKernelGGL((dense_momentum_kernel<4>), grid, block, lds, stream, A, D, out); break;
        default: return kDenseUnsupported;
    }
    return static_cast<int>(hipGetLastError());
}

int dense_launch_adapt(hipStream_t stream, const ChainArrays& A, const DenseArrays& D, double multiplier,
                       int update_window, int* mask, int chain_begin, int n_chains, int expect_iter) {
    const int lds = dense_adapt_lds_bytes(A.d, A.dpad);
    const dim3 grid(n_chains > 0 ? n_chains : A.chains);
    (void)hipGetLastError();
    // lmc_config.tuning.chol_hbm (a test knob): the factorisation through HBM at every size an engine allocated the work area for,
    // so that the FullAdapt goldens of the small shapes check it against the register form (same factor bit for bit)
    const bool force_hbm = D.force_chol_hbm != 0 && D.chol_work != nullptr;
    if (D.mat_f64)   // QuadPotentialFullAdapt(dtype="float64"): covariance, factor and factorisation in float64 (general kernels)
        hipLaunchKernelGGL((dense_adapt_kernel<0, double>), grid, dim3(kCholHbmThreads), lds, stream, A, D, multiplier, update_window, mask, chain_begin, expect_iter);
    else if (dense_adapt_grid(A.d) == 8 && !force_hbm)
        hipLaunchKernelGGL(dense_adapt_kernel<8>, grid, dim3(64), lds, stream, A, D, multiplier, update_window, mask, chain_begin, expect_iter);
    else if (dense_adapt_grid(A.d) == 16 && !force_hbm)
        hipLaunchKernelGGL(dense_adapt_kernel<16>, grid, dim3(256), lds, stream, A, D, multiplier, update_window, mask, chain_begin, expect_iter);
    else if (dense_adapt_grid(A.d) == 32 && !force_hbm)
        hipLaunchKernelGGL(dense_adapt_kernel<32>, grid, dim3(1024), lds, stream, A, D, multiplier, update_window, mask, chain_begin, expect_iter);
    else   // the factorisation through HBM: d > 256 (or the test knob)
        hipLaunchKernelGGL(dense_adapt_kernel<0>, grid, dim3(kCholHbmThreads), lds, stream, A, D, multiplier, update_window, mask, chain_begin, expect_iter);
    return static_cast<int>(hipGetLastError());
}

// ---- the tick state machine (lmc_tick.hpp: tick_step) with a dense mass matrix: densities evaluated by the caller
// (targets.TorchTarget) sampled with QuadPotentialFull / FullInv / FullAdapt. The differences from the diagonal policy are
// the ones between lmc_sampler.hpp and lmc_dense.hpp: velocities are matrix sweeps and therefore stored with the trajectory
// ends and tree nodes, one sweep per leapfrog forms v = C p and w = C g, the momentum is a triangular solve (or L n), and
// FullAdapt's update of a chain that finished a tuning iteration in this tick runs in dense_adapt_kernel, launched masked
// by the host between two ticks.
// per-chain HBM row: 0-4 left end {q, p, g, v, w}, 5-9 right end, p_sum, proposal q, half-stepped momentum, the start
// state's stored velocity, then 6 vectors per subtree level {lp, lv, rp, rv, psum, proposal q} (tick_dense_scratch_vectors)
template <int NS, class MatT>
struct TickDenseMass {
    static constexpr bool kDense = true;
    static constexpr int kEndVecs = 5, kLevelVecs = 6, kPsum = kSlotPsum, kProp = kSlotProp, kHalf = kSlotHalf, kV0s = kSlotV0s,
                         kFixed = kTickDenseFixedSlots;
    static constexpr int kLp = 0, kLv = 1, kRp = 2, kRv = 3, kPs = 4, kQ = 5;
    const DenseArrays& D;
    int c;
    DenseMat<MatT> mm;
    __device__ __forceinline__ TickDenseMass(const ChainArrays& A, const DenseArrays& D_, int c_)
        : D(D_), c(c_), mm{static_cast<const MatT*>(D_.covT) + static_cast<long long>(c_) * D_.mat_stride, nullptr, 0, A.d, A.dpad} {}
    __device__ __forceinline__ void momentum(Team<1>&, int d, double* lds, bool, const double (&)[NS], double (&p0)[NS]) {
        lds_double* xop = (lds_double*)lds;   // the normals lie in lds[0, d) (TickWaveShape::normals)
        if (D.kind == kDenseFullInv)
            dense_momentum_inv<NS>(static_cast<const double*>(D.fac), d, mm.dpad, xop, p0);
        else
            dense_momentum_full<NS>(static_cast<const float*>(D.fac) + static_cast<long long>(c) * D.fac_stride, d, mm.dpad, xop, p0);
    }
    __device__ __forceinline__ double start_state(Team<1>& tm, double* lds, int, int, bool momentum_f32, int sdot_mode,
                                                  const double (&p0)[NS], const double (&g0)[NS], double logp0, double (&v0)[NS],
                                                  double (&w0)[NS], double (&v0s)[NS]) {
        return dense_start_state<NS, MatT>(tm, mm, lds, momentum_f32, sdot_mode, p0, g0, logp0, v0, w0, v0s);
    }
    __device__ __forceinline__ void velocity(double* lds, const double (&p)[NS], const double (&g)[NS], double (&v)[NS], double (&w)[NS]) {
        velocity2<NS, MatT>(mm, (lds_double*)lds, p, g, v, w);
    }
};

template <int NS, class MatT>
__global__ __launch_bounds__(64, dense_waves_per_simd(NS)) void tick_dense_kernel(ChainArrays A, DenseArrays D, TickArrays K,
                                                                                  SamplerParams P, const double* logp_in,
                                                                                  const double* grad_in, int* adapt_mask) {
    extern __shared__ __attribute__((aligned(16))) double lds[];   // 2 * dpad doubles: normals / sweep operands / sdot staging
    TickWaveShape shape(lds, A.dpad);
    TickDenseMass<NS, MatT> mass(A, D, static_cast<int>(blockIdx.x));
    tick_step<NS>(A, K, P, logp_in, grad_in, lds, shape, mass, D.kind == kDenseFullAdapt ? adapt_mask : nullptr);
}

int tick_dense_launch(int ns, bool mat_f64, hipStream_t stream, const ChainArrays& A, const DenseArrays& D,
                      const TickArrays& K, const SamplerParams& P, const double* logp, const double* grad, int* adapt_mask) {
    const dim3 grid(A.chains), block(64);
    const int lds = 2 * A.dpad * 8;
    (void)hipGetLastError();
    DENSE_SHAPE_SWITCH(ns, mat_f64, hipLaunchKernelGGL((tick_dense_kernel<NS, MatT>), grid, block, lds, stream, A, D, K, P, logp,
                                                       grad, adapt_mask))
    return static_cast<int>(hipGetLastError());
}

int dense_launch_reset(hipStream_t stream, const ChainArrays& A, const DenseArrays& D, const void* cov1T,
                       const void* fac1, const double* raw1T, const double* mean1, double weight, int window, int d8) {
    const int per_chain = (d8 * A.dpad + 255) / 256;
    (void)hipGetLastError();
    const dim3 grid(A.chains, per_chain < 64 ? per_chain : 64);
    if (D.mat_f64)
        hipLaunchKernelGGL(dense_reset_kernel<double>, grid, dim3(256), 0, stream, A, D, static_cast<const double*>(cov1T),
                           static_cast<const double*>(fac1), raw1T, mean1, weight, window, d8);
    else
        hipLaunchKernelGGL(dense_reset_kernel<float>, grid, dim3(256), 0, stream, A, D, static_cast<const float*>(cov1T),
                           static_cast<const float*>(fac1), raw1T, mean1, weight, window, d8);
    return static_cast<int>(hipGetLastError());
}

}  // namespace lmc
