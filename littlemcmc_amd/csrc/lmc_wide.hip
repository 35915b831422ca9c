// Third translation unit of liblmc_hip.so: the general ("wide") kernels (lmc_wide.hpp) and their launchers -- model_ndim
// beyond 1024, dense mass matrices beyond 256 dimensions, float64 adaptive diagonals. Compiled in parallel with
// lmc_engine.hip and lmc_dense.hip.
#include <hip/hip_runtime.h>

#include "../../include/lmc_hip.h"
#include "lmc_wide.hpp"
#include "lmc_tick.hpp"
#include "lmc_wide_launch.hpp"
#ifdef LMC_USER_TARGET_HEADER
#include LMC_USER_TARGET_HEADER
#endif

namespace lmc {

static_assert(kWideBlock == kWideThreads, "launcher and kernel agree on the team size");

#ifdef LMC_USER_TARGET_HEADER
#define WIDE_USER_CASE(CALL) case LMC_TARGET_USER: { CALL(UserTarget); } break;
#else
#define WIDE_USER_CASE(CALL)
#endif

#if defined(LMC_USER_TARGET_HEADER) && defined(LMC_ONLY_USER)
#define WIDE_FAMILY_SWITCH(family, CALL) \
    switch (family) {                    \
        WIDE_USER_CASE(CALL)             \
        default: return kWideUnsupported; \
    }
#else
#define WIDE_FAMILY_SWITCH(family, CALL)                                    \
    switch (family) {                                                       \
        case LMC_TARGET_STD_NORMAL: { CALL(StdNormalTarget); } break;       \
        case LMC_TARGET_DIAG_GAUSSIAN: { CALL(DiagGaussianTarget); } break; \
        case LMC_TARGET_AR1: { CALL(AR1Target); } break;                    \
        case LMC_TARGET_FUNNEL: { CALL(FunnelTarget); } break;              \
        case LMC_TARGET_NORMAL1D: { CALL(Normal1DTarget); } break;          \
        WIDE_USER_CASE(CALL)                                                \
        default: return kWideUnsupported;                                   \
    }
#endif

#define WIDE_NS_CASE(n, ...) case n: { constexpr int NS = n; __VA_ARGS__; } break;
#define WIDE_NS_SWITCH(ns, ...)                                                                        \
    switch (ns) {                                                                                      \
        WIDE_NS_CASE(1, __VA_ARGS__) WIDE_NS_CASE(2, __VA_ARGS__) WIDE_NS_CASE(4, __VA_ARGS__)         \
        WIDE_NS_CASE(8, __VA_ARGS__) WIDE_NS_CASE(16, __VA_ARGS__)                                     \
        default: return kWideUnsupported;                                                              \
    }
// (elements per thread, wavefronts per chain): one wavefront with up to 8 elements per lane (model_ndim <= 512), the
// 16-wavefront team beyond -- the measured crossover (tools/wide_team_ab.py)
#define WIDE_SHAPE_SWITCH(ns, w, ...)                                                                  \
    if ((w) == 1) {                                                                                    \
        constexpr int W = 1;                                                                           \
        switch (ns) {                                                                                  \
            WIDE_NS_CASE(1, __VA_ARGS__) WIDE_NS_CASE(2, __VA_ARGS__) WIDE_NS_CASE(4, __VA_ARGS__)     \
            WIDE_NS_CASE(8, __VA_ARGS__)                                                               \
            default: return kWideUnsupported;                                                          \
        }                                                                                              \
    } else if ((w) == kWideWaves) {                                                                    \
        constexpr int W = kWideWaves;                                                                  \
        switch (ns) {                                                                                  \
            WIDE_NS_CASE(1, __VA_ARGS__) WIDE_NS_CASE(2, __VA_ARGS__) WIDE_NS_CASE(4, __VA_ARGS__)     \
            WIDE_NS_CASE(8, __VA_ARGS__) WIDE_NS_CASE(16, __VA_ARGS__)                                 \
            default: return kWideUnsupported;                                                          \
        }                                                                                              \
    } else return kWideUnsupported;

int wide_scratch_slots(int max_levels) { return wide_scratch_vectors(max_levels); }
int wide_lds_bytes(int dpad) { return wide_lds_doubles(dpad) * 8; }

#define WIDE_LDS_ATTR(KERNEL)                                                                                   \
    if (lds > 64 * 1024) {                                                                                      \
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(&KERNEL),                            \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, lds);                  \
        if (err != hipSuccess) return static_cast<int>(err);                                                    \
    }

int wide_launch_run(int family, int ns, int w, hipStream_t stream, const ChainArrays& A, const DenseArrays& D, const SamplerParams& P,
                    const double* tparams, int n_chains) {
    const dim3 grid(n_chains > 0 ? n_chains : A.chains), block(64 * w);
    const int lds = wide_lds_bytes(A.dpad);
    (void)hipGetLastError();
#define RUN_CALL(T) \
    WIDE_SHAPE_SWITCH(ns, w, { WIDE_LDS_ATTR((run_wide_kernel<NS, W, T>)) hipLaunchKernelGGL((run_wide_kernel<NS, W, T>), grid, block, lds, stream, A, D, P, tparams); })
    WIDE_FAMILY_SWITCH(family, RUN_CALL)
#undef RUN_CALL
    return static_cast<int>(hipGetLastError());
}

int wide_launch_logp(int family, int ns, int w, hipStream_t stream, const ChainArrays& A, const double* tparams, const double* q,
                     double* logp, double* grad) {
    const dim3 grid(A.chains), block(64 * w);
    const int lds = 2 * kWideWaves * kTeamSlots * 8;
    (void)hipGetLastError();
#define LOGP_CALL(T) WIDE_SHAPE_SWITCH(ns, w, hipLaunchKernelGGL((wide_logp_kernel<NS, W, T>), grid, block, lds, stream, A, tparams, q, logp, grad))
    WIDE_FAMILY_SWITCH(family, LOGP_CALL)
#undef LOGP_CALL
    return static_cast<int>(hipGetLastError());
}

int wide_launch_trajectory(int family, int ns, int w, hipStream_t stream, const ChainArrays& A, const DenseArrays& D,
                           const double* tparams, const double* q0, const double* p0, int p0_is_f32, int sdot_mode, double eps,
                           int n_fwd, int n_back, double* oq, double* op, double* ov, double* og, double* oe, double* ol) {
    const dim3 grid(A.chains), block(64 * w);
    const int lds = wide_lds_bytes(A.dpad);
    (void)hipGetLastError();
#define TRAJ_CALL(T)                                                                                                       \
    WIDE_SHAPE_SWITCH(ns, w, { WIDE_LDS_ATTR((wide_trajectory_kernel<NS, W, T>)) hipLaunchKernelGGL((wide_trajectory_kernel<NS, W, T>), grid, block, lds, stream, A, D, \
                                           tparams, q0, p0, p0_is_f32, sdot_mode, eps, n_fwd, n_back, oq, op, ov, og, oe, ol); })
    WIDE_FAMILY_SWITCH(family, TRAJ_CALL)
#undef TRAJ_CALL
    return static_cast<int>(hipGetLastError());
}

int wide_launch_momentum(int ns, int w, hipStream_t stream, const ChainArrays& A, const DenseArrays& D, int momentum_f32, double* out) {
    const dim3 grid(A.chains), block(64 * w);
    const int lds = wide_lds_bytes(A.dpad);
    (void)hipGetLastError();
    WIDE_SHAPE_SWITCH(ns, w, { WIDE_LDS_ATTR((wide_momentum_kernel<NS, W>)) hipLaunchKernelGGL((wide_momentum_kernel<NS, W>), grid, block, lds, stream, A, D, momentum_f32, out); })
    return static_cast<int>(hipGetLastError());
}

// ---- the tick state machine (lmc_tick.hpp: tick_step) for the shapes of the general kernels: externally evaluated densities
// (a Python callable, a batched torch callable) beyond 1024 dimensions. One chain = a workgroup of 16 wavefronts, model_ndim
// up to 16 384, diagonal mass matrices. What differs from the one-wavefront shape: the team's reductions and barriers, the
// normals drawn 1024 at a time by wave 0 (numpy's stream is sequential), the uniform stream shared by the team.
struct TickWideShape {
    typedef WideTeam TeamT;
    static constexpr int kThreads = kWideThreads;
    TeamT tm;
    double* bcast;
    __device__ __forceinline__ TickWideShape(double* lds, int dpad) {
        tm.xbuf = lds + wide_stage_doubles(dpad);
        tm.parity = 0;
        bcast = tm.xbuf + 2 * kWideWaves * kTeamSlots;
    }
    template <int NS>
    __device__ __forceinline__ void normals(RngState& rng, int d, int, double* lds, double (&z)[NS]) {
        wide_normals_regs<NS>(tm, rng, d, lds, bcast, z);
    }
};

template <int NS>
__global__ __launch_bounds__(kWideThreads, 1) void tick_wide_kernel(ChainArrays A, TickArrays K, SamplerParams P, const double* logp_in,
                                                                     const double* grad_in) {
    extern __shared__ __attribute__((aligned(16))) double lds[];   // wide_stage_doubles(dpad): normals chunk + staging / sdot staging; team exchange; broadcast words
    TickWideShape shape(lds, A.dpad);
    TickDiagMass<NS> mass(A, static_cast<long long>(blockIdx.x) * A.dpad, static_cast<int>(threadIdx.x));
    tick_step<NS>(A, K, P, logp_in, grad_in, lds, shape, mass, nullptr);
}

int tick_wide_launch(int ns, hipStream_t stream, const ChainArrays& A, const TickArrays& K, const SamplerParams& P,
                     const double* logp, const double* grad) {
    const dim3 grid(A.chains), block(kWideThreads);
    const int lds = wide_lds_bytes(A.dpad);
    (void)hipGetLastError();
    WIDE_NS_SWITCH(ns, { WIDE_LDS_ATTR((tick_wide_kernel<NS>)) hipLaunchKernelGGL((tick_wide_kernel<NS>), grid, block, lds, stream, A, K, P, logp, grad); })
    return static_cast<int>(hipGetLastError());
}

int tick_wide_launch_begin(int ns, hipStream_t stream, const ChainArrays& A, const TickArrays& K, long long iter_begin) {
    const dim3 grid(A.chains), block(kWideThreads);
    (void)hipGetLastError();
    WIDE_NS_SWITCH(ns, hipLaunchKernelGGL((tick_begin_kernel<NS, kWideThreads>), grid, block, 0, stream, A, K, iter_begin))
    return static_cast<int>(hipGetLastError());
}

int wide_launch_mass_update(int ns, int w, hipStream_t stream, const ChainArrays& A, const SamplerParams& P) {
    const dim3 grid(A.chains), block(64 * w);
    (void)hipGetLastError();
    WIDE_SHAPE_SWITCH(ns, w, hipLaunchKernelGGL((wide_mass_update_kernel<NS, W>), grid, block, 0, stream, A, P))
    return static_cast<int>(hipGetLastError());
}

}  // namespace lmc
