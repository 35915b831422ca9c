// Third translation unit of liblmc_hip.so: the dense-mass sampling kernel for a matrix shared by all chains
// (QuadPotentialFull, /root/reference/littlemcmc/quadpotential.py:428-464) on the matrix cores. Eight one-wavefront
// chains share a workgroup and meet once per leapfrog for one v_mfma_f64_16x16x4_f64 product (lmc_dense.hpp:
// coop_product / run_dense_coop_kernel). A block here holds eight chains side by side, so "the thread's index in its
// chain" is the lane -- the one definition the shared device code needs to know about.
#include <hip/hip_runtime.h>

#define LMC_CHAIN_THREAD (static_cast<int>(threadIdx.x) & 63)
#define LMC_DENSE_COOP 1
#include "../../include/lmc_hip.h"
#include "lmc_dense.hpp"
#include "lmc_dense_launch.hpp"

namespace lmc {

int dense_coop_lds_slots(int d, int dpad, int max_slots) {   // tree slots per chain that fit next to the panels (one workgroup per CU)
    long slots = (160L * 1024 - coop_lds_bytes(d, dpad, 0)) / (static_cast<long>(kCoopWaves) * dpad * 8);
    if (slots > max_slots) slots = max_slots;
    return static_cast<int>(slots < 0 ? 0 : slots);
}

int dense_coop_supported(int family, int ns, int d, int dpad) {
    if (ns != 1 && ns != 2) return 0;                       // dpad <= 128: the float32 matrix fits one CU's LDS next to the panels
    if (coop_lds_bytes(d, dpad, 0) > 160 * 1024) return 0;
    switch (family) {
        case LMC_TARGET_STD_NORMAL: case LMC_TARGET_DIAG_GAUSSIAN: case LMC_TARGET_AR1: case LMC_TARGET_FUNNEL: return 1;
        default: return 0;
    }
}

int dense_launch_run_coop(int family, int ns, hipStream_t stream, const ChainArrays& A, const DenseArrays& D,
                          const SamplerParams& P, const double* tparams, int n_chains) {
    const int n = n_chains > 0 ? n_chains : A.chains;
    const dim3 grid((n + kCoopWaves - 1) / kCoopWaves), block(64 * kCoopWaves);
    const int lds = coop_lds_bytes(A.d, A.dpad, D.lds_slots);
    (void)hipGetLastError();
#define COOP_ONE(NSV, T)                                                                                             \
    {                                                                                                                \
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(&run_dense_coop_kernel<NSV, T>),          \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, lds);                       \
        if (err != hipSuccess) return static_cast<int>(err);                                                         \
        hipLaunchKernelGGL((run_dense_coop_kernel<NSV, T>), grid, block, lds, stream, A, D, P, tparams, n);          \
    }
#define COOP_CALL(T) { if (ns == 1) COOP_ONE(1, T) else COOP_ONE(2, T) }
    switch (family) {
        case LMC_TARGET_STD_NORMAL: COOP_CALL(StdNormalTarget) break;
        case LMC_TARGET_DIAG_GAUSSIAN: COOP_CALL(DiagGaussianTarget) break;
        case LMC_TARGET_AR1: COOP_CALL(AR1Target) break;
        case LMC_TARGET_FUNNEL: COOP_CALL(FunnelTarget) break;
        default: return kDenseUnsupported;
    }
#undef COOP_CALL
#undef COOP_ONE
    return static_cast<int>(hipGetLastError());
}

}  // namespace lmc
