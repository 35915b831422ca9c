// Third translation unit of liblmc_hip.so: the tick kernels (externally evaluated log-densities, lmc_tick.hpp).
#include <hip/hip_runtime.h>

#include "lmc_tick.hpp"

namespace lmc {

int tick_launch(int ns, hipStream_t stream, const ChainArrays& A, const TickArrays& K, const SamplerParams& P,
                const double* logp, const double* grad) {
    const dim3 grid(A.chains), block(64);
    const int lds = 2 * A.dpad * 8;
    (void)hipGetLastError();
    switch (ns) {
        case 1: hipLaunchKernelGGL((tick_kernel<1>), grid, block, lds, stream, A, K, P, logp, grad); break;
        case 2: hipLaunchKernelGGL((tick_kernel<2>), grid, block, lds, stream, A, K, P, logp, grad); break;
        case 4: hipLaunchKernelGGL((tick_kernel<4>), grid, block, lds, stream, A, K, P, logp, grad); break;
        case 8: hipLaunchKernelGGL((tick_kernel<8>), grid, block, lds, stream, A, K, P, logp, grad); break;
        case 16: hipLaunchKernelGGL((tick_kernel<16>), grid, block, lds, stream, A, K, P, logp, grad); break;
        default: return -1;
    }
    return static_cast<int>(hipGetLastError());
}

int tick_launch_begin(int ns, hipStream_t stream, const ChainArrays& A, const TickArrays& K, long long iter_begin) {
    const dim3 grid(A.chains), block(64);
    (void)hipGetLastError();
    switch (ns) {
        case 1: hipLaunchKernelGGL((tick_begin_kernel<1>), grid, block, 0, stream, A, K, iter_begin); break;
        case 2: hipLaunchKernelGGL((tick_begin_kernel<2>), grid, block, 0, stream, A, K, iter_begin); break;
        case 4: hipLaunchKernelGGL((tick_begin_kernel<4>), grid, block, 0, stream, A, K, iter_begin); break;
        case 8: hipLaunchKernelGGL((tick_begin_kernel<8>), grid, block, 0, stream, A, K, iter_begin); break;
        case 16: hipLaunchKernelGGL((tick_begin_kernel<16>), grid, block, 0, stream, A, K, iter_begin); break;
        default: return -1;
    }
    return static_cast<int>(hipGetLastError());
}

int tick_launch_count(hipStream_t stream, const TickArrays& K, int chains) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(tick_count_kernel, dim3((chains + 255) / 256), dim3(256), 0, stream, K, chains);
    return static_cast<int>(hipGetLastError());
}

}  // namespace lmc
