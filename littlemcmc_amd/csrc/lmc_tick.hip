// Third translation unit of liblmc_hip.so: the tick kernels (externally evaluated log-densities, lmc_tick.hpp).
#include <hip/hip_runtime.h>

#include "lmc_tick.hpp"

namespace lmc {

// chains that still want evaluations (only launched when the host asks: one atomic per 256 chains, not per chain
// per tick -- 65 536 atomics on one address cost more than the rest of the tick)
__global__ __launch_bounds__(256) void tick_count_kernel(TickArrays K, int chains) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const bool active = c < chains && K.phase[c] != kTickDone;
    const unsigned long long m = ballot64(active);
    __shared__ int part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        const int n = part[0] + part[1] + part[2] + part[3];
        if (n) atomicAdd(K.n_active, n);
    }
}

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
