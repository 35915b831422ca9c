// Host-callable launchers of the tick kernels (lmc_tick.hip); called by the C ABI in lmc_engine.hip.
// Return value: 0 = launched, -1 = unsupported vector width, otherwise a hipError_t.
#pragma once
#include <hip/hip_runtime.h>

#include "lmc_sampler.hpp"

namespace lmc {

enum TickPhase : int { kTickStart = 0, kTickLeap = 1, kTickDone = 2 };
enum TickInt : int { kTiDepth = 0, kTiLeaf, kTiRight, kTiNLeap, kTiLStart, kTiRStart, kTiMaxDepth, kTiSteps, kNumTickInt };
enum TickDbl : int { kTdEps = 0, kTdStep, kTdE0, kTdLogp0, kTdPropE, kTdPropLogp, kTdCoff, kTdWStart, kTdWn, kTdAn,
                     kTdMaxDe, kTdPlen, kTdCtot, kNumTickDbl };
constexpr int kTickLevels = 24;   // per-level subtree scalars kept per chain (max_treedepth <= 20)

struct TickArrays {
    int* phase;           // [C]
    long long* git;       // [C] iteration the chain is in
    long long iter_end;   // one past the last iteration of the current lmc_engine_tick_begin() request
    long long n_tune;
    int* ti;              // [C][kNumTickInt]
    double* td;           // [C][kNumTickDbl]
    double* lvl;          // [C][4][kTickLevels]  w, a, proposal energy, proposal logp of the parked subtrees
    double* q_eval;       // [C][d] unpadded: the points whose density is wanted (input of the callable)
    int* n_active;        // [1] chains that asked for another evaluation in the last tick
};
// per-chain HBM row (A.scratch): 0-2 left end {q, p, g}, 3-5 right end, 6 p_sum, 7 proposal q, 8 half-stepped momentum,
// then 4 vectors per subtree level {lp, rp, psum, proposal q}
constexpr int tick_scratch_vectors(int max_levels) { return 9 + 4 * max_levels; }

// rows of the caller's [chains][d] (unpadded) arrays <-> a thread's slice of a padded vector
template <int NS>
__device__ __forceinline__ void load_rows(const double* src, int d, int lane, double (&x)[NS]) {   // [d] unpadded
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int e = lane * NS + s;
        x[s] = (e < d) ? src[e] : 0.0;
    }
}
template <int NS>
__device__ __forceinline__ void store_rows(double* dst, int d, int lane, const double (&x)[NS]) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int e = lane * NS + s;
        if (e < d) dst[e] = x[s];
    }
}

int tick_launch(int ns, hipStream_t stream, const ChainArrays& A, const TickArrays& K, const SamplerParams& P,
                const double* logp, const double* grad);
int tick_launch_begin(int ns, hipStream_t stream, const ChainArrays& A, const TickArrays& K, long long iter_begin);
int tick_launch_count(hipStream_t stream, const TickArrays& K, int chains);   // K.n_active += chains not yet done

// per-chain HBM row of the dense tick kernel (lmc_dense.hip: TickDenseMass): 0-4 left end {q, p, g, v, w}, 5-9 right end, p_sum,
// proposal q, half-stepped momentum, the start state's stored velocity, then 6 vectors per subtree level
constexpr int kSlotPsum = 10, kSlotProp = 11, kSlotHalf = 12, kSlotV0s = 13, kTickDenseFixedSlots = 14;
constexpr int tick_dense_scratch_vectors(int max_levels) { return kTickDenseFixedSlots + 6 * max_levels; }

}  // namespace lmc
