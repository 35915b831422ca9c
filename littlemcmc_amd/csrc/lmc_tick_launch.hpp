// Host-callable launchers of the tick kernels (lmc_tick.hip); called by the C ABI in lmc_engine.hip.
// Return value: 0 = launched, -1 = unsupported vector width, otherwise a hipError_t.
#pragma once
#include <hip/hip_runtime.h>

#include "lmc_sampler.hpp"

namespace lmc {

enum TickPhase : int { kTickStart = 0, kTickLeap = 1, kTickDone = 2 };
enum TickInt : int { kTiDepth = 0, kTiLeaf, kTiRight, kTiNLeap, kTiLStart, kTiRStart, kTiMaxDepth, kTiSteps, kNumTickInt };
enum TickDbl : int { kTdEps = 0, kTdStep, kTdE0, kTdLogp0, kTdPropE, kTdPropLogp, kTdCoff, kTdWStart, kTdWn, kTdAn,
                     kTdMaxDe, kTdPlen, kNumTickDbl };
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

int tick_launch(int ns, hipStream_t stream, const ChainArrays& A, const TickArrays& K, const SamplerParams& P,
                const double* logp, const double* grad);
int tick_launch_begin(int ns, hipStream_t stream, const ChainArrays& A, const TickArrays& K, long long iter_begin);
int tick_launch_count(hipStream_t stream, const TickArrays& K, int chains);   // K.n_active += chains not yet done

}  // namespace lmc
