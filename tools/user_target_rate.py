#!/usr/bin/env python3
"""GPU box: what a USER's density costs on the fused path (targets.UserTarget: a HIP snippet compiled with hiprtc at run time and
linked into the sampling kernel -- the plug-in path of north_star) next to the built-in functor of the same density: C3's job
(65 536 chains x d = 128 AR(1), tune 1000 + draws 1000) through sample()'s own job loop, draws not stored.
    python tools/user_target_rate.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import littlemcmc_amd as lmc  # noqa: E402
from littlemcmc_amd import _abi, sampling  # noqa: E402

# the density as a user would write it first: gradient and log-density in one pass, the reduction inside the functor
NAIVE = """
namespace lmc {
template <int NS>
struct UserTarget {
    static constexpr bool kLanePartial = false;
    double c_end, c_mid, off; int d;
    template <class Team> __device__ void init(Team&, const double* p, int d_) { c_end = p[0]; c_mid = p[1]; off = p[2]; d = d_; }
    template <class Team> __device__ double logp_grad(Team& tm, const double (&q)[NS], double (&g)[NS]) const {
        double below, above;
        tm.neighbours(q[NS - 1], q[0], below, above);
        double part = 0.0;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int e = tm.tid() * NS + s;
            const double prev = (s == 0) ? below : q[s - 1];
            const double next = (s == NS - 1) ? above : q[s + 1];
            const double diag = (e == 0 || e == d - 1) ? c_end : c_mid;
            const double pq = (e < d) ? ((diag * q[s] + off * prev) + off * next) : 0.0;
            g[s] = -pq;
            part = __builtin_fma(q[s], g[s], part);
        }
        return 0.5 * tm.sum(part);
    }
};
}
"""
# ... and following the functor contract's advice (csrc/lmc_targets.hpp): a log-density that is a plain lane sum hands back its
# per-lane partial (kLanePartial), so that it shares the pair's ONE batched reduction; coefficients are per-thread slices
TUNED = """
namespace lmc {
template <int NS>
struct UserTarget {
    static constexpr bool kLanePartial = true;
    double diag[NS], cpl[NS];
    template <class Team> __device__ void init(Team& tm, const double* p, int d) {
        for (int s = 0; s < NS; ++s) {
            const int e = tm.tid() * NS + s;
            diag[s] = (e >= d) ? 0.0 : ((e == 0 || e == d - 1) ? p[0] : p[1]);
            cpl[s] = (e < d) ? p[2] : 0.0;
        }
    }
    template <class Team> __device__ double logp_grad(Team& tm, const double (&q)[NS], double (&g)[NS]) const {
        return tm.sum(logp_grad_partial(tm, q, g));
    }
    template <class Team> __device__ double logp_grad_partial(Team& tm, const double (&q)[NS], double (&g)[NS]) const {
        double below, above;
        tm.neighbours(q[NS - 1], q[0], below, above);
        double part = 0.0;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const double prev = (s == 0) ? below : q[s - 1];
            const double next = (s == NS - 1) ? above : q[s + 1];
            const double pq = (diag[s] * q[s] + cpl[s] * prev) + cpl[s] * next;
            g[s] = -pq;
            part = __builtin_fma(q[s], g[s], part);
        }
        return 0.5 * part;
    }
};
}
"""

d, chains, tune, draws = 128, 65536, 1000, 1000
params = lmc.targets.AR1(d, 0.9).params
for name, tgt in (("built-in AR1Target", lmc.targets.AR1(d, 0.9)),
                  ("UserTarget, lane-partial functor (hiprtc)", lmc.targets.UserTarget(d, TUNED, params=params)),
                  ("UserTarget, reduction inside the functor (hiprtc)", lmc.targets.UserTarget(d, NAIVE, params=params))):
    seeds = lmc.distributed.global_seeds(20260928, chains)
    start, step = lmc.init_nuts(tgt, d, random_seed=seeds)
    for rep in range(2):
        eng = step._make_engine(chains)
        try:
            eng.seed(seeds)
            eng.set_position(start)
            eng.reset_tuning()
            eng.reserve(tune + draws, keep_trace=False)
            eng.synchronize()
            t0 = time.perf_counter()
            sampling._run_job(eng, tune, tune + draws, [100, 100, 100, 100, 500], False)
            dt = time.perf_counter() - t0
            leaps = float(eng.counters()[:, _abi.CT_LEAPFROGS].sum())
            lds = eng.run_lds_bytes()
        finally:
            eng.close()
    print("%-52s %.4e leapfrog-steps/s (%.2f s; LDS plan at the end: %d bytes)" % (name, leaps / dt, dt, lds))
