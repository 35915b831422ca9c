// Device log-densities: the plug-in `logp_dlogp_func(q) -> (logp, dlogp)` of the reference
// (/root/reference/littlemcmc/integration.py:40,62,115) as a __device__ functor that is inlined
// into the leapfrog of the transition kernel.
//
// Contract (the "thread-distributed" form of the plug-in). A chain is owned by a team of 64*W threads
// (lmc_team.hpp; W = 1 for d <= 128); thread t = tm.tid() owns elements e = t*NS + s, s < NS. Elements with
// e >= d are padding: they arrive as 0 and MUST be returned as 0 in g.
//   template <int NS> struct Target {
//       static constexpr bool kLanePartial = ...;
//       template <class Team> __device__ void init(Team& tm, const double* params, int d);   // once per kernel
//       // logp must be team-uniform: reduce with tm.sum(); neighbours with tm.neighbours(); thread 0's value
//       // with tm.bcast0()
//       template <class Team> __device__ double logp_grad(Team& tm, const double (&q)[NS], double (&g)[NS]) const;
//       // only if kLanePartial: the per-thread partial p_t with logp = sum_t p_t; lets the integrator fuse this
//       // reduction with the kinetic-energy one
//       template <class Team> __device__ double logp_grad_partial(Team& tm, const double (&q)[NS], double (&g)[NS]) const;
//   };
// The CPU statements of the same densities are in oracle/targets.py (same operation order).
#pragma once
#include "lmc_team.hpp"

namespace lmc {

enum TargetFamily : int {
    kStdNormal = 0,
    kDiagGaussian = 1,
    kAR1 = 2,
    kFunnel = 3,
    kNormal1D = 4,
    kUser = 5,
};

// logp = -1/2 sum q^2 ; g = -q
template <int NS>
struct StdNormalTarget {
    static constexpr bool kLanePartial = true;
    template <class Team>
    __device__ void init(Team&, const double*, int) {}
    template <class Team>
    __device__ double logp_grad_partial(Team&, const double (&q)[NS], double (&g)[NS]) const {
        double part = 0.0;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            part = __builtin_fma(q[s], q[s], part);
            g[s] = -q[s];
        }
        return -0.5 * part;   // exact scaling: sum(-q^2/2) == -(sum q^2)/2 bit for bit
    }
    template <class Team>
    __device__ double logp_grad(Team& tm, const double (&q)[NS], double (&g)[NS]) const {
        return tm.sum(logp_grad_partial(tm, q, g));
    }
};

// g = -(prec*q) ; logp = 1/2 q.g  (params = prec[d])
template <int NS>
struct DiagGaussianTarget {
    static constexpr bool kLanePartial = true;
    double prec[NS];
    template <class Team>
    __device__ void init(Team& tm, const double* params, int d) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int e = tm.tid() * NS + s;
            prec[s] = (e < d) ? params[e] : 0.0;
        }
    }
    template <class Team>
    __device__ double logp_grad_partial(Team&, const double (&q)[NS], double (&g)[NS]) const {
        double part = 0.0;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            g[s] = -(prec[s] * q[s]);
            part = __builtin_fma(q[s], g[s], part);
        }
        return 0.5 * part;
    }
    template <class Team>
    __device__ double logp_grad(Team& tm, const double (&q)[NS], double (&g)[NS]) const {
        return tm.sum(logp_grad_partial(tm, q, g));
    }
};

// AR(1): (Pq)_i = (diag_i q_i + off q_{i-1}) + off q_{i+1}; g = -Pq; logp = 1/2 q.g
// params = {c_end, c_mid, off}
template <int NS>
struct AR1Target {
    static constexpr bool kLanePartial = true;
    // per-thread coefficient slices, fixed for the whole kernel: diagonal and coupling. Padding elements carry 0
    // coefficients, so the hot loop has no selects; ONE coupling slice serves both neighbours: the element below
    // e = 0 and the one above e = d-1 are exact zeros (team edge / padding), so their products vanish by themselves.
    double diag[NS], cpl[NS];
    template <class Team>
    __device__ void init(Team& tm, const double* params, int d) {
        const double c_end = params[0], c_mid = params[1], off = params[2];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int e = tm.tid() * NS + s;
            diag[s] = (e >= d) ? 0.0 : ((e == 0 || e == d - 1) ? c_end : c_mid);
            cpl[s] = (e < d) ? off : 0.0;
        }
    }
    template <class Team>
    __device__ double logp_grad(Team& tm, const double (&q)[NS], double (&g)[NS]) const {
        return tm.sum(logp_grad_partial(tm, q, g));
    }
    template <class Team>
    __device__ double logp_grad_partial(Team& tm, const double (&q)[NS], double (&g)[NS]) const {
        double below, above;   // element e-1 of this thread's first slot, e+1 of its last slot
        tm.neighbours(q[NS - 1], q[0], below, above);
        double part = 0.0;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const double prev = (s == 0) ? below : q[s - 1];
            const double next = (s == NS - 1) ? above : q[s + 1];
            // (diag q + off q_{e-1}) + off q_{e+1}; a zero coefficient adds +0.0, which leaves the sum unchanged
            const double pq = (diag[s] * q[s] + cpl[s] * prev) + cpl[s] * next;
            g[s] = -pq;
            part = __builtin_fma(q[s], g[s], part);
        }
        return 0.5 * part;
    }
};

// Neal's funnel: v = q_0, x = q_{1..d-1}
// Everything that depends on v alone -- e^{-v}, -v^2/18, -v/9 -- is wave-uniform scalar work on the critical path of every
// leapfrog (a chain in the funnel's neck builds 4095-leapfrog trees alone on its SIMD, where each dependent instruction
// costs a full pipeline latency: DESIGN.md section 6, C5). It is therefore kept short: the exponential is the table-driven
// uniform form (lmc_wave.hpp: ~15 VALU, < 1 ulp, instead of ~40 for the generic exp) and is issued BEFORE the reduction of
// sum x^2 so that the two overlap; the divisions by the constants 18 and 9 are multiplications by their reciprocals
// (~2 VALU instead of ~12 each; the result differs from numpy's quotient by at most one ulp, like the exponential's).
template <int NS>
struct FunnelTarget {
    static constexpr bool kLanePartial = false;
    int d;
    template <class Team>
    __device__ void init(Team&, const double*, int d_) { d = d_; }
    template <class Team>
    __device__ double logp_grad(Team& tm, const double (&q)[NS], double (&g)[NS]) const {
        const int t = tm.tid();
        const double v = tm.bcast0(q[0]);
        const double ev = exp_uniform_fast(fmax(-v, -700.0));   // (q_0 beyond 700 has diverged long before)
        double part = 0.0;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int e = t * NS + s;
            if (e > 0) part = __builtin_fma(q[s], q[s], part);
        }
        const double dm1 = static_cast<double>(d - 1);
        const double lin = -(v * v) * (1.0 / 18.0) - 0.5 * dm1 * v;      // the part of logp that needs no reduction
        const double g0_lin = -v * (1.0 / 9.0) - 0.5 * dm1;
        const double ssum = tm.sum(part);
        const double hes = 0.5 * ev * ssum;
        const double logp = lin - hes;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int e = t * NS + s;
            g[s] = (e == 0) ? (g0_lin + hes) : ((e < d) ? -(ev * q[s]) : 0.0);
        }
        return logp;
    }
};

// The reference's own test target (/root/reference/tests/test_utils.py:19-28), d == 1:
// logp = -z^2/2 - log(scale sqrt(2 pi)), z = (x-loc)/scale ; dlogp = -(x-loc)/scale (sic).
// params = {loc, scale}
template <int NS>
struct Normal1DTarget {
    static constexpr bool kLanePartial = false;
    double loc, scale, lognorm;
    template <class Team>
    __device__ void init(Team&, const double* params, int) {
        loc = params[0];
        scale = params[1];
        lognorm = log(scale * sqrt(2.0 * 3.141592653589793));
    }
    template <class Team>
    __device__ double logp_grad(Team& tm, const double (&q)[NS], double (&g)[NS]) const {
        const double x = tm.bcast0(q[0]);
        const double z = (x - loc) / scale;
#pragma unroll
        for (int s = 0; s < NS; ++s) g[s] = (tm.tid() == 0 && s == 0) ? -(x - loc) / scale : 0.0;
        return -0.5 * z * z - lognorm;
    }
};

}  // namespace lmc
