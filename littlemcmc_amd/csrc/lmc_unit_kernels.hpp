// Target-dependent unit kernels (one wavefront per chain): the density itself and compute_state + leapfrog steps.
// They back the reference's protocol methods (step.integrator.compute_state / .step, calling a target object) and the
// unit tests; together with run_kernel (lmc_sampler.hpp) they are everything that depends on the density functor, i.e.
// what a run-time compiled user density instantiates (littlemcmc_amd/targets.py: UserTarget, hiprtc).
#pragma once
#include "lmc_sampler.hpp"

namespace lmc {

template <int NS, template <int> class TargetT>
__global__ __launch_bounds__(64) void logp_kernel(ChainArrays A, const double* tparams, const double* qin,
                                                  double* logp_out, double* grad_out) {
    const int c = blockIdx.x;
    const int lane = lane_id();
    const int d = A.d;
    Team<1> tm{nullptr, 0};
    TargetT<NS> tgt;
    tgt.init(tm, tparams, d);
    double q[NS], g[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int e = lane * NS + s;
        q[s] = (e < d) ? qin[static_cast<long long>(c) * d + e] : 0.0;
    }
    const double logp = tgt.logp_grad(tm, q, g);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int e = lane * NS + s;
        if (e < d) grad_out[static_cast<long long>(c) * d + e] = g[s];
    }
    if (lane == 0) logp_out[c] = logp;
}

// compute_state + n_fwd steps (+eps) + n_back steps (-eps); all states written out.
template <int NS, template <int> class TargetT>
__global__ __launch_bounds__(64) void trajectory_kernel(ChainArrays A, const double* tparams, const double* q0,
                                                        const double* p0, int p0_is_f32, int sdot_mode, double eps,
                                                        int n_fwd, int n_back, double* oq, double* op, double* ov,
                                                        double* og, double* oe, double* ol) {
    const int c = blockIdx.x;
    const int lane = lane_id();
    const int d = A.d;
    const long long row = static_cast<long long>(c) * A.dpad;
    Team<1> tm{nullptr, 0};
    TargetT<NS> tgt;
    tgt.init(tm, tparams, d);
    double q[NS], p[NS], g[NS];
    float var[NS];
    double vard[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int e = lane * NS + s;
        q[s] = (e < d) ? q0[static_cast<long long>(c) * d + e] : 0.0;
        p[s] = (e < d) ? p0[static_cast<long long>(c) * d + e] : 0.0;
        var[s] = A.var[row + e];
        vard[s] = static_cast<double>(var[s]);
    }
    const int n_states = n_fwd + n_back + 1;
    double logp = tgt.logp_grad(tm, q, g);
    double energy;
    double v[NS];
    if (p0_is_f32) {
        extern __shared__ __attribute__((aligned(16))) double lds[];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            p[s] = static_cast<double>(static_cast<float>(p[s]));
            v[s] = static_cast<double>(var[s] * static_cast<float>(p[s]));
        }
        const float kin = start_kinetic_f32<NS>(tm, p, var, d, sdot_mode, reinterpret_cast<float*>(lds), A.dpad);
        energy = static_cast<double>(kin) - logp;
    } else {
#pragma unroll
        for (int s = 0; s < NS; ++s) v[s] = static_cast<double>(var[s]) * p[s];
        energy = 0.5 * wave_sum(pdot_v<NS>(p, vard, p)) - logp;
    }
    for (int k = 0; k < n_states; ++k) {
        if (k > 0) {
            leapfrog<NS>(tm, tgt, vard, (k <= n_fwd) ? eps : -eps, q, p, g, energy, logp);
#pragma unroll
            for (int s = 0; s < NS; ++s) v[s] = static_cast<double>(var[s]) * p[s];
        }
        const long long base = (static_cast<long long>(c) * n_states + k) * d;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int e = lane * NS + s;
            if (e < d) {
                oq[base + e] = q[s];
                op[base + e] = p[s];
                ov[base + e] = v[s];
                og[base + e] = g[s];
            }
        }
        if (lane == 0) {
            oe[static_cast<long long>(c) * n_states + k] = energy;
            ol[static_cast<long long>(c) * n_states + k] = logp;
        }
    }
}

}  // namespace lmc
