// The NUTS transition in LEAF form, stated ONCE for every kernel family that does not use the fused pair form of
// lmc_sampler.hpp (nuts_transition2): the dense-mass kernels (lmc_dense.hpp: one wavefront per chain, node vectors in registers,
// stack in LDS / the scratch row, stored velocities) and the general kernels (lmc_wide.hpp: teams of 1 or 16 wavefronts, every
// vector in the chain's scratch row). Until round 5 each of the two carried its own copy of these ~150 statements; they differ
// only in WHERE a vector lives and in HOW a state is integrated, which is what a policy object now answers.
//
//   nuts.py:204-224 (the doubling loop), _Tree :251-435 in the iterative post-order form of SURVEY.md appendix A.4:
//   leaves are integrated one at a time; leaf i closes the right children of levels 0 .. ctz(~i)-1, each merge is
//   nuts.py:377-417 (sub-U-turn checks :389-396, uniform-within-subtree proposal :404), the accepted subtree is merged into
//   the trajectory by nuts.py:321-340 (biased progressive proposal :322, in-place momentum sum :329, the three U-turn checks
//   with aliased operands :332-340). Weights are kept in the linear domain with one offset per transition (lmc_sampler.hpp).
//
// Policy P (all members device functions; NS = P::kNS elements of every vector per thread):
//   typename P::End                      a trajectory end / the state being integrated: members q, p, g, v [NS] (+ whatever the
//                                        integrator carries along, e.g. the dense kernels' w = C g)
//   double uniform()                     next uniform of the chain's stream (math.py:21-25's np.random.uniform())
//   bool any_nonpositive2(a, b), any_nonpositive6(d[6])     team-wide sums of per-thread partial dots, any <= 0
//   void start_state(End&)               the start State (integration.py:52-66) as the integrator wants it (hmc_transition_any)
//   void accept_state(const End&)        the chain's position becomes this state's (hmc_transition_any)
//   void end_load(side, End&) / end_store(side, const End&)  the trajectory's ends, side 0 = left, 1 = right
//   void end_velocity(side, v[NS])       the velocity STORED with that end's State (the float32 start velocity while the end
//                                        still is the start state: SURVEY A.2) -- operand of the U-turn checks
//   void end_momentum(side, p[NS])
//   void leapfrog(eps, End&, energy&, logp&)                 integration.py:100-121
//   node_ld<F>(x) / node_st<F>(x)        the subtree node under construction, F in {kNodeLp, kNodeLv, kNodePs, kNodeQ}: left-end
//                                        momentum / velocity, momentum sum, proposal position (its right end is the state being
//                                        integrated)
//   level_ld(j, f, x) / level_st(j, f, x)   subtree-stack level j, f in NodeField order {lp, lv, rp, rv, psum, q}
//   psum_ld(x) / psum_st(x)              the trajectory's running momentum sum (nuts.py:329)
//   proposal_from_node()                 the transition's proposal position <- the node's proposal position
#pragma once
#include "lmc_sampler.hpp"

namespace lmc {

enum NodeField : int { kNodeLp = 0, kNodeLv = 1, kNodeRp = 2, kNodeRv = 3, kNodePs = 4, kNodeQ = 5 };

// (The HMC transition, hmc.py:140-182, is stated once for ALL families, the fused one included: hmc_transition_any in
//  lmc_sampler.hpp; the policies below serve both.)

// ---- NUTS transition, leaf form ------------------------------------------------------------------------------------------
// In: both trajectory ends hold the start state, psum = p0, the proposal position = q. Out: the proposal position.
template <class P>
__device__ inline void leaf_nuts_transition(P& pol, double e0, double logp0, double step_size, double emax, int max_depth,
                                            bool momentum_f32, TransitionOut& out) {
    constexpr int NS = P::kNS;
    double prop_e = e0, prop_logp = logp0;
    double coff = 0.0, w_start = 1.0, wn = 0.0, an = 0.0, max_de = 0.0;   // linear-domain weights (lmc_sampler.hpp)
    double c_tot = 0.0;   // offset the accepted totals {w_start, wn, an} are expressed in (see nuts_transition2)
    int depth = 0, n_leap = 0;
    bool diverging = false, turning = false, exhausted = true;
    LevelScalars lsc = {0.0, 0.0, 0.0, 0.0};

    for (int dd = 0; dd < max_depth; ++dd) {
        const bool right = pol.uniform() < 0.5;   // nuts.py:213
        const double eps = right ? step_size : -step_size;
        const int side = right ? 1 : 0;
        typename P::End c;   // the end that is being extended
        pol.end_load(side, c);
        double tw = 0.0, ta = 0.0, tpe = 0.0, tplogp = 0.0;
        const int n_leaves = 1 << depth;
        for (int i = 0; i < n_leaves; ++i) {
            double energy, logp;
            pol.leapfrog(eps, c, energy, logp);
            ++n_leap;
            double de = first_f64(energy - e0);
            if (isnan(de)) de = __builtin_inf();
            if (fabs(de) > fabs(max_de)) max_de = de;
            if (!(fabs(de) < emax)) { diverging = true; break; }   // nuts.py:358,370-375
            const double x = -de;
            if (x - coff > 600.0) {
                const double f = exp_uniform(coff - x);
                lsc.w *= f; lsc.a *= f;
                coff = x;
            }
            tw = exp_uniform_fast(x - coff);
            const double sat = (coff == 0.0) ? fmin(1.0, tw) : ((x >= 0.0) ? 1.0 : exp_uniform(x));
            ta = tw * sat;
            // the leaf as a one-state node: both ends and the momentum sum are the new state's
            pol.template node_st<kNodeLp>(c.p); pol.template node_st<kNodeLv>(c.v);
            pol.template node_st<kNodePs>(c.p); pol.template node_st<kNodeQ>(c.q);
            tpe = energy; tplogp = logp;
            int j = 0;
            while ((i >> j) & 1) {   // merge stack[j] (a, earlier) with the node under construction (b); nuts.py:377-417
                double aw, aa, ape, aplogp;
                lsc.get(j, aw, aa, ape, aplogp);
                double ps[NS], alv[NS];
                {
                    double aps[NS], tps[NS];
                    pol.level_ld(j, kNodePs, aps); pol.template node_ld<kNodePs>(tps);
#pragma unroll
                    for (int s = 0; s < NS; ++s) ps[s] = aps[s] + tps[s];
                }
                pol.level_ld(j, kNodeLv, alv);
                bool turn;
                if (j > 0) {   // nuts.py:389-396
                    double dots[6];
                    dots[0] = pdot<NS>(ps, alv); dots[1] = pdot<NS>(ps, c.v);
                    {
                        double aps[NS], tlp[NS], p1[NS], tlv[NS];
                        pol.level_ld(j, kNodePs, aps); pol.template node_ld<kNodeLp>(tlp); pol.template node_ld<kNodeLv>(tlv);
#pragma unroll
                        for (int s = 0; s < NS; ++s) p1[s] = aps[s] + tlp[s];
                        dots[2] = pdot<NS>(p1, alv); dots[3] = pdot<NS>(p1, tlv);
                    }
                    {
                        double arp[NS], tps[NS], p2[NS], arv[NS];
                        pol.level_ld(j, kNodeRp, arp); pol.template node_ld<kNodePs>(tps); pol.level_ld(j, kNodeRv, arv);
#pragma unroll
                        for (int s = 0; s < NS; ++s) p2[s] = arp[s] + tps[s];
                        dots[4] = pdot<NS>(p2, arv); dots[5] = pdot<NS>(p2, c.v);
                    }
                    turn = pol.any_nonpositive6(dots);
                } else {
                    turn = pol.any_nonpositive2(pdot<NS>(ps, alv), pdot<NS>(ps, c.v));
                }
                const double wsum = aw + tw;
                const double asum = aa + ta;
                const bool take_b = uniform_true(pol.uniform() * wsum < tw);   // nuts.py:404 (drawn even if turning)
                {   // the merged node: a's left end, the summed momentum; its proposal is a's unless b's was taken
                    double alp[NS];
                    pol.level_ld(j, kNodeLp, alp);
                    pol.template node_st<kNodeLp>(alp);
                }
                pol.template node_st<kNodeLv>(alv); pol.template node_st<kNodePs>(ps);
                if (!take_b) {
                    double aq[NS];
                    pol.level_ld(j, kNodeQ, aq);
                    pol.template node_st<kNodeQ>(aq);
                    tpe = ape; tplogp = aplogp;
                }
                tw = wsum; ta = asum;
                ++j;
                if (turn) { turning = true; break; }
            }
            if (turning) break;
            if (i + 1 < n_leaves) {   // park the node at level j
                double t[NS];
                pol.template node_ld<kNodeLp>(t); pol.level_st(j, kNodeLp, t);
                pol.template node_ld<kNodeLv>(t); pol.level_st(j, kNodeLv, t);
                pol.level_st(j, kNodeRp, c.p); pol.level_st(j, kNodeRv, c.v);
                pol.template node_ld<kNodePs>(t); pol.level_st(j, kNodePs, t);
                pol.template node_ld<kNodeQ>(t); pol.level_st(j, kNodeQ, t);
                lsc.put(j, tw, ta, tpe, tplogp);
            }
        }
        ++depth;   // nuts.py:315
        if (diverging || turning) { exhausted = false; break; }

        // ---- accepted subtree: merge into the trajectory (nuts.py:321-340)
        if (c_tot != coff) {   // the offset moved inside this subtree: bring the accepted totals to it (rare)
            const double f = exp_uniform(c_tot - coff);
            wn = first_f64(wn * f); an = first_f64(an * f); w_start = first_f64(w_start * f);
            c_tot = coff;
        }
        if (uniform_true(pol.uniform() * (w_start + wn) < tw)) {   // biased progressive
            pol.proposal_from_node(); prop_e = tpe; prop_logp = tplogp;
        }
        wn = first_f64(wn + tw);
        an = first_f64(an + ta);
        double psum[NS], tps[NS];
        pol.psum_ld(psum); pol.template node_ld<kNodePs>(tps);
#pragma unroll
        for (int s = 0; s < NS; ++s) {   // in place; float32 storage when the start momentum is float32 (nuts.py:329)
            const double t = psum[s] + tps[s];
            psum[s] = momentum_f32 ? static_cast<double>(static_cast<float>(t)) : t;
        }
        pol.psum_st(psum);
        double dots[6];
        {
            double oLv[NS], oRv[NS], oP[NS], tlp[NS], tlv[NS], p1[NS], p2[NS];
            pol.end_velocity(0, oLv); pol.end_velocity(1, oRv);   // velocities of both old ends ...
            pol.end_momentum(side, oP);                            // ... and the momentum of the end that is being replaced
            pol.template node_ld<kNodeLp>(tlp); pol.template node_ld<kNodeLv>(tlv);
            if (right) {
#pragma unroll
                for (int s = 0; s < NS; ++s) { p1[s] = psum[s] + tlp[s]; p2[s] = oP[s] + tps[s]; }
                dots[0] = pdot<NS>(psum, oLv); dots[1] = pdot<NS>(psum, c.v);
                dots[2] = pdot<NS>(p1, oLv);   dots[3] = pdot<NS>(p1, tlv);
                dots[4] = pdot<NS>(p2, oRv);   dots[5] = pdot<NS>(p2, c.v);
            } else {
#pragma unroll
                for (int s = 0; s < NS; ++s) { p1[s] = tps[s] + oP[s]; p2[s] = tlp[s] + psum[s]; }
                dots[0] = pdot<NS>(psum, c.v); dots[1] = pdot<NS>(psum, oRv);
                dots[2] = pdot<NS>(p1, c.v);   dots[3] = pdot<NS>(p1, oLv);
                dots[4] = pdot<NS>(p2, tlv);   dots[5] = pdot<NS>(p2, oRv);
            }
        }
        pol.end_store(side, c);
        if (pol.any_nonpositive6(dots)) { turning = true; exhausted = false; break; }
    }

    out.accept = (wn > 0.0) ? first_f64(an / wn) : 0.0;   // nuts.py:421-425
    out.energy = prop_e;
    out.energy_error = first_f64(prop_e - e0);
    out.max_energy_error = max_de;
    out.model_logp = prop_logp;
    out.depth = depth;
    out.n_leapfrog = n_leap;
    out.diverging = diverging;
    out.exhausted = exhausted;
    out.accepted = 0;
}

}  // namespace lmc
