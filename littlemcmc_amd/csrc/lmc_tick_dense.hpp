// The tick kernel (lmc_tick.hpp) with a dense mass matrix: densities evaluated by the caller (targets.TorchTarget)
// sampled with QuadPotentialFull / FullInv / FullAdapt. Same state machine, cut at the density evaluation; the
// differences are the ones between lmc_sampler.hpp and lmc_dense.hpp: velocities are matrix sweeps and therefore
// stored with the trajectory ends and tree nodes, one sweep per leapfrog forms v = C p and w = C g, the momentum is
// a triangular solve (or L n), and FullAdapt's update of a chain that finished a tuning iteration in this tick runs
// in dense_adapt_kernel, launched masked by the host between two ticks. GENERATED from lmc_tick.hpp by
// tools/gen_tick_dense.py so that the two state machines stay statement-parallel; do not edit by hand.
#pragma once
#include "lmc_dense.hpp"
#include "lmc_tick_launch.hpp"

namespace lmc {

template <int NS, class MatT>
__global__ __launch_bounds__(64, dense_waves_per_simd(NS)) void tick_dense_kernel(ChainArrays A, DenseArrays D, TickArrays K,
                                                                                  SamplerParams P, const double* logp_in,
                                                                                  const double* grad_in, int* adapt_mask) {
    extern __shared__ __attribute__((aligned(16))) double lds[];   // 2 * dpad doubles: normals + staging / sdot staging
    const int c = blockIdx.x;
    const int lane = lane_id();
    const int d = A.d, dpad = A.dpad;
    int phase = first_i32(K.phase[c]);
    if (phase == kTickDone) return;
    const long long row = static_cast<long long>(c) * dpad;
    Team<1> tm{nullptr, 0};
    glb_double* scr = (glb_double*)(A.scratch + static_cast<long long>(c) * A.scratch_stride);
    auto slot = [&](int k) { return scr + k * dpad; };
    auto level = [&](int j, int k) { return scr + (kTickDenseFixedSlots + 6 * j + k) * dpad; };
    const MatT* M = static_cast<const MatT*>(D.covT) + static_cast<long long>(c) * D.mat_stride;
    DenseMat<MatT> mm{M, nullptr, 0, d, dpad};
    lds_double* xop = (lds_double*)lds;

    // ---- persistent chain state
    long long git = K.git[c];
    const bool tune = git < K.n_tune;
    RngState rng;
    rng.mt = A.mt + static_cast<long long>(c) * kMtN;   // in place in HBM / L2: a tick touches a few words
    rng.pos = first_i32(A.rng_pos[c]);
    rng.has_gauss = first_i32(A.rng_has_gauss[c]);
    rng.gauss = first_f64(A.rng_gauss[c]);
    UniformWindow win;
    window_reset(win);
    DualAverage da;
    dual_average_load(A, c, da);
    int iter_count = first_i32(A.iter_count[c]);
    int* ti = K.ti + c * kNumTickInt;
    double* td = K.td + c * kNumTickDbl;
    int depth = first_i32(ti[kTiDepth]), leaf = first_i32(ti[kTiLeaf]), n_leap = first_i32(ti[kTiNLeap]);
    bool right = first_i32(ti[kTiRight]) != 0;
    bool l_start = first_i32(ti[kTiLStart]) != 0, r_start = first_i32(ti[kTiRStart]) != 0;
    int max_depth = first_i32(ti[kTiMaxDepth]), n_steps = first_i32(ti[kTiSteps]);
    double eps = first_f64(td[kTdEps]), step_size = first_f64(td[kTdStep]), e0 = first_f64(td[kTdE0]);
    double logp0 = first_f64(td[kTdLogp0]), prop_e = first_f64(td[kTdPropE]), prop_logp = first_f64(td[kTdPropLogp]);
    double coff = first_f64(td[kTdCoff]), w_start = first_f64(td[kTdWStart]), wn = first_f64(td[kTdWn]);
    double an = first_f64(td[kTdAn]), max_de = first_f64(td[kTdMaxDe]), plen = first_f64(td[kTdPlen]);
    double c_tot = first_f64(td[kTdCtot]);   // offset the accepted totals {w_start, wn, an} are expressed in
    LevelScalars lsc = {0.0, 0.0, 0.0, 0.0};
    double* lvl = K.lvl + static_cast<long long>(c) * 4 * kTickLevels;
    if (lane < kTickLevels) {
        lsc.w = lvl[lane]; lsc.a = lvl[kTickLevels + lane]; lsc.pe = lvl[2 * kTickLevels + lane];
        lsc.plogp = lvl[3 * kTickLevels + lane];
    }
    const bool momentum_f32 = P.momentum_f32 != 0;
    int status = 0;

    // what this tick decides
    bool begin_doubling = false, subtree_done = false, end_transition = false, need_leap = false;
    bool diverging = false, turning = false, exhausted = false, accepted = false;
    double cq[NS], cp[NS], cg[NS], cv[NS], cw[NS];   // the state the next leapfrog starts from: q, p, g, v = C p, w = C g
    double q[NS];                           // the chain's position (start of the iteration / its result)
    double tlp[NS], tlv[NS], trp[NS], trv[NS], tps[NS], tq[NS];
    double tw = 0.0, ta = 0.0, tpe = 0.0, tplogp = 0.0;
    const double logp_new = first_f64(logp_in[c]);

    if (phase == kTickStart) {
        // ---- the iteration begins (base_hmc.py:140-153): momentum draw, start state from the delivered density
        vload<NS>(A.q + row, q);
        double g0[NS];
        load_rows<NS>(grad_in + static_cast<long long>(c) * d, d, lane, g0);
        rng_normals(rng, d, lds, lds + dpad);
        double p0[NS];
        if (D.kind == kDenseFullInv)
            dense_momentum_inv<NS>(static_cast<const double*>(D.fac), d, dpad, xop, p0);
        else
            dense_momentum_full<NS>(static_cast<const float*>(D.fac) + static_cast<long long>(c) * D.fac_stride, d, dpad, xop, p0);
        logp0 = logp_new;
        double v0[NS], w0[NS], v0s[NS];
        e0 = dense_start_state<NS, MatT>(tm, mm, lds, momentum_f32, P.sdot_mode, p0, g0, logp0, v0, w0, v0s);
        if (!isfinite(e0)) {   // base_hmc.py:145-148
            if (lane == 0) { A.status[c] |= kStatusBadInitialEnergy; K.phase[c] = kTickDone; }
            return;
        }
        const bool adapt_step = tune && P.adapt_step_size;
        step_size = jitter_step_size(tm, rng, A, P, c, adapt_step ? da.step_now : da.step_bar_now);
        n_leap = 0;
        if (P.kind == 0) {
            max_depth = (tune && iter_count < 200) ? P.early_max_treedepth : P.max_treedepth;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                vstore_as<NS>(slot(5 * r + 0), q); vstore_as<NS>(slot(5 * r + 1), p0); vstore_as<NS>(slot(5 * r + 2), g0);
                vstore_as<NS>(slot(5 * r + 3), v0); vstore_as<NS>(slot(5 * r + 4), w0);
            }
            vstore_as<NS>(slot(kSlotPsum), p0); vstore_as<NS>(slot(kSlotProp), q); vstore_as<NS>(slot(kSlotV0s), v0s);
            l_start = true; r_start = true;   // the end still is the start state: its stored velocity is v0s
            prop_e = e0; prop_logp = logp0;
            coff = 0.0; c_tot = 0.0; w_start = 1.0; wn = 0.0; an = 0.0; max_de = 0.0;
            depth = 0;
            lsc = {0.0, 0.0, 0.0, 0.0};
            begin_doubling = true;
        } else {   // hmc.py:143-149
            plen = first_f64(window_next(rng, win) * P.path_length);
            n_steps = static_cast<int>(plen / step_size);
            n_steps = n_steps < 1 ? 1 : n_steps;
            n_steps = n_steps > P.max_steps ? P.max_steps : n_steps;
            eps = step_size;
            vcopy(cq, q); vcopy(cp, p0); vcopy(cg, g0); vcopy(cv, v0); vcopy(cw, w0);
            need_leap = true;
        }
    } else {
        // ---- second half of the leapfrog (integration.py:115-121) with the delivered gradient
        double half[NS];
        load_rows<NS>(K.q_eval + static_cast<long long>(c) * d, d, lane, cq);
        load_rows<NS>(grad_in + static_cast<long long>(c) * d, d, lane, cg);
        vload_as<NS>(slot(kSlotHalf), half);
        const double dt = 0.5 * eps;
#pragma unroll
        for (int s = 0; s < NS; ++s) cp[s] = half[s] + dt * cg[s];
        velocity2<NS, MatT>(mm, xop, cp, cg, cv, cw);   // the one matrix sweep of this leapfrog: v = C p, w = C g
        const double energy = first_f64(0.5 * tm.sum(pdot<NS>(cp, cv)) - logp_new);
        ++n_leap;
        if (P.kind == 0) {
            // ---- leaf (nuts.py:344-375) and the merges it closes (nuts.py:377-417)
            double de = first_f64(energy - e0);
            if (isnan(de)) de = __builtin_inf();
            if (fabs(de) > fabs(max_de)) max_de = de;
            if (!(fabs(de) < P.emax)) {
                diverging = true;
            } else {
                const double x = -de;
                if (x - coff > 600.0) {
                    const double f = exp_uniform(coff - x);
                    lsc.w *= f; lsc.a *= f;
                    coff = x;
                }
                tw = exp_uniform_fast(x - coff);
                const double sat = (coff == 0.0) ? fmin(1.0, tw) : ((x >= 0.0) ? 1.0 : exp_uniform(x));
                ta = tw * sat;
                vcopy(tlp, cp); vcopy(trp, cp); vcopy(tps, cp); vcopy(tlv, cv); vcopy(trv, cv); vcopy(tq, cq);
                tpe = energy; tplogp = logp_new;
                int j = 0;
                while ((leaf >> j) & 1) {
                    double alp[NS], alv[NS], arp[NS], arv[NS], aps[NS], aq[NS];
                    double aw, aa, ape, aplogp;
                    vload_as<NS>(level(j, 0), alp); vload_as<NS>(level(j, 1), alv); vload_as<NS>(level(j, 2), arp);
                    vload_as<NS>(level(j, 3), arv); vload_as<NS>(level(j, 4), aps); vload_as<NS>(level(j, 5), aq);
                    lsc.get(j, aw, aa, ape, aplogp);
                    double ps[NS];
#pragma unroll
                    for (int s = 0; s < NS; ++s) ps[s] = aps[s] + tps[s];
                    bool turn;
                    if (j > 0) {
                        double p1[NS], p2[NS];
#pragma unroll
                        for (int s = 0; s < NS; ++s) { p1[s] = aps[s] + tlp[s]; p2[s] = arp[s] + tps[s]; }
                        double dots[6] = {pdot<NS>(ps, alv), pdot<NS>(ps, trv), pdot<NS>(p1, alv),
                                          pdot<NS>(p1, tlv), pdot<NS>(p2, arv), pdot<NS>(p2, trv)};
                        turn = tm.any_nonpositive6(dots);
                    } else {
                        turn = tm.any_nonpositive2(pdot<NS>(ps, alv), pdot<NS>(ps, trv));
                    }
                    const double wsum = aw + tw;
                    const double asum = aa + ta;
                    const bool take_b = uniform_true(window_next(rng, win) * wsum < tw);
                    vcopy(tlp, alp); vcopy(tlv, alv); vcopy(tps, ps);
                    if (!take_b) { vcopy(tq, aq); tpe = ape; tplogp = aplogp; }
                    tw = wsum; ta = asum;
                    ++j;
                    if (turn) { turning = true; break; }
                }
                if (!turning) {
                    if (leaf + 1 < (1 << depth)) {   // park the node, continue the subtree from (cq, cp, cg)
                        vstore_as<NS>(level(j, 0), tlp); vstore_as<NS>(level(j, 1), tlv); vstore_as<NS>(level(j, 2), trp);
                        vstore_as<NS>(level(j, 3), trv); vstore_as<NS>(level(j, 4), tps); vstore_as<NS>(level(j, 5), tq);
                        lsc.put(j, tw, ta, tpe, tplogp);
                        ++leaf;
                        need_leap = true;
                    } else {
                        subtree_done = true;
                    }
                }
            }
            if (diverging || turning) { ++depth; end_transition = true; }
            if (subtree_done) {
                // ---- accepted subtree: merge into the trajectory (nuts.py:315-340)
                ++depth;
                double psum[NS], propq[NS];
                vload_as<NS>(slot(kSlotPsum), psum); vload_as<NS>(slot(kSlotProp), propq);
                if (c_tot != coff) {   // the offset moved inside this subtree: bring the accepted totals to it (rare)
                    const double f = exp_uniform(c_tot - coff);
                    wn = first_f64(wn * f); an = first_f64(an * f); w_start = first_f64(w_start * f);
                    c_tot = coff;
                }
                if (uniform_true(window_next(rng, win) * (w_start + wn) < tw)) {
                    vcopy(propq, tq); prop_e = tpe; prop_logp = tplogp;
                    vstore_as<NS>(slot(kSlotProp), propq);
                }
                wn = first_f64(wn + tw);
                an = first_f64(an + ta);
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const double t = psum[s] + tps[s];
                    psum[s] = momentum_f32 ? static_cast<double>(static_cast<float>(t)) : t;
                }
                vstore_as<NS>(slot(kSlotPsum), psum);
                double Lp[NS], Rp[NS], oLv[NS], oRv[NS], vtl[NS], vtr[NS];
                vload_as<NS>(slot(1), Lp); vload_as<NS>(slot(6), Rp);
                vload_as<NS>(slot(l_start ? kSlotV0s : 3), oLv);
                vload_as<NS>(slot(r_start ? kSlotV0s : 8), oRv);
                vcopy(vtl, tlv); vcopy(vtr, trv);
                double dots[6], p1[NS], p2[NS];
                const int side = right ? 1 : 0;
                if (right) {
#pragma unroll
                    for (int s = 0; s < NS; ++s) { p1[s] = psum[s] + tlp[s]; p2[s] = Rp[s] + tps[s]; }
                    dots[0] = pdot<NS>(psum, oLv); dots[1] = pdot<NS>(psum, vtr);
                    dots[2] = pdot<NS>(p1, oLv);   dots[3] = pdot<NS>(p1, vtl);
                    dots[4] = pdot<NS>(p2, oRv);   dots[5] = pdot<NS>(p2, vtr);
                    r_start = false;
                } else {
#pragma unroll
                    for (int s = 0; s < NS; ++s) { p1[s] = tps[s] + Lp[s]; p2[s] = tlp[s] + psum[s]; }
                    dots[0] = pdot<NS>(psum, vtr); dots[1] = pdot<NS>(psum, oRv);
                    dots[2] = pdot<NS>(p1, vtr);   dots[3] = pdot<NS>(p1, oLv);
                    dots[4] = pdot<NS>(p2, vtl);   dots[5] = pdot<NS>(p2, oRv);
                    l_start = false;
                }
                vstore_as<NS>(slot(5 * side + 0), cq); vstore_as<NS>(slot(5 * side + 1), cp); vstore_as<NS>(slot(5 * side + 2), cg);
                vstore_as<NS>(slot(5 * side + 3), cv); vstore_as<NS>(slot(5 * side + 4), cw);
                if (tm.any_nonpositive6(dots)) { turning = true; end_transition = true; }
                else if (depth >= max_depth) { exhausted = true; end_transition = true; }
                else begin_doubling = true;
            }
        } else {
            // ---- HMC: next step or the Metropolis test (hmc.py:150-176)
            if (n_leap < n_steps) {
                need_leap = true;
            } else {
                diverging = !isfinite(energy);
                double de = first_f64(e0 - energy);
                if (isnan(de)) de = -__builtin_inf();
                if (fabs(de) > P.emax) diverging = true;
                const double accept = first_f64(fmin(1.0, exp_uniform(de)));
                if (!diverging) {
                    const double u = window_next(rng, win);
                    if (!(u >= accept)) accepted = true;
                }
                an = accept; prop_e = energy; prop_logp = logp_new; max_de = de;
                end_transition = true;
            }
        }
    }

    if (begin_doubling) {   // nuts.py:211-216: direction, then extend from that end
        right = window_next(rng, win) < 0.5;
        eps = right ? step_size : -step_size;
        const int side = right ? 1 : 0;
        vload_as<NS>(slot(5 * side + 0), cq); vload_as<NS>(slot(5 * side + 1), cp); vload_as<NS>(slot(5 * side + 2), cg);
        vload_as<NS>(slot(5 * side + 3), cv); vload_as<NS>(slot(5 * side + 4), cw);
        leaf = 0;
        need_leap = true;
    }

    if (need_leap) {
        // ---- first half of the next leapfrog (integration.py:107-112): the point whose density is wanted
        const double dt = 0.5 * eps;
        double half[NS], qn[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            half[s] = cp[s] + dt * cg[s];
            const double v = cv[s] + dt * cw[s];   // C (p + dt g)
            qn[s] = cq[s] + eps * v;
        }
        vstore_as<NS>(slot(kSlotHalf), half);
        store_rows<NS>(K.q_eval + static_cast<long long>(c) * d, d, lane, qn);
        phase = kTickLeap;
    }

    if (end_transition) {
        // ---- statistics, adaptation, outputs (base_hmc.py:155-190), then the next iteration asks for its start density
        TransitionOut out;
        if (P.kind == 0) {
            vload_as<NS>(slot(kSlotProp), q);
            out.accept = (wn > 0.0) ? first_f64(an / wn) : 0.0;   // nuts.py:421-425
            out.energy = prop_e;
            out.energy_error = first_f64(prop_e - e0);
            out.max_energy_error = max_de;
            out.model_logp = prop_logp;
            out.depth = depth;
            out.accepted = 0;
        } else {
            vload<NS>(A.q + row, q);
            if (accepted) vcopy(q, cq);
            out.accept = an;
            out.energy = prop_e;
            out.energy_error = max_de;
            out.max_energy_error = plen;
            out.model_logp = prop_logp;
            out.depth = n_steps;
            out.accepted = accepted;
        }
        out.n_leapfrog = n_leap;
        out.diverging = diverging;
        out.exhausted = exhausted;
        long long ct_maxdepth = (P.kind == 0 && exhausted && !tune) ? 1 : 0;
        const bool adapt_step = tune && P.adapt_step_size;
        if (adapt_step) dual_average_update(A, P, out.accept, da);
        // FullAdapt.update for this chain runs in dense_adapt_kernel right after this tick (the host launches it masked)
        if (tune && D.kind == kDenseFullAdapt && lane == 0) adapt_mask[c] = 1;
        ++iter_count;
        if (A.mom_mean != nullptr && !tune) moments_update<NS>(A, tm, c, row, q);
        write_outputs<NS>(A, c, lane, git, q, out, da.step_now, da.step_bar_now, tune);
        vstore<NS>(A.q + row, q);
        if (lane == 0) {
            A.counters[c * kNumCounters + kCtMaxTreedepth] += ct_maxdepth;
            A.counters[c * kNumCounters + kCtDivsSample] += (diverging && !tune) ? 1 : 0;
            A.counters[c * kNumCounters + kCtSamplesAfterTune] += tune ? 0 : 1;
            A.counters[c * kNumCounters + kCtLeapfrogs] += n_leap;
        }
        ++git;
        if (git >= K.iter_end) {
            phase = kTickDone;
        } else {
            phase = kTickStart;
            store_rows<NS>(K.q_eval + static_cast<long long>(c) * d, d, lane, q);
        }
    }

    // ---- store
    if (lane < kTickLevels) {
        lvl[lane] = lsc.w; lvl[kTickLevels + lane] = lsc.a; lvl[2 * kTickLevels + lane] = lsc.pe;
        lvl[3 * kTickLevels + lane] = lsc.plogp;
    }
    if (lane == 0) {
        K.phase[c] = phase;
        K.git[c] = git;
        ti[kTiDepth] = depth; ti[kTiLeaf] = leaf; ti[kTiRight] = right ? 1 : 0; ti[kTiNLeap] = n_leap;
        ti[kTiLStart] = l_start ? 1 : 0; ti[kTiRStart] = r_start ? 1 : 0; ti[kTiMaxDepth] = max_depth; ti[kTiSteps] = n_steps;
        td[kTdEps] = eps; td[kTdStep] = step_size; td[kTdE0] = e0; td[kTdLogp0] = logp0; td[kTdPropE] = prop_e;
        td[kTdPropLogp] = prop_logp; td[kTdCoff] = coff; td[kTdWStart] = w_start; td[kTdWn] = wn; td[kTdAn] = an;
        td[kTdMaxDe] = max_de; td[kTdPlen] = plen; td[kTdCtot] = c_tot;
        A.rng_pos[c] = rng.pos;
        A.rng_has_gauss[c] = rng.has_gauss;
        A.rng_gauss[c] = rng.gauss;
        A.da[c * 4 + 0] = da.log_step;
        A.da[c * 4 + 1] = da.log_bar;
        A.da[c * 4 + 2] = da.hbar;
        A.da_count[c] = da.count;
        A.iter_count[c] = iter_count;
        A.status[c] |= status;
    }
}

}  // namespace lmc
