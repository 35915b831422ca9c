// GENERATED from lmc_tick.hpp by tools/gen_tick_wide.py -- do not edit; edit lmc_tick.hpp / the generator and re-run it.
// The tick state machine (externally evaluated log-densities: a Python callable, a batched torch callable) for the shapes
// of the general kernels: one chain = a workgroup of 16 wavefronts (lmc_wide.hpp), model_ndim up to 16 384, diagonal mass
// matrices. Statement for statement the one-wavefront tick kernel; what differs is who "lane" is (the thread's index in its
// chain), the team's reductions and barriers, the normals drawn 1024 at a time by wave 0, and the uniform stream shared by
// the team.
#pragma once
#include "lmc_tick_launch.hpp"
#include "lmc_wide.hpp"

namespace lmc {

template <int NS>
__global__ __launch_bounds__(kWideThreads, 1) void tick_wide_kernel(ChainArrays A, TickArrays K, SamplerParams P, const double* logp_in,
                                                                     const double* grad_in) {
    extern __shared__ __attribute__((aligned(16))) double lds[];   // wide_stage_doubles(dpad): normals chunk + staging / sdot staging; team exchange; broadcast words
    const int c = blockIdx.x;
    const int lane = static_cast<int>(threadIdx.x);   // the thread's index in its chain (the name is the one-wavefront kernel's)
    const int d = A.d, dpad = A.dpad;
    int phase = first_i32(K.phase[c]);
    if (phase == kTickDone) return;
    const long long row = static_cast<long long>(c) * dpad;
    WideTeam tm;
    tm.xbuf = lds + wide_stage_doubles(dpad);
    tm.parity = 0;
    double* bcast = tm.xbuf + 2 * kWideWaves * kTeamSlots;
    glb_double* scr = (glb_double*)(A.scratch + static_cast<long long>(c) * A.scratch_stride);
    auto slot = [&](int k) { return scr + k * dpad; };
    auto level = [&](int j, int k) { return scr + (9 + 4 * j + k) * dpad; };

    // ---- persistent chain state
    long long git = K.git[c];
    const bool tune = git < K.n_tune;
    float var[NS], inv_std[NS];
    double vard[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        var[s] = A.var[row + lane * NS + s];
        inv_std[s] = A.inv_std[row + lane * NS + s];
        vard[s] = static_cast<double>(var[s]);
    }
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
    if (lane_id() < kTickLevels) {   // every wave holds the level scalars on its own lanes
        lsc.w = lvl[lane_id()]; lsc.a = lvl[kTickLevels + lane_id()]; lsc.pe = lvl[2 * kTickLevels + lane_id()];
        lsc.plogp = lvl[3 * kTickLevels + lane_id()];
    }
    const bool momentum_f32 = P.momentum_f32 != 0;
    int status = 0;

    // what this tick decides
    bool begin_doubling = false, subtree_done = false, end_transition = false, need_leap = false;
    bool diverging = false, turning = false, exhausted = false, accepted = false;
    double cq[NS], cp[NS], cg[NS];          // the state the next leapfrog starts from
    double q[NS];                           // the chain's position (start of the iteration / its result)
    double tlp[NS], trp[NS], tps[NS], tq[NS];
    double tw = 0.0, ta = 0.0, tpe = 0.0, tplogp = 0.0;
    const double logp_new = first_f64(logp_in[c]);

    if (phase == kTickStart) {
        // ---- the iteration begins (base_hmc.py:140-153): momentum draw, start state from the delivered density
        vload<NS>(A.q + row, q);
        double g0[NS];
        load_rows<NS>(grad_in + static_cast<long long>(c) * d, d, lane, g0);
        double zz[NS];
        wide_normals_regs<NS>(tm, rng, d, lds, bcast, zz);   // wave 0 draws (numpy's stream is sequential), 1024 at a time
        double p0[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const double z = zz[s];
            p0[s] = momentum_f32 ? static_cast<double>(inv_std[s] * static_cast<float>(z)) : z * static_cast<double>(inv_std[s]);
        }
        tm.sync();
        logp0 = logp_new;
        if (momentum_f32) {
            const float kin = start_kinetic_f32<NS>(tm, p0, var, d, P.sdot_mode, reinterpret_cast<float*>(lds), dpad);
            e0 = first_f64(static_cast<double>(kin) - logp0);
        } else {
            e0 = first_f64(0.5 * tm.sum(pdot_v<NS>(p0, vard, p0)) - logp0);
        }
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
                vstore_as<NS>(slot(3 * r + 0), q); vstore_as<NS>(slot(3 * r + 1), p0); vstore_as<NS>(slot(3 * r + 2), g0);
            }
            vstore_as<NS>(slot(6), p0); vstore_as<NS>(slot(7), q);
            l_start = momentum_f32; r_start = momentum_f32;
            prop_e = e0; prop_logp = logp0;
            coff = 0.0; c_tot = 0.0; w_start = 1.0; wn = 0.0; an = 0.0; max_de = 0.0;
            depth = 0;
            lsc = {0.0, 0.0, 0.0, 0.0};
            begin_doubling = true;
        } else {   // hmc.py:143-149
            plen = first_f64(team_uniform(tm, rng, win) * P.path_length);
            n_steps = static_cast<int>(plen / step_size);
            n_steps = n_steps < 1 ? 1 : n_steps;
            n_steps = n_steps > P.max_steps ? P.max_steps : n_steps;
            eps = step_size;
            vcopy(cq, q); vcopy(cp, p0); vcopy(cg, g0);
            need_leap = true;
        }
    } else {
        // ---- second half of the leapfrog (integration.py:115-121) with the delivered gradient
        double half[NS];
        load_rows<NS>(K.q_eval + static_cast<long long>(c) * d, d, lane, cq);
        load_rows<NS>(grad_in + static_cast<long long>(c) * d, d, lane, cg);
        vload_as<NS>(slot(8), half);
        const double dt = 0.5 * eps;
        double kin = 0.0;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            cp[s] = half[s] + dt * cg[s];
            kin = __builtin_fma(cp[s], vard[s] * cp[s], kin);
        }
        const double energy = first_f64(0.5 * tm.sum(kin) - logp_new);
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
                vcopy(tlp, cp); vcopy(trp, cp); vcopy(tps, cp); vcopy(tq, cq);
                tpe = energy; tplogp = logp_new;
                int j = 0;
                while ((leaf >> j) & 1) {
                    double alp[NS], arp[NS], aps[NS], aq[NS];
                    double aw, aa, ape, aplogp;
                    vload_as<NS>(level(j, 0), alp); vload_as<NS>(level(j, 1), arp);
                    vload_as<NS>(level(j, 2), aps); vload_as<NS>(level(j, 3), aq);
                    lsc.get(j, aw, aa, ape, aplogp);
                    double ps[NS];
#pragma unroll
                    for (int s = 0; s < NS; ++s) ps[s] = aps[s] + tps[s];
                    bool turn;
                    if (j > 0) {
                        double p1[NS], p2[NS];
#pragma unroll
                        for (int s = 0; s < NS; ++s) { p1[s] = aps[s] + tlp[s]; p2[s] = arp[s] + tps[s]; }
                        double dots[6] = {pdot_v<NS>(ps, vard, alp), pdot_v<NS>(ps, vard, trp), pdot_v<NS>(p1, vard, alp),
                                          pdot_v<NS>(p1, vard, tlp), pdot_v<NS>(p2, vard, arp), pdot_v<NS>(p2, vard, trp)};
                        turn = tm.any_nonpositive6(dots);
                    } else {
                        turn = tm.any_nonpositive2(pdot_v<NS>(ps, vard, alp), pdot_v<NS>(ps, vard, trp));
                    }
                    const double wsum = aw + tw;
                    const double asum = aa + ta;
                    const bool take_b = uniform_true(team_uniform(tm, rng, win) * wsum < tw);
                    vcopy(tlp, alp); vcopy(tps, ps);
                    if (!take_b) { vcopy(tq, aq); tpe = ape; tplogp = aplogp; }
                    tw = wsum; ta = asum;
                    ++j;
                    if (turn) { turning = true; break; }
                }
                if (!turning) {
                    if (leaf + 1 < (1 << depth)) {   // park the node, continue the subtree from (cq, cp, cg)
                        vstore_as<NS>(level(j, 0), tlp); vstore_as<NS>(level(j, 1), trp);
                        vstore_as<NS>(level(j, 2), tps); vstore_as<NS>(level(j, 3), tq);
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
                vload_as<NS>(slot(6), psum); vload_as<NS>(slot(7), propq);
                if (c_tot != coff) {   // the offset moved inside this subtree: bring the accepted totals to it (rare)
                    const double f = exp_uniform(c_tot - coff);
                    wn = first_f64(wn * f); an = first_f64(an * f); w_start = first_f64(w_start * f);
                    c_tot = coff;
                }
                if (uniform_true(team_uniform(tm, rng, win) * (w_start + wn) < tw)) {
                    vcopy(propq, tq); prop_e = tpe; prop_logp = tplogp;
                    vstore_as<NS>(slot(7), propq);
                }
                wn = first_f64(wn + tw);
                an = first_f64(an + ta);
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const double t = psum[s] + tps[s];
                    psum[s] = momentum_f32 ? static_cast<double>(static_cast<float>(t)) : t;
                }
                vstore_as<NS>(slot(6), psum);
                double Lp[NS], Rp[NS], oLv[NS], oRv[NS], vtl[NS], vtr[NS];
                vload_as<NS>(slot(1), Lp); vload_as<NS>(slot(4), Rp);
                end_velocity<NS>(oLv, vard, Lp, l_start);
                end_velocity<NS>(oRv, vard, Rp, r_start);
#pragma unroll
                for (int s = 0; s < NS; ++s) { vtl[s] = vard[s] * tlp[s]; vtr[s] = vard[s] * trp[s]; }
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
                vstore_as<NS>(slot(3 * side + 0), cq); vstore_as<NS>(slot(3 * side + 1), cp); vstore_as<NS>(slot(3 * side + 2), cg);
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
                    const double u = team_uniform(tm, rng, win);
                    if (!(u >= accept)) accepted = true;
                }
                an = accept; prop_e = energy; prop_logp = logp_new; max_de = de;
                end_transition = true;
            }
        }
    }

    if (begin_doubling) {   // nuts.py:211-216: direction, then extend from that end
        right = team_uniform(tm, rng, win) < 0.5;
        eps = right ? step_size : -step_size;
        const int side = right ? 1 : 0;
        vload_as<NS>(slot(3 * side + 0), cq); vload_as<NS>(slot(3 * side + 1), cp); vload_as<NS>(slot(3 * side + 2), cg);
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
            const double v = vard[s] * half[s];
            qn[s] = cq[s] + eps * v;
        }
        vstore_as<NS>(slot(8), half);
        store_rows<NS>(K.q_eval + static_cast<long long>(c) * d, d, lane, qn);
        phase = kTickLeap;
    }

    if (end_transition) {
        // ---- statistics, adaptation, outputs (base_hmc.py:155-190), then the next iteration asks for its start density
        TransitionOut out;
        if (P.kind == 0) {
            vload_as<NS>(slot(7), q);
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
        if (tune && P.adapt_mass) {
            MassScalars ms;
            ms.n_samples = first_i32(A.n_samples[c]);
            ms.wsel = first_i32(A.wsel[c]);
            ms.wsum_f = first_f64(A.wsum[c * 2 + ms.wsel]);
            ms.wsum_b = first_f64(A.wsum[c * 2 + (1 - ms.wsel)]);
            ms.window = first_i32(A.awindow[c]);
            double wm[NS], wr[NS], wmb[NS], wrb[NS];
            diag_mass_prefetch<NS>(A, row, ms, wm, wr, wmb, wrb);
            diag_mass_update<NS>(A, P, row, lane, q, var, inv_std, vard, ms, wm, wr, wmb, wrb);
            tm.sync();   // every wave has read the estimator scalars thread 0 rewrites
            if (lane == 0) {
                A.n_samples[c] = ms.n_samples;
                A.wsel[c] = ms.wsel;
                A.awindow[c] = ms.window;
                A.wsum[c * 2 + ms.wsel] = ms.wsum_f;
                A.wsum[c * 2 + (1 - ms.wsel)] = ms.wsum_b;
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                A.var[row + lane * NS + s] = var[s];
                A.inv_std[row + lane * NS + s] = inv_std[s];
            }
        }
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

// lmc_engine_tick_begin(): every chain asks for the density at its current position
template <int NS>
__global__ __launch_bounds__(kWideThreads, 1) void tick_wide_begin_kernel(ChainArrays A, TickArrays K, long long iter_begin) {
    const int c = blockIdx.x;
    const int lane = static_cast<int>(threadIdx.x);
    double q[NS];
    vload<NS>(A.q + static_cast<long long>(c) * A.dpad, q);
    store_rows<NS>(K.q_eval + static_cast<long long>(c) * A.d, A.d, lane, q);
    if (lane == 0) {
        const bool dead = (A.status[c] & kStatusBadInitialEnergy) != 0 || iter_begin >= K.iter_end;
        K.phase[c] = dead ? kTickDone : kTickStart;
        K.git[c] = iter_begin;
    }
}

}  // namespace lmc
