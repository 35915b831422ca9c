// Externally evaluated log-densities ("a batched callable adapter"): the sampler as a
// resumable state machine. The plug-in `logp_dlogp_func` stays on the host side of the C ABI -- any batched
// function (chains x d) -> (chains), (chains x d) on device memory, e.g. a torch-ROCm callable -- and the
// transition kernel is cut at the one place the reference calls it (integration.py:62 and :115):
//
//   tick_kernel:  consume (logp, grad) of the point each chain asked for  ->  advance that chain's iteration
//                 (finish the leapfrog, leaf, merges, doublings, end of transition, adaptation, next iteration's
//                 momentum draw ...) until it needs the density again  ->  write the next point, return.
//
// Chains do not wait for each other: every tick every unfinished chain performs exactly one density evaluation,
// whatever iteration / tree depth it is in. All per-chain state lives in HBM between ticks (ends of the
// trajectory, subtree stack, p_sum, proposal, scalars); the arithmetic is the one of lmc_sampler.hpp, statement for
// statement, so a chain driven through ticks and a chain inside run_kernel agree to the rounding of the density.
//
//   leapfrog  <- /root/reference/littlemcmc/integration.py:100-121 (split at line 115)
//   NUTS      <- nuts.py:204-435        HMC <- hmc.py:140-182        iteration <- base_hmc.py:140-190
//
// ONE statement of the state machine (tick_step) for every shape it runs in (round 5; until then lmc_tick_dense.hpp and
// lmc_tick_wide.hpp were text-substituted copies of this file produced by tools/gen_tick_*.py). What differs between the
// shapes is carried by two policies:
//   Shape -- who a chain's threads are: one wavefront (TickWaveShape, below) or the general kernels' team of 16 wavefronts
//            (TickWideShape, lmc_wide.hip): the team type, and how normal(size=d) reaches the threads' registers;
//   Mass  -- the mass matrix: diagonal (TickDiagMass, below: velocities are one product away and never stored) or dense
//            (TickDenseMass, lmc_dense.hip: velocities are matrix sweeps, stored with trajectory ends and tree nodes, one
//            sweep per leapfrog forms v = C p and w = C g, FullAdapt's update runs in dense_adapt_kernel between ticks).
//
// HBM traffic per tick (round 5; profiles/r05_tick_*): a tick reads and writes only what its step touches --
//   * a LEAF parked at level 0 is {p, q} (its left end, right end and momentum sum are the same vector), a level-1 node
//     {lp, rp, q} (its momentum sum is lp + rp, the very sum the merge formed: nuts.py:386);
//   * the per-level scalars are read for the levels this leaf merges with only, and written for the level it parks at;
//   * inv_std is read by the ticks that draw a momentum or adapt the mass matrix, not by every leapfrog;
//   * the end a doubling extends is not re-read when it is the end the previous doubling left in registers, nor the start
//     state right after it was stored; the trajectory's proposal is written when it changes and read when the transition ends.
#pragma once
#include "lmc_tick_launch.hpp"

namespace lmc {

// ---- Shape policy: one wavefront per chain ------------------------------------------------------------------------
struct TickWaveShape {
    typedef Team<1> TeamT;
    static constexpr int kThreads = 64;
    TeamT tm;
    __device__ __forceinline__ TickWaveShape(double*, int) : tm{nullptr, 0} {}
    // normal(size=d) of the chain's stream into the owning threads' registers; the one-wavefront form also LEAVES them in
    // lds[0, d) (the dense momentum sweeps read their operand there)
    template <int NS>
    __device__ __forceinline__ void normals(RngState& rng, int d, int dpad, double* lds, double (&z)[NS]) {
        rng_normals(rng, d, lds, lds + dpad);
        const int t = static_cast<int>(threadIdx.x);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int e = t * NS + s;
            z[s] = (e < d) ? lds[e] : 0.0;
        }
    }
};

// ---- Mass policy: diagonal (QuadPotentialDiag / DiagAdapt, float32 storage) ---------------------------------------------
// per-chain HBM row (A.scratch): 0-2 left end {q, p, g}, 3-5 right end, 6 p_sum, 7 proposal q, 8 half-stepped momentum,
// then 4 vectors per subtree level {lp, rp, psum, proposal q} (tick_scratch_vectors)
template <int NS>
struct TickDiagMass {
    static constexpr bool kDense = false;
    static constexpr int kEndVecs = 3, kLevelVecs = 4, kPsum = 6, kProp = 7, kHalf = 8, kV0s = -1, kFixed = 9;
    static constexpr int kLp = 0, kLv = -1, kRp = 1, kRv = -1, kPs = 2, kQ = 3;   // vectors of a parked node
    const ChainArrays& A;
    long long row;
    int tid;
    float var[NS], inv_std[NS];
    double vard[NS];
    __device__ __forceinline__ TickDiagMass(const ChainArrays& A_, long long row_, int tid_) : A(A_), row(row_), tid(tid_) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            var[s] = A.var[row + tid * NS + s];
            vard[s] = static_cast<double>(var[s]);
            inv_std[s] = 0.0f;
        }
    }
    __device__ __forceinline__ void load_inv_std() {
#pragma unroll
        for (int s = 0; s < NS; ++s) inv_std[s] = A.inv_std[row + tid * NS + s];
    }
    // quadpotential.py:221-224 / :374-376
    template <class TeamT>
    __device__ __forceinline__ void momentum(TeamT&, int, double*, bool momentum_f32, const double (&z)[NS], double (&p0)[NS]) {
        load_inv_std();
#pragma unroll
        for (int s = 0; s < NS; ++s)
            p0[s] = momentum_f32 ? static_cast<double>(inv_std[s] * static_cast<float>(z[s])) : z[s] * static_cast<double>(inv_std[s]);
    }
    // integration.py:52-66: e0; (v0, w0, v0s are the dense policy's)
    template <class TeamT>
    __device__ __forceinline__ double start_state(TeamT& tm, double* lds, int d, int dpad, bool momentum_f32, int sdot_mode,
                                                  const double (&p0)[NS], const double (&)[NS], double logp0, double (&)[NS],
                                                  double (&)[NS], double (&)[NS]) {
        if (momentum_f32) {
            const float kin = start_kinetic_f32<NS>(tm, p0, var, d, sdot_mode, reinterpret_cast<float*>(lds), dpad);
            return first_f64(static_cast<double>(kin) - logp0);
        }
        return first_f64(0.5 * tm.sum(pdot_v<NS>(p0, vard, p0)) - logp0);
    }
    // v = M^-1 p (and w = M^-1 g for the dense policy's one-sweep leapfrog)
    __device__ __forceinline__ void velocity(double*, const double (&p)[NS], const double (&)[NS], double (&v)[NS], double (&)[NS]) {
#pragma unroll
        for (int s = 0; s < NS; ++s) v[s] = vard[s] * p[s];
    }
};

// register budget per vector width (waves per SIMD): the tick kernel is latency / bandwidth bound and insensitive to
// occupancy (4 / 6 / 8 waves measured equal at NS = 2), so wide vectors simply get the registers they need
constexpr int tick_waves_per_simd(int ns) { return ns <= 2 ? 4 : ns == 4 ? 2 : 1; }

// ---- one tick of one chain ----------------------------------------------------------------------------------------
// adapt_mask: dense FullAdapt only (the chain finished a tuning iteration in this tick: the host launches dense_adapt_kernel
// masked before the next tick), else nullptr.
template <int NS, class Shape, class Mass>
__device__ __forceinline__ void tick_step(const ChainArrays& A, const TickArrays& K, const SamplerParams& P, const double* logp_in,
                                          const double* grad_in, double* lds, Shape& shape, Mass& mass, int* adapt_mask) {
    typedef typename Shape::TeamT TeamT;
    TeamT& tm = shape.tm;
    constexpr bool kDense = Mass::kDense;
    const int c = blockIdx.x;
    const int lane = static_cast<int>(threadIdx.x);   // the thread's index in its chain
    const int d = A.d, dpad = A.dpad;
    int phase = first_i32(K.phase[c]);
    if (phase == kTickDone) return;
    const long long row = static_cast<long long>(c) * dpad;
    glb_double* scr = (glb_double*)(A.scratch + static_cast<long long>(c) * A.scratch_stride);
    auto slot = [&](int k) { return scr + k * dpad; };
    auto end_slot = [&](int side, int k) { return scr + (Mass::kEndVecs * side + k) * dpad; };
    auto level = [&](int j, int k) { return scr + (Mass::kFixed + Mass::kLevelVecs * j + k) * dpad; };

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
    // Per-level subtree scalars (lane j of every wave holds level j): a leaf merges with levels 0 .. m-1, m = its number of
    // trailing one bits, and a level is always written (parked) before it is read -- so only lanes < m are fetched, only the
    // lane a node is parked at is written back, and the rare weight-offset move rescales the other parked levels in memory.
    LevelScalars lsc = {0.0, 0.0, 0.0, 0.0};
    double* lvl = K.lvl + static_cast<long long>(c) * 4 * kTickLevels;
    const int n_merge = (phase == kTickLeap && P.kind == 0) ? __builtin_ctz(~static_cast<unsigned>(leaf)) : 0;
    if (lane_id() < n_merge) {
        lsc.w = lvl[lane_id()]; lsc.a = lvl[kTickLevels + lane_id()]; lsc.pe = lvl[2 * kTickLevels + lane_id()];
        lsc.plogp = lvl[3 * kTickLevels + lane_id()];
    }
    int parked_at = -1;   // level whose scalars this tick wrote
    const bool momentum_f32 = P.momentum_f32 != 0;
    int status = 0;

    // what this tick decides
    bool begin_doubling = false, subtree_done = false, end_transition = false, need_leap = false;
    bool diverging = false, turning = false, exhausted = false, accepted = false;
    bool have_end = false;                  // {cq, cp, cg, (cv, cw)} hold the end of side `right` as stored in the row
    double cq[NS], cp[NS], cg[NS];          // the state the next leapfrog starts from: q, p, g
    double cv[NS], cw[NS];                  // dense: v = C p, w = C g (unused with a diagonal)
    double q[NS];                           // the chain's position (start of the iteration / its result)
    double tlp[NS], trp[NS], tps[NS], tq[NS];   // node in flight: left / right end momentum, momentum sum, proposal position
    double tlv[NS], trv[NS];                    // ... and the velocities of its two ends
    double tw = 0.0, ta = 0.0, tpe = 0.0, tplogp = 0.0;
    const double logp_new = first_f64(logp_in[c]);

    if (phase == kTickStart) {
        // ---- the iteration begins (base_hmc.py:140-153): momentum draw, start state from the delivered density
        vload<NS>(A.q + row, q);
        double g0[NS];
        load_rows<NS>(grad_in + static_cast<long long>(c) * d, d, lane, g0);
        double z[NS], p0[NS];
        shape.template normals<NS>(rng, d, dpad, lds, z);
        mass.momentum(tm, d, lds, momentum_f32, z, p0);
        tm.sync();
        logp0 = logp_new;
        double v0[NS], w0[NS], v0s[NS];   // dense only
        e0 = mass.start_state(tm, lds, d, dpad, momentum_f32, P.sdot_mode, p0, g0, logp0, v0, w0, v0s);
        if (!isfinite(e0)) {   // base_hmc.py:145-148
            if (lane == 0) { A.status[c] |= kStatusBadInitialEnergy; K.phase[c] = kTickDone; }
            return;
        }
        const bool adapt_step = tune && P.adapt_step_size;
        step_size = jitter_step_size(tm, rng, A, P, c, adapt_step ? da.step_now : da.step_bar_now);
        n_leap = 0;
        vcopy(cq, q); vcopy(cp, p0); vcopy(cg, g0);
        if constexpr (kDense) { vcopy(cv, v0); vcopy(cw, w0); }
        if (P.kind == 0) {
            max_depth = (tune && iter_count < 200) ? P.early_max_treedepth : P.max_treedepth;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                vstore_as<NS>(end_slot(r, 0), q); vstore_as<NS>(end_slot(r, 1), p0); vstore_as<NS>(end_slot(r, 2), g0);
                if constexpr (kDense) { vstore_as<NS>(end_slot(r, 3), v0); vstore_as<NS>(end_slot(r, 4), w0); }
            }
            vstore_as<NS>(slot(Mass::kPsum), p0); vstore_as<NS>(slot(Mass::kProp), q);
            if constexpr (kDense) {
                vstore_as<NS>(slot(Mass::kV0s), v0s);
                l_start = true; r_start = true;   // the end still is the start state: its stored velocity is v0s
            } else {
                l_start = momentum_f32; r_start = momentum_f32;
            }
            prop_e = e0; prop_logp = logp0;
            coff = 0.0; c_tot = 0.0; w_start = 1.0; wn = 0.0; an = 0.0; max_de = 0.0;
            depth = 0;
            have_end = true;      // both ends ARE the start state in registers: the first doubling reads nothing back
            begin_doubling = true;
        } else {   // hmc.py:143-149
            plen = first_f64(team_uniform(tm, rng, win) * P.path_length);
            n_steps = static_cast<int>(plen / step_size);
            n_steps = n_steps < 1 ? 1 : n_steps;
            n_steps = n_steps > P.max_steps ? P.max_steps : n_steps;
            eps = step_size;
            need_leap = true;
        }
    } else {
        // ---- second half of the leapfrog (integration.py:115-121) with the delivered gradient
        double half[NS];
        load_rows<NS>(K.q_eval + static_cast<long long>(c) * d, d, lane, cq);
        load_rows<NS>(grad_in + static_cast<long long>(c) * d, d, lane, cg);
        vload_as<NS>(slot(Mass::kHalf), half);
        const double dt = 0.5 * eps;
#pragma unroll
        for (int s = 0; s < NS; ++s) cp[s] = half[s] + dt * cg[s];
        double cvel[NS];
        if constexpr (kDense) {
            mass.velocity(lds, cp, cg, cv, cw);   // the one matrix sweep of this leapfrog: v = C p, w = C g
            vcopy(cvel, cv);
        } else {
            mass.velocity(lds, cp, cg, cvel, cvel);
        }
        const double energy = first_f64(0.5 * tm.sum(pdot<NS>(cp, cvel)) - logp_new);
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
                if (x - coff > 600.0) {   // cold path: move the offset, rescale every stored weight -- the levels this leaf
                    const double f = exp_uniform(coff - x);   // merges with in registers, the other parked ones where they lie
                    lsc.w *= f; lsc.a *= f;
                    if (lane >= n_merge && lane < kTickLevels) { lvl[lane] *= f; lvl[kTickLevels + lane] *= f; }
                    coff = x;
                }
                tw = exp_uniform_fast(x - coff);
                const double sat = (coff == 0.0) ? fmin(1.0, tw) : ((x >= 0.0) ? 1.0 : exp_uniform(x));
                ta = tw * sat;
                vcopy(tlp, cp); vcopy(trp, cp); vcopy(tps, cp); vcopy(tlv, cvel); vcopy(trv, cvel); vcopy(tq, cq);
                tpe = energy; tplogp = logp_new;
                int j = 0;
                while ((leaf >> j) & 1) {
                    double alp[NS], alv[NS], arp[NS], arv[NS], aps[NS], aq[NS];
                    double aw, aa, ape, aplogp;
                    // node a = the one parked at level j: a leaf {p, q}, a level-1 node {lp, rp, q}, else {lp, rp, psum, q}
                    vload_as<NS>(level(j, Mass::kLp), alp); vload_as<NS>(level(j, Mass::kQ), aq);
                    if constexpr (kDense) vload_as<NS>(level(j, Mass::kLv), alv);
                    if (j == 0) {
                        vcopy(arp, alp); vcopy(aps, alp);
                        if constexpr (kDense) vcopy(arv, alv);
                    } else {
                        vload_as<NS>(level(j, Mass::kRp), arp);
                        if constexpr (kDense) vload_as<NS>(level(j, Mass::kRv), arv);
                        if (j == 1) {
#pragma unroll
                            for (int s = 0; s < NS; ++s) aps[s] = alp[s] + arp[s];   // the very sum the level-0 merge formed
                        } else {
                            vload_as<NS>(level(j, Mass::kPs), aps);
                        }
                    }
                    if constexpr (!kDense) {
#pragma unroll
                        for (int s = 0; s < NS; ++s) { alv[s] = mass.vard[s] * alp[s]; arv[s] = mass.vard[s] * arp[s]; }
                    }
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
                    const bool take_b = uniform_true(team_uniform(tm, rng, win) * wsum < tw);
                    vcopy(tlp, alp); vcopy(tlv, alv); vcopy(tps, ps);
                    if (!take_b) { vcopy(tq, aq); tpe = ape; tplogp = aplogp; }
                    tw = wsum; ta = asum;
                    ++j;
                    if (turn) { turning = true; break; }
                }
                if (!turning) {
                    if (leaf + 1 < (1 << depth)) {   // park the node, continue the subtree from (cq, cp, cg)
                        vstore_as<NS>(level(j, Mass::kLp), tlp); vstore_as<NS>(level(j, Mass::kQ), tq);
                        if constexpr (kDense) vstore_as<NS>(level(j, Mass::kLv), tlv);
                        if (j > 0) {
                            vstore_as<NS>(level(j, Mass::kRp), trp);
                            if constexpr (kDense) vstore_as<NS>(level(j, Mass::kRv), trv);
                            if (j > 1) vstore_as<NS>(level(j, Mass::kPs), tps);
                        }
                        lsc.put(j, tw, ta, tpe, tplogp);
                        parked_at = j;
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
                double psum[NS];
                vload_as<NS>(slot(Mass::kPsum), psum);
                if (c_tot != coff) {   // the offset moved inside this subtree: bring the accepted totals to it (rare)
                    const double f = exp_uniform(c_tot - coff);
                    wn = first_f64(wn * f); an = first_f64(an * f); w_start = first_f64(w_start * f);
                    c_tot = coff;
                }
                if (uniform_true(team_uniform(tm, rng, win) * (w_start + wn) < tw)) {
                    prop_e = tpe; prop_logp = tplogp;
                    vstore_as<NS>(slot(Mass::kProp), tq);
                }
                wn = first_f64(wn + tw);
                an = first_f64(an + ta);
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const double t = psum[s] + tps[s];
                    psum[s] = momentum_f32 ? static_cast<double>(static_cast<float>(t)) : t;
                }
                vstore_as<NS>(slot(Mass::kPsum), psum);
                double Lp[NS], Rp[NS], oLv[NS], oRv[NS];
                vload_as<NS>(end_slot(0, 1), Lp); vload_as<NS>(end_slot(1, 1), Rp);
                if constexpr (kDense) {
                    vload_as<NS>(l_start ? slot(Mass::kV0s) : end_slot(0, 3), oLv);
                    vload_as<NS>(r_start ? slot(Mass::kV0s) : end_slot(1, 3), oRv);
                } else {
                    end_velocity<NS>(oLv, mass.vard, Lp, l_start);
                    end_velocity<NS>(oRv, mass.vard, Rp, r_start);
                }
                double dots[6], p1[NS], p2[NS];
                const int side = right ? 1 : 0;
                if (right) {
#pragma unroll
                    for (int s = 0; s < NS; ++s) { p1[s] = psum[s] + tlp[s]; p2[s] = Rp[s] + tps[s]; }
                    dots[0] = pdot<NS>(psum, oLv); dots[1] = pdot<NS>(psum, trv);
                    dots[2] = pdot<NS>(p1, oLv);   dots[3] = pdot<NS>(p1, tlv);
                    dots[4] = pdot<NS>(p2, oRv);   dots[5] = pdot<NS>(p2, trv);
                    r_start = false;
                } else {
#pragma unroll
                    for (int s = 0; s < NS; ++s) { p1[s] = tps[s] + Lp[s]; p2[s] = tlp[s] + psum[s]; }
                    dots[0] = pdot<NS>(psum, trv); dots[1] = pdot<NS>(psum, oRv);
                    dots[2] = pdot<NS>(p1, trv);   dots[3] = pdot<NS>(p1, oLv);
                    dots[4] = pdot<NS>(p2, tlv);   dots[5] = pdot<NS>(p2, oRv);
                    l_start = false;
                }
                vstore_as<NS>(end_slot(side, 0), cq); vstore_as<NS>(end_slot(side, 1), cp); vstore_as<NS>(end_slot(side, 2), cg);
                if constexpr (kDense) { vstore_as<NS>(end_slot(side, 3), cv); vstore_as<NS>(end_slot(side, 4), cw); }
                have_end = true;   // registers == the row's end of side `right`
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
        const bool was_right = right;
        right = team_uniform(tm, rng, win) < 0.5;
        eps = right ? step_size : -step_size;
        const int side = right ? 1 : 0;
        // the registers already hold this end when it is the one the previous doubling extended (it was stored a moment
        // ago) or when the iteration has just begun (both ends are the start state)
        if (!(have_end && (phase == kTickStart || right == was_right))) {
            vload_as<NS>(end_slot(side, 0), cq); vload_as<NS>(end_slot(side, 1), cp); vload_as<NS>(end_slot(side, 2), cg);
            if constexpr (kDense) { vload_as<NS>(end_slot(side, 3), cv); vload_as<NS>(end_slot(side, 4), cw); }
        }
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
            double v;
            if constexpr (kDense) v = cv[s] + dt * cw[s];   // C (p + dt g)
            else v = mass.vard[s] * half[s];
            qn[s] = cq[s] + eps * v;
        }
        vstore_as<NS>(slot(Mass::kHalf), half);
        store_rows<NS>(K.q_eval + static_cast<long long>(c) * d, d, lane, qn);
        phase = kTickLeap;
    }

    if (end_transition) {
        // ---- statistics, adaptation, outputs (base_hmc.py:155-190), then the next iteration asks for its start density
        TransitionOut out;
        if (P.kind == 0) {
            vload_as<NS>(slot(Mass::kProp), q);
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
        if constexpr (kDense) {
            // FullAdapt.update for this chain runs in dense_adapt_kernel right after this tick (the host launches it masked)
            if (tune && adapt_mask != nullptr && lane == 0) adapt_mask[c] = 1;
        } else {
            if (tune && P.adapt_mass) {
                MassScalars ms;
                ms.n_samples = first_i32(A.n_samples[c]);
                ms.wsel = first_i32(A.wsel[c]);
                ms.wsum_f = first_f64(A.wsum[c * 2 + ms.wsel]);
                ms.wsum_b = first_f64(A.wsum[c * 2 + (1 - ms.wsel)]);
                ms.window = first_i32(A.awindow[c]);
                mass.load_inv_std();
                double wm[NS], wr[NS], wmb[NS], wrb[NS];
                diag_mass_prefetch<NS>(A, row, ms, wm, wr, wmb, wrb);
                diag_mass_update<NS>(A, P, row, lane, q, mass.var, mass.inv_std, mass.vard, ms, wm, wr, wmb, wrb);
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
                    A.var[row + lane * NS + s] = mass.var[s];
                    A.inv_std[row + lane * NS + s] = mass.inv_std[s];
                }
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
    if (lane == parked_at) {   // (thread j of the team's first wave: every wave's lane j holds the same scalars)
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

// one wavefront per chain, diagonal mass matrix: densities evaluated by the caller (targets.TorchTarget / CallableTarget)
template <int NS>
__global__ __launch_bounds__(64, tick_waves_per_simd(NS)) void tick_kernel(ChainArrays A, TickArrays K, SamplerParams P, const double* logp_in,
                                                  const double* grad_in) {
    extern __shared__ __attribute__((aligned(16))) double lds[];   // 2 * dpad doubles: normals + staging / sdot staging
    TickWaveShape shape(lds, A.dpad);
    TickDiagMass<NS> mass(A, static_cast<long long>(blockIdx.x) * A.dpad, static_cast<int>(threadIdx.x));
    tick_step<NS>(A, K, P, logp_in, grad_in, lds, shape, mass, nullptr);
}

// lmc_engine_tick_begin(): every chain asks for the density at its current position
template <int NS, int THREADS = 64>
__global__ __launch_bounds__(THREADS) void tick_begin_kernel(ChainArrays A, TickArrays K, long long iter_begin) {
    const int c = blockIdx.x;
    const int lane = static_cast<int>(threadIdx.x);   // the thread's index in its chain
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
