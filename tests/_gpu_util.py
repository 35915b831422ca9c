"""Helpers shared by the -m gpu parity tests: build a device target / engine from a golden case and
compare device chains with the oracle's, aware of decision margins.

Why margins: the reference evaluates the start state's kinetic energy with a float32 BLAS ``sdot``
whose summation order is CPU/BLAS specific; the device sums the same float32 products in float64 and
rounds once. The two start energies can therefore differ by a few float32 ulps (~1e-5 at d=128),
which can flip a multinomial/Metropolis decision only when |log U - log p| is below that. The oracle
reports the smallest such margin per transition; chains are required to match bit-for-bit (integer
stats) up to the first transition whose margin is below ``FRAGILE`` -- and there must be few of those.
"""
import numpy as np

import littlemcmc_amd as lmc
from littlemcmc_amd import targets as T

FRAGILE = 1e-4          # |log U - log p| below this may legitimately flip (float32 start energy)
RTOL_Q = 1e-7           # positions: different reduction order / libm ulps, amplified over a chain
INT_STATS = ("depth", "tree_size", "diverging", "n_steps", "accepted", "tune")


def device_target(family, d, params):
    family = str(family)
    if family == "std_normal":
        return T.StdNormal(d)
    if family == "diag_gaussian":
        return T.DiagGaussian(params)
    if family == "ar1":
        t = T.AR1(d)
        np.testing.assert_array_equal(t.params, params)
        return t
    if family == "funnel":
        return T.Funnel(d)
    if family == "normal1d":
        return T.Normal1D(*params)
    raise ValueError(family)


def kwargs_from(g, prefix=""):
    kw = {}
    for n, v in zip(g[prefix + "kw_names"], g[prefix + "kw_vals"]):
        kw[str(n)] = int(v) if str(n) in ("max_treedepth", "early_max_treedepth", "max_steps") else float(v)
    return kw


def first_mismatch(got, want):
    """Index of the first iteration where any integer stat differs (None if all equal)."""
    bad = None
    for name in got:
        if name in INT_STATS and name in want:
            idx = np.nonzero(np.ravel(got[name]) != np.ravel(want[name]))[0]
            if len(idx) and (bad is None or idx[0] < bad):
                bad = int(idx[0])
    return bad


GROWTH = 1e4            # a chain that parts from its twin does so geometrically (x2 - x10 per tuned iteration, DESIGN.md
                        # section 5): the iteration BEFORE the first difference already shows >= 1 / GROWTH of the tolerance
JUMP = 1e4              # ... and it grows by at most this factor from one iteration to the next: the largest ratio measured between
                        # consecutive iterations of any captured chain once its separation is measurable is 250 (tools/growth_probe.py,
                        # profiles/r06_growth_probe.txt; x2 - x10 is typical) -- a drift that is 1e4 times the previous
                        # iteration's is a jump, whatever came before (round 5's review: "after a dozen iterations everything
                        # has a reason")
TURN_FRAGILE = 1e-9     # |p_sum . v| below this (relative to the dot's own scale ~ d) is a U-turn test within reduction-order noise


def _separation(got_q, got_stats, want_q, want_stats):
    """Per iteration, in units of the comparison's own tolerances: position error, step-size error (the value dual
    averaging hands to the NEXT iteration) and energy error (the start energy is formed before any decision)."""
    n = len(want_q)
    err = np.max(np.abs(got_q - want_q) / (1e-9 + np.abs(want_q) * RTOL_Q), axis=1)
    step_err = np.zeros(n)
    if "step_size" in got_stats and "step_size" in want_stats:
        g, w = np.ravel(got_stats["step_size"])[:n], np.ravel(want_stats["step_size"])[:n]
        step_err = np.abs(g - w) / (np.abs(w) * RTOL_Q)
    e_err = np.zeros(n)
    if "energy" in got_stats and "energy" in want_stats:
        g, w = np.ravel(got_stats["energy"])[:n], np.ravel(want_stats["energy"])[:n]
        e_err = np.abs(g - w) / (1e-7 * (1.0 + np.abs(w)))     # float32 ulp scale of the start state's kinetic energy
    return err, step_err, e_err


def explain_first_difference(i, err, step_err, e_err, margins, integer=False):
    """Why may iteration i be the first at which a device chain and its oracle twin differ (beyond RTOL_Q, or in an
    integer statistic)?  Returns a reason or None. Admissible reasons -- each one a measured property of the chain, not
    a blanket allowance:
      * "margin": a multinomial / Metropolis decision at or before i sat within FRAGILE of its threshold;
      * "turn":   a U-turn dot product at or before i was zero to reduction-order noise;
      * "f32 start energy": the energies of iteration i itself already differ at the float32-ulp scale (a host whose
        sdot rounds differently from the capture host's: the start energy precedes every decision of the iteration);
      * "growth": the iteration before i already shows >= 1 / GROWTH of the tolerance in position or step size, i.e. the
        two chains were parting geometrically (dual-averaging feedback), not jumping -- and, when the difference at i is a
        position drift (``integer`` False: an integer statistic that flips changes the draw altogether), it is at most JUMP
        times the previous iteration's separation.
    A difference that appears out of nowhere -- positions equal to 1e-13, step sizes equal, no fragile decision, then
    1e-6 apart -- has no reason and fails (round 4's review: such a regression used to shorten `upto` silently)."""
    m = np.asarray(margins)
    lb = m[: i + 1] if m.ndim == 1 else m[: i + 1, 0]
    if (lb < FRAGILE).any():
        return "margin"
    if m.ndim == 2 and m.shape[1] > 1 and (m[: i + 1, 1] < TURN_FRAGILE).any():
        return "turn"
    if e_err[i] >= 1.0:
        return "f32 start energy"
    if i > 0 and max(err[i - 1], step_err[i - 1]) * GROWTH >= 1.0:
        if integer or err[i] <= JUMP * max(err[i - 1], step_err[i - 1]):
            return "growth"
    return None


def assert_chain_matches(got_q, got_stats, want_q, want_stats, margins, label=""):
    """got/want: per-iteration arrays of ONE chain. margins[i] = oracle's smallest logbern margin at i (or the oracle's
    [n, 3] rows {logbern, U-turn, divergence}).
    Returns the number of iterations verified bit-exactly. The first difference -- a position beyond RTOL_Q or an integer
    statistic -- must have a reason (explain_first_difference)."""
    n = len(want_q)
    bad = first_mismatch(got_stats, want_stats)
    # positions drift apart geometrically under tuning (dual-averaging feedback); compare while they agree
    err, step_err, e_err = _separation(got_q, got_stats, want_q, want_stats)
    drift = np.nonzero(err > 1.0)[0]
    upto = n if bad is None else bad
    if len(drift):
        upto = min(upto, int(drift[0]))
    if upto < n:
        why = explain_first_difference(upto, err, step_err, e_err, margins, integer=(bad is not None and bad == upto))
        assert why is not None, (
            "%s: chain parts from the oracle at iteration %d (%s) with no reason: position error %.3g of the tolerance "
            "there and %.3g / step-size error %.3g the iteration before, energy error %.3g of a float32 ulp, smallest "
            "logbern margin so far %.3g" % (
                label, upto, "integer statistic" if (bad is not None and bad == upto) else "position drift",
                err[upto], err[upto - 1] if upto else 0.0, step_err[upto - 1] if upto else 0.0, e_err[upto],
                float(np.min(np.asarray(margins)[: upto + 1] if np.asarray(margins).ndim == 1 else np.asarray(margins)[: upto + 1, 0]))))
    for name in got_stats:
        g, w = np.ravel(got_stats[name])[:upto], np.ravel(want_stats[name])[:upto]
        if name in INT_STATS:
            np.testing.assert_array_equal(g, w, err_msg="%s %s" % (label, name))
        else:
            # energies carry the float32 start-energy rounding: a few float32 ulps of the kinetic energy
            scale = 1.0
            if "energy" in want_stats:
                scale = 1.0 + np.abs(np.ravel(want_stats["energy"])[:upto])
            assert np.all(np.abs(g - w) <= 1e-6 * np.abs(w) + 2e-6 * scale), "%s %s: max abs diff %g" % (
                label, name, np.max(np.abs(g - w)))
    np.testing.assert_allclose(got_q[:upto], want_q[:upto], rtol=RTOL_Q, atol=1e-9, err_msg="%s q" % label)
    return upto


# ---------------------------------------------------------------------------------------------------
# per-iteration ("teacher-forced") parity: every iteration of an oracle chain is replayed on the device
# from the oracle's exact pre-iteration state, all iterations of a chain at once (one per wavefront).
# ---------------------------------------------------------------------------------------------------
def oracle_chain_snapshots(ostep, start, seed, tune, draws):
    """Run one oracle chain like sampling.py:481-521; return (snapshots, outputs) per iteration."""
    from oracle import lmc_oracle as orc  # noqa: F401

    rng = np.random.RandomState(int(seed))
    q = np.array(start, dtype="d")
    ostep.tune = bool(tune)
    ostep.reset_tuning()
    snaps, outs = [], []
    adaptive = hasattr(ostep.pot, "fore")
    for i in range(tune + draws):
        if i == 0:
            ostep.iter_count = 0
        if i == tune:
            ostep.tune = False
        pot, ad = ostep.pot, ostep.adapt
        snap = dict(q=q.copy(), rng=rng.get_state(), tune=ostep.tune, iter_count=ostep.iter_count,
                    var=pot.var.copy(), log_step=float(np.ravel(ad.log_step)[0]), log_bar=float(np.ravel(ad.log_bar)[0]),
                    hbar=float(np.ravel(ad.hbar)[0]), da_count=ad.count, n_samples=pot.n_samples)
        if adaptive:
            snap.update(fore_mean=pot.fore.mean.copy(), fore_raw_var=pot.fore.raw_var.copy(), fore_w_sum=pot.fore.w_sum,
                        back_mean=pot.back.mean.copy(), back_raw_var=pot.back.raw_var.copy(), back_w_sum=pot.back.w_sum,
                        window=pot.window)
        q, st = ostep.astep(q, rng)
        snaps.append(snap)
        outs.append(dict(q=q.copy(), stats={k: np.ravel(v)[0] for k, v in st.items()}, margin=ostep.last_margins.lb,
                         turn_margin=ostep.last_margins.turn, rng_pos=rng.get_state()[2]))
    return snaps, outs


def replay_iterations_on_device(step, snaps, outs, label="", expect_plan=None):
    """One engine, one wavefront per snapshot; run ONE iteration everywhere; compare with the oracle.
    ``expect_plan``: the LDS plan ("shallow" / "deep") the launch must have run under (Engine.last_run_plan()).
    Returns (n_checked, n_fragile)."""
    from littlemcmc_amd import _abi

    checked = fragile = 0
    for tune_flag in (True, False):
        idx = [i for i, s in enumerate(snaps) if s["tune"] == tune_flag]
        if not idx:
            continue
        n = len(idx)
        eng = step._make_engine(n)
        try:
            eng.set_position(np.stack([snaps[i]["q"] for i in idx]))
            for c, i in enumerate(idx):
                eng.set_rng_state(c, snaps[i]["rng"])
            state = {k: np.stack([np.asarray(snaps[i][k]) for i in idx]) for k in
                     ("var", "log_step", "log_bar", "hbar", "da_count", "iter_count", "n_samples")}
            if "fore_mean" in snaps[idx[0]]:
                for k in ("fore_mean", "fore_raw_var", "back_mean", "back_raw_var", "fore_w_sum", "back_w_sum", "window"):
                    state[k] = np.stack([np.asarray(snaps[i][k]) for i in idx])
            if eng.mass_f64:   # QuadPotentialDiagAdapt(dtype="float64"): the diagonal travels in its own dtype
                state["var64"] = state.pop("var")
            eng.set_chain_state(state)
            eng.reserve(1, keep_trace=True)
            eng.run(1 if tune_flag else 0, 0, 1)
            assert not eng.status().any()
            if expect_plan is not None:
                assert eng.last_run_plan() == expect_plan, (label, eng.last_run_plan(), eng.kernel_shape())
            q = eng.trace()[:, 0]
            stats = {k: v[:, 0] for k, v in step._stats_from_engine(eng, 0, 1).items()}
            after = eng.get_chain_state()
            for c, i in enumerate(idx):
                want = outs[i]
                tag = "%s iter %d" % (label, i)
                if want["margin"] < 1e-9 or want["turn_margin"] < 1e-9:   # a coin flip within reduction-order noise
                    fragile += 1
                    continue
                for name, val in want["stats"].items():
                    got = stats[name][c]
                    if name in INT_STATS:
                        assert got == val, (tag, name, got, val)
                    else:
                        assert np.isclose(got, val, rtol=1e-10, atol=1e-10), (tag, name, got, val)
                np.testing.assert_allclose(q[c], want["q"], rtol=1e-11, atol=1e-12, err_msg=tag)
                assert eng.get_rng_state(c)[2] == want["rng_pos"], tag
                if i + 1 < len(snaps):   # adaptation state after the iteration == oracle's next snapshot
                    nxt = snaps[i + 1]
                    np.testing.assert_allclose(after["var"][c], nxt["var"], rtol=2e-7, err_msg=tag + " var")
                    if eng.mass_f64:
                        np.testing.assert_allclose(after["var64"][c], nxt["var"], rtol=1e-12, err_msg=tag + " var64")
                    for k in ("log_step", "log_bar", "hbar"):
                        assert np.isclose(after[k][c], nxt[k], rtol=1e-11, atol=1e-13), (tag, k, after[k][c], nxt[k])
                    assert after["da_count"][c] == nxt["da_count"] and after["n_samples"][c] == nxt["n_samples"], tag
                    if "fore_mean" in nxt and nxt["tune"] == tune_flag:
                        for k in ("fore_mean", "fore_raw_var", "back_mean", "back_raw_var"):
                            np.testing.assert_allclose(after[k][c], nxt[k], rtol=1e-11, atol=1e-13, err_msg=tag + " " + k)
                        assert after["fore_w_sum"][c] == nxt["fore_w_sum"] and after["back_w_sum"][c] == nxt["back_w_sum"], tag
                        assert after["window"][c] == nxt["window"], (tag, after["window"][c], nxt["window"])
                checked += 1
        finally:
            eng.close()
    return checked, fragile


# ---------------------------------------------------------------------------------------------------
# oracle chains inside a full-size job: the seeds are prefix stable over the GLOBAL chain index space, so chain i of
# a 65 536-chain device job is the oracle's one-chain run with seeds[i]
# ---------------------------------------------------------------------------------------------------
def assert_selected_chains_match_oracle(family, d, seeds, start, sel, n_it, trace, stats, okw=None, label=""):
    """trace[chains, >= n_it, d] / stats[name][chains, >= n_it, 1] of a device job (iterations from 0, tuning included);
    chains ``sel`` are compared with oracle chains on the same global seeds over the first n_it iterations: integer
    statistics exact, positions to RTOL_Q. Returns the iterations verified per chain."""
    from oracle import lmc_oracle as orc
    from oracle import targets as OT

    f = OT.make(family, d)
    okw = dict(okw or {})
    _s, ostep = orc.init_nuts(f, d, seeds=seeds, **okw)
    np.testing.assert_array_equal(_s, start)
    ot, os_, margins = orc.sample(f, d, draws=0, tune=n_it, step=ostep, chains=len(sel), start=start,
                                  random_seed=[seeds[i] for i in sel], discard_tuned_samples=False, record_margins=True)
    out = []
    for k, c in enumerate(sel):
        got = {n_: np.asarray(stats[n_])[c, :n_it].reshape(n_it) for n_ in stats}
        want = {n_: os_[n_][k, :, 0] for n_ in os_ if n_ in stats}
        out.append(assert_chain_matches(np.asarray(trace)[c, :n_it], got, ot[k], want, margins[k],
                                        label="%s chain %d" % (label, c)))
    return out


def replay_golden_run(golden_dir, name, lds_plan="auto"):
    """Every iteration of every chain of the reference-captured run ``tests/golden/<name>.npz``, replayed on the device from the
    oracle's exact pre-iteration state (integer statistics exact, positions / energies / adaptation state to 1e-10).
    ``lds_plan`` pins the LDS plan of the one-wave sampling kernels ("shallow" / "deep": lmc_config.lds_plan).
    Returns (checked, fragile, total)."""
    import os

    from oracle import lmc_oracle as orc
    from oracle import targets as OT

    g = np.load(os.path.join(golden_dir, name + ".npz"))
    d, chains, tune, draws = int(g["d"]), int(g["chains"]), int(g["tune"]), int(g["draws"])
    kw = kwargs_from(g)
    fam = str(g["family"])
    f = OT.DiagGaussian(g["params"]) if fam == "diag_gaussian" else OT.make(fam, d)
    tgt = device_target(fam, d, g["params"])
    seeds = [int(s) for s in g["seeds"]]
    start = g["start"]
    dkw = dict(kw)
    if lds_plan != "auto":
        dkw["lds_plan"] = lds_plan
    total_checked = total_fragile = 0
    for c in range(chains):   # every captured chain
        if str(g["kind"]) == "hmc":
            ostep = orc.Step(f, d, kind="hmc", **kw)
            step = lmc.HamiltonianMC(tgt, d, **kw)
        else:
            _s, ostep = orc.init_nuts(f, d, seeds=seeds, **kw)
            _s2, step = lmc.init_nuts(tgt, d, random_seed=seeds, **dkw)
            np.testing.assert_array_equal(_s, _s2)
            np.testing.assert_array_equal(_s, start)
        snaps, outs = oracle_chain_snapshots(ostep, start, seeds[c], tune, draws)
        # the oracle chain IS the golden chain on the capture host; elsewhere (other BLAS) allow drift
        same = np.allclose(np.array([o["q"] for o in outs]), g["trace"][c], rtol=1e-9, atol=1e-12)
        checked, fragile = replay_iterations_on_device(step, snaps, outs, label="%s chain %d" % (name, c),
                                                       expect_plan=None if lds_plan == "auto" else lds_plan)
        total_checked += checked
        total_fragile += fragile
        if same:
            np.testing.assert_array_equal(np.array([o["stats"]["diverging"] for o in outs]),
                                          g["stat_diverging"][c, :, 0])
    return total_checked, total_fragile, chains * (tune + draws)
