"""The numpy oracle (oracle/lmc_oracle.py) pinned against fixtures captured from the imported
reference (tests/golden/capture.py). Same numpy/BLAS => bit-for-bit here; the float tolerance
(rtol 1e-9) only leaves room for a different BLAS summation order on another host."""
import os

import numpy as np
import pytest

from oracle import lmc_oracle as orc
from oracle import targets

RTOL = 1e-9
INT_STATS = ("depth", "tree_size", "diverging", "n_steps", "accepted", "tune")


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def _kw(g, prefix=""):
    kw = {}
    for n, v in zip(g[prefix + "kw_names"], g[prefix + "kw_vals"]):
        kw[str(n)] = int(v) if str(n) in ("max_treedepth", "early_max_treedepth", "max_steps") else float(v)
    return kw


def test_leapfrog_golden(golden_dir):
    g = _load(golden_dir, "leapfrog")
    for ci in range(int(g["n_cases"])):
        k = "c%d_" % ci
        d = int(g[k + "d"])
        f = targets.make(str(g[k + "family"]), d)
        var = g[k + "var"]
        if str(g[k + "pot"]) == "adapt":
            pot = orc.DiagAdaptPotential(d, np.zeros(d), var, 10)
            p0 = g[k + "p"][0].astype("float32")
        else:
            pot = orc.DiagPotential(var)
            p0 = g[k + "p"][0]
        assert str(p0.dtype) == str(g[k + "p0_dtype"])
        np.testing.assert_array_equal(pot.var, var)
        n, eps = int(g[k + "n"]), float(g[k + "eps"])
        s = orc.compute_state(pot, f, g[k + "q"][0], p0)
        states = [s]
        for _ in range(n):
            s = orc.leapfrog(pot, f, eps, s)
            states.append(s)
        for _ in range(n):
            s = orc.leapfrog(pot, f, -eps, s)
            states.append(s)
        for i, s in enumerate(states):
            for name, val in (("q", s.q), ("p", s.p), ("v", s.v), ("g", s.g)):
                np.testing.assert_allclose(val, g[k + name][i], rtol=RTOL, atol=1e-300, err_msg="%s%s[%d]" % (k, name, i))
            np.testing.assert_allclose(float(np.ravel(s.energy)[0]), g[k + "energy"][i], rtol=RTOL)
            np.testing.assert_allclose(float(np.ravel(s.logp)[0]), g[k + "logp"][i], rtol=RTOL)
        # reversibility, as the reference's own tests/test_hmc.py:23-40 demands (rtol 1e-5)
        np.testing.assert_allclose(states[-1].q, states[0].q, rtol=1e-5, atol=1e-12)
        np.testing.assert_allclose(states[-1].p, states[0].p, rtol=1e-5, atol=1e-12)


def test_transitions_golden(golden_dir):
    g = _load(golden_dir, "transitions")
    for ci in range(int(g["n_cases"])):
        k = "c%d_" % ci
        d = int(g[k + "d"])
        kind = str(g[k + "kind"])
        f = targets.make(str(g[k + "family"]), d)
        kw = _kw(g, k)
        if kind == "nuts_scaling":
            step = orc.Step(f, d, kind="nuts", scaling=np.linspace(0.5, 2.0, d), is_cov=True,
                            adapt_step_size=False, **kw)
        else:
            step = orc.Step(f, d, kind=kind, adapt_step_size=False, **kw)
        step.tune = False
        step.adapt.log_bar = np.log(float(g[k + "eps"]))
        step.adapt.log_step = np.log(float(g[k + "eps"]))
        rng = np.random.RandomState(int(g[k + "seed"]))
        q = g[k + "q0"]
        for i in range(int(g[k + "iters"])):
            q, st = step.astep(q, rng)
            np.testing.assert_allclose(q, g[k + "q"][i], rtol=RTOL, atol=1e-300, err_msg="%sq[%d]" % (k, i))
            for name in step.stats_dtypes:
                want = g[k + "stat_" + name][i]
                got = np.ravel(st[name])[0]
                if name in INT_STATS:
                    assert got == want, (k, name, i, got, want)
                else:
                    np.testing.assert_allclose(got, want, rtol=RTOL, atol=1e-12, err_msg="%s%s[%d]" % (k, name, i))
        assert rng.get_state()[2] == int(g[k + "final_rng_pos"])
        np.testing.assert_array_equal(rng.get_state()[1][:4], g[k + "final_rng_key0"])


def test_adapt_golden(golden_dir):
    g = _load(golden_dir, "adapt")
    da = orc.DualAverage(float(g["initial_step"]), 0.8, 0.05, 0.75, 10)
    for a, row in zip(g["accepts"], g["da"]):
        da.update(a, True)
        np.testing.assert_allclose([da.log_step, da.log_bar, da.hbar, da.count], row, rtol=1e-14)
    d = g["samples"].shape[1]
    pot = orc.DiagAdaptPotential(d, g["mean0"], np.ones(d), 10)
    for x, var, istd in zip(g["samples"], g["var"], g["inv_stds"]):
        pot.update(x, True)
        np.testing.assert_array_equal(pot.var, var)
        np.testing.assert_array_equal(pot.inv_stds, istd)
    assert pot.n_samples == int(g["n_samples"])
    np.testing.assert_array_equal(pot.fore.mean, g["fore_mean"])
    np.testing.assert_array_equal(pot.fore.raw_var, g["fore_raw_var"])
    np.testing.assert_array_equal(pot.back.mean, g["back_mean"])
    assert pot.fore.w_sum == float(g["fore_w"]) and pot.back.w_sum == float(g["back_w"])


def test_seed_derivation_golden(golden_dir):
    g = _load(golden_dir, "seeds")
    for chains in (2, 4, 64):
        seeds = orc.derive_seeds(20260928, chains)
        np.testing.assert_array_equal(seeds, g["seeds_%d" % chains])
        np.testing.assert_array_equal(orc.jitter_start(seeds[0], 7), g["jitter_%d" % chains])
    # prefix stability: the first K seeds do not depend on the chain count
    np.testing.assert_array_equal(g["seeds_64"][:4], g["seeds_4"])


E2E = ["e2e_hmc_c1", "e2e_nuts_std64", "e2e_nuts_std128", "e2e_nuts_ar1_16", "e2e_nuts_funnel8",
       "e2e_nuts_diag50", "e2e_nuts_normal1d", "e2e_nuts_ar1_128", "e2e_nuts_funnel256", "e2e_nuts_diag1000"]


@pytest.mark.parametrize("name", E2E)
def test_e2e_golden(golden_dir, name):
    g = _load(golden_dir, name)
    d, chains, tune, draws = int(g["d"]), int(g["chains"]), int(g["tune"]), int(g["draws"])
    kw = _kw(g)
    if "params" in g.files and str(g["family"]) == "diag_gaussian":
        f = targets.DiagGaussian(g["params"])
    else:
        f = targets.make(str(g["family"]), d)
    step = None
    if str(g["kind"]) == "hmc":
        step = orc.Step(f, d, kind="hmc", **kw)
        kw = {}
    trace, stats = orc.sample(f, d, draws=draws, tune=tune, step=step, chains=chains,
                              random_seed=int(g["random_seed"]), discard_tuned_samples=False, **kw)
    assert trace.shape == (chains, tune + draws, d) == g["trace"].shape
    for name_ in stats:
        want = g["stat_" + name_]
        assert stats[name_].shape == want.shape == (chains, tune + draws, 1)
        assert stats[name_].dtype == want.dtype
        if name_ in INT_STATS:
            np.testing.assert_array_equal(stats[name_], want, err_msg=name_)
        else:
            np.testing.assert_allclose(stats[name_], want, rtol=RTOL, atol=1e-12, err_msg=name_)
    np.testing.assert_allclose(trace, g["trace"], rtol=RTOL, atol=1e-300)


def test_diag_adapt_growing_window_golden(golden_dir):
    """QuadPotentialDiagAdapt(adaptation_window_multiplier=2) (quadpotential.py:239-243): estimator sequence and one
    end-to-end chain."""
    g = _load(golden_dir, "diag_window_multiplier")
    d = g["samples"].shape[1]
    pot = orc.DiagAdaptPotential(d, np.full(d, 0.5), np.ones(d), 10, window=15, multiplier=2)
    for i, x in enumerate(g["samples"]):
        pot.update(x, True)
        np.testing.assert_array_equal(np.asarray(pot.var, dtype="d"), g["seq_var"][i])
        assert pot.n_samples == g["seq_ns"][i] and pot.window == g["seq_window"][i]
        assert pot.fore.w_sum == g["seq_fw"][i] and pot.back.w_sum == g["seq_bw"][i]
    assert pot.window > 15
    d2, tune, draws, seed = int(g["e2e_d"]), int(g["e2e_tune"]), int(g["e2e_draws"]), int(g["e2e_seed"])
    f = targets.make("ar1", d2)
    pot2 = orc.DiagAdaptPotential(d2, g["e2e_start"], np.ones(d2), 10, window=20, multiplier=2)
    step = orc.Step(f, d2, kind="nuts", potential=pot2)
    trace, stats = orc.sample(f, d2, draws=draws, tune=tune, step=step, start=g["e2e_start"], chains=1,
                              random_seed=[seed], discard_tuned_samples=False)
    np.testing.assert_allclose(trace, g["e2e_trace"], rtol=RTOL, atol=1e-300)
    np.testing.assert_array_equal(stats["tree_size"], g["e2e_stat_tree_size"])
    assert pot2.window == int(g["e2e_final_window"])
    np.testing.assert_array_equal(np.asarray(pot2.var, dtype="d"), g["e2e_final_var"])


@pytest.mark.parametrize("kind", ["nuts", "hmc"])
def test_step_rand_golden(golden_dir, kind):
    """base_hmc.py:154-155 with step_rand = lambda s: s * np.random.uniform(lo, hi): one uniform of the chain's own stream
    per iteration, between the momentum draw and the trajectory (captured from the imported reference)."""
    g = _load(golden_dir, "e2e_step_rand")
    lo, hi = float(g["lo"]), float(g["hi"])
    d, chains = int(g[kind + "_d"]), int(g[kind + "_chains"])
    tune, draws = int(g[kind + "_tune"]), int(g[kind + "_draws"])
    f = targets.make(str(g[kind + "_family"]), d)
    if kind == "hmc":
        step = orc.Step(f, d, kind="hmc", path_length=float(g["hmc_path_length"]), step_rand=(lo, hi))
        trace, stats = orc.sample(f, d, draws=draws, tune=tune, step=step, chains=chains,
                                  random_seed=int(g["random_seed"]), discard_tuned_samples=False)
    else:
        trace, stats = orc.sample(f, d, draws=draws, tune=tune, chains=chains, random_seed=int(g["random_seed"]),
                                  discard_tuned_samples=False, step_rand=(lo, hi))
    for name_ in stats:
        want = g[kind + "_stat_" + name_]
        if name_ in INT_STATS:
            np.testing.assert_array_equal(stats[name_], want, err_msg=name_)
        else:
            np.testing.assert_allclose(stats[name_], want, rtol=RTOL, atol=1e-12, err_msg=name_)
    np.testing.assert_allclose(trace, g[kind + "_trace"], rtol=RTOL, atol=1e-300)
