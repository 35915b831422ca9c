"""Dense mass matrices: the oracle's Full / FullInv / FullAdapt potentials pinned
against fixtures captured from the imported reference (tests/golden/capture.py dense_*). Same numpy / scipy /
OpenBLAS in this container => bit-for-bit; RTOL leaves room for another host's BLAS summation order."""
import os

import numpy as np
import pytest

from oracle import lmc_oracle as orc
from oracle import targets

RTOL = 1e-9
INT_STATS = ("depth", "tree_size", "diverging", "n_steps", "accepted", "tune")


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def test_dense_units_golden(golden_dir):
    g = _load(golden_dir, "dense_units")
    for ci in range(int(g["n_cases"])):
        k = "c%d_" % ci
        d = int(g[k + "d"])
        pot = orc.quad_potential(g[k + "matrix"], str(g[k + "kind"]) == "full")
        assert isinstance(pot, orc.FullPotential if str(g[k + "kind"]) == "full" else orc.FullInvPotential)
        f = targets.make(str(g[k + "family"]), d)
        rng = np.random.RandomState(int(g[k + "seed"]))
        draws = np.array([pot.random(rng) for _ in range(3)])
        assert str(draws.dtype) == str(g[k + "random_dtype"])
        np.testing.assert_allclose(draws, g[k + "random"], rtol=RTOL)
        q0 = 0.3 * rng.randn(d)
        p0 = pot.random(rng)
        x = rng.randn(d)
        np.testing.assert_array_equal(x, g[k + "x"])
        np.testing.assert_allclose(pot.velocity(x), g[k + "velocity"], rtol=RTOL)
        np.testing.assert_allclose(0.5 * x.dot(pot.velocity(x)), float(g[k + "energy_x"]), rtol=RTOL)
        n, eps = int(g[k + "n"]), float(g[k + "eps"])
        s = orc.compute_state(pot, f, q0, p0)
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
        np.testing.assert_allclose(states[-1].q, states[0].q, rtol=1e-5, atol=1e-9)   # tests/test_hmc.py:23-40


@pytest.mark.parametrize("name", ["w20", "w15u4"])
def test_dense_adapt_sequence_golden(golden_dir, name):
    g = _load(golden_dir, "dense_adapt")
    kw = {str(n): int(v) for n, v in zip(g[name + "_kw_names"], g[name + "_kw_vals"])}
    d = g["samples"].shape[1]
    pot = orc.FullAdaptPotential(d, g["initial_mean"], np.eye(d), 10, **kw)
    for i, x in enumerate(g["samples"]):
        pot.update(x, True)
        np.testing.assert_array_equal(np.asarray(pot.cov, dtype="d"), g[name + "_cov"][i], err_msg="cov[%d]" % i)
        np.testing.assert_allclose(np.asarray(pot.chol, dtype="d"), g[name + "_chol"][i], rtol=1e-6, atol=1e-7)
        np.testing.assert_array_equal(pot.fore.mean, g[name + "_fmean"][i])
        np.testing.assert_array_equal(pot.fore.raw, g[name + "_fraw"][i])
        np.testing.assert_array_equal(pot.back.raw, g[name + "_braw"][i])
        assert pot.fore.n_samples == g[name + "_fn"][i] and pot.back.n_samples == g[name + "_bn"][i]
        assert pot.window == g[name + "_window"][i] and pot.previous_update == g[name + "_prev"][i]
        assert pot.n_samples == g[name + "_ns"][i]
    assert pot.window > kw["adaptation_window"]          # the window did grow (tests/test_quadpotential.py:195-212)


def test_dense_adapt_singular_estimate_keeps_the_factor(golden_dir):
    g = _load(golden_dir, "dense_adapt")
    pot = orc.FullAdaptPotential(2, np.zeros(2), np.eye(2), 0, adaptation_window=10)
    with np.errstate(all="ignore"):
        for _ in range(11):
            pot.update(np.ones(2), True)
    assert bool(g["singular_failed"]) and pot.chol_error is not None   # tests/test_quadpotential.py:215-224
    np.testing.assert_array_equal(np.asarray(pot.cov, dtype="d"), g["singular_cov"])
    np.testing.assert_array_equal(np.asarray(pot.chol, dtype="d"), g["singular_chol"])


DENSE_E2E = ["e2e_nuts_full_ar1_12", "e2e_nuts_fullinv_ar1_12", "e2e_hmc_full_std10",
             "e2e_nuts_adaptfull_ar1_10_a", "e2e_nuts_adaptfull_ar1_10_b", "e2e_nuts_adaptfull_std70",
             "e2e_nuts_full64_ar1_12"]


def dense_run_from_golden(g):
    """(f, step, sample kwargs) reproducing the captured reference call with the oracle."""
    d = int(g["d"])
    f = targets.make(str(g["family"]), d)
    potk = str(g["potential"])
    if potk in ("full", "inv", "full64"):
        pot = orc.FullPotential(g["matrix"], dtype="float64") if potk == "full64" else orc.quad_potential(g["matrix"], potk == "full")
        step = orc.Step(f, d, kind=str(g["kind"]), potential=pot)
        return f, step, dict(random_seed=int(g["random_seed"]))
    return f, None, dict(random_seed=[int(g["seeds"][0])], init=potk)


@pytest.mark.parametrize("name", DENSE_E2E)
def test_dense_e2e_golden(golden_dir, name):
    g = _load(golden_dir, name)
    d, chains, tune, draws = int(g["d"]), int(g["chains"]), int(g["tune"]), int(g["draws"])
    f, step, kw = dense_run_from_golden(g)
    trace, stats = orc.sample(f, d, draws=draws, tune=tune, step=step, chains=chains,
                              discard_tuned_samples=False, **kw)
    for name_ in stats:
        want = g["stat_" + name_]
        assert stats[name_].shape == want.shape and stats[name_].dtype == want.dtype
        if name_ in INT_STATS:
            np.testing.assert_array_equal(stats[name_], want, err_msg=name_)
        else:
            np.testing.assert_allclose(stats[name_], want, rtol=RTOL, atol=1e-12, err_msg=name_)
    np.testing.assert_allclose(trace, g["trace"], rtol=RTOL, atol=1e-300)


def test_full_potential_float64_units(golden_dir):
    """QuadPotentialFull(cov, dtype="float64") (quadpotential.py:431-444): velocity, energy, random() of the oracle against
    the values captured from the reference."""
    g = _load(golden_dir, "e2e_nuts_full64_ar1_12")
    pot = orc.FullPotential(g["matrix"], dtype="float64")
    x = g["unit_x"]
    np.testing.assert_allclose(pot.velocity(x), g["unit_velocity"], rtol=RTOL)
    np.testing.assert_allclose(0.5 * x.dot(pot.velocity(x)), float(g["unit_energy"]), rtol=RTOL)
    rng = np.random.RandomState(int(g["unit_random_seed"]))
    draws = np.array([pot.random(rng) for _ in range(3)])
    assert str(draws.dtype) == str(g["unit_random_dtype"]) == "float64"
    np.testing.assert_allclose(draws, g["unit_random"], rtol=RTOL)


def test_full_adapt_two_chains_sequential_carry_over(golden_dir):
    """init="adapt_full", chains=2, cores=1 (missing item 4). The reference's sequential driver reuses ONE
    step object (/root/reference/littlemcmc/sampling.py:370-383) and QuadPotentialFullAdapt.reset() is the inherited no-op
    (quadpotential.py:137-139): chain 1 starts from the covariance, estimators and grown window chain 0 ended with. The
    oracle reproduces that run bit for bit when told to (sequential_carry_over=True) and, by default, gives every chain
    a fresh potential -- in which case chain 1 is the reference's ONE-chain run with chain 1's seed (what the reference's
    multi-process driver computes, and what the device computes: tests/test_gpu_dense.py pins the device side)."""
    g = _load(golden_dir, "e2e_adaptfull_two_chains")
    d, tune, draws = int(g["d"]), int(g["tune"]), int(g["draws"])
    f = targets.make(str(g["family"]), d)
    seeds = [int(s) for s in g["seeds"]]
    with np.errstate(all="ignore"):
        carried, cstats = orc.sample(f, d, draws=draws, tune=tune, chains=2, init="adapt_full", random_seed=seeds,
                                     discard_tuned_samples=False, sequential_carry_over=True)
        fresh, fstats = orc.sample(f, d, draws=draws, tune=tune, chains=2, init="adapt_full", random_seed=seeds,
                                   discard_tuned_samples=False)
    np.testing.assert_allclose(carried, g["trace"], rtol=RTOL, atol=1e-300)
    for name_ in cstats:
        if name_ in INT_STATS:
            np.testing.assert_array_equal(cstats[name_], g["stat_" + name_], err_msg=name_)
    # chain 0 is the same either way; chain 1 fresh == the reference's one-chain run with chain 1's seed
    np.testing.assert_allclose(fresh[0], g["trace"][0], rtol=RTOL, atol=1e-300)
    np.testing.assert_allclose(fresh[1], g["solo1_trace"][0], rtol=RTOL, atol=1e-300)
    for name_ in fstats:
        if name_ in INT_STATS:
            np.testing.assert_array_equal(fstats[name_][1], g["solo1_stat_" + name_][0], err_msg=name_)
    # ... and the two behaviours do differ, from chain 1's first iteration on (another starting matrix)
    assert not np.allclose(g["trace"][1, :5], g["solo1_trace"][0, :5])
