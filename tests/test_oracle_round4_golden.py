"""The oracle pinned against the round-4 fixtures captured from the imported reference (tests/golden/capture.py groups
diag_float64, full_adapt_float64, step_rand_callable, wide): QuadPotentialDiagAdapt / QuadPotentialFullAdapt with
dtype="float64", a deterministic step_rand callable,
and shapes beyond the fused kernels' vector widths (model_ndim 2000 diagonal, 384 dense)."""
import os

import numpy as np

from oracle import lmc_oracle as orc
from oracle import targets

RTOL = 1e-9
INT_STATS = ("depth", "tree_size", "diverging", "n_steps", "accepted", "tune")


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def _check(stats, trace, g):
    for name_ in stats:
        want = g["stat_" + name_]
        assert stats[name_].shape == want.shape and stats[name_].dtype == want.dtype
        if name_ in INT_STATS:
            np.testing.assert_array_equal(stats[name_], want, err_msg=name_)
        else:
            np.testing.assert_allclose(stats[name_], want, rtol=RTOL, atol=1e-12, err_msg=name_)
    np.testing.assert_allclose(trace, g["trace"], rtol=RTOL, atol=1e-300)


def ar1_cov(d, rho):
    idx = np.arange(d)
    return rho ** np.abs(idx[:, None] - idx[None, :])


def test_diag_adapt_float64_golden(golden_dir):
    """quadpotential.py:159,175-184: dtype="float64" keeps var / stds / inv_stds, the momentum draw and with it the
    start state's velocity and kinetic energy in float64."""
    g = _load(golden_dir, "e2e_nuts_diag64_ar1_12")
    d = int(g["d"])
    pot = orc.DiagAdaptPotential(d, g["unit_initial_mean"], g["unit_initial_diag"], 10, window=20, dtype="float64")
    x = g["unit_x"]
    np.testing.assert_array_equal(pot.velocity(x), g["unit_velocity"])
    rs = np.random.RandomState(int(g["unit_random_seed"]))
    rnd = np.array([pot.random(rs) for _ in range(3)])
    assert str(rnd.dtype) == str(g["unit_random_dtype"]) == "float64"
    np.testing.assert_array_equal(rnd, g["unit_random"])
    for i, smp in enumerate(g["seq_samples"]):
        pot.update(smp, True)
        assert str(pot.var.dtype) == str(g["seq_var_dtype"]) == "float64"
        np.testing.assert_array_equal(pot.var, g["seq_var"][i])
    assert pot.n_samples == int(g["seq_n_samples"])
    f = targets.make(str(g["family"]), d)
    pot2 = orc.DiagAdaptPotential(d, g["start"], np.ones(d), 10, dtype="float64")
    step = orc.Step(f, d, kind="nuts", potential=pot2)
    trace, stats = orc.sample(f, d, draws=int(g["draws"]), tune=int(g["tune"]), step=step, start=g["start"],
                              chains=int(g["chains"]), random_seed=[int(s) for s in g["seeds"]], discard_tuned_samples=False)
    _check(stats, trace, g)
    np.testing.assert_array_equal(pot2.var, g["final_var"])


def test_full_adapt_float64_golden(golden_dir):
    """quadpotential.py:484,497-509: QuadPotentialFullAdapt(dtype="float64") keeps covariance, factor and the momentum draw in
    float64 (capture.py group full_adapt_float64)."""
    g = _load(golden_dir, "e2e_nuts_adaptfull64_ar1_10")
    d = int(g["d"])
    pot = orc.FullAdaptPotential(d, g["unit_initial_mean"], g["unit_initial_cov"], 10, adaptation_window=20, dtype="float64")
    assert str(pot.chol.dtype) == str(g["unit_chol_dtype"]) == "float64" and not pot.momentum_f32
    np.testing.assert_array_equal(pot.chol, g["unit_chol"])
    np.testing.assert_array_equal(pot.velocity(g["unit_x"]), g["unit_velocity"])
    rs = np.random.RandomState(int(g["unit_random_seed"]))
    rnd = np.array([pot.random(rs) for _ in range(3)])
    assert str(rnd.dtype) == str(g["unit_random_dtype"]) == "float64"
    np.testing.assert_array_equal(rnd, g["unit_random"])
    for i, smp in enumerate(g["seq_samples"]):
        pot.update(smp, True)
        assert str(pot.cov.dtype) == str(g["seq_cov_dtype"]) == "float64"
        np.testing.assert_array_equal(pot.cov, g["seq_cov"][i])
        np.testing.assert_array_equal(pot.chol, g["seq_chol"][i])
        assert pot.fore.n_samples == g["seq_fn"][i] and pot.back.n_samples == g["seq_bn"][i]
        assert pot.window == g["seq_window"][i] and pot.previous_update == g["seq_prev"][i]
    assert pot.n_samples == int(g["seq_n_samples"])
    f = targets.make(str(g["family"]), d)
    pot2 = orc.FullAdaptPotential(d, g["start"], np.eye(d), 10, dtype="float64")
    step = orc.Step(f, d, kind="nuts", potential=pot2)
    trace, stats = orc.sample(f, d, draws=int(g["draws"]), tune=int(g["tune"]), step=step, start=g["start"],
                              chains=1, random_seed=[int(s) for s in g["seeds"]], discard_tuned_samples=False)
    _check(stats, trace, g)
    np.testing.assert_array_equal(pot2.cov, g["final_cov"])
    np.testing.assert_array_equal(pot2.chol, g["final_chol"])


def test_step_rand_callable_golden(golden_dir):
    """base_hmc.py:154-155 with an arbitrary (here deterministic) function of the step size."""
    g = _load(golden_dir, "e2e_step_rand_callable")
    d, fac = int(g["d"]), float(g["factor"])
    f = targets.make(str(g["family"]), d)
    trace, stats = orc.sample(f, d, draws=int(g["draws"]), tune=int(g["tune"]), chains=int(g["chains"]),
                              random_seed=int(g["random_seed"]), discard_tuned_samples=False, step_rand=lambda s: fac * s)
    _check(stats, trace, g)


def test_model_ndim_2000_golden(golden_dir):
    g = _load(golden_dir, "e2e_nuts_diag2000")
    d = int(g["d"])
    f = targets.DiagGaussian(g["params"])
    trace, stats = orc.sample(f, d, draws=int(g["draws"]), tune=int(g["tune"]), chains=1,
                              random_seed=int(g["random_seed"]), discard_tuned_samples=False)
    _check(stats, trace, g)


def test_dense_model_ndim_384_golden(golden_dir):
    g = _load(golden_dir, "e2e_nuts_full_ar1_384")
    d = int(g["d"])
    f = targets.make("ar1", d)
    step = orc.Step(f, d, kind="nuts", potential=orc.FullPotential(ar1_cov(d, float(g["rho"]))))
    trace, stats = orc.sample(f, d, draws=int(g["draws"]), tune=int(g["tune"]), step=step, chains=1,
                              random_seed=int(g["random_seed"]), discard_tuned_samples=False)
    _check(stats, trace, g)
