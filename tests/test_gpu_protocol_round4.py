"""-m gpu: protocol methods the reference exports that used to raise NotImplementedError ("boundary
leftovers"): potential.update(sample, grad, tune) as a host call (/root/reference/littlemcmc/quadpotential.py:112,231,528),
arbitrary step_rand callables (base_hmc.py:46,123,154-155), and the reference's own tests/test_quadpotential.py:160-223
call sequences run against littlemcmc_amd unchanged."""
import os

import numpy as np
import pytest

import littlemcmc_amd as lmc
from littlemcmc_amd import quadpotential
from littlemcmc_amd import targets as T

pytestmark = pytest.mark.gpu


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def test_diag_adapt_update_as_a_host_call(golden_dir):
    """QuadPotentialDiagAdapt.update() sample by sample against the sequence captured from the reference (window 15,
    multiplier 2: three window switches in 150 samples)."""
    g = _load(golden_dir, "diag_window_multiplier")
    d = g["samples"].shape[1]
    pot = lmc.QuadPotentialDiagAdapt(d, np.full(d, 0.5), np.ones(d), 10, adaptation_window=15, adaptation_window_multiplier=2)
    for i, x in enumerate(g["samples"]):
        pot.update(x, None, True)
        np.testing.assert_array_equal(np.asarray(pot._var, dtype="d"), g["seq_var"][i], err_msg="sample %d" % i)
        assert pot._n_samples == g["seq_ns"][i] and pot.adaptation_window == g["seq_window"][i]
    n_before = pot._n_samples
    pot.update(g["samples"][0], None, False)          # tune=False: nothing happens (quadpotential.py:233-234)
    assert pot._n_samples == n_before
    lmc.QuadPotentialDiag(np.ones(3)).update(np.zeros(3), None, True)   # the fixed potentials' update is `pass`
    lmc.QuadPotentialFull(np.eye(3)).update(np.zeros(3), None, True)


# ---- /root/reference/tests/test_quadpotential.py:160-223, the call sequences as they stand there -------------------------
def test_full_adapt_sample_p(seed=4566):
    np.random.seed(seed)
    m = np.array([[3.0, -2.0], [-2.0, 4.0]])
    m_inv = np.linalg.inv(m)
    var = np.array([[2 * m[0, 0], m[1, 0] * m[1, 0] + m[1, 1] * m[0, 0]],
                    [m[0, 1] * m[0, 1] + m[1, 1] * m[0, 0], 2 * m[1, 1]]])
    n_samples = 1000
    pot = quadpotential.QuadPotentialFullAdapt(2, np.zeros(2), m_inv, 1)
    samples = [pot.random() for n in range(n_samples)]
    sample_cov = np.cov(samples, rowvar=0)
    assert np.all(np.abs(m - sample_cov) < 5 * np.sqrt(var / n_samples))


def test_full_adapt_update_window(seed=1123):
    np.random.seed(seed)
    init_cov = np.array([[1.0, 0.02], [0.02, 0.8]])
    pot = quadpotential.QuadPotentialFullAdapt(2, np.zeros(2), init_cov, 1, update_window=50)
    assert np.allclose(pot._cov, init_cov)
    for i in range(49):
        pot.update(np.random.randn(2), None, True)
    assert np.allclose(pot._cov, init_cov)
    pot.update(np.random.randn(2), None, True)
    assert not np.allclose(pot._cov, init_cov)


def test_full_adapt_adaptation_window(seed=8978):
    np.random.seed(seed)
    window = 10
    pot = quadpotential.QuadPotentialFullAdapt(2, np.zeros(2), np.eye(2), 1, adaptation_window=window)
    for i in range(window + 1):
        pot.update(np.random.randn(2), None, True)
    assert pot._previous_update == window
    assert pot._adaptation_window == window * pot._adaptation_window_multiplier


def test_full_adapt_not_invertible():
    window = 10
    pot = quadpotential.QuadPotentialFullAdapt(2, np.zeros(2), np.eye(2), 0, adaptation_window=window)
    for i in range(window + 1):
        pot.update(np.ones(2), None, True)
    with pytest.raises(ValueError):
        pot.raise_ok(None)


# ---- step_rand ---------------------------------------------------------------------------------------------------------
def test_step_rand_callable_matches_the_reference(golden_dir):
    """base_hmc.py:154-155 with an arbitrary function of the step size (here 0.9 * s): the captured reference chains through
    sample() -- a prefix, as for every tuned chain -- and every iteration through _astep."""
    g = _load(golden_dir, "e2e_step_rand_callable")
    d, fac, chains = int(g["d"]), float(g["factor"]), int(g["chains"])
    tune, draws = int(g["tune"]), int(g["draws"])
    tgt = T.AR1(d, 0.9)
    trace, stats = lmc.sample(tgt, d, draws=draws, tune=tune, chains=chains, cores=1, random_seed=int(g["random_seed"]),
                              discard_tuned_samples=False, progressbar=False, step_rand=lambda s: fac * s)
    assert trace.shape == g["trace"].shape
    n = 14
    np.testing.assert_array_equal(stats["tree_size"][:, :n], g["stat_tree_size"][:, :n])
    np.testing.assert_array_equal(stats["depth"][:, :n], g["stat_depth"][:, :n])
    np.testing.assert_allclose(trace[:, :n], g["trace"][:, :n], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(stats["step_size"][:, :n], g["stat_step_size"][:, :n], rtol=1e-6)
    # the step-method protocol: one chain driven by _astep exactly like sampling.py:481-521 drives it
    from oracle import lmc_oracle as orc
    from oracle import targets as OT

    seeds = orc.derive_seeds(int(g["random_seed"]), chains)
    start, step = lmc.init_nuts(tgt, d, random_seed=seeds, step_rand=lambda s: fac * s)
    np.random.seed(seeds[0])
    q = start
    step.tune = True
    step.reset_tuning()
    for i in range(12):
        q, st = step._astep(q)
        assert st[0]["tree_size"] == g["stat_tree_size"][0, i, 0] and st[0]["depth"] == g["stat_depth"][0, i, 0], i
        np.testing.assert_allclose(q, g["trace"][0, i], rtol=1e-7, atol=1e-9)
    assert OT is not None


def test_step_rand_random_callable_still_samples_the_target():
    """A callable that draws from np.random is honoured too (host stream, not the chain's: statistically the same sampler)."""
    d = 6
    trace, stats = lmc.sample(T.StdNormal(d), d, draws=300, tune=300, chains=32, random_seed=3, progressbar=False,
                              step_rand=lambda s: s * np.random.uniform(0.8, 1.2))
    assert abs(trace.mean()) < 0.05 and abs(trace.var() - 1.0) < 0.1
    with pytest.raises(TypeError):
        lmc.NUTS(T.StdNormal(d), d, step_rand=3.0)
