"""-m gpu: the general ("wide") kernels (csrc/lmc_wide.hpp) -- every shape of the hot path the fused kernels are not
instantiated for: model_ndim > 1024 (/root/reference/littlemcmc/base_hmc.py:102 has no limit), QuadPotentialFull / FullInv
beyond 256 dimensions (quadpotential.py:388-468), QuadPotentialDiagAdapt(dtype="float64") (quadpotential.py:159,175-184),
a run-time compiled density with a dense mass matrix. Parity exactly as for the fused kernels: every iteration of a chain
of the reference (captured goldens) / of the oracle replayed from the oracle's pre-iteration state, integer statistics
exact. LMC_FORCE_WIDE=1 sends the small goldens through the general kernels as well."""
import os

import numpy as np
import pytest

import littlemcmc_amd as lmc
from littlemcmc_amd import targets as T
from oracle import lmc_oracle as orc
from oracle import targets as OT
from tests._gpu_util import device_target, kwargs_from, oracle_chain_snapshots, replay_iterations_on_device

pytestmark = pytest.mark.gpu


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def ar1_cov(d, rho):
    idx = np.arange(d)
    return rho ** np.abs(idx[:, None] - idx[None, :])


def test_model_ndim_2000_every_iteration_of_the_reference_chain(golden_dir):
    """d = 2000 diagonal-mass NUTS (twice C4's dimension): the captured reference chain, every iteration."""
    g = _load(golden_dir, "e2e_nuts_diag2000")
    d, tune, draws = int(g["d"]), int(g["tune"]), int(g["draws"])
    f = OT.DiagGaussian(g["params"])
    tgt = T.DiagGaussian(g["params"])
    seeds = [int(s) for s in g["seeds"]]
    _s, ostep = orc.init_nuts(f, d, seeds=seeds)
    start, step = lmc.init_nuts(tgt, d, random_seed=seeds)
    np.testing.assert_array_equal(start, g["start"])
    eng = step._make_engine(2)
    try:
        assert eng.wide and eng.kernel_shape() == (2, 2, 16)
    finally:
        eng.close()
    snaps, outs = oracle_chain_snapshots(ostep, start, seeds[0], tune, draws)
    np.testing.assert_allclose(np.array([o["q"] for o in outs]), g["trace"][0], rtol=1e-9, atol=1e-300)   # oracle == reference
    checked, fragile = replay_iterations_on_device(step, snaps, outs, label="diag2000")
    print("d = 2000: %d of %d iterations replayed, %d fragile" % (checked, tune + draws, fragile))
    assert checked >= tune + draws - 1


@pytest.mark.parametrize("family,d,kw", [("std_normal", 1025, {}), ("ar1", 1500, {}), ("funnel", 2100, {"max_treedepth": 8}),
                                           ("ar1", 4100, {}), ("std_normal", 9000, {})])
def test_every_iteration_replay_beyond_1024_dimensions(family, d, kw):
    """2, 4, 8 and 16 elements per thread of the 1024-thread team (d <= 2048 / 4096 / 8192 / 16 384)."""
    f = OT.make(family, d)
    tgt = device_target(family, d, f.params())
    seeds = orc.derive_seeds(991 + d, 2)
    tune, draws = (14, 4) if d > 4096 else (24, 6)
    _s, ostep = orc.init_nuts(f, d, seeds=seeds, **kw)
    start, step = lmc.init_nuts(tgt, d, random_seed=seeds, **kw)
    np.testing.assert_array_equal(_s, start)
    snaps, outs = oracle_chain_snapshots(ostep, start, seeds[1], tune, draws)
    checked, fragile = replay_iterations_on_device(step, snaps, outs, label="%s d=%d" % (family, d))
    assert checked >= tune + draws - 1, (checked, fragile)


def test_sample_api_beyond_1024_dimensions():
    """sample() end to end at d = 1300: shapes, dtypes, a same-seed prefix of the oracle's chains, moments of the target."""
    d, chains, tune, draws = 1300, 6, 60, 40
    trace, stats = lmc.sample(T.StdNormal(d), d, draws=draws, tune=tune, chains=chains, random_seed=8, progressbar=False,
                              discard_tuned_samples=False)
    assert trace.shape == (chains, tune + draws, d) and stats["tree_size"].dtype == np.float64
    seeds = orc.derive_seeds(8, chains)
    ot, os_ = orc.sample(OT.StdNormal(d), d, draws=0, tune=12, chains=2, random_seed=seeds[:2],
                         start=orc.jitter_start(seeds[0], d), step=orc.init_nuts(OT.StdNormal(d), d, seeds=seeds)[1],
                         discard_tuned_samples=False)
    n = 10
    np.testing.assert_array_equal(stats["tree_size"][:2, :n], os_["tree_size"][:, :n])
    np.testing.assert_allclose(trace[:2, :n], ot[:, :n], rtol=1e-7, atol=1e-9)
    post = trace[:, tune:]
    assert abs(post.mean()) < 0.02 and abs(post.var() - 1.0) < 0.05
    # HMC takes the same kernel
    tr2, st2 = lmc.sample(T.StdNormal(d), d, draws=10, tune=30, chains=3, random_seed=8, progressbar=False,
                          step=lmc.HamiltonianMC(T.StdNormal(d), d, path_length=1.0))
    assert tr2.shape == (3, 10, d) and st2["accepted"].dtype == np.bool_ and np.isfinite(tr2).all()


def test_dense_model_ndim_384_every_iteration_of_the_reference_chain(golden_dir):
    """QuadPotentialFull at d = 384 (quadpotential.py:430-468): the captured reference chain through the general kernel."""
    from tests.test_gpu_dense import REPLAY_F32, _replay

    g = _load(golden_dir, "e2e_nuts_full_ar1_384")
    d, tune, draws = int(g["d"]), int(g["tune"]), int(g["draws"])
    cov = ar1_cov(d, float(g["rho"]))
    of = OT.make("ar1", d)
    ostep = orc.Step(of, d, kind="nuts", potential=orc.FullPotential(cov))
    dstep = lmc.NUTS(T.AR1(d, 0.9), d, potential=lmc.QuadPotentialFull(cov))
    seed = int(g["seeds"][0])
    start = orc.jitter_start(seed, d)      # sample() without start: init_nuts' jitter (sampling.py:148-159)
    n = _replay(ostep, dstep, start, seed, tune, draws, True, "full d=384")
    print("dense d = 384: %d of %d iterations replayed (tolerance %g: float32-born momentum)" % (n, tune + draws, REPLAY_F32))
    assert n >= tune + draws - 3


@pytest.mark.parametrize("d,kind", [(300, "full"), (520, "inv"), (1100, "full64")])
def test_dense_fixed_matrices_beyond_256_dimensions(d, kind):
    from tests.test_gpu_dense import _replay

    rs = np.random.RandomState(d)
    a = rs.randn(d, d) / np.sqrt(d)
    mat = a @ a.T + 0.5 * np.eye(d)
    of = OT.make("ar1", d)
    tgt = device_target("ar1", d, of.params())
    if kind == "full64":
        opot, dpot = orc.FullPotential(mat, dtype="float64"), lmc.QuadPotentialFull(mat, dtype="float64")
    else:
        opot = orc.quad_potential(mat, kind == "full")
        dpot = lmc.QuadPotentialFull(mat) if kind == "full" else lmc.QuadPotentialFullInv(mat)
    ostep = orc.Step(of, d, kind="nuts", potential=opot)
    dstep = lmc.NUTS(tgt, d, potential=dpot)
    start = 0.1 * rs.randn(d)
    tune, draws = (6, 2) if d > 1000 else (10, 4)
    n = _replay(ostep, dstep, start, 4242 + d, tune, draws, kind == "full", "%s d=%d" % (kind, d))
    assert n >= tune + draws - 2


def test_diag_adapt_float64_matches_the_reference(golden_dir):
    """QuadPotentialDiagAdapt(dtype="float64"): protocol values, the update() sequence across a window switch and every
    iteration of the captured reference run."""
    g = _load(golden_dir, "e2e_nuts_diag64_ar1_12")
    d = int(g["d"])
    pot = lmc.QuadPotentialDiagAdapt(d, g["unit_initial_mean"], g["unit_initial_diag"], 10, adaptation_window=20, dtype="float64")
    x = g["unit_x"]
    np.testing.assert_allclose(pot.velocity(x), g["unit_velocity"], rtol=1e-15)
    np.testing.assert_allclose(pot.energy(x), float(g["unit_energy"]), rtol=1e-13)
    np.random.seed(int(g["unit_random_seed"]))
    rnd = np.array([pot.random() for _ in range(3)])
    assert str(rnd.dtype) == str(g["unit_random_dtype"]) == "float64"
    np.testing.assert_allclose(rnd, g["unit_random"], rtol=5e-16, atol=1e-300)
    for i, smp in enumerate(g["seq_samples"]):
        pot.update(smp, None, True)
        assert str(pot._var.dtype) == "float64"
        np.testing.assert_allclose(pot._var, g["seq_var"][i], rtol=1e-14, err_msg="update %d" % i)
    assert pot._n_samples == int(g["seq_n_samples"])
    # the captured run, iteration by iteration
    f = OT.make(str(g["family"]), d)
    tgt = device_target(str(g["family"]), d, g["params"])
    tune, draws = int(g["tune"]), int(g["draws"])
    total = 0
    for c in range(int(g["chains"])):
        opot = orc.DiagAdaptPotential(d, g["start"], np.ones(d), 10, dtype="float64")
        ostep = orc.Step(f, d, kind="nuts", potential=opot)
        step = lmc.NUTS(tgt, d, potential=lmc.QuadPotentialDiagAdapt(d, g["start"], np.ones(d), 10, dtype="float64"))
        snaps, outs = oracle_chain_snapshots(ostep, g["start"], int(g["seeds"][c]), tune, draws)
        np.testing.assert_allclose(np.array([o["q"] for o in outs]), g["trace"][c], rtol=1e-9, atol=1e-300)
        checked, fragile = replay_iterations_on_device(step, snaps, outs, label="diag64 chain %d" % c)
        total += checked
    assert total >= int(g["chains"]) * (tune + draws) - 2
    # ... and through sample(): the state the step object is left with carries the potential's dtype
    step = lmc.NUTS(tgt, d, potential=lmc.QuadPotentialDiagAdapt(d, g["start"], np.ones(d), 10, dtype="float64"))
    trace, stats = lmc.sample(tgt, d, draws=draws, tune=tune, step=step, start=g["start"], chains=int(g["chains"]),
                              random_seed=[int(s) for s in g["seeds"]], discard_tuned_samples=False, progressbar=False)
    n = 12
    np.testing.assert_array_equal(stats["tree_size"][:, :n], g["stat_tree_size"][:, :n])
    np.testing.assert_allclose(trace[:, :n], g["trace"][:, :n], rtol=1e-7, atol=1e-9)
    assert step.potential._var.dtype == np.float64 and str(g["final_var_dtype"]) == "float64"


def test_full_adapt_float64_matches_the_reference(golden_dir):
    """QuadPotentialFullAdapt(dtype="float64") (quadpotential.py:484,497-509): protocol values, the update() sequence across a
    window switch -- covariance exact, factor to float64 rounding -- and every iteration of the captured reference run, with
    estimators, covariance and factor compared after each tuning iteration."""
    from tests.test_gpu_dense import _replay

    g = _load(golden_dir, "e2e_nuts_adaptfull64_ar1_10")
    d = int(g["d"])
    pot = lmc.QuadPotentialFullAdapt(d, g["unit_initial_mean"], g["unit_initial_cov"], 10, adaptation_window=20, dtype="float64")
    x = g["unit_x"]
    np.testing.assert_allclose(pot.velocity(x), g["unit_velocity"], rtol=1e-14)
    np.testing.assert_allclose(pot.energy(x), float(g["unit_energy"]), rtol=1e-13)
    np.random.seed(int(g["unit_random_seed"]))
    rnd = np.array([pot.random() for _ in range(3)])
    assert str(rnd.dtype) == str(g["unit_random_dtype"]) == "float64"
    np.testing.assert_allclose(rnd, g["unit_random"], rtol=1e-12, atol=1e-14)
    assert pot._chol.dtype == np.float64
    np.testing.assert_allclose(pot._chol, g["unit_chol"], rtol=1e-13, atol=1e-15)
    for i, smp in enumerate(g["seq_samples"]):
        pot.update(smp, None, True)
        assert pot._cov.dtype == np.float64 and pot._chol.dtype == np.float64
        np.testing.assert_array_equal(pot._cov, g["seq_cov"][i], err_msg="update %d" % i)
        np.testing.assert_allclose(pot._chol, g["seq_chol"][i], rtol=1e-12, atol=1e-14, err_msg="update %d" % i)
        assert pot._adaptation_window == int(g["seq_window"][i]) and pot._previous_update == int(g["seq_prev"][i])
    assert pot._n_samples == int(g["seq_n_samples"])
    # the captured run, iteration by iteration (float64 tolerances: nothing on this path is float32-born)
    f = OT.make(str(g["family"]), d)
    tgt = device_target(str(g["family"]), d, g["params"])
    tune, draws = int(g["tune"]), int(g["draws"])
    ostep = orc.Step(f, d, kind="nuts", potential=orc.FullAdaptPotential(d, g["start"], np.eye(d), 10, dtype="float64"))
    dstep = lmc.NUTS(tgt, d, potential=lmc.QuadPotentialFullAdapt(d, g["start"], np.eye(d), 10, dtype="float64"))
    eng = dstep._make_engine(1)
    try:
        assert eng.wide and eng.mass_f64
    finally:
        eng.close()
    assert _replay(ostep, dstep, g["start"], int(g["seeds"][0]), tune, draws, False, "adapt_full float64") >= tune + draws - 2
    # ... and through sample(): the potential the step object is left with carries the dtype
    dstep = lmc.NUTS(tgt, d, potential=lmc.QuadPotentialFullAdapt(d, g["start"], np.eye(d), 10, dtype="float64"))
    trace, stats = lmc.sample(tgt, d, draws=draws, tune=tune, step=dstep, start=g["start"], chains=1,
                              random_seed=[int(s) for s in g["seeds"]], discard_tuned_samples=False, progressbar=False)
    n = 12
    np.testing.assert_array_equal(stats["tree_size"][:, :n], g["stat_tree_size"][:, :n])
    np.testing.assert_allclose(trace[:, :n], g["trace"][:, :n], rtol=1e-7, atol=1e-9)
    assert dstep.potential._cov.dtype == np.float64 and dstep.potential._chol.dtype == np.float64


def test_full_adapt_float64_at_300_dimensions():
    """The float64 form beyond the fused kernels' sizes too: an oracle chain at d = 300, every iteration."""
    from tests.test_gpu_dense import _replay

    d, seed = 300, 4321
    f = OT.make("ar1", d)
    tgt = device_target("ar1", d, f.params())
    start = orc.jitter_start(seed, d)
    ostep = orc.Step(f, d, kind="nuts", potential=orc.FullAdaptPotential(d, start, np.eye(d), 10, dtype="float64"))
    dstep = lmc.NUTS(tgt, d, potential=lmc.QuadPotentialFullAdapt(d, start, np.eye(d), 10, dtype="float64"))
    assert _replay(ostep, dstep, start, seed, 10, 2, False, "adapt_full float64 d=300") >= 10


def test_fixed_diagonal_in_float64():
    """QuadPotentialDiag(v, dtype="float64") (quadpotential.py:349-365): the diagonal is not rounded to float32."""
    d = 9
    v = np.linspace(0.3, 2.9, d) + 1e-9                     # not float32-representable
    pot = lmc.QuadPotentialDiag(v, dtype="float64")
    x = np.arange(1.0, d + 1)
    np.testing.assert_allclose(pot.velocity(x), v * x, rtol=1e-15)
    pot32 = lmc.QuadPotentialDiag(v)
    np.testing.assert_allclose(pot32.velocity(x), v.astype("float32") * x, rtol=1e-15)
    assert np.abs(pot32.velocity(x) - v * x).max() > 1e-9    # the float32 default does round (as the reference's)


GOLDENS_THROUGH_WIDE = ["e2e_hmc_c1", "e2e_nuts_std64", "e2e_nuts_ar1_16", "e2e_nuts_funnel8", "e2e_nuts_diag50", "e2e_nuts_normal1d"]


@pytest.mark.parametrize("team", [1, 16])
@pytest.mark.parametrize("name", GOLDENS_THROUGH_WIDE)
def test_small_goldens_replay_through_the_general_kernel(golden_dir, name, team, monkeypatch):
    """LMC_FORCE_WIDE=1: the captured reference chains of the fused kernels' own shapes -- window switches, divergences,
    HMC, d = 1 -- every iteration through the general kernel, as one wavefront per chain (the shape dim <= 1024 takes) and as
    the 16-wavefront team (LMC_WIDE_TEAM=16: the shape dim > 1024 takes)."""
    monkeypatch.setenv("LMC_FORCE_WIDE", "1")
    monkeypatch.setenv("LMC_WIDE_TEAM", str(team))
    g = _load(golden_dir, name)
    d, chains, tune, draws = int(g["d"]), int(g["chains"]), int(g["tune"]), int(g["draws"])
    kw = kwargs_from(g)
    fam = str(g["family"])
    f = OT.DiagGaussian(g["params"]) if fam == "diag_gaussian" else OT.make(fam, d)
    tgt = device_target(fam, d, g["params"])
    seeds = [int(s) for s in g["seeds"]]
    total = 0
    for c in range(min(chains, 2)):
        if str(g["kind"]) == "hmc":
            ostep, step = orc.Step(f, d, kind="hmc", **kw), lmc.HamiltonianMC(tgt, d, **kw)
        else:
            _s, ostep = orc.init_nuts(f, d, seeds=seeds, **kw)
            _s2, step = lmc.init_nuts(tgt, d, random_seed=seeds, **kw)
        eng = step._make_engine(1)
        try:
            assert eng.wide and eng.kernel_shape()[2] == team
        finally:
            eng.close()
        snaps, outs = oracle_chain_snapshots(ostep, g["start"], seeds[c], tune, draws)
        checked, fragile = replay_iterations_on_device(step, snaps, outs, label="%s (general kernel) chain %d" % (name, c))
        total += checked
    assert total >= min(chains, 2) * (tune + draws) - 2


@pytest.mark.parametrize("team", [1, 16])
@pytest.mark.parametrize("name", ["e2e_nuts_full_ar1_12", "e2e_nuts_fullinv_ar1_12", "e2e_nuts_full64_ar1_12", "e2e_hmc_full_std10"])
def test_dense_goldens_replay_through_the_general_kernel(golden_dir, name, team, monkeypatch):
    from tests.test_gpu_dense import _oracle_and_device_steps, _replay

    monkeypatch.setenv("LMC_FORCE_WIDE", "1")
    monkeypatch.setenv("LMC_WIDE_TEAM", str(team))
    g = _load(golden_dir, name)
    ostep, dstep, start = _oracle_and_device_steps(g)
    f32_born = str(g["potential"]) not in ("inv", "full64")
    tune, draws = int(g["tune"]), int(g["draws"])
    n = _replay(ostep, dstep, start, int(g["seeds"][0]), tune, draws, f32_born, name + " (general kernel)")
    assert n >= 0.97 * (tune + draws)


# ---------------------------------------------------------------------------------------------------
# QuadPotentialFullAdapt beyond the fused kernels' 256 dimensions (quadpotential.py:470-560): the general sampling kernel on
# per-chain matrices, the refresh factorised through HBM (csrc/lmc_dense.hpp: cholesky_hbm)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("d,family", [(300, "ar1"), (520, "ar1"), (600, "std_normal"), (1024, "std_normal")])
def test_full_adapt_beyond_256_dimensions_replays_the_oracle(d, family):
    """init='jitter+adapt_full' at d = 300 (one wavefront per chain) and d = 520 (the 16-wavefront team): every iteration from
    the oracle's pre-iteration state -- estimators, covariance, factor, window bookkeeping after each tuning iteration."""
    from tests.test_gpu_dense import _replay

    of = OT.make(family, d)
    tgt = device_target(family, d, of.params())
    seed = 900 + d
    start, ostep = orc.init_nuts(of, d, init="jitter+adapt_full", seeds=[seed])
    start_d, dstep = lmc.init_nuts(tgt, d, init="jitter+adapt_full", random_seed=[seed])
    np.testing.assert_array_equal(start, start_d)
    eng = dstep._make_engine(1)
    try:
        assert eng.wide and eng.kernel_shape()[2] == (1 if d <= 512 else 16)
    finally:
        eng.close()
    tune, draws = (12, 3) if d < 1024 else (5, 1)      # (1024: the largest size the potential accepts)
    assert _replay(ostep, dstep, start, seed, tune, draws, True, "adapt_full d=%d" % d) >= tune + draws - 2


@pytest.mark.parametrize("name", ["e2e_nuts_adaptfull_ar1_10_a", "e2e_nuts_adaptfull_ar1_10_b", "e2e_nuts_adaptfull_std70"])
def test_full_adapt_goldens_replay_through_the_general_kernel_and_the_hbm_factorisation(golden_dir, name, monkeypatch):
    """LMC_FORCE_WIDE=1 + LMC_CHOL_HBM=1: the reference's own FullAdapt chains (window switches included) through the general
    sampling kernel and the factorisation that serves d > 256 -- same tolerances as the register form's test."""
    from tests.test_gpu_dense import test_dense_transitions_replay_the_reference_chain as replay_reference_chain

    monkeypatch.setenv("LMC_FORCE_WIDE", "1")
    monkeypatch.setenv("LMC_CHOL_HBM", "1")
    replay_reference_chain(golden_dir, name)


def test_hbm_factorisation_equals_the_register_form(golden_dir, monkeypatch):
    """The two factorisations apply the same operation sequence to every entry: covariance and factor after each update of
    the reference's update sequence (tests/test_quadpotential.py:183-224) are equal bit for bit, and a d = 200 estimate too."""
    from tests.test_gpu_dense import test_full_adapt_update_sequence_matches_reference as update_sequence

    def factors(d, n_updates, hbm):
        if hbm:
            monkeypatch.setenv("LMC_CHOL_HBM", "1")
        else:
            monkeypatch.delenv("LMC_CHOL_HBM", raising=False)
        rs = np.random.RandomState(7)
        a = rs.randn(d, d) / np.sqrt(d)
        root = np.linalg.cholesky(a @ a.T + 0.3 * np.eye(d))
        out = []
        with lmc.Engine(T.StdNormal(d), chains=2, potential="full_adapt") as eng:
            eng.set_dense_potential(np.eye(d), np.zeros(d), 1, 50, 2.0, 1)
            for i in range(n_updates):
                eng.set_position((root @ rs.randn(d, 2)).T.copy())
                eng.dense_update(True)
            st = eng.get_dense_state()
            assert (st["chol_failures"] == 0).all()
            out = [st["cov"].copy(), st["chol"].copy()]
        return out

    for d in (24, 200):
        cov_r, chol_r = factors(d, 12, False)
        cov_h, chol_h = factors(d, 12, True)
        np.testing.assert_array_equal(cov_r, cov_h)
        np.testing.assert_array_equal(chol_r, chol_h)
        assert np.abs(chol_r[0] - np.eye(d)).max() > 1e-3      # (it did learn something)
    monkeypatch.setenv("LMC_CHOL_HBM", "1")
    for name in ("w20", "w15u4"):
        update_sequence(golden_dir, name)


def test_full_adapt_sample_at_300_dimensions_learns_the_covariance():
    """sample() with a QuadPotentialFullAdapt at d = 300 end to end (window 600: the estimate that takes over at the switch has
    more samples than dimensions -- with fewer the reference itself stops on a singular matrix): the adapted matrices approach
    the target's covariance and the trees get shallower than with the diagonal metric (tests/test_gpu_dense.py at d = 8)."""
    d, chains, tune, draws = 300, 16, 1000, 50
    tgt = T.AR1(d, 0.9)
    pot = lmc.QuadPotentialFullAdapt(d, np.zeros(d), np.eye(d), 10, adaptation_window=600)
    trace, stats, eng = lmc.sample(tgt, d, draws=draws, tune=tune, chains=chains, step=lmc.NUTS(tgt, d, potential=pot),
                                   random_seed=5, return_engine=True, progressbar=False)
    try:
        assert eng.wide
        st = eng.get_dense_state(fields=("cov", "chol_failures"))
    finally:
        eng.close()
    assert (st["chol_failures"] == 0).all()
    assert trace.shape == (chains, draws, d) and np.isfinite(trace).all()
    cov = st["cov"].mean(axis=0)
    sd = np.sqrt(np.diag(cov))
    assert sd.min() > 0.5 and sd.max() < 1.3      # (the estimate still holds the warm-up's first, narrow samples)
    corr = cov / np.outer(sd, sd)
    near = np.abs(np.subtract.outer(np.arange(d), np.arange(d))) == 1
    assert abs(corr[near].mean() - 0.9) < 0.05
    far = np.abs(np.subtract.outer(np.arange(d), np.arange(d))) > 60
    assert abs(corr[far].mean()) < 0.05
    _, stats_diag = lmc.sample(tgt, d, draws=draws, tune=tune, chains=chains, random_seed=5, progressbar=False)
    assert stats["tree_size"].mean() < 0.8 * stats_diag["tree_size"].mean()


USER_AR1 = """
namespace lmc {
template <int NS>
struct UserTarget {   // AR(1) written by a "user": g = -P q with a tridiagonal precision, logp = q.g / 2; params = {c_end, c_mid, off}
    static constexpr bool kLanePartial = false;
    double c_end, c_mid, off; int d;
    template <class Team> __device__ void init(Team&, const double* p, int d_) { c_end = p[0]; c_mid = p[1]; off = p[2]; d = d_; }
    template <class Team> __device__ double logp_grad(Team& tm, const double (&q)[NS], double (&g)[NS]) const {
        double below, above;
        tm.neighbours(q[NS - 1], q[0], below, above);
        double part = 0.0;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int e = tm.tid() * NS + s;
            const double prev = (s == 0) ? below : q[s - 1];
            const double next = (s == NS - 1) ? above : q[s + 1];
            const double diag = (e == 0 || e == d - 1) ? c_end : c_mid;
            const double pq = (e < d) ? ((diag * q[s] + off * prev) + off * next) : 0.0;
            g[s] = -pq;
            part = __builtin_fma(q[s], g[s], part);
        }
        return 0.5 * tm.sum(part);
    }
};
}
"""


def test_runtime_compiled_density_beyond_1024_dimensions_and_with_a_dense_matrix():
    """hiprtc instantiates the general kernel for a user's functor: d = 1200 with a diagonal mass matrix, and a dense
    QuadPotentialFull (what the fused run-time path cannot do) -- both equal to the built-in AR(1) density's chains."""
    for d, dense in ((1200, False), (40, True)):
        params = T.AR1(d, 0.9).params
        user = T.UserTarget(d, USER_AR1, params=params)
        kw = dict(draws=12, tune=30, chains=4, random_seed=17, discard_tuned_samples=False, progressbar=False)
        if dense:
            mk = lambda t: dict(step=lmc.NUTS(t, d, potential=lmc.QuadPotentialFull(ar1_cov(d, 0.9))))   # noqa: E731
        else:
            mk = lambda t: {}   # noqa: E731
        a = lmc.sample(user, d, **kw, **mk(user))
        os.environ["LMC_FORCE_WIDE"] = "1" if dense else ""
        try:
            b = lmc.sample(T.AR1(d, 0.9), d, **kw, **mk(T.AR1(d, 0.9)))
        finally:
            os.environ.pop("LMC_FORCE_WIDE", None)
        np.testing.assert_array_equal(a[1]["tree_size"], b[1]["tree_size"])
        np.testing.assert_allclose(a[0], b[0], rtol=1e-9, atol=1e-12)


def test_wide_engines_on_several_devices_and_interrupts():
    """The general kernels behind the same driver: chain blocks on two engines equal one engine; Ctrl-C returns a prefix."""
    d = 1100
    kw = dict(draws=10, tune=20, chains=5, random_seed=4, discard_tuned_samples=False, progressbar=False)
    one = lmc.sample(T.StdNormal(d), d, device=0, **kw)
    two = lmc.sample(T.StdNormal(d), d, devices=[0, 0], **kw)
    np.testing.assert_array_equal(one[0], two[0])
    fired = []

    def cb(trace, draw):
        if not fired and draw.iteration >= 16:
            fired.append(draw.iteration)
            raise KeyboardInterrupt

    tr, st = lmc.sample(T.StdNormal(d), d, draws=4000, tune=50, chains=64, random_seed=4, discard_tuned_samples=False,
                        progressbar=False, callback=cb)
    n = tr.shape[1]
    assert fired and 0 < n < 4050
    full, _ = lmc.sample(T.StdNormal(d), d, draws=max(n - 50, 0), tune=min(n, 50), chains=64, random_seed=4,
                         discard_tuned_samples=False, progressbar=False)
    np.testing.assert_array_equal(tr, full)


# ---- externally evaluated densities (Python / torch callables) beyond 1024 dimensions: tick_step with TickWideShape (lmc_wide.hip) ----------
@pytest.mark.parametrize("name", ["e2e_nuts_ar1_16", "e2e_hmc_c1", "e2e_nuts_std64"])
def test_goldens_replay_through_the_wide_tick_kernel(golden_dir, name, monkeypatch):
    """LMC_FORCE_WIDE=1 with a torch callable: every iteration of the captured reference chains through the tick state
    machine of the general kernels (tick_step of lmc_tick.hpp instantiated on the 16-wavefront team)."""
    from tests.test_gpu_torch_target import torch_ar1, torch_std_normal

    monkeypatch.setenv("LMC_FORCE_WIDE", "1")
    g = _load(golden_dir, name)
    d, tune, draws = int(g["d"]), int(g["tune"]), int(g["draws"])
    kw = kwargs_from(g)
    fam = str(g["family"])
    f = OT.make(fam, d)
    tgt = torch_ar1(d) if fam == "ar1" else torch_std_normal(d)
    seeds = [int(s) for s in g["seeds"]]
    if str(g["kind"]) == "hmc":
        ostep, step = orc.Step(f, d, kind="hmc", **kw), lmc.HamiltonianMC(tgt, d, **kw)
    else:
        _s, ostep = orc.init_nuts(f, d, seeds=seeds, **kw)
        _s2, step = lmc.init_nuts(tgt, d, random_seed=seeds, **kw)
    eng = step._make_engine(1)
    try:
        assert eng.wide
    finally:
        eng.close()
    snaps, outs = oracle_chain_snapshots(ostep, g["start"], seeds[0], tune, draws)
    checked, fragile = replay_iterations_on_device(step, snaps, outs, label=name + " (wide ticks)")
    assert checked >= 0.99 * (tune + draws), (checked, fragile)


@pytest.mark.parametrize("d", [1100, 2500])
def test_callables_beyond_1024_dimensions(d):
    """The reference's own plug-in forms at model_ndim > 1024: a batched torch callable replayed against the oracle, and a
    plain per-point Python callable (integration.py:40) through sample() -- the same chains as the device functor's."""
    from tests.test_gpu_torch_target import torch_ar1

    f = OT.make("ar1", d)
    seeds = orc.derive_seeds(77 + d, 2)
    tune, draws = 12, 4
    _s, ostep = orc.init_nuts(f, d, seeds=seeds)
    start, step = lmc.init_nuts(torch_ar1(d), d, random_seed=seeds)
    snaps, outs = oracle_chain_snapshots(ostep, start, seeds[1], tune, draws)
    checked, fragile = replay_iterations_on_device(step, snaps, outs, label="torch callable d=%d" % d)
    assert checked >= tune + draws - 1
    if d > 2000:
        return
    kw = dict(draws=6, tune=10, chains=2, random_seed=5, discard_tuned_samples=False, progressbar=False)
    a = lmc.sample(lambda q: f(q), d, **kw)                      # a plain Python callable: the reference's signature
    b = lmc.sample(T.AR1(d, 0.9), d, **kw)                       # the device functor
    np.testing.assert_array_equal(a[1]["tree_size"], b[1]["tree_size"])
    np.testing.assert_allclose(a[0], b[0], rtol=1e-6, atol=1e-8)   # (numpy vs device density: last-bit differences, amplified by tuning)


# ---- rare paths through the general kernels: weight-offset moves, divergences in deep trees, depth caps -------------------
@pytest.mark.parametrize("path", ["fused", "dense", "ticks"])
def test_rare_weight_offset_paths_through_the_general_kernels(path, monkeypatch):
    """tests/test_gpu_reference_suite.py::test_weight_offset_moves_when_the_energy_drops_by_more_than_600 with
    LMC_FORCE_WIDE=1: the linear-domain weight offset moving inside accepted and rejected subtrees, in the general sampling
    kernel (diagonal and dense mass) and in the general tick kernel."""
    from tests.test_gpu_reference_suite import test_weight_offset_moves_when_the_energy_drops_by_more_than_600 as body

    monkeypatch.setenv("LMC_FORCE_WIDE", "1")
    body(path)


def test_rare_paths_differential_fuzz_through_the_general_kernels():
    """tools/fuzz_rare.py under LMC_FORCE_WIDE=1: far starts, step sizes up to the stability limit, d = 1 ... 300, depth caps,
    every sampler statistic of the first iterations against the oracle."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LMC_FORCE_WIDE="1")
    res = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_rare.py"), "60", "23"], capture_output=True,
                         text=True, timeout=900, cwd=root, env=env)
    tail = "\n".join(res.stdout.strip().split("\n")[-6:])
    assert res.returncode == 0, tail + "\n" + res.stderr[-2000:]
    assert "failures: 0" in tail
