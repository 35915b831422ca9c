"""-m gpu: dense mass matrices (QuadPotentialFull / FullInv / FullAdapt) through the C ABI
against the oracle and the fixtures captured from the reference (tests/golden/dense_*.npz, e2e_*full*.npz).

Tolerances. Everything downstream of a float64 matrix sweep agrees with numpy's dgemv to reduction-order noise
(rtol 1e-9 here). Two quantities are float32 in the reference and are NOT reproducible bit for bit: the momentum
draw ``solve_triangular(chol.T, float32 normals)`` (BLAS strsv, float32 accumulation) and the start state's
``sgemv`` velocity / ``sdot`` kinetic energy. They agree to a few float32 ulps times the condition of the solve
(F32 = 2e-5 relative below); a chain that starts from such a momentum then tracks the reference to REPLAY_F32 over
one iteration, and decisions are compared only where the oracle's decision margin exceeds DECISION. Observed on
MI355X over the 1770 captured iterations (tools/dense_parity_report.py): no integer statistic differs anywhere;
positions agree to 6e-7 (float32-born momentum) / 1e-14 (FullInv), float statistics to 3e-6 / 1e-14 relative.
"""
import os

import numpy as np
import pytest

import littlemcmc_amd as lmc
from littlemcmc_amd import _abi
from oracle import lmc_oracle as orc
from oracle import targets as otargets
from tests._gpu_util import INT_STATS, device_target

pytestmark = pytest.mark.gpu

F32 = 2e-5        # relative agreement of float32-born quantities (momentum solve, start velocity / energy)
F64 = 1e-9        # everything computed in float64 on both sides
REPLAY_F32 = 1e-5 # one whole transition started from a float32-born momentum (observed: 6e-7 positions, 3e-6 stats)
REPLAY_F64 = 1e-11
DECISION = 1e-4   # smallest oracle decision margin that must reproduce when the momentum is float32-born


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def _pot(kind, matrix):
    return lmc.QuadPotentialFull(matrix) if kind == "full" else lmc.QuadPotentialFullInv(matrix)


# ---------------------------------------------------------------------------------------------------
# unit values: velocity / energy / random / leapfrog path (reference: tests/test_quadpotential.py:67-120,
# tests/test_hmc.py:23-40)
# ---------------------------------------------------------------------------------------------------
def test_dense_units_match_reference_fixtures(golden_dir):
    g = _load(golden_dir, "dense_units")
    for ci in range(int(g["n_cases"])):
        k = "c%d_" % ci
        kind, d = str(g[k + "kind"]), int(g[k + "d"])
        tgt = device_target(g[k + "family"], d, g[k + "params"])
        pot = _pot(kind, g[k + "matrix"])
        step = lmc.HamiltonianMC(tgt, d, potential=pot)
        eng = step._engine()
        tol = F32 if kind == "full" else F64
        # potential.random() x 3 from the captured seed (global numpy stream, like the reference)
        np.random.seed(int(g[k + "seed"]))
        draws = np.array([pot.random() for _ in range(3)])
        assert str(draws.dtype) == str(g[k + "random_dtype"])
        scale = np.abs(g[k + "random"]).max()
        np.testing.assert_allclose(draws, g[k + "random"], rtol=tol, atol=tol * scale, err_msg=k + "random")
        q0 = 0.3 * np.random.randn(d)
        p0 = pot.random()
        x = np.random.randn(d)
        # the device consumed exactly the reference's stream (a cached second variate may differ by an ulp of libm)
        np.testing.assert_allclose(x, g[k + "x"], rtol=1e-15)
        np.testing.assert_allclose(q0, g[k + "q"][0], rtol=1e-15)
        np.testing.assert_allclose(pot.velocity(g[k + "x"]), g[k + "velocity"], rtol=F64, atol=F64 * np.abs(g[k + "velocity"]).max())
        np.testing.assert_allclose(pot.energy(g[k + "x"]), float(g[k + "energy_x"]), rtol=F64)
        # integrator path from the reference's own (q0, p0): compute_state + n steps out + n steps back
        n, eps = int(g[k + "n"]), float(g[k + "eps"])
        p_ref = g[k + "p"][0].astype(np.float32) if kind == "full" else g[k + "p"][0]
        out = eng.trajectory(g[k + "q"][0], p_ref, eps, n, n)
        for name in ("q", "p", "g"):
            np.testing.assert_allclose(out[name][0], g[k + name], rtol=F64, atol=F64 * np.abs(g[k + name]).max(),
                                       err_msg=k + name)
        vs = np.abs(g[k + "v"]).max()
        np.testing.assert_allclose(out["v"][0, 1:], g[k + "v"][1:], rtol=F64, atol=F64 * vs, err_msg=k + "v")
        np.testing.assert_allclose(out["v"][0, 0], g[k + "v"][0], rtol=tol, atol=tol * vs, err_msg=k + "v0")
        np.testing.assert_allclose(out["energy"][0, 1:], g[k + "energy"][1:], rtol=F64, atol=1e-9)
        np.testing.assert_allclose(out["energy"][0, 0], g[k + "energy"][0], rtol=tol, atol=tol * (1 + abs(g[k + "logp"][0])))
        # reversibility (tests/test_hmc.py:23-40, rtol 1e-5)
        np.testing.assert_allclose(out["q"][0, -1], out["q"][0, 0], rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(out["p"][0, -1], out["p"][0, 0], rtol=1e-5, atol=1e-9)


def test_dense_random_has_the_right_covariance():
    """tests/test_quadpotential.py:104-120: cov(potential.random()) == inverse of the velocity matrix."""
    np.random.seed(42)
    for _ in range(2):
        cov = np.random.rand(5, 5)
        cov += cov.T
        cov += 10 * np.eye(5)
        inv = np.linalg.inv(cov)
        for pot in (lmc.quad_potential(cov, True), lmc.quad_potential(inv, False)):
            eng = pot._eng()
            big = lmc.Engine(lmc.targets.StdNormal(5), chains=4000, potential=pot._engine_kind)
            try:
                big.set_dense_potential(pot._matrix)
                big.seed(np.arange(4000, dtype=np.uint32))
                vals = big.draw_momentum()
            finally:
                big.close()
            assert np.allclose(np.cov(vals.T), inv, atol=0.1)
            x = np.random.randn(5)
            v = cov.dot(x)
            np.testing.assert_allclose(pot.velocity(x), v, rtol=1e-4)       # tests/test_quadpotential.py:67-87
            np.testing.assert_allclose(pot.energy(x), 0.5 * x.dot(v), rtol=1e-4)
            assert eng is pot._eng()


def test_dense_rejects_what_the_reference_rejects():
    with pytest.raises(np.linalg.LinAlgError):
        lmc.QuadPotentialFull(np.array([[1.0, 2.0], [2.0, 1.0]]))._validate()      # indefinite
    with pytest.raises(lmc.quadpotential.PositiveDefiniteError):
        lmc.quad_potential(np.array([[1.0, 0.0], [0.0, -1.0]]), True)
    with pytest.raises(ValueError):
        lmc.QuadPotentialFullAdapt(3, np.zeros(3), np.eye(2), 1)
    with pytest.raises(NotImplementedError):      # per-chain adapted matrices: 26 MB per chain at 1024 (tests/test_gpu_wide.py)
        lmc.QuadPotentialFullAdapt(1025, np.zeros(1025), None, 1)
    with pytest.raises(NotImplementedError):      # shared matrices: the general kernels' 2048 (tests/test_gpu_wide.py)
        lmc.QuadPotentialFull(np.eye(2049))


# ---------------------------------------------------------------------------------------------------
# FullAdapt.update sequences (reference: tests/test_quadpotential.py:183-224)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["w20", "w15u4"])
def test_full_adapt_update_sequence_matches_reference(golden_dir, name):
    g = _load(golden_dir, "dense_adapt")
    kw = {str(n): int(v) for n, v in zip(g[name + "_kw_names"], g[name + "_kw_vals"])}
    d = g["samples"].shape[1]
    with lmc.Engine(lmc.targets.StdNormal(d), chains=1, potential="full_adapt") as eng:
        eng.set_dense_potential(np.eye(d), g["initial_mean"], 10, kw.get("adaptation_window", 101), 2.0,
                                kw.get("update_window", 1))
        for i, x in enumerate(g["samples"]):
            eng.set_position(x.reshape(1, d))
            eng.dense_update(True)
            st = eng.get_dense_state()
            tag = "%s[%d]" % (name, i)
            np.testing.assert_array_equal(st["cov"][0].astype("d"), g[name + "_cov"][i], err_msg=tag + " cov")
            np.testing.assert_allclose(st["chol"][0], g[name + "_chol"][i], rtol=1e-5, atol=1e-6, err_msg=tag + " chol")
            np.testing.assert_array_equal(st["fore_mean"][0], g[name + "_fmean"][i], err_msg=tag)
            np.testing.assert_array_equal(st["fore_raw_cov"][0], g[name + "_fraw"][i], err_msg=tag)
            np.testing.assert_array_equal(st["back_mean"][0], g[name + "_bmean"][i], err_msg=tag)
            np.testing.assert_array_equal(st["back_raw_cov"][0], g[name + "_braw"][i], err_msg=tag)
            assert st["fore_n"][0] == g[name + "_fn"][i] and st["back_n"][0] == g[name + "_bn"][i], tag
            assert st["window"][0] == g[name + "_window"][i] and st["previous_update"][0] == g[name + "_prev"][i], tag
            assert eng.adapt_state()["n_samples"][0] == g[name + "_ns"][i], tag
            assert st["chol_failures"][0] == 0


def test_full_adapt_singular_estimate_keeps_the_factor(golden_dir):
    """tests/test_quadpotential.py:215-224: a covariance estimate that cannot be factorised leaves the old factor
    in place and is reported by raise_ok()."""
    g = _load(golden_dir, "dense_adapt")
    pot = lmc.QuadPotentialFullAdapt(2, np.zeros(2), np.eye(2), 0, adaptation_window=10)
    for _ in range(11):
        pot.update(np.ones(2), None, True)
    np.testing.assert_array_equal(np.isnan(pot._cov), np.isnan(g["singular_cov"]))
    np.testing.assert_array_equal(pot._chol.astype("d"), g["singular_chol"])
    with pytest.raises(ValueError):
        pot.raise_ok(None)
    assert pot._previous_update == 10 and pot._adaptation_window == 20      # :195-212


# ---------------------------------------------------------------------------------------------------
# per-iteration parity of whole transitions: every iteration of an oracle chain replayed on the device from the
# oracle's exact pre-iteration state (one wavefront per iteration)
# ---------------------------------------------------------------------------------------------------
DENSE_E2E = ["e2e_nuts_full_ar1_12", "e2e_nuts_fullinv_ar1_12", "e2e_hmc_full_std10",
             "e2e_nuts_adaptfull_ar1_10_a", "e2e_nuts_adaptfull_ar1_10_b", "e2e_nuts_adaptfull_std70",
             "e2e_nuts_full64_ar1_12"]


def _oracle_and_device_steps(g):
    d = int(g["d"])
    potk, kind = str(g["potential"]), str(g["kind"])
    of = otargets.make(str(g["family"]), d)
    tgt = device_target(g["family"], d, g["params"])
    if potk in ("full", "inv", "full64"):
        opot = orc.FullPotential(g["matrix"], dtype="float64") if potk == "full64" else orc.quad_potential(g["matrix"], potk == "full")
        ostep = orc.Step(of, d, kind=kind, potential=opot)
        dpot = lmc.QuadPotentialFull(g["matrix"], dtype="float64") if potk == "full64" else _pot("full" if potk == "full" else "inv", g["matrix"])
        dstep = (lmc.HamiltonianMC if kind == "hmc" else lmc.NUTS)(tgt, d, potential=dpot)
        start = orc.jitter_start(int(g["seeds"][0]), d)
    else:
        start, ostep = orc.init_nuts(of, d, init=potk, seeds=[int(g["seeds"][0])])
        np.random.seed(int(g["seeds"][0]))
        start_d, dstep = lmc.init_nuts(tgt, d, init=potk, random_seed=[int(g["seeds"][0])])
        np.testing.assert_array_equal(start, start_d)
    return ostep, dstep, start


def _snapshots(ostep, start, seed, tune, draws):
    rng = np.random.RandomState(int(seed))
    q = np.array(start, dtype="d")
    ostep.tune = bool(tune)
    ostep.reset_tuning()
    snaps, outs = [], []
    for i in range(tune + draws):
        if i == 0:
            ostep.iter_count = 0
        if i == tune:
            ostep.tune = False
        pot, ad = ostep.pot, ostep.adapt
        snap = dict(q=q.copy(), rng=rng.get_state(), tune=ostep.tune, iter_count=ostep.iter_count,
                    log_step=float(np.ravel(ad.log_step)[0]), log_bar=float(np.ravel(ad.log_bar)[0]),
                    hbar=float(np.ravel(ad.hbar)[0]), da_count=ad.count, n_samples=pot.n_samples)
        if isinstance(pot, orc.FullAdaptPotential):
            snap.update(cov=pot.cov.copy(), chol=pot.chol.copy(), fore_mean=pot.fore.mean.copy(),
                        fore_raw_cov=pot.fore.raw.copy(), fore_n=pot.fore.n_samples, back_mean=pot.back.mean.copy(),
                        back_raw_cov=pot.back.raw.copy(), back_n=pot.back.n_samples, window=pot.window,
                        previous_update=pot.previous_update)
        q, st = ostep.astep(q, rng)
        m = ostep.last_margins
        snaps.append(snap)
        outs.append(dict(q=q.copy(), stats={k: np.ravel(v)[0] for k, v in st.items()},
                         margin=min(m.lb, m.turn, m.div), rng_pos=rng.get_state()[2]))
    return snaps, outs


@pytest.mark.parametrize("name", DENSE_E2E)
def test_dense_transitions_replay_the_reference_chain(golden_dir, name):
    g = _load(golden_dir, name)
    d, tune, draws = int(g["d"]), int(g["tune"]), int(g["draws"])
    ostep, dstep, start = _oracle_and_device_steps(g)
    snaps, outs = _snapshots(ostep, start, int(g["seeds"][0]), tune, draws)
    # the oracle chain IS the reference chain (pinned bit for bit by tests/test_oracle_dense_golden.py)
    np.testing.assert_allclose(np.array([o["q"] for o in outs]), g["trace"][0], rtol=1e-9, atol=1e-300)
    f32_born = str(g["potential"]) not in ("inv", "full64")
    tol = REPLAY_F32 if f32_born else REPLAY_F64
    floor = DECISION if f32_born else 1e-9
    checked = skipped = 0
    for tune_flag in (True, False):
        idx = [i for i, s in enumerate(snaps) if s["tune"] == tune_flag]
        if not idx:
            continue
        eng = dstep._make_engine(len(idx))
        try:
            eng.set_position(np.stack([snaps[i]["q"] for i in idx]))
            for c, i in enumerate(idx):
                eng.set_rng_state(c, snaps[i]["rng"])
            eng.set_chain_state({k: np.stack([np.asarray(snaps[i][k]) for i in idx]) for k in
                                 ("log_step", "log_bar", "hbar", "da_count", "iter_count", "n_samples")})
            adapt = "cov" in snaps[idx[0]]
            if adapt:
                eng.set_dense_state({k: np.stack([np.asarray(snaps[i][k]) for i in idx]) for k in
                                     ("cov", "chol", "fore_mean", "fore_raw_cov", "fore_n", "back_mean", "back_raw_cov",
                                      "back_n", "window", "previous_update")})
            eng.reserve(1, keep_trace=True)
            eng.run(1 if tune_flag else 0, 0, 1)
            assert not eng.status().any()
            q = eng.trace()[:, 0]
            stats = {k: v[:, 0] for k, v in dstep._stats_from_engine(eng, 0, 1).items()}
            after = eng.get_chain_state()
            dense_after = eng.get_dense_state() if adapt else None
            for c, i in enumerate(idx):
                want, tag = outs[i], "%s iter %d" % (name, i)
                assert eng.get_rng_state(c)[2] == want["rng_pos"] or want["margin"] < floor, tag
                if want["margin"] < floor:
                    skipped += 1
                    continue
                for sname, val in want["stats"].items():
                    got = stats[sname][c]
                    if sname in INT_STATS:
                        assert got == val, (tag, sname, got, val, want["margin"])
                    else:
                        assert np.isclose(got, val, rtol=tol, atol=tol * (1 + abs(want["stats"].get("energy", 0.0)))), (
                            tag, sname, got, val)
                np.testing.assert_allclose(q[c], want["q"], rtol=tol, atol=tol * (1 + np.abs(want["q"]).max()), err_msg=tag)
                if i + 1 < len(snaps):
                    nxt = snaps[i + 1]
                    for k in ("log_step", "log_bar", "hbar"):
                        assert np.isclose(after[k][c], nxt[k], rtol=tol, atol=tol), (tag, k)
                    assert after["da_count"][c] == nxt["da_count"] and after["n_samples"][c] == nxt["n_samples"], tag
                    if adapt and nxt["tune"] == tune_flag:
                        cs = np.abs(nxt["cov"]).max()
                        np.testing.assert_allclose(dense_after["cov"][c], nxt["cov"], rtol=0, atol=2 * tol * cs, err_msg=tag)
                        np.testing.assert_allclose(dense_after["chol"][c], nxt["chol"], rtol=0, atol=10 * tol * np.sqrt(cs), err_msg=tag)
                        assert dense_after["window"][c] == nxt["window"], tag
                        assert dense_after["previous_update"][c] == nxt["previous_update"], tag
                        assert dense_after["fore_n"][c] == nxt["fore_n"] and dense_after["back_n"][c] == nxt["back_n"], tag
                checked += 1
        finally:
            eng.close()
    assert checked >= 0.97 * (tune + draws), (checked, skipped)


# ---------------------------------------------------------------------------------------------------
# end to end through sample(): API, shapes, and that dense adaptation actually learns the covariance
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("init", ["adapt_full", "jitter+adapt_full"])
def test_sample_with_dense_init_modes(init):
    """tests/test_sampling.py:20-61 runs every init mode; shapes and dtypes as the reference returns them."""
    d, chains, draws, tune = 6, 8, 60, 260
    tgt = lmc.targets.AR1(d, 0.9)
    trace, stats = lmc.sample(tgt, d, draws=draws, tune=tune, chains=chains, init=init, random_seed=3)
    assert trace.shape == (chains, draws, d)
    assert stats["depth"].shape == (chains, draws, 1) and stats["depth"].dtype == np.int64
    assert not stats["diverging"].any()
    # the adapted matrix of the last chain is a covariance estimate of the target (unit variances, rho = 0.9)
    start, step = lmc.init_nuts(tgt, d, init=init, random_seed=3)
    assert isinstance(step.potential, lmc.QuadPotentialFullAdapt)


def test_dense_adaptation_learns_the_target_covariance():
    d, chains = 8, 64
    tgt = lmc.targets.AR1(d, 0.9)
    idx = np.arange(d)
    true_cov = 0.9 ** np.abs(idx[:, None] - idx[None, :])
    trace, stats, eng = lmc.sample(tgt, d, draws=400, tune=600, chains=chains, init="adapt_full", random_seed=11,
                                   return_engine=True)
    try:
        cov = eng.get_dense_state(fields=("cov",))["cov"]
    finally:
        eng.close()
    assert np.abs(cov.mean(axis=0) - true_cov).max() < 0.12
    emp = np.cov(trace.reshape(-1, d).T)
    assert np.abs(emp - true_cov).max() < 0.06
    # a dense metric decorrelates the AR(1) target: shallower trees than the diagonal metric needs
    _, stats_diag = lmc.sample(tgt, d, draws=400, tune=600, chains=chains, random_seed=11)
    assert stats["tree_size"].mean() < 0.7 * stats_diag["tree_size"].mean()


# ---------------------------------------------------------------------------------------------------
# wider vectors (2 and 4 elements per thread, padded lanes): oracle chains generated on the fly
# ---------------------------------------------------------------------------------------------------
def _replay(ostep, dstep, start, seed, tune, draws, f32_born, label):
    snaps, outs = _snapshots(ostep, start, seed, tune, draws)
    tol = REPLAY_F32 if f32_born else REPLAY_F64
    floor = DECISION if f32_born else 1e-9
    checked = 0
    for tune_flag in (True, False):
        idx = [i for i, s in enumerate(snaps) if s["tune"] == tune_flag]
        if not idx:
            continue
        eng = dstep._make_engine(len(idx))
        try:
            eng.set_position(np.stack([snaps[i]["q"] for i in idx]))
            for c, i in enumerate(idx):
                eng.set_rng_state(c, snaps[i]["rng"])
            eng.set_chain_state({k: np.stack([np.asarray(snaps[i][k]) for i in idx]) for k in
                                 ("log_step", "log_bar", "hbar", "da_count", "iter_count", "n_samples")})
            if "cov" in snaps[idx[0]]:
                eng.set_dense_state({k: np.stack([np.asarray(snaps[i][k]) for i in idx]) for k in
                                     ("cov", "chol", "fore_mean", "fore_raw_cov", "fore_n", "back_mean", "back_raw_cov",
                                      "back_n", "window", "previous_update")})
            eng.reserve(1, keep_trace=True)
            eng.run(1 if tune_flag else 0, 0, 1)
            assert not eng.status().any()
            q = eng.trace()[:, 0]
            stats = {k: v[:, 0] for k, v in dstep._stats_from_engine(eng, 0, 1).items()}
            dense_after = eng.get_dense_state(fields=("cov",)) if "cov" in snaps[idx[0]] else None
            for c, i in enumerate(idx):
                want, tag = outs[i], "%s iter %d" % (label, i)
                if want["margin"] < floor:
                    continue
                for sname, val in want["stats"].items():
                    got = stats[sname][c]
                    if sname in INT_STATS:
                        assert got == val, (tag, sname, got, val, want["margin"])
                    else:
                        assert np.isclose(got, val, rtol=tol, atol=tol * (1 + abs(want["stats"].get("energy", 0.0)))), (
                            tag, sname, got, val)
                np.testing.assert_allclose(q[c], want["q"], rtol=tol, atol=tol * (1 + np.abs(want["q"]).max()), err_msg=tag)
                if dense_after is not None and i + 1 < len(snaps) and snaps[i + 1]["tune"] == tune_flag:
                    cs = np.abs(snaps[i + 1]["cov"]).max()
                    np.testing.assert_allclose(dense_after["cov"][c], snaps[i + 1]["cov"], rtol=0, atol=2 * tol * cs, err_msg=tag)
                checked += 1
        finally:
            eng.close()
    return checked


@pytest.mark.parametrize("d,family", [(100, "ar1"), (130, "std_normal"), (200, "ar1"), (256, "std_normal")])
def test_dense_adapt_wide_vectors(d, family):
    of = otargets.make(family, d)
    tgt = device_target(family, d, of.params())
    seed = 900 + d
    start, ostep = orc.init_nuts(of, d, init="jitter+adapt_full", seeds=[seed])
    start_d, dstep = lmc.init_nuts(tgt, d, init="jitter+adapt_full", random_seed=[seed])
    np.testing.assert_array_equal(start, start_d)
    tune, draws = 14, 4
    assert _replay(ostep, dstep, start, seed, tune, draws, True, "adapt_full d=%d" % d) >= tune + draws - 2


@pytest.mark.parametrize("d,kind", [(130, "full"), (200, "inv"), (256, "full")])
def test_dense_fixed_wide_vectors(d, kind):
    rs = np.random.RandomState(d)
    a = rs.randn(d, d) / np.sqrt(d)
    mat = a @ a.T + 0.5 * np.eye(d)
    of = otargets.make("ar1", d)
    tgt = device_target("ar1", d, of.params())
    ostep = orc.Step(of, d, kind="nuts", potential=orc.quad_potential(mat, kind == "full"))
    dstep = lmc.NUTS(tgt, d, potential=_pot(kind, mat))
    start = 0.1 * rs.randn(d)
    n = _replay(ostep, dstep, start, 4242 + d, 10, 4, kind == "full", "%s d=%d" % (kind, d))
    assert n >= 12


def test_full_adapt_two_chains_are_independent_not_carried_over(golden_dir):
    """init="adapt_full", chains=2, cores=1: the ONE place where this engine and the reference's sequential driver
    knowingly differ (missing item 4). The reference reuses one step object for all chains
    (/root/reference/littlemcmc/sampling.py:370-383) and QuadPotentialFullAdapt.reset() is the inherited no-op
    (quadpotential.py:137-139), so its chain 1 starts from chain 0's final matrix and grown window. Here every chain is
    fresh: chain 0 is the reference's chain 0, and chain 1 is the reference's ONE-chain run with chain 1's seed (captured
    next to the two-chain run, tests/golden/capture.py group full_adapt_two_chains; the oracle reproduces both behaviours
    bit for bit, tests/test_oracle_dense_golden.py) -- what the reference's own multi-process driver computes."""
    g = _load(golden_dir, "e2e_adaptfull_two_chains")
    d, tune, draws = int(g["d"]), int(g["tune"]), int(g["draws"])
    tgt = device_target(g["family"], d, g["params"])
    seeds = [int(s) for s in g["seeds"]]
    trace, stats = lmc.sample(tgt, d, draws=draws, tune=tune, chains=2, cores=1, init="adapt_full", random_seed=seeds,
                              discard_tuned_samples=False, progressbar=False)
    n = 8   # a prefix, as for every tuned chain (float32-born momentum: REPLAY_F32 per iteration)
    for c, (want_q, want_ts) in enumerate([(g["trace"][0], g["stat_tree_size"][0]),
                                           (g["solo1_trace"][0], g["solo1_stat_tree_size"][0])]):
        np.testing.assert_array_equal(stats["tree_size"][c, :n, 0], want_ts[:n, 0], err_msg="chain %d" % c)
        np.testing.assert_allclose(trace[c, :n], want_q[:n], rtol=20 * REPLAY_F32, atol=1e-6, err_msg="chain %d" % c)
    # ... and NOT the sequential driver's chain 1 (it integrates with another matrix from its first iteration on)
    assert np.abs(trace[1, :n] - g["trace"][1, :n]).max() > 1e-3
    assert np.abs(g["solo1_trace"][0, :n] - g["trace"][1, :n]).max() > 1e-3


# ---------------------------------------------------------------------------------------------------
# size-independent properties at scale: chains are independent, so chain c of a large run IS chain c of a small one
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("init,d", [("jitter+adapt_full", 128), ("adapt_full", 40)])
def test_dense_chains_are_prefix_stable_at_scale(init, d):
    """Per-chain matrices, estimators, scratch rows and RNG streams are indexed by chain: the first 48 chains of a
    3000-chain run must equal a 48-chain run bit for bit (same seeds by construction, sampling.py:131-136)."""
    tgt = lmc.targets.AR1(d, 0.9)
    kw = dict(draws=12, tune=40, init=init, random_seed=77, discard_tuned_samples=False)
    big_tr, big_st = lmc.sample(tgt, d, chains=3000, **kw)
    small_tr, small_st = lmc.sample(tgt, d, chains=48, **kw)
    np.testing.assert_array_equal(big_tr[:48], small_tr)
    np.testing.assert_array_equal(big_st["tree_size"][:48], small_st["tree_size"])
    np.testing.assert_array_equal(big_st["energy"][:48], small_st["energy"])
    assert np.isfinite(big_tr).all() and not big_st["diverging"][:, 40:].any()


def test_dense_checkpoint_resume_is_exact():
    """A FullAdapt run split into two engine lifetimes (chain state + dense state + RNG saved and restored) equals
    the uninterrupted run bit for bit."""
    d, chains, tune, draws = 9, 10, 70, 20
    tgt = lmc.targets.AR1(d, 0.9)
    start, step = lmc.init_nuts(tgt, d, init="jitter+adapt_full", random_seed=[5])
    seeds = np.arange(chains, dtype=np.uint32) + 100

    def fresh():
        eng = step._make_engine(chains)
        eng.seed(seeds)
        eng.set_position(start)
        eng.reset_tuning()
        eng.reserve(tune + draws, keep_trace=True)
        return eng

    with fresh() as eng:
        eng.run(tune, 0, tune + draws)
        want = eng.trace()
    cut = 33
    with fresh() as eng:
        eng.run(tune, 0, cut)
        q = eng.get_position()
        chain_state = eng.get_chain_state()
        dense_state = eng.get_dense_state()
        rng = [eng.get_rng_state(c) for c in range(chains)]
        first = eng.trace(0, cut)
    with fresh() as eng:
        eng.set_position(q)
        eng.set_chain_state(chain_state)
        eng.set_dense_state({k: v for k, v in dense_state.items() if k != "chol_failures"})
        for c in range(chains):
            eng.set_rng_state(c, rng[c])
        eng.run(tune, cut, tune + draws - cut)
        second = eng.trace(cut, tune + draws - cut)
    np.testing.assert_array_equal(np.concatenate([first, second], axis=1), want)


def test_full_potential_float64(golden_dir):
    """QuadPotentialFull(cov, dtype="float64") (quadpotential.py:431-444, the dtype argument; LMC_POT_FULL_F64): velocity,
    energy and random() on the device against the values captured from the reference -- float64 throughout, so to 1e-12 --
    and same-seed chains through sample(): the momentum is solve_triangular(chol.T, z) served by the rows of L^-1."""
    g = _load(golden_dir, "e2e_nuts_full64_ar1_12")
    d = int(g["d"])
    pot = lmc.QuadPotentialFull(g["matrix"], dtype="float64")
    assert pot.dtype == "float64"
    x = g["unit_x"]
    np.testing.assert_allclose(pot.velocity(x), g["unit_velocity"], rtol=1e-12)
    np.testing.assert_allclose(pot.energy(x), float(g["unit_energy"]), rtol=1e-12)
    np.random.seed(int(g["unit_random_seed"]))
    draws = np.array([pot.random() for _ in range(3)])
    assert draws.dtype == np.float64
    np.testing.assert_allclose(draws, g["unit_random"], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(pot._chol, g["unit_chol"], rtol=1e-12, atol=1e-14)    # the float64 factor as the device holds it
    tgt = device_target(g["family"], d, g["params"])
    step = lmc.NUTS(tgt, d, potential=lmc.QuadPotentialFull(g["matrix"], dtype="float64"))
    chains, tune, draws_n = int(g["chains"]), int(g["tune"]), int(g["draws"])
    trace, stats = lmc.sample(tgt, d, draws=draws_n, tune=tune, step=step, chains=chains, cores=1, progressbar=False,
                              random_seed=int(g["random_seed"]), discard_tuned_samples=False)
    n = 12   # tuned chains decorrelate geometrically: a solid prefix, every iteration is replayed above
    np.testing.assert_array_equal(stats["tree_size"][:, :n], g["stat_tree_size"][:, :n])
    np.testing.assert_array_equal(stats["depth"][:, :n], g["stat_depth"][:, :n])
    np.testing.assert_allclose(trace[:, :n], g["trace"][:, :n], rtol=1e-7, atol=1e-9)


def test_keyboard_interrupt_in_the_shared_matrix_kernel():
    """Ctrl-C while eight chains per workgroup meet at barriers (run_dense_coop_kernel): a chain that sees the stop word
    leaves its iteration loop and keeps answering the group's barriers until every chain of the group has left -- no wave is
    left waiting, the draws every chain completed come back (sampling.py:324-328 of the reference) and equal the
    uninterrupted job's. The interrupt is raised from the callback when the device reports iteration 40 (no wall clock)."""
    d = 32
    idx = np.arange(d)
    cov = 0.9 ** np.abs(idx[:, None] - idx[None, :])
    tgt = lmc.targets.AR1(d, 0.9)
    fired = []

    def cb(trace, draw):
        if not fired and draw.iteration >= 40:
            fired.append(draw.iteration)
            raise KeyboardInterrupt

    total = 100 + 40000
    trace, stats = lmc.sample(tgt, d, draws=40000, tune=100, chains=2048, step=lmc.NUTS(tgt, d, potential=lmc.QuadPotentialFull(cov)),
                              random_seed=3, discard_tuned_samples=False, callback=cb, progressbar=False)
    n = trace.shape[1]
    print("interrupt at device iteration %s: %d of %d iterations" % (fired, n, total))
    assert fired and 0 < n < total and np.isfinite(trace).all()   # (the hint is where the FASTEST relay chain is; n what EVERY chain completed)
    assert stats["tree_size"].shape == (2048, n, 1)
    full, fstats = lmc.sample(tgt, d, draws=max(n - 100, 0), tune=min(n, 100), chains=2048,
                              step=lmc.NUTS(tgt, d, potential=lmc.QuadPotentialFull(cov)), random_seed=3,
                              discard_tuned_samples=False, progressbar=False)
    np.testing.assert_array_equal(trace, full[:, :n])
    np.testing.assert_array_equal(stats["tree_size"], fstats["tree_size"][:, :n])
