"""-m gpu: densities given as batched torch callables (littlemcmc_amd.targets.TorchTarget)
driven through the tick protocol of the C ABI (lmc_engine_tick_begin / lmc_engine_tick, csrc/lmc_tick.hpp).

The tick kernel is the transition kernel cut at the density evaluation, so the same parity bar applies: every
iteration of the reference chains, replayed from the oracle's exact pre-iteration state, must reproduce the oracle's
iteration -- integer statistics exactly, positions / energies / adaptation state to 1e-10 (the torch density and the
numpy density differ only in summation order)."""
import os

import numpy as np
import pytest

import littlemcmc_amd as lmc
from littlemcmc_amd.targets import TorchTarget
from oracle import lmc_oracle as orc
from oracle import targets as OT
from tests._gpu_util import kwargs_from, oracle_chain_snapshots, replay_iterations_on_device

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def torch_std_normal(d):
    return TorchTarget(d, lambda q: (-0.5 * (q * q).sum(dim=1), -q))


def torch_ar1(d, rho=0.9):
    c = 1.0 / (1.0 - rho * rho)
    c_end, c_mid, off = c, (1.0 + rho * rho) * c, -rho * c

    diag = torch.full((d,), c_mid, dtype=torch.float64, device="cuda")   # built once: fn itself stays graph-capturable
    diag[0] = c_end
    diag[d - 1] = c_end

    def fn(q):
        pq = diag * q
        if d > 1:
            pq[:, 1:] += off * q[:, :-1]
            pq[:, :-1] += off * q[:, 1:]
        g = -pq
        return 0.5 * (q * g).sum(dim=1), g

    return TorchTarget(d, fn)


def torch_funnel(d):
    """Neal's funnel through autograd (oracle/targets.py: Funnel)."""

    def logp(q):
        v = q[:, 0]
        rest = q[:, 1:]
        return -0.5 * v * v / 9.0 - 0.5 * (rest * rest).sum(dim=1) * torch.exp(-v) - 0.5 * (d - 1) * v

    return TorchTarget.from_logp(d, logp)


@pytest.mark.parametrize("name", ["e2e_nuts_ar1_16", "e2e_hmc_c1", "e2e_nuts_std64"])
def test_every_iteration_of_the_golden_runs_through_ticks(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    d, tune, draws = int(g["d"]), int(g["tune"]), int(g["draws"])
    kw = kwargs_from(g)
    fam = str(g["family"])
    f = OT.make(fam, d)
    tgt = torch_ar1(d) if fam == "ar1" else torch_std_normal(d)
    seeds = [int(s) for s in g["seeds"]]
    if str(g["kind"]) == "hmc":
        ostep = orc.Step(f, d, kind="hmc", **kw)
        step = lmc.HamiltonianMC(tgt, d, **kw)
    else:
        _s, ostep = orc.init_nuts(f, d, seeds=seeds, **kw)
        _s2, step = lmc.init_nuts(tgt, d, random_seed=seeds, **kw)
        np.testing.assert_array_equal(_s, _s2)
    snaps, outs = oracle_chain_snapshots(ostep, g["start"], seeds[0], tune, draws)
    checked, fragile = replay_iterations_on_device(step, snaps, outs, label=name + " (ticks)")
    assert checked >= 0.99 * (tune + draws), (checked, fragile)


@pytest.mark.parametrize("d", [130, 250, 500, 1000])
def test_ticks_on_wide_vectors(d):
    f = OT.make("ar1", d)
    tgt = torch_ar1(d)
    seeds = orc.derive_seeds(99, 2)
    _s, ostep = orc.init_nuts(f, d, seeds=seeds)
    start, step = lmc.init_nuts(tgt, d, random_seed=seeds)
    snaps, outs = oracle_chain_snapshots(ostep, start, seeds[1], 30, 8)
    checked, fragile = replay_iterations_on_device(step, snaps, outs, label="ticks d=%d" % d)
    assert checked >= 37, (checked, fragile)


def test_sample_with_a_torch_target_matches_the_fused_kernel():
    """Whole sample() runs: the tick path and the fused kernel start identical and stay identical while tuning's
    feedback has not amplified the density's last-bit differences (cf. tests/test_gpu_parity.py e2e prefix)."""
    d, chains, tune, draws = 12, 24, 40, 30
    a_tr, a_st = lmc.sample(torch_ar1(d), d, draws=draws, tune=tune, chains=chains, random_seed=5, discard_tuned_samples=False)
    b_tr, b_st = lmc.sample(lmc.targets.AR1(d), d, draws=draws, tune=tune, chains=chains, random_seed=5, discard_tuned_samples=False)
    assert a_tr.shape == b_tr.shape == (chains, tune + draws, d)
    pre = 12
    np.testing.assert_array_equal(a_st["tree_size"][:, :pre], b_st["tree_size"][:, :pre])
    np.testing.assert_array_equal(a_st["depth"][:, :pre], b_st["depth"][:, :pre])
    np.testing.assert_allclose(a_tr[:, :pre], b_tr[:, :pre], rtol=1e-6, atol=1e-9)
    # and the draws are draws from the target: unit marginal variances, lag-1 correlation 0.9
    x = a_tr[:, tune:].reshape(-1, d)
    assert abs(x.var(axis=0).mean() - 1.0) < 0.25


def test_autograd_target_samples_the_funnel():
    d, chains = 6, 256
    trace, stats = lmc.sample(torch_funnel(d), d, draws=150, tune=250, chains=chains, random_seed=9, max_treedepth=8,
                              discard_tuned_samples=False)
    ftrace, fstats = lmc.sample(lmc.targets.Funnel(d), d, draws=150, tune=250, chains=chains, random_seed=9, max_treedepth=8,
                                discard_tuned_samples=False)
    assert stats["tree_size"].min() >= 1 and stats["depth"].max() <= 8
    # the same density as a fused device functor: identical first transitions ...
    np.testing.assert_array_equal(stats["tree_size"][:, :5], fstats["tree_size"][:, :5])
    np.testing.assert_allclose(trace[:, :5], ftrace[:, :5], rtol=1e-7, atol=1e-9)
    # ... and the same distribution of draws afterwards (NUTS on the raw funnel is biased in the neck; the two
    # implementations must agree with each other, q_0 roughly N(0, 3^2))
    v, vf = trace[:, 250:, 0].ravel(), ftrace[:, 250:, 0].ravel()
    assert abs(v.mean() - vf.mean()) < 0.35 and abs(v.std() - vf.std()) < 0.35
    assert 1.5 < v.std() < 3.6


def test_torch_target_contract_errors():
    d = 4
    bad_shape = TorchTarget(d, lambda q: (q.sum(dim=1), q[:, :2]))
    with pytest.raises(ValueError, match="must return"):
        lmc.sample(bad_shape, d, draws=2, tune=2, chains=2, random_seed=1)
    on_cpu = TorchTarget(d, lambda q: (q.sum(dim=1).cpu(), q.cpu()))
    with pytest.raises(TypeError, match="no CPU path"):
        lmc.sample(on_cpu, d, draws=2, tune=2, chains=2, random_seed=1)
    with pytest.raises(TypeError):
        TorchTarget(d, "not callable")
    with pytest.raises(NotImplementedError, match="up to model_ndim = 256"):
        lmc.sample(torch_std_normal(300), 300, draws=2, tune=2, chains=2, random_seed=1, init="adapt_full")
    # reference plug-in signature on one point
    logp, grad = torch_std_normal(d)(np.arange(4.0))
    assert np.isclose(logp, -7.0) and np.allclose(grad, -np.arange(4.0))


def test_tick_protocol_through_the_c_abi_directly():
    """include/lmc_hip.h protocol without the Python loop helper: reserve, tick_begin, evaluate, tick ... until no
    chain is active; finished chains ignore further ticks."""
    d, chains, n = 5, 7, 6
    tgt = torch_std_normal(d)
    with lmc.Engine(tgt, chains=chains) as eng:
        eng.seed(np.arange(chains, dtype=np.uint32))
        eng.set_position(np.zeros(d))
        eng.reset_tuning()
        eng.reserve(n, keep_trace=True)
        with pytest.raises(lmc._abi.HipLibraryError, match="tick"):
            eng._check(eng._lib.lmc_engine_run(eng._h, n, 0, n))      # the fused entry point refuses an external density
        stream = torch.cuda.Stream()
        eng.set_stream(stream.cuda_stream)
        with torch.cuda.stream(stream):
            eng.tick_begin(n, 0, n)
            from littlemcmc_amd.engine import _TickView
            q = torch.as_tensor(_TickView(eng.tick_positions_ptr(), (chains, d)), device="cuda")
            ticks, active = 0, chains
            while active:
                logp, grad = tgt.evaluate(q)
                active = eng.tick(logp.data_ptr(), grad.data_ptr(), wait=True)
                ticks += 1
            assert eng.tick(logp.data_ptr(), grad.data_ptr(), wait=True) == 0
        eng.set_stream(None)
        leap = eng.stat_i32(lmc._abi.STAT_TREE_SIZE, 0, n)
        assert ticks == (leap.sum(axis=1) + n).max()       # one evaluation per leapfrog + one per iteration start
        assert eng.counters()[:, lmc._abi.CT_LEAPFROGS].sum() == leap.sum()


def test_tick_chains_are_prefix_stable_at_scale():
    """Chains driven through ticks are independent of how many other chains share the launch: the first 32 chains
    of a 5000-chain run equal a 32-chain run bit for bit (the callable is evaluated row by row)."""
    d = 20
    kw = dict(draws=10, tune=30, random_seed=31, discard_tuned_samples=False)
    big_tr, big_st = lmc.sample(torch_std_normal(d), d, chains=5000, **kw)
    small_tr, small_st = lmc.sample(torch_std_normal(d), d, chains=32, **kw)
    np.testing.assert_array_equal(big_st["tree_size"][:32], small_st["tree_size"])
    np.testing.assert_array_equal(big_tr[:32], small_tr)


@pytest.mark.parametrize("kind", ["nuts", "hmc"])
def test_ticks_with_a_fixed_diagonal_scaling(kind):
    """QuadPotentialDiag (scaling=..., float64 momentum draw, no mass adaptation) through ticks == fused kernel."""
    d, chains = 9, 16
    scaling = np.linspace(0.5, 2.0, d)
    cls = lmc.NUTS if kind == "nuts" else lmc.HamiltonianMC
    out = []
    for tgt in (torch_ar1(d), lmc.targets.AR1(d)):
        step = cls(tgt, d, scaling=scaling, is_cov=True)
        out.append(lmc.sample(tgt, d, draws=10, tune=15, step=step, chains=chains, random_seed=4, discard_tuned_samples=False))
    (a_tr, a_st), (b_tr, b_st) = out
    key = "tree_size" if kind == "nuts" else "n_steps"
    np.testing.assert_array_equal(a_st[key][:, :10], b_st[key][:, :10])
    np.testing.assert_allclose(a_tr[:, :10], b_tr[:, :10], rtol=1e-7, atol=1e-9)


def test_graph_replay_of_the_callable_gives_the_same_chains():
    """TorchTarget(graph=True): fn captured once into a HIP graph and replayed every tick == eager evaluation."""
    d, chains = 10, 40
    kw = dict(draws=12, tune=25, chains=chains, random_seed=13, discard_tuned_samples=False)
    eager_tr, eager_st = lmc.sample(torch_ar1(d), d, **kw)
    t = torch_ar1(d)
    t.graph = True
    graph_tr, graph_st = lmc.sample(t, d, **kw)
    np.testing.assert_array_equal(graph_st["tree_size"], eager_st["tree_size"])
    np.testing.assert_array_equal(graph_tr, eager_tr)


# ---------------------------------------------------------------------------------------------------
# torch-callable density x dense mass matrix: tick_step of csrc/lmc_tick.hpp with the dense Mass policy (lmc_dense.hip)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["e2e_nuts_full_ar1_12", "e2e_nuts_fullinv_ar1_12", "e2e_hmc_full_std10",
                                  "e2e_nuts_adaptfull_ar1_10_b", "e2e_nuts_adaptfull_std70"])
def test_dense_mass_through_ticks_replays_the_reference_chain(golden_dir, name):
    """The captured dense-mass reference chains (tests/golden/e2e_*full*.npz) replayed iteration by iteration with the
    density evaluated by a torch callable: same bar as the fused dense kernel (tests/test_gpu_dense.py)."""
    from tests.test_gpu_dense import _pot, _replay

    g = np.load(os.path.join(golden_dir, name + ".npz"))
    d, tune, draws = int(g["d"]), int(g["tune"]), int(g["draws"])
    potk, kind, fam = str(g["potential"]), str(g["kind"]), str(g["family"])
    of = OT.make(fam, d)
    tgt = torch_ar1(d) if fam == "ar1" else torch_std_normal(d)
    seed = int(g["seeds"][0])
    if potk in ("full", "inv"):
        ostep = orc.Step(of, d, kind=kind, potential=orc.quad_potential(g["matrix"], potk == "full"))
        dstep = (lmc.HamiltonianMC if kind == "hmc" else lmc.NUTS)(tgt, d, potential=_pot("full" if potk == "full" else "inv", g["matrix"]))
        start = orc.jitter_start(seed, d)
    else:
        start, ostep = orc.init_nuts(of, d, init=potk, seeds=[seed])
        start_d, dstep = lmc.init_nuts(tgt, d, init=potk, random_seed=[seed])
        np.testing.assert_array_equal(start, start_d)
    checked = _replay(ostep, dstep, start, seed, tune, draws, potk != "inv", name + " (ticks)")
    assert checked >= 0.97 * (tune + draws)


def test_sample_torch_target_with_dense_adaptation():
    """sample(TorchTarget, init="adapt_full"): chains finish their tuning iterations in different ticks and each gets
    its FullAdapt.update right after its own iteration (masked launch of dense_adapt_kernel); the run starts identical
    to the fused dense kernel and learns the target's covariance."""
    d, chains = 8, 64
    kw = dict(draws=200, tune=500, chains=chains, init="adapt_full", random_seed=11, discard_tuned_samples=False)
    t_tr, t_st, eng = lmc.sample(torch_ar1(d), d, return_engine=True, **kw)
    try:
        cov = eng.get_dense_state(fields=("cov",))["cov"]
        ns = eng.adapt_state()["n_samples"]
    finally:
        eng.close()
    assert (ns == 500).all()                                  # exactly one update per tuning iteration per chain
    f_tr, f_st = lmc.sample(lmc.targets.AR1(d), d, **kw)
    np.testing.assert_array_equal(t_st["tree_size"][:, :10], f_st["tree_size"][:, :10])
    np.testing.assert_allclose(t_tr[:, :10], f_tr[:, :10], rtol=1e-6, atol=1e-8)
    idx = np.arange(d)
    true_cov = 0.9 ** np.abs(idx[:, None] - idx[None, :])
    assert np.abs(cov.mean(axis=0) - true_cov).max() < 0.15
    x = t_tr[:, 500:].reshape(-1, d)
    assert np.abs(np.cov(x.T) - true_cov).max() < 0.1
