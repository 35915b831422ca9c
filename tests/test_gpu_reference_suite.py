"""-m gpu: the reference's own test-suite (/root/reference/tests/test_hmc.py, test_quadpotential.py diagonal
cases, test_sampling.py), re-read against littlemcmc_amd: same calls, same assertions, device arithmetic."""
import numpy as np
import numpy.testing as npt
import pytest

import littlemcmc_amd as lmc
from littlemcmc_amd import quadpotential
from littlemcmc_amd.targets import Normal1D

pytestmark = pytest.mark.gpu

logp_dlogp_func = Normal1D()   # tests/test_utils.py:19-28


# ---- tests/test_hmc.py -------------------------------------------------------------------------------
def test_leapfrog_reversible():
    np.random.seed(42)
    model_ndim = 1
    scaling = np.random.rand(model_ndim)
    step = lmc.HamiltonianMC(logp_dlogp_func=logp_dlogp_func, model_ndim=model_ndim, scaling=scaling)
    p = step.potential.random()
    q = np.random.randn(model_ndim)
    start = step.integrator.compute_state(p, q)
    for epsilon in [0.01, 0.1]:
        for n_steps in [1, 2, 3, 4, 20]:
            state = start
            for _ in range(n_steps):
                state = step.integrator.step(epsilon, state)
            for _ in range(n_steps):
                state = step.integrator.step(-epsilon, state)
            npt.assert_allclose(state.q, start.q, rtol=1e-5)
            npt.assert_allclose(state.p, start.p, rtol=1e-5)


def test_nuts_tuning():
    model_ndim = 1
    step = lmc.NUTS(logp_dlogp_func=logp_dlogp_func, model_ndim=model_ndim)
    trace, stats = lmc.sample(logp_dlogp_func, model_ndim, 5, 5, step=step, chains=1, cores=1)
    assert not step.tune


# ---- tests/test_quadpotential.py (diagonal cases) -----------------------------------------------------
def test_elemwise_energy():
    scaling = np.array([1, 2, 3])
    x = np.ones_like(scaling)
    pot = quadpotential.quad_potential(scaling, True)
    npt.assert_allclose(pot.energy(x), 0.5 * scaling.sum())


def test_equal_diag():
    np.random.seed(42)
    for _ in range(3):
        diag = np.random.rand(5)
        x = np.random.randn(5)
        pots = [quadpotential.quad_potential(diag, False), quadpotential.quad_potential(1.0 / diag, True)]
        v = np.diag(1.0 / diag).dot(x)
        e = x.dot(np.diag(1.0 / diag).dot(x)) / 2
        for pot in pots:
            npt.assert_allclose(pot.velocity(x), v, rtol=1e-6)
            npt.assert_allclose(pot.energy(x), e, rtol=1e-6)


def test_random_diag():
    d = np.arange(10) + 1
    np.random.seed(42)
    pots = [quadpotential.quad_potential(d, True), quadpotential.quad_potential(1.0 / d, False)]
    for pot in pots:
        vals = np.array([pot.random() for _ in range(1000)])
        npt.assert_allclose(vals.std(0), np.sqrt(1.0 / d), atol=0.1)


def test_random_consumes_the_global_numpy_stream():
    """potential.random() draws from np.random like the reference (quadpotential.py:221-224, :374-376)."""
    pot = quadpotential.QuadPotentialDiagAdapt(5, np.zeros(5), np.full(5, 4.0), 10)
    np.random.seed(7)
    got = pot.random()
    np.random.seed(7)
    want = (1.0 / np.sqrt(np.full(5, 4.0, dtype="f4"))) * np.random.normal(size=5).astype("f4")
    assert got.dtype == np.float32
    npt.assert_array_equal(got, want)
    npt.assert_array_equal(np.random.normal(size=3), np.random.RandomState(7).normal(size=8)[5:])


# ---- tests/test_sampling.py ------------------------------------------------------------------------------
@pytest.mark.parametrize("method", ["adapt_diag", "jitter+adapt_diag"])
def test_init_nuts(method):
    start, step = lmc.init_nuts(logp_dlogp_func=logp_dlogp_func, model_ndim=1, init=method)
    assert isinstance(start, np.ndarray) and len(start) == 1 and isinstance(step, lmc.NUTS)


@pytest.mark.parametrize("cls", [lmc.HamiltonianMC, lmc.NUTS])
def test_sampling_runs(cls):
    model_ndim, draws, tune, chains = 1, 3, 1, 2
    step = cls(logp_dlogp_func=logp_dlogp_func, model_ndim=model_ndim)
    trace, stats = lmc.sample(logp_dlogp_func, model_ndim, draws, tune, step=step, chains=chains, cores=1)
    assert trace.shape == (chains, draws, model_ndim)
    assert all(stats[name].shape == (chains, draws, model_ndim) for name in step.stats_dtypes[0])
    assert all(stats[name].dtype == dt for name, dt in step.stats_dtypes[0].items())


def test_multiprocess_sampling_runs():
    step = lmc.NUTS(logp_dlogp_func=logp_dlogp_func, model_ndim=1)
    trace, stats = lmc.sample(logp_dlogp_func, 1, 1, 1, step=step, chains=4, cores=4)
    assert np.var(trace) > 0   # the reference's cores=4 path returns the start point here; we do not


@pytest.mark.parametrize("cls", [lmc.HamiltonianMC, lmc.NUTS])
def test_recovers_1d_normal(cls):
    step = cls(logp_dlogp_func=logp_dlogp_func, model_ndim=1)
    trace, stats = lmc.sample(logp_dlogp_func, 1, 1000, 1000, step=step, chains=1, cores=1)
    assert np.allclose(np.mean(trace), 0, atol=1)
    assert np.allclose(np.std(trace), 1, atol=1)
    # much tighter than the reference asks: 1000 draws of a unit normal
    assert abs(np.mean(trace)) < 0.2 and abs(np.std(trace) - 1) < 0.2


def test_samples_not_all_same():
    trace, stats = lmc.sample(logp_dlogp_func, 1, 50, 10, chains=1, cores=1)
    assert np.var(trace) > 0


def test_reset_tuning():
    model_ndim, draws, tune, chains = 1, 2, 50, 2
    start, step = lmc.init_nuts(logp_dlogp_func=logp_dlogp_func, model_ndim=1)
    lmc.sample(logp_dlogp_func, model_ndim, draws=draws, tune=tune, chains=chains, step=step, start=start, cores=1)
    assert step.potential._n_samples == tune
    assert step.step_adapt._count == tune + 1


# ---- step-method protocol (_astep) with the global numpy stream ----------------------------------------------
def test_astep_protocol_matches_oracle():
    from oracle import lmc_oracle as orc
    from oracle import targets as OT

    d = 6
    step = lmc.NUTS(lmc.targets.StdNormal(d), d)
    ostep = orc.Step(OT.StdNormal(d), d, kind="nuts")
    np.random.seed(99)
    rng = np.random.RandomState(99)
    q = oq = np.full(d, 0.3)
    for i in range(12):
        q, st = step._astep(q)
        oq, ost = ostep.astep(oq, rng)
        assert st[0]["depth"] == ost["depth"] and st[0]["tree_size"] == ost["tree_size"]
        npt.assert_allclose(q, oq, rtol=1e-9, atol=1e-12)
        npt.assert_allclose(st[0]["step_size"], ost["step_size"], rtol=1e-10)
    assert step.iter_count == 12 and step.potential._n_samples == 12 and step.step_adapt._count == 13
    assert np.random.get_state()[2] == rng.get_state()[2]      # the global stream advanced identically
    step.stop_tuning()
    q, st = step._astep(q)
    assert st[0]["tune"] is np.False_ or not st[0]["tune"]


# ---- the reference's own plug-in signature: a plain per-point Python callable ----------------------------------
def test_plain_python_callable_reproduces_the_reference_chain(golden_dir):
    """tests/test_utils.py:19-28's function, passed to sample() as it is (numpy, per point, logp of shape (1,)):
    wrapped in a CallableTarget, sampled by the HIP tick kernel with the density evaluated by the caller's code --
    and the chain is the reference's (golden e2e_nuts_normal1d, captured from the reference with this density)."""
    import os

    from oracle import lmc_oracle as orc
    from oracle import targets as OT
    from tests._gpu_util import assert_chain_matches

    import scipy.stats

    # tests/test_utils.py:19-28's density as the golden capture states it (oracle/targets.py: Normal1D, a plain numpy
    # per-point callable -- analytic log-density, so that far tuning excursions do not underflow norm.pdf to -inf)
    plain = OT.make("normal1d", 1)
    assert not isinstance(plain, lmc.targets.DeviceTarget)

    def scipy_plain(x, loc=0, scale=1):   # the reference's literal spelling: log(norm.pdf(x)) of shape (1,), -(x-loc)/scale
        return np.log(scipy.stats.norm.pdf(x, loc=loc, scale=scale)), -(x - loc) / scale

    g = np.load(os.path.join(golden_dir, "e2e_nuts_normal1d.npz"))
    chains, tune, draws = int(g["chains"]), int(g["tune"]), int(g["draws"])
    calls = []

    def counted(x):
        calls.append(1)
        return plain(x)

    trace, stats = lmc.sample(counted, 1, draws=draws, tune=tune, chains=chains, cores=1, progressbar=False,
                              random_seed=int(g["random_seed"]), discard_tuned_samples=False)
    assert trace.shape == g["trace"].shape and len(calls) > chains * (tune + draws)
    _t, _s, margins = orc.sample(OT.make("normal1d", 1), 1, draws=draws, tune=tune, chains=chains,
                                 random_seed=int(g["random_seed"]), discard_tuned_samples=False, record_margins=True)
    verified = 0
    for c in range(chains):
        got = {n_: stats[n_][c, :, 0] for n_ in stats}
        want = {n_: g["stat_" + n_][c, :, 0] for n_ in stats}
        verified += assert_chain_matches(trace[c], got, g["trace"][c], want, margins[c], label="callable chain %d" % c)
    assert verified >= chains * 15
    # the step-method protocol with a plain callable, and a callable returning CPU torch tensors
    import torch

    step = lmc.NUTS(scipy_plain, 1)
    assert isinstance(step._logp_dlogp_func, lmc.targets.CallableTarget)
    np.random.seed(4)
    q, st = step._astep(np.array([0.3]))
    assert np.isfinite(q).all() and st[0]["tree_size"] >= 1

    def torch_fn(x):
        t = torch.tensor(x, requires_grad=True)
        lp = -0.5 * (t * t).sum()
        lp.backward()
        return lp, t.grad

    tr, _st = lmc.sample(torch_fn, 3, draws=30, tune=30, chains=2, random_seed=2, progressbar=False)
    assert tr.shape == (2, 30, 3) and np.isfinite(tr).all()
    with pytest.raises(TypeError):
        lmc.sample("not a callable", 2, draws=2, tune=2)


def test_pointwise_torch_callable_is_batched_with_vmap():
    import torch

    d = 7
    tgt = lmc.targets.TorchTarget.from_pointwise(d, lambda q: (-0.5 * (q * q).sum(), -q))
    kw = dict(draws=40, tune=60, chains=64, random_seed=5, progressbar=False, discard_tuned_samples=False)
    tr, st = lmc.sample(tgt, d, **kw)
    tr2, st2 = lmc.sample(lmc.targets.StdNormal(d), d, **kw)        # the fused device functor of the same density
    np.testing.assert_array_equal(st["tree_size"][:, :10], st2["tree_size"][:, :10])
    npt.assert_allclose(tr[:, :5], tr2[:, :5], rtol=1e-9, atol=1e-12)
    assert abs(tr[:, 60:].var() - 1.0) < 0.15


def test_astep_after_a_32_bit_legacy_draw():
    """np.random.randint leaves the MT19937 position odd; the device stream must stay word-exact with numpy's
    (rk_double twists between its two words), through uniforms, normals and a whole NUTS iteration."""
    from oracle import lmc_oracle as orc
    from oracle import targets as OT

    # raw stream: position 1 after the randint, 311 doubles bring it to 623 -> the next double straddles the twist
    with lmc.Engine(lmc.targets.StdNormal(3), chains=1) as eng:
        rs = np.random.RandomState(7)
        rs.randint(2 ** 30)
        assert rs.get_state()[2] % 2 == 1
        eng.set_rng_state(0, rs.get_state())
        got = eng.rng_draw([-311, -3, 5, -1, 129])[0]
        want = np.concatenate([rs.random_sample(311), rs.random_sample(3), rs.normal(size=5), rs.random_sample(1),
                               rs.normal(size=129)])
        npt.assert_allclose(got, want, rtol=5e-16, atol=0)
        st, ws = eng.get_rng_state(0), rs.get_state()
        assert st[2] == ws[2] and st[3] == ws[3]
        npt.assert_array_equal(st[1], ws[1])
    # the step-method protocol from an odd position, across the twist inside a tree (window refill with one word left)
    d = 5
    step = lmc.NUTS(lmc.targets.StdNormal(d), d)
    ostep = orc.Step(OT.StdNormal(d), d, kind="nuts")
    np.random.seed(5)
    np.random.randint(10)
    rng = np.random.RandomState(5)
    rng.randint(10)
    q = oq = np.full(d, -0.2)
    for i in range(60):   # ~60 iterations x (5 normals + tree uniforms) consume well over 624 words
        q, st = step._astep(q)
        oq, ost = ostep.astep(oq, rng)
        assert st[0]["depth"] == ost["depth"] and st[0]["tree_size"] == ost["tree_size"], i
        npt.assert_allclose(q, oq, rtol=1e-8, atol=1e-11)
    assert np.random.get_state()[2] == rng.get_state()[2]
    npt.assert_array_equal(np.random.get_state()[1], rng.get_state()[1])


def test_astep_reports_the_step_size_it_used_and_reset_keeps_the_step_adaptation():
    d = 4
    step = lmc.NUTS(lmc.targets.StdNormal(d), d)
    np.random.seed(11)
    q = np.zeros(d)
    eps0 = 0.25 / d ** 0.25
    q, st = step._astep(q)
    assert step.step_size == pytest.approx(eps0, rel=1e-15)        # base_hmc.py:151-153: the step of THIS iteration
    used_next = float(st[0]["step_size"])
    q, st = step._astep(q)
    assert step.step_size == pytest.approx(used_next, rel=1e-12)
    count, log_step = step.step_adapt._count, step.step_adapt._log_step
    step.reset()                                                   # base_hmc.py:197-200: potential only
    q, st = step._astep(q)
    assert step.step_adapt._count == count + 1                     # the device's dual averaging went on, not back to 1
    assert step.potential._n_samples == 1
    step.potential.reset()
    q, st = step._astep(q)
    assert step.step_adapt._count == count + 2 and step.step_adapt._log_step != log_step


def test_bad_initial_energy_raises_value_error():
    step = lmc.NUTS(lmc.targets.StdNormal(2), 2)
    with pytest.raises(ValueError, match="Bad initial energy"):
        step._astep(np.array([np.inf, 0.0]))
    # the reference raises per call and recovers: a good position afterwards samples normally
    np.random.seed(3)
    q, st = step._astep(np.array([0.1, -0.2]))
    assert np.isfinite(q).all() and st[0]["tree_size"] >= 1
    step.reset_tuning()
    q, st = step._astep(q)
    assert np.isfinite(q).all()
    with pytest.raises(ValueError, match="Bad initial energy"):
        lmc.sample(lmc.targets.StdNormal(2), 2, draws=2, tune=2, chains=3, start=np.array([np.nan, 0.0]), random_seed=1)


# ---- a user-written device log-density, compiled at run time and linked into the kernels -------------------
USER_SRC = r"""
#include "lmc_team.hpp"
namespace lmc {
// independent Gaussians with means mu (params) and unit variance: logp = -1/2 sum (q - mu)^2
template <int NS>
struct UserTarget {
    static constexpr bool kLanePartial = true;
    double mu[NS];
    template <class Team>
    __device__ void init(Team& tm, const double* params, int d) {
        for (int s = 0; s < NS; ++s) { const int e = tm.tid() * NS + s; mu[s] = (e < d) ? params[e] : 0.0; }
    }
    template <class Team>
    __device__ double logp_grad_partial(Team&, const double (&q)[NS], double (&g)[NS]) const {
        double part = 0.0;
        for (int s = 0; s < NS; ++s) { const double z = q[s] - mu[s]; g[s] = -z; part = __builtin_fma(z, z, part); }
        return -0.5 * part;
    }
    template <class Team>
    __device__ double logp_grad(Team& tm, const double (&q)[NS], double (&g)[NS]) const {
        return tm.sum(logp_grad_partial(tm, q, g));
    }
};
}  // namespace lmc
"""


@pytest.mark.parametrize("d", [70, 1000])   # one wavefront per chain / a team of four (72 KB of dynamic LDS)
def test_user_target_is_compiled_and_sampled(d):
    mu = np.linspace(-2.0, 3.0, d)
    tgt = lmc.targets.UserTarget(d, USER_SRC, params=mu)
    q = np.random.RandomState(0).randn(d)
    logp, grad = tgt(q)                                     # reference plug-in signature, evaluated on the GPU
    npt.assert_allclose(logp, -0.5 * np.sum((q - mu) ** 2), rtol=1e-13)
    npt.assert_allclose(grad, -(q - mu), rtol=1e-15, atol=1e-15)
    trace, stats = lmc.sample(tgt, d, draws=300, tune=300, chains=256, random_seed=5)
    npt.assert_allclose(trace.mean(axis=(0, 1)), mu, atol=0.03)
    npt.assert_allclose(trace.var(axis=(0, 1)), 1.0, atol=0.05)
    # same chain as the oracle driven by the numpy statement of the same density
    from oracle import lmc_oracle as orc

    f = lambda x: (-0.5 * np.dot(x - mu, x - mu), -(x - mu))   # noqa: E731
    ot, ost = orc.sample(f, d, draws=5, tune=25, chains=2, random_seed=5, discard_tuned_samples=False)
    gt, gst = lmc.sample(tgt, d, draws=5, tune=25, chains=2, random_seed=5, discard_tuned_samples=False)
    n = 15
    npt.assert_array_equal(gst["depth"][:, :n], ost["depth"][:, :n])
    npt.assert_allclose(gt[:, :n], ot[:, :n], rtol=1e-6, atol=1e-8)


# ---- driver extras: per-chain starts, seed lists, kwargs forwarding, warnings -------------------
def test_sample_driver_extras():
    d = 4
    tgt = lmc.targets.StdNormal(d)
    starts = [np.full(d, float(i)) for i in range(3)]
    seeds = [11, 22, 33]
    tr, st = lmc.sample(tgt, d, draws=5, tune=0, chains=3, start=starts, random_seed=seeds,
                        discard_tuned_samples=False, step=lmc.NUTS(tgt, d, adapt_step_size=False))
    # a list of seeds is used verbatim, chain by chain; a chain is a function of (seed, start) only
    tr1, st1 = lmc.sample(tgt, d, draws=5, tune=0, chains=1, start=[starts[2]], random_seed=[33],
                          discard_tuned_samples=False, step=lmc.NUTS(tgt, d, adapt_step_size=False))
    npt.assert_array_equal(tr[2], tr1[0])
    # **kwargs reach the NUTS constructor (sampling.py:149-155): a depth cap is honoured
    tr2, st2 = lmc.sample(tgt, d, draws=30, tune=30, chains=2, random_seed=5, max_treedepth=2, early_max_treedepth=2)
    assert st2["depth"].max() <= 2
    # size= is an alias of model_ndim (north_star's spelling)
    tr3, _ = lmc.sample(tgt, size=d, draws=3, tune=3, chains=2, random_seed=5)
    assert tr3.shape == (2, 3, d)


def test_warnings_follow_the_reference():
    from littlemcmc_amd.report import WarningType

    # divergences after tuning -> DIVERGENCES warning (base_hmc.py:207-227); funnel with a large fixed step size
    d = 8
    tgt = lmc.targets.Funnel(d)
    step = lmc.NUTS(tgt, d, adapt_step_size=False, step_scale=6.0)
    lmc.sample(tgt, d, draws=60, tune=5, chains=4, step=step, random_seed=3)
    kinds = [w.kind for w in step.warnings()]
    assert WarningType.DIVERGENCES in kinds
    assert step._num_divs_sample > 0 and step._samples_after_tune == 4 * 60
    # max tree depth reached in > 5 % of the draws -> TREEDEPTH warning (nuts.py:226-239)
    tgt2 = lmc.targets.StdNormal(16)
    step2 = lmc.NUTS(tgt2, 16, adapt_step_size=False, step_scale=0.02, max_treedepth=3)
    lmc.sample(tgt2, 16, draws=40, tune=0, chains=2, step=step2, random_seed=4)
    assert WarningType.TREEDEPTH in [w.kind for w in step2.warnings()]
    assert step2._reached_max_treedepth > 0
    # acceptance far from target -> BAD_ACCEPTANCE (step_sizes.py:101-121)
    assert WarningType.BAD_ACCEPTANCE in [w.kind for w in step2.warnings()]


def test_separable_user_target_student_t():
    """UserTarget.separable: per-coordinate C expressions -> device functor. Student-t(nu=5) marginals."""
    d, nu = 40, 5.0
    tgt = lmc.targets.UserTarget.separable(d, logp="-0.5*(P[0]+1.0)*log1p(q*q/P[0])",
                                           grad="-(P[0]+1.0)*q/(P[0]+q*q)", params=[nu])
    q = np.random.RandomState(1).randn(d)
    logp, grad = tgt(q)
    npt.assert_allclose(logp, np.sum(-0.5 * (nu + 1) * np.log1p(q * q / nu)), rtol=1e-12)
    npt.assert_allclose(grad, -(nu + 1) * q / (nu + q * q), rtol=1e-13)
    trace, stats = lmc.sample(tgt, d, draws=400, tune=400, chains=512, random_seed=2)
    # Var of t_5 = nu/(nu-2) = 5/3; heavy tails: pooled estimate within 8 %
    assert abs(trace.var() / (nu / (nu - 2)) - 1) < 0.08 and abs(trace.mean()) < 0.02
    assert stats["diverging"].mean() < 0.01


@pytest.mark.parametrize("path", ["fused", "dense", "ticks"])
def test_weight_offset_moves_when_the_energy_drops_by_more_than_600(path):
    """Tree weights are kept in the linear domain as e^{-dE - c} with ONE offset c per transition; a leaf whose -dE
    exceeds c by more than 600 moves the offset and rescales the stored weights. That path is rare by construction:
    force it with a unit Gaussian entered far out with a step size close to the stability limit (leapfrog's modified
    energy makes H drop by 700-860 on the way in) and compare with the oracle, which carries log-weights and has no
    such path. In four of the six chains the drop exceeds 745 inside a subtree that is then REJECTED (it turns): the
    accepted totals must survive that (they once underflowed to 0/0 and the acceptance statistic read 0)."""
    from oracle import lmc_oracle as orc
    from oracle import targets as OT

    d = 3
    f = OT.make("std_normal", d)
    sc = 1.85 * d ** 0.25
    starts = [np.full(d, v) for v in (24.0, 24.5, 25.0, 25.5, 26.0, 23.5)]
    seeds = [5, 6, 7, 8, 9, 10]
    if path == "dense":
        opot, pot = orc.FullPotential(np.eye(d)), lmc.QuadPotentialFull(np.eye(d))
    else:
        opot, pot = None, None
    if path == "ticks":
        import torch

        tgt = lmc.targets.TorchTarget(d, lambda q: (-0.5 * (q * q).sum(dim=1), -q))
    else:
        tgt = lmc.targets.StdNormal(d)
    okw = dict(potential=opot) if opot is not None else {}
    kw = dict(potential=pot) if pot is not None else {}
    ostep = orc.Step(f, d, kind="nuts", adapt_step_size=False, step_scale=sc, **okw)
    ot, ost = orc.sample(f, d, draws=6, tune=0, step=ostep, chains=len(seeds), start=starts, random_seed=seeds,
                         discard_tuned_samples=False)
    step = lmc.NUTS(tgt, d, adapt_step_size=False, step_scale=sc, **kw)
    gt, gst = lmc.sample(tgt, d, draws=6, tune=0, step=step, chains=len(seeds), start=starts, random_seed=seeds,
                         discard_tuned_samples=False)
    assert (ost["max_energy_error"].min(axis=(1, 2)) < -745.0).sum() >= 3 and not ost["diverging"].any()   # the rare case
    npt.assert_array_equal(gst["depth"], ost["depth"])
    npt.assert_array_equal(gst["tree_size"], ost["tree_size"])
    npt.assert_array_equal(gst["diverging"], ost["diverging"])
    npt.assert_allclose(gst["max_energy_error"], ost["max_energy_error"], rtol=1e-9)
    npt.assert_allclose(gst["mean_tree_accept"], ost["mean_tree_accept"], rtol=1e-8, atol=1e-300)
    npt.assert_allclose(gt, ot, rtol=1e-9, atol=1e-10)


def test_rare_paths_differential_fuzz():
    """tools/fuzz_rare.py: chains entered far out in the tails with step sizes up to the stability limit (weight-offset
    moves in shallow and deep trees, divergences in either leaf of a pair, depth-capped trees, one-wave and team
    kernels), every sampler statistic of the first iterations against the oracle."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_rare.py"), "80", "11"], capture_output=True,
                         text=True, timeout=900, cwd=root)
    tail = "\n".join(res.stdout.strip().split("\n")[-6:])
    assert res.returncode == 0, tail + "\n" + res.stderr[-2000:]
    assert "failures: 0" in tail


@pytest.mark.parametrize("kind", ["nuts", "hmc", "nuts_dense", "nuts_ticks"])
def test_step_rand_uniform_draws_from_the_chain_stream(golden_dir, kind):
    """step_rand (base_hmc.py:154-155) as StepRandUniform(lo, hi) == lambda s: s * np.random.uniform(lo, hi): the jitter
    uniform comes out of each chain's own stream between the start state and the trajectory, in the fused, dense and
    tick kernels alike. The NUTS / HMC cases reproduce chains captured from the imported reference running that very
    lambda (tests/golden/e2e_step_rand.npz); the dense / tick cases are checked against the oracle."""
    import os

    from oracle import lmc_oracle as orc
    from oracle import targets as OT

    g = np.load(os.path.join(golden_dir, "e2e_step_rand.npz"))
    lo, hi, seed = float(g["lo"]), float(g["hi"]), int(g["random_seed"])
    sr = lmc.StepRandUniform(lo, hi)
    n = 12   # iterations compared: dual averaging amplifies 1-ulp differences beyond that
    if kind == "hmc":
        d, chains, tune, draws = int(g["hmc_d"]), int(g["hmc_chains"]), int(g["hmc_tune"]), int(g["hmc_draws"])
        tgt = lmc.targets.StdNormal(d)
        step = lmc.HamiltonianMC(tgt, d, path_length=float(g["hmc_path_length"]), step_rand=sr)
        gt, gst = lmc.sample(tgt, d, draws=draws, tune=tune, step=step, chains=chains, random_seed=seed,
                             discard_tuned_samples=False)
        want_t, want_n = g["hmc_trace"], g["hmc_stat_n_steps"]
        npt.assert_array_equal(gst["n_steps"][:, :n], want_n[:, :n])
        npt.assert_allclose(gst["step_size"][:, :n], g["hmc_stat_step_size"][:, :n], rtol=1e-9)
        npt.assert_allclose(gt[:, :n], want_t[:, :n], rtol=1e-7, atol=1e-9)
        return
    d, chains, tune, draws = int(g["nuts_d"]), int(g["nuts_chains"]), int(g["nuts_tune"]), int(g["nuts_draws"])
    f = OT.make("ar1", d)
    if kind == "nuts":
        tgt = lmc.targets.AR1(d)
        gt, gst = lmc.sample(tgt, d, draws=draws, tune=tune, chains=chains, random_seed=seed, discard_tuned_samples=False,
                             step_rand=sr)
        want_t, want_ts = g["nuts_trace"], g["nuts_stat_tree_size"]
    else:
        if kind == "nuts_dense":
            tgt, init = lmc.targets.AR1(d), "adapt_full"
        else:
            import torch

            prec = torch.as_tensor(np.linalg.inv(0.9 ** np.abs(np.subtract.outer(np.arange(d), np.arange(d)))), device="cuda")
            tgt, init = lmc.targets.TorchTarget(d, lambda q: (-0.5 * ((q @ prec) * q).sum(dim=1), -(q @ prec))), "auto"
        gt, gst = lmc.sample(tgt, d, draws=draws, tune=tune, chains=chains, random_seed=seed, discard_tuned_samples=False,
                             step_rand=sr, init=init)
        want_t, ost = orc.sample(f, d, draws=draws, tune=tune, chains=chains, random_seed=seed, discard_tuned_samples=False,
                                 step_rand=(lo, hi), init=init)
        want_ts = ost["tree_size"]
        n = 6 if kind == "nuts_dense" else 10
    npt.assert_array_equal(gst["tree_size"][:, :n], want_ts[:, :n])
    tol = 5e-4 if kind == "nuts_dense" else 1e-6
    npt.assert_allclose(gt[:, :n], want_t[:, :n], rtol=tol, atol=tol)
