"""-m gpu: round-5 fixes of the round-4 advisor findings -- the stop-word relay of launches whose size is not a power of two,
step_rand bookkeeping (the step object keeps the ADAPTED step size; an override handed back restores the device's own
jitter; `callback` on the host-step_rand path), and the standalone integrator's dtype for QuadPotentialFullAdapt."""
import numpy as np
import pytest

import littlemcmc_amd as lmc
from littlemcmc_amd import _abi
from littlemcmc_amd import targets as T

pytestmark = pytest.mark.gpu


def _engine(tgt, d, chains, seed=7, **kw):
    seeds = lmc.distributed.global_seeds(seed, chains)
    start, step = lmc.init_nuts(tgt, d, random_seed=seeds, **kw)
    eng = step._make_engine(chains)
    eng.seed(seeds)
    eng.set_position(start)
    eng.reset_tuning()
    return eng, step


@pytest.mark.parametrize("chains", [130, 3, 255])
def test_relay_rotation_covers_every_mark_of_a_launch_of_any_size(chains):
    """lmc_sampler.hpp: stop_request_load. Every 16th iteration ONE chain in (relay_mask + 1) of a launch reads the host's
    stop word and leaves the iteration index in the progress word; WHICH chain rotates with the iteration. Round 4 rounded
    relay_mask + 1 UP to a power of two, so for 130 chains (two sub-blocks of 65, mask 127) the residues 65..127 had no
    chain: after the first mark nobody relayed for ~60 marks -- ~1 000 iterations without Ctrl-C or progress (found by
    review). With the largest power of two <= n every mark has a relay: after a 160-iteration launch the progress word
    stands at one of the last marks (it would still read 0 with the round-4 mask)."""
    d = 8
    eng, _step = _engine(T.StdNormal(d), d, chains)
    try:
        eng.reserve(160, keep_trace=False)
        eng.run(160, 0, 160)
        eng.synchronize()
        assert not eng.status().any()
        # marks are iterations 0, 16, ..., 144; chains of a launch run a few iterations apart, so the LAST writer may be a
        # mark or two behind the newest one
        assert eng.progress() >= 112, eng.progress()
    finally:
        eng.close()


def test_interrupting_a_130_chain_job_returns_a_prefix():
    """The functional side of the same finding: Ctrl-C on a job whose launch size is not a power of two."""
    from tests.test_gpu_scale import _InterruptAt

    d, chains, tune, draws = 16, 130, 50, 200000
    tgt = T.StdNormal(d)
    cb = _InterruptAt(64)
    trace, stats = lmc.sample(tgt, d, draws=draws, tune=tune, chains=chains, random_seed=12, discard_tuned_samples=False,
                              callback=cb, progressbar=False, stream_results=False)   # (a 200 050-iteration job only to be interrupted: nothing to pin)
    n = trace.shape[1]
    assert cb.fired_at is not None and 0 < n < tune + draws
    full, _fs = lmc.sample(tgt, d, draws=max(n - tune, 0), tune=min(n, tune), chains=chains, random_seed=12,
                           discard_tuned_samples=False, progressbar=False)
    np.testing.assert_array_equal(trace, full)


def test_step_size_on_the_step_object_stays_the_adapted_value():
    """base_hmc.py:151-155 of the reference: `self.step_size = self.step_adapt.current(...)`, then a LOCAL
    `step_size = self._step_rand(step_size)`. The object keeps the adapted value; only the integrator sees the jittered one."""
    d = 5
    tgt = T.StdNormal(d)
    seeds = [11, 12]
    start, step = lmc.init_nuts(tgt, d, random_seed=seeds, step_rand=lambda s: 0.5 * s)
    np.random.seed(seeds[0])
    step.tune = True
    step.reset_tuning()
    q = start
    for _ in range(3):
        adapted = float(np.exp(step.step_adapt._log_step))
        q, _st = step._astep(q)
        assert step.step_size == adapted          # round 4 left 0.5 * adapted here


def test_handing_back_a_step_size_override_restores_the_device_jitter():
    """lmc_engine_set_step_sizes(e, NULL): "back to the adapted step sizes, or to the device's own jitter if that was set".
    Round 4 dropped the jitter (step_jitter 2 -> 0). A: device jitter only. B: device jitter, an override set and handed
    back before the launch. The two must be the same chains."""
    d, chains, n = 6, 8, 12
    out = []
    for with_override in (False, True):
        eng, _step = _engine(T.StdNormal(d), d, chains)
        try:
            eng.set_step_jitter(0.5, 0.5)         # lo == hi: step * 0.5, and one uniform of the chain's stream consumed
            if with_override:
                eng.set_step_sizes(np.full(chains, 0.123))
                eng.set_step_sizes(None)
            eng.reserve(n, keep_trace=True)
            eng.run(n, 0, n)
            eng.synchronize()
            out.append((eng.trace().copy(), eng.stat_i32(_abi.STAT_TREE_SIZE, 0, n).copy()))
        finally:
            eng.close()
    np.testing.assert_array_equal(out[0][1], out[1][1])
    np.testing.assert_array_equal(out[0][0], out[1][0])
    # ... and the jitter is in force in both (without it the chains differ)
    eng, _step = _engine(T.StdNormal(d), d, chains)
    try:
        eng.reserve(n, keep_trace=True)
        eng.run(n, 0, n)
        eng.synchronize()
        assert np.abs(eng.trace() - out[0][0]).max() > 1e-3
    finally:
        eng.close()


def test_host_step_rand_job_calls_back_and_hands_the_engine_back_clean():
    """sampling._run_job_host_step_rand: `callback` is honoured (once per iteration), a KeyboardInterrupt raised in it ends
    the job with the completed rows, and the engine that comes back (return_engine=True) no longer carries the per-chain
    step-size override -- round 4 dropped the callback and left the override in place after an interrupt."""
    d, chains, tune, draws = 4, 6, 20, 20
    tgt = T.StdNormal(d)
    seen = []

    def cb(trace, draw):
        seen.append(draw.iteration)
        if draw.iteration == 9:
            raise KeyboardInterrupt

    trace, stats, eng = lmc.sample(tgt, d, draws=draws, tune=tune, chains=chains, random_seed=4, discard_tuned_samples=False,
                                   progressbar=False, step_rand=lambda s: 0.9 * s, callback=cb, return_engine=True)
    try:
        assert seen == list(range(10))
        assert trace.shape == (chains, 10, d)
        # the engine integrates with its adapted step sizes again: one more iteration equals the same iteration of an
        # engine that never had an override and is put into the same state
        state = eng.get_chain_state()
        rng = [eng.get_rng_state(c) for c in range(chains)]
        pos = eng.get_position().copy()
        eng.reserve(1, keep_trace=True)
        eng.run(1, 0, 1)
        eng.synchronize()
        got = eng.trace()[:, 0].copy()
    finally:
        eng.close()
    seeds = lmc.distributed.global_seeds(4, chains)
    _start, step = lmc.init_nuts(tgt, d, random_seed=seeds)
    ref = step._make_engine(chains)
    try:
        ref.set_position(pos)
        for c in range(chains):
            ref.set_rng_state(c, rng[c])
        ref.set_chain_state(state)
        ref.reserve(1, keep_trace=True)
        ref.run(1, 0, 1)
        ref.synchronize()
        np.testing.assert_array_equal(got, ref.trace()[:, 0])
    finally:
        ref.close()


def test_standalone_integrator_keeps_full_adapt_float64():
    """integration.HipLeapfrogIntegrator built around QuadPotentialFullAdapt(dtype="float64") on its own (no step object):
    the engine it creates must be a float64 FullAdapt engine, like the one base_hmc / quadpotential create (round 4 listed
    only the diagonal kinds there, so cov / chol came back float32 and the float32-momentum rules applied)."""
    from littlemcmc_amd.integration import CpuLeapfrogIntegrator

    d = 4
    tgt = T.StdNormal(d)
    cov = np.diag(np.linspace(0.5, 2.0, d))
    pot = lmc.QuadPotentialFullAdapt(d, np.zeros(d), cov, 1, dtype="float64")
    integ = CpuLeapfrogIntegrator(pot, tgt)
    q = np.linspace(-1.0, 1.0, d)
    p = np.array([0.3, -0.2, 0.1, 0.7])             # float64 momentum
    st = integ.compute_state(q, p)
    assert integ._eng().mass_f64
    assert st.v.dtype == np.float64
    np.testing.assert_allclose(st.v, cov.dot(p), rtol=1e-14)
    np.testing.assert_allclose(float(np.ravel(st.energy)[0]), 0.5 * p.dot(cov.dot(p)) + 0.5 * q.dot(q), rtol=1e-14)


# ---- LDS plans of the one-wave sampling kernels (lmc_sampler.hpp: run_kernel<.., PL>; lmc_engine.hip: choose_lds_plan) ------------
def _plan_job(monkeypatch, plan, tgt, d, chains, n, kw=None):
    """plan: "0" / "1" pin the LDS plan through lmc_config.lds_plan (ABI 8; an environment variable until round 5), None
    leaves the choice to the engine."""
    eng, step = _engine(tgt, d, chains, lds_plan={"0": "shallow", "1": "deep", None: "auto"}[plan], **(kw or {}))
    try:
        eng.reserve(n, keep_trace=True)
        for first in range(0, n, 52):          # launches enqueued one by one, each after the one before has reported tree sizes
            eng.run(n // 2, first, min(52, n - first))   # (the engine considers the deep-tree plan from iteration 200 on)
            eng.synchronize()
        assert not eng.status().any()
        out = (eng.trace().copy(), eng.stat_i32(_abi.STAT_TREE_SIZE, 0, n).copy(), eng.stat_f64(_abi.STAT_ENERGY, 0, n).copy(),
               [eng.get_rng_state(c)[2] for c in range(min(chains, 4))])
        if plan is not None:               # a pinned plan is the plan of every launch
            assert eng.last_run_plan() == {"0": "shallow", "1": "deep"}[plan]
        return out, eng.run_lds_bytes()
    finally:
        eng.close()


@pytest.mark.parametrize("family,d,kw", [("ar1", 128, {}), ("funnel", 256, {"max_treedepth": 12}), ("std_normal", 64, {}),
                                           ("ar1", 40, {})])
def test_lds_plans_are_bit_identical(monkeypatch, family, d, kw):
    """The two LDS plans of the one-wave sampling kernels differ in WHERE a chain's private data lives -- plan 0: MT19937
    state and three cold slots in LDS, stack level 2 in the scratch row; plan 1: the generator used in place, stack level 2
    in LDS -- never in arithmetic. Pinned to plan 0, pinned to plan 1, or chosen per launch by the engine from the tree
    sizes the chains report: the same draws, statistics and generator positions, bit for bit."""
    tgt = {"ar1": lambda: T.AR1(d, 0.9), "funnel": lambda: T.Funnel(d), "std_normal": lambda: T.StdNormal(d)}[family]()
    chains, n = 96, 260
    ref, lds0 = _plan_job(monkeypatch, "0", tgt, d, chains, n, kw)
    one, lds1 = _plan_job(monkeypatch, "1", tgt, d, chains, n, kw)
    auto, _ldsa = _plan_job(monkeypatch, None, tgt, d, chains, n, kw)
    assert lds1 != lds0                      # the pinned plan really is another LDS layout
    for other in (one, auto):
        np.testing.assert_array_equal(ref[0], other[0])
        np.testing.assert_array_equal(ref[1], other[1])
        np.testing.assert_array_equal(ref[2], other[2])
        assert ref[3] == other[3]


def test_engine_follows_the_tree_sizes_the_chains_report(monkeypatch):
    """choose_lds_plan: deep trees (AR(1) d = 128 settles at 60-130 leapfrogs per iteration) move the launches that are
    enqueued after the chains have reported to the deep-tree plan -- from iteration 200 on, past the early-treedepth regime
    -- and shallow trees (standard normal, 7 per iteration) stay."""
    _out, lds_deep = _plan_job(monkeypatch, None, T.AR1(128, 0.9), 128, 256, 260)
    _out, lds_shallow = _plan_job(monkeypatch, None, T.StdNormal(128), 128, 256, 260)
    _out, lds_plan1 = _plan_job(monkeypatch, "1", T.StdNormal(128), 128, 256, 60)
    _out, lds_plan0 = _plan_job(monkeypatch, "0", T.StdNormal(128), 128, 256, 60)
    assert lds_deep == lds_plan1 and lds_shallow == lds_plan0 and lds_plan0 != lds_plan1


def test_sample_is_the_same_job_whatever_plan_the_engine_picks(monkeypatch):
    """lmc.sample() on a job big enough for its launch schedule (4 x 100 iterations, then 500s, two launches in flight) with
    the engine free to move launches to the deep-tree LDS plan, against the same call pinned to plan 0: the same draws and
    statistics, bit for bit -- the choice (and the pacing that feeds it) is invisible in the results."""
    d, chains, tune, draws = 128, 20000, 350, 250
    tgt = T.AR1(d, 0.9)
    out = []
    for plan in ("shallow", "auto"):
        # launch_iters explicit: sample()'s own schedule depends on how many chains the device keeps resident, and a job that
        # fits the slots is ONE launch from iteration 0 -- which takes plan 0 whatever the trees (round 5's advisor)
        trace, stats, eng = lmc.sample(tgt, d, draws=draws, tune=tune, chains=chains, random_seed=99, progressbar=False,
                                       return_engine=True, launch_iters=100, lds_plan=plan)
        try:
            lds = eng.run_lds_bytes()
            assert eng.last_run_plan() == ("shallow" if plan == "shallow" else "deep")
        finally:
            eng.close()
        out.append((trace[::97].copy(), stats["tree_size"].copy(), stats["energy"][::97].copy(), lds))
        del trace, stats
    assert out[0][3] != out[1][3]            # the free engine did end up in the other plan (deep trees: AR(1) d = 128)
    np.testing.assert_array_equal(out[0][0], out[1][0])
    np.testing.assert_array_equal(out[0][1], out[1][1])
    np.testing.assert_array_equal(out[0][2], out[1][2])


def test_runtime_compiled_density_gets_the_deep_tree_plan_too(monkeypatch):
    """A user's density compiled at run time (hiprtc: the plug-in path north_star describes) hands the engine BOTH
    instantiations of the one-wave sampling kernel -- lmc_engine_load_user_kernels + lmc_engine_load_user_run_plan1 (ABI 7)
    -- so its deep trees move to the deep-tree LDS plan like the built-in densities': same chains as pinned to plan 0."""
    from tests.test_gpu_wide import USER_AR1

    d, chains, n = 128, 96, 260
    out = []
    for plan in ("0", None):
        user = T.UserTarget(d, USER_AR1, params=T.AR1(d, 0.9).params)
        res, lds = _plan_job(monkeypatch, plan, user, d, chains, n)
        out.append((res, lds))
    builtin, _lds = _plan_job(monkeypatch, "0", T.AR1(d, 0.9), d, chains, n)
    assert out[0][1] != out[1][1]                      # the free engine ended in plan 1
    np.testing.assert_array_equal(out[0][0][0], out[1][0][0])
    np.testing.assert_array_equal(out[0][0][1], out[1][0][1])
    np.testing.assert_array_equal(out[0][0][1], builtin[1])   # ... and the user's AR(1) builds the built-in AR(1)'s trees
