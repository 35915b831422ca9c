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
                              callback=cb, progressbar=False)
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
