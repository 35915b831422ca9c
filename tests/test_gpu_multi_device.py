"""-m gpu: several GPUs behind the drop-in call. The reference fans its chains out over ``cores`` worker processes from
inside ``sample()`` (/root/reference/littlemcmc/sampling.py:124-129,186-201); here ``sample(..., devices=[...])`` deals
contiguous chain blocks to one engine per GPU, all driven from one process. The test box has ONE GPU, so the engines
share it (``devices=[0, 0]`` / ``[0, 0, 0]``): every line of the multi-device code path runs -- block dealing, global
seeds, launches enqueued on all engines before any wait, concatenation in chain order, diagnostics reduced over the
per-engine blocks -- and the result must be the one-engine result bit for bit."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import littlemcmc_amd as lmc
from littlemcmc_amd import diagnostics as dg
from littlemcmc_amd import targets as T

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _same(a, b):
    (ta, sa), (tb, sb) = a, b
    np.testing.assert_array_equal(ta, tb)
    assert set(sa) == set(sb)
    for name in sa:
        np.testing.assert_array_equal(sa[name], sb[name], err_msg=name)


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]])
def test_sample_on_several_engines_equals_the_one_engine_run(devices):
    d, chains, tune, draws = 24, 301, 120, 80           # 301 chains: uneven blocks (151 + 150; 101 + 100 + 100)
    kw = dict(draws=draws, tune=tune, chains=chains, random_seed=77, discard_tuned_samples=False, progressbar=False)
    one = lmc.sample(T.AR1(d, 0.9), d, device=0, **kw)
    many = lmc.sample(T.AR1(d, 0.9), d, devices=devices, **kw)
    _same(one, many)
    assert one[0].shape == (chains, tune + draws, d)


def test_step_object_and_diagnostics_over_a_group():
    """What sample() leaves behind -- the step object's counters and last-chain adaptation state -- and R-hat / ESS
    reduced over the per-engine blocks equal the one-engine job's."""
    d, chains, tune, draws = 16, 512, 150, 200
    tgt = T.Funnel(d)                                    # divergences: the counters are not all zero
    res = {}
    for key, dev_kw in (("one", dict(device=0)), ("two", dict(devices=[0, 0]))):
        step = lmc.NUTS(tgt, d, max_treedepth=6)
        trace, stats, eng = lmc.sample(tgt, d, draws=draws, tune=tune, chains=chains, step=step, random_seed=9,
                                       progressbar=False, return_engine=True, **dev_kw)
        try:
            x = dg.trace_tensor(eng)
            diag = dg.summarize(x)
            diag_rn = dg.summarize(x, rank_normalized=True)
            res[key] = (trace, stats, step, {k: v.cpu().numpy() for k, v in diag.items() if hasattr(v, "cpu")},
                        {k: v.cpu().numpy() for k, v in diag_rn.items() if hasattr(v, "cpu")}, type(eng).__name__)
        finally:
            eng.close()
    a, b = res["one"], res["two"]
    assert (a[5], b[5]) == ("Engine", "EngineGroup")
    _same(a[:2], b[:2])
    sa, sb = a[2], b[2]
    assert (sa._num_divs_sample, sa._samples_after_tune, sa._reached_max_treedepth, sa.iter_count, sa.tune) == \
           (sb._num_divs_sample, sb._samples_after_tune, sb._reached_max_treedepth, sb.iter_count, sb.tune)
    assert sa._num_divs_sample > 0
    np.testing.assert_array_equal(sa.potential._var, sb.potential._var)
    assert (sa.step_adapt._log_step, sa.step_adapt._count) == (sb.step_adapt._log_step, sb.step_adapt._count)
    for k in ("rhat", "ess", "mean", "var"):             # sums over blocks in another order: equal to rounding
        np.testing.assert_allclose(a[3][k], b[3][k], rtol=1e-10, atol=1e-13, err_msg=k)
        np.testing.assert_allclose(a[4][k], b[4][k], rtol=1e-9, atol=1e-12, err_msg="rank-normalised " + k)


def test_cores_and_default_pick_the_gpus(monkeypatch):
    """`cores=N` caps the GPUs (sampling.py:117-129 of the reference), the default takes as many as the chains fill; on a
    box that reports 4 devices (they all are GPU 0 here) the job is dealt accordingly and still equals the one-GPU run."""
    from littlemcmc_amd import sampling
    from littlemcmc_amd.engine import Engine, EngineGroup

    d, chains = 8, 64
    kw = dict(draws=40, tune=60, chains=chains, random_seed=3, discard_tuned_samples=False, progressbar=False)
    one = lmc.sample(T.StdNormal(d), d, device=0, **kw)
    made = []
    real_make = lmc.NUTS._make_engine

    def make_on_gpu0(self, n, device=0):                 # "device k" of the pretend 4-GPU box is GPU 0
        made.append((n, device))
        return real_make(self, n, device=0)

    monkeypatch.setattr(sampling, "visible_devices", lambda: 4)
    monkeypatch.setattr(lmc.NUTS, "_make_engine", make_on_gpu0)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    two = lmc.sample(T.StdNormal(d), d, cores=2, **kw)
    assert [m for m in made if m[0] > 1] == [(32, 0), (32, 1)]
    _same(one, two)
    made.clear()
    dflt = lmc.sample(T.StdNormal(d), d, **kw)            # 64 chains do not fill one GPU's resident slots: one GPU
    assert [m for m in made if m[0] > 1] == [(64, 0)]
    _same(one, dflt)


def test_dense_and_hmc_jobs_over_a_group():
    d = 12
    idx = np.arange(d)
    cov = 0.9 ** np.abs(idx[:, None] - idx[None, :])
    kw = dict(draws=30, tune=60, chains=40, random_seed=21, discard_tuned_samples=False, progressbar=False)
    for make_step in (lambda: lmc.NUTS(T.AR1(d, 0.9), d, potential=lmc.QuadPotentialFull(cov)),
                      lambda: lmc.NUTS(T.AR1(d, 0.9), d, potential=lmc.QuadPotentialFullAdapt(d, np.zeros(d), np.eye(d), 10)),
                      lambda: lmc.HamiltonianMC(T.AR1(d, 0.9), d, path_length=1.0)):
        one = lmc.sample(T.AR1(d, 0.9), d, step=make_step(), device=0, **kw)
        two = lmc.sample(T.AR1(d, 0.9), d, step=make_step(), devices=[0, 0], **kw)
        _same(one, two)


def _bench(extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    base = ["--steps", "2", "--warmup", "0", "--chains", "512", "--no-cpu-baseline", "--no-secondary"]
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + base + extra, env=env, capture_output=True,
                         text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    return json.loads(lines[0])


def test_bench_inproc_launcher_adds_up_to_the_one_gpu_job():
    """`bench.py --gpus 2 --launcher inproc`: the second way to produce the scaling curve -- one process, one engine per
    GPU, no process group -- does exactly the leapfrogs of the N = 1 job (same global seeds, strong scaling) and says in
    the line which launcher ran."""
    one = _bench(["--gpus", "1", "--no-rccl-check"])
    two = _bench(["--gpus", "2", "--launcher", "inproc"])
    assert (one["n_gpus"], two["n_gpus"]) == (1, 2)
    assert two["leapfrogs"] == one["leapfrogs"]
    assert [r["chains"] for r in two["per_rank"]] == [256, 256]
    assert sum(r["leapfrogs"] for r in two["per_rank"]) == two["leapfrogs"]
    assert two["launcher"].startswith("inproc") and one["launcher"].startswith("ranks")
    assert two["backend"] is None and two["rccl_ranks"] is None
    assert abs(two["ess_per_sec"]["ess_min"] / one["ess_per_sec"]["ess_min"] - 1.0) < 1e-9
    assert one["source_hash"] == two["source_hash"] == one["source_tree_hash"]          # the binary's own stamp


def test_bench_falls_back_to_the_inproc_launcher_when_rccl_does_not_come_up():
    """The driver's own form for N > 1 (torch.distributed.run, nccl). If the RCCL process group cannot be brought up, rank 0
    runs the same job in-process and the line says so -- the scaling curve does not hinge on a collective the sampling path
    never needs. (LMC_BENCH_FAIL_RCCL simulates the failure.)"""
    one = _bench(["--gpus", "1", "--no-rccl-check"])
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["LMC_BENCH_FAIL_RCCL"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29583", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0",
           "--chains", "512", "--no-cpu-baseline", "--no-secondary"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    two = json.loads(lines[0])
    assert two["n_gpus"] == 2 and two["leapfrogs"] == one["leapfrogs"]
    assert two["launcher"].startswith("inproc") and "RCCL" in two["launcher_fallback"] and two["rccl_error"]
    assert [r["chains"] for r in two["per_rank"]] == [256, 256]
