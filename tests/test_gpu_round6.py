"""-m gpu, round 6: the deep-tree LDS plan of the one-wave sampling kernels DIRECTLY under the oracle (it runs 82 % of the
headline's time and was only ever compared with the other plan), results streamed into pinned host arrays while the job
runs, and hiprtc compiling a never-seen density on the GPU box."""
import os
import subprocess
import sys
import time

import numpy as np
import pytest

import littlemcmc_amd as lmc
from littlemcmc_amd import _abi
from littlemcmc_amd import targets as T

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def _golden_sdot(monkeypatch):
    """The goldens were captured on an AVX-512 host (tests/test_gpu_parity.py: same fixture)."""
    from littlemcmc_amd import engine

    monkeypatch.setattr(engine, "DEFAULT_SDOT", "skylakex")


# ---- run_kernel<NS, 1, Target, 0, 1> (LDS plan "deep") against the reference-captured chains, every iteration -------------------
@pytest.mark.parametrize("name", ["e2e_nuts_ar1_128", "e2e_nuts_std64", "e2e_nuts_std128", "e2e_nuts_funnel256", "e2e_nuts_ar1_16",
                                  "e2e_nuts_funnel8"])
def test_every_iteration_of_the_golden_runs_under_the_deep_tree_plan(golden_dir, name):
    """nuts.py:284-342,377-417 through the kernel that runs the headline: the engine only moves launches to the deep-tree
    plan from iteration 200 on, so the suite's oracle comparisons (launches from iteration 0) all ran plan "shallow". Here
    lmc_config.lds_plan pins "deep" and every iteration of the captured reference chains -- C3's shape (AR(1), d = 128), C2's
    (d = 64), the north_star shape (d = 128), C5's (funnel d = 256, depth 12) -- is replayed from the oracle's state; the
    launch must report the plan it ran under."""
    from tests._gpu_util import replay_golden_run

    checked, fragile, total = replay_golden_run(golden_dir, name, lds_plan="deep")
    print("%s under the deep-tree plan: replay checked %d of %d iterations, %d fragile" % (name, checked, total, fragile))
    assert checked >= total - 2, (checked, fragile)


def test_rare_paths_differential_fuzz_under_the_deep_tree_plan():
    """tools/fuzz_rare.py with lds_plan = deep: far starts, step sizes up to the stability limit, weight-offset moves,
    divergences in either leaf of a pair, depth-capped trees -- the one-wave shapes run run_kernel<.., 1> (teams ignore the
    field), every statistic of the first iterations against the oracle."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_rare.py"), "60", "29", "deep"], capture_output=True,
                         text=True, timeout=900, cwd=ROOT)
    tail = "\n".join(res.stdout.strip().split("\n")[-6:])
    assert res.returncode == 0, tail + "\n" + res.stderr[-2000:]
    assert "lds_plan deep" in tail and "failures: 0" in tail


def test_pinned_plan_is_reported_and_ignored_where_there_is_one_plan_only():
    """lmc_engine_last_run_plan: "deep" / "shallow" as pinned for a one-wave kernel; a team kernel (d = 1000) accepts the field,
    launches its one plan and reports "shallow" (plan 0 is what it runs); dense engines report None."""
    for d, plan, want in ((128, "deep", "deep"), (128, "shallow", "shallow"), (1000, "deep", "shallow")):
        start, step = lmc.init_nuts(T.StdNormal(d), d, random_seed=[5, 6], lds_plan=plan)
        eng = step._make_engine(2)
        try:
            assert eng.last_run_plan() is None
            eng.seed([5, 6])
            eng.set_position(start)
            eng.reset_tuning()
            eng.reserve(4, keep_trace=False)
            eng.run(2, 0, 4)
            eng.synchronize()
            assert eng.last_run_plan() == want, (d, plan, eng.last_run_plan())
        finally:
            eng.close()
    with pytest.raises(KeyError):
        lmc.Engine(T.StdNormal(8), chains=2, lds_plan="deepest")


# ---- streamed results: lmc_engine_copy_window_async / StreamedResults / sample(stream_results=...) -----------------------------
def _same(a, b):
    np.testing.assert_array_equal(a[0], b[0])
    assert sorted(a[1]) == sorted(b[1])
    for k in a[1]:
        assert a[1][k].dtype == b[1][k].dtype and a[1][k].shape == b[1][k].shape, k
        np.testing.assert_array_equal(a[1][k], b[1][k], err_msg=k)


@pytest.mark.parametrize("kind,discard", [("nuts", True), ("nuts", False), ("hmc", True)])
def test_streamed_results_equal_the_copy_after_the_job(kind, discard):
    """sample() fills its result arrays while the job runs (every launch's window copied under the next launch, statistics
    converted to the reference's dtypes on the device) or, with stream_results=False, in one copy when the job is over:
    the same arrays bit for bit, same dtypes and shapes (sampling.py:207-222), through launch boundaries that do and do not
    coincide with the tune / draw boundary."""
    d, chains = 48, 700
    tgt = T.AR1(d, 0.9)
    kw = dict(draws=130, tune=170, chains=chains, random_seed=31, progressbar=False, discard_tuned_samples=discard, launch_iters=64)
    mk = (lambda: None) if kind == "nuts" else (lambda: lmc.HamiltonianMC(tgt, d, path_length=1.0))
    a = lmc.sample(tgt, d, step=mk(), stream_results="direct", **kw)    # the kernel writes the returned trace itself
    b = lmc.sample(tgt, d, step=mk(), stream_results=False, **kw)
    w = lmc.sample(tgt, d, step=mk(), stream_results="windows", **kw)   # trace in HBM, copied window by window
    _same(a, b)
    _same(w, b)
    n_out = 130 if discard else 300
    assert a[0].shape == (chains, n_out, d) and a[0].flags["C_CONTIGUOUS"]
    want = lmc.NUTS.stats_dtypes[0] if kind == "nuts" else lmc.HamiltonianMC.stats_dtypes[0]
    assert {k: v.dtype for k, v in a[1].items()} == {k: np.dtype(v) for k, v in want.items()}
    # the arrays outlive the engine and everything else of the call (they own their pinned memory)
    import gc

    keep = a[0][5, 7].copy()
    gc.collect()
    np.testing.assert_array_equal(a[0][5, 7], keep)


def test_streamed_results_on_two_engines_and_after_an_interrupt():
    """An EngineGroup (two engines on this one GPU, contiguous chain blocks) streams every block into its rows of the same
    arrays; a job interrupted from the callback returns the prefix every chain completed -- both equal the unstreamed path."""
    d, chains = 24, 300
    tgt = T.StdNormal(d)
    kw = dict(draws=60, tune=60, chains=chains, random_seed=8, progressbar=False, discard_tuned_samples=False, launch_iters=25)
    one = lmc.sample(tgt, d, devices=[0], stream_results=False, **kw)
    two = lmc.sample(tgt, d, devices=[0, 0], stream_results="direct", **kw)
    _same(one, two)
    _same(one, lmc.sample(tgt, d, devices=[0, 0], stream_results="windows", **kw))
    # the team kernels (d = 600: two wavefronts per chain) and the general kernels (d = 1100) store their draws the same way
    for dd in (600, 1100):
        kk = dict(draws=6, tune=10, chains=5, random_seed=4, progressbar=False, launch_iters=7)
        _same(lmc.sample(T.StdNormal(dd), dd, stream_results="direct", **kk), lmc.sample(T.StdNormal(dd), dd, stream_results=False, **kk))

    def stop_at_50(trace, draw):
        if draw.iteration >= 50:
            raise KeyboardInterrupt

    # an engine kept after a "direct" job reads its draws through the caller's array (lmc_engine_get_trace on an attached trace)
    tr, _st, eng = lmc.sample(tgt, d, devices=[0], stream_results="direct", return_engine=True, **kw)
    try:
        np.testing.assert_array_equal(eng.trace(0, 120), tr)
        np.testing.assert_array_equal(tr, one[0])
    finally:
        eng.close()
    with pytest.raises(ValueError):
        lmc.sample(tgt, d, stream_results="sideways", **kw)
    # tiny results, many of them, arrays dropped in any order: every pinned array owns whole pages (a small numpy allocation
    # shares its pages with its neighbours -- unregistering one unpinned memory the device was still writing: a GPU fault that
    # killed the process, intermittently, until pinned_empty() took page-exclusive memory)
    keep = []
    for i in range(40):
        tr_, st_ = lmc.sample(T.StdNormal(3), 3, draws=4 + i % 3, tune=3, chains=2 + i % 5, random_seed=i, progressbar=False,
                              stream_results="direct")
        keep.append((tr_, st_))
        if i % 3 == 0:
            del keep[i // 2]
        assert np.isfinite(tr_).all()
    # (a job long enough to be still running whenever the stop arrives -- 20 060 iterations, 1.6 GB of pinned results of which
    #  a prefix comes back; its first 120 iterations are the short job's: a chain is a function of its seed, start and tune)
    got = lmc.sample(tgt, d, stream_results="direct", callback=stop_at_50, **dict(kw, draws=20000))
    n = got[0].shape[1]
    assert 25 <= n < 20060, n
    m = min(n, 120)
    np.testing.assert_array_equal(got[0][:, :m], one[0][:, :m])
    for k in got[1]:
        assert got[1][k].shape == (chains, n, 1)
        np.testing.assert_array_equal(got[1][k][:, :m], one[1][k][:, :m], err_msg=k)


def test_copy_window_async_runs_under_the_next_launch():
    """The C ABI itself: windows enqueued right behind their launches, destination pinned (lmc_host_alloc); the host is not
    blocked by the enqueue (it returns long before the launch it follows is over), the copies are complete after copy_wait(),
    and they equal lmc_engine_get_trace / get_stat_*."""
    from littlemcmc_amd.engine import StreamedResults

    d, chains, n = 128, 16000, 150
    tgt = T.AR1(d, 0.9)
    seeds = lmc.distributed.global_seeds(3, chains)
    start, step = lmc.init_nuts(tgt, d, random_seed=seeds)
    eng = step._make_engine(chains)
    try:
        eng.seed(seeds)
        eng.set_position(start)
        eng.reset_tuning()
        eng.reserve(n, keep_trace=True, trace_begin=0)
        out = StreamedResults(chains, n, 0, d, step._result_planes())
        assert out.pinned and out.trace.flags["C_CONTIGUOUS"]
        t0 = time.perf_counter()
        for first in range(0, n, 50):
            eng.run(n // 2, first, 50)
            eng.copy_window_async(out, first, 50)
        t_enq = time.perf_counter() - t0
        eng.synchronize()
        t_job = time.perf_counter() - t0
        eng.copy_wait()
        assert t_enq < 0.5 * t_job, (t_enq, t_job)          # enqueueing did not wait for kernels or copies
        np.testing.assert_array_equal(out.trace, eng.trace(0, n))
        raw = step._stats_from_engine(eng, 0, n)
        for name, dt in step.stats_dtypes[0].items():
            got = out.stats[name]
            assert got.dtype == np.dtype(dt), name
            np.testing.assert_array_equal(got, raw[name].astype(dt), err_msg=name)
        # a window outside the destination is refused, not written
        with pytest.raises(_abi.HipLibraryError, match="outside"):
            eng.copy_window_async(StreamedResults(chains, 10, 50, d, [], pinned=False), 40, 20)
        # ... and so is a pageable trace destination, which leaves the engine's own trace in place
        with pytest.raises(_abi.HipLibraryError, match="not device-accessible"):
            eng.attach_trace(np.empty((chains, n, d)), 0)
        np.testing.assert_array_equal(out.trace[::501], eng.trace(0, n)[::501])
        # pageable memory is refused (the copies are kernels that write the destination themselves), never silently staged
        with pytest.raises(_abi.HipLibraryError, match="not device-accessible"):
            eng.copy_window_async(StreamedResults(chains, 10, 50, d, [], pinned=False), 50, 10)
    finally:
        eng.close()


# ---- hiprtc on THIS box -----------------------------------------------------------------------------------------------------
def test_hiprtc_compiles_a_never_seen_density_on_this_box():
    """A user's density whose source carries a per-run nonce: the content-addressed cache of code objects cannot hold it, so
    hiprtc compiles it here, on the GPU box (the code objects the build container produced are not shipped: .gpurunignore,
    and the session fixture empties the cache directory). Sampled, and checked against the oracle driven by the numpy
    statement of the same density."""
    from oracle import lmc_oracle as orc
    from tests.test_gpu_reference_suite import USER_SRC

    nonce = "%016x" % int.from_bytes(os.urandom(8), "little")
    src = USER_SRC + "\n// nonce %s\n" % nonce
    d = 24
    mu = np.linspace(-1.0, 2.0, d)
    cache = os.path.join(ROOT, "littlemcmc_amd", "_user_targets")
    before = set(os.listdir(cache)) if os.path.isdir(cache) else set()
    tgt = T.UserTarget(d, src, params=mu)
    t0 = time.perf_counter()
    gt, gst = lmc.sample(tgt, d, draws=5, tune=25, chains=2, random_seed=5, discard_tuned_samples=False, progressbar=False)
    dt = time.perf_counter() - t0
    new = [f for f in set(os.listdir(cache)) - before if f.endswith(".hsaco")]
    assert new, "no code object was produced: the density did not go through hiprtc on this box"
    f = lambda x: (-0.5 * np.dot(x - mu, x - mu), -(x - mu))   # noqa: E731
    ot, ost = orc.sample(f, d, draws=5, tune=25, chains=2, random_seed=5, discard_tuned_samples=False)
    n = 15
    np.testing.assert_array_equal(gst["depth"][:, :n], ost["depth"][:, :n])
    np.testing.assert_array_equal(gst["tree_size"][:, :n], ost["tree_size"][:, :n])
    np.testing.assert_allclose(gt[:, :n], ot[:, :n], rtol=1e-6, atol=1e-8)
    print("hiprtc compiled %s in a %.1f s sample() call on this box" % (new, dt))
