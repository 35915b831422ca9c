"""Host logic that needs no GPU: the job loop of sample() (progress, Ctrl-C) against a stand-in engine, and bench.py's
self-launcher command line."""
import logging
import os
import sys
import types

import numpy as np
import pytest

from littlemcmc_amd import _abi, sampling

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class FakeEngine:
    """Records what _run_job asks of an engine; `interrupt_at_sync` raises KeyboardInterrupt from the first wait."""

    def __init__(self, chains=4, done_when_stopped=37, interrupt_at_sync=False):
        self.chains = chains
        self.target = types.SimpleNamespace(family=_abi.TARGET_STD_NORMAL)
        self.cfg = types.SimpleNamespace(device=0)
        self.calls = []
        self._done = done_when_stopped
        self._interrupt = interrupt_at_sync

    def run_streams(self):
        return [0]          # no GPU here: torch.cuda.ExternalStream fails and _run_job takes the blocking wait

    def run(self, tune, it, n):
        self.calls.append(("run", tune, it, n))

    def synchronize(self):
        self.calls.append(("sync",))
        if self._interrupt:
            self._interrupt = False
            raise KeyboardInterrupt

    def request_stop(self, stop=True):
        self.calls.append(("stop", bool(stop)))

    def completed_iterations(self):
        return self._done


def test_job_loop_enqueues_back_to_back_and_waits_once():
    eng = FakeEngine()
    n_done, interrupted = sampling._run_job(eng, tune=50, n_total=230, per_launch=100, progressbar=True)
    assert (n_done, interrupted) == (230, False)
    assert [c for c in eng.calls if c[0] == "run"] == [("run", 50, 0, 100), ("run", 50, 100, 100), ("run", 50, 200, 30)]
    assert eng.calls[-1] == ("sync",) and ("stop", True) not in eng.calls


def test_keyboard_interrupt_stops_the_device_and_reports_what_every_chain_completed(caplog):
    """sampling.py:324-328 / :470-471 of the reference: Ctrl-C ends sampling, the draws so far are kept."""
    eng = FakeEngine(done_when_stopped=37, interrupt_at_sync=True)
    with caplog.at_level(logging.WARNING, logger="littlemcmc_amd"):
        n_done, interrupted = sampling._run_job(eng, tune=50, n_total=230, per_launch=100, progressbar=False)
    assert (n_done, interrupted) == (37, True)
    kinds = [c[0] for c in eng.calls]
    i = kinds.index("stop")
    assert eng.calls[i] == ("stop", True) and "sync" in kinds[i:]          # stop request, then wait for the kernels to drain
    assert eng.calls[-1] == ("stop", False)                                  # re-armed for whoever keeps the engine
    assert "interrupted after 37 of 230" in caplog.text


def test_bench_self_launcher_command_line(monkeypatch):
    """`python bench.py --gpus 4 ...` with no WORLD_SIZE re-runs itself under torch.distributed.run on 127.0.0.1."""
    sys.path.insert(0, ROOT)
    import importlib

    bench = importlib.import_module("bench")
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0

    import subprocess

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    assert bench.self_launch(4) == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_chain_block_seeds_are_prefix_stable():
    from littlemcmc_amd.distributed import chain_block, global_seeds

    seeds = global_seeds(20260928, 64)
    assert seeds[:8] == global_seeds(20260928, 8)
    blocks = [chain_block(64, r, 8) for r in range(8)]
    assert sum((seeds[a:b] for a, b in blocks), []) == seeds


# ---- several GPUs behind sample(): device choice and the engine group (no GPU needed: stand-in engines) --------------
def test_device_choice_follows_the_references_cores_logic(monkeypatch):
    """sampling.py:117-129 of the reference with GPUs in the place of worker processes."""
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sampling, "visible_devices", lambda: 8)
    slots = lambda: 3072   # noqa: E731
    R = sampling._resolve_devices
    assert R([3, 3], None, None, 100, slots) == [3, 3]                 # explicit, repeats allowed (two engines on one GPU)
    assert R(4, None, None, 100, slots) == [0, 1, 2, 3]                # devices=N: the first N
    assert R(None, 5, None, 100000, slots) == [5]                      # device=k pins one GPU (launcher ranks)
    assert R(None, None, 2, 100000, slots) == [0, 1]                   # cores=N: at most N GPUs
    assert R(None, None, 64, 100000, slots) == list(range(8))          # ... of those that exist
    assert R(None, None, None, 65536, slots) == list(range(8))         # default: as many GPUs as the chains fill
    assert R(None, None, None, 8192, slots) == [0, 1]                  # 8192 // 3072 = 2: no GPU below full residency
    assert R(None, None, None, 64, slots) == [0]
    assert R(4, None, None, 3, slots) == [0, 1, 2]                     # never more devices than chains
    monkeypatch.setattr(sampling, "visible_devices", lambda: 1)
    assert R(None, None, None, 65536, slots) == [0]
    monkeypatch.setenv("WORLD_SIZE", "8")                              # under a process-per-GPU launcher: this rank's GPU
    monkeypatch.setenv("LOCAL_RANK", "6")
    monkeypatch.setattr(sampling, "visible_devices", lambda: 8)
    assert R(None, None, None, 65536, slots) == [6]
    with pytest.raises(ValueError):
        R([], None, None, 4, slots)


class FakeBlockEngine(FakeEngine):
    """A stand-in that also holds per-chain arrays, so that the group's slicing / concatenation can be checked."""

    def __init__(self, chains, device, done):
        super().__init__(chains=chains, done_when_stopped=done)
        self.cfg = types.SimpleNamespace(device=device, chains=chains)
        self.dim, self.kind, self.potential = 3, "nuts", "diag_adapt"
        self.seeds = self.q = None

    def seed(self, s):
        self.seeds = np.asarray(s).copy()

    def set_position(self, q):
        self.q = np.broadcast_to(np.asarray(q, dtype="d"), (self.chains, self.dim)).copy()

    def trace(self, lo=None, n=None):
        return self.q[:, None, :] + np.arange(n)[None, :, None]

    def status(self):
        return np.zeros(self.chains, dtype=np.int32)

    def counters(self):
        return np.tile(self.seeds[:, None].astype(np.int64), (1, _abi.NUM_COUNTERS))

    def adapt_state(self):
        return {"var": self.q.astype(np.float32), "count": self.seeds.astype(np.int32)}

    def dense_chain(self, chain=0):
        return ("cov", int(self.seeds[chain]))

    def close(self):
        self.calls.append(("close",))


def test_engine_group_deals_chain_blocks_and_concatenates_in_chain_order():
    from littlemcmc_amd.distributed import chain_block
    from littlemcmc_amd.engine import EngineGroup

    chains, devs = 11, [0, 1, 1]
    blocks = [chain_block(chains, k, len(devs)) for k in range(len(devs))]
    engines = [FakeBlockEngine(hi - lo, dv, done=40 + 7 * k) for k, (dv, (lo, hi)) in enumerate(zip(devs, blocks))]
    g = EngineGroup(engines, blocks)
    assert (g.chains, g.devices) == (chains, devs)
    seeds = np.arange(100, 100 + chains)
    q0 = np.arange(chains * 3, dtype="d").reshape(chains, 3)
    g.seed(seeds)
    g.set_position(q0)
    for e, (lo, hi) in zip(engines, blocks):          # every engine got ITS slice of the global arrays
        np.testing.assert_array_equal(e.seeds, seeds[lo:hi])
        np.testing.assert_array_equal(e.q, q0[lo:hi])
    np.testing.assert_array_equal(g.trace(0, 2)[:, 0], q0)             # results come back in global chain order
    np.testing.assert_array_equal(g.counters()[:, 0], seeds)
    np.testing.assert_array_equal(g.adapt_state()["count"], seeds)
    assert g.dense_chain(chains - 1) == ("cov", int(seeds[-1])) and g.dense_chain(4) == ("cov", int(seeds[4]))
    assert g.completed_iterations() == 40                               # what EVERY chain of EVERY block completed
    g.set_position(q0[0])                                               # one start for all chains
    assert all((e.q == q0[0]).all() for e in engines)
    # the job loop drives the group like one engine: every launch on every device, one wait, stop reaches all of them
    n_done, interrupted = sampling._run_job(g, tune=5, n_total=25, per_launch=10, progressbar=False)
    assert (n_done, interrupted) == (25, False)
    for e in engines:
        assert [c for c in e.calls if c[0] == "run"] == [("run", 5, 0, 10), ("run", 5, 10, 10), ("run", 5, 20, 5)]
    engines[1]._interrupt = True
    n_done, interrupted = sampling._run_job(g, tune=5, n_total=100, per_launch=50, progressbar=False)
    assert (n_done, interrupted) == (40, True)
    for e in engines:
        assert ("stop", True) in e.calls and e.calls[-1] == ("stop", False)
    g.close()
    assert all(e.calls[-1] == ("close",) for e in engines)


def test_the_build_hash_is_compiled_into_the_binary(tmp_path):
    """needs_build() compares the stamp INSIDE the library with the sources as they are (not mtimes), and the loaded
    library reports the same stamp through the C ABI: a stale binary cannot pass for the tree's."""
    from littlemcmc_amd import _build

    lib = _abi.load()
    stamp = lib.lmc_build_hash().decode()
    assert stamp == _build.binary_hash() and len(stamp) == 16 and stamp != "unstamped"
    assert (stamp == _build.source_hash()) == (not _build.needs_build())
    # a binary without a stamp, or with another build's stamp, needs a build whatever its mtime says
    fake = tmp_path / "liblmc_hip.so"
    fake.write_bytes(b"\x7fELF....LMC_BUILD_HASH=0123456789abcdef\0....")
    assert _build.binary_hash(str(fake)) == "0123456789abcdef" and _build.needs_build(str(fake))
    fake.write_bytes(b"\x7fELF no stamp at all")
    assert _build.binary_hash(str(fake)) is None and _build.needs_build(str(fake))
    assert _build.source_hash(("-DX",)) != _build.source_hash()          # private builds carry their own identity


def test_a_chain_may_only_part_from_its_oracle_twin_for_a_reason():
    """tests/_gpu_util.py: explain_first_difference / assert_chain_matches (round 5). The rule the -m gpu parity tests apply to
    a device chain's first difference from the oracle, pinned on synthetic chains: a jump out of nowhere has no reason and
    fails; geometric growth, a fragile decision, a U-turn at reduction noise and a float32-ulp energy difference are reasons."""
    import pytest

    from tests._gpu_util import FRAGILE, RTOL_Q, assert_chain_matches

    n, d = 30, 4
    rs = np.random.RandomState(0)
    want_q = rs.randn(n, d) + 3.0
    stats = {"depth": np.full(n, 3), "tree_size": np.full(n, 7), "step_size": np.full(n, 0.5), "energy": np.full(n, 10.0)}
    calm = np.full((n, 3), 0.3)                      # no decision anywhere near its threshold

    def chain(err_from, err_rel, step_rel=None):
        got_q = want_q.copy()
        got_q[err_from:] *= 1.0 + err_rel
        got = {k: v.copy() for k, v in stats.items()}
        if step_rel is not None:
            got["step_size"][step_rel[0]:] *= 1.0 + step_rel[1]
        return got_q, got

    # identical chains: all iterations verified
    assert assert_chain_matches(want_q.copy(), dict(stats), want_q, stats, calm) == n
    # positions 1e-6 apart from iteration 12 on, nothing before: no reason -> fails
    got_q, got = chain(12, 1e-6)
    with pytest.raises(AssertionError, match="no reason"):
        assert_chain_matches(got_q, got, want_q, stats, calm)
    # the same jump with the step size already 1e-9 apart the iteration before: geometric growth -> accepted, prefix 12
    got_q, got = chain(12, 1e-6, step_rel=(11, 1e-9))
    assert assert_chain_matches(got_q, got, want_q, stats, calm) == 12
    # growth is geometric, not arbitrary: from 2e-4 of the tolerance to 1000 times the tolerance in ONE iteration (the largest
    # ratio any captured chain shows is 250) is a jump again, whatever the iteration before looked like (round 6: JUMP)
    got_q, got = chain(12, 1e-4, step_rel=(11, 2e-11))
    with pytest.raises(AssertionError, match="no reason"):
        assert_chain_matches(got_q, got, want_q, stats, calm)
    # ... or with a multinomial decision within FRAGILE of its threshold at iteration 5
    fragile = calm.copy()
    fragile[5, 0] = 0.1 * FRAGILE
    got_q, got = chain(12, 1e-6)
    assert assert_chain_matches(got_q, got, want_q, stats, fragile) == 12
    # ... or a U-turn dot product at reduction-order noise
    turn = calm.copy()
    turn[9, 1] = 1e-12
    assert assert_chain_matches(got_q, got, want_q, stats, turn) == 12
    # ... or the energy of that very iteration a float32 ulp apart (another host's sdot)
    got_e = {k: v.copy() for k, v in got.items()}
    got_e["energy"][12] *= 1.0 + 5e-7
    assert assert_chain_matches(got_q, got_e, want_q, stats, calm) == 12
    # an integer statistic that differs out of nowhere fails too
    got_i = {k: v.copy() for k, v in stats.items()}
    got_i["tree_size"][20] = 15
    with pytest.raises(AssertionError, match="no reason"):
        assert_chain_matches(want_q.copy(), got_i, want_q, stats, calm)
    assert RTOL_Q == 1e-7


def test_bench_line_is_compact_on_a_default_shaped_result():
    """Round 5's default run printed a 24 KB line and the driver's record lost it (BENCH_r05.parsed = null). The line
    bench.py prints now is built by compact_line(): at most 8192 bytes on that very result object (kept under profiles/),
    with the headline, `roofline`, `cpu_baseline` and a short block per secondary workload."""
    import importlib
    import json

    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    verbose = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_driver_form.json")))
    assert len(json.dumps(verbose)) > 20000
    line = bench.compact_line(verbose, "bench_detail.json")
    text = json.dumps(line)
    assert len(text) + 1 <= 8192 and len(text) < 6000, len(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["value"] == verbose["value"] and line["config"]["workload"].startswith("C3")
    r = line["roofline"]
    assert r["bound"] == "fp64_valu" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5 and r["traffic"] is not None
    assert line["cpu_baseline"]["cores"] == verbose["cpu_baseline"]["cores"] and line["cpu_baseline"]["kind"] == "port"
    assert len(line["secondary"]) == len(verbose["secondary"]) == 6
    labels = [s_["workload"] for s_ in line["secondary"]]
    assert [lb.split(":")[0] for lb in labels] == ["north_star shape", "north_star shape", "C2", "C4", "C5", "C5 (one launch)"]
    for s_ in line["secondary"]:
        assert s_["value"] > 0 and 0 < s_["roofline"]["frac"] < 1 and "mean_wave_slot_occupancy" in s_["tail"]
    assert not any(isinstance(v, str) and len(v) > 170 for v in json.loads(text).values())

    # eight ranks with long error texts still fit; the headline blocks are never what gets dropped
    verbose["per_rank"] = [dict(verbose["per_rank"][0], rank=r_) for r_ in range(8)]
    verbose["rccl_error"] = "x" * 2000
    verbose["launcher_fallback"] = "y" * 2000
    line8 = bench.compact_line(verbose, "bench_detail.json")
    assert len(json.dumps(line8)) + 1 <= 8192 and len(line8["per_rank"]) == 8
    assert "roofline" in line8 and "cpu_baseline" in line8
    tiny = bench.compact_line(verbose, "bench_detail.json", limit=3000)
    assert len(json.dumps(tiny)) + 1 <= 3000 and tiny["roofline"]["frac"] == line["roofline"]["frac"] and "cpu_baseline" in tiny


class _StreamFakeEngine(FakeEngine):
    """FakeEngine + the streamed-results calls: records attach_trace / copy_window_async / copy_wait in call order."""

    def attach_trace(self, out, trace_begin):
        self.calls.append(("attach", None if out is None else "array", trace_begin))

    def copy_window_async(self, out, first, n):
        self.calls.append(("copy", first, n))

    def copy_wait(self):
        self.calls.append(("copy_wait",))


@pytest.mark.parametrize("pin_fails", [False, True])
def test_result_streamer_orders_attach_launch_and_window_copies(monkeypatch, pin_fails):
    """_ResultStreamer (sampling.py): the returned arrays are pinned by a helper thread; the trace is attached BEFORE the first
    launch that reaches `first` is enqueued (the kernel arguments of a launch are fixed when it is enqueued); every launch's
    window of statistics is copied right behind its launch, clipped to [first, ...); if the arrays cannot be pinned the trace
    goes to HBM instead (attach_trace(None)), nothing is copied per window, and finish() says so by returning None."""
    from littlemcmc_amd import engine as engine_mod

    import threading

    main = threading.current_thread()
    where = {}

    class FakeResults:
        def __init__(self, chains, n_out, first, dim, planes, direct=False, register=True):
            assert register is False                              # the helper thread allocates and pre-faults only ...
            where["alloc"] = threading.current_thread()
            self.trace, self.n_out, self.first, self.registered = object(), n_out, first, False

        def register(self):                                       # ... the HIP call comes from the thread that drives the engine
            where["register"] = threading.current_thread()
            if pin_fails:
                raise _abi.HipLibraryError("cannot pin")
            self.registered = True

    monkeypatch.setattr(engine_mod, "StreamedResults", FakeResults)
    eng = _StreamFakeEngine()
    st = sampling._ResultStreamer(eng, chains=4, n_out=130, first=100, dim=3, planes=[], direct=True)
    n_done, interrupted = sampling._run_job(eng, tune=100, n_total=230, per_launch=60, progressbar=False,
                                            on_enqueued=st.window, before_enqueue=st.before_launch)
    out = st.finish()
    assert (n_done, interrupted) == (230, False)
    assert where["alloc"] is not main and where["register"] is main
    runs = [i for i, c in enumerate(eng.calls) if c[0] == "run"]
    attach = [i for i, c in enumerate(eng.calls) if c[0] == "attach"]
    assert [eng.calls[i] for i in runs] == [("run", 100, 0, 60), ("run", 100, 60, 60), ("run", 100, 120, 60), ("run", 100, 180, 50)]
    # attached exactly once, after the launch [0, 60) (which never reaches iteration 100) and before the launch [60, 120)
    assert len(attach) == 1 and runs[0] < attach[0] < runs[1]
    assert eng.calls[attach[0]] == ("attach", None if pin_fails else "array", 100)
    copies = [c for c in eng.calls if c[0] == "copy"]
    if pin_fails:
        assert out is None and copies == [] and ("copy_wait",) not in eng.calls
    else:
        assert out is not None and copies == [("copy", 100, 20), ("copy", 120, 60), ("copy", 180, 50)]   # clipped to >= first
        assert eng.calls[-1] == ("copy_wait",)
        # every window's copy follows its own launch
        for c in copies:
            assert eng.calls.index(c) > eng.calls.index(("run", 100, c[1] if c[1] != 100 else 60, 60 if c[1] != 180 else 50))


def test_pinned_arrays_own_whole_pages_and_register_where_they_are_told():
    """engine.pinned_empty (no GPU: a stand-in for the two library calls): the block handed to lmc_host_register starts on a
    page boundary, spans whole pages and covers the array -- so no two pinned arrays share a page, whatever their size --;
    register=False allocates and pre-faults only, pinned_register() makes the call later (sampling._ResultStreamer: on the
    driving thread), and the block is unregistered exactly once, when the last view of the array is gone."""
    import gc
    import mmap

    from littlemcmc_amd import engine

    class Lib:
        def __init__(self):
            self.reg, self.unreg = [], []

        def lmc_host_register(self, p, n):
            self.reg.append((p.value, n))
            return 0

        def lmc_host_unregister(self, p):
            self.unreg.append(p.value)
            return 0

    lib, page = Lib(), mmap.PAGESIZE
    a = engine.pinned_empty((3, 5, 7), np.float64, lib=lib, threads=4)
    b = engine.pinned_empty((11,), np.bool_, lib=lib, threads=4)
    assert len(lib.reg) == 2
    for arr, (base, span) in zip((a, b), lib.reg):
        assert base % page == 0 and span % page == 0 and base <= arr.ctypes.data and arr.ctypes.data + arr.nbytes <= base + span
        assert arr.flags["C_CONTIGUOUS"] and arr.flags["WRITEABLE"]
    (b0, s0), (b1, s1) = lib.reg
    assert b0 + s0 <= b1 or b1 + s1 <= b0                      # disjoint page ranges
    a[:] = 2.0
    b[:] = True
    view = a[:, :2, None]
    del a
    gc.collect()
    assert lib.unreg == [] and view.sum() == 2.0 * 3 * 2 * 7   # a view keeps the block pinned
    del view, b, arr                                            # (the loop variable above still named the second array)
    gc.collect()
    assert sorted(lib.unreg) == sorted(x for x, _ in lib.reg)
    # deferred registration
    c = engine.pinned_empty((1000,), np.float64, lib=lib, register=False)
    assert len(lib.reg) == 2
    engine.pinned_register(c[10:20])                            # found through any view
    engine.pinned_register(c)                                   # idempotent
    assert len(lib.reg) == 3 and lib.reg[2][1] == 2 * page
    del c
    gc.collect()
    assert len(lib.unreg) == 3
    assert engine.pinned_empty((0, 4), np.float64, lib=lib).shape == (0, 4) and len(lib.reg) == 3   # nothing to pin
