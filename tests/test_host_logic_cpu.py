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
