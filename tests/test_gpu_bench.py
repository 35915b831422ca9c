"""-m gpu: bench.py itself -- the self-launcher, the multi-rank path and the fields the driver's record relies on."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "2", "--warmup", "0", "--chains", "512", "--no-cpu-baseline", "--no-secondary"]


def run_bench(extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + extra, env=env, capture_output=True,
                         text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout            # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_launches_its_own_ranks_and_the_blocks_add_up():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE: bench.py re-runs itself as 2 ranks under
    torch.distributed.run (gloo here: two ranks share this one GPU), prints one line, exits 0 -- and the two chain blocks
    together did exactly the leapfrogs of the one-process job (same global seeds, strong scaling)."""
    one = run_bench(["--gpus", "1"])
    two = run_bench(["--gpus", "2", "--backend", "gloo"])
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["leapfrogs"] == one["leapfrogs"]
    assert [r["chains"] for r in two["per_rank"]] == [256, 256]
    assert sum(r["leapfrogs"] for r in two["per_rank"]) == two["leapfrogs"]
    assert abs(two["ess_per_sec"]["ess_min"] / one["ess_per_sec"]["ess_min"] - 1.0) < 1e-9      # all-reduced == one block
    for line in (one, two):
        assert line["metric"].startswith("leapfrog-steps/sec") and line["config"]["chains_total"] == 512
        t = line["tail"]
        assert t["busiest_chain_leapfrogs"] >= t["mean_chain_leapfrogs"] > 0
        assert t["critical_path_leapfrogs"] >= t["busiest_chain_leapfrogs"] * 0.5
        assert 0 < t["implied_wall_lower_bound_s"] and t["lone_wave_us_per_leapfrog"] > 0
        assert 0 < t["mean_wave_slot_occupancy"] <= 1.05
    # N = 1 under the default backend brings up a one-rank RCCL group and says so in the line
    assert one["backend"] == "nccl" and (one["rccl_ranks"] == 1 or one["rccl_error"])
    assert one["rccl_ranks"] == 1, one["rccl_error"]


def test_bench_under_torchrun_still_works():
    """The driver's own form for N > 1: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ..."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29581", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--no-ess"] + SMALL
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2
