"""-m gpu: diagnostics computed on the draws where they live (HBM, zero-copy view of the engine's trace) equal
the numpy restatement on the same draws copied to the host."""
import numpy as np
import pytest

import littlemcmc_amd as lmc
from littlemcmc_amd import diagnostics as dg
from oracle import diagnostics_oracle as odg

pytestmark = pytest.mark.gpu


def test_device_diagnostics_match_numpy_on_real_draws():
    d, chains, tune, draws = 6, 16, 150, 120
    trace, stats, eng = lmc.sample(lmc.targets.AR1(d, 0.9), d, draws=draws, tune=tune, chains=chains, random_seed=8,
                                   return_engine=True)
    try:
        x = dg.trace_tensor(eng)
        assert tuple(x.shape) == (chains, draws, d) and x.is_cuda
        np.testing.assert_array_equal(x.cpu().numpy(), trace)          # the view IS the engine's trace
        got = dg.summarize(x)
        rhat, ess = odg.rhat_ess(trace)
        np.testing.assert_allclose(got["rhat"].cpu().numpy(), rhat, rtol=1e-9)
        np.testing.assert_allclose(got["ess"].cpu().numpy(), ess, rtol=1e-7)
        assert np.all(rhat < 1.1) and np.all(ess > 50)
    finally:
        eng.close()


@pytest.mark.parametrize("chains,n,d,rho", [(7, 61, 5, 0.2), (33, 200, 130, 0.6), (12, 300, 64, 0.97), (3, 16, 1, -0.5)])
def test_hip_chain_statistics_kernel_against_numpy(chains, n, d, rho):
    """lmc_diag_chain_stats (csrc/lmc_diag.hip) on synthetic AR(1) series: ragged sizes (n not a multiple of the 16-draw
    block, d over several 64-lane slabs), slow mixing (several lag passes), split and unsplit, rank-normalised."""
    import torch

    from tests.test_diagnostics_cpu import ar1_chains

    x = ar1_chains(chains, n, d, rho, 21)
    xd = torch.from_numpy(x).cuda()
    for split in (True, False):
        got = dg.summarize(xd, split=split)
        rhat, ess = odg.rhat_ess(x, do_split=split)
        np.testing.assert_allclose(got["rhat"].cpu().numpy(), rhat, rtol=1e-9)
        np.testing.assert_allclose(got["ess"].cpu().numpy(), ess, rtol=1e-7)
        host = dg.summarize(torch.from_numpy(x), split=split, stats_fn=odg.torch_chain_stats)   # the oracle's FFT restatement the gloo tests inject
        assert host["lag_passes"] == got["lag_passes"]
        np.testing.assert_allclose(got["ess"].cpu().numpy(), host["ess"].numpy(), rtol=1e-9)
    if rho > 0.9:
        assert got["lag_passes"] > 2
    # one pass of the raw statistics, lags 16..31, against direct sums
    blk = dg.chain_stats_pass(xd, [(3, n - 5)], 16).cpu().numpy()
    sub = x[:, 3:3 + n - 5]
    cen = sub - sub.mean(axis=1, keepdims=True)
    m = n - 5
    for k in (0, 5, 15):
        lag = 16 + k
        want = (cen[:, :m - lag] * cen[:, lag:]).sum(axis=(0, 1)) / m if lag < m else np.zeros(d)
        np.testing.assert_allclose(blk[3 + k], want, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(blk[0], sub.mean(axis=1).sum(axis=0), rtol=1e-12)
    z = dg.summarize(xd, rank_normalized=True)
    rhat, ess = odg.rhat_ess(x, rank_normalized=True)
    np.testing.assert_allclose(z["rhat"].cpu().numpy(), rhat, rtol=1e-8)
    np.testing.assert_allclose(z["ess"].cpu().numpy(), ess, rtol=1e-6)


def test_hip_chain_statistics_are_bit_reproducible():
    import torch

    from tests.test_diagnostics_cpu import ar1_chains

    xd = torch.from_numpy(ar1_chains(5000, 64, 70, 0.4, 2)).cuda()
    a = dg.chain_stats_pass(xd, [(0, 32), (32, 32)], 0)
    b = dg.chain_stats_pass(xd, [(0, 32), (32, 32)], 0)
    assert torch.equal(a, b)


def test_sample_distributed_single_rank_equals_sample():
    d = 5
    tgt = lmc.targets.StdNormal(d)
    tr, st, diag = lmc.distributed.sample_distributed(tgt, d, draws=40, tune=40, chains=6, random_seed=77)
    tr2, st2 = lmc.sample(tgt, d, draws=40, tune=40, chains=6, random_seed=77)
    np.testing.assert_array_equal(tr, tr2)
    np.testing.assert_array_equal(st["tree_size"], st2["tree_size"])
    assert diag["rhat"].shape == (d,) and diag["n_chains"] == 12.0


def test_running_moments_match_the_trace_and_give_rhat_without_one():
    """lmc_engine_keep_moments: per-chain Welford mean / M2 of the post-warm-up draws, accumulated inside the
    transition kernel, equal the same statistics of the trace; R-hat follows from them with no trace in HBM."""
    d, chains, tune, draws = 70, 48, 120, 150          # d = 70: padded lanes, 2 elements per thread
    seeds = np.arange(chains, dtype=np.uint32) + 11
    start = np.zeros((chains, d))

    def run(keep_trace):
        eng = lmc.Engine(lmc.targets.AR1(d, 0.5), chains=chains)
        eng.keep_moments(True)
        eng.seed(seeds)
        eng.set_position(start)
        eng.reset_tuning()
        eng.reserve(tune + draws, keep_trace=keep_trace, trace_begin=tune)
        eng.run(tune, 0, 100)                          # launch boundaries inside warm-up and inside the draws
        eng.run(tune, 100, 100)
        eng.run(tune, 200, tune + draws - 200)
        eng.synchronize()
        return eng

    with run(True) as eng:
        trace = eng.trace(tune, draws)
        mean, m2, n = eng.moments()
    assert np.all(n == draws)
    np.testing.assert_allclose(mean, trace.mean(axis=1), rtol=0, atol=1e-12)
    np.testing.assert_allclose(m2, ((trace - trace.mean(axis=1, keepdims=True)) ** 2).sum(axis=1), rtol=1e-10)

    with run(False) as eng:                            # no trace at all: same moments (same seeds, same chains)
        mean2, m22, n2 = eng.moments()
        eng.reset_tuning()
        assert np.all(eng.moments()[2] == 0)           # reset with the tuning state
    np.testing.assert_array_equal(mean2, mean)
    np.testing.assert_array_equal(m22, m2)

    rhat = dg.rhat_from_moments(mean, m2, n).numpy()
    want, _ = odg.rhat_ess(trace, do_split=False)
    np.testing.assert_allclose(rhat, want, rtol=1e-9)
    assert np.all(rhat < 1.1)


def test_moments_require_opt_in():
    with lmc.Engine(lmc.targets.StdNormal(4), chains=2) as eng:
        with pytest.raises(lmc._abi.HipLibraryError, match="keep_moments"):
            eng.moments()


def test_sample_distributed_with_moment_diagnostics():
    """diagnostics="moments": R-hat from the running moments the kernel keeps, no trace involved; equals the plain
    (non-split) R-hat of the returned draws."""
    d = 7
    tgt = lmc.targets.AR1(d, 0.5)
    tr, st, diag = lmc.distributed.sample_distributed(tgt, d, draws=120, tune=100, chains=12, random_seed=3,
                                                      diagnostics="moments")
    want, _ = odg.rhat_ess(tr, do_split=False)
    np.testing.assert_allclose(diag["rhat"], want, rtol=1e-9)
    assert diag["n_chains"] == 12.0


TWO_RANK_WORKER = """
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
import littlemcmc_amd as lmc
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank = dist.get_rank()
d, chains = 16, 11                                   # 11 chains over 2 ranks: blocks of 6 and 5
tgt = lmc.targets.AR1(d, 0.9)
tr, st, diag = lmc.distributed.sample_distributed(tgt, d, draws=60, tune=80, chains=chains, random_seed=20260928, device=0)
tr2, st2, diag2 = lmc.distributed.sample_distributed(tgt, d, draws=60, tune=80, chains=chains, random_seed=20260928,
                                                     device=0, diagnostics="moments", discard_tuned_samples=False)
tr3, st3, diag3 = lmc.distributed.sample_distributed(tgt, d, draws=60, tune=80, chains=chains, random_seed=20260928, device=0,
                                                     diagnostics="rank_normalized")
# more ranks than chains: rank 1 owns nothing and still joins every collective (discard_tuned_samples=False: its
# placeholder must carry draws + tune rows like the other rank's trace)
tr4, st4, diag4 = lmc.distributed.sample_distributed(tgt, d, draws=40, tune=40, chains=1, random_seed=5, device=0,
                                                     discard_tuned_samples=False)
assert tr4.shape == ((1, 80, d) if rank == 0 else (0, 80, d)) and diag4["n_chains"] == 2.0
np.savez({out!r} + "/rank%d.npz" % rank, trace=tr, tree_size=st["tree_size"], depth=st["depth"], rhat=diag["rhat"],
         ess=diag["ess"], n_chains=diag["n_chains"], rhat_m=diag2["rhat"], trace_all=tr2, rhat_z=diag3["rhat"],
         ess_z=diag3["ess"], ess_one=diag4["ess"])
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_two_ranks_on_one_gpu_equal_the_single_process_run(tmp_path):
    """The multi-GPU path end to end with 2 ranks (gloo) sharing this GPU: sample_distributed's chain blocks,
    concatenated, ARE the single-process run -- draws and statistics bit for bit -- and the all-reduced R-hat / ESS
    equal the single-process diagnostics of all chains."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "worker.py"
    script.write_text(TWO_RANK_WORKER.format(root=root, out=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK="0"),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o)
    parts = [np.load(str(tmp_path / ("rank%d.npz" % r))) for r in range(2)]
    d, chains = 16, 11
    tgt = lmc.targets.AR1(d, 0.9)
    seeds = lmc.distributed.global_seeds(20260928, chains)
    start, step = lmc.init_nuts(tgt, d, random_seed=seeds)
    trace, stats, eng = lmc.sample(tgt, d, draws=60, tune=80, chains=chains, random_seed=seeds, start=start, step=step,
                                   return_engine=True)
    try:
        assert [p["trace"].shape[0] for p in parts] == [6, 5]
        np.testing.assert_array_equal(np.concatenate([p["trace"] for p in parts]), trace)
        np.testing.assert_array_equal(np.concatenate([p["tree_size"] for p in parts]), stats["tree_size"])
        np.testing.assert_array_equal(np.concatenate([p["depth"] for p in parts]), stats["depth"])
        want = dg.summarize(dg.trace_tensor(eng))
        for p in parts:          # every rank holds the diagnostics of ALL chains
            assert float(p["n_chains"]) == 2.0 * chains
            np.testing.assert_allclose(p["rhat"], want["rhat"].cpu().numpy(), rtol=1e-12)
            np.testing.assert_allclose(p["ess"], want["ess"].cpu().numpy(), rtol=1e-10)
        rhat_m, _ = odg.rhat_ess(trace, do_split=False)
        np.testing.assert_allclose(parts[0]["rhat_m"], rhat_m, rtol=1e-9)
        rhat_z, ess_z = odg.rhat_ess(trace, rank_normalized=True)      # GLOBAL ranks over both blocks
        for p in parts:
            np.testing.assert_allclose(p["rhat_z"], rhat_z, rtol=1e-8)
            np.testing.assert_allclose(p["ess_z"], ess_z, rtol=1e-6)
        np.testing.assert_array_equal(parts[0]["ess_one"], parts[1]["ess_one"])
        assert parts[0]["trace_all"].shape == (6, 140, d)          # sample()'s own keywords reach sample()
    finally:
        eng.close()


RCCL_WORKER = """
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
import littlemcmc_amd as lmc
from littlemcmc_amd import diagnostics as dg
from oracle import diagnostics_oracle as odg
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))     # "nccl" IS RCCL on ROCm
ones = torch.ones(4, device="cuda")
dist.all_reduce(ones)                                                                        # RCCL has executed
torch.cuda.synchronize()
assert ones.tolist() == [1.0] * 4
d = 9
tgt = lmc.targets.AR1(d, 0.6)
tr, st, eng = lmc.sample(tgt, d, draws=120, tune=100, chains=10, random_seed=4, return_engine=True)
got = dg.summarize(dg.trace_tensor(eng), reduce_device="cuda")           # every pass: one RCCL all-reduce on the GPU
rhat, ess = odg.rhat_ess(tr)
np.testing.assert_allclose(got["rhat"].cpu().numpy(), rhat, rtol=1e-9)
np.testing.assert_allclose(got["ess"].cpu().numpy(), ess, rtol=1e-7)
z = dg.summarize(dg.trace_tensor(eng), reduce_device="cuda", rank_normalized=True)       # all_gather over RCCL
rz, ez = odg.rhat_ess(tr, rank_normalized=True)
np.testing.assert_allclose(z["rhat"].cpu().numpy(), rz, rtol=1e-8)
eng.close()
for mode in ("moments", True):
    tr2, st2, diag = lmc.distributed.sample_distributed(tgt, d, draws=120, tune=100, chains=10, random_seed=4, diagnostics=mode)
    np.testing.assert_array_equal(tr2, tr)
    want, _ = odg.rhat_ess(tr, do_split=(mode is True))
    np.testing.assert_allclose(diag["rhat"], want, rtol=1e-9)
dist.barrier(); dist.destroy_process_group()
print("rccl ok")
"""


def test_rccl_backend_single_rank(tmp_path):
    """The "nccl" (= RCCL) process group, world_size 1, on this GPU: an all-reduce of ones, diagnostics.summarize with the
    reduction on the device, the rank-normalised variant (all_gather), and sample_distributed with both diagnostics modes
    -- the very collectives the 8-GPU job issues have executed once on this box. Runs in a subprocess so that a RCCL
    failure cannot take the test session with it."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rccl_worker.py"
    script.write_text(RCCL_WORKER.format(root=root))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29563", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "rccl ok" in res.stdout
