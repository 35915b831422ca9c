"""Diagnostics (split R-hat / ESS) vs the numpy restatement, and the multi-rank reduction over gloo
(world_size 2, CPU) -- the only collective of the multi-GPU path."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from littlemcmc_amd import diagnostics as dg
from littlemcmc_amd.distributed import chain_block
from oracle import diagnostics_oracle as odg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def ar1_chains(chains, n, d, rho, seed):
    rs = np.random.RandomState(seed)
    x = np.zeros((chains, n, d))
    x[:, 0] = rs.randn(chains, d)
    for t in range(1, n):
        x[:, t] = rho * x[:, t - 1] + np.sqrt(1 - rho ** 2) * rs.randn(chains, d)
    return x + np.linspace(-1, 1, d)


@pytest.mark.parametrize("rho", [0.0, 0.6, -0.4])
def test_rhat_ess_match_numpy_restatement(rho):
    x = ar1_chains(6, 80, 3, rho, 1)
    got = dg.summarize(torch.from_numpy(x), stats_fn=odg.torch_chain_stats)
    rhat, ess = odg.rhat_ess(x)
    np.testing.assert_allclose(got["rhat"].numpy(), rhat, rtol=1e-10)
    np.testing.assert_allclose(got["ess"].numpy(), ess, rtol=1e-8)


def test_ess_recovers_known_autocorrelation_time():
    rho = 0.5
    x = ar1_chains(256, 400, 2, rho, 3)
    got = dg.summarize(torch.from_numpy(x), stats_fn=odg.torch_chain_stats)
    want = 256 * 400 * (1 - rho) / (1 + rho)
    assert np.all(np.abs(got["ess"].numpy() / want - 1) < 0.1)
    assert np.all(np.abs(got["rhat"].numpy() - 1) < 0.01)


def test_lag_passes_follow_the_autocorrelation_length():
    """16 lags per pass; passes stop once Geyer's initial positive sequence has ended in every dimension."""
    fast = dg.summarize(torch.from_numpy(ar1_chains(10, 400, 4, 0.3, 5)), stats_fn=odg.torch_chain_stats)
    slow = dg.summarize(torch.from_numpy(ar1_chains(10, 400, 4, 0.95, 5)), stats_fn=odg.torch_chain_stats)
    assert fast["lag_passes"] == 1 and slow["lag_passes"] >= 3
    x = ar1_chains(10, 400, 4, 0.95, 5)
    rhat, ess = odg.rhat_ess(x)
    np.testing.assert_allclose(slow["ess"].numpy(), ess, rtol=1e-8)
    capped = dg.summarize(torch.from_numpy(x), max_lag=16, stats_fn=odg.torch_chain_stats)
    assert capped["lag_passes"] == 1 and np.all(capped["ess"].numpy() >= slow["ess"].numpy())


@pytest.mark.parametrize("rho", [0.0, 0.7])
def test_rank_normalised_variant_matches_numpy_restatement(rho):
    x = np.exp(ar1_chains(5, 90, 3, rho, 11))          # heavy right tail: rank normalisation matters
    got = dg.summarize(torch.from_numpy(x), rank_normalized=True, stats_fn=odg.torch_chain_stats)
    rhat, ess = odg.rhat_ess(x, rank_normalized=True)
    np.testing.assert_allclose(got["rhat"].numpy(), rhat, rtol=1e-9)
    np.testing.assert_allclose(got["ess"].numpy(), ess, rtol=1e-7)
    plain = dg.summarize(torch.from_numpy(x), stats_fn=odg.torch_chain_stats)
    assert not np.allclose(plain["ess"].numpy(), got["ess"].numpy(), rtol=1e-3)


def test_chain_blocks_partition_the_chains():
    for total, world in [(65536, 8), (10, 3), (7, 8), (4096, 1)]:
        blocks = [chain_block(total, r, world) for r in range(world)]
        assert blocks[0][0] == 0 and blocks[-1][1] == total
        assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
        sizes = [b - a for a, b in blocks]
        assert max(sizes) - min(sizes) <= 1


WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from littlemcmc_amd import diagnostics as dg
from littlemcmc_amd.distributed import chain_block, global_seeds
from oracle import diagnostics_oracle as odg
from tests.test_diagnostics_cpu import ar1_chains
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
solo = [dist.new_group([r]) for r in range(world)][rank]       # a one-rank group per rank (created collectively)
fft = odg.torch_chain_stats                                    # the oracle's restatement of the kernel's block
x = ar1_chains(12, 90, 3, 0.5, 7)                    # the WHOLE job, identical on both ranks
lo, hi = chain_block(12, rank, world)
got = dg.summarize(torch.from_numpy(x[lo:hi]), stats_fn=fft)       # each rank only sees its block
ref = dg.summarize(torch.from_numpy(x), group=solo, stats_fn=fft)  # single-rank reduction of everything
np.testing.assert_allclose(got["ess"].numpy(), ref["ess"].numpy(), rtol=1e-10)
np.testing.assert_allclose(got["rhat"].numpy(), ref["rhat"].numpy(), rtol=1e-12)
assert got["n_chains"] == 24.0
# several lag passes: every rank must issue the same number of collectives
xs = ar1_chains(12, 400, 3, 0.95, 9)
gs = dg.summarize(torch.from_numpy(xs[lo:hi]), stats_fn=fft)
rs = dg.summarize(torch.from_numpy(xs), group=solo, stats_fn=fft)
assert gs["lag_passes"] == rs["lag_passes"] >= 3
np.testing.assert_allclose(gs["ess"].numpy(), rs["ess"].numpy(), rtol=1e-10)
# rank-normalised: GLOBAL ranks (the blocks differ in location: per-rank normalisation would erase it), ties averaged
xt = np.round(np.exp(ar1_chains(12, 90, 3, 0.5, 11)), 1)           # rounding makes exact ties
xt[6:] += 2.0                                                     # the second rank's chains sit somewhere else
gr = dg.summarize(torch.from_numpy(xt[lo:hi]), rank_normalized=True, stats_fn=fft)
want_rhat, want_ess = odg.rhat_ess(xt, rank_normalized=True)       # scipy rankdata(method="average") over everything
np.testing.assert_allclose(gr["rhat"].numpy(), want_rhat, rtol=1e-9)
np.testing.assert_allclose(gr["ess"].numpy(), want_ess, rtol=1e-7)
assert want_rhat.max() > 1.05                                      # the shift is visible, i.e. was not normalised away
# a rank that owns NO chain (more ranks than chains) joins every collective with an empty block of any draw count
one = ar1_chains(1, 400, 3, 0.95, 13)
l1, h1 = chain_block(1, rank, world)
blk = torch.from_numpy(one[l1:h1]) if h1 > l1 else torch.zeros((0, 7, 3), dtype=torch.float64)
g1 = dg.summarize(blk, stats_fn=fft)
r1 = dg.summarize(torch.from_numpy(one), group=solo, stats_fn=fft)
np.testing.assert_allclose(g1["ess"].numpy(), r1["ess"].numpy(), rtol=1e-10)
assert g1["lag_passes"] == r1["lag_passes"] and g1["n_chains"] == 2.0
# ranks that disagree on the draw count fail TOGETHER (no rank is left waiting in a collective)
try:
    dg.summarize(torch.from_numpy(x[lo:hi, : 90 - 2 * rank]), stats_fn=fft)
    raise SystemExit("draw-count mismatch went unnoticed")
except ValueError as err:
    assert "different numbers of draws" in str(err)
seeds = global_seeds(20260928, 12)
assert seeds[:4] == global_seeds(20260928, 4)          # prefix stable: blocks do not depend on the job size
out = [None] * world
dist.all_gather_object(out, seeds[lo:hi])
assert sum(out, []) == seeds
# trace-free R-hat from per-chain running moments (what lmc_engine_get_moments hands out), block per rank
blk = x[lo:hi]
mean = blk.mean(axis=1)
m2 = ((blk - mean[:, None, :]) ** 2).sum(axis=1)
rh = dg.rhat_from_moments(mean, m2, np.full(hi - lo, blk.shape[1])).numpy()
want, _ = odg.rhat_ess(x, do_split=False)
np.testing.assert_allclose(rh, want, rtol=1e-10)
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_the_package_has_no_host_implementation_of_the_statistics():
    """CPU tensors are refused: the HIP kernel is the only backend (the FFT restatement lives in oracle/)."""
    from littlemcmc_amd._abi import HipLibraryError

    with pytest.raises(HipLibraryError):
        dg.summarize(torch.zeros((4, 40, 2), dtype=torch.float64))
    assert not hasattr(dg, "_torch_chain_stats")


def test_rank_normalisation_averages_ties():
    rs = np.random.RandomState(3)
    x = np.round(rs.randn(4, 50, 2), 1)                  # many exact ties
    z = dg.rank_normalize(torch.from_numpy(x)).numpy()
    np.testing.assert_allclose(z, odg.rank_normalize(x), rtol=1e-12, atol=1e-14)


def test_two_rank_gloo_reduction_equals_single_process(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o)
        assert "ok" in o


WORKER8 = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
torch.set_num_threads(1)
from littlemcmc_amd import diagnostics as dg
from littlemcmc_amd.distributed import chain_block
from oracle import diagnostics_oracle as odg
from tests.test_diagnostics_cpu import ar1_chains
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
fft = odg.torch_chain_stats
for total in (13, 5):          # 13 chains over 8 ranks: blocks of 2 and 1; 5 chains: three ranks own nothing
    x = ar1_chains(total, 200, 2, 0.9, 17 + total)
    lo, hi = chain_block(total, rank, world)
    blk = torch.from_numpy(x[lo:hi]) if hi > lo else torch.zeros((0, 3, 2), dtype=torch.float64)
    got = dg.summarize(blk, stats_fn=fft)
    rhat, ess = odg.rhat_ess(x)
    np.testing.assert_allclose(got["rhat"].numpy(), rhat, rtol=1e-9)
    np.testing.assert_allclose(got["ess"].numpy(), ess, rtol=1e-7)
    assert got["n_chains"] == 2.0 * total and got["lag_passes"] >= 2
    z = dg.summarize(blk, stats_fn=fft, rank_normalized=True)
    rz, ez = odg.rhat_ess(x, rank_normalized=True)
    np.testing.assert_allclose(z["rhat"].numpy(), rz, rtol=1e-8)
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_eight_rank_gloo_reduction_with_uneven_and_empty_blocks(tmp_path):
    """The diagnostics reduction as the 8-GPU job issues it (world_size 8, gloo, CPU): uneven chain blocks, ranks that own
    no chain, several lag passes, the globally rank-normalised variant -- every rank issues the same collectives and gets
    the all-chain result."""
    script = tmp_path / "worker8.py"
    script.write_text(WORKER8.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29537", WORLD_SIZE="8", OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(8)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o)
        assert "ok" in o


def test_per_gpu_blocks_in_one_process_reduce_like_one_block():
    """sample(devices=[...]) hands the diagnostics a LIST of per-GPU trace views (diagnostics.trace_tensor(group)): the
    (3 + 16) x d blocks of the parts are added, global ranks are counted over every part's sorted pool (ties averaged) --
    the result is the one-block result whatever the split, uneven and empty parts included."""
    x = ar1_chains(11, 160, 5, 0.8, 3)
    x[3, 40:60] = x[3, 40:41]                       # a stuck stretch: tied values for the rank normalisation
    whole = torch.from_numpy(x)
    for cuts in ([0, 4, 11], [0, 1, 1, 7, 11], [0, 11, 11]):
        parts = [whole[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
        for rn in (False, True):
            ref = dg.summarize(whole, rank_normalized=rn, stats_fn=odg.torch_chain_stats)
            got = dg.summarize(parts, rank_normalized=rn, stats_fn=odg.torch_chain_stats)
            for k in ("rhat", "ess", "mean", "var"):
                np.testing.assert_allclose(got[k].numpy(), ref[k].numpy(), rtol=1e-10, atol=1e-12, err_msg="%s %s %s" % (cuts, rn, k))
            assert got["n_chains"] == ref["n_chains"] and got["n_draws"] == ref["n_draws"]
    with pytest.raises(ValueError, match="different numbers of draws"):
        dg.summarize([whole[:4], whole[4:, :100]], stats_fn=odg.torch_chain_stats)
