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


def test_sample_distributed_single_rank_equals_sample():
    d = 5
    tgt = lmc.targets.StdNormal(d)
    tr, st, diag = lmc.distributed.sample_distributed(tgt, d, draws=40, tune=40, chains=6, random_seed=77)
    tr2, st2 = lmc.sample(tgt, d, draws=40, tune=40, chains=6, random_seed=77)
    np.testing.assert_array_equal(tr, tr2)
    np.testing.assert_array_equal(st["tree_size"], st2["tree_size"])
    assert diag["rhat"].shape == (d,) and diag["n_chains"] == 12.0
