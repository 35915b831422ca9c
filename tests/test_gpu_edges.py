"""-m gpu: edge cases of the driver / engine boundary: empty outputs, single chain, tune-only and draw-only runs,
the largest supported dimension, invalid shapes, capacity errors."""
import numpy as np
import pytest

import littlemcmc_amd as lmc
from littlemcmc_amd import _abi
from littlemcmc_amd import targets as T

pytestmark = pytest.mark.gpu


def test_empty_and_degenerate_runs():
    d = 3
    tgt = T.StdNormal(d)
    # draws = 0 with discard_tuned_samples: empty trace, stats keep their dtypes (sampling.py:141-143 only warns)
    tr, st = lmc.sample(tgt, d, draws=0, tune=7, chains=2, random_seed=1)
    assert tr.shape == (2, 0, d) and st["depth"].shape == (2, 0, 1) and st["depth"].dtype == np.int64
    # tune = 0: no adaptation at all, step size stays at step_scale / d**0.25
    tr, st = lmc.sample(tgt, d, draws=6, tune=0, chains=1, random_seed=1)
    assert tr.shape == (1, 6, d) and not st["tune"].any()
    np.testing.assert_allclose(st["step_size_bar"], 0.25 / d ** 0.25, rtol=1e-12)
    # one chain, one draw
    tr, st = lmc.sample(tgt, d, draws=1, tune=1, chains=1, random_seed=1)
    assert tr.shape == (1, 1, d)


def test_largest_dimension_and_beyond():
    d = 1024
    tr, st = lmc.sample(T.StdNormal(d), d, draws=3, tune=5, chains=2, random_seed=2)
    assert tr.shape == (2, 3, d) and np.isfinite(tr).all()
    eng = lmc.Engine(T.StdNormal(1025), chains=1)        # beyond the fused kernels: the general kernels (tests/test_gpu_wide.py)
    try:
        assert eng.wide and eng.kernel_shape()[2] == 16
    finally:
        eng.close()
    with pytest.raises(_abi.HipLibraryError, match="beyond the general kernels"):
        lmc.Engine(T.StdNormal(16385), chains=1)
    with pytest.raises(_abi.HipLibraryError, match="requires dim == 1"):
        t = T.Normal1D()
        t.d = 2
        lmc.Engine(t, chains=1)


def test_engine_call_sequence_errors():
    eng = lmc.Engine(T.StdNormal(4), chains=2)
    try:
        with pytest.raises(_abi.HipLibraryError, match="reserve"):
            eng.run(0, 0, 1)
        eng.reserve(5, keep_trace=False)
        with pytest.raises(_abi.HipLibraryError, match="exceed reserved capacity"):
            eng.run(0, 3, 4)
        eng.run(2, 0, 5)
        with pytest.raises(_abi.HipLibraryError, match="no trace"):
            eng.trace(0, 5)
        assert eng.stat_i32(_abi.STAT_TREE_SIZE).shape == (2, 5)
        with pytest.raises(_abi.HipLibraryError, match="positive definite"):
            eng.set_potential(np.zeros(4), np.array([1.0, 0.0, 1.0, 1.0]), 10.0)
        dg = lmc.Engine(T.DiagGaussian(np.ones(3)), chains=1)
        try:
            with pytest.raises(_abi.HipLibraryError, match="precisions"):
                dg._check(dg._lib.lmc_engine_set_target_params(dg._h, _abi.ptr(np.ones(2)), 2))
        finally:
            dg.close()
    finally:
        eng.close()


def test_trace_window_and_partial_reads():
    d, chains = 5, 3
    eng = lmc.NUTS(T.StdNormal(d), d)._make_engine(chains)
    try:
        eng.seed([1, 2, 3])
        eng.set_position(np.zeros(d))
        eng.reset_tuning()
        eng.reserve(20, keep_trace=True, trace_begin=8)
        eng.run(8, 0, 20)
        full = eng.trace()
        assert full.shape == (chains, 12, d)
        np.testing.assert_array_equal(eng.trace(10, 4), full[:, 2:6])
        with pytest.raises(_abi.HipLibraryError, match="not stored"):
            eng.trace(3, 2)
        np.testing.assert_array_equal(eng.get_position(), full[:, -1])      # the chain's position is its last draw
    finally:
        eng.close()


def test_results_do_not_depend_on_where_the_tree_stack_lives():
    """Subtree-stack levels in LDS or spilled to the HBM scratch row: same bits (AR(1) d=48, depth up to 8)."""
    d, chains, n = 48, 24, 40
    tgt = T.AR1(d, 0.95)
    ref = None
    for lds_levels in (1, 2, 4, 0):
        eng = lmc.Engine(tgt, chains=chains, lds_levels=lds_levels)
        try:
            eng.set_potential(np.zeros(d), np.ones(d), 10.0)
            eng.seed(list(range(chains)))
            eng.set_position(np.full(d, 0.1))
            eng.reset_tuning()
            eng.reserve(n)
            eng.run(25, 0, n)
            out = (eng.trace(), eng.stat_i32(_abi.STAT_TREE_SIZE), eng.stat_f64(_abi.STAT_ENERGY))
            assert out[1].max() > 31        # deep enough to reach spilled levels when lds_levels is small
        finally:
            eng.close()
        if ref is None:
            ref = out
        else:
            for a, b in zip(ref, out):
                np.testing.assert_array_equal(a, b)


def test_degenerate_shapes_on_the_dense_and_tick_paths():
    """d = 1 and 2, one chain, tune = 0 / draws = 0, max_treedepth = 1: the widened paths (dense mass, torch-callable
    density, both together) at the smallest sizes the reference accepts."""
    import torch

    from littlemcmc_amd.targets import TorchTarget

    def torch_normal(d):
        return TorchTarget(d, lambda q: (-0.5 * (q * q).sum(dim=1), -q))

    for d in (1, 2):
        for tgt in (lmc.targets.StdNormal(d), torch_normal(d)):
            for init in ("adapt_diag", "adapt_full"):
                tr, st = lmc.sample(tgt, d, draws=6, tune=9, chains=1, init=init, random_seed=3)
                assert tr.shape == (1, 6, d) and np.isfinite(tr).all() and st["tree_size"].min() >= 1
                tr, st = lmc.sample(tgt, d, draws=0, tune=5, chains=2, init=init, random_seed=3)
                assert tr.shape == (2, 0, d) and st["depth"].shape == (2, 0, 1)
                tr, st = lmc.sample(tgt, d, draws=5, tune=0, chains=2, init=init, random_seed=3, max_treedepth=1)
                assert tr.shape == (2, 5, d) and (st["depth"] == 1).all() and (st["tree_size"] == 1).all()
    # a 1 x 1 "dense" matrix is a diagonal one: same chain as QuadPotentialDiag... up to the float32 / float64 momentum
    # dtype, so compare the two dense classes with each other instead (cov = 1 / A)
    a = lmc.sample(lmc.targets.StdNormal(1), 1, draws=20, tune=0, chains=3, random_seed=8,
                   step=lmc.HamiltonianMC(lmc.targets.StdNormal(1), 1, potential=lmc.QuadPotentialFullInv(np.array([[4.0]]))))[0]
    assert np.isfinite(a).all() and a.std() > 0
    assert torch.cuda.is_available()
