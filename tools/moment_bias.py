#!/usr/bin/env python3
"""GPU box: pooled marginal variance of many-chain NUTS runs, float32-momentum potential (QuadPotentialDiagAdapt, the
reference default) vs float64-momentum potential (QuadPotentialDiag / scaling=): is the ~+0.25 % variance seen at
65 536 x 128 a property of the algorithm's float32 quirks (nuts.py:329) or of the device code?"""
import sys

import numpy as np

import littlemcmc_amd as lmc
from littlemcmc_amd import targets as T


def pooled(mean, m2, n):
    n = np.asarray(n, dtype="d")[:, None]
    tot = n.sum()
    grand = (mean * n).sum(axis=0) / tot
    return grand, (m2.sum(axis=0) + (n * (mean - grand) ** 2).sum(axis=0)) / (tot - 1.0)


def run(tgt, d, chains, tune, draws, mode):
    seeds = lmc.distributed.global_seeds(20260928, chains)
    if mode == "adapt_f32":
        start, step = lmc.init_nuts(tgt, d, random_seed=seeds)
    else:
        np.random.seed(int(seeds[0]))
        start = 2 * np.random.rand(d) - 1
        step = lmc.NUTS(tgt, d, scaling=np.ones(d), is_cov=True)
    eng = step._make_engine(chains)
    eng.seed(seeds); eng.set_position(start); eng.reset_tuning(); eng.keep_moments(True)
    eng.reserve(tune + draws, keep_trace=False)
    eng.run(tune, 0, tune + draws)
    mean, m2, n = eng.moments()
    from littlemcmc_amd import _abi
    depth = eng.stat_i32(_abi.STAT_DEPTH, tune, draws).mean()
    eng.close()
    gm, gv = pooled(np.asarray(mean), np.asarray(m2), n)
    print("%-10s d=%d chains=%d draws=%d: mean|max %.2e  var-1: mean %+.2e min %+.2e max %+.2e  depth %.2f" % (
        mode, d, chains, draws, np.abs(gm).max(), (gv - 1).mean(), (gv - 1).min(), (gv - 1).max(), depth), flush=True)


if __name__ == "__main__":
    chains = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    for name, tgt, d in [("ar1", T.AR1(128, 0.9), 128), ("std", T.StdNormal(128), 128), ("ar1-32", T.AR1(32, 0.9), 32)]:
        for mode in ("adapt_f32", "fixed_f64"):
            print(name, end=" ")
            run(tgt, d, chains, 400, 600, mode)
