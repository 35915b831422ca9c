#!/usr/bin/env python3
"""The two jobs tools/pc_sample.sh samples program counters of (GPU box):
    lone   ONE funnel chain (d = 256, max_treedepth 12, fixed small step: every tree a full 4 095-leapfrog tree) under the
           deep-tree LDS plan -- the straggler whose dependent instruction chain is C5's wall time
    c3     8 192 AR(1) d = 128 chains, 300 iterations past the settling phase, deep-tree plan -- the headline kernel at full
           occupancy"""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import littlemcmc_amd as lmc  # noqa: E402
from littlemcmc_amd import _abi  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "lone"
if which == "lone":
    d = 256
    step = lmc.NUTS(lmc.targets.Funnel(d), d, max_treedepth=12, adapt_step_size=False, lds_plan="deep")
    eng = step._make_engine(1)
    eng.seed([1234])
    q0 = np.zeros(d)
    q0[0] = -2.0
    eng.set_position(q0[None, :])
    eng.reset_tuning()
    eng.set_dual_average(np.log(2e-3), np.log(2e-3))
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    eng.reserve(n, keep_trace=False)
    t0 = time.perf_counter()
    eng.run(0, 0, n)
    eng.synchronize()
    t = time.perf_counter() - t0
    leaps = int(eng.counters()[0, _abi.CT_LEAPFROGS])
    print("lone funnel chain: %d leapfrogs in %.3f s -> %.3f us per leapfrog, plan %s" % (leaps, t, 1e6 * t / leaps, eng.last_run_plan()))
else:
    d, chains = 128, 8192
    tgt = lmc.targets.AR1(d, 0.9)
    seeds = lmc.distributed.global_seeds(7, chains)
    start, step = lmc.init_nuts(tgt, d, random_seed=seeds, lds_plan="deep")
    eng = step._make_engine(chains)
    eng.seed(seeds)
    eng.set_position(start)
    eng.reset_tuning()
    n = 500
    eng.reserve(n, keep_trace=False)
    t0 = time.perf_counter()
    for first in range(0, n, 100):
        eng.run(250, first, 100)
    eng.synchronize()
    t = time.perf_counter() - t0
    leaps = int(eng.counters()[:, _abi.CT_LEAPFROGS].sum())
    print("C3 slice: %d chains, %d leapfrogs in %.3f s -> %.3e leapfrog-steps/s, plan %s" % (chains, leaps, t, leaps / t, eng.last_run_plan()))
eng.close()
