#!/usr/bin/env python3
"""GPU box: what does ONE chain's leapfrog cost when nothing but the integration runs? The unit entry point
lmc_engine_trajectory integrates n steps of one chain (one wavefront, the sampler's own leapfrog<>, every state stored) --
no tree, no weights, no U-turn dots. Differencing two lengths cancels launch and copy overheads. Next to the sampling
kernel's lone-wave leapfrog (tools/c5_team_latency.py: 0.875 us at d = 256) this is the share of a deep tree's critical path
that is integration, i.e. the ceiling of any scheme that moves the tree's bookkeeping to other wavefronts.
    python tools/lone_leapfrog_probe.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import littlemcmc_amd as lmc  # noqa: E402

for name, tgt, d in (("funnel d=256 (C5, NS=4)", lmc.targets.Funnel(256), 256), ("AR(1) d=128 (C3, NS=2)", lmc.targets.AR1(128, 0.9), 128),
                     ("std normal d=64 (C2, NS=1)", lmc.targets.StdNormal(64), 64)):
    step = lmc.NUTS(tgt, d)
    eng = step._make_engine(1)
    try:
        q0 = np.zeros(d)
        q0[0] = -1.0
        p0 = np.random.RandomState(1).normal(size=d)
        ts = {}
        for n in (500, 2000, 8000, 2000, 8000):
            t0 = time.perf_counter()
            eng.trajectory(q0, p0, 1e-3, n, 0, p0_is_f32=False)
            ts[n] = time.perf_counter() - t0
        us = 1e6 * (ts[8000] - ts[2000]) / 6000.0
        print("%-28s %.3f us per leapfrog integrated alone (%.1f ms for 8000 steps, %.1f ms for 2000)" % (name, us, 1e3 * ts[8000], 1e3 * ts[2000]))
    finally:
        eng.close()
