#!/usr/bin/env python3
"""GPU box: leapfrog latency of ONE chain alone on the GPU -- the straggler that bounds C5 -- through the kernel shapes a
d = 256 chain can take: one wavefront of four elements per lane (<4,1>, what ships), two waves of two (<2,2>), four waves
of one (<1,4>). Needs a -DLMC_EXPERIMENTAL_SHAPES build (LMC_HIP_LIB). Funnel d = 256, max_treedepth 12, fixed small step
size so that every tree is a full depth-12 tree (4 095 leapfrogs), 20 iterations timed.

    LMC_HIP_LIB=build_variants/liblmc_shapes.so python tools/c5_team_latency.py
"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(shape):
    import numpy as np

    import littlemcmc_amd as lmc
    from littlemcmc_amd import _abi

    d = 256
    tgt = lmc.targets.Funnel(d)
    step = lmc.NUTS(tgt, d, max_treedepth=12, adapt_step_size=False)
    eng = step._make_engine(1)
    try:
        eng.seed([1234])
        q0 = np.zeros(d)
        q0[0] = -2.0                      # in the neck: small steps, no U-turn before depth 12
        eng.set_position(q0[None, :])
        eng.reset_tuning()
        eng.set_dual_average(np.log(2e-3), np.log(2e-3))
        n = 24
        eng.reserve(n, keep_trace=False)
        eng.run(0, 0, 4)
        eng.synchronize()
        l0 = int(eng.counters()[0, _abi.CT_LEAPFROGS])
        t0 = time.perf_counter()
        eng.run(0, 4, n - 4)
        eng.synchronize()
        t = time.perf_counter() - t0
        l1 = int(eng.counters()[0, _abi.CT_LEAPFROGS]) - l0
        depth = eng.stat_i32(_abi.STAT_DEPTH, 4, n - 4).mean()
        print("shape %s: %d leapfrogs in %.1f ms -> %.3f us per leapfrog (mean depth %.2f, kernel shape %s)" % (
            shape, l1, 1e3 * t, 1e6 * t / max(l1, 1), depth, eng.kernel_shape()))
    finally:
        eng.close()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        one(sys.argv[1])
    else:
        for shape in ("4,1", "2,2", "1,4", "4,1", "2,2", "1,4"):
            env = dict(os.environ, LMC_RUN_SHAPE=shape)
            subprocess.call([sys.executable, os.path.abspath(__file__), shape], env=env)
