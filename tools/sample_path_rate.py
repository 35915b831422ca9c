#!/usr/bin/env python3
"""GPU box: leapfrog-steps/s of the job loop lmc.sample() itself runs (sampling._run_job: its launch schedule, two launches in
flight, the engine choosing the LDS plan per launch) for C3 and the north_star shape, with the plan pinned and chosen.
    python tools/sample_path_rate.py"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


SCHEDULES = {"s4x100": [100, 100, 100, 100, 500], "s200_300": [200, 300, 500], "s500": [500]}


def one(target_name, sched="s4x100"):
    import numpy as np

    import littlemcmc_amd as lmc
    from littlemcmc_amd import _abi, sampling

    d, chains, tune, draws = 128, 65536, 1000, 1000
    kw = {}
    if target_name == "funnel":
        d, chains, kw = 256, 16384, {"max_treedepth": 12}
    tgt = {"ar1": lambda: lmc.targets.AR1(d, 0.9), "std_normal": lambda: lmc.targets.StdNormal(d), "funnel": lambda: lmc.targets.Funnel(d)}[target_name]()
    seeds = lmc.distributed.global_seeds(20260928, chains)
    start, step = lmc.init_nuts(tgt, d, random_seed=seeds, **kw)
    for rep in range(2):
        eng = step._make_engine(chains)
        try:
            eng.seed(seeds)
            eng.set_position(start)
            eng.reset_tuning()
            eng.reserve(tune + draws, keep_trace=False)
            eng.synchronize()
            t0 = time.perf_counter()
            sampling._run_job(eng, tune, tune + draws, SCHEDULES[sched], False)
            dt = time.perf_counter() - t0
            leaps = float(eng.counters()[:, _abi.CT_LEAPFROGS].sum())
        finally:
            eng.close()
    print("%-10s launches %-9s LMC_LDS_PLAN=%-4s %.4e leapfrog-steps/s (%.2f s)" % (target_name, sched, os.environ.get("LMC_LDS_PLAN", "auto"), leaps / dt, dt))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        one(*sys.argv[1:3])
    else:
        for name in ("ar1", "std_normal", "funnel"):
            for sched in SCHEDULES:
                for plan in ("0", None):
                    env = dict(os.environ)
                    env.pop("LMC_LDS_PLAN", None)
                    if plan is not None:
                        env["LMC_LDS_PLAN"] = plan
                    subprocess.call([sys.executable, os.path.abspath(__file__), name, sched], env=env)
