#!/usr/bin/env python3
"""Leapfrogs/s of the general kernels (csrc/lmc_wide.hpp) at the shapes with dim <= 1024, as one wavefront per chain
(the default there) and as the 16-wavefront team (LMC_WIDE_TEAM=16, the shape dim > 1024 takes). One line per case:
engines are created in fresh subprocesses because the knob is read at engine creation."""
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CASES = [  # (label, family, dim, potential, dtype, chains)
    ("diag f64 d=64", "ar1", 64, "diag_adapt", "float64", 16384),
    ("diag f64 d=256", "ar1", 256, "diag_adapt", "float64", 8192),
    ("diag f64 d=1024", "ar1", 1024, "diag_adapt", "float64", 4096),
    ("diag f64 d=512", "ar1", 512, "diag_adapt", "float64", 8192),
    ("dense f32 d=300", "ar1", 300, "full", "float32", 4096),
    ("dense f32 d=384", "ar1", 384, "full", "float32", 4096),
    ("dense f32 d=512", "ar1", 512, "full", "float32", 4096),
    ("dense f32 d=768", "ar1", 768, "full", "float32", 2048),
    ("dense f32 d=1024", "ar1", 1024, "full", "float32", 2048),
]


def one(label, fam, d, pot, dtype, chains):
    import numpy as np
    import littlemcmc_amd as lmc
    from littlemcmc_amd import targets as T
    from littlemcmc_amd import quadpotential as Q

    tgt = T.AR1(d, 0.5)
    if pot == "diag_adapt":
        potential = Q.QuadPotentialDiagAdapt(d, np.zeros(d), np.ones(d, dtype=dtype), 10, dtype=dtype)
    else:
        potential = Q.QuadPotentialFull(np.eye(d, dtype=np.float32))
    step = lmc.NUTS(tgt, d, potential=potential, step_scale=0.25)
    tune, draws = 50, 50
    best = None
    for rep in range(2):
        t0 = time.perf_counter()
        trace, stats = lmc.sample(tgt, d, draws=draws, tune=tune, step=step, chains=chains, random_seed=list(range(chains)),
                                  start=np.zeros(d), progressbar=False, discard_tuned_samples=False)
        dt = time.perf_counter() - t0
        leap = float(np.asarray(stats["tree_size"]).sum())
        if best is None or dt < best[0]:
            best = (dt, leap)
    print(json.dumps({"case": label, "team": os.environ.get("LMC_WIDE_TEAM", "1"), "chains": chains, "seconds": round(best[0], 3),
                      "leapfrogs_per_s": best[1] / best[0], "leapfrog_elements_per_s": best[1] * d / best[0]}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        one(*CASES[int(sys.argv[1])])
    else:
        for i in range(len(CASES)):
            for team in ("1", "16"):
                env = dict(os.environ, LMC_WIDE_TEAM=team)
                subprocess.run([sys.executable, os.path.abspath(__file__), str(i)], env=env, timeout=600)
