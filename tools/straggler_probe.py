#!/usr/bin/env python3
"""GPU box: how much of a many-chain launch is tail? Runs one config (default C5: 16 384 x d=256 funnel, treedepth 12)
launch by launch and prints, per launch, the leapfrogs of the mean and of the busiest chain, the launch time, and the
time the same leapfrogs would take if they were spread evenly over the resident wave slots.

    PYTHONPATH=. python tools/straggler_probe.py [target dim chains max_treedepth tune draws launches]
"""
import sys
import time

import numpy as np
import torch

import littlemcmc_amd as lmc
from littlemcmc_amd import _abi
from littlemcmc_amd import targets as T

a = sys.argv[1:]
name = a[0] if a else "funnel"
d = int(a[1]) if len(a) > 1 else 256
chains = int(a[2]) if len(a) > 2 else 16384
depth = int(a[3]) if len(a) > 3 else 12
tune = int(a[4]) if len(a) > 4 else 1000
draws = int(a[5]) if len(a) > 5 else 1000
launches = int(a[6]) if len(a) > 6 else 20
tgt = {"funnel": lambda: T.Funnel(d), "ar1": lambda: T.AR1(d, 0.9), "std_normal": lambda: T.StdNormal(d),
       "diag": lambda: T.DiagGaussian.ill_conditioned(d, 1e4)}[name]()
seeds = lmc.distributed.global_seeds(20260928, chains)
start, step = lmc.init_nuts(tgt, d, random_seed=seeds, max_treedepth=depth)
eng = step._make_engine(chains)
eng.seed(seeds); eng.set_position(start); eng.reset_tuning()
n_total = tune + draws
eng.reserve(n_total, keep_trace=False)
ips = n_total // launches
prev = np.zeros(chains)
tot_t = 0.0
for k in range(launches):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.run(tune, k * ips, ips)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    tot_t += dt
    cur = eng.counters()[:, _abi.CT_LEAPFROGS].astype("d")
    lf = cur - prev; prev = cur
    q = np.quantile(lf, [0.5, 0.9, 0.99, 0.999])
    print("launch %2d  %.1f ms  leapfrogs/chain: mean %.0f  p50 %.0f p90 %.0f p99 %.0f p99.9 %.0f max %.0f   max/mean %.1f  "
          "rate %.3g /s   busiest chain alone: %.2f us/leapfrog" % (k, dt * 1e3, lf.mean(), q[0], q[1], q[2], q[3], lf.max(),
                                                                    lf.max() / lf.mean(), lf.sum() / dt, dt / lf.max() * 1e6), flush=True)
print("total %.3f s, %.3g leapfrogs/s; per-chain totals: mean %.0f max %.0f (x%.1f)" % (
    tot_t, prev.sum() / tot_t, prev.mean(), prev.max(), prev.max() / prev.mean()))
eng.close()
