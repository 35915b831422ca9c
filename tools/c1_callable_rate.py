#!/usr/bin/env python3
"""GPU box: BASELINE.json configs[0] ("C1": 4 chains, dim = 10 standard normal, HamiltonianMC path_length = 2.0, tune 1000 +
draws 1000) through the two ways a user can hand the density over -- the reference's own plug-in form, a plain per-point
Python callable (targets.CallableTarget: sampler on the device, the callable evaluated on the host once per chain per tick),
and the built-in device functor (fused kernel) -- next to the numpy oracle on one host core. Round 4's review: "its
throughput is not stated anywhere".   python tools/c1_callable_rate.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import littlemcmc_amd as lmc  # noqa: E402
from oracle import lmc_oracle as orc  # noqa: E402
from oracle import targets as OT  # noqa: E402

d, chains, tune, draws, seed = 10, 4, 1000, 1000, 20260928


def logp_dlogp(q):   # the reference's plug-in signature (integration.py:40): q[d] -> (logp, dlogp[d])
    return -0.5 * np.dot(q, q), -q


def run(name, target):
    step = lmc.HamiltonianMC(target, d, path_length=2.0)
    lmc.sample(target, d, draws=20, tune=20, step=step, chains=chains, random_seed=seed, progressbar=False)   # warm-up (code objects)
    step = lmc.HamiltonianMC(target, d, path_length=2.0)
    t0 = time.perf_counter()
    trace, stats = lmc.sample(target, d, draws=draws, tune=tune, step=step, chains=chains, random_seed=seed,
                              discard_tuned_samples=False, progressbar=False)
    dt = time.perf_counter() - t0
    leaps = float(stats["n_steps"].sum())
    print("%-44s %9.3e leapfrog-steps/s  (%d chains x %d iterations, %.0f leapfrogs, %.2f s of sample() wall time; mean %.3f var %.3f)"
          % (name, leaps / dt, chains, tune + draws, leaps, dt, trace[:, tune:].mean(), trace[:, tune:].var()))
    return trace


a = run("plain Python callable (CallableTarget, ticks)", logp_dlogp)
b = run("device functor StdNormal (fused kernel)", lmc.targets.StdNormal(d))
print("same chains either way (prefix of 30 iterations): %s" % np.allclose(a[:, :30], b[:, :30], rtol=1e-6, atol=1e-9))
f = OT.StdNormal(d)
ostep = orc.Step(f, d, kind="hmc", path_length=2.0)
t0 = time.perf_counter()
_tr, st = orc.sample(f, d, draws=draws, tune=tune, step=ostep, chains=chains, random_seed=seed, discard_tuned_samples=False)
dt = time.perf_counter() - t0
print("%-44s %9.3e leapfrog-steps/s  (one host core, %.2f s)" % ("numpy oracle (port of the reference)", float(st["n_steps"].sum()) / dt, dt))
