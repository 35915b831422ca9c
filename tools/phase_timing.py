#!/usr/bin/env python3
"""GPU box: where does an iteration go? Needs a library built with -DLMC_PHASE_TIMING (run_kernel then accumulates
s_memtime ticks per phase in three of the per-chain counters):

    python -c "from littlemcmc_amd import _build; _build.build(out='littlemcmc_amd/liblmc_hip_phase.so', extra_flags=['-DLMC_PHASE_TIMING'])"
    LMC_HIP_LIB=littlemcmc_amd/liblmc_hip_phase.so PYTHONPATH=. python tools/phase_timing.py [target dim chains tune draws]
"""
import sys

import numpy as np

import littlemcmc_amd as lmc
from littlemcmc_amd import _abi
from littlemcmc_amd import targets as T

a = sys.argv[1:]
name = a[0] if a else "std_normal"
d = int(a[1]) if len(a) > 1 else 128
chains = int(a[2]) if len(a) > 2 else 65536
tune = int(a[3]) if len(a) > 3 else 300
draws = int(a[4]) if len(a) > 4 else 300
tgt = {"funnel": lambda: T.Funnel(d), "ar1": lambda: T.AR1(d, 0.9), "std_normal": lambda: T.StdNormal(d),
       "diag": lambda: T.DiagGaussian.ill_conditioned(d, 1e4)}[name]()
seeds = lmc.distributed.global_seeds(20260928, chains)
start, step = lmc.init_nuts(tgt, d, random_seed=seeds)
eng = step._make_engine(chains)
eng.seed(seeds); eng.set_position(start); eng.reset_tuning()
eng.reserve(tune + draws, keep_trace=True, trace_begin=tune)
names = ["momentum draw", "start state (logp, float32 energy)", "NUTS transition", "dual averaging", "mass adaptation",
         "bookkeeping + outputs + loop"]
prev = np.zeros((chains, _abi.NUM_COUNTERS), dtype=np.int64)
for label, lo, n in (("tuning", 0, tune), ("draws", tune, draws)):
    eng.run(tune, lo, n)
    eng.synchronize()
    ct = eng.counters().astype(np.int64)
    dlt = ct - prev
    prev = ct
    ph = np.zeros((chains, 6))
    for k in range(3):
        v = dlt[:, k].astype(np.uint64)
        ph[:, 2 * k] = (v >> np.uint64(32)).astype("d")
        ph[:, 2 * k + 1] = (v & np.uint64(0xffffffff)).astype("d")
    tot = ph.sum()
    leap = dlt[:, _abi.CT_LEAPFROGS].sum()
    print("%s: %d iterations, %.1f leapfrogs per iteration, %.0f ticks per iteration per chain" % (label, n, leap / chains / n, tot / chains / n))
    for i in range(6):
        print("   %-38s %5.1f %%   %8.0f ticks per iteration" % (names[i], 100 * ph[:, i].sum() / tot, ph[:, i].sum() / chains / n))
eng.close()
