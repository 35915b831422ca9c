#!/usr/bin/env python3
"""GPU box: where does a leapfrog of the shared-matrix dense kernel (run_dense_coop_kernel) go? Needs a library built with
-DLMC_COOP_TIMING (coop_product then accumulates clock ticks per chain: waiting for the other seven chains, multiplying,
waiting for the product, the chain's own work between two products):

    python -c "from littlemcmc_amd import _build; _build.build(out='build_variants/liblmc_coop_timing.so', extra_flags=['-DLMC_COOP_TIMING'])"
    LMC_HIP_LIB=build_variants/liblmc_coop_timing.so PYTHONPATH=. python tools/coop_timing.py [dim chains tune draws]
"""
import sys

import numpy as np

import littlemcmc_amd as lmc
from littlemcmc_amd import _abi
from littlemcmc_amd import targets as T

a = sys.argv[1:]
d = int(a[0]) if a else 128
chains = int(a[1]) if len(a) > 1 else 65536
tune = int(a[2]) if len(a) > 2 else 200
draws = int(a[3]) if len(a) > 3 else 200
tgt = T.AR1(d, 0.9)
idx = np.arange(d)
cov = 0.9 ** np.abs(idx[:, None] - idx[None, :])
seeds = lmc.distributed.global_seeds(20260928, chains)
step = lmc.NUTS(tgt, d, potential=lmc.QuadPotentialFull(cov.astype("float32")))
eng = step._make_engine(chains)
eng.seed(seeds); eng.set_position(np.zeros((chains, d))); eng.reset_tuning()
eng.reserve(tune + draws, keep_trace=False)
prev = np.zeros((chains, _abi.NUM_COUNTERS), dtype=np.int64)
names = ["waiting for the group (barrier 1)", "multiplying (MFMA + LDS)", "waiting for the product (barrier 2)", "the chain's own work between products"]
for label, lo, n in (("tuning", 0, tune), ("draws", tune, draws)):
    eng.run(tune, lo, n)
    eng.synchronize()
    ct = eng.counters().astype(np.int64)
    dlt = ct - prev
    prev = ct
    ph = np.zeros((chains, 5))
    for k in range(2):
        v = dlt[:, k].astype(np.uint64)
        ph[:, 2 * k] = (v >> np.uint64(32)).astype("d")
        ph[:, 2 * k + 1] = (v & np.uint64(0xffffffff)).astype("d")
    ph[:, 4] = dlt[:, 2]
    leap = dlt[:, _abi.CT_LEAPFROGS].sum()
    products = leap + chains * n      # one product per leapfrog + one for the start state
    tot = ph[:, :4].sum()
    print("%s: %d iterations, %.2f leapfrogs per iteration, %.0f ticks per product per chain" % (label, n, leap / chains / n, tot / products))
    for i in range(4):
        print("   %-40s %5.1f %%   %8.0f ticks per product" % (names[i], 100 * ph[:, i].sum() / tot, ph[:, i].sum() / products))
    print("   after the last product of the launch     %8.0f ticks per iteration" % (ph[:, 4].sum() / chains / n))
eng.close()
