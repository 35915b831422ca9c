#!/usr/bin/env python3
"""End-to-end wall time of the literal drop-in call (GPU box): ``lmc.sample(...)`` returning the trace and every statistic
as numpy arrays (sampling.py:207-222 of the reference), with the results streamed into pinned host arrays while the job
runs (the default) and copied in one piece after it (stream_results=False), next to the kernel-only time of the same job
(draws left in HBM). "hidden" = 1 - (streamed - kernel) / (after - kernel): the share of the copy-out that no longer shows.

    python tools/sample_e2e.py [c2|c3|std128] [chains] [tune] [draws] [repeats]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import littlemcmc_amd as lmc  # noqa: E402
from littlemcmc_amd import _abi  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
shape = {"c2": ("std_normal", 64, 4096), "c3": ("ar1", 128, 65536), "std128": ("std_normal", 128, 65536)}[cfg]
chains = int(sys.argv[2]) if len(sys.argv) > 2 else shape[2]
tune = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
draws = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 2
d = shape[1]
tgt = lmc.targets.AR1(d, 0.9) if shape[0] == "ar1" else lmc.targets.StdNormal(d)
SEED = 20260928
out_gib = chains * draws * (d * 8 + 82) / 2.0 ** 30
print("%s: %d chains x d=%d, tune %d + draws %d; the call returns %.2f GiB (trace + 11 statistics)" % (cfg, chains, d, tune, draws, out_gib))


def kernel_only():
    """The same job, same launch schedule, draws left in HBM: what sample() costs before anything is copied out."""
    from littlemcmc_amd import sampling

    seeds = sampling._derive_seeds(SEED, chains)
    start, step = lmc.init_nuts(tgt, d, random_seed=seeds)
    eng = step._make_engine(chains)
    try:
        eng.seed(seeds)
        eng.set_position(start)
        eng.reset_tuning()
        eng.reserve(tune + draws, keep_trace=True, trace_begin=tune)
        slots = eng.resident_chains()
        per = max(1, min(tune + draws, 4000))
        if slots and slots < chains < 6 * slots:
            per = 100
        elif slots and chains >= 6 * slots:
            per = [100, 100, 100, 100, 500]
        eng.synchronize()
        t0 = time.perf_counter()
        sampling._run_job(eng, tune, tune + draws, per, False)
        dt = time.perf_counter() - t0
        leaps = int(eng.counters()[:, _abi.CT_LEAPFROGS].sum())
        return dt, leaps
    finally:
        eng.close()


def call(stream):
    t0 = time.perf_counter()
    trace, stats = lmc.sample(tgt, d, draws=draws, tune=tune, chains=chains, random_seed=SEED, progressbar=False,
                              stream_results=stream)
    dt = time.perf_counter() - t0
    chk = float(trace[::max(1, chains // 64), -1].sum()) + float(stats["tree_size"][::max(1, chains // 64)].sum())
    del trace, stats
    return dt, chk


kernel_only()   # warm-up: code objects, allocator, first pinned allocation
call("direct") if out_gib < 4 else None
rows = []
for r in range(reps):
    tk, leaps = kernel_only()
    ts, c1 = call("direct")
    tw, c3 = call("windows")
    ta, c2 = call(False)
    assert c1 == c2 == c3, (c1, c2, c3)
    hidden = 1.0 - (ts - tk) / max(ta - tk, 1e-9)
    rows.append((tk, ts, ta, hidden))
    print("run %d: kernel only %.3f s (%.3e leapfrog-steps/s) | sample() direct %.3f s | windows %.3f s | copy-after %.3f s | copy-out %.3f s -> %.3f s (direct; windows %.3f s), %.0f %% hidden"
          % (r, tk, leaps / tk, ts, tw, ta, ta - tk, ts - tk, tw - tk, 100 * hidden))
best = min(rows, key=lambda x: x[1])
print("best: kernel %.3f s, direct %.3f s (%.2fx kernel), copy-after %.3f s (%.2fx kernel); end-to-end rate streamed %.3e leapfrog-steps/s"
      % (best[0], best[1], best[1] / best[0], best[2], best[2] / best[0], leaps / best[1]))
