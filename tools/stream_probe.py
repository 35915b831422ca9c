#!/usr/bin/env python3
"""Where the time of a streamed sample() goes (GPU box): pinning the result arrays, the window copies on an idle GPU, the same
copies under a running launch, freeing.   python tools/stream_probe.py [chains] [draws]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import littlemcmc_amd as lmc  # noqa: E402
from littlemcmc_amd import sampling  # noqa: E402
from littlemcmc_amd.engine import StreamedResults  # noqa: E402

chains = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
draws = int(sys.argv[2]) if len(sys.argv) > 2 else 250
d, tune = 128, 300
tgt = lmc.targets.AR1(d, 0.9)
seeds = sampling._derive_seeds(1, chains)
start, step = lmc.init_nuts(tgt, d, random_seed=seeds)
n_total = tune + draws
gib = chains * draws * (d * 8 + 82) / 2.0 ** 30

t0 = time.perf_counter()
out = StreamedResults(chains, draws, tune, d, step._result_planes())
t_pin = time.perf_counter() - t0
print("pinning %.2f GiB in 12 arrays: %.3f s (%.1f GiB/s)" % (gib, t_pin, gib / t_pin))
t0 = time.perf_counter()
page = np.empty((chains, draws, d))
page[:] = 0.0
t_touch = time.perf_counter() - t0
print("first touch of a pageable %.2f GiB array: %.3f s" % (page.nbytes / 2.0 ** 30, t_touch))
del page

# pinning by parts: a numpy allocation (numpy madvises huge pages for large blocks) pre-faulted by T threads on disjoint
# ranges, then registered in one call
import ctypes as C
import threading

from littlemcmc_amd import _abi

lib = _abi.load()
nbytes = chains * draws * d * 8
for T_ in (1, 4, 8, 16, 32):
    t0 = time.perf_counter()
    raw = np.empty(nbytes, dtype=np.uint8)
    base = raw.ctypes.data
    step_b = -(-nbytes // T_ // (1 << 21)) * (1 << 21)
    ranges = [(o, min(step_b, nbytes - o)) for o in range(0, nbytes, step_b)]

    def touch(o, n):
        C.memset(base + o, 0, n)

    th = [threading.Thread(target=touch, args=r) for r in ranges]
    [t.start() for t in th]
    [t.join() for t in th]
    t_touch = time.perf_counter() - t0
    t0 = time.perf_counter()
    rc = lib.lmc_host_register(C.c_void_p(base), nbytes)
    t_reg = time.perf_counter() - t0
    print("%2d threads, %.2f GiB: allocate + prefault %.3f s (%.1f GiB/s), hipHostRegister %.3f s (%.1f GiB/s), rc %d" % (
        len(ranges), nbytes / 2.0 ** 30, t_touch, nbytes / 2.0 ** 30 / t_touch, t_reg, nbytes / 2.0 ** 30 / t_reg, rc))
    t0 = time.perf_counter()
    lib.lmc_host_unregister(C.c_void_p(base))
    t_un = time.perf_counter() - t0
    t0 = time.perf_counter()
    del raw
    print("   unregister %.3f s, free %.3f s" % (t_un, time.perf_counter() - t0))

eng = step._make_engine(chains)
eng.seed(seeds)
eng.set_position(start)
eng.reset_tuning()
eng.reserve(n_total, keep_trace=True, trace_begin=tune)
t0 = time.perf_counter()
sampling._run_job(eng, tune, n_total, 100, False)
t_job = time.perf_counter() - t0
print("job alone (%d iterations, launches of 100): %.3f s" % (n_total, t_job))

for rep in range(2):
    t0 = time.perf_counter()
    for first in range(tune, n_total, 50):
        eng.copy_window_async(out, first, min(50, n_total - first))
    t_enq = time.perf_counter() - t0
    eng.copy_wait()
    t_idle = time.perf_counter() - t0
    print("window copies on an idle GPU: enqueue %.3f s, done %.3f s (%.1f GiB/s)" % (t_enq, t_idle, gib / t_idle))

# the same under a running job, for several copy geometries
def job(with_copies):
    eng.reset_tuning()
    eng.seed(seeds)
    eng.set_position(start)
    eng.synchronize()
    t0 = time.perf_counter()
    cb = (lambda f, n: eng.copy_window_async(out, max(f, tune), f + n - max(f, tune)) if f + n > tune else None) if with_copies else None
    sampling._run_job(eng, tune, n_total, 100, False, on_enqueued=cb)
    t_k = time.perf_counter() - t0
    eng.copy_wait()
    return t_k, time.perf_counter() - t0


for wg in (0, 4, 8, 16, 32, 64, 256, 1024):
    out.copy_workgroups = wg
    t0 = time.perf_counter()
    for first in range(tune, n_total, 50):
        eng.copy_window_async(out, first, min(50, n_total - first))
    eng.copy_wait()
    t_idle = time.perf_counter() - t0
    a = job(False)
    b = job(True)
    print("copy_workgroups %4d: idle-GPU copy %.3f s (%.1f GiB/s) | job alone %.3f s | job with its windows copied under the following launches %.3f s (+%.3f s)"
          % (wg, t_idle, gib / t_idle, a[1], b[1], b[1] - a[1]))
out.copy_workgroups = 0
t0 = time.perf_counter()
tr = eng.trace(tune, draws)
t_old = time.perf_counter() - t0
print("lmc_engine_get_trace into a fresh pageable array: %.3f s (%.1f GiB/s)" % (t_old, tr.nbytes / 2.0 ** 30 / t_old))
assert np.array_equal(tr[::997], out.trace[::997])
eng.close()
t0 = time.perf_counter()
del out
print("freeing the pinned arrays: %.3f s" % (time.perf_counter() - t0))
