import sys, time
sys.path.insert(0, '.')
import numpy as np
import littlemcmc_amd as lmc
from littlemcmc_amd import targets as T
real_sleep = time.sleep
for label, kw, d in (("full (coop MFMA kernel)", {}, 32), ("adapt_full (per-wave kernel)", {"init": "jitter+adapt_full"}, 16), ("team kernel d=600", {}, 600)):
    fired = []
    def sleep_then_interrupt(dt):
        if not fired:
            fired.append(1); real_sleep(0.3); raise KeyboardInterrupt
        real_sleep(dt)
    time.sleep = sleep_then_interrupt
    tgt = T.AR1(d, 0.9)
    step = None
    if label.startswith("full"):
        idx = np.arange(d); cov = 0.9 ** np.abs(idx[:, None] - idx[None, :])
        step = lmc.NUTS(tgt, d, potential=lmc.QuadPotentialFull(cov))
    t0 = time.perf_counter()
    tr, st = lmc.sample(tgt, d, draws=20000 if d < 100 else 3000, tune=100, chains=2048 if d < 100 else 512, step=step, random_seed=3, discard_tuned_samples=False, **kw)
    time.sleep = real_sleep
    print(label, "interrupted after", tr.shape[1], "iterations in %.2f s" % (time.perf_counter() - t0), "finite:", bool(np.isfinite(tr).all()))
