#!/usr/bin/env python3
"""GPU box: how do a device chain and its oracle twin part? For every chain of the e2e goldens run through sample(): the
separation (in units of the comparison's tolerance) of the iterations up to the first one beyond it, and the largest ratio
between consecutive iterations once the separation is measurable -- the evidence behind the JUMP bound of
tests/_gpu_util.py: explain_first_difference."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import littlemcmc_amd as lmc  # noqa: E402
from littlemcmc_amd import engine  # noqa: E402
from oracle import lmc_oracle as orc  # noqa: E402
from oracle import targets as OT  # noqa: E402
from tests._gpu_util import _separation, device_target, kwargs_from  # noqa: E402

engine.DEFAULT_SDOT = "skylakex"
names = ["e2e_nuts_std64", "e2e_nuts_std128", "e2e_nuts_ar1_16", "e2e_nuts_funnel8", "e2e_nuts_diag50", "e2e_nuts_normal1d",
         "e2e_nuts_ar1_128", "e2e_nuts_funnel256", "e2e_nuts_diag1000"]
worst = 0.0
for name in names:
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    d, chains, tune, draws = int(g["d"]), int(g["chains"]), int(g["tune"]), int(g["draws"])
    kw = kwargs_from(g)
    fam = str(g["family"])
    tgt = device_target(fam, d, g["params"])
    trace, stats = lmc.sample(tgt, d, draws=draws, tune=tune, chains=chains, cores=1, progressbar=False,
                              random_seed=int(g["random_seed"]), discard_tuned_samples=False, **kw)
    for c in range(chains):
        got = {n_: stats[n_][c, :, 0] for n_ in stats}
        want = {n_: g["stat_" + n_][c, :, 0] for n_ in stats}
        err, step_err, _e = _separation(trace[c], got, g["trace"][c], want)
        sep = np.maximum(err, step_err)
        first = int(np.argmax(sep > 1.0)) if (sep > 1.0).any() else len(sep)
        seq = sep[max(0, first - 6):first + 1]
        ratios = [seq[i + 1] / seq[i] for i in range(len(seq) - 1) if seq[i] > 1e-6]
        r = max(ratios) if ratios else 0.0
        worst = max(worst, r)
        print("%-20s chain %d: first beyond tolerance at %3d; separation before it %s; largest step ratio %.1f" % (
            name, c, first, " ".join("%.1e" % x for x in seq), r))
print("largest ratio between consecutive iterations of any chain: %.1f" % worst)
