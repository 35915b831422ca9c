import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import littlemcmc_amd as lmc
from oracle import lmc_oracle as orc, targets as OT

d, chains, tune, draws, seed = 16, 2, 120, 10, 1234
trace, stats, eng = lmc.sample(lmc.targets.StdNormal(d), d, draws=draws, tune=tune, chains=chains, random_seed=seed,
                          discard_tuned_samples=False, return_engine=True)
otrace, ostats, m = orc.sample(OT.StdNormal(d), d, draws=draws, tune=tune, chains=chains, random_seed=seed,
                               discard_tuned_samples=False, record_margins=True)
np.set_printoptions(precision=17, linewidth=200)
for c in range(chains):
    for i in range(tune + draws):
        dq = np.max(np.abs(trace[c, i] - otrace[c, i]))
        row = {k: (stats[k][c, i, 0], ostats[k][c, i, 0]) for k in stats}
        bad = [k for k, (a, b) in row.items() if not np.isclose(a, b, rtol=1e-9, atol=1e-9)]
        if dq > 1e-9 or bad:
            print("chain", c, "iter", i, "dq", dq, "margin", m[c, i])
            for k, (a, b) in row.items():
                print("   %-18s gpu %-24r ref %-24r %s" % (k, a, b, "<--" if k in bad else ""))
            if i > 0:
                print("   prev iter stats:", {k: (stats[k][c, i-1, 0], ostats[k][c, i-1, 0]) for k in ("step_size", "step_size_bar", "mean_tree_accept")})
            break
    else:
        print("chain", c, "all", tune + draws, "iterations match")
print(eng.adapt_state())
