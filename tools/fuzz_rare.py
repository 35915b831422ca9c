#!/usr/bin/env python3
"""Differential fuzz of the rare paths (GPU box): chains entered far out in the tails with step sizes up to the
stability limit -- energy drops of hundreds to a thousand (weight-offset moves, rescaled subtree stacks), divergences in
the first or second leaf of a pair, NaN energies, trees cut by max_treedepth -- fused kernels (one wave and teams)
against the numpy oracle for the first iterations, every sampler statistic compared.
Usage: python tools/fuzz_rare.py [n_cases] [seed] [lds_plan]      (lds_plan: auto | shallow | deep -- lmc_config.lds_plan)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import littlemcmc_amd as lmc  # noqa: E402
from oracle import lmc_oracle as orc  # noqa: E402
from oracle import targets as OT  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
lds_plan = sys.argv[3] if len(sys.argv) > 3 else "auto"
bad = 0
seen = {"rescale": 0, "deep_rescale": 0, "diverging": 0, "maxdepth": 0}
for case in range(n_cases):
    fam = str(rs.choice(["std_normal", "ar1", "funnel", "diag_gaussian"]))
    d = int(rs.choice([1, 2, 3, 5, 8, 16, 33, 64, 100, 128, 200, 300]))
    if fam == "funnel":
        d = max(d, 2)
    chains = 8
    far = float(rs.choice([0.0, 3.0, 10.0, 25.0, 40.0])) / (1.0 if fam != "funnel" else 8.0)
    frac = float(rs.choice([0.15, 0.3, 0.5, 0.8, 0.95, 0.99, 1.2]))        # fraction of the unit-Gaussian stability limit 2
    md = int(rs.choice([3, 6, 10]))
    sc = 2.0 * frac * d ** 0.25 / (1.0 if fam in ("std_normal", "funnel") else (4.4 if fam == "ar1" else 100.0) ** (0.0 if rs.rand() < 0.5 else 0.5))
    f = OT.make(fam, d)
    from tests._gpu_util import device_target
    tgt = device_target(fam, d, f.params())
    starts = [far * rs.randn(d) / np.sqrt(max(d, 1)) * np.sqrt(d) * (1.0 if rs.rand() < 0.5 else 1.0 / np.sqrt(d)) for _ in range(chains)]
    if fam == "std_normal" and rs.rand() < 0.7:
        # aim at energy drops of 200-990 (below Emax, around the 600 that moves the weight offset and the 745 that underflows):
        # leapfrog on a unit Gaussian loses up to H0 eps^2 / 4 on the way in
        eps = sc / d ** 0.25
        starts = []
        for _ in range(chains):
            u = rs.randn(d)
            starts.append(u / np.linalg.norm(u) * np.sqrt(8.0 * rs.uniform(200.0, 990.0)) / max(eps, 0.3))
    seeds = [int(x) for x in rs.randint(1, 10 ** 6, size=chains)]
    draws = 4
    ostep = orc.Step(f, d, kind="nuts", adapt_step_size=False, step_scale=sc, max_treedepth=md)
    step = lmc.NUTS(tgt, d, adapt_step_size=False, step_scale=sc, max_treedepth=md, lds_plan=lds_plan)
    try:
        ot, ost = orc.sample(f, d, draws=draws, tune=0, step=ostep, chains=chains, start=starts, random_seed=seeds, discard_tuned_samples=False)
        o_err = None
    except Exception as e:   # Bad initial energy etc.
        o_err = type(e).__name__
    try:
        gt, gst = lmc.sample(tgt, d, draws=draws, tune=0, step=step, chains=chains, start=starts, random_seed=seeds, discard_tuned_samples=False)
        g_err = None
    except Exception as e:
        g_err = type(e).__name__
    if o_err or g_err:
        ok = o_err == g_err
        print("case %3d %-13s d=%3d far=%5.1f frac=%.2f md=%2d : both raised %s / %s : %s" % (case, fam, d, far, frac, md, o_err, g_err, "ok" if ok else "FAIL"))
        bad += 0 if ok else 1
        continue
    # compare iteration by iteration per chain until the first integer mismatch (after one, the chains have parted)
    ok = True
    msg = ""
    for c in range(chains):
        for t in range(draws):
            ints = all(int(gst[k][c, t, 0]) == int(ost[k][c, t, 0]) for k in ("depth", "tree_size", "diverging"))
            if not ints:
                # a knife-edge decision shows as a margin-free difference; report it for inspection
                ok = False
                msg += " chain %d it %d ints dev(%d,%d,%d) orc(%d,%d,%d);" % (c, t, gst["depth"][c, t, 0], gst["tree_size"][c, t, 0], gst["diverging"][c, t, 0], ost["depth"][c, t, 0], ost["tree_size"][c, t, 0], ost["diverging"][c, t, 0])
                break
            for k, tol in (("max_energy_error", 1e-7), ("mean_tree_accept", 1e-6), ("energy", 1e-8), ("energy_error", 1e-6)):
                a, b = float(gst[k][c, t, 0]), float(ost[k][c, t, 0])
                if not (np.isclose(a, b, rtol=tol, atol=1e-7) or (np.isnan(a) and np.isnan(b)) or (np.isinf(a) and a == b)):
                    ok = False
                    msg += " chain %d it %d %s dev %.12g orc %.12g;" % (c, t, k, a, b)
            if not np.allclose(gt[c, t], ot[c, t], rtol=1e-7, atol=1e-8):
                ok = False
                msg += " chain %d it %d |dq| %.2e;" % (c, t, np.abs(gt[c, t] - ot[c, t]).max())
                break
    mde = ost["max_energy_error"][:, :, 0]
    seen["rescale"] += int((mde < -600).any())
    seen["deep_rescale"] += int(((mde < -600) & (ost["depth"][:, :, 0] >= 3)).any())
    seen["diverging"] += int(ost["diverging"].any())
    seen["maxdepth"] += int((ost["depth"] == md).any())
    print("case %3d %-13s d=%3d far=%5.1f frac=%.2f md=%2d  min dE %9.1f  depth<=%d div %d : %s%s" % (
        case, fam, d, far, frac, md, mde.min(), ost["depth"].max(), int(ost["diverging"].sum()), "ok" if ok else "FAIL", msg[:300]))
    bad += 0 if ok else 1
print("lds_plan %s; cases that exercised:" % lds_plan, seen, " failures:", bad)
sys.exit(1 if bad else 0)
