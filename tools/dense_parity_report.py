#!/usr/bin/env python3
"""Report the observed device-vs-oracle differences of the dense-mass replay (tests/test_gpu_dense.py), per golden
case: used to choose the tolerances stated in that test. Run on a GPU box from the repo root."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_gpu_dense as T  # noqa: E402
from tests._gpu_util import INT_STATS  # noqa: E402

golden = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
for name in T.DENSE_E2E:
    g = T._load(golden, name)
    d, tune, draws = int(g["d"]), int(g["tune"]), int(g["draws"])
    ostep, dstep, start = T._oracle_and_device_steps(g)
    snaps, outs = T._snapshots(ostep, start, int(g["seeds"][0]), tune, draws)
    qerr, serr, int_bad, margins_bad = [], [], 0, []
    for tune_flag in (True, False):
        idx = [i for i, s in enumerate(snaps) if s["tune"] == tune_flag]
        if not idx:
            continue
        eng = dstep._make_engine(len(idx))
        eng.set_position(np.stack([snaps[i]["q"] for i in idx]))
        for c, i in enumerate(idx):
            eng.set_rng_state(c, snaps[i]["rng"])
        eng.set_chain_state({k: np.stack([np.asarray(snaps[i][k]) for i in idx]) for k in
                             ("log_step", "log_bar", "hbar", "da_count", "iter_count", "n_samples")})
        if "cov" in snaps[idx[0]]:
            eng.set_dense_state({k: np.stack([np.asarray(snaps[i][k]) for i in idx]) for k in
                                 ("cov", "chol", "fore_mean", "fore_raw_cov", "fore_n", "back_mean", "back_raw_cov",
                                  "back_n", "window", "previous_update")})
        eng.reserve(1, keep_trace=True)
        eng.run(1 if tune_flag else 0, 0, 1)
        q = eng.trace()[:, 0]
        stats = {k: v[:, 0] for k, v in dstep._stats_from_engine(eng, 0, 1).items()}
        for c, i in enumerate(idx):
            want = outs[i]
            bad = any(stats[s][c] != v for s, v in want["stats"].items() if s in INT_STATS)
            if bad:
                int_bad += 1
                margins_bad.append(want["margin"])
                continue
            qerr.append(np.max(np.abs(q[c] - want["q"])) / (1 + np.abs(want["q"]).max()))
            for s, v in want["stats"].items():
                if s not in INT_STATS:
                    serr.append((abs(stats[s][c] - v) / (1 + abs(want["stats"].get("energy", 0.0))), s, i, stats[s][c], v))
        eng.close()
    print("%-32s iters %4d int-mismatch %3d (max margin among them %.2e)  q err max %.2e median %.2e  stat err max %.2e" % (
        name, tune + draws, int_bad, max(margins_bad) if margins_bad else 0.0, max(qerr), np.median(qerr), max(serr)[0]), max(serr)[1:])
