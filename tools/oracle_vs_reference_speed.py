#!/usr/bin/env python3
"""Build-container only: time the imported reference against the numpy oracle on the same chain (same seed,
same target), so that the GPU box's CPU baseline (oracle) can be related to the true reference."""
import os
import sys
import tempfile
import time

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
stub = tempfile.mkdtemp()
os.makedirs(os.path.join(stub, "fastprogress"))
open(os.path.join(stub, "fastprogress", "__init__.py"), "w").close()
open(os.path.join(stub, "fastprogress", "fastprogress.py"), "w").write(
    "class progress_bar:\n    def __init__(self, gen, total=None, display=True, **kw):\n        self.gen=gen; self.comment=''\n"
    "    def __iter__(self):\n        return iter(self.gen)\n    def update(self, v):\n        pass\n")
sys.path.insert(0, stub)
sys.path.insert(0, "/root/reference")
import logging

import numpy as np

import littlemcmc as ref
from oracle import lmc_oracle as orc
from oracle import targets as OT

logging.getLogger("littlemcmc").setLevel(logging.ERROR)
for fam, d, tune, draws in [("ar1", 128, 300, 200), ("std_normal", 128, 300, 300)]:
    f = OT.make(fam, d)
    t0 = time.perf_counter()
    tr, st = ref.sample(f, d, draws=draws, tune=tune, chains=1, cores=1, progressbar=False, random_seed=7,
                        discard_tuned_samples=False)
    t_ref = time.perf_counter() - t0
    leap = float(st["tree_size"].sum())
    t0 = time.perf_counter()
    otr, ost = orc.sample(f, d, draws=draws, tune=tune, chains=1, random_seed=7, discard_tuned_samples=False)
    t_or = time.perf_counter() - t0
    assert float(ost["tree_size"].sum()) == leap and np.array_equal(otr, tr)
    print("%-10s d=%d: %d leapfrogs | reference %.2f s = %.0f leap/s | oracle %.2f s = %.0f leap/s | oracle/reference = %.2f"
          % (fam, d, leap, t_ref, leap / t_ref, t_or, leap / t_or, t_ref / t_or))
