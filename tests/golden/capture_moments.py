#!/usr/bin/env python3
"""Stationary moments of the REFERENCE sampler on the AR(1) target, as golden data for the full-size GPU tests.

The reference's NUTS (``_Tree.extend`` aliases ``p_sum``) does not leave a correlated Gaussian
exactly invariant: with a diagonal mass matrix its pooled marginal variance on AR(1) rho = 0.9 sits above 1 (about
+4.6 % at d = 8, +2 % at d = 32, +0.3 % at d = 128; a standard normal shows nothing). north_star asks for moments
"within 1e-3 of the CPU reference", so the reference's own stationary moments are captured here -- many independent
chains of the imported reference (sequential path, ``cores=1``), one process per block of chains -- and committed as
plain numbers in ``stationary_moments.npz``. Runs only in the build container (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 nice -n 19 python tests/golden/capture_moments.py [workers]
"""
import multiprocessing as mp
import os
import sys
import tempfile
import time

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

CASES = [   # name, d, chains, tune, draws
    ("ar1_32", 32, 1024, 300, 3500),
    ("ar1_128", 128, 512, 300, 2800),
]
SEED = 20260929


def _import_reference():
    stub = tempfile.mkdtemp(prefix="fp_stub_")
    os.makedirs(os.path.join(stub, "fastprogress"))
    open(os.path.join(stub, "fastprogress", "__init__.py"), "w").close()
    with open(os.path.join(stub, "fastprogress", "fastprogress.py"), "w") as fh:
        fh.write("class progress_bar:\n"
                 "    def __init__(self, gen, total=None, display=True, **kw):\n"
                 "        self.gen, self.total, self.comment = gen, total, ''\n"
                 "    def __iter__(self):\n        return iter(self.gen)\n"
                 "    def update(self, val):\n        pass\n")
    sys.path.insert(0, stub)
    sys.path.insert(0, "/root/reference")
    sys.path.insert(0, ROOT)
    import logging

    import littlemcmc as ref

    logging.getLogger("littlemcmc").setLevel(logging.ERROR)
    return ref


def work(job):
    d, chains, tune, draws, seed = job
    import numpy as np

    ref = _import_reference()
    from oracle import targets

    f = targets.make("ar1", d)
    trace, stats = ref.sample(f, d, draws=draws, tune=tune, chains=chains, cores=1, progressbar=False, random_seed=seed)
    # per-chain sufficient statistics: n, sum, sum of squares per dimension
    return (trace.shape[1] * np.ones(chains), trace.sum(axis=1), (trace ** 2).sum(axis=1),
            float(stats["depth"].mean()), float(stats["tree_size"].sum()))


if __name__ == "__main__":
    import numpy as np

    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    out = {}
    for name, d, chains, tune, draws in CASES:
        t0 = time.time()
        block = 8
        jobs = [(d, block, tune, draws, SEED + 7919 * i) for i in range(chains // block)]
        with mp.get_context("fork").Pool(workers) as pool:
            res = pool.map(work, jobs, chunksize=1)
        n = np.concatenate([r[0] for r in res])
        s1 = np.concatenate([r[1] for r in res])
        s2 = np.concatenate([r[2] for r in res])
        tot = n.sum()
        mean = s1.sum(axis=0) / tot
        var = (s2.sum(axis=0) - tot * mean ** 2) / (tot - 1)
        # Monte-Carlo error of the dimension-averaged variance from the chain-to-chain spread of per-chain estimates
        per_chain = (s2 / n[:, None] - (s1 / n[:, None]) ** 2).mean(axis=1)
        se = per_chain.std(ddof=1) / np.sqrt(len(per_chain))
        out.update({name + "_d": d, name + "_chains": len(n), name + "_draws": draws, name + "_tune": tune,
                    name + "_mean": mean, name + "_var": var, name + "_var_avg": var.mean(), name + "_var_avg_se": se,
                    name + "_depth": np.mean([r[3] for r in res])})
        print("%s: %d chains x %d draws: mean|max %.2e, var-1 avg %+.5f +- %.5f, depth %.2f, %.0f s" % (
            name, len(n), draws, np.abs(mean).max(), var.mean() - 1, se, out[name + "_depth"], time.time() - t0), flush=True)
        np.savez_compressed(os.path.join(HERE, "stationary_moments.npz"), **out)
