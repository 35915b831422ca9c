#!/usr/bin/env python3
"""Differential fuzz (GPU box): random small configurations, every engine path against the fused diagonal/dense kernels
and the numpy oracle for the first iterations. Prints one line per case; exits non-zero on the first disagreement.
Usage: python tools/fuzz_paths.py [n_cases] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import littlemcmc_amd as lmc  # noqa: E402
from littlemcmc_amd.targets import TorchTarget  # noqa: E402
from oracle import lmc_oracle as orc  # noqa: E402
from oracle import targets as OT  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def torch_ar1(d, rho=0.9):
    c = 1.0 / (1.0 - rho * rho)
    diag = torch.full((d,), (1.0 + rho * rho) * c, dtype=torch.float64, device="cuda")
    diag[0] = c
    diag[d - 1] = c
    off = -rho * c

    def fn(q):
        pq = diag * q
        if d > 1:
            pq[:, 1:] += off * q[:, :-1]
            pq[:, :-1] += off * q[:, 1:]
        return -0.5 * (q * pq).sum(dim=1), -pq

    return TorchTarget(d, fn)


bad = 0
for case in range(n_cases):
    d = int(rs.choice([1, 2, 3, 7, 16, 31, 64, 65, 100, 128, 129, 200, 256]))
    kind = str(rs.choice(["nuts", "hmc"]))
    mass = str(rs.choice(["adapt_diag", "jitter+adapt_diag", "adapt_full", "jitter+adapt_full"]))
    chains = int(rs.choice([1, 3, 17]))
    tune, draws = int(rs.choice([0, 5, 12])), int(rs.choice([1, 6]))
    seed = int(rs.randint(1, 10 ** 6))
    kw = dict(draws=draws, tune=tune, chains=chains, random_seed=seed, discard_tuned_samples=False)
    fused_t, torch_t, of = lmc.targets.AR1(d), torch_ar1(d), OT.make("ar1", d)
    if kind == "hmc":
        start0 = np.zeros(d)
        def steps(t):
            pot = (lmc.QuadPotentialFullAdapt(d, start0, np.eye(d), 10) if "full" in mass
                   else lmc.QuadPotentialDiagAdapt(d, start0, np.ones(d), 10))
            return lmc.HamiltonianMC(t, d, potential=pot, path_length=1.0)
        a = lmc.sample(fused_t, d, step=steps(fused_t), start=start0, **kw)
        b = lmc.sample(torch_t, d, step=steps(torch_t), start=start0, **kw)
        opot = orc.FullAdaptPotential(d, start0, np.eye(d), 10) if "full" in mass else orc.DiagAdaptPotential(d, start0, np.ones(d), 10)
        o = orc.sample(of, d, step=orc.Step(of, d, kind="hmc", potential=opot, path_length=1.0), start=start0, **kw)
        key = "n_steps"
    else:
        a = lmc.sample(fused_t, d, init=mass, **kw)
        b = lmc.sample(torch_t, d, init=mass, **kw)
        o = orc.sample(of, d, init=mass, **kw)
        key = "tree_size"
    n = min(4, tune + draws)
    # dense: float32-born momentum / start energy, chained over n iterations (tests/test_gpu_dense.py: 6e-7 per iteration)
    tol = 5e-4 if "full" in mass else 1e-8
    ok = True
    for name, (tr, st) in (("tick", b), ("oracle", o)):
        same_int = np.array_equal(st[key][:, :n], a[1][key][:, :n])
        close = np.allclose(tr[:, :n], a[0][:, :n], rtol=tol, atol=tol)
        if not (same_int and close):
            ok = False
            print("   MISMATCH vs %s: ints %s, max |dq| %.3e" % (name, same_int, np.abs(tr[:, :n] - a[0][:, :n]).max()))
    fin = np.isfinite(a[0]).all() and np.isfinite(b[0]).all()
    print("case %2d d=%3d %-4s %-18s chains=%2d tune=%2d draws=%d seed=%6d : %s" % (
        case, d, kind, mass, chains, tune, draws, seed, "ok" if ok and fin else "FAIL"))
    bad += 0 if (ok and fin) else 1
sys.exit(1 if bad else 0)
