#!/usr/bin/env python3
"""leapfrog-steps/s of the tick path (TorchTarget) next to the fused kernel on the same workload.
Usage (GPU box): python tools/bench_torch_target.py [chains] [dim] [iters]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import littlemcmc_amd as lmc  # noqa: E402
from littlemcmc_amd import _abi  # noqa: E402
from littlemcmc_amd.targets import TorchTarget  # noqa: E402

chains = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 200
rho = 0.9
c = 1.0 / (1.0 - rho * rho)
diag = torch.full((d,), (1.0 + rho * rho) * c, dtype=torch.float64, device="cuda")
diag[0] = c
diag[-1] = c
off = -rho * c


def ar1(q):
    pq = diag * q
    pq[:, 1:] += off * q[:, :-1]
    pq[:, :-1] += off * q[:, 1:]
    return -0.5 * (q * pq).sum(dim=1), -pq


# The same density written for the GPU instead of transcribed from the formula: g = -P q as ONE float64 GEMM with the
# (negated) tridiagonal precision matrix as a dense d x d operand (rocBLAS: matrix cores), logp = 0.5 q.g as a product and a
# row sum -- three kernels and ~4 passes over the (chains x d) arrays instead of nine kernels and ~20 passes. What a user's
# callable costs is the user's; this line shows how much of the tick is the callable (the tick kernel itself: profiles/r05_tick_*).
negP = torch.zeros((d, d), dtype=torch.float64, device="cuda")
idx = torch.arange(d, device="cuda")
negP[idx, idx] = -diag
negP[idx[1:], idx[:-1]] = -off
negP[idx[:-1], idx[1:]] = -off


def ar1_gemm(q):
    g = q @ negP
    return 0.5 * (q * g).sum(dim=1), g


only = os.environ.get("LMC_TICK_BENCH_ONLY")   # substring filter (profiling runs)
variants = (("fused AR1Target", lmc.targets.AR1(d, rho)), ("TorchTarget (eager, 9 torch kernels / tick)", TorchTarget(d, ar1)),
            ("TorchTarget (graph=True: fn replayed as a HIP graph)", TorchTarget(d, ar1, graph=True)),
            ("TorchTarget (GEMM form, 4 torch kernels / tick)", TorchTarget(d, ar1_gemm)),
            ("TorchTarget (GEMM form, graph=True)", TorchTarget(d, ar1_gemm, graph=True)))
for name, tgt in variants:
    if only and only not in name:
        continue
    step = lmc.NUTS(tgt, d)
    eng = step._make_engine(chains)
    eng.seed(np.arange(chains, dtype=np.uint32) + 1)
    eng.set_position(np.zeros(d))
    eng.reset_tuning()
    eng.reserve(iters, keep_trace=False)
    eng.run(iters // 2, 0, 5)
    eng.synchronize()
    torch.cuda.synchronize()
    base = eng.counters()[:, _abi.CT_LEAPFROGS].sum()
    t0 = time.perf_counter()
    eng.run(iters // 2, 5, iters - 5)
    eng.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    leaps = eng.counters()[:, _abi.CT_LEAPFROGS].sum() - base
    ticks = getattr(eng, "ticks", 0)
    print("%-54s %9.3e leapfrog-steps/s  (%.2f s, %d chains x d=%d x %d iterations%s)" % (
        name, leaps / dt, dt, chains, d, iters - 5, (", %d ticks, %.0f us/tick" % (ticks, 1e6 * dt / max(ticks, 1))) if ticks else ""))
    eng.close()
