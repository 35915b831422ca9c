#!/usr/bin/env python3
"""GPU box: does splitting a job's chains into B sub-blocks on B streams remove the per-launch tail?
B engines of chains/B chains each (own stream), K steps of `ips` iterations enqueued round-robin without host syncs,
against ONE engine with all chains. PYTHONPATH=. python tools/overlap_probe.py [chains] [B] [steps] [ips]"""
import sys
import time

import numpy as np
import torch

import littlemcmc_amd as lmc
from littlemcmc_amd import _abi
from littlemcmc_amd import targets as T

chains = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
K = int(sys.argv[3]) if len(sys.argv) > 3 else 10
ips = int(sys.argv[4]) if len(sys.argv) > 4 else 100
d = 128
tgt = T.AR1(d, 0.9)
seeds = lmc.distributed.global_seeds(20260928, chains)


def make(lo, hi):
    start, step = lmc.init_nuts(tgt, d, random_seed=seeds[lo:hi])
    eng = step._make_engine(hi - lo)
    eng.seed(seeds[lo:hi]); eng.set_position(start); eng.reset_tuning()
    eng.reserve(K * ips, keep_trace=False)
    return eng


def run(engs):
    for e in engs:
        e.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        for e in engs:
            e.run(K * ips // 2, k * ips, ips)
    for e in engs:
        e.synchronize()
    dt = time.perf_counter() - t0
    leap = sum(float(e.counters()[:, _abi.CT_LEAPFROGS].sum()) for e in engs)
    return leap / dt, dt


one = [make(0, chains)]
r1, t1 = run(one)
one[0].close()
bounds = [chains * b // B for b in range(B + 1)]
many = [make(bounds[b], bounds[b + 1]) for b in range(B)]
rB, tB = run(many)
print("chains %d: one engine %.3e leapfrogs/s (%.3f s);  %d engines on %d streams %.3e (%.3f s)  ratio %.3f" % (
    chains, r1, t1, B, B, rB, tB, rB / r1))
