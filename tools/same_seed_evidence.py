#!/usr/bin/env python3
"""How long do two same-seed chains stay together when ONE float32 rounding differs? (CPU only; writes
profiles/r04_same_seed_decorrelation.json; with --strict-order-ceiling also the ceiling of a strict-summation-order device
build, profiles/r05_same_seed_decorrelation.json)

north_star asks for draws whose per-chain moments "match the reference CPU sampler on identical RNG seeds". Whole tuned
chains cannot agree bit for bit across machines: dual averaging feeds the acceptance statistic back into the step size
(gain ~ sqrt(t) / ((t + t0) gamma) ~ 3 early on), so a 1-ulp difference in one energy doubles every iteration or two
until a tree decision flips. This tool measures that on the CPU alone, with the ORACLE against ITSELF: the same chain is
run twice, once with numpy's own float32 dot for the start state's kinetic energy (the host BLAS: OpenBLAS
sdot_k_SKYLAKEX on AVX-512 hosts) and once with the summation order of the other x86-64 OpenBLAS kernel
(sdot_k_HASWELL, what the reference computes on an AVX2 host; littlemcmc_amd/_blas_probe.py restates both). Nothing
else differs -- same seeds, same algorithm, same float64 arithmetic -- and the two CPU chains part after a few dozen
iterations, exactly like a device chain and the oracle do. The table lists, per golden configuration and chain, the
first iteration whose integer statistics differ."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from littlemcmc_amd import _abi  # noqa: E402
from littlemcmc_amd._blas_probe import detect_sdot_mode, emulate_sdot  # noqa: E402
from oracle import lmc_oracle as orc  # noqa: E402
from oracle import targets as OT  # noqa: E402

RUNS = [("e2e_nuts_std64", "std_normal", 64, 2, 250, 150), ("e2e_nuts_std128", "std_normal", 128, 2, 110, 20),
        ("e2e_nuts_ar1_16", "ar1", 16, 4, 250, 150), ("e2e_nuts_ar1_128", "ar1", 128, 2, 250, 50),
        ("e2e_nuts_diag50", "diag_gaussian", 50, 2, 250, 100)]
SEED = 20260928


def first_difference(a, b):
    bad = None
    for k in ("depth", "tree_size", "diverging"):
        idx = np.nonzero(a[k] != b[k])[0]
        if len(idx) and (bad is None or idx[0] < bad):
            bad = int(idx[0])
    return bad


def one_ulp_experiment():
    """The same question with the smallest possible difference: ONE accept statistic of ONE iteration moved by one unit in
    the last place (what a device exp() that differs from glibc's by an ulp does, once). Everything else identical."""
    rows = []
    for name, fam, d, chains, tune, draws in RUNS:
        f = OT.make(fam, d)
        base = orc.sample(f, d, draws=draws, tune=tune, chains=1, random_seed=SEED, discard_tuned_samples=False)[1]
        real_update = orc.DualAverage.update
        state = {"n": 0}

        def perturbed(self, accept, tune_flag, _real=real_update, _st=state):
            _st["n"] += 1
            if _st["n"] == 11:                     # the eleventh iteration's statistic, nudged by one ulp
                accept = np.nextafter(accept, 2.0)
            return _real(self, accept, tune_flag)

        orc.DualAverage.update = perturbed
        try:
            pert = orc.sample(f, d, draws=draws, tune=tune, chains=1, random_seed=SEED, discard_tuned_samples=False)[1]
        finally:
            orc.DualAverage.update = real_update
        a = {k: base[k][0, :, 0] for k in base}
        b = {k: pert[k][0, :, 0] for k in pert}
        fd = first_difference(a, b)
        rel = np.abs(a["step_size"] - b["step_size"]) / a["step_size"]
        growth = [float(rel[i]) for i in (11, 20, 30, 40) if i < len(rel)]
        rows.append({"golden": name, "dim": d, "first_iteration_with_a_different_tree": fd,
                     "relative_step_size_difference_at_iterations_11_20_30_40": growth})
        print("%-20s one ulp in one accept statistic (iteration 10): trees differ from iteration %s; step-size difference %s"
              % (name, fd, ["%.1e" % g for g in growth]))
    return rows


def strict_order_ceiling():
    """Review item (round 4, N3): would a device build that evaluates every dot product in the host BLAS's summation order
    ("strict order") track a golden chain for all its iterations? Here is its CEILING, measured on the CPU: the oracle
    against itself with EVERY sum bit-identical (it is the same numpy) and the only difference the one a strict-order device
    build cannot remove -- its exp / log differ from glibc's by up to an ulp, so the acceptance statistic of an iteration
    (nuts.py:421-425: an exp of a difference of logaddexp chains) lands on a neighbouring double about every other time.
    Every iteration's statistic is moved by -1 / 0 / +1 ulp at random (seeded); everything else is identical. The chains
    part at the same distance as with the float32 start-energy difference above, i.e. where device chains part today:
    bit-identical sums would not lengthen the prefix; bit-identical transcendentals would be needed as well (numpy's exp /
    log, glibc's exp / log1p / log with their tables and FMA contraction patterns)."""
    rows = []
    for name, fam, d, chains, tune, draws in RUNS:
        f = OT.make(fam, d)
        base = orc.sample(f, d, draws=draws, tune=tune, chains=1, random_seed=SEED, discard_tuned_samples=False)[1]
        real_update = orc.DualAverage.update
        firsts = []
        for trial in range(3):
            rs = np.random.RandomState(1000 + trial)

            def perturbed(self, accept, tune_flag, _real=real_update, _rs=rs):
                k = _rs.randint(-1, 2)
                if k:
                    accept = np.nextafter(accept, 2.0 if k > 0 else -1.0)
                return _real(self, accept, tune_flag)

            orc.DualAverage.update = perturbed
            try:
                pert = orc.sample(f, d, draws=draws, tune=tune, chains=1, random_seed=SEED, discard_tuned_samples=False)[1]
            finally:
                orc.DualAverage.update = real_update
            a = {k: base[k][0, :, 0] for k in base}
            b = {k: pert[k][0, :, 0] for k in pert}
            firsts.append(first_difference(a, b))
        rows.append({"golden": name, "dim": d, "iterations": tune + draws,
                     "first_iteration_with_a_different_tree_3_trials": firsts})
        print("%-20s +-1 ulp in every accept statistic: trees differ from iteration %s (of %d)" % (name, firsts, tune + draws))
    return rows


def main():
    host = detect_sdot_mode()
    other = _abi.SDOT_OPENBLAS_HASWELL if host == _abi.SDOT_OPENBLAS_SKYLAKEX else _abi.SDOT_OPENBLAS_SKYLAKEX
    names = {_abi.SDOT_OPENBLAS_SKYLAKEX: "sdot_k_SKYLAKEX", _abi.SDOT_OPENBLAS_HASWELL: "sdot_k_HASWELL"}
    rows = []
    for name, fam, d, chains, tune, draws in RUNS:
        f = OT.make(fam, d)
        out = {}
        for label, hook in (("host", None), ("other", lambda x, y: emulate_sdot(x, y, other))):
            orc.START_SDOT = hook
            try:
                _tr, st = orc.sample(f, d, draws=draws, tune=tune, chains=chains, random_seed=SEED, discard_tuned_samples=False)
            finally:
                orc.START_SDOT = None
            out[label] = st
        for c in range(chains):
            a = {k: out["host"][k][c, :, 0] for k in out["host"]}
            b = {k: out["other"][k][c, :, 0] for k in out["other"]}
            fd = first_difference(a, b)
            rel = np.abs(a["energy"] - b["energy"]) / (1.0 + np.abs(a["energy"]))
            first_ulp = int(np.nonzero(rel > 0)[0][0]) if (rel > 0).any() else None
            rows.append({"golden": name, "target": fam, "dim": d, "chain": c, "iterations": tune + draws,
                         "first_iteration_with_any_energy_difference": first_ulp,
                         "first_iteration_with_a_different_tree": fd})
            print("%-20s chain %d: energies differ from iteration %s, trees from iteration %s of %d"
                  % (name, c, first_ulp, fd, tune + draws))
    ulp_rows = one_ulp_experiment()
    doc = {"what": __doc__.split("\n\n")[1].replace("\n", " "),
           "one_ulp_experiment": {"what": one_ulp_experiment.__doc__.replace("\n", " "), "rows": ulp_rows},
           "host_sdot": names[host], "other_sdot": names[other], "rows": rows,
           "summary": {"median_first_different_tree": float(np.median([r["first_iteration_with_a_different_tree"] for r in rows
                                                                       if r["first_iteration_with_a_different_tree"] is not None])),
                       "chains_never_differing": sum(r["first_iteration_with_a_different_tree"] is None for r in rows)}}
    if "--strict-order-ceiling" in sys.argv:
        doc["strict_order_ceiling"] = {"what": strict_order_ceiling.__doc__.replace("\n", " "), "rows": strict_order_ceiling()}
    path = os.path.join(ROOT, "profiles", "r05_same_seed_decorrelation.json" if "strict_order_ceiling" in doc else "r04_same_seed_decorrelation.json")
    with open(path, "w") as fh:
        json.dump(doc, fh, indent=1)
    print("wrote", path, doc["summary"])


if __name__ == "__main__":
    main()
