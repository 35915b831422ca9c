"""Helpers shared by the -m gpu parity tests: build a device target / engine from a golden case and
compare device chains with the oracle's, aware of decision margins.

Why margins: the reference evaluates the start state's kinetic energy with a float32 BLAS ``sdot``
whose summation order is CPU/BLAS specific; the device sums the same float32 products in float64 and
rounds once. The two start energies can therefore differ by a few float32 ulps (~1e-5 at d=128),
which can flip a multinomial/Metropolis decision only when |log U - log p| is below that. The oracle
reports the smallest such margin per transition; chains are required to match bit-for-bit (integer
stats) up to the first transition whose margin is below ``FRAGILE`` -- and there must be few of those.
"""
import numpy as np

import littlemcmc_amd as lmc
from littlemcmc_amd import targets as T

FRAGILE = 1e-4          # |log U - log p| below this may legitimately flip (float32 start energy)
RTOL_Q = 1e-7           # positions: different reduction order / libm ulps, amplified over a chain
INT_STATS = ("depth", "tree_size", "diverging", "n_steps", "accepted", "tune")


def device_target(family, d, params):
    family = str(family)
    if family == "std_normal":
        return T.StdNormal(d)
    if family == "diag_gaussian":
        return T.DiagGaussian(params)
    if family == "ar1":
        t = T.AR1(d)
        np.testing.assert_array_equal(t.params, params)
        return t
    if family == "funnel":
        return T.Funnel(d)
    if family == "normal1d":
        return T.Normal1D(*params)
    raise ValueError(family)


def kwargs_from(g, prefix=""):
    kw = {}
    for n, v in zip(g[prefix + "kw_names"], g[prefix + "kw_vals"]):
        kw[str(n)] = int(v) if str(n) in ("max_treedepth", "early_max_treedepth", "max_steps") else float(v)
    return kw


def first_mismatch(got, want):
    """Index of the first iteration where any integer stat differs (None if all equal)."""
    bad = None
    for name in got:
        if name in INT_STATS and name in want:
            idx = np.nonzero(np.ravel(got[name]) != np.ravel(want[name]))[0]
            if len(idx) and (bad is None or idx[0] < bad):
                bad = int(idx[0])
    return bad


def assert_chain_matches(got_q, got_stats, want_q, want_stats, margins, label=""):
    """got/want: per-iteration arrays of ONE chain. margins[i] = oracle's smallest logbern margin at i.
    Returns the number of iterations verified bit-exactly."""
    n = len(want_q)
    bad = first_mismatch(got_stats, want_stats)
    upto = n if bad is None else bad
    if bad is not None:
        fragile = np.nonzero(margins[: bad + 1] < FRAGILE)[0]
        assert len(fragile), "%s: integer stats diverge at iteration %d but no decision margin < %g before it" % (
            label, bad, FRAGILE)
    for name in got_stats:
        g, w = np.ravel(got_stats[name])[:upto], np.ravel(want_stats[name])[:upto]
        if name in INT_STATS:
            np.testing.assert_array_equal(g, w, err_msg="%s %s" % (label, name))
        else:
            # energies carry the float32 start-energy rounding: a few float32 ulps of the kinetic energy
            scale = 1.0
            if "energy" in want_stats:
                scale = 1.0 + np.abs(np.ravel(want_stats["energy"])[:upto])
            assert np.all(np.abs(g - w) <= 1e-6 * np.abs(w) + 2e-6 * scale), "%s %s: max abs diff %g" % (
                label, name, np.max(np.abs(g - w)))
    np.testing.assert_allclose(got_q[:upto], want_q[:upto], rtol=RTOL_Q, atol=1e-9, err_msg="%s q" % label)
    return upto
