"""Synthetic log-densities (numpy, float64) used to drive the reference and the oracle.

TEST INFRASTRUCTURE ONLY. Nothing under ``littlemcmc_amd/`` imports this module; the
product path evaluates its targets on the GPU (``littlemcmc_amd/csrc/lmc_targets.hpp``).
These are the CPU statements of the same densities, written so that the operation order
matches the device code (see SURVEY.md Appendix C for the definitions).

The plug-in contract they satisfy is the reference's ``logp_dlogp_func(q) -> (logp, dlogp)``
(/root/reference/littlemcmc/integration.py:40,62,115): ``q`` is ``float64[d]``, ``logp`` is a
``np.float64`` scalar and ``dlogp`` is ``float64[d]``.
"""
import numpy as np


class StdNormal:
    """logp = -1/2 sum q_i^2 ; g = -q."""

    family = "std_normal"

    def __init__(self, d):
        self.d = int(d)

    def params(self):
        return np.zeros(0, dtype=np.float64)

    def __call__(self, q):
        return -0.5 * np.dot(q, q), -q


class DiagGaussian:
    """Independent Gaussian with precisions ``prec``: g = -(prec*q), logp = 1/2 q.g.

    ``ill_conditioned(d, kappa)`` builds sigma_i^2 = kappa^{i/(d-1)} (config C4).
    """

    family = "diag_gaussian"

    def __init__(self, prec):
        self.prec = np.ascontiguousarray(prec, dtype=np.float64)
        self.d = self.prec.shape[0]

    @classmethod
    def ill_conditioned(cls, d, kappa=1e4):
        i = np.arange(d, dtype=np.float64)
        sigma2 = kappa ** (i / max(d - 1, 1))
        return cls(1.0 / sigma2)

    def params(self):
        return self.prec

    def __call__(self, q):
        g = -(self.prec * q)
        return 0.5 * np.dot(q, g), g


class AR1:
    """Stationary AR(1) Gaussian, unit marginal variance, tridiagonal precision (config C3).

    c = 1/(1-rho^2); P_11 = P_dd = c; P_ii = (1+rho^2) c; P_{i,i+-1} = -rho c.
    (Pq)_i = (diag_i q_i + off q_{i-1}) + off q_{i+1};  g = -Pq;  logp = 1/2 q.g
    """

    family = "ar1"

    def __init__(self, d, rho=0.9):
        self.d = int(d)
        self.rho = float(rho)
        c = 1.0 / (1.0 - self.rho * self.rho)
        self.c_end = c
        self.c_mid = (1.0 + self.rho * self.rho) * c
        self.off = -self.rho * c

    def params(self):
        return np.array([self.c_end, self.c_mid, self.off], dtype=np.float64)

    def __call__(self, q):
        d = self.d
        diag = np.full(d, self.c_mid)
        diag[0] = self.c_end
        diag[d - 1] = self.c_end
        pq = diag * q
        if d > 1:
            pq[1:] += self.off * q[:-1]
            pq[:-1] += self.off * q[1:]
        g = -pq
        return 0.5 * np.dot(q, g), g


class Funnel:
    """Neal's funnel (config C5): v = q_0 ~ N(0, 3^2), q_i | v ~ N(0, e^v), i >= 1.

    logp = -v^2/18 - (d-1) v / 2 - 1/2 e^{-v} S,  S = sum_{i>=1} q_i^2
    g_0 = -v/9 - (d-1)/2 + 1/2 e^{-v} S ;  g_i = -e^{-v} q_i
    """

    family = "funnel"

    def __init__(self, d):
        self.d = int(d)

    def params(self):
        return np.zeros(0, dtype=np.float64)

    def __call__(self, q):
        d = self.d
        v = q[0]
        x = q[1:]
        s = np.dot(x, x)
        ev = np.exp(-v)
        hes = 0.5 * ev * s
        logp = -(v * v) / 18.0 - 0.5 * (d - 1) * v - hes
        g = np.empty(d, dtype=np.float64)
        g[0] = -v / 9.0 - 0.5 * (d - 1) + hes
        g[1:] = -(ev * x)
        return np.float64(logp), g


class Normal1D:
    """The reference's own test target (/root/reference/tests/test_utils.py:19-28), stated
    analytically: logp = -(x-loc)^2/(2 scale^2) - log(scale sqrt(2 pi)) elementwise, returned as a
    float64 array of the same shape as ``q`` (the reference's tests rely on shape-(1,) logp);
    dlogp = -(x - loc)/scale (sic: the reference divides by scale, not scale^2)."""

    family = "normal1d"

    def __init__(self, d=1, loc=0.0, scale=1.0):
        self.d = int(d)
        self.loc = float(loc)
        self.scale = float(scale)

    def params(self):
        return np.array([self.loc, self.scale], dtype=np.float64)

    def __call__(self, q):
        z = (q - self.loc) / self.scale
        logp = -0.5 * z * z - np.log(self.scale * np.sqrt(2.0 * np.pi))
        return logp, -(q - self.loc) / self.scale


def make(family, d, **kw):
    if family == "std_normal":
        return StdNormal(d)
    if family == "ar1":
        return AR1(d, kw.get("rho", 0.9))
    if family == "diag_gaussian":
        if "prec" in kw:
            return DiagGaussian(kw["prec"])
        return DiagGaussian.ill_conditioned(d, kw.get("kappa", 1e4))
    if family == "funnel":
        return Funnel(d)
    if family == "normal1d":
        return Normal1D(d, kw.get("loc", 0.0), kw.get("scale", 1.0))
    raise ValueError("unknown target family %r" % (family,))
