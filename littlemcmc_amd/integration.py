"""Leapfrog integrator protocol -- mirror of /root/reference/littlemcmc/integration.py.

``compute_state`` (integration.py:52-66) and ``step`` (:68-121) keep their signatures and return the
same ``State`` record, but the arithmetic is the HIP leapfrog (csrc/lmc_sampler.hpp: leapfrog<>)
run for one chain through ``lmc_engine_trajectory``. The reference name ``CpuLeapfrogIntegrator``
is kept as an alias so that ``step.integrator`` users need no change.
"""
from collections import namedtuple

import numpy as np

State = namedtuple("State", "q, p, v, q_grad, energy, model_logp")


class IntegrationError(RuntimeError):
    """Numerical errors during leapfrog integration (integration.py:28). With diagonal potentials the
    reference never raises it either: non-finite energies surface as divergences."""


class HipLeapfrogIntegrator:
    def __init__(self, potential, logp_dlogp_func, engine=None):
        self._potential = potential
        self._logp_dlogp_func = logp_dlogp_func
        self._engine = engine

    def _eng(self):
        if self._engine is None:
            from .engine import Engine

            kind = self._potential._engine_kind
            self._engine = Engine(self._logp_dlogp_func, chains=1, potential=kind,
                                  mass_dtype=getattr(self._potential, "dtype", "float32") if kind in ("diag_adapt", "diag", "full_adapt") else "float32")
            self._potential._bind(self._engine)
        return self._engine

    def _wrap(self, out, k, p_dtype):
        logp = self._logp_dlogp_func._wrap_logp(out["logp"][0, k])
        energy = out["energy"][0, k]
        if np.ndim(logp):
            energy = np.array([energy])
        p = out["p"][0, k]
        v = out["v"][0, k]
        if k == 0 and p_dtype == np.float32:  # the start state keeps float32 p and v
            p = p.astype(np.float32)
            v = v.astype(np.float32)
        return State(out["q"][0, k], p, v, out["g"][0, k], energy, logp)

    def compute_state(self, q, p):
        p = np.asarray(p)
        out = self._eng().trajectory(np.asarray(q, dtype="d"), p, 0.0, 0, 0, p0_is_f32=(p.dtype == np.float32))
        return self._wrap(out, 0, p.dtype)

    def step(self, epsilon, state, out=None):
        p = np.asarray(state.p)
        res = self._eng().trajectory(np.asarray(state.q, dtype="d"), p, float(epsilon), 1, 0,
                                     p0_is_f32=(p.dtype == np.float32))
        return self._wrap(res, 1, p.dtype)


CpuLeapfrogIntegrator = HipLeapfrogIntegrator
