"""Hamiltonian Monte Carlo step method -- host mirror of /root/reference/littlemcmc/hmc.py.

Constructor, defaults and ``stats_dtypes`` follow hmc.py:36-138; the transition (hmc.py:140-182) is
``lmc::hmc_transition`` in csrc/lmc_sampler.hpp."""
import numpy as np

from . import _abi
from .base_hmc import BaseHMC

__all__ = ["HamiltonianMC"]


class HamiltonianMC(BaseHMC):
    name = "hmc"
    _kind = "hmc"
    default_blocked = True
    generates_stats = True
    stats_dtypes = [
        {
            "step_size": np.float64,
            "n_steps": np.int64,
            "tune": np.bool_,
            "step_size_bar": np.float64,
            "accept": np.float64,
            "diverging": np.bool_,
            "energy_error": np.float64,
            "energy": np.float64,
            "path_length": np.float64,
            "accepted": np.bool_,
            "model_logp": np.float64,
        }
    ]

    def __init__(self, logp_dlogp_func, model_ndim=None, scaling=None, is_cov=False, potential=None,
                 target_accept=0.8, Emax=1000, adapt_step_size=True, step_scale=0.25, gamma=0.05, k=0.75,
                 t0=10, step_rand=None, path_length=2.0, max_steps=1024, size=None, momentum_rng="numpy"):
        if model_ndim is None:
            model_ndim = size if size is not None else getattr(logp_dlogp_func, "d", None)
        super().__init__(logp_dlogp_func=logp_dlogp_func, model_ndim=model_ndim, scaling=scaling, is_cov=is_cov,
                         potential=potential, target_accept=target_accept, Emax=Emax,
                         adapt_step_size=adapt_step_size, step_scale=step_scale, gamma=gamma, k=k, t0=t0,
                         step_rand=step_rand)
        self._momentum_rng = momentum_rng   # "numpy": the reference's stream; "philox": counter-based throughput mode (include/lmc_hip.h)
        self.path_length = path_length
        self.max_steps = max_steps

    def _engine_kwargs(self):
        kw = super()._engine_kwargs()
        kw.update(path_length=self.path_length, max_steps=self.max_steps)
        return kw

    def _result_planes(self):
        """(name, LMC_PLANE_*, index, LMC_AS_*, dtype) per entry of stats_dtypes: what the device writes into sample()'s arrays."""
        f = lambda name, slot: (name, _abi.PLANE_F64, slot, _abi.AS_NATIVE, np.float64)   # noqa: E731
        b = lambda name, bit: (name, _abi.PLANE_U8, bit, _abi.AS_NATIVE, np.bool_)       # noqa: E731
        return [
            f("step_size", _abi.STAT_STEP_SIZE),
            ("n_steps", _abi.PLANE_I32, _abi.STAT_DEPTH, _abi.AS_I64, np.int64),
            b("tune", _abi.STAT_TUNE),
            f("step_size_bar", _abi.STAT_STEP_SIZE_BAR),
            f("accept", _abi.STAT_ACCEPT),
            b("diverging", _abi.STAT_DIVERGING),
            f("energy_error", _abi.STAT_ENERGY_ERROR),
            f("energy", _abi.STAT_ENERGY),
            f("path_length", _abi.STAT_MAX_ENERGY_ERROR),
            b("accepted", _abi.STAT_ACCEPTED),
            f("model_logp", _abi.STAT_MODEL_LOGP),
        ]

    def _stats_from_engine(self, eng, iter_begin, n):
        f = lambda s: eng.stat_f64(s, iter_begin, n)   # noqa: E731
        return {
            "step_size": f(_abi.STAT_STEP_SIZE),
            "n_steps": eng.stat_i32(_abi.STAT_DEPTH, iter_begin, n).astype(np.int64),
            "tune": eng.stat_u8(_abi.STAT_TUNE, iter_begin, n).astype(np.bool_),
            "step_size_bar": f(_abi.STAT_STEP_SIZE_BAR),
            "accept": f(_abi.STAT_ACCEPT),
            "diverging": eng.stat_u8(_abi.STAT_DIVERGING, iter_begin, n).astype(np.bool_),
            "energy_error": f(_abi.STAT_ENERGY_ERROR),
            "energy": f(_abi.STAT_ENERGY),
            "path_length": f(_abi.STAT_MAX_ENERGY_ERROR),
            "accepted": eng.stat_u8(_abi.STAT_ACCEPTED, iter_begin, n).astype(np.bool_),
            "model_logp": f(_abi.STAT_MODEL_LOGP),
        }
