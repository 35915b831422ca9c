"""No-U-Turn sampler step method -- host mirror of /root/reference/littlemcmc/nuts.py.

Constructor, defaults, ``stats_dtypes`` and ``warnings`` follow nuts.py:87-239; the transition itself
(nuts.py:204-224 and the ``_Tree`` class, :251-435) is ``lmc::nuts_transition`` in
csrc/lmc_sampler.hpp."""
import numpy as np

from . import _abi
from .base_hmc import BaseHMC
from .report import SamplerWarning, WarningType

__all__ = ["NUTS"]


class NUTS(BaseHMC):
    name = "nuts"
    _kind = "nuts"
    default_blocked = True
    generates_stats = True
    stats_dtypes = [
        {
            "depth": np.int64,
            "step_size": np.float64,
            "tune": np.bool_,
            "mean_tree_accept": np.float64,
            "step_size_bar": np.float64,
            "tree_size": np.float64,
            "diverging": np.bool_,
            "energy_error": np.float64,
            "energy": np.float64,
            "max_energy_error": np.float64,
            "model_logp": np.float64,
        }
    ]

    def __init__(self, logp_dlogp_func, model_ndim=None, scaling=None, is_cov=False, potential=None,
                 target_accept=0.8, Emax=1000, adapt_step_size=True, step_scale=0.25, gamma=0.05, k=0.75,
                 t0=10, step_rand=None, path_length=2.0, max_treedepth=10, early_max_treedepth=8, size=None, momentum_rng="numpy",
                 lds_plan="auto"):
        if model_ndim is None:
            model_ndim = size if size is not None else getattr(logp_dlogp_func, "d", None)
        super().__init__(logp_dlogp_func=logp_dlogp_func, model_ndim=model_ndim, scaling=scaling, is_cov=is_cov,
                         potential=potential, target_accept=target_accept, Emax=Emax,
                         adapt_step_size=adapt_step_size, step_scale=step_scale, gamma=gamma, k=k, t0=t0,
                         step_rand=step_rand)
        self._lds_plan = lds_plan           # include/lmc_hip.h: LMC_LDS_PLAN_* ("auto" / "shallow" / "deep"); results do not depend on it
        self._momentum_rng = momentum_rng   # "numpy": the reference's stream; "philox": counter-based throughput mode (include/lmc_hip.h)
        self.max_treedepth = max_treedepth
        self.early_max_treedepth = early_max_treedepth
        self.path_length = path_length
        self._reached_max_treedepth = 0

    def _engine_kwargs(self):
        kw = super()._engine_kwargs()
        kw.update(max_treedepth=self.max_treedepth, early_max_treedepth=self.early_max_treedepth, lds_plan=self._lds_plan)
        return kw

    def _result_planes(self):
        """(name, LMC_PLANE_*, index, LMC_AS_*, dtype) per entry of stats_dtypes: what the device writes into sample()'s arrays."""
        f = lambda name, slot: (name, _abi.PLANE_F64, slot, _abi.AS_NATIVE, np.float64)   # noqa: E731
        return [
            ("depth", _abi.PLANE_I32, _abi.STAT_DEPTH, _abi.AS_I64, np.int64),
            f("step_size", _abi.STAT_STEP_SIZE),
            ("tune", _abi.PLANE_U8, _abi.STAT_TUNE, _abi.AS_NATIVE, np.bool_),
            f("mean_tree_accept", _abi.STAT_ACCEPT),
            f("step_size_bar", _abi.STAT_STEP_SIZE_BAR),
            ("tree_size", _abi.PLANE_I32, _abi.STAT_TREE_SIZE, _abi.AS_F64, np.float64),
            ("diverging", _abi.PLANE_U8, _abi.STAT_DIVERGING, _abi.AS_NATIVE, np.bool_),
            f("energy_error", _abi.STAT_ENERGY_ERROR),
            f("energy", _abi.STAT_ENERGY),
            f("max_energy_error", _abi.STAT_MAX_ENERGY_ERROR),
            f("model_logp", _abi.STAT_MODEL_LOGP),
        ]

    def _stats_from_engine(self, eng, iter_begin, n):
        f = lambda s: eng.stat_f64(s, iter_begin, n)   # noqa: E731
        return {
            "depth": eng.stat_i32(_abi.STAT_DEPTH, iter_begin, n).astype(np.int64),
            "step_size": f(_abi.STAT_STEP_SIZE),
            "tune": eng.stat_u8(_abi.STAT_TUNE, iter_begin, n).astype(np.bool_),
            "mean_tree_accept": f(_abi.STAT_ACCEPT),
            "step_size_bar": f(_abi.STAT_STEP_SIZE_BAR),
            "tree_size": eng.stat_i32(_abi.STAT_TREE_SIZE, iter_begin, n).astype(np.float64),
            "diverging": eng.stat_u8(_abi.STAT_DIVERGING, iter_begin, n).astype(np.bool_),
            "energy_error": f(_abi.STAT_ENERGY_ERROR),
            "energy": f(_abi.STAT_ENERGY),
            "max_energy_error": f(_abi.STAT_MAX_ENERGY_ERROR),
            "model_logp": f(_abi.STAT_MODEL_LOGP),
        }

    def _astep(self, q0):
        before = int(self._engine().counters()[0, _abi.CT_REACHED_MAX_TREEDEPTH])
        out = super()._astep(q0)
        self._reached_max_treedepth += int(self._eng1.counters()[0, _abi.CT_REACHED_MAX_TREEDEPTH]) - before
        return out

    def warnings(self):   # nuts.py:226-239
        warnings = super().warnings()
        n_samples = self._samples_after_tune
        n_treedepth = self._reached_max_treedepth
        if n_samples > 0 and n_treedepth / float(n_samples) > 0.05:
            msg = ("The chain reached the maximum tree depth. Increase max_treedepth, increase target_accept "
                   "or reparameterize.")
            warnings.append(SamplerWarning(WarningType.TREEDEPTH, msg, "warn", None, None, None))
        return warnings
