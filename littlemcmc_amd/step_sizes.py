"""Dual-averaging step-size adaptation -- host view of the per-chain device state
(/root/reference/littlemcmc/step_sizes.py). ``update`` runs inside the sampling kernel
(csrc/lmc_sampler.hpp, step_sizes.py:71-92); this object carries the constructor arguments, mirrors
the fields of the last chain after a run and produces the acceptance warning on the host."""
import numpy as np
from scipy import stats

from .report import SamplerWarning, WarningType


class DualAverageAdaptation:
    def __init__(self, initial_step, target, gamma, k, t0):
        self._initial_step = initial_step
        self._target = target
        self._k = k
        self._t0 = t0
        self._gamma = gamma
        self.reset()

    def reset(self):  # step_sizes.py:49-56
        self._log_step = np.log(self._initial_step)
        self._log_bar = self._log_step
        self._hbar = 0.0
        self._count = 1
        self._mu = np.log(10 * self._initial_step)
        self._tuned_stats = []

    def _pull(self, engine, chain=0):
        st = engine.adapt_state()
        self._log_step = float(st["log_step"][chain])
        self._log_bar = float(st["log_bar"][chain])
        self._hbar = float(st["hbar"][chain])
        self._count = int(st["count"][chain])

    def current(self, tune):  # step_sizes.py:58-69
        return np.exp(self._log_step) if tune else np.exp(self._log_bar)

    def stats(self):  # step_sizes.py:94-99
        return {"step_size": np.exp(self._log_step), "step_size_bar": np.exp(self._log_bar)}

    def warnings(self):  # step_sizes.py:101-121
        accept = np.array(self._tuned_stats)
        if accept.size == 0:
            return []
        mean_accept = np.mean(accept)
        n_bound = min(100, len(accept))
        n_good, n_bad = mean_accept * n_bound, (1 - mean_accept) * n_bound
        lower, upper = stats.beta(n_good + 1, n_bad + 1).interval(0.95)
        if self._target < lower or self._target > upper:
            msg = ("The acceptance probability does not match the target. It is %s, but should be close to %s. "
                   "Try to increase the number of tuning steps." % (mean_accept, self._target))
            info = {"target": self._target, "actual": mean_accept}
            return [SamplerWarning(WarningType.BAD_ACCEPTANCE, msg, "warn", None, None, info)]
        return []
