"""Dual-averaging step-size adaptation -- host view of the per-chain device state.

The update rule of /root/reference/littlemcmc/step_sizes.py:71-92 runs inside the sampling kernel
(csrc/lmc_sampler.hpp, run_kernel epilogue). This object only (a) carries the constructor arguments to the
engine, (b) mirrors the fields of one chain after a run so that ``step.step_adapt._count`` etc. read as in the
reference, and (c) produces the acceptance-rate warning (step_sizes.py:101-121) on the host."""
import math

from scipy import stats as _scipy_stats

from .report import SamplerWarning, WarningType


class DualAverageAdaptation:
    def __init__(self, initial_step, target, gamma, k, t0):
        self._initial_step, self._target = initial_step, target
        self._gamma, self._k, self._t0 = gamma, k, t0
        self.reset()

    # -- state ------------------------------------------------------------------------------------------------
    def reset(self):
        """Start of a chain (step_sizes.py:49-56): log_step = log_bar = log(eps0), hbar = 0, count = 1."""
        self._log_step = self._log_bar = math.log(self._initial_step)
        self._mu = math.log(10 * self._initial_step)
        self._hbar, self._count = 0.0, 1
        self._tuned_stats = []

    def _pull(self, engine, chain=0):
        """Copy chain ``chain``'s adaptation scalars from the device."""
        st = engine.adapt_state()
        self._log_step, self._log_bar = float(st["log_step"][chain]), float(st["log_bar"][chain])
        self._hbar, self._count = float(st["hbar"][chain]), int(st["count"][chain])

    # -- views ------------------------------------------------------------------------------------------------
    def current(self, tune):
        """Step size the next iteration will use: exp(log_step) while adapting, else exp(log_bar)."""
        return math.exp(self._log_step if tune else self._log_bar)

    def stats(self):
        return dict(step_size=math.exp(self._log_step), step_size_bar=math.exp(self._log_bar))

    def warnings(self):
        """BAD_ACCEPTANCE if the target acceptance lies outside the 95 % Beta interval implied by the mean
        post-tuning acceptance over (at most) 100 pseudo-draws (the reference's heuristic)."""
        n = len(self._tuned_stats)
        if n == 0:
            return []
        mean_accept = sum(float(a) for a in self._tuned_stats) / n
        n_bound = min(100, n)
        lo, hi = _scipy_stats.beta(mean_accept * n_bound + 1, (1 - mean_accept) * n_bound + 1).interval(0.95)
        if lo <= self._target <= hi:
            return []
        msg = ("The acceptance probability does not match the target. It is %s, but should be close to %s. "
               "Try to increase the number of tuning steps." % (mean_accept, self._target))
        return [SamplerWarning(WarningType.BAD_ACCEPTANCE, msg, "warn", None, None,
                               {"target": self._target, "actual": mean_accept})]
