"""Step-method base -- host mirror of /root/reference/littlemcmc/base_hmc.py.

``BaseHMC`` keeps the reference's constructor (base_hmc.py:32-126), attributes and lifecycle
(``tune``, ``iter_count``, ``stop_tuning``, ``reset_tuning``, ``reset``, ``warnings``, ``_astep``),
but an iteration is not Python: ``_astep`` enqueues one iteration of the HIP kernel
(csrc/lmc_sampler.hpp: run_kernel) on a one-chain engine, and ``sample()`` runs all chains and
all iterations of a run inside that kernel. The global legacy numpy RNG state is handed to the
device and back around every ``_astep`` so that ``np.random.seed`` semantics are preserved.
"""
from collections import namedtuple

import numpy as np

from . import _abi
from .integration import HipLeapfrogIntegrator
from .quadpotential import QuadPotential, QuadPotentialDiagAdapt, quad_potential
from .report import SamplerWarning, WarningType
from .step_sizes import DualAverageAdaptation
from .targets import require_device_target

HMCStepData = namedtuple("HMCStepData", "end, accept_stat, divergence_info, stats")
DivergenceInfo = namedtuple("DivergenceInfo", "message, exec_info, state")


def raise_for_status(status, what="chain"):
    """Translate per-chain device status bits into the reference's exceptions."""
    bad = np.nonzero(status & _abi.STATUS_BAD_INITIAL_ENERGY)[0]
    if len(bad):  # base_hmc.py:145-148
        raise ValueError("Bad initial energy (non-finite) in %s %s. The model might be misspecified."
                         % (what, bad[:8].tolist()))
    # math.py:23-24 (FloatingPointError for a NaN log_p) has no device counterpart: every leaf that reaches a merge
    # passed |dE| < Emax (nuts.py:358), so all log-weights are finite and log_p cannot be NaN (include/lmc_hip.h)


class _StepIntegrator(HipLeapfrogIntegrator):
    """``step.integrator``: shares the step's one-chain engine."""

    def __init__(self, step):
        super().__init__(step.potential, step._logp_dlogp_func)
        self._step = step

    def _eng(self):
        return self._step._engine()


class StepRandUniform:
    """``step_rand`` (base_hmc.py:46,123,154-155) in the form the device can honour with same-seed parity:
    ``StepRandUniform(lo, hi)`` is ``lambda s: s * np.random.uniform(lo, hi)`` -- one double of the chain's own legacy stream
    per iteration, drawn where the reference calls the function (after the momentum draw and the start state, before the
    trajectory). Calling the object does exactly that on the host, so it can be handed to the reference too.

    Any OTHER callable is honoured as well, on the host: it is evaluated once per chain per iteration BEFORE that iteration
    is launched (one launch per iteration, lmc_engine_set_step_sizes), with the step size the reference would pass it. A
    deterministic function (``lambda s: 0.9 * s``, a schedule, ...) gives the reference's chain; one that draws from
    ``np.random`` sees the host's global stream, not the chain's, and at another point of the iteration -- statistically
    the same sampler, not the same draws (use StepRandUniform for those)."""

    def __init__(self, lo, hi):
        self.lo, self.hi = float(lo), float(hi)
        if not (np.isfinite(self.lo) and np.isfinite(self.hi)):
            raise ValueError("step_rand bounds must be finite")

    def __call__(self, step_size):
        return step_size * np.random.uniform(self.lo, self.hi)

    def __repr__(self):
        return "StepRandUniform(%r, %r)" % (self.lo, self.hi)


class BaseHMC:
    """Superclass of the Hamiltonian samplers (base_hmc.py:29)."""

    _kind = None            # "nuts" | "hmc"
    stats_dtypes = None

    def __init__(self, logp_dlogp_func, model_ndim, scaling, is_cov, potential, target_accept, Emax,
                 adapt_step_size, step_scale, gamma, k, t0, step_rand):
        self._logp_dlogp_func = require_device_target(logp_dlogp_func, model_ndim)
        if step_rand is not None and not callable(step_rand):
            raise TypeError("step_rand must be callable (base_hmc.py:154-155 calls it with the step size)")
        self.adapt_step_size = adapt_step_size
        self.Emax = Emax
        self.iter_count = 0
        self.model_ndim = int(model_ndim)
        self.step_size = step_scale / (model_ndim ** 0.25)   # base_hmc.py:102
        self.target_accept = target_accept
        self._step_scale, self._gamma, self._k, self._t0 = step_scale, gamma, k, t0
        self.step_adapt = DualAverageAdaptation(self.step_size, target_accept, gamma, k, t0)
        self.tune = True
        if scaling is None and potential is None:   # base_hmc.py:109-113
            potential = QuadPotentialDiagAdapt(model_ndim, np.zeros(model_ndim), np.ones(model_ndim), 10)
        if scaling is not None and potential is not None:
            raise ValueError("Cannot specify both `potential` and `scaling`.")
        elif potential is not None:
            if not isinstance(potential, QuadPotential):
                raise TypeError("potential must be a littlemcmc_amd.quadpotential.QuadPotential")
            self.potential = potential
        else:
            self.potential = quad_potential(np.asarray(scaling), is_cov)
        self._eng1 = None
        self.integrator = _StepIntegrator(self)
        self._step_rand = step_rand
        self._warnings = []
        self._samples_after_tune = 0
        self._num_divs_sample = 0

    # -- engines ------------------------------------------------------------------------------------
    def _engine_kwargs(self):
        return dict(
            kind=self._kind, potential=self.potential._engine_kind, target_accept=self.target_accept,
            Emax=self.Emax, adapt_step_size=self.adapt_step_size, step_scale=self._step_scale,
            gamma=self._gamma, k=self._k, t0=self._t0,
            adaptation_window=getattr(self.potential, "_initial_adaptation_window", 101),
            adaptation_window_multiplier=getattr(self.potential, "adaptation_window_multiplier", 1.0),
            rng=getattr(self, "_momentum_rng", "numpy"),
            mass_dtype=getattr(self.potential, "dtype", "float32") if self.potential._engine_kind in ("diag_adapt", "diag", "full_adapt") else "float32",
        )

    def _make_engine(self, chains, device=0):
        """A fresh engine for ``chains`` chains configured like this step (used by sample())."""
        from .engine import Engine

        eng = Engine(self._logp_dlogp_func, chains=chains, device=device, **self._engine_kwargs())
        self.potential._push_initial(eng)
        if isinstance(self._step_rand, StepRandUniform):
            eng.set_step_jitter(self._step_rand.lo, self._step_rand.hi)
        return eng

    def _host_step_rand(self):
        """The step_rand callable the HOST has to evaluate per iteration (None: none, or the device's own uniform form)."""
        return None if isinstance(self._step_rand, StepRandUniform) else self._step_rand

    def _host_step_sizes(self, eng, tune):
        """step_rand(step size of the coming iteration) for every chain of ``eng`` (base_hmc.py:151-155)."""
        st = eng.adapt_state()
        base = np.exp(st["log_step"] if (tune and self.adapt_step_size) else st["log_bar"])
        return np.array([float(self._step_rand(float(b))) for b in base])

    def _engine(self):
        if self._eng1 is None:
            self._eng1 = self._make_engine(1)
            self.potential._bind(self._eng1)
            self._eng1.reserve(1, keep_trace=False)
        return self._eng1

    # -- lifecycle ------------------------------------------------------------------------------------
    def stop_tuning(self):   # base_hmc.py:128-131
        if hasattr(self, "tune"):
            self.tune = False

    def reset_tuning(self, start=None):   # base_hmc.py:192-195
        self.step_adapt.reset()
        if self._eng1 is not None:
            self._eng1.reset_tuning()
        self.reset(start=None)

    def reset(self, start=None):   # base_hmc.py:197-200
        self.tune = True
        self.potential.reset()

    # -- one iteration ----------------------------------------------------------------------------------
    def _stats_from_engine(self, eng, iter_begin, n):
        raise NotImplementedError

    def _astep(self, q0):
        """One HMC/NUTS iteration on the device (base_hmc.py:140-190)."""
        eng = self._engine()
        eng.set_position(np.asarray(q0, dtype="d").reshape(1, self.model_ndim))
        eng.set_rng_state(0, np.random.get_state())
        # the step size this iteration integrates with (base_hmc.py:151-153), from the adaptation state before the update
        self.step_size = float(np.exp(self.step_adapt._log_step if (self.tune and self.adapt_step_size)
                                      else self.step_adapt._log_bar))
        if self._host_step_rand() is not None:   # base_hmc.py:154-155 (evaluated before the launch: see StepRandUniform)
            # the reference keeps the ADAPTED value on the object (base_hmc.py:151-153) and integrates with the jittered one
            # (a local, :154-155): only the engine sees step_rand's result
            eng.set_step_sizes([float(self._step_rand(self.step_size))])
            eng.set_rng_state(0, np.random.get_state())   # the callable may have drawn from the global stream
        eng.run(1 if self.tune else 0, 0, 1)
        np.random.set_state(eng.get_rng_state(0))
        raise_for_status(eng.status())
        q = eng.get_position()[0]
        stats = {k: v[0, 0] for k, v in self._stats_from_engine(eng, 0, 1).items()}
        if not (self.tune and self.adapt_step_size):
            self.step_adapt._tuned_stats.append(stats["mean_tree_accept" if self._kind == "nuts" else "accept"])
        self.step_adapt._pull(eng)
        self.potential._pull(eng)
        if stats["diverging"]:
            if self.tune:
                kind = WarningType.TUNING_DIVERGENCE
            else:
                kind = WarningType.DIVERGENCE
                self._num_divs_sample += 1
            self._warnings.append(SamplerWarning(kind, "Divergence encountered.", "debug", self.iter_count, None, None))
        self.iter_count += 1
        if not self.tune:
            self._samples_after_tune += 1
        return q, [stats]

    def warnings(self):   # base_hmc.py:202-230
        warnings = list(self._warnings)
        message = ""
        n_divs = self._num_divs_sample
        if n_divs and self._samples_after_tune == n_divs:
            message = "The chain contains only diverging samples. The model is probably misspecified."
        elif n_divs == 1:
            message = "There was 1 divergence after tuning. Increase `target_accept` or reparameterize."
        elif n_divs > 1:
            message = ("There were %s divergences after tuning. Increase `target_accept` or reparameterize."
                       % n_divs)
        if message:
            warnings.append(SamplerWarning(WarningType.DIVERGENCES, message, "error", None, None, None))
        warnings.extend(self.step_adapt.warnings())
        return warnings
