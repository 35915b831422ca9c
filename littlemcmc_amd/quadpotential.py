"""Mass matrices (potentials) -- host-side mirror of /root/reference/littlemcmc/quadpotential.py:
the diagonal family of the hot path and the dense family (section 8f-3).

The objects keep the reference's constructor signatures and protocol (``velocity``, ``energy``,
``velocity_energy``, ``random``, ``update``, ``reset``; quadpotential.py:93-140) but hold no
numerics of their own: every method is a call into liblmc_hip.so (a one-chain engine). ``random()``
consumes the *global* legacy numpy stream exactly like the reference does -- the MT19937 state
is handed to the device and back -- so ``np.random.seed(...)`` keeps its meaning.

Dense potentials (QuadPotentialFull / FullInv / FullAdapt, quadpotential.py:390-615) run on the device: matrix
sweeps inside the leapfrog, triangular-solve momentum draws, and for FullAdapt a batched covariance refresh + Cholesky
kernel after every tuning iteration (littlemcmc_amd/csrc/lmc_dense.hpp) -- fused kernels up to model_ndim = 256, the
general kernels beyond (Full / FullInv up to 2048, FullAdapt up to 1024; littlemcmc_amd/csrc/lmc_wide.hpp).
Sparse scalings (QuadPotentialSparse needs scikit-sparse in the reference) are not implemented.
"""
import numpy as np

from . import targets as _targets

__all__ = [
    "quad_potential",
    "QuadPotentialDiag",
    "QuadPotentialFull",
    "QuadPotentialFullInv",
    "QuadPotentialDiagAdapt",
    "QuadPotentialFullAdapt",
    "PositiveDefiniteError",
]


class PositiveDefiniteError(ValueError):
    """quadpotential.py:80-90."""

    def __init__(self, msg, idx):
        super().__init__(msg)
        self.idx = idx
        self.msg = msg

    def __str__(self):
        return "Scaling is not positive definite: %s. Check indexes %s." % (self.msg, self.idx)


def partial_check_positive_definite(C):
    """quadpotential.py:68-77."""
    d = C if C.ndim == 1 else np.diag(C)
    (i,) = np.nonzero(np.logical_or(np.isnan(d), d <= 0))
    if len(i):
        raise PositiveDefiniteError("Simple check failed. Diagonal contains negatives", i)


def quad_potential(C, is_cov):
    """quadpotential.py:33-65: build a potential from a scaling vector (diagonal) or matrix."""
    if hasattr(C, "toarray") and hasattr(C, "nnz"):   # scipy.sparse (quadpotential.py:49-53; the reference's own sparse
        if not is_cov:                                #  class does not exist -- a sparse covariance is densified here)
            raise ValueError("Sparse precision matrices are not supported")
        C = C.toarray()
    C = np.asarray(C)
    partial_check_positive_definite(C)
    if C.ndim == 1:
        return QuadPotentialDiag(C if is_cov else 1.0 / C)
    if is_cov:
        return QuadPotentialFull(C)
    return QuadPotentialFullInv(C)


def isquadpotential(value):
    """quadpotential.py:143-145."""
    return isinstance(value, QuadPotential)


class QuadPotential:
    """Protocol base (quadpotential.py:93-140). Numerics live on the device."""

    _engine_kind = None  # "diag_adapt" | "diag" | "full" | "full_inv" | "full_adapt"
    _momentum_f32 = False

    def __init__(self, n):
        self._n = int(n)
        self._engine = None      # engine this potential is bound to (a step's, or a private one)
        self._own_engine = False

    # -- engine plumbing ------------------------------------------------------------------------
    def _bind(self, engine):
        """Attach to the one-chain engine of a step method and load the initial values into it."""
        self._engine = engine
        self._own_engine = False
        self._push_initial(engine)

    def _eng(self):
        if self._engine is None:
            from .engine import Engine

            self._engine = Engine(_targets.StdNormal(self._n), chains=1, potential=self._engine_kind,
                                  mass_dtype=getattr(self, "dtype", "float32") if self._engine_kind in ("diag_adapt", "diag", "full_adapt") else "float32",
                                  adaptation_window=getattr(self, "_initial_adaptation_window", 101),
                                  adaptation_window_multiplier=getattr(self, "adaptation_window_multiplier", 1.0))
            self._own_engine = True
            self._push_initial(self._engine)
        return self._engine

    def _push_initial(self, engine):
        raise NotImplementedError

    def _pull(self, engine, chain=0):
        """Refresh the host-visible attributes from the device state of ``chain``."""
        st = engine.adapt_state()
        self._var = st["var"][chain].copy()
        self._stds = np.sqrt(self._var)
        self._inv_stds = (1.0 / self._stds).astype(np.float32)
        self._n_samples = int(st["n_samples"][chain])

    def _state0(self, x):
        """compute_state(q=0, p=x) under a standard-normal target: v = M^-1 x, energy = x.v/2."""
        eng = self._eng()
        if eng.target.family != _targets.StdNormal.family:
            # bound to a step with another target: kinetic part = energy + logp(0)
            out = eng.trajectory(np.zeros(self._n), x, 0.0, 0, 0)
            return out["v"][0, 0], out["energy"][0, 0] + out["logp"][0, 0]
        out = eng.trajectory(np.zeros(self._n), x, 0.0, 0, 0)
        return out["v"][0, 0], out["energy"][0, 0]

    # -- protocol -------------------------------------------------------------------------------
    def velocity(self, x, out=None):
        v, _ = self._state0(np.asarray(x))
        if out is not None:
            out[:] = v
            return out
        return v

    def energy(self, x, velocity=None):
        _, e = self._state0(np.asarray(x))
        return e

    def velocity_energy(self, x, v_out):
        v, e = self._state0(np.asarray(x))
        v_out[:] = v
        return e

    def random(self):
        """Momentum draw from the global legacy numpy stream, generated on the device."""
        eng = self._eng()
        eng.set_rng_state(0, np.random.get_state())
        p = eng.draw_momentum()[0]
        np.random.set_state(eng.get_rng_state(0))
        return p.astype(np.float32) if self._momentum_f32 else p

    def update(self, sample, grad, tune):
        """quadpotential.py:112-118: the fixed potentials learn nothing from a sample (the base class's `pass`)."""
        return None

    def raise_ok(self, vmap=None):
        return None

    def reset(self):
        if self._engine is not None:
            self._push_initial(self._engine)


class QuadPotentialDiagAdapt(QuadPotential):
    """quadpotential.py:148-291: float32 diagonal adapted from the tuning draws' variance."""

    _engine_kind = "diag_adapt"
    _momentum_f32 = True

    def __init__(self, n, initial_mean, initial_diag=None, initial_weight=0, adaptation_window=101,
                 adaptation_window_multiplier=1, dtype=None):
        initial_mean = np.asarray(initial_mean)
        if initial_diag is not None:
            initial_diag = np.asarray(initial_diag)
            if initial_diag.ndim != 1:
                raise ValueError("Initial diagonal must be one-dimensional.")
        if initial_mean.ndim != 1:
            raise ValueError("Initial mean must be one-dimensional.")
        if initial_diag is not None and len(initial_diag) != n:
            raise ValueError("Wrong shape for initial_diag: expected %s got %s" % (n, len(initial_diag)))
        if len(initial_mean) != n:
            raise ValueError("Wrong shape for initial_mean: expected %s got %s" % (n, len(initial_mean)))
        if dtype is None:   # quadpotential.py:175-176
            dtype = "float32"
        dtype = np.dtype(dtype).name
        if dtype not in ("float32", "float64"):
            raise NotImplementedError("the device mass matrix is float32 (the reference's default) or float64")
        super().__init__(n)
        self.dtype = dtype
        self._momentum_f32 = dtype == "float32"   # quadpotential.py:223: normal(size=n).astype(dtype)
        if initial_diag is None:  # quadpotential.py:178-180
            initial_diag = np.ones(n, dtype=dtype)
            initial_weight = 1
        self._initial_mean = np.array(initial_mean, dtype="d")
        self._initial_diag = initial_diag.astype(dtype)
        self._initial_weight = initial_weight
        self.adaptation_window = int(adaptation_window)
        self._initial_adaptation_window = int(adaptation_window)
        self.adaptation_window_multiplier = float(adaptation_window_multiplier)
        self._var = np.array(self._initial_diag, copy=True)
        self._stds = np.sqrt(self._initial_diag)
        self._inv_stds = 1.0 / self._stds
        self._n_samples = 0

    def _pull(self, engine, chain=0):
        super()._pull(engine, chain)
        if self.dtype == "float64":   # quadpotential.py:226-229 in the potential's dtype
            self._stds = np.sqrt(self._var)
            self._inv_stds = 1.0 / self._stds
        if self.adaptation_window_multiplier != 1.0:
            self.adaptation_window = int(engine.get_chain_state(fields=("window",))["window"][chain])

    def update(self, sample, grad, tune):
        """quadpotential.py:231-245 as one device call: both Welford estimators take the sample, the foreground one becomes
        the float32 variance, the window switches (the kernel function lmc_engine_run applies after every tuning iteration).
        During sample() this runs inside the sampling kernel; the host call is the reference's protocol method."""
        if not tune:
            return
        eng = self._eng()
        eng.set_position(np.asarray(sample, dtype="d").reshape(1, self._n))
        eng.diag_update(True)
        self._pull(eng)

    def _push_initial(self, engine):
        engine.set_potential(self._initial_mean, self._initial_diag.astype("d"), float(self._initial_weight))
        self.adaptation_window = self._initial_adaptation_window
        self._var = np.array(self._initial_diag, copy=True)
        self._stds = np.sqrt(self._initial_diag)
        self._inv_stds = 1.0 / self._stds
        self._n_samples = 0


class QuadPotentialDiag(QuadPotential):
    """quadpotential.py:346-387: fixed float32 diagonal (covariance), float64 momentum draw."""

    _engine_kind = "diag"

    def __init__(self, v, dtype=None):
        v = np.asarray(v)
        dtype = np.dtype("float32" if dtype is None else dtype).name   # quadpotential.py:358-359
        if dtype not in ("float32", "float64"):
            raise NotImplementedError("the device mass matrix is float32 (the reference's default) or float64")
        super().__init__(v.shape[0])
        self.dtype = dtype
        self.v = v.astype(dtype)
        self.s = self.v ** 0.5
        self.inv_s = 1.0 / self.s
        self._n_samples = 0

    def _push_initial(self, engine):
        engine.set_potential(None, self.v.astype("d"), 0.0)

    def _pull(self, engine, chain=0):
        pass


MAX_DENSE_NDIM = 2048        # QuadPotentialFull / FullInv: fused kernels up to 256, the general kernels beyond (include/lmc_hip.h)
MAX_DENSE_ADAPT_NDIM = 1024  # QuadPotentialFullAdapt: one matrix per chain, refreshed + factorised every tuning iteration (fused kernels
                             # and a register-resident factorisation up to 256, the general kernels and one through HBM beyond)


def _square(a, what):
    a = np.asarray(a)
    if a.ndim != 2 or a.shape[0] != a.shape[1]:
        raise ValueError("%s must be a square two-dimensional array" % what)
    if a.shape[0] > MAX_DENSE_NDIM:
        raise NotImplementedError("dense mass matrices run on the device up to model_ndim = %d" % MAX_DENSE_NDIM)
    return a


class _DensePotential(QuadPotential):
    """Shared plumbing of the dense potentials: the matrix is factorised by liblmc_hip when it is pushed to an
    engine; a matrix that is not positive definite surfaces as numpy.linalg.LinAlgError, the exception
    scipy.linalg.cholesky raises in the reference's constructors."""

    def _matrix_args(self):
        raise NotImplementedError

    def _push_initial(self, engine):
        from ._abi import HipLibraryError

        try:
            engine.set_dense_potential(*self._matrix_args())
        except HipLibraryError as err:
            if "positive definite" in str(err) or "infs or NaNs" in str(err):
                raise np.linalg.LinAlgError(str(err)) from None
            raise
        self._pull(engine)

    def _pull(self, engine, chain=0):
        self._cov, self._chol = engine.dense_chain(chain)

    def _validate(self):
        """Factorise now (one-chain engine), like the reference's constructors do."""
        self._eng()


class QuadPotentialFull(_DensePotential):
    """quadpotential.py:428-468: covariance ``cov`` in ``dtype`` (float32 by default, like the reference; "float64" keeps
    the matrix, the velocity cov @ x and the momentum solve(chol.T, z) in float64)."""

    _engine_kind = "full"
    _momentum_f32 = True

    def __init__(self, cov, dtype=None):
        if dtype in (None, "float32", np.float32):
            self.dtype = "float32"
        elif dtype in ("float64", np.float64, "d", float):
            self.dtype = "float64"
            self._engine_kind = "full_f64"      # include/lmc_hip.h: LMC_POT_FULL_F64
            self._momentum_f32 = False
        else:
            raise NotImplementedError("the device mass matrix is float32 (the reference's default) or float64")
        cov = _square(cov, "cov")
        super().__init__(cov.shape[0])
        self._matrix = np.array(cov, dtype="d")
        self._cov = np.array(cov, dtype=self.dtype, copy=True)
        self._chol = None
        self._n_samples = 0

    def _matrix_args(self):
        return (self._matrix,)

    def _pull(self, engine, chain=0):
        cov, chol = engine.dense_chain(chain)          # float32 views of the device's matrices
        if self.dtype == "float32":
            self._cov, self._chol = cov, chol
        else:                                          # the float64 matrix is the caller's own; the factor as the device holds it
            self._chol = engine.dense_factor_f64()

    __call__ = QuadPotential.random


class QuadPotentialFullInv(_DensePotential):
    """quadpotential.py:388-425: mass matrix ``A`` (inverse covariance); velocity = A^-1 x, momentum = L z."""

    _engine_kind = "full_inv"
    _momentum_f32 = False

    def __init__(self, A, dtype=None):
        A = _square(A, "A")
        super().__init__(A.shape[0])
        self.dtype = "float32" if dtype is None else dtype
        self._matrix = np.array(A, dtype="d")
        self._n_samples = 0

    def _matrix_args(self):
        return (self._matrix,)

    def _pull(self, engine, chain=0):
        self._cov, _chol32 = engine.dense_chain(chain)
        self.L = engine.dense_factor_f64()


class QuadPotentialFullAdapt(_DensePotential):
    """quadpotential.py:471-557: dense covariance re-estimated from the tuning draws (two running estimators,
    growing adaptation windows), refreshed together with its Cholesky factor every ``update_window`` samples.

    Every chain adapts its own matrix. ``reset()`` restores the constructor state (the reference's reset is the
    inherited no-op, so its sequential driver carries chain k-1's matrix into chain k; its multi-process driver
    -- and this engine -- start every chain fresh)."""

    _engine_kind = "full_adapt"
    _momentum_f32 = True

    def __init__(self, n, initial_mean, initial_cov=None, initial_weight=0, adaptation_window=101,
                 adaptation_window_multiplier=2, update_window=1, dtype=None):
        initial_mean = np.asarray(initial_mean)
        if initial_cov is not None:
            initial_cov = np.asarray(initial_cov)
            if initial_cov.ndim != 2:
                raise ValueError("Initial covariance must be two-dimensional.")
        if initial_mean.ndim != 1:
            raise ValueError("Initial mean must be one-dimensional.")
        if initial_cov is not None and initial_cov.shape != (n, n):
            raise ValueError("Wrong shape for initial_cov: expected %s got %s" % (n, initial_cov.shape))
        if len(initial_mean) != n:
            raise ValueError("Wrong shape for initial_mean: expected %s got %s" % (n, len(initial_mean)))
        if dtype is None:   # quadpotential.py:497-498
            dtype = "float32"
        dtype = np.dtype(dtype).name
        if dtype not in ("float32", "float64"):
            raise NotImplementedError("QuadPotentialFullAdapt runs on the device in float32 (the reference's default) or float64")
        if n > MAX_DENSE_ADAPT_NDIM:
            raise NotImplementedError("per-chain adapted dense mass matrices run on the device up to model_ndim = %d"
                                      % MAX_DENSE_ADAPT_NDIM)
        super().__init__(n)
        self.dtype = dtype
        self._momentum_f32 = dtype == "float32"   # quadpotential.py:451: normal(size=n).astype(self.dtype)
        if initial_cov is None:  # quadpotential.py:501-503
            initial_cov = np.eye(n, dtype=dtype)
            initial_weight = 1
        self._initial_mean = np.array(initial_mean, dtype="d")
        self._matrix = np.array(initial_cov, dtype="d")
        self._initial_weight = float(initial_weight)
        self._adaptation_window = int(adaptation_window)
        self._adaptation_window_multiplier = float(adaptation_window_multiplier)
        self._update_window = int(update_window)
        self._cov = np.array(initial_cov, dtype=dtype, copy=True)
        self._chol = None
        self._chol_error = None
        self._previous_update = 0
        self._n_samples = 0
        self._initial_window = int(adaptation_window)

    def _matrix_args(self):
        return (self._matrix, self._initial_mean, self._initial_weight, self._initial_window,
                self._adaptation_window_multiplier, self._update_window)

    def _pull(self, engine, chain=0):
        self._cov, self._chol = engine.dense_chain(chain)
        st = engine.get_dense_state(fields=("window", "previous_update", "chol_failures"))
        self._adaptation_window = int(st["window"][chain])
        self._previous_update = int(st["previous_update"][chain])
        self._n_samples = int(engine.adapt_state()["n_samples"][chain])
        if int(st["chol_failures"][chain]):
            self._chol_error = np.linalg.LinAlgError("the covariance estimate was not positive definite")

    def update(self, sample, grad, tune):
        """quadpotential.py:528-552 as one device call (covariance refresh + Cholesky kernel)."""
        eng = self._eng()
        eng.set_position(np.asarray(sample, dtype="d").reshape(1, self._n))
        eng.dense_update(tune)
        self._pull(eng)

    def raise_ok(self, vmap=None):   # quadpotential.py:554-557
        if self._chol_error is not None:
            raise ValueError("{0}".format(self._chol_error))
