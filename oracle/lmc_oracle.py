"""CPU oracle: a numpy restatement of the reference's many-chain HMC/NUTS hot path.

TEST INFRASTRUCTURE ONLY. Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this file, and only as the checker / timed CPU baseline.
Nothing under ``littlemcmc_amd/`` imports it; the product path fails loudly without the HIP
library.

It follows /root/reference/littlemcmc (0.2.2) function by function, one chain at a time, with
the same dtypes (float32 mass matrix / momentum draw, float64 everywhere else), the same
legacy-MT19937 consumption order and the same quirks (in-place aliased ``p_sum``, "next"
step size in the stats, ...). The recursion of ``_Tree._build_subtree`` is restated as an
iterative post-order walk with an explicit stack -- the form the HIP kernel uses -- and is
pinned against the imported reference by tests/golden/capture.py (goldens committed under
tests/golden/) and tests/test_oracle_golden.py.

Parity status: PINNED. Checked bit-for-bit against the imported reference in the build
container (same numpy/OpenBLAS) on the fixtures in tests/golden/.
"""
from collections import namedtuple

import numpy as np
import scipy.linalg

# integration.py:25
State = namedtuple("State", "q p v g energy logp")


# --------------------------------------------------------------------------------------
# quadpotential.py
# --------------------------------------------------------------------------------------
class _Welford:
    """quadpotential.py:294-340 (_WeightedVariance), float64 accumulators."""

    def __init__(self, n, mean=None, var=None, weight=0.0):
        self.w_sum = float(weight)
        self.mean = np.zeros(n) if mean is None else np.array(mean, dtype="d", copy=True)
        self.raw_var = np.zeros(n) if var is None else np.array(var, dtype="d", copy=True)
        self.raw_var[:] *= self.w_sum

    def add(self, x):
        # quadpotential.py:324-332 with weight == 1
        self.w_sum += 1
        prop = 1 / self.w_sum
        old = x - self.mean
        self.mean[:] += prop * old
        new = x - self.mean
        self.raw_var[:] += 1 * old * new


class DiagAdaptPotential:
    """quadpotential.py:148-245 (QuadPotentialDiagAdapt); all mass arrays in ``dtype`` (float32 unless the caller says
    "float64", quadpotential.py:175-176), and with them the momentum draw (:223) and the start state's velocity / energy."""

    def __init__(self, n, initial_mean, initial_diag=None, initial_weight=0, window=101, multiplier=1, dtype="float32"):
        self.n = n
        self.dtype = np.dtype(dtype).name
        self.momentum_f32 = self.dtype == "float32"
        if initial_diag is None:  # quadpotential.py:178-180
            initial_diag = np.ones(n, dtype=self.dtype)
            initial_weight = 1
        else:
            initial_diag = np.asarray(initial_diag).astype(self.dtype)
        self._initial_mean = np.array(initial_mean, dtype="d")
        self._initial_diag = initial_diag
        self._initial_weight = initial_weight
        self._initial_window = window
        self.multiplier = float(multiplier)
        self.reset()

    def reset(self):  # quadpotential.py:195-204
        self.var = np.array(self._initial_diag, dtype=self.dtype, copy=True)
        self.stds = np.sqrt(self._initial_diag)
        self.inv_stds = 1.0 / self.stds
        self.fore = _Welford(self.n, self._initial_mean, self._initial_diag, self._initial_weight)
        self.back = _Welford(self.n)
        self.n_samples = 0
        # the reference's reset() leaves a grown adaptation_window in place (it only matters for multiplier != 1 and
        # for chains after the first of its sequential driver); every chain here starts like the reference's first
        self.window = self._initial_window

    def velocity(self, x):  # :206-208
        return np.multiply(self.var, x)

    def velocity_into(self, x, out):  # :216-219
        np.multiply(self.var, x, out=out)

    def random(self, rng):  # :221-224
        vals = rng.normal(size=self.n).astype(self.dtype)
        return self.inv_stds * vals

    def update(self, sample, tune):  # :231-245
        if not tune:
            return
        self.fore.add(sample)
        self.back.add(sample)
        np.divide(self.fore.raw_var, self.fore.w_sum, out=self.var)  # f64 quotient -> dtype
        np.sqrt(self.var, out=self.stds)
        np.divide(1, self.stds, out=self.inv_stds)
        if self.n_samples > 0 and self.n_samples % self.window == 0:
            self.fore = self.back
            self.back = _Welford(self.n)
            self.window = int(self.window * self.multiplier)  # :243
        self.n_samples += 1


class DiagPotential:
    """quadpotential.py:346-387 (QuadPotentialDiag): fixed float32 diagonal, float64 momentum."""

    momentum_f32 = False

    def __init__(self, v):
        v = np.asarray(v).astype("float32")
        self.var = v
        self.stds = v ** 0.5
        self.inv_stds = 1.0 / self.stds
        self.n = v.shape[0]
        self.n_samples = 0

    def reset(self):
        pass

    def velocity(self, x):
        return self.var * x

    def velocity_into(self, x, out):  # :378-387
        np.multiply(self.var, x, out=out)

    def random(self, rng):  # :374-376
        return rng.normal(size=self.var.shape) * self.inv_stds

    def update(self, sample, tune):
        pass


class _WelfordCov:
    """quadpotential.py:560-615 (_WeightedCovariance): float64 mean and raw second moment, full d x d."""

    def __init__(self, n, mean=None, cov=None, weight=0):
        self.n_samples = float(weight)
        self.mean = np.zeros(n) if mean is None else np.array(mean, dtype="d", copy=True)
        self.raw = np.eye(n) if cov is None else np.array(cov, dtype="d", copy=True)
        self.raw[:] *= self.n_samples

    def add(self, x):  # :594-600 with weight == 1 (rows scale with the NEW residual, columns with the OLD one)
        self.n_samples += 1
        old = x - self.mean
        self.mean[:] += old / self.n_samples
        new = x - self.mean
        self.raw[:] += 1 * new[:, None] * old[None, :]


class FullPotential:
    """quadpotential.py:428-468 (QuadPotentialFull): float32 covariance, float32 momentum through a
    triangular solve with the transposed Cholesky factor."""

    momentum_f32 = True
    dense = True

    def __init__(self, cov, dtype="float32"):
        self.cov = np.array(cov, dtype=dtype, copy=True)
        self.chol = scipy.linalg.cholesky(self.cov, lower=True)
        self.n = len(self.cov)
        self.n_samples = 0
        self.momentum_f32 = self.cov.dtype == np.float32   # :451: normal(size=n).astype(self.dtype)

    def reset(self):
        pass

    def velocity(self, x):  # :446-448 (float32 x -> float32 sgemv, float64 x -> promoted dgemv)
        return np.dot(self.cov, x)

    def velocity_into(self, x, out):  # :461-464
        np.dot(self.cov, x, out=out)

    def random(self, rng):  # :450-453
        vals = rng.normal(size=self.n).astype(self.cov.dtype)
        return scipy.linalg.solve_triangular(self.chol.T, vals, overwrite_b=True)

    def update(self, sample, tune):
        pass


class FullInvPotential:
    """quadpotential.py:388-425 (QuadPotentialFullInv): mass matrix A given, velocity by two triangular solves,
    float64 momentum L n."""

    momentum_f32 = False
    dense = True

    def __init__(self, A):
        self.L = scipy.linalg.cholesky(A, lower=True)
        self.n = self.L.shape[0]
        self.n_samples = 0

    def reset(self):
        pass

    def velocity(self, x):  # :404-409
        return scipy.linalg.cho_solve((self.L, True), x)

    def velocity_into(self, x, out):
        out[:] = scipy.linalg.cho_solve((self.L, True), x)

    def random(self, rng):  # :411-414
        return np.dot(self.L, rng.normal(size=self.n))

    def update(self, sample, tune):
        pass


class FullAdaptPotential(FullPotential):
    """quadpotential.py:471-557 (QuadPotentialFullAdapt): foreground/background covariance estimators, the
    float32 covariance and its Cholesky factor refreshed every ``update_window`` tuning samples, the
    adaptation window growing by ``multiplier`` each time the background estimator takes over.

    The reference's ``reset()`` is the base-class no-op (quadpotential.py:136-138), so its SEQUENTIAL driver hands
    chain k the matrix chain k-1 ended with; its multi-process driver gives every chain a fresh copy. ``reset``
    here restores the constructor state, i.e. every chain is what the reference computes for a one-chain run with
    that chain's seed (the goldens are captured that way)."""

    def __init__(self, n, initial_mean, initial_cov=None, initial_weight=0, adaptation_window=101,
                 adaptation_window_multiplier=2, update_window=1, dtype="float32"):
        if initial_cov is None:  # :501-503
            initial_cov = np.eye(n, dtype=dtype)
            initial_weight = 1
        self._init = (n, np.array(initial_mean, dtype="d"), np.array(initial_cov), initial_weight,
                      int(adaptation_window), float(adaptation_window_multiplier), int(update_window), dtype)
        self.reset()

    def reset(self):
        n, mean, cov, weight, window, mult, upd, dtype = self._init
        self.n = n
        self.cov = np.array(cov, dtype=dtype, copy=True)
        self.momentum_f32 = self.cov.dtype == np.float32
        self.chol = scipy.linalg.cholesky(self.cov, lower=True)
        self.chol_error = None
        self.fore = _WelfordCov(n, mean, cov, weight)
        self.back = _WelfordCov(n)
        self.n_samples = 0
        self.window, self.multiplier, self.update_window = window, mult, upd
        self.previous_update = 0

    def _refresh(self, est):  # :521-526
        np.divide(est.raw, est.n_samples - 1, out=self.cov)  # float64 quotient stored in the potential's dtype
        try:
            self.chol = scipy.linalg.cholesky(self.cov, lower=True)
        except (scipy.linalg.LinAlgError, ValueError) as error:
            self.chol_error = error  # the old factor stays in use

    def update(self, sample, tune):  # :528-552
        if not tune:
            return
        delta = self.n_samples - self.previous_update
        self.fore.add(sample)
        self.back.add(sample)
        if (delta + 1) % self.update_window == 0:
            self._refresh(self.fore)
        if delta >= self.window:
            self.fore = self.back
            self.back = _WelfordCov(self.n)
            self.previous_update = self.n_samples
            self.window = int(self.window * self.multiplier)
        self.n_samples += 1


def quad_potential(scaling, is_cov):
    """quadpotential.py:33-65 for dense arrays: covariance -> Full, precision/Hessian -> FullInv."""
    c = np.asarray(scaling)
    if c.ndim == 1:
        return quad_potential_diag(c, is_cov)
    dg = np.diag(c)
    bad = np.nonzero(np.logical_or(np.isnan(dg), dg <= 0))[0]
    if len(bad):
        raise ValueError("Scaling is not positive definite. Check indexes %s." % (bad,))
    return FullPotential(c) if is_cov else FullInvPotential(c)


def quad_potential_diag(scaling, is_cov):
    """quadpotential.py:33-77 restricted to 1-D scalings."""
    c = np.asarray(scaling)
    bad = np.nonzero(np.logical_or(np.isnan(c), c <= 0))[0]
    if len(bad):
        raise ValueError("Scaling is not positive definite. Check indexes %s." % (bad,))
    return DiagPotential(c if is_cov else 1.0 / c)


# --------------------------------------------------------------------------------------
# integration.py
# --------------------------------------------------------------------------------------
# Test hook (tools/same_seed_evidence.py): a function (x, y) -> float32 that stands in for the host BLAS's sdot in the ONE
# place its summation order decides how long two same-seed chains stay together -- the float32 kinetic energy of the start
# state. None (always, except in that tool): numpy's own dot, i.e. what the reference computes on this host.
START_SDOT = None


def compute_state(pot, f, q, p):
    """integration.py:52-66. With a float32 p this leaves v and the kinetic term float32."""
    logp, g = f(q)
    v = pot.velocity(p)
    if START_SDOT is not None and p.dtype == np.float32 and v.dtype == np.float32:
        kinetic = np.float32(0.5) * START_SDOT(p, v)
    else:
        kinetic = 0.5 * p.dot(v)
    return State(q, p, v, g, kinetic - logp, logp)


def leapfrog(pot, f, eps, s):
    """integration.py:100-121."""
    dt = 0.5 * eps
    p = s.p + dt * s.g
    v = pot.velocity(p)
    q = (s.q + eps * v).astype(s.q.dtype)
    logp, g = f(q)
    p = p + dt * g
    pot.velocity_into(p, v)   # in place, integration.py:118
    kinetic = 0.5 * np.dot(p, v)
    return State(q, p, v, g, kinetic - logp, logp)


# --------------------------------------------------------------------------------------
# step_sizes.py
# --------------------------------------------------------------------------------------
class DualAverage:
    """step_sizes.py:23-99."""

    def __init__(self, initial_step, target, gamma, k, t0):
        self.initial_step, self.target, self.gamma, self.k, self.t0 = initial_step, target, gamma, k, t0
        self.reset()

    def reset(self):
        self.log_step = np.log(self.initial_step)
        self.log_bar = self.log_step
        self.hbar = 0.0
        self.count = 1
        self.mu = np.log(10 * self.initial_step)
        self.tuned_stats = []

    def current(self, tune):
        return np.exp(self.log_step) if tune else np.exp(self.log_bar)

    def update(self, accept, tune):
        if not tune:
            self.tuned_stats.append(accept)
            return
        count = self.count
        w = 1.0 / (count + self.t0)
        self.hbar = (1 - w) * self.hbar + w * (self.target - accept)
        self.log_step = self.mu - self.hbar * np.sqrt(count) / self.gamma
        mk = count ** -self.k
        self.log_bar = mk * self.log_step + (1 - mk) * self.log_bar
        self.count += 1


# --------------------------------------------------------------------------------------
# math.py
# --------------------------------------------------------------------------------------
def _log1mexp(x):  # math.py:28-35
    return np.log(-np.expm1(-x)) if x < 0.683 else np.log1p(-np.exp(-x))


class _Margins:
    """Smallest decision margins seen in one transition (test aid, not part of the algorithm)."""

    __slots__ = ("lb", "turn", "div")

    def __init__(self):
        self.lb = np.inf
        self.turn = np.inf
        self.div = np.inf


def _logbern(rng, log_p, m):  # math.py:21-25
    if np.isnan(log_p):
        raise FloatingPointError("log_p can't be nan.")
    with np.errstate(divide="ignore"):
        lu = np.log(rng.uniform())
    gap = abs(float(np.ravel(lu - log_p)[0]))
    if gap < m.lb:
        m.lb = gap
    return lu < log_p


def _turn(m, ps, v):
    d = ps.dot(v)
    a = abs(float(np.ravel(d)[0]))
    if a < m.turn:
        m.turn = a
    return d <= 0


# --------------------------------------------------------------------------------------
# nuts.py
# --------------------------------------------------------------------------------------
class _Node:
    """nuts.py:243-248 Subtree + Proposal, flattened. ``l``/``r`` are States (integration order)."""

    __slots__ = ("l", "r", "psum", "prop", "ls", "lwas")

    def __init__(self, l, r, psum, prop, ls, lwas):
        self.l, self.r, self.psum, self.prop, self.ls, self.lwas = l, r, psum, prop, ls, lwas


def nuts_transition(pot, f, rng, start, step_size, emax, max_depth):
    """One NUTS transition: nuts.py:204-224 (_hamiltonian_step) + :251-435 (_Tree).

    Returns (proposal State, stats dict, diverging, reached_max_depth, margins).
    """
    m = _Margins()
    e0 = np.array(start.energy)  # nuts.py:273
    left = right = start
    prop = start
    depth = 0
    log_size = 0
    lwas = -np.inf
    n_leap = 0
    psum = start.p.copy()  # float32 when the momentum draw is float32 (nuts.py:281)
    max_de = 0
    diverging = False
    turning = False
    exhausted = True

    for _ in range(max_depth):
        direction = 1 if _logbern(rng, np.log(0.5), _Margins()) else -1  # nuts.py:213
        edge = right if direction > 0 else left
        eps = np.asarray(direction * step_size)

        # ---- balanced subtree of 2**depth leaves, post-order with an explicit stack
        #      (replaces the recursion of nuts.py:377-417)
        stack = [None] * (depth + 1)
        cur = edge
        sub = None
        for i in range(1 << depth):
            cur = leapfrog(pot, f, eps, cur)  # nuts.py:344-347
            n_leap += 1
            de = cur.energy - e0
            if np.isnan(de):
                de = np.inf
            if np.abs(de) > np.abs(max_de):
                max_de = de
            gap = abs(abs(float(np.ravel(de)[0])) - emax)
            if gap < m.div:
                m.div = gap
            if not (np.abs(de) < emax):  # nuts.py:358,370-375
                diverging = True
                break
            node = _Node(cur, cur, cur.p, cur, -de, -de + min(0.0, -de))
            j = 0
            while (i >> j) & 1:  # node closes a right child: merge stack[j] (earlier) with it
                a, b = stack[j], node
                ps = a.psum + b.psum
                t = _turn(m, ps, a.l.v) | _turn(m, ps, b.r.v)  # nuts.py:389
                if j > 0:  # nuts.py:391-396
                    p1 = a.psum + b.l.p
                    t |= _turn(m, p1, a.l.v) | _turn(m, p1, b.l.v)
                    p2 = a.r.p + b.psum
                    t |= _turn(m, p2, a.r.v) | _turn(m, p2, b.r.v)
                ls = np.logaddexp(a.ls, b.ls)
                lw = np.logaddexp(a.lwas, b.lwas)
                pr = b.prop if _logbern(rng, b.ls - ls, m) else a.prop  # nuts.py:404
                node = _Node(a.l, b.r, ps, pr, ls, lw)
                j += 1
                if t:
                    turning = True
                    break
            if turning:
                break
            stack[j] = node
        else:
            sub = stack[depth]
        depth += 1  # nuts.py:315
        if diverging or turning:
            exhausted = False
            break

        # ---- accepted subtree: merge into the trajectory (nuts.py:321-340)
        old_l, old_r = left, right
        if direction > 0:
            right = sub.r
        else:
            left = sub.r
        if _logbern(rng, sub.ls - log_size, m):
            prop = sub.prop
        log_size = np.logaddexp(log_size, sub.ls)
        lwas = np.logaddexp(lwas, sub.lwas)
        psum[:] += sub.psum  # in place: rounds to float32 when psum is float32 (nuts.py:329)
        t = _turn(m, psum, left.v) | _turn(m, psum, right.v)
        if direction > 0:
            # leftmost_p_sum aliases the already-updated total (nuts.py:302,329,336)
            p1 = psum + sub.l.p
            t |= _turn(m, p1, old_l.v) | _turn(m, p1, sub.l.v)
            p2 = old_r.p + sub.psum
            t |= _turn(m, p2, old_r.v) | _turn(m, p2, sub.r.v)
        else:
            p1 = sub.psum + old_l.p
            t |= _turn(m, p1, sub.r.v) | _turn(m, p1, old_l.v)
            # rightmost_p_sum aliases the already-updated total (nuts.py:312,329,338)
            p2 = sub.l.p + psum
            t |= _turn(m, p2, sub.l.v) | _turn(m, p2, old_r.v)
        if t:
            turning = True
            exhausted = False
            break

    mean_accept = 0.0
    if log_size > 0:  # nuts.py:421-425
        mean_accept = np.exp(lwas - (log_size + _log1mexp(log_size - 0.0)))
    stats = {
        "depth": depth,
        "mean_tree_accept": mean_accept,
        "energy_error": prop.energy - start.energy,
        "energy": prop.energy,
        "tree_size": n_leap,
        "max_energy_error": max_de,
        "model_logp": prop.logp,
    }
    return prop, stats, diverging, exhausted, m


# --------------------------------------------------------------------------------------
# hmc.py
# --------------------------------------------------------------------------------------
def hmc_transition(pot, f, rng, start, step_size, emax, path_length, max_steps):
    """hmc.py:140-182."""
    m = _Margins()
    plen = rng.rand() * path_length
    n_steps = max(1, int(plen / step_size))
    n_steps = min(max_steps, n_steps)
    state = start
    for _ in range(n_steps):
        state = leapfrog(pot, f, step_size, state)
    diverging = False
    if not np.isfinite(state.energy):
        diverging = True
    de = start.energy - state.energy
    if np.isnan(de):
        de = -np.inf
    if np.abs(de) > emax:
        diverging = True
    with np.errstate(over="ignore"):
        accept = min(1, np.exp(de))
    accepted = False
    end = start
    if not diverging:
        u = rng.rand()
        m.lb = abs(u - float(np.ravel(accept)[0]))
        if not (u >= accept):
            end = state
            accepted = True
    stats = {
        "path_length": plen,
        "n_steps": n_steps,
        "accept": accept,
        "energy_error": de,
        "energy": state.energy,
        "accepted": accepted,
        "model_logp": state.logp,
    }
    return end, stats, diverging, accept, m


# --------------------------------------------------------------------------------------
# base_hmc.py + nuts.py/hmc.py constructors
# --------------------------------------------------------------------------------------
NUTS_STATS = {  # nuts.py:87-101
    "depth": np.int64, "step_size": np.float64, "tune": np.bool_, "mean_tree_accept": np.float64,
    "step_size_bar": np.float64, "tree_size": np.float64, "diverging": np.bool_,
    "energy_error": np.float64, "energy": np.float64, "max_energy_error": np.float64,
    "model_logp": np.float64,
}
HMC_STATS = {  # hmc.py:36-50
    "step_size": np.float64, "n_steps": np.int64, "tune": np.bool_, "step_size_bar": np.float64,
    "accept": np.float64, "diverging": np.bool_, "energy_error": np.float64, "energy": np.float64,
    "path_length": np.float64, "accepted": np.bool_, "model_logp": np.float64,
}


class Step:
    """BaseHMC (base_hmc.py:29-200) with kind in {"nuts","hmc"}; one chain at a time."""

    def __init__(self, f, d, kind="nuts", scaling=None, is_cov=False, potential=None,
                 target_accept=0.8, Emax=1000, adapt_step_size=True, step_scale=0.25, gamma=0.05,
                 k=0.75, t0=10, path_length=2.0, max_treedepth=10, early_max_treedepth=8,
                 max_steps=1024, step_rand=None):
        self.f, self.d, self.kind = f, d, kind
        # base_hmc.py:46,123,154-155: step_rand(step_size) -> step_size, any callable in the reference. Restated for the
        # one form that can keep same-seed parity: (lo, hi) stands for  lambda s: s * np.random.uniform(lo, hi)  -- ONE
        # uniform from the chain's stream, drawn where the reference calls it (after the momentum draw, before the tree)
        self.step_rand = step_rand
        self.Emax = Emax
        self.adapt_step_size = adapt_step_size
        self.step_size = step_scale / (d ** 0.25)  # base_hmc.py:102
        self.adapt = DualAverage(self.step_size, target_accept, gamma, k, t0)
        self.tune = True
        self.iter_count = 0
        if scaling is None and potential is None:  # base_hmc.py:109-113
            potential = DiagAdaptPotential(d, np.zeros(d), np.ones(d), 10)
        if scaling is not None and potential is not None:
            raise ValueError("Cannot specify both `potential` and `scaling`.")
        self.pot = potential if potential is not None else quad_potential(scaling, is_cov)
        self.path_length = path_length
        self.max_treedepth = max_treedepth
        self.early_max_treedepth = early_max_treedepth
        self.max_steps = max_steps
        self.reached_max_treedepth = 0
        self.samples_after_tune = 0
        self.num_divs_sample = 0
        self.last_margins = None
        # True: the reference's SEQUENTIAL driver to the letter -- one step object for all chains (sampling.py:370-383) whose
        # FullAdapt potential is never reset, so chain k starts from chain k-1's matrix, estimators and grown window.
        # False (default, what the device does and what the reference's multi-process driver computes): every chain fresh.
        self.sequential_carry_over = False
        self.stats_dtypes = NUTS_STATS if kind == "nuts" else HMC_STATS

    def reset_tuning(self):  # base_hmc.py:192-200
        self.adapt.reset()
        self.tune = True
        if not (self.sequential_carry_over and isinstance(self.pot, FullAdaptPotential)):
            self.pot.reset()   # (QuadPotentialFullAdapt inherits the base class's no-op reset, quadpotential.py:137-139)

    def astep(self, q0, rng):
        """base_hmc.py:140-190."""
        p0 = self.pot.random(rng)
        start = compute_state(self.pot, self.f, q0, p0)
        if not np.isfinite(start.energy):
            raise ValueError("Bad initial energy: {}. The model might be misspecified.".format(start.energy))
        adapt_step = self.tune and self.adapt_step_size
        step_size = self.adapt.current(adapt_step)
        self.step_size = step_size
        if callable(self.step_rand):     # base_hmc.py:154-155, any function of the step size (a deterministic one here:
            step_size = self.step_rand(step_size)   # this restatement threads its RandomState explicitly, np.random is not the chain's)
        elif self.step_rand is not None:
            lo, hi = self.step_rand
            step_size = step_size * rng.uniform(lo, hi)
        if self.kind == "nuts":
            md = self.early_max_treedepth if (self.tune and self.iter_count < 200) else self.max_treedepth
            end, stats, diverging, exhausted, m = nuts_transition(
                self.pot, self.f, rng, start, step_size, self.Emax, md)
            if exhausted and not self.tune:
                self.reached_max_treedepth += 1
            accept = stats["mean_tree_accept"]
        else:
            end, stats, diverging, accept, m = hmc_transition(
                self.pot, self.f, rng, start, step_size, self.Emax, self.path_length, self.max_steps)
        self.last_margins = m
        self.adapt.update(accept, adapt_step)
        self.pot.update(end.q, self.tune)
        if diverging and not self.tune:
            self.num_divs_sample += 1
        self.iter_count += 1
        if not self.tune:
            self.samples_after_tune += 1
        out = {"tune": self.tune, "diverging": bool(diverging)}
        out.update(stats)
        out["step_size"] = np.exp(self.adapt.log_step)  # values for the NEXT iteration
        out["step_size_bar"] = np.exp(self.adapt.log_bar)
        return end.q, out


# --------------------------------------------------------------------------------------
# sampling.py
# --------------------------------------------------------------------------------------
def derive_seeds(random_seed, chains):
    """sampling.py:131-138 for an int seed."""
    rs = np.random.RandomState(random_seed)
    return [int(rs.randint(2 ** 30)) for _ in range(chains)]


def jitter_start(seed0, d):
    """sampling.py:574-576,584: seed(seeds[0]); 2*rand(d)-1."""
    rs = np.random.RandomState(int(seed0))
    return 2 * rs.rand(d) - 1


def init_nuts(f, d, init="auto", seeds=None, **kwargs):
    """sampling.py:524-605."""
    if not isinstance(init, str):
        raise TypeError("init must be a string.")
    init = init.lower()
    if init == "auto":
        init = "jitter+adapt_diag"
    if init == "adapt_diag":
        start = np.zeros(d)
    elif init == "jitter+adapt_diag":
        start = jitter_start(seeds[0], d) if seeds is not None else 2 * np.random.rand(d) - 1
    elif init == "adapt_full":
        start = np.zeros(d)
    elif init == "jitter+adapt_full":
        start = jitter_start(seeds[0], d) if seeds is not None else 2 * np.random.rand(d) - 1
    else:
        raise ValueError("Unknown initializer: {}.".format(init))
    if init.endswith("adapt_full"):  # sampling.py:588-597
        pot = FullAdaptPotential(d, start, np.eye(d), 10)
    else:
        pot = DiagAdaptPotential(d, start, np.ones(d), 10)
    return start, Step(f, d, kind="nuts", potential=pot, **kwargs)


def sample(f, d, draws=1000, tune=1000, step=None, init="auto", chains=2, start=None,
           random_seed=None, discard_tuned_samples=True, record_margins=False, sequential_carry_over=False, **kwargs):
    """sampling.py:35-222 sequential path (cores=1): returns (trace[chains,draws,d], stats).
    ``sequential_carry_over``: see Step.sequential_carry_over (pinned by tests/golden/e2e_adaptfull_two_chains.npz)."""
    if isinstance(random_seed, (int, np.integer)):
        seeds = derive_seeds(int(random_seed), chains)
    else:
        seeds = [int(s) for s in list(random_seed)[:chains]]
    if step is None or start is None:
        start_, step_ = init_nuts(f, d, init=init, seeds=seeds, **kwargs)
        step = step_ if step is None else step
        start = start_ if start is None else start
    starts = [start] * chains if isinstance(start, np.ndarray) and start.ndim == 1 else list(start)
    step.sequential_carry_over = bool(sequential_carry_over)
    n = tune + draws
    trace = np.zeros((chains, n, d))
    stats = {k: np.zeros((chains, n, 1), dtype=dt) for k, dt in step.stats_dtypes.items()}
    margins = np.full((chains, n, 3), np.inf)
    for c in range(chains):
        rng = np.random.RandomState(seeds[c])  # sampling.py:496-497
        q = starts[c]
        step.tune = bool(tune)
        step.reset_tuning()
        for i in range(n):
            if i == 0:
                step.iter_count = 0
            if i == tune:
                step.tune = False
            q, st = step.astep(q, rng)
            trace[c, i] = q
            for k in stats:
                stats[k][c, i, 0] = np.ravel(st[k])[0]
            mm = step.last_margins
            margins[c, i] = (mm.lb, mm.turn, mm.div)
    if discard_tuned_samples:
        trace = trace[:, tune:]
        stats = {k: v[:, tune:] for k, v in stats.items()}
        margins = margins[:, tune:]
    if record_margins:
        return trace, stats, margins
    return trace, stats
