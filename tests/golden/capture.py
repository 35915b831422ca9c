#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the IMPORTED reference.

Runs only in the build container (needs /root/reference, which does not exist on the GPU
box). Nothing of the reference's source travels: the outputs are plain arrays (inputs and
expected outputs) saved as ``*.npz``. Re-run with

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/capture.py

``fastprogress`` (a progress-bar dependency of the reference, not installed here) is replaced
by a no-op stub created in a temp dir; nothing is written under /root/reference.

Only the sequential path (``cores=1``) is captured: the reference's multi-process path returns
the start point for every draw.
"""
import os
import sys
import tempfile

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

_stub = tempfile.mkdtemp(prefix="fp_stub_")
os.makedirs(os.path.join(_stub, "fastprogress"))
open(os.path.join(_stub, "fastprogress", "__init__.py"), "w").close()
with open(os.path.join(_stub, "fastprogress", "fastprogress.py"), "w") as fh:
    fh.write(
        "class progress_bar:\n"
        "    def __init__(self, gen, total=None, display=True, **kw):\n"
        "        self.gen, self.total, self.comment = gen, total, ''\n"
        "    def __iter__(self):\n"
        "        return iter(self.gen)\n"
        "    def update(self, val):\n"
        "        pass\n"
    )
sys.path.insert(0, _stub)
sys.path.insert(0, "/root/reference")

import logging  # noqa: E402

import numpy as np  # noqa: E402

import littlemcmc as ref  # noqa: E402
from littlemcmc.quadpotential import QuadPotentialDiagAdapt, _WeightedVariance  # noqa: E402
from littlemcmc.step_sizes import DualAverageAdaptation  # noqa: E402

from oracle import targets  # noqa: E402

logging.getLogger("littlemcmc").setLevel(logging.ERROR)
SEED = 20260928


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("%-28s %8.1f KiB" % (name + ".npz", os.path.getsize(path) / 1024.0))


def state_rows(states):
    return dict(
        q=np.array([np.asarray(s.q, dtype="d") for s in states]),
        p=np.array([np.asarray(s.p, dtype="d") for s in states]),
        v=np.array([np.asarray(s.v, dtype="d") for s in states]),
        g=np.array([np.asarray(s.q_grad, dtype="d") for s in states]),
        energy=np.array([float(np.ravel(s.energy)[0]) for s in states]),
        logp=np.array([float(np.ravel(s.model_logp)[0]) for s in states]),
    )


# ---------------------------------------------------------------------------------------
# 1. integrator unit fixtures: compute_state + n steps forward then n steps back
# ---------------------------------------------------------------------------------------
def capture_leapfrog():
    out = {}
    cases = []
    for fam, d in [("std_normal", 1), ("std_normal", 10), ("std_normal", 128), ("ar1", 10),
                   ("funnel", 8), ("diag_gaussian", 50), ("ar1", 200)]:
        for pot_kind in ("adapt", "scaling"):
            cases.append((fam, d, pot_kind))
    for ci, (fam, d, pot_kind) in enumerate(cases):
        f = targets.make(fam, d)
        np.random.seed(1000 + ci)
        scaling = (0.5 + np.random.rand(d))
        if pot_kind == "adapt":
            pot = QuadPotentialDiagAdapt(d, np.zeros(d), scaling, 10)
            step = ref.HamiltonianMC(f, d, potential=pot)
        else:
            step = ref.HamiltonianMC(f, d, scaling=scaling, is_cov=True)
        q0 = 0.3 * np.random.randn(d)
        p0 = step.potential.random()
        eps = 0.07
        n = 20 if d <= 16 else 5
        s = step.integrator.compute_state(q0, p0)
        states = [s]
        for _ in range(n):
            s = step.integrator.step(eps, s)
            states.append(s)
        for _ in range(n):
            s = step.integrator.step(-eps, s)
            states.append(s)
        rows = state_rows(states)
        key = "c%d_" % ci
        out[key + "family"] = np.array(fam)
        out[key + "pot"] = np.array(pot_kind)
        out[key + "d"] = np.array(d)
        out[key + "eps"] = np.array(eps)
        out[key + "n"] = np.array(n)
        out[key + "var"] = np.asarray(step.potential._var if pot_kind == "adapt" else step.potential.v)
        out[key + "p0_dtype"] = np.array(str(p0.dtype))
        out[key + "params"] = f.params()
        for k, v in rows.items():
            out[key + k] = v
    out["n_cases"] = np.array(len(cases))
    save("leapfrog", **out)


# ---------------------------------------------------------------------------------------
# 2. iteration sequences at a FIXED step size (tune off): momentum draw + transition
# ---------------------------------------------------------------------------------------
def capture_transitions():
    out = {}
    cases = [
        # family, d, kind, eps, iters, extra kwargs
        ("std_normal", 2, "nuts", 0.9, 400, {}),
        ("std_normal", 16, "nuts", 0.55, 300, {}),
        ("std_normal", 128, "nuts", 0.6, 60, {}),
        ("funnel", 8, "nuts", 0.45, 300, {"max_treedepth": 12}),
        ("ar1", 16, "nuts", 0.2, 150, {}),
        ("std_normal", 2, "nuts", 0.05, 40, {"max_treedepth": 5}),  # hits max depth
        ("std_normal", 10, "hmc", 0.4, 300, {"path_length": 2.0}),
        ("funnel", 8, "hmc", 0.5, 200, {"path_length": 3.0}),
        ("std_normal", 3, "nuts_scaling", 0.7, 200, {}),
    ]
    for ci, (fam, d, kind, eps, iters, kw) in enumerate(cases):
        f = targets.make(fam, d)
        if kind == "nuts":
            step = ref.NUTS(f, d, adapt_step_size=False, **kw)
        elif kind == "nuts_scaling":
            step = ref.NUTS(f, d, scaling=np.linspace(0.5, 2.0, d), is_cov=True, adapt_step_size=False, **kw)
        else:
            step = ref.HamiltonianMC(f, d, adapt_step_size=False, **kw)
        step.tune = False
        step.step_adapt._log_bar = np.log(eps)
        step.step_adapt._log_step = np.log(eps)
        seed = 4242 + ci
        np.random.seed(seed)
        q = np.full(d, 0.1)
        qs, stats = [], []
        for _ in range(iters):
            q, st = step._astep(q)
            qs.append(np.array(q, dtype="d"))
            stats.append(st[0])
        key = "c%d_" % ci
        out[key + "family"] = np.array(fam)
        out[key + "kind"] = np.array(kind)
        out[key + "d"] = np.array(d)
        out[key + "eps"] = np.array(eps)
        out[key + "seed"] = np.array(seed)
        out[key + "iters"] = np.array(iters)
        out[key + "q0"] = np.full(d, 0.1)
        out[key + "kw_names"] = np.array(list(kw.keys()), dtype="U32")
        out[key + "kw_vals"] = np.array(list(kw.values()), dtype="d")
        out[key + "q"] = np.array(qs)
        for name, dt in step.stats_dtypes[0].items():
            out[key + "stat_" + name] = np.array([np.ravel(s[name])[0] for s in stats]).astype(dt)
        out[key + "final_rng_pos"] = np.array(np.random.get_state()[2])
        out[key + "final_rng_key0"] = np.array(np.random.get_state()[1][:4])
    out["n_cases"] = np.array(len(cases))
    save("transitions", **out)


# ---------------------------------------------------------------------------------------
# 3. adaptation unit fixtures
# ---------------------------------------------------------------------------------------
def capture_adapt():
    rs = np.random.RandomState(7)
    accepts = rs.beta(4, 1.5, size=300)
    da = DualAverageAdaptation(0.25 / 128 ** 0.25, 0.8, 0.05, 0.75, 10)
    rows = []
    for a in accepts:
        da.update(a, True)
        rows.append((da._log_step, da._log_bar, da._hbar, da._count))
    d = 12
    samples = rs.randn(250, d) * np.linspace(0.1, 30.0, d) + np.linspace(-3, 3, d)
    mean0 = rs.randn(d)
    pot = QuadPotentialDiagAdapt(d, mean0, np.ones(d), 10)
    var_rows, istd_rows = [], []
    for x in samples:
        pot.update(x, None, True)
        var_rows.append(pot._var.copy())
        istd_rows.append(pot._inv_stds.copy())
    save(
        "adapt",
        accepts=accepts, initial_step=np.array(0.25 / 128 ** 0.25), da=np.array(rows, dtype="d"),
        samples=samples, mean0=mean0, var=np.array(var_rows), inv_stds=np.array(istd_rows),
        n_samples=np.array(pot._n_samples), fore_mean=pot._foreground_var.mean,
        fore_raw_var=pot._foreground_var.raw_var, fore_w=np.array(pot._foreground_var.w_sum),
        back_mean=pot._background_var.mean, back_raw_var=pot._background_var.raw_var,
        back_w=np.array(pot._background_var.w_sum),
    )


# ---------------------------------------------------------------------------------------
# 4/5. end-to-end sample() runs (sequential path)
# ---------------------------------------------------------------------------------------
def capture_e2e(only=None):
    runs = [
        # name, family, d, kind, chains, tune, draws, kwargs
        ("e2e_hmc_c1", "std_normal", 10, "hmc", 4, 300, 200, {"path_length": 2.0}),
        ("e2e_nuts_std64", "std_normal", 64, "nuts", 2, 250, 150, {}),
        ("e2e_nuts_std128", "std_normal", 128, "nuts", 2, 110, 20, {}),
        ("e2e_nuts_ar1_16", "ar1", 16, "nuts", 4, 250, 150, {}),
        ("e2e_nuts_funnel8", "funnel", 8, "nuts", 4, 250, 150, {"max_treedepth": 12}),
        ("e2e_nuts_diag50", "diag_gaussian", 50, "nuts", 2, 250, 100, {}),
        ("e2e_nuts_normal1d", "normal1d", 1, "nuts", 2, 120, 80, {}),
        # the benchmarked instantiation (BASELINE config C3's shape: AR(1) at d = 128, two elements per lane)
        ("e2e_nuts_ar1_128", "ar1", 128, "nuts", 2, 250, 50, {}),
        # round 6: the exact instantiations of BASELINE configs C5 and C4 -- run_kernel<4, 1, FunnelTarget> (d = 256, depth 12)
        # and run_kernel<4, 4, DiagGaussianTarget> (d = 1000, a team of four wavefronts per chain)
        ("e2e_nuts_funnel256", "funnel", 256, "nuts", 2, 60, 20, {"max_treedepth": 12}),
        ("e2e_nuts_diag1000", "diag_gaussian", 1000, "nuts", 2, 60, 20, {}),
    ]
    for name, fam, d, kind, chains, tune, draws, kw in runs:
        if only and name not in only:
            continue
        f = targets.make(fam, d)
        if kind == "hmc":
            step = ref.HamiltonianMC(f, d, **kw)
            trace, stats = ref.sample(f, d, draws=draws, tune=tune, step=step, chains=chains, cores=1,
                                      progressbar=False, random_seed=SEED, discard_tuned_samples=False)
        else:
            # the plain API call; init_nuts is wrapped only to get hold of the step object it builds
            import littlemcmc.sampling as ref_sampling
            made = {}
            orig = ref_sampling.init_nuts

            def spy(*a, **k):
                st, sp = orig(*a, **k)
                made["step"] = sp
                return st, sp

            ref_sampling.init_nuts = spy
            try:
                trace, stats = ref.sample(f, d, draws=draws, tune=tune, chains=chains, cores=1,
                                          progressbar=False, random_seed=SEED,
                                          discard_tuned_samples=False, **kw)
            finally:
                ref_sampling.init_nuts = orig
            step = made["step"]
        np.random.seed(SEED)
        seeds = np.array([np.random.randint(2 ** 30) for _ in range(chains)])
        np.random.seed(int(seeds[0]))
        jitter = 2 * np.random.rand(d) - 1
        arrays = dict(
            family=np.array(fam), kind=np.array(kind), d=np.array(d), chains=np.array(chains),
            tune=np.array(tune), draws=np.array(draws), random_seed=np.array(SEED), seeds=seeds,
            start=jitter, params=f.params(), trace=trace,
            kw_names=np.array(list(kw.keys()), dtype="U32"), kw_vals=np.array(list(kw.values()), dtype="d"),
            final_var=np.asarray(step.potential._var),
            final_da=np.array([float(np.ravel(x)[0]) for x in (
                step.step_adapt._log_step, step.step_adapt._log_bar, step.step_adapt._hbar,
                step.step_adapt._count)]),
            final_n_samples=np.array(step.potential._n_samples),
        )
        for k, v in stats.items():
            arrays["stat_" + k] = v
        save(name, **arrays)


# ---------------------------------------------------------------------------------------
# 6. seed derivation
# ---------------------------------------------------------------------------------------
def capture_seeds():
    out = {}
    for chains in (2, 4, 64):
        np.random.seed(SEED)
        seeds = np.array([np.random.randint(2 ** 30) for _ in range(chains)])
        np.random.seed(int(seeds[0]))
        out["seeds_%d" % chains] = seeds
        out["jitter_%d" % chains] = 2 * np.random.rand(7) - 1
    # raw stream checks for the device RNG: seed -> u32 words, doubles, normals
    rs = np.random.RandomState(12345)
    out["stream_seed"] = np.array(12345)
    out["stream_doubles"] = rs.random_sample(700)
    out["stream_normals"] = rs.normal(size=1001)
    out["stream_after_uniform"] = np.array(rs.uniform())
    out["stream_normals2"] = rs.normal(size=10)
    out["stream_state_pos"] = np.array(rs.get_state()[2])
    save("seeds", **out)

# ---------------------------------------------------------------------------------------
# 7. dense mass matrices: unit values, estimator sequences, end-to-end runs
# ---------------------------------------------------------------------------------------
def _spd(rs, d):
    a = rs.randn(d, d) / np.sqrt(d)
    return a @ a.T + 0.5 * np.eye(d)


def ar1_cov(d, rho):
    idx = np.arange(d)
    return rho ** np.abs(idx[:, None] - idx[None, :])


def capture_dense_units():
    from littlemcmc import quadpotential as rq

    out = {}
    cases = [("full", 5, "std_normal"), ("full", 12, "ar1"), ("full", 70, "ar1"), ("full", 128, "std_normal"),
             ("inv", 5, "std_normal"), ("inv", 12, "ar1"), ("inv", 70, "ar1")]
    for ci, (kind, d, fam) in enumerate(cases):
        rs = np.random.RandomState(300 + ci)
        mat = _spd(rs, d)                      # covariance for "full", mass (inverse covariance) for "inv"
        pot = rq.quad_potential(mat, kind == "full")
        f = targets.make(fam, d)
        step = ref.HamiltonianMC(f, d, potential=pot)
        np.random.seed(4000 + ci)
        draws = np.array([pot.random() for _ in range(3)])
        q0 = 0.3 * np.random.randn(d)
        p0 = pot.random()
        x = np.random.randn(d)
        eps, n = 0.05, (12 if d <= 16 else 4)
        st = step.integrator.compute_state(q0, p0)
        states = [st]
        for _ in range(n):
            st = step.integrator.step(eps, st)
            states.append(st)
        for _ in range(n):
            st = step.integrator.step(-eps, st)
            states.append(st)
        key = "c%d_" % ci
        out[key + "kind"] = np.array(kind)
        out[key + "family"] = np.array(fam)
        out[key + "d"] = np.array(d)
        out[key + "matrix"] = mat
        out[key + "seed"] = np.array(4000 + ci)
        out[key + "random"] = draws.astype("d")
        out[key + "random_dtype"] = np.array(str(draws.dtype))
        out[key + "x"] = x
        out[key + "velocity"] = np.asarray(pot.velocity(x), dtype="d")
        out[key + "energy_x"] = np.array(float(pot.energy(x)))
        out[key + "eps"] = np.array(eps)
        out[key + "n"] = np.array(n)
        out[key + "params"] = f.params()
        for k, v in state_rows(states).items():
            out[key + k] = v
    out["n_cases"] = np.array(len(cases))
    save("dense_units", **out)


def capture_dense_adapt():
    from littlemcmc.quadpotential import QuadPotentialFullAdapt

    out = {}
    rs = np.random.RandomState(77)
    d = 6
    true_cov = _spd(rs, d) * 3.0
    samples = rs.multivariate_normal(np.linspace(-1, 1, d), true_cov, size=140)
    for name, kw in [("w20", dict(adaptation_window=20)), ("w15u4", dict(adaptation_window=15, update_window=4))]:
        pot = QuadPotentialFullAdapt(d, samples[0] * 0.0 + 0.25, np.eye(d), 10, **kw)
        rows = {k: [] for k in ("cov", "chol", "fmean", "fraw", "fn", "bmean", "braw", "bn", "window", "prev", "ns")}
        for x in samples:
            pot.update(x, None, True)
            rows["cov"].append(np.array(pot._cov, dtype="d"))
            rows["chol"].append(np.array(pot._chol, dtype="d"))
            rows["fmean"].append(pot._foreground_cov.mean.copy())
            rows["fraw"].append(pot._foreground_cov.raw_cov.copy())
            rows["fn"].append(pot._foreground_cov.n_samples)
            rows["bmean"].append(pot._background_cov.mean.copy())
            rows["braw"].append(pot._background_cov.raw_cov.copy())
            rows["bn"].append(pot._background_cov.n_samples)
            rows["window"].append(pot._adaptation_window)
            rows["prev"].append(pot._previous_update)
            rows["ns"].append(pot._n_samples)
        for k, v in rows.items():
            out[name + "_" + k] = np.array(v)
        out[name + "_kw_names"] = np.array(list(kw.keys()), dtype="U32")
        out[name + "_kw_vals"] = np.array(list(kw.values()), dtype="d")
    out["samples"] = samples
    out["initial_mean"] = samples[0] * 0.0 + 0.25
    # singular estimate (tests/test_quadpotential.py:215-224): the factor must survive, the error must be recorded
    pot = QuadPotentialFullAdapt(2, np.zeros(2), np.eye(2), 0, adaptation_window=10)
    for _ in range(11):
        pot.update(np.ones(2), None, True)
    out["singular_cov"] = np.array(pot._cov, dtype="d")
    out["singular_chol"] = np.array(pot._chol, dtype="d")
    out["singular_failed"] = np.array(pot._chol_error is not None)
    save("dense_adapt", **out)


def capture_dense_e2e():
    from littlemcmc import quadpotential as rq

    runs = [
        # name, family, d, kind, potential, chains, tune, draws
        ("e2e_nuts_full_ar1_12", "ar1", 12, "nuts", "full", 2, 200, 100),
        ("e2e_nuts_fullinv_ar1_12", "ar1", 12, "nuts", "inv", 2, 200, 100),
        ("e2e_hmc_full_std10", "std_normal", 10, "hmc", "full", 2, 200, 100),
        ("e2e_nuts_adaptfull_ar1_10_a", "ar1", 10, "nuts", "adapt_full", 1, 260, 100),
        ("e2e_nuts_adaptfull_ar1_10_b", "ar1", 10, "nuts", "jitter+adapt_full", 1, 260, 100),
        ("e2e_nuts_adaptfull_std70", "std_normal", 70, "nuts", "jitter+adapt_full", 1, 120, 30),
    ]
    for ri, (name, fam, d, kind, potk, chains, tune, draws) in enumerate(runs):
        f = targets.make(fam, d)
        rho = 0.9
        extra = {}
        if potk in ("full", "inv"):
            cov = ar1_cov(d, rho) if fam == "ar1" else _spd(np.random.RandomState(9), d)
            mat = cov if potk == "full" else np.linalg.inv(cov)
            pot = rq.quad_potential(mat, potk == "full")
            cls = ref.HamiltonianMC if kind == "hmc" else ref.NUTS
            step = cls(f, d, potential=pot)
            extra["matrix"] = mat
            trace, stats = ref.sample(f, d, draws=draws, tune=tune, step=step, chains=chains, cores=1,
                                      progressbar=False, random_seed=SEED + ri, discard_tuned_samples=False)
            np.random.seed(SEED + ri)
            seeds = np.array([np.random.randint(2 ** 30) for _ in range(chains)])
        else:
            import littlemcmc.sampling as ref_sampling
            made = {}
            orig = ref_sampling.init_nuts

            def spy(*a, **k):
                st, sp = orig(*a, **k)
                made["step"] = sp
                return st, sp

            ref_sampling.init_nuts = spy
            seeds = np.array([SEED % 100000 + 31 * ri])
            try:
                trace, stats = ref.sample(f, d, draws=draws, tune=tune, chains=1, cores=1, init=potk,
                                          progressbar=False, random_seed=[int(seeds[0])],
                                          discard_tuned_samples=False)
            finally:
                ref_sampling.init_nuts = orig
            step = made["step"]
            extra["final_cov"] = np.array(step.potential._cov, dtype="d")
            extra["final_chol"] = np.array(step.potential._chol, dtype="d")
            extra["final_window"] = np.array(step.potential._adaptation_window)
            extra["final_prev"] = np.array(step.potential._previous_update)
        arrays = dict(
            family=np.array(fam), kind=np.array(kind), potential=np.array(potk), d=np.array(d),
            chains=np.array(chains), tune=np.array(tune), draws=np.array(draws), seeds=seeds,
            random_seed=np.array(SEED + ri), params=f.params(), trace=trace,
            final_da=np.array([float(np.ravel(x)[0]) for x in (
                step.step_adapt._log_step, step.step_adapt._log_bar, step.step_adapt._hbar,
                step.step_adapt._count)]),
        )
        arrays.update(extra)
        for k, v in stats.items():
            arrays["stat_" + k] = v
        save(name, **arrays)


# ---------------------------------------------------------------------------------------
# 7b. QuadPotentialFull(cov, dtype="float64") (quadpotential.py:431-444: the dtype argument): float64 covariance,
#     float64 momentum by a float64 triangular solve. Unit values of the potential + one end-to-end NUTS run.
# ---------------------------------------------------------------------------------------
def capture_dense_full64():
    from littlemcmc import quadpotential as rq

    fam, d, chains, tune, draws = "ar1", 12, 2, 200, 100
    f = targets.make(fam, d)
    cov = ar1_cov(d, 0.9)
    pot = rq.QuadPotentialFull(cov, dtype="float64")
    rs = np.random.RandomState(77)
    x = rs.randn(d)
    np.random.seed(4711)
    rnd = np.array([pot.random() for _ in range(3)])
    unit = dict(unit_x=x, unit_velocity=pot.velocity(x), unit_energy=np.array(pot.energy(x)), unit_random=rnd,
                unit_random_seed=np.array(4711), unit_random_dtype=np.array(str(rnd.dtype)), unit_chol=np.array(pot._chol, dtype="d"))
    step = ref.NUTS(f, d, potential=pot)
    trace, stats = ref.sample(f, d, draws=draws, tune=tune, step=step, chains=chains, cores=1, progressbar=False,
                              random_seed=SEED + 41, discard_tuned_samples=False)
    np.random.seed(SEED + 41)
    seeds = np.array([np.random.randint(2 ** 30) for _ in range(chains)])
    arrays = dict(family=np.array(fam), kind=np.array("nuts"), potential=np.array("full64"), d=np.array(d),
                  chains=np.array(chains), tune=np.array(tune), draws=np.array(draws), seeds=seeds,
                  random_seed=np.array(SEED + 41), params=f.params(), trace=trace, matrix=cov,
                  final_da=np.array([float(np.ravel(v)[0]) for v in (step.step_adapt._log_step, step.step_adapt._log_bar,
                                                                     step.step_adapt._hbar, step.step_adapt._count)]))
    arrays.update(unit)
    for k, v in stats.items():
        arrays["stat_" + k] = v
    save("e2e_nuts_full64_ar1_12", **arrays)


# ---------------------------------------------------------------------------------------
# 8. QuadPotentialDiagAdapt with a growing adaptation window (adaptation_window_multiplier != 1)
# ---------------------------------------------------------------------------------------
def capture_diag_window_multiplier():
    out = {}
    rs = np.random.RandomState(21)
    d = 6
    samples = rs.randn(150, d) * np.linspace(0.3, 4.0, d) + 1.0
    pot = QuadPotentialDiagAdapt(d, np.full(d, 0.5), np.ones(d), 10, adaptation_window=15, adaptation_window_multiplier=2)
    rows = {k: [] for k in ("var", "ns", "window", "fw", "bw")}
    for x in samples:
        pot.update(x, None, True)
        rows["var"].append(np.array(pot._var, dtype="d"))
        rows["ns"].append(pot._n_samples)
        rows["window"].append(pot.adaptation_window)
        rows["fw"].append(pot._foreground_var.w_sum)
        rows["bw"].append(pot._background_var.w_sum)
    for k, v in rows.items():
        out["seq_" + k] = np.array(v)
    out["samples"] = samples
    # one chain end to end (one chain: the reference's reset() keeps the grown window for the next chain)
    d2, tune, draws, seed = 8, 220, 40, 424242
    f = targets.make("ar1", d2)
    np.random.seed(seed)
    start = 2 * np.random.rand(d2) - 1
    pot2 = QuadPotentialDiagAdapt(d2, start, np.ones(d2), 10, adaptation_window=20, adaptation_window_multiplier=2)
    step = ref.NUTS(f, d2, potential=pot2)
    trace, stats = ref.sample(f, d2, draws=draws, tune=tune, step=step, start=start, chains=1, cores=1, progressbar=False,
                              random_seed=[seed], discard_tuned_samples=False)
    out.update(e2e_d=np.array(d2), e2e_tune=np.array(tune), e2e_draws=np.array(draws), e2e_seed=np.array(seed),
               e2e_start=start, e2e_trace=trace, e2e_final_window=np.array(pot2.adaptation_window),
               e2e_final_var=np.array(pot2._var, dtype="d"))
    for k, v in stats.items():
        out["e2e_stat_" + k] = v
    save("diag_window_multiplier", **out)


# ---------------------------------------------------------------------------------------
# step_rand (base_hmc.py:46,123,154-155): the per-iteration step-size jitter, in the one form that consumes the
# chain's own stream in a fixed way -- lambda s: s * np.random.uniform(lo, hi)
# ---------------------------------------------------------------------------------------
def capture_step_rand():
    lo, hi = 0.8, 1.25
    jitter = lambda s: s * np.random.uniform(lo, hi)   # noqa: E731
    out = {"lo": np.array(lo), "hi": np.array(hi), "random_seed": np.array(SEED)}
    # NUTS through the plain API (kwargs reach the NUTS constructor, sampling.py:149-155)
    f = targets.make("ar1", 12)
    trace, stats = ref.sample(f, 12, draws=15, tune=60, chains=2, cores=1, progressbar=False, random_seed=SEED,
                              discard_tuned_samples=False, step_rand=jitter)
    out.update(nuts_family=np.array("ar1"), nuts_d=np.array(12), nuts_chains=np.array(2), nuts_tune=np.array(60),
               nuts_draws=np.array(15), nuts_trace=trace)
    for k, v in stats.items():
        out["nuts_stat_" + k] = v
    # HMC with an explicit step object
    f = targets.make("std_normal", 6)
    step = ref.HamiltonianMC(f, 6, path_length=1.5, step_rand=jitter)
    trace, stats = ref.sample(f, 6, draws=15, tune=60, step=step, chains=2, cores=1, progressbar=False,
                              random_seed=SEED, discard_tuned_samples=False)
    out.update(hmc_family=np.array("std_normal"), hmc_d=np.array(6), hmc_chains=np.array(2), hmc_tune=np.array(60),
               hmc_draws=np.array(15), hmc_path_length=np.array(1.5), hmc_trace=trace)
    for k, v in stats.items():
        out["hmc_stat_" + k] = v
    save("e2e_step_rand", **out)


# ---------------------------------------------------------------------------------------
# round 4: QuadPotentialDiagAdapt(dtype="float64") (quadpotential.py:159,175-184), a deterministic step_rand callable
# (base_hmc.py:154-155), and shapes beyond the fused kernels' vector widths (model_ndim 2000 diagonal, 384 dense)
# ---------------------------------------------------------------------------------------
def capture_diag_float64():
    fam, d, chains, tune, draws = "ar1", 12, 2, 230, 40
    f = targets.make(fam, d)
    rs = np.random.RandomState(5)
    # unit values + an update sequence across a window switch
    pot = QuadPotentialDiagAdapt(d, np.full(d, 0.25), 0.5 + rs.rand(d), 10, adaptation_window=20, dtype="float64")
    x = rs.randn(d)
    np.random.seed(99)
    rnd = np.array([pot.random() for _ in range(3)])
    out = dict(unit_x=x, unit_velocity=pot.velocity(x), unit_energy=np.array(pot.energy(x)), unit_random=rnd,
               unit_random_dtype=np.array(str(rnd.dtype)), unit_random_seed=np.array(99),
               unit_initial_mean=np.full(d, 0.25), unit_initial_diag=np.array(pot._initial_diag, dtype="d"))
    samples = rs.randn(60, d) * np.linspace(0.2, 5.0, d) - 0.5
    seq = []
    for smp in samples:
        pot.update(smp, None, True)
        seq.append(np.array(pot._var, dtype="d"))
    out.update(seq_samples=samples, seq_var=np.array(seq), seq_var_dtype=np.array(str(pot._var.dtype)),
               seq_n_samples=np.array(pot._n_samples))
    # one run end to end
    np.random.seed(SEED + 7)
    seeds = np.array([np.random.randint(2 ** 30) for _ in range(chains)])
    np.random.seed(int(seeds[0]))
    start = 2 * np.random.rand(d) - 1
    pot2 = QuadPotentialDiagAdapt(d, start, np.ones(d), 10, dtype="float64")
    step = ref.NUTS(f, d, potential=pot2)
    trace, stats = ref.sample(f, d, draws=draws, tune=tune, step=step, start=start, chains=chains, cores=1,
                              progressbar=False, random_seed=list(seeds), discard_tuned_samples=False)
    out.update(family=np.array(fam), d=np.array(d), chains=np.array(chains), tune=np.array(tune), draws=np.array(draws),
               seeds=seeds, start=start, params=f.params(), trace=trace, final_var=np.array(pot2._var),
               final_var_dtype=np.array(str(pot2._var.dtype)))
    for k, v in stats.items():
        out["stat_" + k] = v
    save("e2e_nuts_diag64_ar1_12", **out)


def capture_full_adapt_float64():
    """QuadPotentialFullAdapt(dtype="float64") (quadpotential.py:484,497-509): unit values, an update sequence across a window
    switch, one NUTS run with the potential handed to the step."""
    from littlemcmc.quadpotential import QuadPotentialFullAdapt

    fam, d, tune, draws = "ar1", 10, 230, 40
    f = targets.make(fam, d)
    rs = np.random.RandomState(15)
    init_cov = _spd(rs, d)
    pot = QuadPotentialFullAdapt(d, np.full(d, 0.25), init_cov, 10, adaptation_window=20, dtype="float64")
    x = rs.randn(d)
    np.random.seed(199)
    rnd = np.array([pot.random() for _ in range(3)])
    out = dict(unit_x=x, unit_velocity=np.asarray(pot.velocity(x)), unit_energy=np.array(pot.energy(x)), unit_random=rnd,
               unit_random_dtype=np.array(str(rnd.dtype)), unit_random_seed=np.array(199), unit_initial_mean=np.full(d, 0.25),
               unit_initial_cov=init_cov, unit_chol=np.array(pot._chol), unit_chol_dtype=np.array(str(pot._chol.dtype)))
    samples = rs.randn(60, d) @ np.linalg.cholesky(_spd(rs, d)).T * 1.5 - 0.5
    rows = {k: [] for k in ("cov", "chol", "fn", "bn", "window", "prev")}
    for smp in samples:
        pot.update(smp, None, True)
        rows["cov"].append(np.array(pot._cov))
        rows["chol"].append(np.array(pot._chol))
        rows["fn"].append(pot._foreground_cov.n_samples)
        rows["bn"].append(pot._background_cov.n_samples)
        rows["window"].append(pot._adaptation_window)
        rows["prev"].append(pot._previous_update)
    out.update(seq_samples=samples, seq_cov_dtype=np.array(str(pot._cov.dtype)), seq_n_samples=np.array(pot._n_samples),
               **{"seq_" + k: np.array(v) for k, v in rows.items()})
    np.random.seed(SEED + 17)
    seeds = np.array([np.random.randint(2 ** 30)])
    np.random.seed(int(seeds[0]))
    start = 2 * np.random.rand(d) - 1
    pot2 = QuadPotentialFullAdapt(d, start, np.eye(d), 10, dtype="float64")
    step = ref.NUTS(f, d, potential=pot2)
    trace, stats = ref.sample(f, d, draws=draws, tune=tune, step=step, start=start, chains=1, cores=1,
                              progressbar=False, random_seed=list(seeds), discard_tuned_samples=False)
    out.update(family=np.array(fam), d=np.array(d), chains=np.array(1), tune=np.array(tune), draws=np.array(draws),
               seeds=seeds, start=start, params=f.params(), trace=trace, final_cov=np.array(pot2._cov),
               final_chol=np.array(pot2._chol), final_cov_dtype=np.array(str(pot2._cov.dtype)))
    for k, v in stats.items():
        out["stat_" + k] = v
    save("e2e_nuts_adaptfull64_ar1_10", **out)


def capture_step_rand_callable():
    shrink = lambda s: 0.9 * s   # noqa: E731
    f = targets.make("ar1", 12)
    trace, stats = ref.sample(f, 12, draws=15, tune=60, chains=2, cores=1, progressbar=False, random_seed=SEED + 3,
                              discard_tuned_samples=False, step_rand=shrink)
    out = dict(factor=np.array(0.9), random_seed=np.array(SEED + 3), family=np.array("ar1"), d=np.array(12),
               chains=np.array(2), tune=np.array(60), draws=np.array(15), trace=trace)
    for k, v in stats.items():
        out["stat_" + k] = v
    save("e2e_step_rand_callable", **out)


def capture_wide():
    from littlemcmc import quadpotential as rq

    # model_ndim = 2000, diagonal mass adaptation (C4's family at twice its size), one chain through the plain API
    fam, d, tune, draws = "diag_gaussian", 2000, 45, 5
    f = targets.make(fam, d)
    trace, stats = ref.sample(f, d, draws=draws, tune=tune, chains=1, cores=1, progressbar=False, random_seed=SEED + 11,
                              discard_tuned_samples=False)
    np.random.seed(SEED + 11)
    seeds = np.array([np.random.randint(2 ** 30)])
    np.random.seed(int(seeds[0]))
    start = 2 * np.random.rand(d) - 1
    out = dict(family=np.array(fam), d=np.array(d), chains=np.array(1), tune=np.array(tune), draws=np.array(draws),
               random_seed=np.array(SEED + 11), seeds=seeds, start=start, params=f.params(), trace=trace)
    for k, v in stats.items():
        out["stat_" + k] = v
    save("e2e_nuts_diag2000", **out)
    # model_ndim = 384, QuadPotentialFull with the target's covariance (quadpotential.py:430-468)
    fam, d, tune, draws = "ar1", 384, 40, 10
    f = targets.make(fam, d)
    pot = rq.QuadPotentialFull(ar1_cov(d, 0.9))
    step = ref.NUTS(f, d, potential=pot)
    trace, stats = ref.sample(f, d, draws=draws, tune=tune, step=step, chains=1, cores=1, progressbar=False,
                              random_seed=SEED + 12, discard_tuned_samples=False)
    np.random.seed(SEED + 12)
    seeds = np.array([np.random.randint(2 ** 30)])
    out = dict(family=np.array(fam), d=np.array(d), chains=np.array(1), tune=np.array(tune), draws=np.array(draws),
               random_seed=np.array(SEED + 12), seeds=seeds, rho=np.array(0.9), params=f.params(), trace=trace)
    for k, v in stats.items():
        out["stat_" + k] = v
    save("e2e_nuts_full_ar1_384", **out)


# ---------------------------------------------------------------------------------------
# 13. init="adapt_full" with TWO chains through the sequential driver (cores=1): the reference reuses one step object
#     (sampling.py:370-383) and QuadPotentialFullAdapt.reset() is the base class's no-op (quadpotential.py:137-139), so
#     chain 1 starts from the matrix, estimators and grown window chain 0 ended with. Captured next to it: the reference's
#     ONE-chain run with chain 1's seed -- what its multi-process driver would compute for chain 1, and what the device
#     computes (every chain fresh). The pair pins the difference instead of describing it.
# ---------------------------------------------------------------------------------------
def capture_full_adapt_two_chains():
    fam, d, tune, draws = "ar1", 6, 150, 50
    f = targets.make(fam, d)
    np.random.seed(SEED + 77)
    seeds = [int(np.random.randint(2 ** 30)) for _ in range(2)]
    out = dict(family=np.array(fam), d=np.array(d), chains=np.array(2), tune=np.array(tune), draws=np.array(draws),
               seeds=np.array(seeds), init=np.array("adapt_full"), params=f.params())
    trace, stats = ref.sample(f, d, draws=draws, tune=tune, chains=2, cores=1, init="adapt_full", progressbar=False,
                              random_seed=list(seeds), discard_tuned_samples=False)
    out["trace"] = trace
    for k, v in stats.items():
        out["stat_" + k] = v
    trace1, stats1 = ref.sample(f, d, draws=draws, tune=tune, chains=1, cores=1, init="adapt_full", progressbar=False,
                                random_seed=[seeds[1]], discard_tuned_samples=False)
    out["solo1_trace"] = trace1
    for k, v in stats1.items():
        out["solo1_stat_" + k] = v
    assert not np.allclose(trace[1], trace1[0]), "the carry-over should show"
    save("e2e_adaptfull_two_chains", **out)


CAPTURES = {"full_adapt_two_chains": capture_full_adapt_two_chains, "leapfrog": capture_leapfrog, "transitions": capture_transitions, "adapt": capture_adapt,
            "e2e": capture_e2e, "seeds": capture_seeds, "dense_units": capture_dense_units,
            "dense_adapt": capture_dense_adapt, "dense_e2e": capture_dense_e2e, "dense_full64": capture_dense_full64,
            "diag_window_multiplier": capture_diag_window_multiplier, "step_rand": capture_step_rand,
            "diag_float64": capture_diag_float64, "full_adapt_float64": capture_full_adapt_float64, "step_rand_callable": capture_step_rand_callable, "wide": capture_wide}

if __name__ == "__main__":
    # capture.py [group | e2e:<name>[,<name>...]] ...   (no argument: everything)
    for which in (sys.argv[1:] or list(CAPTURES)):
        if which.startswith("e2e:"):
            capture_e2e(only=which[4:].split(","))
        else:
            CAPTURES[which]()
