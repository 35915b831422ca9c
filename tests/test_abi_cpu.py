"""No-GPU checks of the drop-in boundary: the C-ABI library loads and exports every symbol declared in
include/lmc_hip.h, constants agree between header and binding, host-side argument handling mirrors the
reference's errors, and the product path fails loudly (no CPU fallback) when there is no device."""
import ctypes
import os
import re

import numpy as np
import pytest

import littlemcmc_amd as lmc
from littlemcmc_amd import _abi
from littlemcmc_amd import targets as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "lmc_hip.h")).read()


def declared_functions():
    names = re.findall(r"^\s*(?:const\s+char\*|int32_t|int64_t|int|void\*?|void)\s+(lmc_\w+)\s*\(", HEADER, re.M)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_abi.LIB_PATH)
    fns = declared_functions()
    assert len(fns) >= 35
    for name in fns:
        assert hasattr(lib, name), "liblmc_hip.so does not export %s" % name
    assert sorted(_abi.EXPORTED_SYMBOLS) == fns, set(fns) ^ set(_abi.EXPORTED_SYMBOLS)


def test_header_constants_match_binding():
    consts = dict(re.findall(r"#define\s+(LMC_\w+)\s+(\d+)", HEADER))
    assert int(consts["LMC_ABI_VERSION"]) == _abi.ABI_VERSION == _abi.load().lmc_abi_version()
    for name in ("KIND_NUTS", "KIND_HMC", "POT_DIAG_ADAPT", "POT_DIAG", "TARGET_STD_NORMAL", "TARGET_DIAG_GAUSSIAN",
                 "TARGET_AR1", "TARGET_FUNNEL", "TARGET_NORMAL1D", "TARGET_USER", "STAT_STEP_SIZE", "STAT_ACCEPT",
                 "STAT_MODEL_LOGP", "STAT_DEPTH", "STAT_TREE_SIZE", "STAT_DIVERGING", "STAT_TUNE", "STAT_ACCEPTED",
                 "CT_LEAPFROGS", "NUM_COUNTERS", "SDOT_NATIVE", "SDOT_OPENBLAS_SKYLAKEX", "SDOT_OPENBLAS_HASWELL",
                 "STATUS_BAD_INITIAL_ENERGY", "LDS_PLAN_AUTO", "LDS_PLAN_SHALLOW", "LDS_PLAN_DEEP", "PLANE_F64", "PLANE_I32",
                 "PLANE_U8", "AS_NATIVE", "AS_F64", "AS_I64", "MAX_PLANES", "MAX_RUN_STREAMS"):
        assert int(consts["LMC_" + name]) == getattr(_abi, name), name


def test_config_struct_layout_and_defaults():
    lib = _abi.load()
    cfg = _abi.Config()
    lib.lmc_config_defaults(ctypes.byref(cfg), 7, 13)
    assert (cfg.chains, cfg.dim, cfg.abi_version) == (7, 13, _abi.ABI_VERSION)
    # reference defaults: nuts.py:110-120, hmc.py:67-68, quadpotential.py:156
    assert (cfg.target_accept, cfg.emax, cfg.step_scale, cfg.gamma, cfg.k, cfg.t0) == (0.8, 1000.0, 0.25, 0.05, 0.75, 10.0)
    assert (cfg.max_treedepth, cfg.early_max_treedepth, cfg.max_steps, cfg.adaptation_window) == (10, 8, 1024, 101)
    assert cfg.path_length == 2.0 and cfg.start_energy_sdot == _abi.SDOT_OPENBLAS_SKYLAKEX
    block = HEADER[HEADER.index("typedef struct lmc_config"):HEADER.index("} lmc_config;")]
    fields = re.findall(r"^\s+(?:int32_t|double|lmc_tuning)\s+(\w+);", block, re.M)
    assert [f for f, _ in _abi.Config._fields_] == fields and fields[-1] == "tuning"
    tblock = HEADER[HEADER.index("typedef struct lmc_tuning"):HEADER.index("} lmc_tuning;")]
    tfields = re.findall(r"^\s+int32_t\s+([\w, ]+?)(?:\[\d+\])?;", tblock, re.M)
    tfields = [x.strip() for grp in tfields for x in grp.split(",")]
    assert [f for f, _ in _abi.Tuning._fields_] == tfields and ctypes.sizeof(_abi.Tuning) == 48
    assert all(getattr(cfg.tuning, f) == 0 for f, _ in _abi.Tuning._fields_[:-1])   # defaults: the engine decides everything
    # no entry point reads the environment any more: the knobs are fields, the HOST maps the LMC_* variables onto them
    from littlemcmc_amd.engine import tuning_from_env

    assert tuning_from_env({"LMC_SUB_BLOCKS": "2", "LMC_FORCE_WIDE": "1", "LMC_RUN_SHAPE": "2,2", "LMC_DENSE_COOP": "0",
                            "LMC_DENSE_LDS_SLOTS": "0", "LMC_WIDE_TEAM": "", "LMC_CHOL_HBM": "x"}) == {
        "sub_blocks": 2, "force_general": 1, "run_ns": 2, "run_w": 2, "dense_coop_off": 1, "dense_lds_slots_p1": 1}
    for unit in ("lmc_engine.hip", "lmc_dense.hip", "lmc_wide.hip", "lmc_tick.hip", "lmc_dense_coop.hip", "lmc_diag.hip"):
        assert "getenv" not in open(os.path.join(ROOT, "littlemcmc_amd", "csrc", unit)).read(), unit
    assert cfg.lds_plan == _abi.LDS_PLAN_AUTO and cfg.reserved0 == 0
    bad = _abi.Config()
    lib.lmc_config_defaults(ctypes.byref(bad), 7, 13)
    bad.lds_plan = 3                       # validated before any HIP call: no GPU needed to see the refusal
    h = ctypes.c_void_p()
    assert lib.lmc_engine_create(ctypes.byref(bad), ctypes.byref(h)) == 1 and b"lds_plan" in lib.lmc_last_error(None)
    bad.lds_plan = 0
    bad.tuning.reserved[1] = 7
    assert lib.lmc_engine_create(ctypes.byref(bad), ctypes.byref(h)) == 1 and b"reserved" in lib.lmc_last_error(None)


def test_window_destination_struct_layout():
    """struct lmc_window_dst / lmc_window_plane (ABI 8): the ctypes mirror has the header's fields in the header's order and
    the C compiler's size (8-byte alignment, 16 planes)."""
    blk = HEADER[HEADER.index("typedef struct lmc_window_plane"):HEADER.index("} lmc_window_dst;")]
    names = re.findall(r"^\s+(?:int32_t|int64_t|void\*|double\*|lmc_window_plane)\s+(\w+)", blk, re.M)
    assert names == ["dst", "kind", "idx", "as", "reserved", "n_out", "first", "trace", "n_planes", "copy_workgroups", "plane"]
    assert [f.rstrip("_") for f, _ in _abi.WindowPlane._fields_] == names[:5]
    assert [f for f, _ in _abi.WindowDst._fields_] == names[5:]
    assert ctypes.sizeof(_abi.WindowPlane) == 24 and ctypes.sizeof(_abi.WindowDst) == 32 + 16 * 24


def test_built_in_targets_and_the_run_time_user_family():
    lib = _abi.load()
    # 0..4 compiled in; 5 = LMC_TARGET_USER: its kernels are loaded at run time (lmc_engine_load_user_kernels);
    # 6 = LMC_TARGET_EXTERNAL (ticks); nothing beyond
    assert [lib.lmc_has_target(i) for i in range(8)] == [1, 1, 1, 1, 1, 1, 1, 0]


def test_user_target_compiles_with_hiprtc_without_a_gpu():
    """The run-time compiler needs no device: the functor-dependent kernels of an engine shape (three, and the sampling kernel's second LDS plan) come out of
    hiprtc as a gfx950 code object with the lowered names the engine looks up; the result is cached by content."""
    from littlemcmc_amd.targets import UserTarget

    t = UserTarget.separable(70, logp="-0.5*q*q", grad="-q")
    assert t.jit == "hiprtc" and t.lib_path is None
    code, run, traj, logp, run1 = t.kernels_for(2, 2, 1)
    assert code[:4] == b"\x7fELF" and len(code) > 10000
    assert "run_kernel" in run and "UserTarget" in run and "trajectory_kernel" in traj and "logp_kernel" in logp
    assert run1 is not None and "run_kernel" in run1 and run1 != run   # one-wave shape: also the deep-tree LDS plan (ABI 7)
    assert UserTarget.separable(600, logp="-0.5*q*q", grad="-q").kernels_for(4, 4, 4)[4] is None   # teams: plan 0 only
    assert t.kernels_for(2, 2, 1)[0] is code                      # in-memory cache
    assert UserTarget.separable(70, logp="-0.5*q*q", grad="-q").kernels_for(2, 2, 1)[0] == code   # disk cache
    with pytest.raises(RuntimeError, match="hiprtc compile failed"):
        UserTarget(3, "namespace lmc { this is not C++ }").kernels_for(1, 1, 1)
    with pytest.raises(ValueError):
        UserTarget(3, "", jit="nvcc")


def test_no_cpu_fallback_without_device():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(_abi.HipLibraryError, match="no HIP device"):
        lmc.Engine(T.StdNormal(4), chains=2)
    with pytest.raises(_abi.HipLibraryError):
        lmc.sample(T.StdNormal(4), 4, draws=3, tune=3, chains=2, random_seed=1)


def test_plain_python_callable_is_wrapped_not_evaluated_on_a_cpu_sampler():
    """The reference's plug-in signature (integration.py:40,62,115) is accepted: a per-point callable becomes a
    CallableTarget (sampler in the HIP tick kernel, density evaluated by the caller's code). Non-callables and
    dimension mismatches are rejected; without a GPU the engine still fails loudly -- there is no CPU sampler."""
    from littlemcmc_amd.targets import CallableTarget, require_device_target

    f = lambda q: (-0.5 * np.dot(q, q), -q)   # noqa: E731
    t = require_device_target(f, 3)
    assert isinstance(t, CallableTarget) and t.family == lmc._abi.TARGET_EXTERNAL and t.d == 3 and t.tick_poll == 1
    lp, g = t(np.array([1.0, 2.0, 3.0]))
    assert lp == -7.0 and np.array_equal(g, [-1.0, -2.0, -3.0])
    step = lmc.NUTS(f, 3)
    assert isinstance(step._logp_dlogp_func, CallableTarget)
    with pytest.raises(TypeError, match="DeviceTarget"):
        lmc.NUTS("not callable", 3)
    with pytest.raises(TypeError, match="DeviceTarget"):
        require_device_target(f)                      # a bare callable needs model_ndim
    with pytest.raises(ValueError, match="model_ndim"):
        lmc.NUTS(T.StdNormal(4), 5)
    import torch

    if not torch.cuda.is_available():
        with pytest.raises(_abi.HipLibraryError):
            lmc.sample(f, 3, draws=2, tune=2, chains=2, random_seed=1)


def test_reference_argument_errors():
    tgt = T.StdNormal(3)
    with pytest.raises(ValueError, match="both"):     # base_hmc.py:115-116
        lmc.NUTS(tgt, 3, scaling=np.ones(3), potential=lmc.QuadPotentialDiag(np.ones(3)))
    with pytest.raises(TypeError, match="string"):    # sampling.py:563-564
        lmc.init_nuts(tgt, 3, init=3)
    with pytest.raises(ValueError, match="Unknown initializer"):   # sampling.py:599
        lmc.init_nuts(tgt, 3, init="nope")
    start, step = lmc.init_nuts(tgt, 3, init="adapt_full")   # sampling.py:588-592 (host objects only: no GPU needed)
    assert isinstance(step.potential, lmc.QuadPotentialFullAdapt) and not start.any()
    with pytest.raises(NotImplementedError):          # per-chain adapted dense matrices: the fused kernels, up to 256 dimensions
        lmc.QuadPotentialFullAdapt(1025, np.zeros(1025), None, 1)
    with pytest.raises(NotImplementedError):          # shared dense matrices: the general kernels take over up to 2048
        lmc.QuadPotentialFull(np.eye(2049))
    assert lmc.QuadPotentialFull(np.eye(300))._n == 300
    assert lmc.QuadPotentialDiagAdapt(3, np.zeros(3), np.ones(3), 1, dtype="float64").dtype == "float64"   # quadpotential.py:159
    with pytest.raises(NotImplementedError):
        lmc.QuadPotentialDiagAdapt(3, np.zeros(3), np.ones(3), 1, dtype="float16")
    with pytest.raises(ValueError, match="two-dimensional"):   # quadpotential.py:485-486
        lmc.QuadPotentialFullAdapt(3, np.zeros(3), np.ones(3), 1)
    from littlemcmc_amd.quadpotential import PositiveDefiniteError

    with pytest.raises(PositiveDefiniteError):        # tests/test_quadpotential.py:21-24
        lmc.quad_potential(np.array([0, 2, 3]), True)
    with pytest.raises(TypeError):                    # sampling.py:138
        from littlemcmc_amd.sampling import _derive_seeds
        _derive_seeds(1.5, 2)
    import scipy.sparse as sp
    from littlemcmc_amd.quadpotential import isquadpotential

    with pytest.raises(ValueError, match="Sparse precision"):   # quadpotential.py:49-53
        lmc.quad_potential(sp.identity(3, format="csr"), False)
    pot = lmc.quad_potential(sp.identity(3, format="csr") * 2.0, True)   # a sparse covariance is densified
    assert isinstance(pot, lmc.QuadPotentialFull) and isquadpotential(pot) and not isquadpotential(np.ones(3))


def test_seed_derivation_and_start_match_golden(golden_dir):
    from littlemcmc_amd.sampling import _derive_seeds

    g = np.load(os.path.join(golden_dir, "seeds.npz"))
    for chains in (2, 4, 64):
        seeds = _derive_seeds(20260928, chains)
        np.testing.assert_array_equal(seeds, g["seeds_%d" % chains])
        start, step = lmc.init_nuts(T.StdNormal(7), 7, random_seed=seeds)
        np.testing.assert_array_equal(start, g["jitter_%d" % chains])
        assert isinstance(step, lmc.NUTS) and isinstance(step.potential, lmc.QuadPotentialDiagAdapt)
        assert step.potential._initial_weight == 10 and step.step_size == 0.25 / 7 ** 0.25


def test_stats_dtypes_match_reference_tables():
    # nuts.py:87-101, hmc.py:36-50
    assert list(lmc.NUTS.stats_dtypes[0]) == ["depth", "step_size", "tune", "mean_tree_accept", "step_size_bar",
                                              "tree_size", "diverging", "energy_error", "energy",
                                              "max_energy_error", "model_logp"]
    assert lmc.NUTS.stats_dtypes[0]["depth"] == np.int64 and lmc.NUTS.stats_dtypes[0]["tree_size"] == np.float64
    assert list(lmc.HamiltonianMC.stats_dtypes[0]) == ["step_size", "n_steps", "tune", "step_size_bar", "accept",
                                                       "diverging", "energy_error", "energy", "path_length",
                                                       "accepted", "model_logp"]


def test_blas_probe_emulation_matches_numpy_here():
    """The float32 dot emulation used to pick the device's start-energy rounding is exact on the capture host."""
    from littlemcmc_amd._blas_probe import detect_sdot_mode, emulate_sdot

    mode = detect_sdot_mode()
    rs = np.random.RandomState(0)
    hits = total = 0
    for n in (1, 5, 31, 32, 33, 64, 100, 128, 257, 1000):
        x, y = rs.randn(n).astype("f4"), rs.randn(n).astype("f4")
        hits += int(emulate_sdot(x, y, mode) == np.dot(x, y))
        total += 1
    if mode == _abi.SDOT_OPENBLAS_SKYLAKEX and hits < total:
        pytest.skip("host BLAS is not one of the two restated OpenBLAS kernels")
    assert hits == total


def test_bench_and_entry_modules_import_without_gpu():
    import importlib

    bench = importlib.import_module("bench")
    assert bench.usable_cores() >= 1 and bench.HBM_PEAK == 8.0e12
    entry = importlib.import_module("__graft_entry__")
    assert callable(entry.build) and callable(entry.smoke)


def test_torch_target_host_contract_without_a_gpu():
    """TorchTarget is a host object around a callable: construction, shape checking and the refusal of CPU tensors
    need no GPU (the sampler itself does)."""
    import torch

    from littlemcmc_amd.targets import TorchTarget, require_device_target

    t = TorchTarget(3, lambda q: (-0.5 * (q * q).sum(dim=1), -q))
    assert t.family == lmc._abi.TARGET_EXTERNAL and t.d == 3 and not t.graph
    assert require_device_target(t, 3) is t
    with pytest.raises(TypeError, match="no CPU path"):
        t.evaluate(torch.zeros(2, 3, dtype=torch.float64))
    bad = TorchTarget(3, lambda q: (q.sum(dim=1), q[:, :2]))
    with pytest.raises(ValueError, match="must return"):
        bad.evaluate(torch.zeros(2, 3, dtype=torch.float64))
    with pytest.raises(TypeError):
        TorchTarget(3, None)
    auto = TorchTarget.from_logp(3, lambda q: -0.5 * (q * q).sum(dim=1), graph=True)
    assert auto.graph
    lp, g = auto.fn(torch.ones(2, 3, dtype=torch.float64))          # the autograd wrapper itself is device agnostic
    assert torch.allclose(lp, torch.full((2,), -1.5, dtype=torch.float64)) and torch.allclose(g, -torch.ones(2, 3, dtype=torch.float64))


def test_the_tick_state_machine_has_one_statement():
    """Round 5: the tick state machine (externally evaluated densities) exists ONCE -- csrc/lmc_tick.hpp: tick_step,
    parameterised by a Shape policy (one wavefront / the 16-wavefront team) and a Mass policy (diagonal / dense). Until
    round 4 lmc_tick_dense.hpp and lmc_tick_wide.hpp were text-substituted copies produced by tools/gen_tick_*.py; they
    must not come back, and the three kernels must be instantiations of the one function."""
    import os
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "littlemcmc_amd", "csrc")
    for gone in ("lmc_tick_dense.hpp", "lmc_tick_wide.hpp"):
        assert not os.path.exists(os.path.join(csrc, gone)), gone
    for gone in ("gen_tick_dense.py", "gen_tick_wide.py"):
        assert not os.path.exists(os.path.join(root, "tools", gone)), gone
    text = {f: open(os.path.join(csrc, f)).read() for f in ("lmc_tick.hpp", "lmc_dense.hip", "lmc_wide.hip")}
    assert len(re.findall(r"void tick_step\(", text["lmc_tick.hpp"])) == 1
    for f, kernel in (("lmc_tick.hpp", "tick_kernel"), ("lmc_dense.hip", "tick_dense_kernel"), ("lmc_wide.hip", "tick_wide_kernel")):
        body = text[f][text[f].index("void %s(" % kernel):]
        assert "tick_step<NS>(" in body[:1200], kernel
    # the NUTS statements of the machine live in lmc_tick.hpp only
    for f in ("lmc_dense.hip", "lmc_wide.hip"):
        assert "subtree_done" not in text[f] and "begin_doubling" not in text[f]
