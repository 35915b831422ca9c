"""Thin object wrapper over the C ABI handle (include/lmc_hip.h): one Engine == one GPU == one
block of chains, each chain a wavefront. Host arrays in, host arrays out; the zero-copy device
pointers are exposed for consumers that keep results in HBM (bench.py, diagnostics). EngineGroup (below) is several
engines -- one per GPU -- behind the same methods: what sample(..., devices=[...]) drives."""
import ctypes as C

import numpy as np

from . import _abi

# Rounding of the float32 start-state kinetic energy: "auto" = match this host's numpy/BLAS
# (_blas_probe.py), or "skylakex" / "haswell" / "native" (include/lmc_hip.h: LMC_SDOT_*).
DEFAULT_SDOT = "auto"


class _TickView:
    """__cuda_array_interface__ shim over the engine's evaluation-point buffer (float64 [chains, dim])."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


class _PinnedBlock:
    """Owner of one page-locked block behind a result array: a numpy allocation registered with the HIP runtime
    (lmc_host_register). Exposes the memory through __array_interface__ (numpy keeps this object alive as the base of every
    view) and unregisters it when the last view is gone."""

    def __init__(self, lib, raw, base, span, shape, dtype):
        self._lib, self._raw, self._base, self._span, self._registered = lib, raw, int(base), int(span), False
        self.__array_interface__ = {"data": (self._base, False), "shape": tuple(int(x) for x in shape),
                                    "typestr": np.dtype(dtype).str, "version": 3}

    def register(self):
        """Page-lock the block and map it into the GPUs (lmc_host_register). Called from the thread that drives the engine:
        the allocation and the page faults may come from a helper thread, the HIP call does not (the runtime is entered from
        one thread at a time by this package)."""
        if not self._registered:
            if self._lib.lmc_host_register(C.c_void_p(self._base), self._span) != _abi.OK:
                raise _abi.HipLibraryError("cannot pin %d bytes of host memory: %s" % (
                    self._span, (self._lib.lmc_last_error(None) or b"?").decode()))
            self._registered = True

    def __del__(self):
        try:
            if self._registered:
                self._lib.lmc_host_unregister(C.c_void_p(self._base))
                self._registered = False
            self._raw = None
        except Exception:
            pass


def _prefault(address, nbytes, threads):
    """Touch [address, address + nbytes) from several threads (C memset outside the GIL, 2 MiB-aligned shares): a fresh
    allocation costs the kernel one zeroed page per fault, which one thread does at ~16 GiB/s and sixteen at ~170 GiB/s."""
    import threading

    if nbytes <= 0:
        return
    threads = max(1, min(int(threads), nbytes >> 24))          # at least 16 MiB per thread
    share = -(-nbytes // threads // (1 << 21)) * (1 << 21)
    if threads == 1:
        C.memset(address, 0, nbytes)
        return
    work = [threading.Thread(target=C.memset, args=(address + o, 0, min(share, nbytes - o))) for o in range(0, nbytes, share)]
    for t in work:
        t.start()
    for t in work:
        t.join()


def pinned_empty(shape, dtype, lib=None, threads=None, register=True):
    """A C-contiguous numpy array in page-locked host memory every GPU can write (hipHostRegister, portable + mapped), unpinned
    and freed when the array and every view of it are gone. The memory is an ordinary numpy allocation (numpy asks for
    transparent huge pages for large blocks), pre-faulted from ``threads`` threads (default: up to 16 of the usable cores) and
    then registered: 15.6 GiB take 0.12 s this way against 1.2 s for hipHostMalloc, which faults every page in from one thread
    (profiles/r06_sample_e2e.txt). Raises HipLibraryError if the memory cannot be pinned.
    ``register=False`` stops before the HIP call (allocate + pre-fault only: safe in a helper thread); the caller registers
    later with ``pinned_register(array)`` from the thread that drives the engine."""
    lib = lib or _abi.load()
    dtype = np.dtype(dtype)
    nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    if nbytes == 0:
        return np.empty(shape, dtype=dtype)
    if threads is None:
        import os

        threads = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    # Whole pages of its own: pinning works on pages, and a small numpy allocation shares its pages with whatever the allocator
    # put next to it -- registering two neighbours pins a page twice, and unregistering one of them unpins it under the other,
    # which the device is still writing (a GPU memory fault that killed the test process once in two runs of the suite,
    # round 6). The block is over-allocated by a page and the page-aligned interior, rounded up to whole pages, is what is
    # touched, registered and handed out.
    import mmap

    page = mmap.PAGESIZE
    span = -(-nbytes // page) * page
    raw = np.empty(span + page, dtype=np.uint8)
    base = raw.ctypes.data + (-raw.ctypes.data) % page
    _prefault(base, span, threads)
    block = _PinnedBlock(lib, raw, base, span, shape, dtype)
    if register:
        block.register()
    return np.asarray(block)


def pinned_register(array):
    """Register (page-lock + map into the GPUs) an array made by pinned_empty(register=False); a no-op for arrays that are
    registered already or own no pinned block (zero-size results)."""
    owner = array
    while owner is not None and not isinstance(owner, _PinnedBlock):
        owner = getattr(owner, "base", None)
    if owner is not None:
        owner.register()


class StreamedResults:
    """The arrays ``sample()`` returns, allocated up front and filled while the job runs (sampling.py:207-222 of the
    reference returns host arrays; Engine.copy_window_async streams every launch's window of iterations into them under the
    next launch): ``trace[chains, n_out, dim]`` float64 and one ``[chains, n_out]`` array per statistic, already in the
    dtype of the reference's stats dict (the device converts), iteration ``first`` in row 0.

    ``planes``: (name, kind, idx, as_, numpy dtype) with kind / as_ the LMC_PLANE_* / LMC_AS_* of include/lmc_hip.h."""

    def __init__(self, chains, n_out, first, dim, planes, keep_trace=True, pinned=True, lib=None, copy_workgroups=0, direct=False,
                 register=True):
        self.direct = bool(direct)      # the sampling kernel writes ``trace`` itself (Engine.attach_trace): windows carry statistics only
        self.copy_workgroups = int(copy_workgroups)     # lmc_window_dst.copy_workgroups (0 = the library's default)
        self.chains, self.n_out, self.first, self.dim = int(chains), int(n_out), int(first), int(dim)
        self.planes = list(planes)
        assert len(self.planes) <= _abi.MAX_PLANES
        # register=False: allocate and pre-fault only (what a helper thread may do); register() later, from the driving thread
        alloc = (lambda sh, dt: pinned_empty(sh, dt, lib, register=register)) if pinned else (lambda sh, dt: np.empty(sh, dtype=dt))
        self.registered = bool(register) or not pinned
        self.pinned = bool(pinned)
        self.trace = alloc((self.chains, self.n_out, self.dim), np.float64) if keep_trace else None
        self.stats = {name: alloc((self.chains, self.n_out), dt) for name, _k, _i, _a, dt in self.planes}

    def register(self):
        """Page-lock every array (HIP calls: from the thread that drives the engine). Raises HipLibraryError on failure."""
        if not self.registered:
            for a in ([] if self.trace is None else [self.trace]) + list(self.stats.values()):
                pinned_register(a)
            self.registered = True

    def window_dst(self, eng, chain_lo=0):
        """struct lmc_window_dst for the engine that owns chains [chain_lo, chain_lo + eng.chains) of these arrays."""
        w = _abi.WindowDst()
        w.n_out, w.first = self.n_out, self.first
        w.trace = None if (self.trace is None or self.direct) else self.trace[chain_lo:].ctypes.data
        w.n_planes = len(self.planes)
        w.copy_workgroups = self.copy_workgroups
        for p, (name, kind, idx, as_, _dt) in enumerate(self.planes):
            w.plane[p].dst = self.stats[name][chain_lo:].ctypes.data
            w.plane[p].kind, w.plane[p].idx, w.plane[p].as_ = int(kind), int(idx), int(as_)
        return w


# Test / A-B knobs of struct lmc_tuning and the environment variables the HOST reads them from when the caller does not pass
# ``tuning=`` (until ABI 7 the library read these variables itself -- hidden inputs to a C ABI; now they are plain fields of
# lmc_config and this table is the one place that looks at the environment).
_TUNING_ENV = {
    "LMC_SUB_BLOCKS": ("sub_blocks", int),
    "LMC_FORCE_WIDE": ("force_general", lambda v: int(int(v) != 0)),
    "LMC_WIDE_TEAM": ("general_team", int),
    "LMC_DENSE_COOP": ("dense_coop_off", lambda v: int(int(v) == 0)),
    "LMC_DENSE_CACHE_ROWS": ("dense_cache_rows_p1", lambda v: int(v) + 1),
    "LMC_DENSE_LDS_SLOTS": ("dense_lds_slots_p1", lambda v: int(v) + 1),
    "LMC_CHOL_HBM": ("chol_hbm", lambda v: int(int(v) != 0)),
}


def tuning_from_env(environ=None):
    """dict of lmc_tuning fields from the LMC_* environment variables (empty / malformed values are ignored)."""
    import os

    environ = os.environ if environ is None else environ
    out = {}
    for name, (field, conv) in _TUNING_ENV.items():
        raw = environ.get(name, "")
        if raw != "":
            try:
                out[field] = conv(raw)
            except ValueError:
                pass
    shape = environ.get("LMC_RUN_SHAPE", "")
    if shape:
        try:
            a, b = (int(x) for x in shape.split(","))
            out["run_ns"], out["run_w"] = a, b
        except ValueError:
            pass
    return out


class Engine:
    def __init__(self, target, chains, kind="nuts", potential="diag_adapt", device=0, lib_path=None,
                 target_accept=0.8, Emax=1000.0, adapt_step_size=True, step_scale=0.25, gamma=0.05, k=0.75,
                 t0=10, path_length=2.0, max_treedepth=10, early_max_treedepth=8, max_steps=1024,
                 adaptation_window=101, adaptation_window_multiplier=1.0, lds_levels=0, sdot=None, rng="numpy",
                 mass_dtype="float32", lds_plan="auto", tuning=None):
        self._lib = _abi.load(lib_path or getattr(target, "lib_path", None))
        self._h = C.c_void_p()
        self.target = target
        self.chains = int(chains)
        self.dim = int(target.d)
        self.kind = kind
        cfg = _abi.Config()
        self._lib.lmc_config_defaults(C.byref(cfg), self.chains, self.dim)
        cfg.device = int(device)
        cfg.kind = {"nuts": _abi.KIND_NUTS, "hmc": _abi.KIND_HMC}[kind]
        cfg.target_family = int(target.family)
        cfg.potential = {"diag_adapt": _abi.POT_DIAG_ADAPT, "diag": _abi.POT_DIAG, "full": _abi.POT_FULL,
                         "full_inv": _abi.POT_FULL_INV, "full_adapt": _abi.POT_FULL_ADAPT, "full_f64": _abi.POT_FULL_F64}[potential]
        self.potential = potential
        cfg.adapt_step_size = int(bool(adapt_step_size))
        cfg.target_accept = float(target_accept)
        cfg.emax = float(Emax)
        cfg.step_scale = float(step_scale)
        cfg.gamma = float(gamma)
        cfg.k = float(k)
        cfg.t0 = float(t0)
        cfg.max_treedepth = int(max_treedepth)
        cfg.early_max_treedepth = int(early_max_treedepth)
        cfg.path_length = float(path_length)
        cfg.max_steps = int(max_steps)
        cfg.adaptation_window = int(adaptation_window)
        cfg.adaptation_window_multiplier = float(adaptation_window_multiplier)
        cfg.lds_levels = int(lds_levels)
        # include/lmc_hip.h: LMC_LDS_PLAN_* (results do not depend on it; pinned values are for A/B runs and parity tests). Like
        # the lmc_tuning knobs, the variable the LIBRARY used to read (LMC_LDS_PLAN=0 / 1) is still honoured -- by the host, and
        # only when the caller left the choice open (tools/ab_lds_plan.sh, tools/sample_path_rate.py)
        if lds_plan in ("auto", None):
            import os

            lds_plan = {"0": 0, "1": 1}.get(os.environ.get("LMC_LDS_PLAN", ""), "auto")
        cfg.lds_plan = {"auto": _abi.LDS_PLAN_AUTO, None: _abi.LDS_PLAN_AUTO, 0: _abi.LDS_PLAN_SHALLOW, "shallow": _abi.LDS_PLAN_SHALLOW,
                        1: _abi.LDS_PLAN_DEEP, "deep": _abi.LDS_PLAN_DEEP}[lds_plan]
        for field, value in (tuning_from_env() if tuning is None else dict(tuning)).items():   # struct lmc_tuning: test / A-B knobs
            setattr(cfg.tuning, field, int(value))
        cfg.rng_mode = {"numpy": _abi.RNG_NUMPY, "philox": _abi.RNG_PHILOX}[rng]   # include/lmc_hip.h: LMC_RNG_*
        # QuadPotentialDiagAdapt(dtype=...) (quadpotential.py:159,175-184); float64 runs in the general kernels
        # QuadPotentialFullAdapt(dtype=...) (quadpotential.py:484,497-509) likewise: float64 covariance, factor and momentum
        self.mass_f64 = np.dtype(mass_dtype) == np.float64 and potential in ("diag_adapt", "diag", "full_adapt")
        cfg.mass_f64 = int(self.mass_f64)
        if cfg.target_family == _abi.TARGET_EXTERNAL:
            # a density evaluated by the caller (a torch / Python callable) is driven by the tick kernels, which exist for the
            # fused shapes and for float32 diagonals beyond them (include/lmc_hip.h: "Which kernels an engine runs")
            if potential not in ("diag_adapt", "diag") and self.dim > 256:
                raise NotImplementedError("a density given as a torch / Python callable runs with dense mass matrices up to "
                                          "model_ndim = 256 (got %d); give it as a device functor (targets.UserTarget) for "
                                          "larger ones" % self.dim)
            if self.mass_f64:
                raise NotImplementedError("a density given as a torch / Python callable runs with float32 diagonal mass matrices; "
                                          "give it as a device functor (targets.UserTarget) for dtype='float64'")
        if sdot is None:
            sdot = DEFAULT_SDOT
        if sdot == "auto":   # float32 start-energy rounding of the host's numpy (see _blas_probe.py)
            from ._blas_probe import detect_sdot_mode

            sdot = detect_sdot_mode()
        cfg.start_energy_sdot = {"native": _abi.SDOT_NATIVE, "skylakex": _abi.SDOT_OPENBLAS_SKYLAKEX,
                                 "haswell": _abi.SDOT_OPENBLAS_HASWELL}.get(sdot, sdot)
        self.cfg = cfg
        h = C.c_void_p()
        self._check(self._lib.lmc_engine_create(C.byref(cfg), C.byref(h)), handle=None)
        self._h = h
        params = np.ascontiguousarray(target.params, dtype=np.float64)
        self._check(self._lib.lmc_engine_set_target_params(self._h, _abi.ptr(params), params.size))
        if hasattr(target, "_attach"):   # a run-time compiled density hands its kernels to the engine (targets.UserTarget)
            target._attach(self)
        self.capacity = 0
        self.keep_trace = False
        self.trace_begin = 0
        self.wide = bool(self._lib.lmc_engine_uses_general_kernels(self._h))   # include/lmc_hip.h: "Which kernels an engine runs"

    def kernel_shape(self):
        """(unit_ns, run_ns, run_w) of this engine's kernels (lmc_engine_kernel_shape)."""
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self._lib.lmc_engine_kernel_shape(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return int(a.value), int(b.value), int(c.value)

    # ---- plumbing ---------------------------------------------------------------------------------
    def _check(self, rc, handle=True):
        if rc != _abi.OK:
            msg = self._lib.lmc_last_error(self._h if handle else None)
            raise _abi.HipLibraryError("liblmc_hip error %d: %s" % (rc, (msg or b"?").decode()))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.lmc_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_stream(self, stream_handle):
        self._check(self._lib.lmc_engine_set_stream(self._h, C.c_void_p(int(stream_handle) if stream_handle else 0)))

    def synchronize(self):
        self._check(self._lib.lmc_engine_synchronize(self._h))

    # ---- state --------------------------------------------------------------------------------------
    def set_potential(self, initial_mean, initial_diag, initial_weight=10.0):
        diag = np.ascontiguousarray(initial_diag, dtype=np.float64)
        per_chain = int(diag.ndim == 2)
        mean = None if initial_mean is None else np.ascontiguousarray(
            np.broadcast_to(np.asarray(initial_mean, dtype=np.float64), diag.shape))
        self._check(self._lib.lmc_engine_set_potential(self._h, _abi.ptr(mean), _abi.ptr(diag),
                                                       float(initial_weight), per_chain))

    def seed(self, seeds):
        s = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint64) & 0xFFFFFFFF, dtype=np.uint32)
        assert s.shape == (self.chains,)
        self._check(self._lib.lmc_engine_seed(self._h, _abi.ptr(s)))

    def set_rng_state(self, chain, state):
        """state: the tuple of np.random.get_state() ('MT19937', key[624], pos, has_gauss, gauss)."""
        key = np.ascontiguousarray(state[1], dtype=np.uint32)
        self._check(self._lib.lmc_engine_set_rng_state(self._h, int(chain), _abi.ptr(key), int(state[2]),
                                                       int(state[3]), float(state[4])))

    def get_rng_state(self, chain):
        key = np.zeros(624, dtype=np.uint32)
        pos, hg, g = C.c_int32(), C.c_int32(), C.c_double()
        self._check(self._lib.lmc_engine_get_rng_state(self._h, int(chain), _abi.ptr(key), C.byref(pos),
                                                       C.byref(hg), C.byref(g)))
        return ("MT19937", key, int(pos.value), int(hg.value), float(g.value))

    def set_position(self, q):
        q = np.ascontiguousarray(q, dtype=np.float64)
        self._check(self._lib.lmc_engine_set_position(self._h, _abi.ptr(q), int(q.ndim == 2)))

    def get_position(self):
        q = np.empty((self.chains, self.dim))
        self._check(self._lib.lmc_engine_get_position(self._h, _abi.ptr(q)))
        return q

    def reset_tuning(self):
        self._check(self._lib.lmc_engine_reset_tuning(self._h))

    def set_dual_average(self, log_step, log_bar, hbar=0.0, count=1):
        self._check(self._lib.lmc_engine_set_dual_average(self._h, float(log_step), float(log_bar), float(hbar),
                                                          int(count)))

    # ---- sampling -----------------------------------------------------------------------------------
    def reserve(self, capacity, keep_trace=True, trace_begin=0):
        """Output storage for ``capacity`` iterations; draws kept for iterations >= trace_begin."""
        tb = int(trace_begin) if keep_trace else -1
        self._check(self._lib.lmc_engine_reserve(self._h, int(capacity), tb))
        self.capacity = int(capacity)
        self.keep_trace = bool(keep_trace) and tb < capacity
        self.trace_begin = max(tb, 0)

    def attach_trace(self, out, trace_begin):
        """Where the draws of iterations >= trace_begin go, after reserve(keep_trace=False): ``out`` = a device-accessible
        [chains, capacity - trace_begin, dim] float64 array (pinned_empty) the sampling kernel writes directly, or None = a
        trace in HBM (include/lmc_hip.h: lmc_engine_attach_trace)."""
        if out is not None:
            assert out.shape == (self.chains, self.capacity - int(trace_begin), self.dim) and out.dtype == np.float64 and out.flags["C_CONTIGUOUS"]
        self._check(self._lib.lmc_engine_attach_trace(self._h, None if out is None else C.c_void_p(out.ctypes.data), int(trace_begin)))
        self._trace_out = out            # (kept alive for as long as the engine may write it)
        self.keep_trace, self.trace_begin = True, int(trace_begin)

    def run(self, n_tune, iter_begin, n_iters):
        if self.target.family == _abi.TARGET_EXTERNAL:
            return self._run_ticks(int(n_tune), int(iter_begin), int(n_iters))
        self._check(self._lib.lmc_engine_run(self._h, int(n_tune), int(iter_begin), int(n_iters)))

    def set_step_jitter(self, lo, hi, enable=True):
        """step_rand as step * uniform(lo, hi) drawn from each chain's own stream (include/lmc_hip.h)."""
        self._check(self._lib.lmc_engine_set_step_jitter(self._h, int(bool(enable)), float(lo), float(hi)))

    def set_step_sizes(self, step_sizes):
        """The step size every chain integrates its NEXT iteration with (an arbitrary host step_rand evaluated by the
        caller, include/lmc_hip.h: lmc_engine_set_step_sizes); None switches back to the adapted ones."""
        if step_sizes is None:
            self._check(self._lib.lmc_engine_set_step_sizes(self._h, None))
            return
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(step_sizes, dtype=np.float64), (self.chains,)))
        self._check(self._lib.lmc_engine_set_step_sizes(self._h, _abi.ptr(a)))

    def diag_update(self, tune=True):
        """potential.update(current position, grad, tune) of QuadPotentialDiagAdapt for every chain (quadpotential.py:231-245)."""
        self._check(self._lib.lmc_engine_diag_update(self._h, int(bool(tune))))

    def request_stop(self, stop=True):
        """Ctrl-C for the device: every chain leaves its launch at its next iteration boundary (include/lmc_hip.h)."""
        self._check(self._lib.lmc_engine_request_stop(self._h, int(bool(stop))))

    def progress(self):
        """Iteration index the running job has reached -- a hint read from pinned host memory, no stream is touched
        (lmc_engine_progress); what every chain has COMPLETED is completed_iterations(), which waits for the launches."""
        return int(self._lib.lmc_engine_progress(self._h))

    def completed_iterations(self):
        """Iterations EVERY chain has completed since reset_tuning() (the smallest per-chain iteration count)."""
        return int(self.get_chain_state(fields=("iter_count",))["iter_count"].min())

    def occupancy(self):
        """(resident_chains, waves_per_chain, wall_clock_hz) of this engine's sampling kernel, asked of the HIP runtime
        for the very kernel / block size / dynamic LDS run() launches with (lmc_engine_occupancy)."""
        rc, wpc, hz = C.c_int32(), C.c_int32(), C.c_double()
        self._check(self._lib.lmc_engine_occupancy(self._h, C.byref(rc), C.byref(wpc), C.byref(hz)))
        return int(rc.value), int(wpc.value), float(hz.value)

    def run_lds_bytes(self):
        """Dynamic LDS bytes per workgroup of the sampling kernel (None: no fused diagonal-mass kernel)."""
        n = int(self._lib.lmc_engine_run_lds_bytes(self._h))
        return n if n >= 0 else None

    def resident_chains(self):
        """How many chains the sampling kernel keeps resident on the GPU at once; None for engines without a fused
        diagonal-mass sampling kernel."""
        if self.target.family == _abi.TARGET_EXTERNAL:
            return None
        return self.occupancy()[0] or None

    def last_run_plan(self):
        """LDS plan of the most recent run() launch: "shallow" / "deep" (None: nothing launched, or a kernel with one plan)."""
        return {_abi.LDS_PLAN_SHALLOW: "shallow", _abi.LDS_PLAN_DEEP: "deep"}.get(int(self._lib.lmc_engine_last_run_plan(self._h)))

    def run_streams(self):
        """Raw HIP stream handles run() launches its kernels on (one per sub-block of chains)."""
        arr = (C.c_void_p * _abi.MAX_RUN_STREAMS)()
        n = min(self._lib.lmc_engine_run_streams(self._h, arr, _abi.MAX_RUN_STREAMS), _abi.MAX_RUN_STREAMS)
        if n < 0:
            self._check(n)
        return [int(arr[i] or 0) for i in range(n)]

    # ---- externally evaluated density (targets.TorchTarget): the tick protocol of include/lmc_hip.h ---------
    def tick_begin(self, n_tune, iter_begin, n_iters):
        self._check(self._lib.lmc_engine_tick_begin(self._h, int(n_tune), int(iter_begin), int(n_iters)))

    def tick_positions_ptr(self):
        return int(self._lib.lmc_engine_tick_positions(self._h) or 0)

    def tick(self, logp_ptr, grad_ptr, wait=True):
        """One evaluation step for every unfinished chain; returns the number still running (None if not waited for)."""
        n = C.c_int32(-1)
        self._check(self._lib.lmc_engine_tick(self._h, C.c_void_p(int(logp_ptr)), C.c_void_p(int(grad_ptr)),
                                              C.byref(n) if wait else None))
        return int(n.value) if wait else None

    def _run_ticks(self, n_tune, iter_begin, n_iters, poll=32):
        """Drive n_iters iterations of every chain with the target's batched torch callable. Engine kernels and the
        callable's kernels alternate on ONE non-default HIP stream; the host only looks every ``poll`` ticks."""
        import torch

        if n_iters == 0:
            return
        dev = torch.device("cuda", int(self.cfg.device))
        if getattr(self, "_tick_stream", None) is None:
            self._tick_stream = torch.cuda.Stream(device=dev)
        self.synchronize()
        self.set_stream(self._tick_stream.cuda_stream)
        try:
            with torch.cuda.device(dev), torch.cuda.stream(self._tick_stream):
                self.tick_begin(n_tune, iter_begin, n_iters)
                q = torch.as_tensor(_TickView(self.tick_positions_ptr(), (self.chains, self.dim)), device=dev)
                graph = None
                if getattr(self.target, "graph", False):
                    graph = getattr(self, "_tick_graph", None)
                    if graph is None:   # warm up (allocator, lazy init), then capture fn(q) once
                        for _ in range(3):
                            self.target.evaluate(q)
                        self._tick_stream.synchronize()
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g, stream=self._tick_stream):
                            out = self.target.evaluate(q)
                        graph = self._tick_graph = (g, out)
                ticks = 0
                while True:
                    if graph is not None:
                        graph[0].replay()
                        logp, grad = graph[1]
                    else:
                        logp, grad = self.target.evaluate(q)
                    ticks += 1
                    active = self.tick(logp.data_ptr(), grad.data_ptr(),
                                       wait=(ticks % getattr(self.target, "tick_poll", poll) == 0))
                    self._tick_keep = (logp, grad)   # alive until the next evaluation is enqueued behind the tick
                    if active == 0:
                        break
                self._tick_stream.synchronize()
                self.ticks = getattr(self, "ticks", 0) + ticks
        finally:
            self.set_stream(None)

    def trace(self, iter_begin=None, n_iters=None):
        iter_begin = self.trace_begin if iter_begin is None else iter_begin
        n = self.capacity - iter_begin if n_iters is None else n_iters
        out = np.empty((self.chains, n, self.dim))
        self._check(self._lib.lmc_engine_get_trace(self._h, _abi.ptr(out), int(iter_begin), int(n)))
        return out

    def stat_f64(self, stat, iter_begin=0, n_iters=None):
        n = self.capacity - iter_begin if n_iters is None else n_iters
        out = np.empty((self.chains, n))
        self._check(self._lib.lmc_engine_get_stat_f64(self._h, int(stat), _abi.ptr(out), int(iter_begin), int(n)))
        return out

    def stat_i32(self, stat, iter_begin=0, n_iters=None):
        n = self.capacity - iter_begin if n_iters is None else n_iters
        out = np.empty((self.chains, n), dtype=np.int32)
        self._check(self._lib.lmc_engine_get_stat_i32(self._h, int(stat), _abi.ptr(out), int(iter_begin), int(n)))
        return out

    def stat_u8(self, stat, iter_begin=0, n_iters=None):
        n = self.capacity - iter_begin if n_iters is None else n_iters
        out = np.empty((self.chains, n), dtype=np.uint8)
        self._check(self._lib.lmc_engine_get_stat_u8(self._h, int(stat), _abi.ptr(out), int(iter_begin), int(n)))
        return out

    # ---- streamed results (include/lmc_hip.h: lmc_engine_copy_window_async) ---------------------------------
    def copy_window_async(self, out, iter_begin, n_iters, chain_lo=0):
        """Enqueue the device->host copy of iterations [iter_begin, iter_begin + n_iters) of every chain into the final
        arrays of ``out`` (a StreamedResults; this engine's chains start at its row ``chain_lo``), ordered after the launches
        enqueued so far, asynchronous to the host."""
        self._check(self._lib.lmc_engine_copy_window_async(self._h, C.byref(out.window_dst(self, chain_lo)), int(iter_begin), int(n_iters)))

    def copy_wait(self):
        self._check(self._lib.lmc_engine_copy_wait(self._h))

    def trace_device_ptr(self):
        return self._lib.lmc_engine_trace_device_ptr(self._h)

    def stat_records_device_ptr(self):
        """[chains][capacity] records of 64 bytes, one per draw (layout: include/lmc_hip.h)."""
        return self._lib.lmc_engine_stat_records_device_ptr(self._h)

    def tree_size_view(self):
        """__cuda_array_interface__ object: tree_size [chains, capacity] int32 where it lies in the per-draw records (strided)."""
        rec = _abi.STAT_RECORD_BYTES

        class _V:
            pass

        v = _V()
        v.__cuda_array_interface__ = {"shape": (self.chains, self.capacity), "typestr": "<i4", "version": 2,
                                      "data": (int(self.stat_records_device_ptr()) + 56, False),
                                      "strides": (self.capacity * rec, rec)}
        return v

    def adapt_state(self):
        var = np.empty((self.chains, self.dim), dtype=np.float32)
        da = np.empty((self.chains, 4))
        cnt = np.empty(self.chains, dtype=np.int32)
        ns = np.empty(self.chains, dtype=np.int32)
        self._check(self._lib.lmc_engine_get_adapt_state(self._h, _abi.ptr(var), _abi.ptr(da), _abi.ptr(cnt),
                                                         _abi.ptr(ns)))
        if self.mass_f64:   # the potential's own dtype
            var = self.get_chain_state(fields=("var64",))["var64"]
        return {"var": var, "log_step": da[:, 0], "log_bar": da[:, 1], "hbar": da[:, 2], "mu": da[:, 3],
                "count": cnt, "n_samples": ns}

    def get_chain_state(self, fields=None):
        """Full adaptation state of every chain as a dict of host arrays (checkpoint); ``fields`` restricts the copy."""
        st = _abi.ChainState()
        out = {}
        for name, dt, vec in _abi.ChainState.FIELDS:
            if fields is not None and name not in fields:
                continue
            if name == "var64" and not self.mass_f64:   # state of a float64 adaptive diagonal only
                continue
            out[name] = np.empty((self.chains, self.dim) if vec else (self.chains,), dtype=dt)
            setattr(st, name, out[name].ctypes.data)
        self._check(self._lib.lmc_engine_get_chain_state(self._h, C.byref(st)))
        return out

    def set_chain_state(self, state):
        """Restore (a subset of) the fields returned by get_chain_state(); arrays are [chains, dim] / [chains]."""
        st = _abi.ChainState()
        keep = []
        for name, dt, vec in _abi.ChainState.FIELDS:
            if name in state and state[name] is not None:
                shape = (self.chains, self.dim) if vec else (self.chains,)
                a = np.ascontiguousarray(np.broadcast_to(np.asarray(state[name], dtype=dt), shape))
                keep.append(a)
                setattr(st, name, a.ctypes.data)
        self._check(self._lib.lmc_engine_set_chain_state(self._h, C.byref(st)))

    # ---- dense mass matrices (QuadPotentialFull / FullInv / FullAdapt) ---------------------------------
    def set_dense_potential(self, matrix, initial_mean=None, initial_weight=0.0, adaptation_window=101,
                            adaptation_window_multiplier=2.0, update_window=1):
        m = np.ascontiguousarray(matrix, dtype=np.float64)
        if m.shape != (self.dim, self.dim):
            raise ValueError("matrix must have shape (%d, %d)" % (self.dim, self.dim))
        mean = None if initial_mean is None else np.ascontiguousarray(initial_mean, dtype=np.float64)
        self._check(self._lib.lmc_engine_set_dense_potential(
            self._h, _abi.ptr(m), None if mean is None else _abi.ptr(mean), float(initial_weight),
            int(adaptation_window), float(adaptation_window_multiplier), int(update_window)))

    def _dense_shape(self, kind):
        return {"m": (self.chains, self.dim, self.dim), "v": (self.chains, self.dim), "s": (self.chains,)}[kind]

    def _dense_f64(self):
        """cov / chol travel as float64: QuadPotentialFullAdapt(dtype="float64")."""
        return self.mass_f64 and self.potential == "full_adapt"

    def dense_chain(self, chain=0):
        """(cov, chol) of one chain [dim, dim], in the potential's dtype (float32; float64 for FullAdapt(dtype="float64"))."""
        dt = np.float64 if self._dense_f64() else np.float32
        cov = np.empty((self.dim, self.dim), dtype=dt)
        chol = np.empty((self.dim, self.dim), dtype=dt)
        get = self._lib.lmc_engine_get_dense_chain_f64 if self._dense_f64() else self._lib.lmc_engine_get_dense_chain
        self._check(get(self._h, int(chain), _abi.ptr(cov), _abi.ptr(chol)))
        return cov, chol

    def dense_factor_f64(self):
        """Lower Cholesky factor [dim, dim] float64 of the float64 dense potentials (Full(dtype="float64"), FullInv)."""
        L = np.empty((self.dim, self.dim))
        self._check(self._lib.lmc_engine_get_dense_factor_f64(self._h, _abi.ptr(L)))
        return L

    def get_dense_state(self, fields=None):
        """cov / chol of every chain; for "full_adapt" also the two covariance estimators and the window state.
        ``fields`` restricts the copy (the matrices are chains x dim x dim)."""
        st = _abi.DenseState()
        out = {}
        f64 = self._dense_f64()
        for name, dt, kind in _abi.DenseState.FIELDS:
            if name in ("cov64", "chol64"):   # "cov" / "chol" come in the potential's dtype
                continue
            if self.potential != "full_adapt" and name not in ("cov", "chol"):
                continue
            if fields is not None and name not in fields:
                continue
            wide = f64 and name in ("cov", "chol")
            out[name] = np.empty(self._dense_shape(kind), dtype=np.float64 if wide else dt)
            setattr(st, name + "64" if wide else name, out[name].ctypes.data)
        self._check(self._lib.lmc_engine_get_dense_state(self._h, C.byref(st)))
        return out

    def set_dense_state(self, state):
        st = _abi.DenseState()
        keep = []
        f64 = self._dense_f64()
        for name, dt, kind in _abi.DenseState.FIELDS:
            if name in ("cov64", "chol64"):
                continue
            if name in state and state[name] is not None:
                wide = f64 and name in ("cov", "chol")
                a = np.ascontiguousarray(np.broadcast_to(np.asarray(state[name], dtype=np.float64 if wide else dt), self._dense_shape(kind)))
                keep.append(a)
                setattr(st, name + "64" if wide else name, a.ctypes.data)
        self._check(self._lib.lmc_engine_set_dense_state(self._h, C.byref(st)))

    def dense_update(self, tune=True):
        """potential.update(current position, grad, tune) for every chain (quadpotential.py:528-552)."""
        self._check(self._lib.lmc_engine_dense_update(self._h, int(bool(tune))))

    def keep_moments(self, enable=True):
        """Accumulate per-chain mean / M2 of the post-warm-up draws on the device (no trace needed for R-hat)."""
        self._check(self._lib.lmc_engine_keep_moments(self._h, int(bool(enable))))

    def moments(self):
        mean = np.empty((self.chains, self.dim))
        m2 = np.empty((self.chains, self.dim))
        n = np.empty(self.chains, dtype=np.int32)
        self._check(self._lib.lmc_engine_get_moments(self._h, _abi.ptr(mean), _abi.ptr(m2), _abi.ptr(n)))
        return mean, m2, n

    def status(self):
        st = np.empty(self.chains, dtype=np.int32)
        self._check(self._lib.lmc_engine_get_status(self._h, _abi.ptr(st)))
        return st

    def counters(self):
        ct = np.empty((self.chains, _abi.NUM_COUNTERS), dtype=np.int64)
        self._check(self._lib.lmc_engine_get_counters(self._h, _abi.ptr(ct)))
        return ct

    # ---- unit entry points --------------------------------------------------------------------------
    def trajectory(self, q0, p0, eps, n_fwd, n_back=0, p0_is_f32=None):
        q0 = np.ascontiguousarray(np.broadcast_to(np.asarray(q0, dtype=np.float64), (self.chains, self.dim)))
        if p0_is_f32 is None:
            p0_is_f32 = np.asarray(p0).dtype == np.float32
        p0 = np.ascontiguousarray(np.broadcast_to(np.asarray(p0, dtype=np.float64), (self.chains, self.dim)))
        ns = n_fwd + n_back + 1
        out = {k: np.empty((self.chains, ns, self.dim)) for k in ("q", "p", "v", "g")}
        out["energy"] = np.empty((self.chains, ns))
        out["logp"] = np.empty((self.chains, ns))
        self._check(self._lib.lmc_engine_trajectory(
            self._h, _abi.ptr(q0), _abi.ptr(p0), int(bool(p0_is_f32)), float(eps), int(n_fwd), int(n_back),
            _abi.ptr(out["q"]), _abi.ptr(out["p"]), _abi.ptr(out["v"]), _abi.ptr(out["g"]),
            _abi.ptr(out["energy"]), _abi.ptr(out["logp"])))
        return out

    def logp_dlogp(self, q):
        q = np.ascontiguousarray(np.broadcast_to(np.asarray(q, dtype=np.float64), (self.chains, self.dim)))
        logp = np.empty(self.chains)
        grad = np.empty((self.chains, self.dim))
        self._check(self._lib.lmc_engine_logp_dlogp(self._h, _abi.ptr(q), _abi.ptr(logp), _abi.ptr(grad)))
        return logp, grad

    def rng_draw(self, ops):
        ops = np.ascontiguousarray(ops, dtype=np.int32)
        total = int(np.abs(ops).sum())
        out = np.empty((self.chains, total))
        self._check(self._lib.lmc_engine_rng_draw(self._h, _abi.ptr(ops), ops.size, _abi.ptr(out)))
        return out

    def draw_momentum(self):
        out = np.empty((self.chains, self.dim))
        self._check(self._lib.lmc_engine_draw_momentum(self._h, _abi.ptr(out)))
        return out


class EngineGroup:
    """One job on several GPUs driven from ONE process: engine k owns the contiguous chain block ``blocks[k]`` of the
    job's global chain index space on device ``devices[k]`` (the reference fans its chains out from inside ``sample()``
    too: one worker per chain up to ``cores``, sampling.py:124-129,186-201 -> parallel_sampling.py). Chains share
    nothing, so there is no data path between the engines: every call is the same call on each engine with its slice of
    the arguments, launches are enqueued on all devices before anything is waited for, and results are concatenated in
    chain order -- the group reads like one Engine of ``chains`` chains (same seeds -> the same draws, chain for chain,
    as one engine holding them all). A device may appear more than once (two engines sharing a GPU: how the one-GPU test
    box exercises this path)."""

    def __init__(self, engines, blocks):
        assert len(engines) == len(blocks) and engines
        self.engines = list(engines)
        self.blocks = [(int(lo), int(hi)) for lo, hi in blocks]
        self.chains = self.blocks[-1][1]
        self.dim = engines[0].dim
        self.target = engines[0].target
        self.kind = engines[0].kind
        self.potential = engines[0].potential
        self.cfg = engines[0].cfg        # shape / method fields; cfg.device and cfg.chains are the FIRST engine's
        self.devices = [int(e.cfg.device) for e in engines]

    # ---- plumbing -----------------------------------------------------------------------------------------------
    def _each(self, name, *args, **kw):
        return [getattr(e, name)(*args, **kw) for e in self.engines]

    def _cat(self, name, *args, **kw):
        return np.concatenate(self._each(name, *args, **kw), axis=0)

    def _locate(self, chain):
        chain = int(chain) % self.chains
        for k, (lo, hi) in enumerate(self.blocks):
            if lo <= chain < hi:
                return self.engines[k], chain - lo
        raise IndexError(chain)

    def close(self):
        for e in self.engines:
            e.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def synchronize(self):
        self._each("synchronize")

    def copy_window_async(self, out, iter_begin, n_iters):
        for e, (lo, _hi) in zip(self.engines, self.blocks):
            e.copy_window_async(out, iter_begin, n_iters, chain_lo=lo)

    def attach_trace(self, out, trace_begin):
        for e, (lo, hi) in zip(self.engines, self.blocks):
            e.attach_trace(None if out is None else out[lo:hi], trace_begin)

    def copy_wait(self):
        self._each("copy_wait")

    # ---- state ----------------------------------------------------------------------------------------------------
    def seed(self, seeds):
        seeds = np.asarray(seeds)
        assert seeds.shape == (self.chains,)
        for e, (lo, hi) in zip(self.engines, self.blocks):
            e.seed(seeds[lo:hi])

    def set_position(self, q):
        q = np.asarray(q, dtype=np.float64)
        for e, (lo, hi) in zip(self.engines, self.blocks):
            e.set_position(q if q.ndim == 1 else np.ascontiguousarray(q[lo:hi]))

    def get_position(self):
        return self._cat("get_position")

    def reset_tuning(self):
        self._each("reset_tuning")

    def keep_moments(self, enable=True):
        self._each("keep_moments", enable)

    def moments(self):
        parts = self._each("moments")
        return tuple(np.concatenate([p[i] for p in parts], axis=0) for i in range(3))

    def set_step_jitter(self, lo, hi, enable=True):
        self._each("set_step_jitter", lo, hi, enable)

    def set_step_sizes(self, step_sizes):
        if step_sizes is None:
            self._each("set_step_sizes", None)
            return
        a = np.broadcast_to(np.asarray(step_sizes, dtype=np.float64), (self.chains,))
        for e, (lo, hi) in zip(self.engines, self.blocks):
            e.set_step_sizes(a[lo:hi])

    # ---- sampling ---------------------------------------------------------------------------------------------------
    def reserve(self, capacity, keep_trace=True, trace_begin=0):
        self._each("reserve", capacity, keep_trace=keep_trace, trace_begin=trace_begin)
        self.capacity, self.keep_trace, self.trace_begin = (self.engines[0].capacity, self.engines[0].keep_trace,
                                                            self.engines[0].trace_begin)

    def run(self, n_tune, iter_begin, n_iters):
        """Asynchronous on every device (fused kernels): the launch is enqueued everywhere before anyone waits."""
        self._each("run", n_tune, iter_begin, n_iters)

    def request_stop(self, stop=True):
        self._each("request_stop", stop)

    def progress(self):
        return min(self._each("progress"))

    def completed_iterations(self):
        return min(self._each("completed_iterations"))

    def resident_chains(self):
        """Resident slots of ONE device's kernel (the engines are alike): what launch sizing compares a block with."""
        return self.engines[0].resident_chains()

    def occupancy(self):
        return self.engines[0].occupancy()

    def trace(self, iter_begin=None, n_iters=None):
        return self._cat("trace", iter_begin, n_iters)

    def stat_f64(self, stat, iter_begin=0, n_iters=None):
        return self._cat("stat_f64", stat, iter_begin, n_iters)

    def stat_i32(self, stat, iter_begin=0, n_iters=None):
        return self._cat("stat_i32", stat, iter_begin, n_iters)

    def stat_u8(self, stat, iter_begin=0, n_iters=None):
        return self._cat("stat_u8", stat, iter_begin, n_iters)

    def status(self):
        return self._cat("status")

    def counters(self):
        return self._cat("counters")

    def adapt_state(self):
        parts = self._each("adapt_state")
        return {k: np.concatenate([p[k] for p in parts], axis=0) for k in parts[0]}

    def get_chain_state(self, fields=None):
        parts = self._each("get_chain_state", fields)
        return {k: np.concatenate([p[k] for p in parts], axis=0) for k in parts[0]}

    def get_dense_state(self, fields=None):
        parts = self._each("get_dense_state", fields)
        return {k: np.concatenate([p[k] for p in parts], axis=0) for k in parts[0]}

    def dense_chain(self, chain=0):
        e, c = self._locate(chain)
        return e.dense_chain(c)

    def dense_factor_f64(self):
        return self.engines[0].dense_factor_f64()
