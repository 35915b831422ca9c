"""``sample`` / ``init_nuts`` -- the driver boundary of /root/reference/littlemcmc/sampling.py.

Same call signature, same seed derivation, same start-point logic, same return layout
(``trace[chains, draws, ndim]``, ``stats[name][chains, draws, 1]`` cast to ``step.stats_dtypes``;
sampling.py:207-220). What changed is everything underneath: instead of one Python loop per chain
(sampling.py:331-521) or one OS process per chain (parallel_sampling.py), all chains run as
wavefronts of one HIP kernel (csrc/lmc_sampler.hpp) and the host only slices the run into a few
launches. ``cores`` caps the number of GPUs a job is spread over (the reference: worker processes); ``mp_ctx`` and
``pickle_backend`` are accepted and ignored.
"""
import logging
import os
from collections.abc import Iterable

import numpy as np

from . import _abi
from .base_hmc import raise_for_status
from .nuts import NUTS
from .quadpotential import QuadPotentialDiagAdapt, QuadPotentialFullAdapt
from .targets import require_device_target

_log = logging.getLogger("littlemcmc_amd")
_STREAM_MIN_BYTES = 8 << 20   # results smaller than this are copied when the job is over (stream_results=True only: an explicit mode is honoured)


def _derive_seeds(random_seed, chains):
    """sampling.py:131-138: per-chain seeds from an int (global legacy stream) or a list."""
    if random_seed is None or isinstance(random_seed, (int, np.integer)):
        if random_seed is not None:
            np.random.seed(int(random_seed))
        return [int(np.random.randint(2 ** 30)) for _ in range(chains)]
    if isinstance(random_seed, Iterable):
        seeds = [int(s) for s in random_seed]
        if len(seeds) < chains:
            raise ValueError("random_seed list is shorter than the number of chains")
        return seeds[:chains]
    raise TypeError("Invalid value for `random_seed`. Must be tuple, list or int")


def init_nuts(logp_dlogp_func, model_ndim=None, init="auto", random_seed=None, size=None, **kwargs):
    """Set up start point and NUTS sampler (sampling.py:524-605): "adapt_diag", "jitter+adapt_diag",
    "adapt_full", "jitter+adapt_full" (dense modes: model_ndim <= 256)."""
    if model_ndim is None:
        model_ndim = size if size is not None else getattr(logp_dlogp_func, "d", None)
    logp_dlogp_func = require_device_target(logp_dlogp_func, model_ndim)
    if not isinstance(init, str):
        raise TypeError("init must be a string.")
    init = init.lower()
    if init == "auto":
        init = "jitter+adapt_diag"
    _log.info("Initializing NUTS using {}...".format(init))
    if random_seed is not None:   # sampling.py:574-576
        random_seed = int(np.atleast_1d(random_seed)[0])
        np.random.seed(random_seed)
    if init == "adapt_diag":
        start = np.zeros(model_ndim)
    elif init == "jitter+adapt_diag":
        start = 2 * np.random.rand(model_ndim) - 1
    elif init == "adapt_full":
        start = np.zeros(model_ndim)
    elif init == "jitter+adapt_full":
        start = 2 * np.random.rand(model_ndim) - 1
    else:
        raise ValueError("Unknown initializer: {}.".format(init))
    if init.endswith("adapt_full"):   # sampling.py:588-597
        potential = QuadPotentialFullAdapt(model_ndim, start, np.eye(model_ndim), 10)
    else:
        potential = QuadPotentialDiagAdapt(model_ndim, start, np.ones(model_ndim), 10)
    step = NUTS(logp_dlogp_func=logp_dlogp_func, model_ndim=model_ndim, potential=potential, **kwargs)
    return start, step


class JobProgress:
    """What ``callback(trace=None, draw=...)`` receives while a job runs (the reference calls its callback once per draw
    with the draw, sampling.py:272-277, :307-308; here all chains advance inside device launches, so the callback is
    called from the host's wait loop whenever the job has moved on): ``iteration`` = the iteration the chains have
    reached (a device-written hint, Engine.progress()), ``total``, ``tuning`` (still below ``tune``), ``chains``,
    ``launches_done``. Raising KeyboardInterrupt in the callback interrupts sampling, as in the reference."""

    __slots__ = ("iteration", "total", "tuning", "chains", "launches_done")

    def __init__(self, iteration, total, tuning, chains, launches_done):
        self.iteration, self.total, self.tuning, self.chains, self.launches_done = iteration, total, tuning, chains, launches_done

    def __repr__(self):
        return "JobProgress(iteration=%d, total=%d, tuning=%s, chains=%d)" % (self.iteration, self.total, self.tuning, self.chains)


class _ResultStreamer:
    """Fills the arrays sample() returns while the job runs (sampling.py:207-222 of the reference returns host arrays; a
    job's draws are tens of GiB, so nothing may wait for the end of the job). The arrays are page-locked host memory:
    allocated and PRE-FAULTED by a helper thread while the first launches run (the page faults are the expensive part and pure
    CPU work), then registered with the HIP runtime (hipHostRegister: 0.03 s per 15 GiB of resident memory) on the thread that
    drives the engine -- this package never enters the HIP runtime from two threads at once.

    * draws: ``direct=True`` -- the sampling kernel stores every draw straight into the returned trace array as it is
      produced (Engine.attach_trace: coalesced rows over the host link, no trace in HBM, nothing left to copy when the job
      ends); ``direct=False`` -- the trace stays in HBM and every launch's window is copied under the following launches.
    * statistics (82 bytes per draw): the window of every launch is gathered out of the per-draw records, converted to the
      reference's dtypes on the device and written into the returned arrays under the following launches
      (Engine.copy_window_async). Windows that come before the arrays exist wait in a list."""

    def __init__(self, eng, chains, n_out, first, dim, planes, direct):
        import threading

        self.eng, self.first, self.out, self.err, self.waiting = eng, int(first), None, None, []
        self.direct, self.attached = bool(direct), False

        def alloc():
            import gc

            from .engine import StreamedResults

            # CPU work only -- and no cyclic garbage collection started from this thread: a collection could finalize an
            # Engine or a pinned block of an earlier job HERE, i.e. enter the HIP runtime beside the driving thread
            was_enabled = gc.isenabled()
            gc.disable()
            try:
                self.out = StreamedResults(chains, n_out, first, dim, planes, direct=direct, register=False)
            except BaseException as err:     # the host cannot allocate that much: the draws are copied after the job instead
                self.err = err
            finally:
                if was_enabled:
                    gc.enable()

        self._thread = threading.Thread(target=alloc, name="lmc-pin-results", daemon=True)
        self._thread.start()

    def before_launch(self, first, n):
        """before_enqueue(first, n) of _run_job: the first launch that reaches iteration ``self.first`` needs to know where
        the draws go (direct mode): wait for the arrays and attach the trace -- or, if they could not be pinned, a trace in HBM."""
        if self.direct and not self.attached and int(first) + int(n) > self.first:
            self._thread.join()
            self._register()
            self.eng.attach_trace(self.out.trace if self.out is not None else None, self.first)
            self.attached = True

    def window(self, first, n):
        """on_enqueued(first, n) of _run_job: iterations [first, first + n) have been launched."""
        a, b = max(int(first), self.first), int(first) + int(n)
        if b > a:
            self.waiting.append((a, b - a))
        self._flush()

    def _register(self):
        """Page-lock the arrays the helper thread prepared (a HIP call: made here, on the driving thread). On failure the job
        goes on without streamed results (trace in HBM, everything copied when it is over)."""
        if self.out is not None and not self.out.registered:
            try:
                self.out.register()
            except Exception as err:
                self.err, self.out = err, None

    def _flush(self):
        if self._thread.is_alive():
            return
        self._register()
        if self.out is None:
            return
        for a, n in self.waiting:
            self.eng.copy_window_async(self.out, a, n)
        self.waiting = []

    def finish(self):
        """Wait for the arrays and for every copy; returns the StreamedResults (None if the memory could not be pinned)."""
        self._thread.join()
        self._register()
        if self.out is None:
            _log.warning("results were not streamed (%s); copying them now that the job is over" % (self.err,))
            return None
        self._flush()
        self.eng.copy_wait()
        return self.out


def _run_job(eng, tune, n_total, per_launch, progressbar, callback=None, on_enqueued=None, before_enqueue=None):
    """Enqueue the job's launches, then wait for them in a way Ctrl-C can reach (sampling.py:324-328, :470-471 in the
    reference: a KeyboardInterrupt ends sampling and what has been drawn so far is returned).

    The launches are asynchronous and back to back (the engine's sub-block streams keep the chip full across launch
    boundaries); the host keeps two of them queued, polls the completion events of the launches it queued,
    logs progress as they complete (``progressbar=True``; the reference's per-draw bar, sampling.py:455-459, at launch
    granularity) and, on Ctrl-C, asks the device to stop -- every chain leaves its launch at its next iteration
    boundary and the launches still queued do nothing (lmc_engine_request_stop). With several GPUs (an EngineGroup)
    every launch is enqueued on all devices before anything is waited for, and a launch counts as complete when it is
    complete everywhere. ``on_enqueued(first, n)`` is called right after the launch of iterations [first, first + n) has
    been enqueued (sample() enqueues the copy of that window into the result arrays there, so the copy of launch k runs under
    launch k + 1); ``before_enqueue(first, n)`` right before it is. Returns (iterations completed by EVERY chain, interrupted)."""
    import time

    engines = getattr(eng, "engines", [eng])
    marks = []   # (iterations enqueued so far, [event per run stream of every engine]) per launch
    events_ok = eng.target.family != _abi.TARGET_EXTERNAL
    torch = None
    if events_ok:
        try:
            import torch
            streams = [torch.cuda.ExternalStream(h, device=torch.device("cuda", int(e_.cfg.device)))
                       for e_ in engines for h in e_.run_streams()]
        except Exception:   # no torch: plain blocking wait (no progress lines, Ctrl-C acts when the job ends)
            torch = None
    t0 = time.perf_counter()
    # The launches are enqueued a few ahead of execution, not all up front: the engine picks the LDS plan of a launch when it
    # is ENQUEUED, from the tree sizes the running chains report (lmc_engine.hip: choose_lds_plan; results do not depend on
    # it), so the queue must not run far ahead of the job. Two in flight per stream keep the device busy across launch
    # boundaries (the next one is enqueued while the last one runs).
    sizes = [int(x) for x in per_launch] if isinstance(per_launch, (list, tuple)) else [int(per_launch)]   # (the last size repeats)
    if not sizes or min(sizes) < 1:
        raise ValueError("launch sizes must be >= 1 iteration (got %r)" % (per_launch,))
    pending = []
    it = 0
    while it < n_total:
        n = min(sizes[min(len(pending), len(sizes) - 1)], n_total - it)
        pending.append((it, n))
        it += n

    def enqueue_next():
        first, n = pending.pop(0)
        if before_enqueue is not None:
            before_enqueue(first, n)
        eng.run(tune, first, n)
        if on_enqueued is not None:
            on_enqueued(first, n)
        if torch is not None:
            evs = []
            for s_ in streams:
                with torch.cuda.device(s_.device):
                    ev_ = torch.cuda.Event()
                    ev_.record(s_)
                evs.append(ev_)
            marks.append((first + n, evs))

    try:
        depth = 2 if torch is not None else len(pending)
        while pending and len(marks) < depth:
            enqueue_next()
        if torch is None:
            while pending:
                enqueue_next()
            eng.synchronize()
        else:
            reported = 0
            launches_done = 0
            seen = -1
            while marks:
                if callback is not None:
                    at = eng.progress()
                    if at != seen:
                        seen = at
                        callback(trace=None, draw=JobProgress(at, n_total, at < tune, eng.chains, launches_done))
                if all(e_.query() for e_ in marks[0][1]):
                    done = marks.pop(0)[0]
                    launches_done += 1
                    if pending:
                        enqueue_next()
                    if progressbar and done != reported:
                        reported = done
                        _log.info("Sampling %d chains: %d/%d iterations (%s), %.1f s" % (
                            eng.chains, done, n_total, "tuning" if done <= tune else "drawing", time.perf_counter() - t0))
                    continue
                time.sleep(0.0005)
            eng.synchronize()
        return n_total, False
    except KeyboardInterrupt:
        if eng.target.family != _abi.TARGET_EXTERNAL:
            eng.request_stop(True)
        eng.synchronize()
        n_done = min(eng.completed_iterations(), n_total)
        if eng.target.family != _abi.TARGET_EXTERNAL:
            eng.request_stop(False)     # re-armed for whoever keeps the engine (return_engine=True)
        _log.warning("Sampling interrupted after %d of %d iterations; returning the draws so far." % (n_done, n_total))
        return n_done, True


def _run_job_host_step_rand(eng, step, tune, n_total, progressbar, callback=None):
    """The job loop when ``step_rand`` is an arbitrary Python callable (base_hmc.py:154-155): it has to be evaluated
    between iterations, so every iteration is a launch of its own -- adaptation state down, the callable once per chain,
    step sizes up (lmc_engine_set_step_sizes). The compatibility path; StepRandUniform runs inside the kernel.
    ``callback`` is called once per iteration, right after that iteration's launch has been ENQUEUED (the launch is
    asynchronous: ``iteration`` counts launches handed to the device, which runs at most one iteration behind, since the next
    set_step_sizes() reads the adaptation state back) and may raise KeyboardInterrupt like in _run_job."""
    import time

    t0 = time.perf_counter()
    it = 0
    try:
        for it in range(n_total):
            eng.set_step_sizes(step._host_step_sizes(eng, it < tune))
            eng.run(tune, it, 1)
            if callback is not None:
                callback(trace=None, draw=JobProgress(it, n_total, it < tune, eng.chains, it))
            if progressbar and (it + 1) % 100 == 0:
                _log.info("Sampling %d chains: %d/%d iterations, %.1f s" % (eng.chains, it + 1, n_total, time.perf_counter() - t0))
        eng.synchronize()
        return n_total, False
    except KeyboardInterrupt:
        eng.synchronize()
        n_done = min(eng.completed_iterations(), n_total)
        _log.warning("Sampling interrupted after %d of %d iterations; returning the draws so far." % (n_done, n_total))
        return n_done, True
    finally:
        # whoever keeps the engine (return_engine=True) gets it back without the per-chain override, interrupted or not;
        # a failure in here must not mask the error that brought us here (a HIP error usually makes synchronize() fail too)
        try:
            eng.synchronize()
            eng.set_step_sizes(None)
        except Exception as cleanup_err:
            _log.warning("cleanup after the per-iteration job failed: %s" % cleanup_err)


def visible_devices():
    """HIP devices this process can see (lmc_device_count: 0 when there is none)."""
    return int(_abi.load().lmc_device_count())


def _resolve_devices(devices, device, cores, chains, probe):
    """Which GPUs a job of ``chains`` chains runs on -- the reference's ``cores`` logic (sampling.py:117-129: one worker
    per chain, at most ``cores`` at a time) with GPUs in the place of cores.

    * ``devices=[...]`` (HIP ordinals, repeats allowed) or ``devices=N`` (the first N GPUs): exactly those.
    * ``device=k``: that one GPU (what a one-process-per-GPU launcher passes: distributed.sample_distributed).
    * neither: under a multi-process launcher (WORLD_SIZE > 1) GPU LOCAL_RANK; otherwise ``cores=N`` -> the first
      min(N, visible) GPUs, and with ``cores`` unset as many GPUs as the chains fill -- one more GPU per
      ``probe()`` = resident wavefront slots of the sampling kernel, i.e. no GPU is brought in to run below full residency.
    Never more devices than chains."""
    if devices is not None:
        if isinstance(devices, (int, np.integer)):
            devices = list(range(int(devices)))
        devices = [int(d_) for d_ in devices]
        if not devices:
            raise ValueError("devices must name at least one GPU")
    elif device is not None:
        devices = [int(device)]
    elif int(os.environ.get("WORLD_SIZE", "1")) > 1:
        devices = [int(os.environ.get("LOCAL_RANK", "0"))]
    else:
        n_vis = visible_devices()
        if n_vis <= 1:
            devices = [0]
        elif cores is not None:
            devices = list(range(max(1, min(int(cores), n_vis))))
        else:
            slots = probe() or 1024   # (kernels that do not report their residency: a block of 1024 chains per GPU)
            devices = list(range(max(1, min(n_vis, chains // slots))))
    return devices[:max(1, min(len(devices), chains))]

def sample(logp_dlogp_func, model_ndim=None, draws=1000, tune=1000, step=None, init="auto", chains=None,
           cores=None, start=None, progressbar=True, random_seed=None, discard_tuned_samples=True,
           chain_idx=0, callback=None, mp_ctx=None, pickle_backend="pickle", size=None, device=None, devices=None,
           launch_iters=None, return_engine=False, keep_moments=False, stream_results=True, **kwargs):
    """Draw samples with many chains on the MI355X(s) of this node (reference signature: sampling.py:35-53).

    Where the reference fans its chains out over ``cores`` worker processes (sampling.py:124-129,186-201), this fans
    them out over GPUs: the chains are dealt to the devices in contiguous blocks (distributed.chain_block) on the seeds
    of the global chain index space, one engine per device, all driven from this process -- the result is the one-GPU
    result chain for chain. ``cores=N`` means "at most N GPUs"; by default a job takes as many visible GPUs as its chains
    fill (see _resolve_devices); ``devices=[...]`` / ``device=k`` pin the choice.

    Extra keywords: ``size`` (alias of ``model_ndim``), ``device`` / ``devices`` (HIP ordinals), ``launch_iters``
    (iterations per kernel launch; default: the whole run in at most a few launches),
    ``return_engine`` (also return the live Engine -- an EngineGroup on several GPUs -- e.g. to read device pointers),
    ``keep_moments`` (the kernel also keeps every chain's running mean / M2 of the post-warm-up draws:
    ``Engine.moments()``). ``callback(trace=None, draw=JobProgress)`` is called from the wait loop as the job advances
    and may raise KeyboardInterrupt to stop it (sampling.py:272-277 of the reference); ``mp_ctx`` and
    ``pickle_backend`` are accepted and ignored (no worker processes). ``stream_results``: True / "direct" (default) --
    the returned arrays are page-locked host memory, the sampling kernel writes every draw straight into the returned trace
    (lmc_engine_attach_trace) and every launch's statistics are copied under the launches that follow
    (lmc_engine_copy_window_async); "windows" -- the trace stays in HBM and is copied window by window like the statistics;
    False -- everything stays on the device until the job is over, then one blocking copy. The same arrays bit for bit
    (tests/test_gpu_round6.py).
    """
    if model_ndim is None:
        model_ndim = size if size is not None else getattr(logp_dlogp_func, "d", None)
    target = require_device_target(logp_dlogp_func, model_ndim)
    gpu_cap = cores                # the caller's own `cores` (None = not given) caps the number of GPUs
    if cores is None:
        cores = min(4, os.cpu_count() or 1)
    if chains is None:
        chains = max(2, cores)
    seeds = _derive_seeds(random_seed, chains)

    if draws == 0:
        _log.warning("Tuning was enabled throughout the whole trace.")
    elif draws < 500:
        _log.warning("Only {} samples in chain.".format(draws))

    if step is None or start is None:   # sampling.py:148-159 (always builds a NUTS for the start point)
        start_, step_ = init_nuts(target, model_ndim, init=init, random_seed=seeds, **kwargs)
        if step is None:
            step = step_
        if start is None:
            start = start_
    if isinstance(start, (list, tuple)):   # one start point per chain (sampling.py:161-164 accepts a list)
        if len(start) != chains:
            raise ValueError("start list must have one entry per chain")
        start = np.stack([np.asarray(x, dtype="d") for x in start])
    start = np.asarray(start, dtype="d")
    if start.ndim == 1:   # same start for every chain (sampling.py:163-164)
        starts = np.broadcast_to(start, (chains, model_ndim))
    else:
        starts = start
        if starts.shape != (chains, model_ndim):
            raise ValueError("start must have shape (model_ndim,) or (chains, model_ndim)")

    n_total = int(tune) + int(draws)

    def _probe_slots():   # resident wavefront slots of this job's sampling kernel on one GPU (a one-chain engine knows)
        if target.family == _abi.TARGET_EXTERNAL:
            return chains + 1     # a host / torch callable is evaluated on one device: no automatic fan-out
        probe = step._make_engine(1, device=0)
        try:
            return probe.resident_chains()
        finally:
            probe.close()

    devs = _resolve_devices(devices, device, gpu_cap, chains, _probe_slots)
    if len(devs) == 1:
        eng = step._make_engine(chains, device=devs[0])
    else:
        from .distributed import chain_block
        from .engine import EngineGroup

        blocks = [chain_block(chains, k, len(devs)) for k in range(len(devs))]
        made = []
        try:
            for dv, (b_lo, b_hi) in zip(devs, blocks):
                made.append(step._make_engine(b_hi - b_lo, device=dv))
        except BaseException:
            for e_ in made:
                e_.close()
            raise
        eng = EngineGroup(made, blocks)
        _log.info("Sampling %d chains on %d GPUs %s (contiguous blocks of ~%d chains)" % (chains, len(devs), devs, blocks[0][1]))
    try:
        eng.seed(seeds)                       # np.random.seed(random_seed[i]) per chain (sampling.py:496-497)
        eng.set_position(np.ascontiguousarray(starts))
        if keep_moments:
            eng.keep_moments(True)
        eng.reset_tuning()                    # step.reset_tuning(); iter_count = 0 (sampling.py:503-509)
        lo = int(tune) if discard_tuned_samples else 0   # sampling.py:473-476
        host_rand = getattr(step, "_host_step_rand", lambda: None)() is not None
        # Streamed results (the default): the arrays the caller gets are pinned while the first launches run, the sampling
        # kernel writes the draws straight into them ("direct") and every launch's statistics are copied under the launches
        # that follow (_ResultStreamer). Not for jobs that launch per iteration (a host step_rand) or per tick (a torch /
        # Python callable): their results are copied when the job is over.
        mode = {True: "direct", False: None, None: None}.get(stream_results, stream_results)
        if mode not in (None, "direct", "windows"):
            raise ValueError("stream_results must be True / 'direct', 'windows' or False")
        if host_rand or target.family == _abi.TARGET_EXTERNAL or n_total - lo <= 0 or not hasattr(step, "_result_planes"):
            mode = None
        if return_engine and stream_results is True:
            mode = "windows"     # whoever keeps the engine reads the draws where diagnostics want them: in HBM (an explicit "direct" is honoured)
        if stream_results is True and chains * (n_total - lo) * (model_ndim * 8 + 82) < _STREAM_MIN_BYTES:
            mode = None          # a result of a few MiB: pinning twelve arrays and a helper thread cost more than one copy
        eng.reserve(max(n_total, 1), keep_trace=mode != "direct", trace_begin=min(lo, max(n_total - 1, 0)))
        # one launch for the whole job unless asked otherwise: every launch ends with a tail in which the chains with
        # the longest trees run alone (ragged targets: DESIGN.md section 6), so fewer, longer launches are faster
        per_launch = int(launch_iters) if launch_iters else max(1, min(n_total, 4000))
        if not launch_iters:
            # ...except when the chains outnumber the resident wavefront slots. Only a few times over: whole-job launches
            # would run in a few job-long rounds with the last one part empty, while in segments of 100 iterations the
            # engine's sub-block streams keep the slots filled across segment boundaries (+22 % at 8 192 x d=128).
            # Many times over: a launch is also the granularity of Ctrl-C for the chains whose wavefronts have not started
            # yet (a workgroup that starts under a stop request does nothing: interrupting ONE job-long launch of 20 rounds of
            # residency would return no draw at all), so the job is cut into launches of 500 iterations -- few enough that
            # the per-launch tail of ragged targets stays small.
            slots = eng.resident_chains()
            per_dev = -(-chains // len(devs))     # what one GPU holds
            if slots and slots < per_dev < 6 * slots:
                per_launch = min(per_launch, 100)
            elif slots and per_dev >= 6 * slots:
                # (round 5) the engine chooses the LDS plan of a launch when it is enqueued, from iteration 200 on and from what
                # the running chains report (it ignores reports from the first 100): four launches of 100 let the choice settle
                # by iteration 300, the rest of the job runs in launches of 500
                per_launch = [100, 100, 100, 100, 500] if per_launch >= 500 else per_launch
            elif not slots and getattr(getattr(eng, "engines", [eng])[0], "wide", False):
                per_launch = min(per_launch, 200)   # general kernels (one workgroup per chain, a few hundred resident): same reason
        if target.family == _abi.TARGET_EXTERNAL and not launch_iters:
            per_launch = max(n_total, 1)   # ticks: chains never wait for each other inside one request
        streamer = None
        if mode is not None:
            streamer = _ResultStreamer(eng, chains, n_total - lo, lo, model_ndim, step._result_planes(), direct=mode == "direct")
        try:
            if host_rand:
                n_done, interrupted = _run_job_host_step_rand(eng, step, int(tune), n_total, progressbar, callback)
            else:
                n_done, interrupted = _run_job(eng, int(tune), n_total, per_launch, progressbar, callback,
                                               on_enqueued=streamer.window if streamer is not None else None,
                                               before_enqueue=streamer.before_launch if streamer is not None else None)
        except BaseException:
            # the job failed: wait for the pinning thread and the copies so that nothing writes the arrays after they are
            # dropped -- without letting a second failure in here mask the error that is on its way up
            if streamer is not None:
                try:
                    streamer.finish()
                except Exception as cleanup_err:
                    _log.warning("waiting for the streamed results after a failed job failed too: %s" % cleanup_err)
            raise
        streamed = streamer.finish() if streamer is not None else None
        raise_for_status(eng.status())

        n_out = max(n_done - lo, 0)
        if n_out > 0 and streamed is not None:
            # (an interrupted job returns the iterations every chain completed: a prefix of the arrays)
            trace = streamed.trace if n_out == streamed.n_out else streamed.trace[:, :n_out]
            stats = {name: streamed.stats[name][:, :n_out, None].astype(dtype, copy=False) for name, dtype in step.stats_dtypes[0].items()}
        elif n_out > 0:
            trace = eng.trace(lo, n_out)
            raw = step._stats_from_engine(eng, lo, n_out)
            stats = {name: raw[name][:, :, None].astype(dtype) for name, dtype in step.stats_dtypes[0].items()}
        else:
            trace = np.zeros((chains, 0, model_ndim))
            stats = {name: np.zeros((chains, 0, 1), dtype=dtype) for name, dtype in step.stats_dtypes[0].items()}

        # leave the step object as the reference's sequential driver leaves it: state of the LAST chain
        if not interrupted:
            step.tune = False if n_total > 0 else step.tune
        elif n_done > 0:
            step.tune = n_done <= int(tune)          # stop_tuning() happens at iteration index `tune` (sampling.py:510-511)
        step.iter_count = n_done
        step.step_adapt._pull(eng, chains - 1)
        step.potential._pull(eng, chains - 1)
        ct = eng.counters()
        step._samples_after_tune += int(ct[:, _abi.CT_SAMPLES_AFTER_TUNE].sum())
        step._num_divs_sample += int(ct[:, _abi.CT_DIVS_AFTER_TUNE].sum())
        if hasattr(step, "_reached_max_treedepth"):
            step._reached_max_treedepth += int(ct[:, _abi.CT_REACHED_MAX_TREEDEPTH].sum())
        if n_done > int(tune):
            tail = eng.stat_f64(_abi.STAT_ACCEPT, int(tune), n_done - int(tune))
            step.step_adapt._tuned_stats = list(tail[chains - 1])
    except BaseException:   # KeyboardInterrupt / SystemExit included: never leak the engine (its HBM, its streams)
        eng.close()
        raise
    if return_engine:
        return trace, stats, eng
    eng.close()
    return trace, stats
