"""Device log-densities: what ``logp_dlogp_func`` is in this framework.

The reference takes any Python callable ``q -> (logp, dlogp)`` and calls it once per leapfrog
step (/root/reference/littlemcmc/integration.py:40,62,115). On the GPU the callable is a
``__device__`` functor compiled into the transition kernel (littlemcmc_amd/csrc/lmc_targets.hpp).
The objects here name such a functor plus its parameters; they are ALSO callable with the
reference's signature -- the call evaluates the device functor on the GPU for one point -- so
they can be passed wherever the reference expects ``logp_dlogp_func``.

Arbitrary densities are supported three ways: :class:`UserTarget` compiles a user-supplied HIP snippet into the
kernels (the fast path), :class:`TorchTarget` takes a batched torch callable on the GPU, and a plain per-point
Python callable -- the reference's own plug-in signature -- is wrapped in :class:`CallableTarget`: the sampler still
runs in the HIP tick kernel, the user's function is evaluated on the host between ticks (compatibility path).
"""
import hashlib
import os

import numpy as np

from . import _abi, _build


class DeviceTarget:
    """Base: a device functor family + its parameter vector."""

    family = None
    lib_path = None  # default library

    def __init__(self, d, params=()):
        self.d = int(d)
        self.params = np.ascontiguousarray(params, dtype=np.float64)
        self._eval_engine = None

    # reference plug-in signature, evaluated on the GPU
    def __call__(self, q):
        from .engine import Engine

        if self._eval_engine is None:
            self._eval_engine = Engine(self, chains=1)
        logp, grad = self._eval_engine.logp_dlogp(np.asarray(q, dtype=np.float64).reshape(1, self.d))
        return self._wrap_logp(logp[0]), grad[0]

    def _wrap_logp(self, logp):
        return np.float64(logp)

    def __getstate__(self):  # picklable like the reference requires (docs/tutorials/quickstart.rst:42-46)
        st = dict(self.__dict__)
        st["_eval_engine"] = None
        return st


class StdNormal(DeviceTarget):
    """logp = -1/2 sum q^2."""

    family = _abi.TARGET_STD_NORMAL

    def __init__(self, d):
        super().__init__(d)


class DiagGaussian(DeviceTarget):
    """Independent Gaussian with precisions ``prec`` (1/sigma^2)."""

    family = _abi.TARGET_DIAG_GAUSSIAN

    def __init__(self, prec):
        prec = np.ascontiguousarray(prec, dtype=np.float64)
        super().__init__(prec.shape[0], prec)

    @classmethod
    def ill_conditioned(cls, d, kappa=1e4):
        i = np.arange(d, dtype=np.float64)
        return cls(1.0 / (kappa ** (i / max(d - 1, 1))))


class AR1(DeviceTarget):
    """Stationary AR(1) Gaussian with unit marginal variances (tridiagonal precision)."""

    family = _abi.TARGET_AR1

    def __init__(self, d, rho=0.9):
        c = 1.0 / (1.0 - rho * rho)
        super().__init__(d, [c, (1.0 + rho * rho) * c, -rho * c])
        self.rho = float(rho)


class Funnel(DeviceTarget):
    """Neal's funnel: q_0 ~ N(0, 9), q_i | q_0 ~ N(0, exp(q_0))."""

    family = _abi.TARGET_FUNNEL

    def __init__(self, d):
        super().__init__(d)


class Normal1D(DeviceTarget):
    """The reference's 1-D test target (tests/test_utils.py:19-28): logp has shape (1,)."""

    family = _abi.TARGET_NORMAL1D

    def __init__(self, loc=0.0, scale=1.0):
        super().__init__(1, [loc, scale])

    def _wrap_logp(self, logp):
        return np.array([logp])


class UserTarget(DeviceTarget):
    """A user-written device log-density, compiled at run time and linked into the kernels.

    ``source`` must define, in namespace ``lmc``::

        template <int NS> struct UserTarget {
            static constexpr bool kLanePartial = false;
            template <class Team> __device__ void init(Team& tm, const double* params, int d);
            template <class Team> __device__ double logp_grad(Team& tm, const double (&q)[NS], double (&g)[NS]) const;
        };

    following the thread-distributed contract documented in csrc/lmc_targets.hpp.

    ``jit="hiprtc"`` (default): when an engine is created, the three kernels that depend on the functor -- the
    transition kernel of the engine's shape, the trajectory and the log-density unit kernels -- are compiled in
    process with hiprtc (2-4 s -- one-wavefront shapes compile the sampling kernel twice, once per LDS plan, so that the
    engine can choose the plan per launch as for the built-in densities; cached by content under ``_user_targets/``, which
    is never shipped to another machine) and handed to the stock library
    (``lmc_engine_load_user_kernels``); no compiler is needed on the machine. Diagonal mass matrices.
    ``jit="hipcc"``: a private build of the whole library around the functor (about 10 s, needs hipcc); this is the
    path for dense mass matrices with a user density."""

    family = _abi.TARGET_USER

    def __init__(self, d, source, params=(), jit="hiprtc"):
        super().__init__(d, params)
        self.source = source
        if jit not in ("hiprtc", "hipcc"):
            raise ValueError("jit must be 'hiprtc' or 'hipcc'")
        self.jit = jit
        self._code = {}   # (unit_ns, run_ns, run_w) -> (code object bytes, lowered names)
        if jit == "hipcc":
            self.lib_path = self._build_private_library()

    # ---- hiprtc: the target-dependent kernels only ---------------------------------------------------------
    def _digest(self, extra=""):
        # the cache key covers everything the binary depends on: the snippet, the kernel sources and the compiler flags
        h = hashlib.sha256(self.source.encode())
        for path in _build.sources():
            with open(path, "rb") as fh:
                h.update(fh.read())
        h.update(" ".join(_build.HIPCC_FLAGS).encode())
        h.update(extra.encode())
        return h.hexdigest()[:16]

    def kernels_for(self, unit_ns, run_ns, run_w, general=False):
        """(code object, run name, trajectory name, logp name, run name under LDS plan 1 or None) for one engine shape,
        compiled once and cached."""
        key = (int(unit_ns), int(run_ns), int(run_w), int(bool(general)))
        if key in self._code:
            return self._code[key]
        if key[3]:   # the general kernels (csrc/lmc_wide.hpp): model_ndim > 1024, dense mass matrices, float64 masses
            names = ["lmc::run_wide_kernel<%d, %d, lmc::UserTarget>" % (key[1], key[2]),
                     "lmc::wide_trajectory_kernel<%d, %d, lmc::UserTarget>" % (key[0], key[2]),
                     "lmc::wide_logp_kernel<%d, %d, lmc::UserTarget>" % (key[0], key[2])]
            tu = ('#include "lmc_wide.hpp"\n' + self.source +
                  "\nnamespace lmc {\n"
                  "template __global__ void run_wide_kernel<%d, %d, UserTarget>(ChainArrays, DenseArrays, SamplerParams, const double*);\n"
                  "template __global__ void wide_trajectory_kernel<%d, %d, UserTarget>(ChainArrays, DenseArrays, const double*, const double*, "
                  "const double*, int, int, double, int, int, double*, double*, double*, double*, double*, double*);\n"
                  "template __global__ void wide_logp_kernel<%d, %d, UserTarget>(ChainArrays, const double*, const double*, double*, double*);\n"
                  "}\n" % (key[1], key[2], key[0], key[2], key[0], key[2]))
        else:
            names = ["lmc::run_kernel<%d, %d, lmc::UserTarget>" % (key[1], key[2]),
                     "lmc::trajectory_kernel<%d, lmc::UserTarget>" % key[0],
                     "lmc::logp_kernel<%d, lmc::UserTarget>" % key[0]]
            tu = ('#include "lmc_sampler.hpp"\n#include "lmc_unit_kernels.hpp"\n' + self.source +
                  "\nnamespace lmc {\n"
                  "template __global__ void run_kernel<%d, %d, UserTarget>(ChainArrays, SamplerParams, const double*);\n"
                  "template __global__ void trajectory_kernel<%d, UserTarget>(ChainArrays, const double*, const double*, const double*, "
                  "int, int, double, int, int, double*, double*, double*, double*, double*, double*);\n"
                  "template __global__ void logp_kernel<%d, UserTarget>(ChainArrays, const double*, const double*, double*, double*);\n"
                  "}\n" % (key[1], key[2], key[0], key[0]))
            if key[2] == 1:   # one-wave kernels: also the deep-tree LDS plan (csrc/lmc_sampler.hpp: run_kernel<.., PL = 1>)
                names.append("lmc::run_kernel<%d, 1, lmc::UserTarget, 0, 1>" % key[1])
                tu += ("namespace lmc {\ntemplate __global__ void run_kernel<%d, 1, UserTarget, 0, 1>(ChainArrays, SamplerParams, "
                       "const double*);\n}\n" % key[1])
        cache = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_user_targets")
        os.makedirs(cache, exist_ok=True)
        path = os.path.join(cache, "user_%s_%d_%d_%d%s.hsaco" % (self._digest("hiprtc"), key[0], key[1], key[2], "g" if key[3] else ""))
        lowered = None
        if os.path.exists(path) and os.path.exists(path + ".names"):
            with open(path, "rb") as fh:
                code = fh.read()
            with open(path + ".names") as fh:
                lowered = [ln for ln in fh.read().split("\n") if ln]
        else:
            code, lowered = _hiprtc_compile(tu, names)
            tmp = "%s.%d.tmp" % (path, os.getpid())     # several ranks may compile the same target at once
            with open(tmp, "wb") as fh:
                fh.write(code)
            os.replace(tmp, path)
            with open(tmp, "w") as fh:
                fh.write("\n".join(lowered))
            os.replace(tmp, path + ".names")
        self._code[key] = (code, lowered[0], lowered[1], lowered[2], lowered[3] if len(lowered) > 3 else None)
        return self._code[key]

    def _attach(self, engine):
        """Called by Engine after lmc_engine_create: compile (or fetch) the kernels of the engine's shape and load them."""
        if self.jit != "hiprtc":
            return
        import ctypes as C

        ns, rns, rw = C.c_int32(), C.c_int32(), C.c_int32()
        engine._check(engine._lib.lmc_engine_kernel_shape(engine._h, C.byref(ns), C.byref(rns), C.byref(rw)))
        general = bool(engine._lib.lmc_engine_uses_general_kernels(engine._h))
        code, run, traj, logp, run1 = self.kernels_for(ns.value, rns.value, rw.value, general)
        buf = C.create_string_buffer(code, len(code))
        engine._check(engine._lib.lmc_engine_load_user_kernels(engine._h, buf, run.encode(), traj.encode(), logp.encode()))
        if run1 is not None:   # the sampling kernel under the deep-tree LDS plan: the engine may now choose per launch
            engine._check(engine._lib.lmc_engine_load_user_run_plan1(engine._h, run1.encode()))

    def __getstate__(self):
        st = super().__getstate__()
        st["_code"] = {}
        return st

    # ---- hipcc: private build of the library ------------------------------------------------------------------
    def _build_private_library(self):
        digest = self._digest("hipcc")
        cache = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_user_targets")
        os.makedirs(cache, exist_ok=True)
        header = os.path.join(cache, "user_%s.hpp" % digest)
        lib = os.path.join(cache, "liblmc_hip_user_%s.so" % digest)
        if not os.path.exists(lib):
            # several ranks may build the same target at once: private temporaries, atomic rename into place
            tmp_tag = "%d_%s" % (os.getpid(), digest)
            tmp_header = os.path.join(cache, "user_%s.tmp.hpp" % tmp_tag)
            tmp_lib = os.path.join(cache, "liblmc_hip_user_%s.tmp.so" % tmp_tag)
            with open(tmp_header, "w") as fh:
                fh.write(self.source)
            try:
                _build.build(out=tmp_lib, extra_flags=["-DLMC_USER_TARGET_HEADER=\"%s\"" % tmp_header, "-DLMC_ONLY_USER"],
                             force=True)
                os.replace(tmp_lib, lib)
                os.replace(tmp_header, header)
            finally:
                for f in (tmp_lib, tmp_header):
                    if os.path.exists(f):
                        os.remove(f)
        return lib


def _hiprtc_compile(source, name_expressions):
    """Compile a translation unit for gfx950 with hiprtc -> (code object bytes, lowered kernel names)."""
    import ctypes as C

    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    try:
        rtc = C.CDLL(os.path.join(rocm, "lib", "libhiprtc.so"))
    except OSError as err:
        raise _abi.HipLibraryError("cannot load libhiprtc.so (%s): UserTarget(jit='hiprtc') needs the ROCm runtime compiler; "
                                   "use jit='hipcc' where hipcc is installed" % err)
    rtc.hiprtcGetErrorString.restype = C.c_char_p
    prog = C.c_void_p()

    def check(rc, what):
        if rc != 0:
            n = C.c_size_t()
            rtc.hiprtcGetProgramLogSize(prog, C.byref(n))
            log = C.create_string_buffer(n.value + 1)
            rtc.hiprtcGetProgramLog(prog, log)
            raise RuntimeError("hiprtc %s failed (%s):\n%s" % (what, rtc.hiprtcGetErrorString(rc).decode(), log.value.decode()[-4000:]))

    check(rtc.hiprtcCreateProgram(C.byref(prog), source.encode(), b"lmc_user_target.hip", 0, None, None), "create")
    try:
        for n in name_expressions:
            check(rtc.hiprtcAddNameExpression(prog, n.encode()), "name expression")
        flags = [f for f in _build.HIPCC_FLAGS if f not in ("-fPIC", "-shared")]
        opts = [f.encode() for f in flags] + [("-I" + _build.CSRC).encode(), ("-I" + os.path.join(rocm, "include")).encode()]
        arr = (C.c_char_p * len(opts))(*opts)
        check(rtc.hiprtcCompileProgram(prog, len(opts), arr), "compile")
        size = C.c_size_t()
        check(rtc.hiprtcGetCodeSize(prog, C.byref(size)), "code size")
        code = C.create_string_buffer(size.value)
        check(rtc.hiprtcGetCode(prog, code), "code")
        lowered = []
        for n in name_expressions:
            p = C.c_char_p()
            check(rtc.hiprtcGetLoweredName(prog, n.encode(), C.byref(p)), "lowered name")
            lowered.append(p.value.decode())
        return code.raw, lowered
    finally:
        rtc.hiprtcDestroyProgram(C.byref(prog))


class TorchTarget(DeviceTarget):
    """A log-density given as a BATCHED torch callable on the GPU, for densities that are easier to write (or
    differentiate) in torch than as a HIP functor (reference cookbook: docs/_static/scripts/
    sample_pytorch_logp_dlogp_func.py).

        fn(q: float64 cuda tensor [chains, d]) -> (logp [chains], dlogp [chains, d])

    The sampler then runs as a resumable kernel (csrc/lmc_tick.hpp): every "tick" each chain hands over the one
    point it needs the density at, ``fn`` evaluates all chains at once, and the kernel carries every chain on to
    its next evaluation -- finishing leapfrogs, building trees, adapting, starting new iterations -- without
    the chains ever waiting for each other. Nothing is computed on the CPU.

    ``TorchTarget.from_logp(d, logp_fn)`` builds the gradient with autograd from ``logp_fn(q) -> [chains]``.
    Limits: d <= 16 384 with diagonal mass matrices (beyond 1 024 the chain is a workgroup of 16 wavefronts:
    csrc/lmc_tick.hpp: tick_step with TickWideShape, lmc_wide.hip); dense mass matrices (QuadPotentialFull*, init="adapt_full") as for
    the fused kernels up to d = 256. A fused ``UserTarget`` is several times faster (no HBM round trip of
    the chain state per leapfrog); this is the path for "cannot write device code".
    """

    family = _abi.TARGET_EXTERNAL

    def __init__(self, d, fn, graph=False):
        super().__init__(d)
        if not callable(fn):
            raise TypeError("fn must be callable: q[chains, d] -> (logp[chains], dlogp[chains, d])")
        self.fn = fn
        # graph=True: capture fn once into a HIP graph (torch.cuda.CUDAGraph) and replay it every tick. The points
        # always live in the same engine-owned buffer, so the capture is valid for the whole run; it removes the
        # host-side launch cost of fn's kernels (what bounds small batches). fn must be capturable: no host
        # synchronisation, no data-dependent shapes.
        self.graph = bool(graph)

    @classmethod
    def from_logp(cls, d, logp_fn, graph=False):
        import torch

        def fn(q):
            with torch.enable_grad():
                x = q.detach().requires_grad_(True)
                lp = logp_fn(x)
                (g,) = torch.autograd.grad(lp.sum(), x)
            return lp.detach(), g

        return cls(d, fn, graph=graph)

    @classmethod
    def from_pointwise(cls, d, fn, graph=False):
        """``fn(q: tensor[d]) -> (logp: scalar tensor, dlogp: tensor[d])`` written in torch ops for ONE point -- the
        reference's plug-in signature (integration.py:40,62,115) -- batched over the chains with ``torch.vmap``."""
        import torch

        batched = torch.vmap(fn)
        return cls(d, lambda q: batched(q), graph=graph)

    def evaluate(self, q):
        """fn on a [chains, d] tensor, results checked and made contiguous float64."""
        import torch

        logp, grad = self.fn(q)
        if logp.shape != (q.shape[0],) and logp.numel() == q.shape[0]:
            logp = logp.reshape(q.shape[0])
        if logp.shape != (q.shape[0],) or grad.shape != q.shape:
            raise ValueError("TorchTarget fn must return (logp[chains], dlogp[chains, d]); got %s and %s for q %s"
                             % (tuple(logp.shape), tuple(grad.shape), tuple(q.shape)))
        if not (logp.is_cuda and grad.is_cuda):
            raise TypeError("TorchTarget fn must return CUDA (ROCm) tensors: there is no CPU path")
        return logp.to(torch.float64).contiguous(), grad.to(torch.float64).contiguous()

    def __call__(self, q):   # reference plug-in signature, one point
        import torch

        x = torch.as_tensor(np.asarray(q, dtype=np.float64).reshape(1, self.d), device="cuda")
        logp, grad = self.evaluate(x)
        return np.float64(logp[0].item()), grad[0].cpu().numpy()

    def __getstate__(self):
        return dict(self.__dict__)


class CallableTarget(TorchTarget):
    """The reference's own plug-in, unchanged: a per-point Python callable ``fn(q: ndarray[d]) -> (logp, dlogp[d])``
    (integration.py:40,62,115; e.g. tests/test_utils.py:19-28 or docs/_static/scripts/
    sample_pytorch_logp_dlogp_func.py:35-45 with CPU tensors).

    The sampler itself -- leapfrog, trees, adaptation, RNG -- still runs in the HIP tick kernel (csrc/lmc_tick.hpp);
    only the user's density is evaluated where the user's code lives: every tick the points of all chains are copied
    to the host, ``fn`` is called once per chain, and the values go back for the next tick. This is the compatibility
    path (C1-sized jobs: a handful of chains); a :class:`UserTarget` or a batched :class:`TorchTarget` is what the
    many-chain configurations need. ``sample()`` / ``NUTS`` / ``HamiltonianMC`` wrap a plain callable in it."""

    tick_poll = 1   # look at the number of unfinished chains after every tick: no wasted host evaluations

    def __init__(self, d, fn):
        if not callable(fn):
            raise TypeError("fn must be callable: q[d] -> (logp, dlogp[d])")
        self.pointwise = fn
        super().__init__(d, self._batched, graph=False)

    @staticmethod
    def _host(x):
        if hasattr(x, "detach"):   # a torch tensor (CPU or device)
            x = x.detach().cpu().numpy()
        return np.asarray(x, dtype=np.float64)

    def _batched(self, q):
        import torch

        host = q.detach().cpu().numpy()
        logp = np.empty(host.shape[0])
        grad = np.empty_like(host)
        for c in range(host.shape[0]):
            lp, dlp = self.pointwise(host[c].copy())
            logp[c] = self._host(lp).reshape(-1)[0]          # scalar or shape-(1,) array (tests/test_utils.py:27-28)
            grad[c] = self._host(dlp).reshape(self.d)
        return torch.from_numpy(logp).to(q.device), torch.from_numpy(grad).to(q.device)

    def __call__(self, q):   # the callable itself, as the reference would call it
        return self.pointwise(np.asarray(q, dtype=np.float64))


_SEPARABLE_TEMPLATE = r"""
#include "lmc_team.hpp"
namespace lmc {
// generated by littlemcmc_amd.targets.UserTarget.separable: logp(q) = sum_e term(q_e, e; P)
template <int NS>
struct UserTarget {
    static constexpr bool kLanePartial = true;
    const double* P;
    int d;
    template <class Team>
    __device__ void init(Team&, const double* params, int d_) { P = params; d = d_; }
    template <class Team>
    __device__ double logp_grad_partial(Team& tm, const double (&qv)[NS], double (&g)[NS]) const {
        double part = 0.0;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int e = tm.tid() * NS + s;
            if (e < d) {
                const double q = qv[s];
                %(prelude)s
                g[s] = (%(grad)s);
                part += (%(logp)s);
            } else {
                g[s] = 0.0;
            }
        }
        return part;
    }
    template <class Team>
    __device__ double logp_grad(Team& tm, const double (&qv)[NS], double (&g)[NS]) const {
        return tm.sum(logp_grad_partial(tm, qv, g));
    }
};
}  // namespace lmc
"""


def _separable(cls, d, logp, grad, params=(), prelude=""):
    """A product density prod_e f(q_e): give the per-coordinate log term and its derivative as C expressions in
    ``q`` (the coordinate), ``e`` (its index) and ``P`` (the parameter vector), e.g. a Student-t with P[0] = nu:

        UserTarget.separable(d, logp="-0.5*(P[0]+1.0)*log1p(q*q/P[0])", grad="-(P[0]+1.0)*q/(P[0]+q*q)", params=[4.0])
    """
    src = _SEPARABLE_TEMPLATE % {"logp": logp, "grad": grad, "prelude": prelude}
    return cls(d, src, params=params)


UserTarget.separable = classmethod(_separable)


def require_device_target(logp_dlogp_func, model_ndim=None):
    """What the step methods and ``sample()`` accept as ``logp_dlogp_func``: a device functor (inlined into the
    leapfrog kernel), a batched torch callable (TorchTarget) or -- the reference's own signature -- a plain per-point
    Python callable, which is wrapped in a :class:`CallableTarget` (sampler on the GPU, density evaluated by the
    caller's code between ticks). Anything else is rejected, loudly."""
    if not isinstance(logp_dlogp_func, DeviceTarget):
        if callable(logp_dlogp_func) and model_ndim is not None:
            return CallableTarget(int(model_ndim), logp_dlogp_func)
        raise TypeError(
            "logp_dlogp_func must be a littlemcmc_amd.targets.DeviceTarget (StdNormal, DiagGaussian, AR1, Funnel, "
            "Normal1D, UserTarget, TorchTarget) or a callable q[d] -> (logp, dlogp[d]) together with model_ndim; got %r"
            % (logp_dlogp_func,))
    if model_ndim is not None and int(model_ndim) != logp_dlogp_func.d:
        raise ValueError("model_ndim=%s does not match the target's dimension %d" % (model_ndim, logp_dlogp_func.d))
    return logp_dlogp_func
