"""Cross-chain convergence diagnostics: split R-hat and multi-chain effective sample size.

The reference has none (ArviZ appears only in a docs recipe, docs/tutorials/framework_cookbook.rst:201-213;
SURVEY.md section 0.9): the definitions below are this build's own -- split R-hat and the Geyer
initial-monotone-sequence ESS of the Stan reference manual, optionally on rank-normalised draws (Vehtari et al.
2021: z = Phi^-1((rank - 3/8) / (S + 1/4)) over the pooled draws of a dimension) -- and are pinned against a
plain-numpy restatement (oracle/diagnostics_oracle.py).

Everything reduces to per-dimension *sufficient statistics that add over chains* (and chain halves, and ranks):

    n_chains, sum_c mean_c, sum_c mean_c^2, sum_c var_c, sum_c acov_c[t]

For draws in HBM (``Engine.trace_device_ptr()`` through ``trace_tensor``) they come from the HIP kernel
``lmc_diag_chain_stats`` (csrc/lmc_diag.hip), 16 lags per pass over the trace; passes continue until Geyer's initial
positive sequence has ended in every dimension (one pass for well-mixing NUTS chains), each pass followed by ONE
small all-reduce (RCCL on GPUs) of its (3 + 16) x d block. CPU tensors (the gloo tests of the reduction logic) take
the same statistics from an FFT in torch tensor code."""
import ctypes
import math

import torch

LAGS_PER_PASS = 16


def split_chains(x):
    """[chains, draws, d] -> [2*chains, draws//2, d] (second half of every chain becomes its own chain)."""
    c, n, d = x.shape
    h = n // 2
    return torch.cat([x[:, :h], x[:, n - h:]], dim=0)


def _hip_chain_stats(x, t0, n, lag0):
    """One pass of lmc_diag_chain_stats over x[chains, draws, d] (float64, contiguous, on a ROCm device):
    -> [3 + 16, d] float64 tensor on the same device."""
    from . import _abi

    lib = _abi.load()
    c, stride, d = x.shape
    out = torch.empty((3 + LAGS_PER_PASS, d), dtype=torch.float64, device=x.device)
    with torch.cuda.device(x.device):
        stream = torch.cuda.current_stream(x.device).cuda_stream
        rc = lib.lmc_diag_chain_stats(ctypes.c_void_p(x.data_ptr()), c, stride, d, int(t0), int(n), int(lag0),
                                      ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(stream))
    if rc != 0:
        raise RuntimeError("lmc_diag_chain_stats failed (status %d)" % rc)
    return out


def _torch_chain_stats(x, t0, n, lag0):
    """Host mirror of the kernel's statistics for CPU tensors: FFT autocovariances of the lags [lag0, lag0 + 16)."""
    blk = x[:, t0:t0 + n].to(torch.float64)
    d = blk.shape[2]
    out = torch.zeros((3 + LAGS_PER_PASS, d), dtype=torch.float64, device=x.device)
    if blk.shape[0] == 0:
        return out
    mean = blk.mean(dim=1)
    cen = blk - mean[:, None, :]
    nfft = 1 << (2 * n - 1).bit_length()
    f = torch.fft.rfft(cen, n=nfft, dim=1)
    acov = torch.fft.irfft(f.real ** 2 + f.imag ** 2, n=nfft, dim=1)[:, :n] / n       # [c, n, d], biased
    out[0] = mean.sum(dim=0)
    out[1] = (mean ** 2).sum(dim=0)
    if lag0 == 0:
        out[2] = acov[:, 0].sum(dim=0) * (n / (n - 1.0))
    hi = min(lag0 + LAGS_PER_PASS, n)
    if hi > lag0:
        out[3:3 + hi - lag0] = acov[:, lag0:hi].sum(dim=0)
    return out


def chain_stats_pass(x, ranges, lag0):
    """Statistics block [3 + 16, d] of one pass, summed over the sub-series ``ranges`` = [(t0, n), ...] of every
    chain of x (the two halves for split diagnostics)."""
    if x.shape[0] == 0:
        return torch.zeros((3 + LAGS_PER_PASS, x.shape[2]), dtype=torch.float64, device=x.device)
    if x.is_cuda:
        if x.dtype != torch.float64 or not x.is_contiguous():
            x = x.to(torch.float64).contiguous()
        fn = _hip_chain_stats
    else:
        fn = _torch_chain_stats
    tot = None
    for t0, n in ranges:
        blk = fn(x, t0, n, lag0)
        tot = blk if tot is None else tot + blk
    return tot


def _all_reduce(t, group=None, reduce_device=None):
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return t
    home = t.device
    r = t.clone() if reduce_device is None else t.to(reduce_device)
    dist.all_reduce(r, op=dist.ReduceOp.SUM, group=group)
    return r.to(home)


def _geyer_ended(stats):
    """True when the initial positive sequence has ended within the available lags in every dimension."""
    m, n = float(stats["n_chains"]), float(stats["n_draws"])
    w = stats["sum_var"] / m
    gmean = stats["sum_mean"] / m
    b_over_n = (stats["sum_mean_sq"] - m * gmean ** 2) / (m - 1.0) if m > 1 else torch.zeros_like(w)
    var_plus = w * (n - 1.0) / n + b_over_n
    acov = stats["sum_acov"] / m
    rho = 1.0 - (w[None, :] - acov * (n / (n - 1.0))) / var_plus[None, :]
    rho[0] = 1.0
    T = rho.shape[0] - (rho.shape[0] % 2)
    pairs = rho[0:T:2] + rho[1:T:2]
    return bool(((pairs <= 0).any(dim=0) | ~torch.isfinite(pairs).all(dim=0)).all())


def sufficient_stats(x, split=True, max_lag=None, group=None, reduce_device=None):
    """Reduced (over chains, halves and ranks) sufficient statistics of x[chains, draws, d] (this rank's block)."""
    c, n_all, d = x.shape
    if split:
        h = n_all // 2
        ranges, n, halves = [(0, h), (n_all - h, h)], h, 2
    else:
        ranges, n, halves = [(0, n_all)], n_all, 1
    if n < 4:
        raise ValueError("need at least %d draws per chain" % (4 * halves))
    limit = n if max_lag is None else min(int(max_lag), n)
    dev = x.device
    nch = _all_reduce(torch.tensor([float(c * halves)], dtype=torch.float64, device=dev), group, reduce_device)
    stats = {"n_chains": nch[0], "n_draws": torch.tensor(float(n), dtype=torch.float64, device=dev)}
    acov_blocks = []
    lag0 = 0
    while True:
        blk = _all_reduce(chain_stats_pass(x, ranges, lag0), group, reduce_device)   # the pass's one collective
        if lag0 == 0:
            stats["sum_mean"], stats["sum_mean_sq"], stats["sum_var"] = blk[0], blk[1], blk[2]
        acov_blocks.append(blk[3:])
        lag0 += LAGS_PER_PASS
        stats["sum_acov"] = torch.cat(acov_blocks, dim=0)[:limit]
        if lag0 >= limit or _geyer_ended(stats):
            break
    stats["lag_passes"] = len(acov_blocks)
    return stats


def finalize(stats):
    """R-hat[d] and ESS[d] from (reduced) sufficient statistics."""
    m = float(stats["n_chains"])
    n = float(stats["n_draws"])
    w = stats["sum_var"] / m                                              # mean within-chain variance
    gmean = stats["sum_mean"] / m
    if m > 1:
        b_over_n = (stats["sum_mean_sq"] - m * gmean ** 2) / (m - 1.0)    # variance of chain means
    else:
        b_over_n = torch.zeros_like(w)
    var_plus = w * (n - 1.0) / n + b_over_n
    rhat = torch.sqrt(var_plus / w)
    acov = stats["sum_acov"] / m                                          # [T, d]
    rho = 1.0 - (w[None, :] - acov * (n / (n - 1.0))) / var_plus[None, :]
    rho[0] = 1.0
    T, d = rho.shape
    if T % 2:
        rho = rho[:-1]
        T -= 1
    pairs = rho[0::2] + rho[1::2]                                         # Geyer P_t, [T/2, d]
    positive = torch.cumprod((pairs > 0).to(pairs.dtype), dim=0)          # initial positive sequence
    pairs = torch.nan_to_num(pairs, nan=0.0) * positive
    pairs = torch.cummin(pairs, dim=0).values                             # initial monotone sequence
    tau = -1.0 + 2.0 * pairs.sum(dim=0)
    tau = torch.clamp(tau, min=1.0 / math.log10(max(m * n, 10.0)))
    ess = m * n / tau
    return {"rhat": rhat, "ess": ess, "mean": gmean, "var": var_plus, "n_chains": m, "n_draws": n,
            "lag_passes": stats.get("lag_passes", 0)}


def rank_normalize(x, chunk_dims=8):
    """z-scores of the pooled ranks of every dimension (Vehtari et al. 2021, eq. 14; average ranks for ties are not
    needed for continuous draws): z = Phi^-1((rank - 3/8) / (S + 1/4)), S = chains x draws. The ranks pool the chains
    of the calling rank (a multi-GPU job normalises each rank's block on its own and pools the z-scores)."""
    c, n, d = x.shape
    S = c * n
    out = torch.empty((c, n, d), dtype=torch.float64, device=x.device)
    for lo in range(0, d, chunk_dims):
        blk = x[:, :, lo:lo + chunk_dims].reshape(S, -1).to(torch.float64)
        order = torch.argsort(blk, dim=0)
        ranks = torch.empty_like(order)
        ar = torch.arange(1, S + 1, device=x.device, dtype=order.dtype)[:, None].expand_as(order)
        ranks.scatter_(0, order, ar)
        p = (ranks.to(torch.float64) - 0.375) / (S + 0.25)
        out[:, :, lo:lo + chunk_dims] = (math.sqrt(2.0) * torch.erfinv(2.0 * p - 1.0)).reshape(c, n, -1)
    return out


def summarize(x, split=True, max_lag=None, group=None, reduce_device=None, rank_normalized=False, chunk=None):
    """x[chains, draws, d] (this rank's chain block) -> dict(rhat[d], ess[d], mean[d], var[d]) over ALL ranks.
    ``rank_normalized=True``: the rank-normalised split-R-hat / bulk ESS (diagnostics of the z-scores)."""
    if rank_normalized and x.shape[0] > 0:
        x = rank_normalize(x)
    out = finalize(sufficient_stats(x, split=split, max_lag=max_lag, group=group, reduce_device=reduce_device))
    out["definition"] = ("rank-normalised " if rank_normalized else "") + (
        "split-R-hat / Geyer initial-monotone-sequence ESS" if split else "R-hat / Geyer initial-monotone-sequence ESS")
    return out


def rhat_from_moments(mean, m2, n, group=None, reduce_device=None):
    """(Non-split) R-hat[d] from per-chain running moments -- mean[chains, d], m2[chains, d] (sum of squared
    deviations), n[chains] equal draws per chain -- reduced over ranks with one all-reduce of
    {chains, draws, sum mean, sum mean^2, sum var}. This is the trace-free diagnostic of SURVEY.md section 8e."""
    mean = torch.as_tensor(mean, dtype=torch.float64)
    m2 = torch.as_tensor(m2, dtype=torch.float64)
    nt = torch.as_tensor(n).to(torch.float64)
    d = mean.shape[1]
    nd_local = float(nt.mean()) if nt.numel() else 2.0
    blk = torch.zeros((3, d), dtype=torch.float64, device=mean.device)
    head = torch.zeros((2,), dtype=torch.float64, device=mean.device)
    if mean.shape[0]:
        blk[0], blk[1], blk[2] = mean.sum(dim=0), (mean ** 2).sum(dim=0), (m2 / (nd_local - 1.0)).sum(dim=0)
        head[0], head[1] = float(mean.shape[0]), float(nt.sum())
    blk = _all_reduce(blk, group, reduce_device)
    head = _all_reduce(head, group, reduce_device)
    m = float(head[0])
    nd = float(head[1]) / m
    w = blk[2] / m
    gmean = blk[0] / m
    b_over_n = (blk[1] - m * gmean ** 2) / (m - 1.0) if m > 1 else torch.zeros_like(w)
    return torch.sqrt((w * (nd - 1.0) / nd + b_over_n) / w)


class _DevicePtr:
    """__cuda_array_interface__ shim: view engine-owned HBM as a torch tensor without copying."""

    def __init__(self, ptr, shape, typestr="<f8"):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2}


def trace_tensor(engine):
    """Zero-copy torch view [chains, capacity - trace_begin, dim] of the engine's draws in HBM."""
    ptr = engine.trace_device_ptr()
    if not ptr:
        raise RuntimeError("the engine keeps no trace (reserve(keep_trace=False))")
    engine.synchronize()
    shape = (engine.chains, engine.capacity - engine.trace_begin, engine.dim)
    return torch.as_tensor(_DevicePtr(ptr, shape), device="cuda:%d" % engine.cfg.device)
