"""Cross-chain convergence diagnostics: split R-hat and multi-chain effective sample size.

The reference has none (ArviZ appears only in a docs recipe, docs/tutorials/framework_cookbook.rst:201-213;
SURVEY.md section 0.9): the definitions below are this build's own -- the classic split-R-hat and the
Geyer initial-monotone-sequence ESS of the Stan reference manual, without rank normalisation -- and are
pinned against a plain-numpy restatement (oracle/diagnostics_oracle.py).

Written as torch tensor code so the same functions run on the draws where they live (HBM, zero-copy from
``Engine.trace_device_ptr()``) and on CPU tensors in the gloo tests. Everything reduces to per-dimension
*sufficient statistics that add over chains*, which is what makes the multi-GPU version one all-reduce:

    n_chains, sum_c mean_c, sum_c mean_c^2, sum_c var_c, sum_c acov_c[t]      (t = 0 .. T-1)
"""
import math

import torch


def split_chains(x):
    """[chains, draws, d] -> [2*chains, draws//2, d] (second half of every chain becomes its own chain)."""
    c, n, d = x.shape
    h = n // 2
    return torch.cat([x[:, :h], x[:, n - h:]], dim=0)


def local_sufficient_stats(x, max_lag=None, chunk=2048):
    """Sufficient statistics of a block of chains x[chains, draws, d] (already split if desired).

    Returns a dict of float64 tensors: n_chains (scalar), n_draws, sum_mean[d], sum_mean_sq[d], sum_var[d],
    sum_acov[T, d] with acov the biased (1/N) within-chain autocovariance.
    """
    c, n, d = x.shape
    T = n if max_lag is None else min(int(max_lag), n)
    nfft = 1 << (2 * n - 1).bit_length()
    dev = x.device
    out = {
        "n_chains": torch.tensor(float(c), dtype=torch.float64, device=dev),
        "n_draws": torch.tensor(float(n), dtype=torch.float64, device=dev),
        "sum_mean": torch.zeros(d, dtype=torch.float64, device=dev),
        "sum_mean_sq": torch.zeros(d, dtype=torch.float64, device=dev),
        "sum_var": torch.zeros(d, dtype=torch.float64, device=dev),
        "sum_acov": torch.zeros(T, d, dtype=torch.float64, device=dev),
    }
    for lo in range(0, c, chunk):
        blk = x[lo:lo + chunk].to(torch.float64)
        mean = blk.mean(dim=1)                                   # [b, d]
        cen = blk - mean[:, None, :]
        f = torch.fft.rfft(cen, n=nfft, dim=1)
        acov = torch.fft.irfft(f.real ** 2 + f.imag ** 2, n=nfft, dim=1)[:, :T] / n   # [b, T, d], biased
        out["sum_mean"] += mean.sum(dim=0)
        out["sum_mean_sq"] += (mean ** 2).sum(dim=0)
        out["sum_var"] += acov[:, 0].sum(dim=0) * (n / (n - 1.0))
        out["sum_acov"] += acov.sum(dim=0)
    return out


def all_reduce_stats(stats, group=None, reduce_device=None):
    """Sum the sufficient statistics over ranks (RCCL on GPUs, gloo on CPU). n_draws must agree.
    ``reduce_device``: where the collective runs (default: where the statistics live)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return stats
    out = dict(stats)
    for k in ("n_chains", "sum_mean", "sum_mean_sq", "sum_var", "sum_acov"):
        home = stats[k].device
        t = stats[k].clone() if reduce_device is None else stats[k].to(reduce_device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        out[k] = t.to(home)
    return out


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
    pairs = pairs * positive
    pairs = torch.cummin(pairs, dim=0).values                             # initial monotone sequence
    tau = -1.0 + 2.0 * pairs.sum(dim=0)
    tau = torch.clamp(tau, min=1.0 / math.log10(max(m * n, 10.0)))
    ess = m * n / tau
    return {"rhat": rhat, "ess": ess, "mean": gmean, "var": var_plus, "n_chains": m, "n_draws": n}


def summarize(x, split=True, max_lag=None, group=None, chunk=2048, reduce_device=None):
    """x[chains, draws, d] (this rank's chain block) -> dict(rhat[d], ess[d], mean[d], var[d]) over ALL ranks."""
    if split:   # the halves are views; sufficient statistics add over chains, so no concatenated copy is made
        n = x.shape[1]
        h = n // 2
        a = local_sufficient_stats(x[:, :h], max_lag=max_lag, chunk=chunk)
        b = local_sufficient_stats(x[:, n - h:], max_lag=max_lag, chunk=chunk)
        stats = {k: (a[k] + b[k] if k != "n_draws" else a[k]) for k in a}
    else:
        stats = local_sufficient_stats(x, max_lag=max_lag, chunk=chunk)
    return finalize(all_reduce_stats(stats, group=group, reduce_device=reduce_device))


def rhat_from_moments(mean, m2, n, group=None, reduce_device=None):
    """(Non-split) R-hat[d] from per-chain running moments -- mean[chains, d], m2[chains, d] (sum of squared
    deviations), n[chains] equal draws per chain -- reduced over ranks with one all-reduce of
    {chains, sum mean, sum mean^2, sum var}. This is the trace-free diagnostic of SURVEY.md section 8e."""
    mean = torch.as_tensor(mean, dtype=torch.float64)
    m2 = torch.as_tensor(m2, dtype=torch.float64)
    nd = float(torch.as_tensor(n).to(torch.float64).mean())
    stats = {"n_chains": torch.tensor(float(mean.shape[0]), dtype=torch.float64, device=mean.device),
             "n_draws": torch.tensor(nd, dtype=torch.float64, device=mean.device),
             "sum_mean": mean.sum(dim=0), "sum_mean_sq": (mean ** 2).sum(dim=0),
             "sum_var": (m2 / (nd - 1.0)).sum(dim=0),
             "sum_acov": torch.zeros(2, mean.shape[1], dtype=torch.float64, device=mean.device)}
    stats = all_reduce_stats(stats, group=group, reduce_device=reduce_device)
    m = float(stats["n_chains"])
    w = stats["sum_var"] / m
    gmean = stats["sum_mean"] / m
    b_over_n = (stats["sum_mean_sq"] - m * gmean ** 2) / (m - 1.0) if m > 1 else torch.zeros_like(w)
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
