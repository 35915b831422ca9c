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
small all-reduce (RCCL on GPUs) of its (3 + 16) x d block. There is one backend: the HIP kernel. The reduction /
finalisation logic above it is backend-agnostic tensor code, and the CPU tests of that logic (gloo, world_size 2) inject
the test oracle's restatement of the kernel's block (oracle/diagnostics_oracle.py: torch_chain_stats) through the
``stats_fn`` argument; nothing in this package computes the statistics on the host."""
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
    c, _n, d = x.shape
    stride = x.stride(0) // d if c > 1 else _n     # rows of a chain lie d apart; chains `stride` rows apart (_row_major)
    out = torch.empty((3 + LAGS_PER_PASS, d), dtype=torch.float64, device=x.device)
    with torch.cuda.device(x.device):
        stream = torch.cuda.current_stream(x.device).cuda_stream
        rc = lib.lmc_diag_chain_stats(ctypes.c_void_p(x.data_ptr()), c, stride, d, int(t0), int(n), int(lag0),
                                      ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(stream))
    if rc != 0:
        raise RuntimeError("lmc_diag_chain_stats failed (status %d)" % rc)
    return out


def _row_major(x):
    """x[chains, draws, d] as the kernel reads it: a draw is d contiguous doubles, a chain's draws follow each other, and
    chains are a whole number of rows apart -- a contiguous tensor, or the leading draws of one (trace_tensor(e, n))."""
    c, n, d = x.shape
    return x.stride(2) == 1 and x.stride(1) == d and (c <= 1 or (x.stride(0) % d == 0 and x.stride(0) >= n * d))


def _blocks(x):
    """x is one tensor [chains, draws, d], or a list of them: the chain blocks of ONE job that this process holds on
    several GPUs (sample(..., devices=[...]) -> trace_tensor(group)). Reductions land on the first block's device."""
    return list(x) if isinstance(x, (list, tuple)) else [x]


def chain_stats_pass(x, ranges, lag0, stats_fn=None):
    if isinstance(x, (list, tuple)):   # one kernel per device, all enqueued before the first result is moved
        parts = [chain_stats_pass(b, ranges, lag0, stats_fn) for b in x]
        tot = parts[0]
        for p_ in parts[1:]:
            tot = tot + p_.to(tot.device)
        return tot
    return _chain_stats_pass_one(x, ranges, lag0, stats_fn)


def _chain_stats_pass_one(x, ranges, lag0, stats_fn=None):
    """Statistics block [3 + 16, d] of one pass, summed over the sub-series ``ranges`` = [(t0, n), ...] of every
    chain of x (the two halves for split diagnostics). ``stats_fn(x, t0, n, lag0)`` replaces the HIP kernel (tests of
    the reduction logic only)."""
    if x.shape[0] == 0:
        return torch.zeros((3 + LAGS_PER_PASS, x.shape[2]), dtype=torch.float64, device=x.device)
    fn = stats_fn
    if fn is None:
        if not x.is_cuda:
            from ._abi import HipLibraryError

            raise HipLibraryError("diagnostics run on draws in HBM (a ROCm tensor, e.g. diagnostics.trace_tensor(engine)); "
                                  "got a CPU tensor and there is no host implementation")
        if x.dtype != torch.float64 or not _row_major(x):
            x = x.to(torch.float64).contiguous()
        fn = _hip_chain_stats
    tot = None
    for t0, n in ranges:
        blk = fn(x, t0, n, lag0)
        tot = blk if tot is None else tot + blk
    return tot


def _group_active(group=None):
    import torch.distributed as dist

    return dist.is_available() and dist.is_initialized()


def _all_reduce(t, group=None, reduce_device=None, op="sum"):
    """All-reduce over the ranks of ``group`` (a no-op without a process group). The collective runs even in a group
    of one rank: a single-GPU job under RCCL executes the very calls the 8-GPU job does."""
    import torch.distributed as dist

    if not _group_active(group):
        return t
    home = t.device
    r = t.clone() if reduce_device is None else t.to(reduce_device)
    dist.all_reduce(r, op={"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN}[op], group=group)
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


def sufficient_stats(x, split=True, max_lag=None, group=None, reduce_device=None, stats_fn=None):
    """Reduced (over chains, halves and ranks) sufficient statistics of x[chains, draws, d] (this rank's block, or the
    list of this process's per-GPU blocks). Every quantity that steers the pass loop -- the draw count, the lag limit,
    "Geyer's sequence has ended" -- is a REDUCED one, so all ranks issue the same collectives whatever their blocks hold
    (a rank may own no chain at all)."""
    blocks = _blocks(x)
    held = [b for b in blocks if b.shape[0] > 0]
    if len({int(b.shape[1]) for b in held}) > 1:
        raise ValueError("the chain blocks hold different numbers of draws per chain: %s" % [int(b.shape[1]) for b in held])
    c = sum(int(b.shape[0]) for b in blocks)
    n_all, d = (held[0].shape[1], held[0].shape[2]) if held else (blocks[0].shape[1], blocks[0].shape[2])
    dev = blocks[0].device
    if len(blocks) > 1:
        x = held if held else blocks[:1]
    else:
        x = blocks[0]
    # the draw count is agreed first: ranks that own chains must have the same, ranks that own none adopt it
    big = float(2 ** 52)
    mine = torch.tensor([float(n_all), -float(n_all)] if c > 0 else [0.0, -big], dtype=torch.float64, device=dev)
    agreed = _all_reduce(mine, group, reduce_device, op="max")
    n_max, n_min = int(agreed[0]), int(-agreed[1])
    if n_max == 0 and n_min >= int(big):
        raise ValueError("no rank holds any chain")
    if n_max != n_min:
        raise ValueError("ranks hold different numbers of draws per chain (%d .. %d)" % (n_min, n_max))
    n_all = n_max
    if split:
        h = n_all // 2
        ranges, n, halves = [(0, h), (n_all - h, h)], h, 2
    else:
        ranges, n, halves = [(0, n_all)], n_all, 1
    if n < 4:
        raise ValueError("need at least %d draws per chain" % (4 * halves))
    limit = n if max_lag is None else min(int(max_lag), n)
    nch = _all_reduce(torch.tensor([float(c * halves)], dtype=torch.float64, device=dev), group, reduce_device)
    stats = {"n_chains": nch[0], "n_draws": torch.tensor(float(n), dtype=torch.float64, device=dev)}
    acov_blocks = []
    lag0 = 0
    while True:
        blk = _all_reduce(chain_stats_pass(x, ranges, lag0, stats_fn), group, reduce_device)   # the pass's one collective
        if lag0 == 0:
            stats["sum_mean"], stats["sum_mean_sq"], stats["sum_var"] = blk[0], blk[1], blk[2]
        acov_blocks.append(blk[3:])
        lag0 += LAGS_PER_PASS
        stats["sum_acov"] = torch.cat(acov_blocks, dim=0)[:limit]
        if lag0 >= limit:
            break
        ended = torch.tensor([1.0 if _geyer_ended(stats) else 0.0], dtype=torch.float64, device=dev)
        if _group_active(group):   # the verdict is a function of reduced data, but ranks must not differ by a rounding
            ended = _all_reduce(ended, group, reduce_device, op="min")
        if float(ended[0]) > 0.5:
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


def rank_normalize(x, chunk_dims=8, group=None, reduce_device=None):
    """z-scores of the pooled ranks of every dimension (Vehtari et al. 2021, eq. 14): z = Phi^-1((r - 3/8) / (S + 1/4)),
    S = ALL chains x draws of ALL ranks, r = the average rank of the draw among them (ties -- a rejected HMC proposal, a
    NUTS tree that returns its start point -- share the mean of their positions, scipy's rankdata(method="average")).

    Multi-GPU: the ranks are global. Every rank sorts its own pooled draws of a block of dimensions, the sorted blocks
    are all-gathered (padded to the largest block with +inf), and a draw's global rank is the sum over ranks of its
    insertion points: r = #less + (#equal + 1) / 2. Normalising each rank's block on its own would map every block to
    N(0, 1) separately and erase exactly the between-rank differences R-hat is there to detect."""
    import torch.distributed as dist

    if isinstance(x, (list, tuple)):
        return _rank_normalize_blocks(list(x), chunk_dims, group)
    c, n, d = x.shape
    S_local = c * n
    active = _group_active(group)
    S = S_local
    sizes = None
    if active:
        world = dist.get_world_size(group)
        cnt = torch.zeros((world,), dtype=torch.float64, device=x.device)
        cnt[dist.get_rank(group)] = float(S_local)
        sizes = [int(v) for v in _all_reduce(cnt, group, reduce_device)]
        S = sum(sizes)
    out = torch.empty((c, n, d), dtype=torch.float64, device=x.device)
    if active:   # the gathered pools of a chunk are world x chunk x (largest block) doubles on every rank: keep them <= ~1 GiB
        chunk_dims = max(1, min(int(chunk_dims), (1 << 30) // max(1, 8 * len(sizes) * max(sizes))))
    for lo in range(0, d, chunk_dims):
        hi = min(lo + chunk_dims, d)
        blk = x[:, :, lo:hi].reshape(S_local, hi - lo).to(torch.float64).t().contiguous()      # [k, S_local]
        srt = torch.sort(blk, dim=1).values
        if active:
            pad = max(sizes)
            mine = torch.full((hi - lo, pad), float("inf"), dtype=torch.float64, device=x.device)
            mine[:, :S_local] = srt
            send = mine if reduce_device is None else mine.to(reduce_device)
            gathered = [torch.empty_like(send) for _ in sizes]
            dist.all_gather(gathered, send, group=group)
            pools = [g.to(x.device)[:, :sz] for g, sz in zip(gathered, sizes)]
        else:
            pools = [srt]
        less = torch.zeros_like(blk)
        leq = torch.zeros_like(blk)
        for pool in pools:
            if pool.shape[1] == 0:
                continue
            less += torch.searchsorted(pool, blk, right=False).to(torch.float64)
            leq += torch.searchsorted(pool, blk, right=True).to(torch.float64)
        ranks = less + 0.5 * (leq - less + 1.0)
        p = (ranks - 0.375) / (S + 0.25)
        z = math.sqrt(2.0) * torch.erfinv(2.0 * p - 1.0)
        out[:, :, lo:hi] = z.t().reshape(c, n, hi - lo)
    return out


def _rank_normalize_blocks(blocks, chunk_dims, group):
    """rank_normalize for the per-GPU chain blocks of one process: global ranks over ALL blocks. Every device sorts its
    own block of a chunk of dimensions; every device then counts its draws' insertion points in every sorted pool
    (copied device to device), exactly as the multi-process form does with all-gathered pools."""
    if _group_active(group):
        raise ValueError("per-GPU blocks inside one rank of a process group are not supported: use one device per rank")
    S = sum(int(b.shape[0]) * int(b.shape[1]) for b in blocks)
    d = int(blocks[0].shape[2])
    outs = [torch.empty(tuple(b.shape), dtype=torch.float64, device=b.device) for b in blocks]
    largest = max(int(b.shape[0]) * int(b.shape[1]) for b in blocks)
    chunk_dims = max(1, min(int(chunk_dims), (1 << 30) // max(1, 8 * len(blocks) * max(largest, 1))))
    for lo in range(0, d, chunk_dims):
        hi = min(lo + chunk_dims, d)
        flat = [b[:, :, lo:hi].reshape(-1, hi - lo).to(torch.float64).t().contiguous() for b in blocks]   # [k, S_b]
        pools = [torch.sort(f, dim=1).values for f in flat]
        for b, f, out in zip(blocks, flat, outs):
            if f.shape[1] == 0:
                continue
            less = torch.zeros_like(f)
            leq = torch.zeros_like(f)
            for pool in pools:
                if pool.shape[1] == 0:
                    continue
                pl = pool.to(f.device)
                less += torch.searchsorted(pl, f, right=False).to(torch.float64)
                leq += torch.searchsorted(pl, f, right=True).to(torch.float64)
            ranks = less + 0.5 * (leq - less + 1.0)
            p = (ranks - 0.375) / (S + 0.25)
            z = math.sqrt(2.0) * torch.erfinv(2.0 * p - 1.0)
            out[:, :, lo:hi] = z.t().reshape(b.shape[0], b.shape[1], hi - lo)
    return outs


def summarize(x, split=True, max_lag=None, group=None, reduce_device=None, rank_normalized=False, chunk=None,
              stats_fn=None):
    """x[chains, draws, d] (this rank's chain block; or the list of this process's per-GPU blocks, trace_tensor(group))
    -> dict(rhat[d], ess[d], mean[d], var[d]) over ALL chains of ALL ranks.
    ``rank_normalized=True``: the rank-normalised split-R-hat / bulk ESS (diagnostics of the z-scores of the GLOBAL
    ranks). ``stats_fn`` replaces the HIP kernel (tests of the reduction logic only)."""
    if rank_normalized:
        x = rank_normalize(x, group=group, reduce_device=reduce_device)
    out = finalize(sufficient_stats(x, split=split, max_lag=max_lag, group=group, reduce_device=reduce_device,
                                    stats_fn=stats_fn))
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


def trace_tensor(engine, n_draws=None):
    """Zero-copy torch view [chains, capacity - trace_begin, dim] of the engine's draws in HBM (an EngineGroup: the list
    of its engines' views, one per GPU -- summarize() takes it as it is). ``n_draws`` keeps the first n_draws draws of
    every chain only: the rows an interrupted run has actually written."""
    if hasattr(engine, "engines"):
        return [trace_tensor(e, n_draws) for e in engine.engines]
    if n_draws is not None:
        return trace_tensor(engine)[:, :max(int(n_draws), 0)]
    ptr = engine.trace_device_ptr()
    if not ptr:
        raise RuntimeError("the engine keeps no trace (reserve(keep_trace=False))")
    engine.synchronize()
    shape = (engine.chains, engine.capacity - engine.trace_begin, engine.dim)
    return torch.as_tensor(_DevicePtr(ptr, shape), device="cuda:%d" % engine.cfg.device)
