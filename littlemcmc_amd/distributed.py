"""Multi-GPU: one process per GPU, contiguous chain blocks, no collective on the sampling path.

Chains are independent: rank r of W owns chains [r*C/W, (r+1)*C/W) together with
that slice of the prefix-stable per-chain seed list (sampling.py:131-136), samples them on its own MI355X
through the ordinary ``sample()`` path, and only the end-of-run diagnostics meet in one all-reduce of
per-dimension sufficient statistics (diagnostics.py) -- RCCL over xGMI when the process group is "nccl",
gloo in the CPU tests. Launch with ``python -m torch.distributed.run --nproc-per-node N ...``.
"""
import os

import numpy as np


def chain_block(total_chains, rank, world):
    """Contiguous block [lo, hi) of rank ``rank``; blocks differ by at most one chain."""
    base, rem = divmod(int(total_chains), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def global_seeds(random_seed, total_chains):
    """Per-chain seeds over the GLOBAL chain index space, identical on every rank (sampling.py:131-136)."""
    from .sampling import _derive_seeds

    return _derive_seeds(random_seed, total_chains)


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def _reduce_device(group, device_index):
    """Where the diagnostics collectives run: on this rank's GPU for RCCL ("nccl" -- it has no host path, every tensor
    handed to it must live on the device), on the host for gloo."""
    import torch
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        if str(dist.get_backend(group)) == "gloo":
            return torch.device("cpu")
        return torch.device("cuda", int(device_index))
    return None


def sample_distributed(logp_dlogp_func, model_ndim=None, draws=1000, tune=1000, chains=None, random_seed=None,
                       start=None, group=None, diagnostics=True, **kwargs):   # diagnostics: True | False | "moments" | "rank_normalized"
    """``sample()`` for a job of ``chains`` chains spread over the ranks of the current process group.

    Returns (trace, stats, diag): this rank's block of the trace/stats (same layouts as ``sample``) and, if
    requested, R-hat/ESS over ALL chains of ALL ranks. With the same ``random_seed`` the union of the blocks
    equals a single-GPU run of ``chains`` chains, chain for chain.
    """
    import inspect

    import torch.distributed as dist

    from .sampling import init_nuts, sample

    rank, world, local_rank = env_rank_world()
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    if chains is None:   # sample()'s own default (sampling.py:124-129) per rank would make the job size depend on the launch
        raise ValueError("sample_distributed needs the TOTAL number of chains (chains=...)")
    chains = int(chains)
    if chains < 1:
        raise ValueError("chains must be >= 1")
    seeds = global_seeds(random_seed, chains)
    lo, hi = chain_block(chains, rank, world)
    # keyword routing as in sample(): its own parameters stay with sample(), everything else configures the step method
    sample_names = set(inspect.signature(sample).parameters) - {"kwargs"}
    sample_kw = {k: v for k, v in kwargs.items() if k in sample_names}
    step_kw = {k: v for k, v in kwargs.items() if k not in sample_names}
    if start is None and sample_kw.get("step") is None:
        # init_nuts must see the GLOBAL first seed so every rank jitters from the same start (sampling.py:574-584)
        start, step = init_nuts(logp_dlogp_func, model_ndim, init=sample_kw.pop("init", "auto"), random_seed=seeds, **step_kw)
        sample_kw["step"] = step
        step_kw = {}
    sample_kw.setdefault("device", local_rank)
    eng = None
    n_keep = int(draws) if sample_kw.get("discard_tuned_samples", True) else int(draws) + int(tune)
    if hi > lo:
        trace, stats, eng = sample(logp_dlogp_func, model_ndim, draws=draws, tune=tune, chains=hi - lo,
                                   random_seed=seeds[lo:hi], start=start, return_engine=True,
                                   keep_moments=(diagnostics == "moments"), **sample_kw, **step_kw)
    else:   # more ranks than chains: this rank owns nothing but still joins the diagnostics reduction
        trace, stats = np.zeros((0, n_keep, int(model_ndim))), {}
    diag = None
    import torch

    dev = torch.device("cuda", int(sample_kw["device"]))
    red = _reduce_device(group, sample_kw["device"])
    if diagnostics == "moments":
        # trace-free cross-chain R-hat: the kernel kept (mean, M2, n) per chain; ranks exchange
        # 3 x d doubles
        from . import diagnostics as dg

        if eng is not None:
            mean, m2, n = (torch.as_tensor(a).to(dev) for a in eng.moments())
        else:
            mean = m2 = torch.zeros((0, int(model_ndim)), dtype=torch.float64, device=dev)
            n = torch.zeros((0,), dtype=torch.int32, device=dev)
        rhat = dg.rhat_from_moments(mean, m2, n, group=group, reduce_device=red)
        diag = {"rhat": rhat.cpu().numpy(), "n_chains": float(chains)}
    elif diagnostics:
        from . import diagnostics as dg

        # a rank without chains joins every collective with an empty block (the draw count is agreed among the ranks
        # that hold chains). Only the rows sample() returned are looked at: after a Ctrl-C the trace buffer's tail is
        # unwritten, and ranks that stopped at different iterations fail the agreed-draw-count check instead of
        # producing statistics of garbage
        x = dg.trace_tensor(eng, n_draws=trace.shape[1]) if eng is not None else torch.zeros((0, 0, int(model_ndim)), dtype=torch.float64, device=dev)
        diag = dg.summarize(x, group=group, reduce_device=red, rank_normalized=(diagnostics == "rank_normalized"))
        diag = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in diag.items()}
    if eng is not None:
        eng.close()
    return trace, stats, diag
