#!/usr/bin/env python3
"""Headline benchmark: leapfrog-steps/sec of the many-chain NUTS hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N = 1: this process)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2] / SURVEY.md 8d "C3"): 65 536 chains per GPU x dim 128, AR(1) rho=0.9
correlated Gaussian, NUTS defaults, diagonal mass adaptation, init jitter+adapt_diag; the timed job is
the reference recipe ``sample(tune=T, draws=D)`` cut into K equal launches ("steps") of the one persistent
kernel (tune = first half; the default K = 20 x 100 iterations is exactly SURVEY 8d's tune=1000, draws=1000). W warm-up steps run first on a throw-away copy of the job (same kernel, same
shapes) and are not timed. Chains are independent: with N GPUs every rank owns its own block of
65 536 chains (weak scaling, no data-path collective); ranks only meet in the barrier and in the
max/sum reductions of the timing.

value  = leapfrog steps of ALL chains on ALL GPUs in the timed region (sum of tree_size) / wall seconds
roofline.achieved = algorithmic bytes (60*d per leapfrog step) / kernel time by HIP events
cpu_baseline      = the numpy oracle (a port of the reference, oracle/lmc_oracle.py) on this box's host
                    cores, one chain per core, same recipe, bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK = 8.0e12      # B/s, MI355X spec (guides/MI355X_MICROARCH.md: 8.0 TB/s spec, 6.3 TB/s achievable)
SEED = 20260928


def make_target(lmc, name, dim):
    if name == "ar1":
        return lmc.targets.AR1(dim, 0.9), "AR(1) rho=0.9 correlated Gaussian"
    if name == "std_normal":
        return lmc.targets.StdNormal(dim), "standard normal"
    if name == "funnel":
        return lmc.targets.Funnel(dim), "Neal's funnel"
    if name == "diag":
        return lmc.targets.DiagGaussian.ill_conditioned(dim, 1e4), "ill-conditioned diagonal Gaussian kappa=1e4"
    raise SystemExit("unknown target %s" % name)


def cpu_baseline_worker(args):
    """One oracle chain (a port of the reference's sequential path) -> (leapfrogs, seconds)."""
    name, dim, tune, draws, seed, start, mass = args
    sys.path.insert(0, ROOT)
    try:   # one BLAS thread per chain process: the reference's chain-per-process model, no oversubscription
        from threadpoolctl import threadpool_limits

        threadpool_limits(1)
    except Exception:
        pass
    from oracle import lmc_oracle as orc
    from oracle import targets as OT

    f = {"ar1": lambda: OT.AR1(dim, 0.9), "std_normal": lambda: OT.StdNormal(dim), "funnel": lambda: OT.Funnel(dim),
         "diag": lambda: OT.DiagGaussian.ill_conditioned(dim, 1e4)}[name]()
    if mass == "full_adapt":
        pot = orc.FullAdaptPotential(dim, start, np.eye(dim), 10)
    elif mass == "full":
        idx = np.arange(dim)
        pot = orc.FullPotential(0.9 ** np.abs(idx[:, None] - idx[None, :]) if name == "ar1" else np.eye(dim))
    else:
        pot = orc.DiagAdaptPotential(dim, start, np.ones(dim), 10)
    step = orc.Step(f, dim, kind="nuts", potential=pot)
    t0 = time.perf_counter()
    _tr, st = orc.sample(f, dim, draws=draws, tune=tune, step=step, chains=1, start=start, random_seed=[seed],
                         discard_tuned_samples=False)
    return float(st["tree_size"].sum()), time.perf_counter() - t0


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p_))
        except Exception:
            pass
    return n


def cpu_baseline(name, dim, seeds, start, budget_iters, mass="diag"):
    import multiprocessing as mp

    cores = usable_cores()
    tune = draws = budget_iters // 2
    n_chains = 3 * cores   # ~15 s of CPU work on the GPU box (41 k leapfrogs/s per EPYC core)
    jobs = [(name, dim, tune, draws, int(seeds[i % len(seeds)]), start, mass) for i in range(n_chains)]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        res = pool.map(cpu_baseline_worker, jobs)
    wall = time.perf_counter() - t0
    leap = sum(r[0] for r in res)
    return {
        "value": leap / wall, "unit": "leapfrog-steps/s", "cores": cores, "kind": "port",
        "sample": "%d chains on %d worker processes (1 per usable core) x (tune %d + draws %d), %s d=%d, %s mass, numpy "
                  "oracle (port of the reference's sequential path); %.0f leapfrogs in %.1f s"
                  % (n_chains, cores, tune, draws, name, dim, mass, leap, wall),
        "per_core": leap / wall / cores,
    }


DIAG_LIMITER = ("measured: VALU issue (f64 at 16 lanes/clk), ~220 VALU instr per leapfrog at ~90% issue "
                "utilisation with 3 waves/SIMD; the trajectory lives in registers/LDS, so HBM traffic is "
                "a few % of the algorithmic bytes and frac can exceed 1 (profiles/, DESIGN.md section 6)")
DENSE_LIMITER = ("one float32 d x d matrix sweep per leapfrog (4 d^2 B; the reference does two): per-chain matrices "
                 "(full_adapt) stream from HBM / Infinity Cache, a shared matrix (full) from L2; DESIGN.md section 9")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--chains", type=int, default=65536, help="chains PER GPU")
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--target", default="ar1", choices=["ar1", "std_normal", "funnel", "diag"])
    ap.add_argument("--iters-per-step", type=int, default=100, help="NUTS iterations per chain per launch")
    ap.add_argument("--max-treedepth", type=int, default=10)
    ap.add_argument("--kind", default="nuts", choices=["nuts", "hmc"], help="step method (hmc: path_length 2.0)")
    ap.add_argument("--mass", default="diag", choices=["diag", "full", "full_adapt"],
                    help="mass matrix: diag = QuadPotentialDiagAdapt (headline), full = QuadPotentialFull with the "
                         "target's covariance, full_adapt = QuadPotentialFullAdapt (init='adapt_full'); dense: dim <= 256")
    ap.add_argument("--no-trace", action="store_true", help="do not store draws (statistics only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ess", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=2000, help="iterations per oracle chain in the CPU baseline")
    ap.add_argument("--lds-levels", type=int, default=0)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend (nccl = RCCL; gloo only to exercise the multi-rank path on one GPU)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N > 1)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
    if args.backend == "gloo":          # test mode: several ranks may share one GPU
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    red_dev = "cuda" if args.backend == "nccl" else "cpu"

    import littlemcmc_amd as lmc
    from littlemcmc_amd import _abi

    K, W, ips = args.steps, args.warmup, args.iters_per_step
    n_total = K * ips
    n_tune = n_total // 2
    chains = args.chains
    target, target_desc = make_target(lmc, args.target, args.dim)

    # seeds: sample()'s own derivation over the GLOBAL chain index space (prefix stable), block per rank
    np.random.seed(SEED)
    seeds_all = np.array([np.random.randint(2 ** 30) for _ in range(chains * world)], dtype=np.uint32)
    np.random.seed(int(seeds_all[0]))
    start = 2 * np.random.rand(args.dim) - 1            # init_nuts jitter (sampling.py:574-584)
    seeds = seeds_all[rank * chains:(rank + 1) * chains]

    if args.mass == "diag":
        pot = lmc.QuadPotentialDiagAdapt(args.dim, start, np.ones(args.dim), 10)
        mass_desc = "diag mass adapt"
    elif args.mass == "full_adapt":
        pot = lmc.QuadPotentialFullAdapt(args.dim, start, np.eye(args.dim), 10)      # sampling.py:588-597
        mass_desc = "dense mass adapt (one float32 %dx%d matrix per chain, refreshed + factorised every tuning iteration)" % (
            args.dim, args.dim)
    else:
        idx = np.arange(args.dim)
        cov = 0.9 ** np.abs(idx[:, None] - idx[None, :]) if args.target == "ar1" else np.eye(args.dim)
        pot = lmc.QuadPotentialFull(cov)
        mass_desc = "fixed dense mass (the target's covariance, one float32 matrix shared by all chains)"
    if args.kind == "nuts":
        step = lmc.NUTS(target, args.dim, potential=pot, max_treedepth=args.max_treedepth)
    else:
        step = lmc.HamiltonianMC(target, args.dim, potential=pot, path_length=2.0)
    kw = step._engine_kwargs()
    kw["lds_levels"] = args.lds_levels
    stream = torch.cuda.Stream()        # a real (non-null) HIP stream: the engine launches on it, the events time it

    def new_job(capacity, trace_from, keep_trace):
        eng = lmc.Engine(target, chains=chains, device=local_rank, **kw)
        step.potential._push_initial(eng)
        eng.set_stream(stream.cuda_stream)
        eng.seed(seeds)
        eng.set_position(start)
        eng.reset_tuning()
        eng.reserve(capacity, keep_trace=keep_trace, trace_begin=trace_from)
        return eng

    # ---- warm-up: W launches of a throw-away copy of the job
    if W > 0:
        warm = new_job(W * ips, (W * ips) // 2, keep_trace=False)
        for s in range(W):
            warm.run((W * ips) // 2, s * ips, ips)
        warm.synchronize()
        warm.close()

    # draws stay in HBM; if the requested job is longer than the memory allows, keep the most recent draws only
    trace_begin = n_tune
    if not args.no_trace:
        free_b, _tot = torch.cuda.mem_get_info()
        per_draw = chains * args.dim * 8
        fit = int(0.6 * free_b // per_draw)
        if n_total - n_tune > fit:
            trace_begin = n_total - max(fit, 1)
    eng = new_job(n_total, trace_begin, keep_trace=not args.no_trace)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(K):
        ev[s][0].record(stream)
        eng.run(n_tune, s * ips, ips)
        ev[s][1].record(stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0

    kernel_ms = [a.elapsed_time(b) for a, b in ev]
    ct = eng.counters()
    leap_local = float(ct[:, _abi.CT_LEAPFROGS].sum())
    status = eng.status()
    if status.any():
        raise SystemExit("chains reported failure status bits: %s" % np.unique(status))
    depth_mean = float(eng.stat_i32(_abi.STAT_DEPTH, n_total - min(ips, n_total - n_tune), min(ips, n_total - n_tune)).mean()) \
        if n_total > n_tune else 0.0
    div_after = int(ct[:, _abi.CT_DIVS_AFTER_TUNE].sum())

    # ---- ESS/sec (the second half of BASELINE.json's metric): split-R-hat / Geyer ESS of the post-warm-up
    #      draws, computed where they live (HBM) and reduced across ranks with ONE all-reduce (RCCL) of
    #      per-dimension sufficient statistics -- the only collective of the multi-GPU path.
    ess = None
    if not args.no_trace and not args.no_ess and n_total - n_tune >= 8:
        from littlemcmc_amd import diagnostics as dg

        t_ess = time.perf_counter()
        diag = dg.summarize(dg.trace_tensor(eng), chunk=1024, reduce_device=red_dev)
        torch.cuda.synchronize()
        draw_steps = [s for s in range(K) if s * ips >= trace_begin]
        draw_s = sum(kernel_ms[s] for s in draw_steps) / 1e3
        e = diag["ess"]
        ess = {"min": float(e.min()), "median": float(e.median()), "rhat_max": float(diag["rhat"].max()),
               "draw_seconds_this_rank": draw_s, "chains_total": diag["n_chains"] / 2, "draws": n_total - trace_begin,
               "diagnostics_seconds": time.perf_counter() - t_ess}
    eng.close()

    wall_t = torch.tensor([wall], dtype=torch.float64, device=red_dev)
    leap_t = torch.tensor([leap_local], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(wall_t, op=dist.ReduceOp.MAX)
        dist.all_reduce(leap_t, op=dist.ReduceOp.SUM)
    wall_max, leap_all = float(wall_t.item()), float(leap_t.item())

    if rank == 0:
        value = leap_all / wall_max
        kern_s = sum(kernel_ms) / 1e3
        # algorithmic bytes of one reference leapfrog: the State vectors, plus for a dense mass matrix the
        # two float32 d x d sweeps of integration.py:111,118 (the device kernel needs one)
        bytes_per_leap = 60 * args.dim + (0 if args.mass == "diag" else 8 * args.dim * args.dim)
        achieved = leap_local * bytes_per_leap / kern_s          # this GPU's kernel, algorithmic bytes / kernel time
        traffic, traffic_src = None, None
        try:   # HBM bytes per launch from the committed PMC profile of this same workload (not measurable in-process)
            tr_key = ("%s:%d" % (args.target, args.dim)) + ("" if args.mass == "diag" else ":" + args.mass)
            tr = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(tr_key)
            if tr:
                traffic = tr["hbm_bytes_per_leapfrog"] * leap_local / K
                traffic_src = tr["source"]
        except Exception:
            pass
        valu_issue = None
        try:   # the limiter that actually binds (profiles/: VALU issue), as measured by the committed PMC run
            if args.mass != "diag":
                raise KeyError("VALU-issue figures are for the diagonal kernel")
            ps = json.load(open(os.path.join(ROOT, "profiles", "r01_default_pmc_summary.json")))
            occ = 3 if args.dim > 64 else 4   # waves per SIMD of the instantiation (168 / 128 VGPRs)
            valu_issue = {"valu_inst_per_leapfrog": ps["per_leapfrog"]["SQ_INSTS_VALU"],
                          "simd_issue_utilisation": min(1.0, occ * ps["wave_time_split"]["valu_active"]),
                          "source": "profiles/r01_default_pmc_summary.json (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES x waves per SIMD)"}
        except Exception:
            pass
        method = ("NUTS max_treedepth=%d" % args.max_treedepth) if args.kind == "nuts" else "HMC path_length=2"
        out = {
            "metric": "leapfrog-steps/sec (all chains)", "value": value, "unit": "leapfrog-steps/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": wall_max * 1e3 / K,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "C3: %d chains/GPU x dim %d %s, %s, %s, tune %d + draws %d in %d launches of "
                            "%d iterations" % (chains, args.dim, target_desc, method, mass_desc, n_tune, n_total - n_tune, K, ips),
                "chains_per_gpu": chains, "dim": args.dim, "target": args.target, "tune": n_tune,
                "draws": n_total - n_tune, "rng": "MT19937 (numpy legacy stream, same-seed parity mode)",
                "trace_in_hbm": not args.no_trace, "parallelism": "chain-block x%d" % world,
            },
            "leapfrogs": leap_all, "wall_s": wall_max, "mean_depth_draws": depth_mean,
            "ess_per_sec": None if ess is None else {
                "min": ess["min"] / ess["draw_seconds_this_rank"], "median": ess["median"] / ess["draw_seconds_this_rank"],
                "ess_min": ess["min"], "ess_median": ess["median"], "rhat_max": ess["rhat_max"],
                "definition": "multi-chain split-R-hat / Geyer ESS over all %d chains x %d post-warm-up draws, "
                              "divided by the post-warm-up kernel time" % (int(ess["chains_total"]), ess["draws"]),
                "diagnostics_seconds": ess["diagnostics_seconds"]},
            "divergences_after_tune": div_after,
            "roofline": {
                "bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": achieved / HBM_PEAK, "traffic": traffic, "traffic_unit": "B per launch", "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": leap_local * bytes_per_leap / K,
                "kernel": ("lmc::run_kernel<NS=%d>" if args.mass == "diag" else "lmc::run_dense_kernel<NS=%d>") % max(1, (args.dim + 63) // 64),
                "kernel_ms_avg": sum(kernel_ms) / K, "algorithmic_bytes_per_leapfrog": bytes_per_leap,
                "read_only_frac": leap_local * 28 * args.dim / kern_s / HBM_PEAK,
                "valu_issue": valu_issue,
                "limiter": DIAG_LIMITER if args.mass == "diag" else DENSE_LIMITER,
            },
        }
        if not args.no_cpu_baseline and world == 1:   # reported baseline: rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(args.target, args.dim, seeds_all[:64], start, args.cpu_iters, args.mass)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
