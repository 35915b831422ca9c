#!/usr/bin/env python3
"""Headline benchmark: leapfrog-steps/sec of the many-chain NUTS hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N = 1: this process)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2] / SURVEY.md 8d "C3"): 65 536 chains x dim 128, AR(1) rho=0.9 correlated
Gaussian, NUTS defaults, diagonal mass adaptation, init jitter+adapt_diag; the timed job is the reference recipe
``sample(tune=T, draws=D)`` cut into K equal launches ("steps") of the one persistent kernel (tune = first half;
the default K = 20 x 100 iterations is SURVEY 8d's tune=1000, draws=1000). W warm-up steps run first on a
throw-away copy of the job (same kernel, same shapes) and are not timed.

Multi-GPU (``--scaling strong``, the default, is C3 as BASELINE states it: "65 536 chains ... 1 -> 8 MI355X"): the
65 536 chains are dealt to the ranks in contiguous blocks (littlemcmc_amd.distributed.chain_block) with the seeds of
the GLOBAL chain index space, so the union of the blocks is the N = 1 job chain for chain; ``--scaling weak`` gives
every rank its own 65 536 chains. Chains are independent: no data-path collective, ranks meet in the barrier, in the
max/sum reductions of the timing and in ONE all-reduce of diagnostics statistics.

value            = leapfrog steps of ALL chains on ALL GPUs in the timed region (sum of tree_size) / wall seconds
roofline         = the bound that binds the kernel: FP64 vector issue. achieved = leapfrogs/s x 26*d flop (SURVEY 8d's
                   flop count of one leapfrog incl. amortised U-turn dots) against 78.6 TFLOP/s; the HBM contract
                   figures (60*d and 28*d bytes per leapfrog) are side fields -- the State never leaves registers/LDS,
                   so they are not a bound. Counter-derived fields (HBM traffic, VALU instructions per leapfrog) are
                   quoted from profiles/pmc_counters.json ONLY if that profile was taken on this very build
                   (source hash match); otherwise they are null.
secondary        = north_star's named shape (standard normal, d = 128, same chains / recipe) in both RNG modes and, on the
                   default command line, the other BASELINE.json configurations -- C2 (4 096 x 64), C4 (8 192 x 1000),
                   C5 (16 384 x 256 funnel, treedepth 12; as K launches and as ONE launch) -- each timed the same way with
                   its own roofline and tail block.
stdout           = ONE compact JSON line (compact_line(): <= 8192 bytes -- headline fields, config, roofline, cpu_baseline, ESS
                   numbers, a short block per secondary workload); the verbose object (every note, full secondary blocks)
                   is written to bench_detail.json (--detail-out), not to stderr.
sample_e2e       = the literal drop-in call lmc.sample(...) -> numpy arrays on C3's chains (tune 300 + draws 200), wall time next to
                   the kernel-only time of the same job (N = 1, default command line).
cpu_baseline     = the numpy oracle (a port of the reference, oracle/lmc_oracle.py) on this box's host cores, one chain
                   per core, same recipe, bounded sample.
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

HBM_PEAK = 8.0e12        # B/s, MI355X spec (guides/MI355X_MICROARCH.md: 8.0 TB/s spec, 6.3 TB/s achievable)
FP64_VALU_PEAK = 78.6e12  # flop/s: 256 CUs x 4 SIMDs x 16 lanes x 2 (FMA) x 2.4 GHz
FLOP_PER_LEAPFROG_PER_DIM = 26   # SURVEY.md 8d "Bound": ~26*d FP64 flop per leapfrog incl. amortised U-turn dots
SEED = 20260928


LINE_LIMIT = 8192   # bytes: the driver's record keeps a bounded tail of stdout; round 5's 24 KB line left BENCH_r05.parsed null


def _short(text, n):
    text = "" if text is None else str(text)
    return text if len(text) <= n else text[:n - 3] + "..."


def _pick(d, keys):
    return {k: d.get(k) for k in keys if k in d}


def compact_line(out, detail_path=None, limit=LINE_LIMIT):
    """The ONE stdout line of a run: the contract's headline fields, `config`, `roofline`, `cpu_baseline`, the ESS numbers
    and a few figures per secondary workload -- numbers and short labels only, every explanatory string stays in the
    verbose object (`bench_detail.json`). Guaranteed <= `limit` bytes: if an unusual run (many ranks, long error texts)
    still overshoots, per-rank rows and then secondary extras are dropped, never the headline / roofline / cpu_baseline."""
    ROOF = ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_unit", "kernel", "kernel_ms_avg",
            "leapfrogs_per_launch", "dispatches_per_step", "dispatch_ms_avg", "flop_per_leapfrog", "bytes_per_leapfrog",
            "valu_inst_per_leapfrog", "simd_valu_busy")
    TAIL = ("busiest_chain_leapfrogs", "mean_chain_leapfrogs", "launch_max_over_mean", "critical_path_leapfrogs",
            "lone_wave_us_per_leapfrog", "implied_wall_lower_bound_s", "tail_bound_frac", "kernel_s", "resident_chains", "waves_per_chain",
            "lds_bytes_per_workgroup", "mean_wave_slot_occupancy")

    def roof(r):
        c = _pick(r, ROOF)
        for k in ("hbm_contract_60d", "hbm_contract_28d_read_only"):
            if isinstance(r.get(k), dict):
                c[k + "_frac_of_8TBps"] = r[k].get("frac_of_8TBps")
        if r.get("pmc_source"):
            c["pmc_source"] = _short(str(r["pmc_source"]).split(" ")[0], 60)
        return c

    line = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                       "vs_baseline", "dtype", "data"))
    cfg = dict(out.get("config", {}))
    cfg["workload"] = _short(cfg.get("workload_short") or cfg.get("workload"), 140)
    cfg.pop("workload_short", None)
    cfg["rng"] = _short(cfg.get("rng"), 40)
    line["config"] = cfg
    line.update(_pick(out, ("leapfrogs", "wall_s", "mean_depth_draws", "divergences_after_tune")))
    if out.get("ess_per_sec"):
        line["ess_per_sec"] = {k: v for k, v in out["ess_per_sec"].items() if not isinstance(v, str)}
        line["ess_per_sec"]["parity"] = "unpinned (the reference has no ESS)"
    else:
        line["ess_per_sec"] = None
    line["roofline"] = roof(out.get("roofline", {}))
    if out.get("cpu_baseline"):
        cb = dict(out["cpu_baseline"])
        cb["sample"] = _short(cb.get("sample_short") or cb.get("sample"), 160)
        cb.pop("sample_short", None)
        line["cpu_baseline"] = cb
    if out.get("sample_e2e"):
        line["sample_e2e"] = {k: v for k, v in out["sample_e2e"].items() if k not in ("note", "call")}
    line["tail"] = _pick(out.get("tail", {}), TAIL)
    line["per_rank"] = [_pick(r, ("rank", "chains", "leapfrogs", "kernel_s", "wall_s")) for r in out.get("per_rank", [])]
    line.update(_pick(out, ("rccl_ranks", "backend", "source_hash", "source_tree_hash")))
    line["rccl_error"] = None if out.get("rccl_error") is None else _short(out["rccl_error"], 160)
    line["launcher"] = _short(str(out.get("launcher", "")).split(":")[0], 16)
    line["launcher_fallback"] = None if out.get("launcher_fallback") is None else _short(out["launcher_fallback"], 160)
    sec = []
    for s_ in out.get("secondary", []):
        r = s_.get("roofline", {})
        t = s_.get("tail", {})
        sec.append({"workload": _short(s_.get("workload_short") or s_.get("workload"), 130), "value": s_.get("value"),
                    "ms_per_step": s_.get("ms_per_step"), "steps": s_.get("steps"),
                    "roofline": _pick(r, ("kernel", "frac", "bound")),
                    "tail": _pick(t, ("mean_wave_slot_occupancy", "implied_wall_lower_bound_s", "tail_bound_frac", "lone_wave_us_per_leapfrog"))})
    if sec:
        line["secondary"] = sec
    if detail_path:
        line["detail"] = detail_path

    def trim(x):   # six significant digits for non-integral floats (integral ones -- leapfrog counts -- stay exact)
        if isinstance(x, float) and x == x and abs(x) != float("inf") and x != int(x):
            return float("%.6g" % x)
        if isinstance(x, dict):
            return {k: trim(v) for k, v in x.items()}
        if isinstance(x, list):
            return [trim(v) for v in x]
        return x

    for k_ in list(line):
        if k_ not in ("value", "ms_per_step"):
            line[k_] = trim(line[k_])
    sec = line.get("secondary", [])

    def size():
        return len(json.dumps(line)) + 1

    if size() > limit:
        line["per_rank"] = "dropped (line limit): see detail"
    if size() > limit and sec:
        for s_ in sec:
            s_.pop("tail", None)
    if size() > limit:
        line["tail"] = _pick(line["tail"], ("mean_wave_slot_occupancy", "implied_wall_lower_bound_s"))
        line.pop("secondary", None)
    assert size() <= limit, size()
    return line


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


def kernel_name(dim, mass):
    """The sampling kernel lmc_engine_run dispatches for a fused shape (lmc_engine.hip:773-776): NS elements per lane,
    W wavefronts per chain -- one wavefront up to d = 256, teams of 2 / 4 above."""
    if mass != "diag":
        return "lmc::run_dense_coop_kernel<NS=%d>" % max(1, (dim + 63) // 64) if mass == "full" else \
               "lmc::run_dense_kernel<NS=%d>" % max(1, (dim + 63) // 64)
    if dim <= 256:
        ns = 1 if dim <= 64 else (2 if dim <= 128 else 4)
        return "lmc::run_kernel<NS=%d,W=1>" % ns
    if dim <= 1024:
        return "lmc::run_kernel<NS=4,W=%d>" % (2 if dim <= 512 else 4)
    return "lmc::run_wide_kernel"


def config_label(target, dim, chains_total, max_treedepth, kind, mass):
    """BASELINE.json's name for the workload when it is one of its configurations, else a neutral label."""
    if kind == "nuts" and mass == "diag":
        if (target, dim, chains_total, max_treedepth) == ("ar1", 128, 65536, 10):
            return "C3"
        if (target, dim, chains_total, max_treedepth) == ("std_normal", 64, 4096, 10):
            return "C2"
        if (target, dim, chains_total, max_treedepth) == ("diag", 1000, 8192, 10):
            return "C4"
        if (target, dim, chains_total, max_treedepth) == ("funnel", 256, 16384, 12):
            return "C5"
        if (target, dim, chains_total, max_treedepth) == ("std_normal", 128, 65536, 10):
            return "north_star shape"
    return "custom"


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
        "sample_short": "%d oracle chains (numpy port of the reference) on %d procs x (tune %d + draws %d), %s d=%d %s mass; "
                        "%.0f leapfrogs in %.1f s" % (n_chains, cores, tune, draws, name, dim, mass, leap, wall),
        "per_core": leap / wall / cores,
    }


class _DevView:
    """__cuda_array_interface__ shim: engine-owned HBM as a torch tensor, no copy."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def pmc_profile(key, source_hash):
    """Counter-derived figures of this workload, if profiles/pmc_counters.json holds a profile of THIS build."""
    try:
        entry = json.load(open(os.path.join(ROOT, "profiles", "pmc_counters.json"))).get(key)
    except Exception:
        return None
    if not entry or entry.get("source_hash") != source_hash:
        return None
    return entry


def self_launch(n):
    """``python bench.py --gpus N`` without a launcher: re-run this very command line as N ranks (one per GPU) under
    ``torch.distributed.run`` on 127.0.0.1 with a free port; rank 0 prints the one JSON line, which passes through."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--chains", type=int, default=65536, help="chains of the job (strong scaling) / per GPU (weak scaling)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="strong: --chains is the whole job, dealt to the ranks in contiguous blocks (C3 as BASELINE "
                         "states it); weak: every rank owns --chains chains")
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
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end line: the literal lmc.sample() call returning numpy "
                                                          "arrays (sampling.py:207-222), next to the kernel-only time of the same job")
    ap.add_argument("--no-secondary", action="store_true", help="skip every secondary workload")
    ap.add_argument("--no-baseline-configs", action="store_true",
                    help="skip the C2 / C4 / C5 lines the default command line appends to `secondary` (keeps the north_star shape)")
    ap.add_argument("--cpu-iters", type=int, default=2000, help="iterations per oracle chain in the CPU baseline")
    ap.add_argument("--lds-levels", type=int, default=0)
    ap.add_argument("--rng", default="numpy", choices=["numpy", "philox"],
                    help="momentum stream of the PRIMARY workload: numpy = the reference's (same-seed parity; the headline), "
                         "philox = counter-based throughput mode (include/lmc_hip.h: LMC_RNG_PHILOX)")
    ap.add_argument("--no-philox-line", action="store_true", help="skip the separately labelled counter-based-RNG line of the north_star shape")
    ap.add_argument("--no-tail", action="store_true", help="skip the lone-chain latency measurement of the tail block (profiling "
                                                           "runs: keeps every run_kernel dispatch the same size)")
    ap.add_argument("--detail-out", default=None, help="where the verbose result object goes (default: bench_detail.json next to "
                                                       "bench.py); stdout carries one compact line of at most %d bytes" % LINE_LIMIT)
    ap.add_argument("--no-rccl-check", action="store_true", help="N = 1: do not bring up the one-rank RCCL group")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend (nccl = RCCL; gloo only to exercise the multi-rank path on one GPU)")
    ap.add_argument("--launcher", default="ranks", choices=["ranks", "inproc"],
                    help="how N GPUs are driven. ranks (default): one process per GPU under torch.distributed.run, RCCL for "
                         "the diagnostics all-reduce and the timing reductions. inproc: ONE process, one engine per GPU on "
                         "its chain block, launches enqueued on all devices before anything is waited for (what "
                         "lmc.sample(..., devices=[...]) does); no process group, so a RCCL / launcher bring-up failure "
                         "cannot cost the scaling curve. The line records which one ran (`launcher`). If there are fewer "
                         "GPUs than N (the one-GPU test box) the engines share devices round-robin.")
    args = ap.parse_args()

    inproc = args.launcher == "inproc"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not inproc:
        sys.exit(self_launch(args.gpus))
    # stdout carries ONE line, the JSON record: whatever the libraries print (RCCL's version banner, c10d warnings) goes
    # to stderr -- at the file-descriptor level, C stdio included -- and the record is written to the real stdout last
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if inproc:
        if world != 1:
            raise SystemExit("--launcher inproc is ONE process driving all GPUs; do not start it under torch.distributed.run")
        args.no_rccl_check = True
    elif world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (start bench.py without WORLD_SIZE to let it launch its own "
                         "ranks, or under torch.distributed.run --nproc-per-node %d)" % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
    if args.backend == "gloo" or os.environ.get("LMC_BENCH_FAIL_RCCL"):          # test modes: several ranks may share one GPU
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    rccl_error = None
    launcher_fallback = None
    if world > 1:
        if args.backend == "nccl":
            # The multi-GPU curve must not hinge on RCCL coming up: the sampling path needs no collective at all. If the
            # process group cannot be brought up (or its first all-reduce fails), rank 0 runs the SAME job through the
            # in-process launcher -- one engine per GPU from this one process -- the other ranks leave, and the line says
            # what happened (`launcher`, `rccl_error`). Chains, seeds, blocks and the timed region are the same either way.
            try:
                import datetime

                if os.environ.get("LMC_BENCH_FAIL_RCCL"):   # test hook: exercise the fallback without breaking RCCL
                    raise RuntimeError("LMC_BENCH_FAIL_RCCL is set")
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank),
                                        timeout=datetime.timedelta(seconds=180))
                probe = torch.ones(1, device="cuda")
                dist.all_reduce(probe)
                torch.cuda.synchronize()
                if int(probe.item()) != world:
                    raise RuntimeError("all-reduce over %d ranks returned %d" % (world, int(probe.item())))
            except Exception as err:
                rccl_error = "%s: %s" % (type(err).__name__, str(err)[:300])
                try:
                    if dist.is_initialized():
                        dist.destroy_process_group()
                except Exception:
                    pass
                if rank != 0:
                    print("rank %d: RCCL bring-up failed (%s); rank 0 runs the job in-process" % (rank, rccl_error), file=sys.stderr)
                    os._exit(0)
                launcher_fallback = "ranks -> inproc: the RCCL process group did not come up (%s)" % rccl_error
                inproc = True
                world, rank, local_rank = 1, 0, 0
                torch.cuda.set_device(0)
                args.no_rccl_check = True
                time.sleep(5.0)   # the other ranks release their GPUs
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    elif args.backend == "nccl" and not args.no_rccl_check:
        # N = 1: the job needs no collective, but the line still says whether RCCL works on this box (a one-rank group;
        # a failure is recorded, it cannot cost the measurement)
        try:
            if "MASTER_PORT" not in os.environ:
                import socket

                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
        except Exception as err:
            rccl_error = "%s: %s" % (type(err).__name__, str(err)[:200])
    red_dev = "cuda" if args.backend == "nccl" else "cpu"
    group_up = dist.is_available() and dist.is_initialized()
    rccl_ranks = None
    if group_up and args.backend == "nccl":
        try:   # an all-reduce of ones on the GPUs: how many ranks RCCL actually connected
            one = torch.ones(1, device="cuda")
            dist.all_reduce(one)
            torch.cuda.synchronize()
            rccl_ranks = int(one.item())
        except Exception as err:
            if world > 1:
                raise
            rccl_error = "%s: %s" % (type(err).__name__, str(err)[:200])

    import littlemcmc_amd as lmc
    from littlemcmc_amd import _abi, _build
    from littlemcmc_amd.distributed import chain_block

    K, W, ips = args.steps, args.warmup, args.iters_per_step
    n_total = K * ips
    n_tune = n_total // 2
    n_units = args.gpus if inproc else world          # GPUs (chain blocks) of the job
    chains_total = args.chains if args.scaling == "strong" else args.chains * n_units

    n_dev = torch.cuda.device_count()

    def job_parts(job_chains):
        """The chain blocks THIS process drives for a job of `job_chains` chains (per GPU under weak scaling): one (its
        rank's) under the process-per-GPU launcher, all of them in-process. -> (parts, chains_total)"""
        total = job_chains if args.scaling == "strong" else job_chains * n_units

        def unit_block(u):
            return chain_block(total, u, n_units) if args.scaling == "strong" else (u * job_chains, (u + 1) * job_chains)

        if inproc:
            parts_ = [{"unit": u, "dev": u % n_dev, "lo": unit_block(u)[0], "hi": unit_block(u)[1]} for u in range(n_units)]
        else:
            parts_ = [{"unit": rank, "dev": local_rank, "lo": unit_block(rank)[0], "hi": unit_block(rank)[1]}]
        for p_ in parts_:
            if p_["hi"] - p_["lo"] < 1:
                raise SystemExit("GPU %d owns no chain (%d chains over %d GPUs)" % (p_["unit"], total, n_units))
        return parts_, total

    parts, _ct = job_parts(args.chains)
    chains = parts[0]["hi"] - parts[0]["lo"]          # the first block: the GPU the roofline / tail figures are taken on
    all_devs = sorted({u % n_dev for u in range(n_units)}) if inproc else [local_rank]

    def sync_all():
        for dv in all_devs:
            torch.cuda.synchronize(dv)

    # seeds: sample()'s own derivation over the GLOBAL chain index space (prefix stable: a job of fewer chains uses the
    # first of them), block per GPU
    most_chains = max(chains_total, 65536 if args.scaling == "strong" else 65536 * n_units)
    np.random.seed(SEED)
    seeds_all = np.array([np.random.randint(2 ** 30) for _ in range(most_chains)], dtype=np.uint32)

    def all_reduce(vals, op):
        t = torch.tensor(vals, dtype=torch.float64, device=red_dev)
        if world > 1:
            dist.all_reduce(t, op=op)
        return [float(v) for v in t]

    def all_gather(vals):
        """[world][len(vals)] -- per-rank figures for the line (imbalance must be visible in the record itself)"""
        t = torch.tensor(vals, dtype=torch.float64, device=red_dev)
        if world == 1:
            return [[float(v) for v in t]]
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return [[float(v) for v in o] for o in out]

    RNG_LABEL = {"numpy": "MT19937 (numpy legacy stream, same-seed parity mode)",
                 "philox": "Philox4x32-10 momentum stream (throughput mode: NOT the reference's draws), tree uniforms MT19937"}

    def run_job(target_name, dim, with_ess, rng="numpy", job_chains=None, max_treedepth=None, K=K, ips=ips, tag=""):
        """The timed job on this process's chain block(s) -> dict of measurements (wall / leapfrogs reduced over ranks).
        Defaults = the primary workload of the command line; the BASELINE configurations that ride along as `secondary`
        lines name their own chains / tree depth / launch split."""
        parts, chains_total = job_parts(args.chains if job_chains is None else job_chains)
        chains = parts[0]["hi"] - parts[0]["lo"]
        devs_here = sorted({p_["dev"] for p_ in parts})
        md = args.max_treedepth if max_treedepth is None else max_treedepth
        n_total = K * ips
        n_tune = n_total // 2
        w_ips = min(ips, 100)                              # warm-up launches stay short whatever the timed split is
        target, target_desc = make_target(lmc, target_name, dim)
        np.random.seed(int(seeds_all[0]))
        start = 2 * np.random.rand(dim) - 1            # init_nuts jitter (sampling.py:574-584)
        if args.mass == "diag":
            pot = lmc.QuadPotentialDiagAdapt(dim, start, np.ones(dim), 10)
            mass_desc = "diag mass adapt"
        elif args.mass == "full_adapt":
            pot = lmc.QuadPotentialFullAdapt(dim, start, np.eye(dim), 10)      # sampling.py:588-597
            mass_desc = ("dense mass adapt (one float32 %dx%d matrix per chain, refreshed + factorised every tuning "
                         "iteration)" % (dim, dim))
        else:
            idx = np.arange(dim)
            cov = 0.9 ** np.abs(idx[:, None] - idx[None, :]) if target_name == "ar1" else np.eye(dim)
            pot = lmc.QuadPotentialFull(cov)
            mass_desc = "fixed dense mass (the target's covariance, one float32 matrix shared by all chains)"
        if args.kind == "nuts":
            step = lmc.NUTS(target, dim, potential=pot, max_treedepth=md)
        else:
            step = lmc.HamiltonianMC(target, dim, potential=pot, path_length=2.0)
        kw = step._engine_kwargs()
        kw["lds_levels"] = args.lds_levels
        kw["rng"] = rng

        def new_job(part, capacity, trace_from, keep_trace):
            eng = lmc.Engine(target, chains=part["hi"] - part["lo"], device=part["dev"], **kw)
            step.potential._push_initial(eng)
            eng.seed(seeds_all[part["lo"]:part["hi"]])
            eng.set_position(start)
            eng.reset_tuning()
            eng.reserve(capacity, keep_trace=keep_trace, trace_begin=trace_from)
            return eng

        def tail_block(eng, ct, kernel_s, nst, target, step, kw, start):
            """Why the job cannot be faster than its busiest chain (ragged trees): per launch every sub-block waits for its
            slowest chain, and a chain is one wavefront (or team) whose leapfrogs run back to back."""
            dev0 = int(eng.cfg.device)
            resident, wpc, hz = eng.occupancy()
            leap_chain = ct[:, _abi.CT_LEAPFROGS]
            # tree_size of every iteration, where it lives: [chains][capacity] int32 in HBM -> leapfrogs per chain per launch
            ts = torch.as_tensor(eng.tree_size_view(), device="cuda:%d" % dev0)
            per_launch = ts.reshape(chains, K, ips).sum(dim=2, dtype=torch.int64)                 # [chains, K]
            bounds = [chains * b // nst for b in range(nst + 1)]
            crit = max(int(per_launch[bounds[b]:bounds[b + 1]].max(dim=0).values.sum()) for b in range(nst))
            ratio = float((per_launch.max(dim=0).values.double() / per_launch.double().mean(dim=0).clamp(min=1.0)).mean())
            lds_bytes = eng.run_lds_bytes()
            if args.no_tail:
                return {"busiest_chain_leapfrogs": int(leap_chain.max()), "mean_chain_leapfrogs": float(leap_chain.mean()),
                        "launch_max_over_mean": ratio, "critical_path_leapfrogs": crit, "kernel_s": kernel_s,
                        "resident_chains": resident, "waves_per_chain": wpc, "lds_bytes_per_workgroup": lds_bytes,
                        "mean_wave_slot_occupancy": (float(ct[:, _abi.CT_WAVE_TICKS].sum()) / hz / (resident * wpc * kernel_s)) if resident else None}
            # a lone chain on an otherwise idle GPU: the issue latency of one wavefront (team) of this kernel
            lone = lmc.Engine(target, chains=1, device=dev0, **kw)
            step.potential._push_initial(lone)
            lone.seed(seeds_all[:1])
            lone.set_position(start)
            lone.reset_tuning()
            n_l = 2 * w_ips
            lone.reserve(n_l, keep_trace=False)
            lone.run(n_l // 2, 0, n_l // 2)
            lone.synchronize()
            l0 = int(lone.counters()[0, _abi.CT_LEAPFROGS])
            t_l = time.perf_counter()
            lone.run(n_l // 2, n_l // 2, n_l - n_l // 2)
            lone.synchronize()
            t_l = time.perf_counter() - t_l
            l1 = int(lone.counters()[0, _abi.CT_LEAPFROGS]) - l0
            lone.close()
            lone_us = 1e6 * t_l / max(l1, 1)
            slots = resident * wpc
            return {
                "busiest_chain_leapfrogs": int(leap_chain.max()), "mean_chain_leapfrogs": float(leap_chain.mean()),
                "launch_max_over_mean": ratio,
                "critical_path_leapfrogs": crit,
                "lone_wave_us_per_leapfrog": lone_us,
                "implied_wall_lower_bound_s": crit * lone_us * 1e-6,
                # share of this job's kernel time that is ONE chain's sequential leapfrogs: near 1 (C5: 0.97) the job cannot
                # get faster with more GPUs -- every chain block still contains a chain like its busiest one, or waits for it
                "tail_bound_frac": crit * lone_us * 1e-6 / kernel_s if kernel_s > 0 else None,
                "kernel_s": kernel_s,
                "resident_chains": resident, "waves_per_chain": wpc, "lds_bytes_per_workgroup": lds_bytes,
                "mean_wave_slot_occupancy": (float(ct[:, _abi.CT_WAVE_TICKS].sum()) / hz / (slots * kernel_s)) if slots else None,
                "note": "per launch each sub-block of chains ends with its busiest chain; critical_path_leapfrogs = max over "
                        "sub-blocks of the sum over launches of that chain's leapfrogs; x the leapfrog latency of a lone "
                        "wavefront (measured on a 1-chain engine of the same kernel, %d post-tuning iterations) = a lower "
                        "bound on the wall time WHATEVER THE NUMBER OF GPUS (tail_bound_frac = that bound / kernel time: near 1 the "
                        "multi-GPU line of this workload is flat by construction); occupancy = resident wave time (device "
                        "counter) / (wave slots x kernel time); first chain block of the job" % (n_l - n_l // 2),
            }

        if W > 0:   # warm-up: W launches of a throw-away copy of the job
            warm = [new_job(p_, W * w_ips, (W * w_ips) // 2, keep_trace=False) for p_ in parts]
            for s_ in range(W):
                for w_ in warm:
                    w_.run((W * w_ips) // 2, s_ * w_ips, w_ips)
            for w_ in warm:
                w_.synchronize()
                w_.close()

        # draws stay in HBM; if the requested job is longer than the memory allows, keep the most recent draws only
        trace_begin = n_tune
        keep_trace = with_ess and not args.no_trace
        if keep_trace:
            for dv in devs_here:
                free_b, _tot = torch.cuda.mem_get_info(dv)
                per_draw = sum(p_["hi"] - p_["lo"] for p_ in parts if p_["dev"] == dv) * dim * 8
                fit = int(0.6 * free_b // per_draw)
                if n_total - trace_begin > fit:
                    trace_begin = n_total - max(fit, 1)
        engs = [new_job(p_, n_total, trace_begin, keep_trace=keep_trace) for p_ in parts]
        # HIP events on the streams the kernel is launched on: an engine launches its chains as sub-blocks (contiguous quarters on
        # four internal streams, lmc_engine_run_streams), so a step is `len(run_streams)` concurrent dispatches per GPU
        run_streams = [[torch.cuda.ExternalStream(h, device=torch.device("cuda", p_["dev"])) for h in e_.run_streams()]
                       for e_, p_ in zip(engs, parts)]
        nst = len(run_streams[0])

        def new_event(dv):
            with torch.cuda.device(dv):
                return torch.cuda.Event(enable_timing=True)

        ev = [[[(new_event(p_["dev"]), new_event(p_["dev"])) for _ in st_] for st_, p_ in zip(run_streams, parts)] for _ in range(K)]
        sync_all()
        if world > 1:
            dist.barrier()
        sync_all()
        t0 = time.perf_counter()
        # Two steps are kept in flight per engine, not all K: the engine picks the LDS plan of a launch when it is ENQUEUED from
        # the tree sizes the running chains report (results do not depend on it), so the queue must not run ahead of the job;
        # the wait is for the step before last, while the last one still runs -- the device never idles.
        for s_ in range(K):          # every launch goes out on all GPUs before anything is waited for
            if s_ >= 2:
                for k_ in range(len(engs)):
                    for b_ in range(len(run_streams[k_])):
                        ev[s_ - 2][k_][b_][1].synchronize()
            for k_, e_ in enumerate(engs):
                for b_, st_ in enumerate(run_streams[k_]):
                    ev[s_][k_][b_][0].record(st_)
                e_.run(n_tune, s_ * ips, ips)
                for b_, st_ in enumerate(run_streams[k_]):
                    ev[s_][k_][b_][1].record(st_)
        sync_all()
        if world > 1:
            dist.barrier()
        sync_all()
        wall = time.perf_counter() - t0

        dispatch_ms = [[a.elapsed_time(b) for a, b in ev[s_][0]] for s_ in range(K)]     # per dispatch, what rocprofv3 lists (first block)

        def span_ms(s0, k_=0):   # kernel-busy time of GPU k_ from the start of step s0 to the end of the last step
            n_b = len(run_streams[k_])
            return max(ev[s0][k_][b0][0].elapsed_time(ev[K - 1][k_][b1][1]) for b0 in range(n_b) for b1 in range(n_b))

        kernel_ms = [span_ms(0) / K] * K                                                  # per step, all sub-blocks, first block's GPU
        cts = [e_.counters() for e_ in engs]
        ct = cts[0]
        leaps = [float(c_[:, _abi.CT_LEAPFROGS].sum()) for c_ in cts]
        leap_local = leaps[0]
        tail = tail_block(engs[0], ct, span_ms(0) / 1e3, nst, target, step, kw, start)
        for e_ in engs:
            status = e_.status()
            if status.any():
                raise SystemExit("chains reported failure status bits: %s" % np.unique(status))
        n_last = min(ips, n_total - n_tune)
        depth_mean = (float(np.concatenate([e_.stat_i32(_abi.STAT_DEPTH, n_total - n_last, n_last).ravel() for e_ in engs]).mean())
                      if n_last > 0 else 0.0)
        div_after = int(sum(c_[:, _abi.CT_DIVS_AFTER_TUNE].sum() for c_ in cts))

        # ESS/sec (the second half of BASELINE.json's metric): split-R-hat / ESS of the post-warm-up draws, computed
        # where they live (HBM) and reduced across GPUs with ONE exchange of per-dimension sufficient statistics -- an
        # all-reduce (RCCL) between ranks, a device-to-device copy of the (3 + 16) x d block in-process: the only
        # collective of the multi-GPU path.
        ess = None
        if keep_trace and not args.no_ess and n_total - n_tune >= 8:
            from littlemcmc_amd import diagnostics as dg

            # one-time costs (loading the code objects of the statistics kernel and of the torch ops of the finalize
            # step, ~0.7 s in a fresh process) are paid on a 64-chain slice first and reported separately
            views = [dg.trace_tensor(e_) for e_ in engs]
            sync_all()
            t_first = time.perf_counter()
            dg.summarize([v_[:64] for v_ in views] if len(views) > 1 else views[0][:64], reduce_device=red_dev)
            sync_all()
            diag_first_s = time.perf_counter() - t_first
            t_ess = time.perf_counter()
            diag = dg.summarize(views if len(views) > 1 else views[0], reduce_device=red_dev)
            sync_all()
            diag_s = time.perf_counter() - t_ess
            s_draw = min(K - 1, -(-trace_begin // ips))
            draw_s = max(span_ms(s_draw, k_) for k_ in range(len(engs))) / 1e3   # kernel-busy time of the steps that produced the kept draws
            e = diag["ess"]
            ess = {"min": float(e.min()), "median": float(e.median()), "rhat_max": float(diag["rhat"].max()),
                   "lag_passes": int(diag.get("lag_passes", 0)),
                   "draw_seconds": draw_s, "chains_total": int(diag["n_chains"] / 2), "draws": n_total - trace_begin,
                   "diagnostics_seconds": diag_s, "diagnostics_process_warmup_seconds": diag_first_s, "definition": diag.get("definition", "")}
        for e_ in engs:
            e_.close()

        if inproc:
            per_rank = [[leaps[k_], span_ms(0, k_) / 1e3, wall, float(p_["hi"] - p_["lo"]), float(p_["dev"])] for k_, p_ in enumerate(parts)]
        else:
            per_rank = all_gather([leap_local, span_ms(0) / 1e3, wall, float(chains), float(local_rank)])
        wall_max, = all_reduce([wall], dist.ReduceOp.MAX)
        leap_all, div_all = all_reduce([sum(leaps), float(div_after)], dist.ReduceOp.SUM)
        if ess is not None:   # post-warm-up time of the slowest rank
            ess["draw_seconds"], ess["diagnostics_seconds"] = all_reduce([ess["draw_seconds"], ess["diagnostics_seconds"]],
                                                                         dist.ReduceOp.MAX)
        method = ("NUTS max_treedepth=%d" % md) if args.kind == "nuts" else "HMC path_length=2"
        label = config_label(target_name, dim, chains_total if args.scaling == "strong" else chains_total // n_units, md, args.kind, args.mass) + tag
        part = ("%d chains on this GPU" % chains) if n_units == 1 else ("%d chains in blocks of ~%d per GPU" % (chains_total, chains))
        return {
            "label": label, "target": target_name, "dim": dim, "start": start, "mass_desc": mass_desc, "rng": RNG_LABEL[rng], "rng_mode": rng,
            "workload": "%s: %d chains x dim %d %s, %s, %s, tune %d + draws %d in %d launches of %d iterations; %s" % (
                label, chains_total, dim, target_desc, method, mass_desc, n_tune, n_total - n_tune, K, ips, part),
            "workload_short": "%s: %d chains x dim %d %s, %s, tune %d + draws %d (%d x %d it)" % (
                label, chains_total, dim, {"ar1": "AR(1) rho=0.9", "std_normal": "std normal", "funnel": "Neal's funnel",
                                           "diag": "diag Gaussian kappa=1e4"}[target_name],
                ("NUTS td%d" % md) if args.kind == "nuts" else "HMC", n_tune, n_total - n_tune, K, ips),
            "K": K, "ips": ips, "chains_total": chains_total, "chains_this_gpu": chains, "n_tune": n_tune, "n_total": n_total,
            "wall": wall_max, "leap_all": leap_all, "leap_local": leap_local, "kernel_ms": kernel_ms,
            "dispatch_ms_avg": float(np.mean(dispatch_ms)), "dispatches_per_step": nst,
            "depth_mean": depth_mean, "div_after": int(div_all), "ess": ess, "tail": tail,
            "per_rank": [{"rank": r, "chains": int(v[3]), "leapfrogs": v[0], "kernel_s": v[1], "wall_s": v[2], "device": int(v[4])}
                         for r, v in enumerate(per_rank)],
        }

    # identity of the BINARY this process loaded (compiled into it: lmc_build_hash), not of the source tree it sits in:
    # counters from profiles/ are quoted only for that very build
    src_hash = _abi.load().lmc_build_hash().decode()

    def roofline(job):
        kern_s = sum(job["kernel_ms"]) / 1e3
        dim = job["dim"]
        rate_local = job["leap_local"] / kern_s                  # this GPU's kernel: leapfrogs per second of kernel time
        flop = FLOP_PER_LEAPFROG_PER_DIM * dim
        extra = 0 if args.mass == "diag" else 8 * dim * dim      # dense: the reference's two float32 d x d sweeps
        key = ("%s:%d" % (job["target"], dim)) + ("" if args.mass == "diag" else ":" + args.mass) + ("" if job["rng_mode"] == "numpy" else ":" + job["rng_mode"])
        prof = pmc_profile(key, src_hash)
        r = {
            "kernel": kernel_name(dim, args.mass),
            "kernel_ms_avg": sum(job["kernel_ms"]) / job["K"], "leapfrogs_per_launch": job["leap_local"] / job["K"],
            "dispatches_per_step": job["dispatches_per_step"], "dispatch_ms_avg": job["dispatch_ms_avg"],
            "launch_note": "a step (launch) is %d concurrent dispatches of the kernel, one per sub-block of chains on its own "
                           "stream; kernel_ms_avg = HIP-event span of the timed region / steps, dispatch_ms_avg = mean event "
                           "time of one dispatch (what rocprofv3 --kernel-trace lists per row)" % job["dispatches_per_step"],
            "flop_per_leapfrog": flop, "flop_model": "26*d FP64 flop per leapfrog incl. amortised U-turn dots (SURVEY.md 8d)",
            "hbm_contract_60d": {"bytes_per_leapfrog": 60 * dim + extra, "GB_per_s": rate_local * (60 * dim + extra) / 1e9,
                                 "frac_of_8TBps": rate_local * (60 * dim + extra) / HBM_PEAK},
            "hbm_contract_28d_read_only": {"bytes_per_leapfrog": 28 * dim, "GB_per_s": rate_local * 28 * dim / 1e9,
                                           "frac_of_8TBps": rate_local * 28 * dim / HBM_PEAK},
            "traffic": None, "traffic_unit": "B per launch", "valu_inst_per_leapfrog": None, "simd_valu_busy": None,
            "pmc_source": None,
        }
        if args.mass == "diag":
            r.update({"bound": "fp64_valu", "achieved": rate_local * flop / 1e12, "peak": FP64_VALU_PEAK / 1e12,
                      "unit": "TFLOP/s", "frac": rate_local * flop / FP64_VALU_PEAK,
                      "note": "the leapfrog State lives in registers/LDS, HBM carries draws and adaptation state only, so "
                              "the HBM contract figures are not a bound (they may exceed the 8 TB/s peak); the kernel is "
                              "limited by per-wave instruction issue at 3-4 waves per SIMD (DESIGN.md section 6)"})
        elif args.mass == "full_adapt":
            # one float32 matrix PER CHAIN: the kernel's own algorithmic traffic is ONE d x d float32 sweep per leapfrog
            # (4 d^2 B; the reference does two, which is what the contract figure above counts) plus the State
            kb = 60 * dim + 4 * dim * dim
            r.update({"bound": "hbm", "achieved": rate_local * kb / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                      "frac": rate_local * kb / HBM_PEAK, "bytes_per_leapfrog": kb,
                      "note": "per-chain matrices stream from HBM / Infinity Cache: 60 d + 4 d^2 bytes per leapfrog "
                              "(DESIGN.md section 9); hbm_contract_60d above is the reference's two-sweep figure"})
        else:
            # ONE matrix shared by all chains is L2 resident: the sweep is vector work, 2 d^2 flop on top of the leapfrog's
            mv = 2 * dim * dim + flop
            r.update({"bound": "fp64_valu", "achieved": rate_local * mv / 1e12, "peak": FP64_VALU_PEAK / 1e12,
                      "unit": "TFLOP/s", "frac": rate_local * mv / FP64_VALU_PEAK, "flop_per_leapfrog": mv,
                      "flop_model": "2 d^2 (matrix sweep, float32 operands accumulated in float64) + 26 d",
                      "note": "a matrix shared by all chains is read from L2; the kernel is vector-issue bound "
                              "(DESIGN.md section 9); the HBM contract figures above are not a bound here"})
        if prof:   # measured on this very build (source hash match)
            r["traffic"] = prof["hbm_bytes_per_leapfrog"] * job["leap_local"] / job["K"]
            r["valu_inst_per_leapfrog"] = prof.get("valu_inst_per_leapfrog")
            r["simd_valu_busy"] = prof.get("simd_valu_busy")
            r["pmc_source"] = prof.get("source")
        return r

    primary = run_job(args.target, args.dim, with_ess=True, rng=args.rng)
    # The other BASELINE.json configurations ride along as `secondary` lines, each timed exactly like the primary (barrier,
    # HIP events on the launch streams, max over ranks) with its own roofline and tail block: north_star's named shape in
    # both RNG modes, C2, C4, and C5 both as K launches and as ONE launch (what sample() does; C5 is tail-bound, so the
    # launch split is part of the result). Only on the default command line (the primary IS C3): a custom primary
    # workload keeps the north_star lines only.
    secondaries = []
    default_primary = (args.target, args.dim, args.chains, args.max_treedepth) == ("ar1", 128, 65536, 10)
    if not args.no_secondary and args.mass == "diag" and args.kind == "nuts":
        if (args.target, args.dim) != ("std_normal", 128):
            secondaries.append(run_job("std_normal", 128, with_ess=False))
            if not args.no_philox_line and args.rng == "numpy":
                secondaries.append(run_job("std_normal", 128, with_ess=False, rng="philox"))
        if default_primary and not args.no_baseline_configs:
            secondaries.append(run_job("std_normal", 64, with_ess=False, job_chains=4096, max_treedepth=10))
            secondaries.append(run_job("diag", 1000, with_ess=False, job_chains=8192, max_treedepth=10))
            secondaries.append(run_job("funnel", 256, with_ess=False, job_chains=16384, max_treedepth=12))
            secondaries.append(run_job("funnel", 256, with_ess=False, job_chains=16384, max_treedepth=12, K=1, ips=K * ips,
                                       tag=" (one launch)"))

    # The literal drop-in call, end to end (N = 1, default primary only): lmc.sample() returning the trace and every statistic as
    # numpy arrays (the reference's sampling.py:207-222) for C3's chains at a shortened recipe (tune 300 + draws 200: 13.5 GiB
    # returned, a few seconds), against the kernel-only time of the very same job and launch schedule with the draws left in
    # HBM. The full-length numbers (67.5 GiB: 3.66 s against 3.29 s) are profiles/r06_sample_e2e.txt.
    e2e = None
    if (rank == 0 and n_units == 1 and not inproc and default_primary and not args.no_e2e and args.mass == "diag"
            and args.kind == "nuts" and args.rng == "numpy"):
        from littlemcmc_amd import sampling as smp

        e_tune, e_draws = 300, 200
        tgt_, _desc = make_target(lmc, args.target, args.dim)
        t0 = time.perf_counter()
        tr_, st_ = lmc.sample(tgt_, args.dim, draws=e_draws, tune=e_tune, chains=args.chains, random_seed=SEED, progressbar=False)
        t_call = time.perf_counter() - t0
        e_leaps = float(st_["tree_size"].sum())
        gib = (tr_.nbytes + sum(v.nbytes for v in st_.values())) / 2.0 ** 30
        del tr_, st_
        seeds_ = smp._derive_seeds(SEED, args.chains)
        start_, step_ = lmc.init_nuts(tgt_, args.dim, random_seed=seeds_)
        eng_ = step_._make_engine(args.chains)
        try:
            eng_.seed(seeds_)
            eng_.set_position(start_)
            eng_.reset_tuning()
            eng_.reserve(e_tune + e_draws, keep_trace=True, trace_begin=e_tune)
            slots_ = eng_.resident_chains()
            per_ = [100, 100, 100, 100, 500] if (slots_ and args.chains >= 6 * slots_) else 100
            eng_.synchronize()
            t0 = time.perf_counter()
            smp._run_job(eng_, e_tune, e_tune + e_draws, per_, False)
            t_kernel = time.perf_counter() - t0
        finally:
            eng_.close()
        e2e = {"call": "lmc.sample(target, %d, draws=%d, tune=%d, chains=%d) -> numpy trace + statistics" % (args.dim, e_draws, e_tune, args.chains),
               "returned_GiB": gib, "wall_s": t_call, "kernel_only_s": t_kernel, "wall_over_kernel": t_call / t_kernel,
               "leapfrogs": e_leaps, "leapfrog_steps_per_s_end_to_end": e_leaps / t_call,
               "note": "wall_s covers engine creation, seeding, pinning the returned arrays (in a helper thread), the job and the "
                       "last window of statistics; the draws are written into the returned array by the sampling kernel itself"}

    if rank == 0:
        value = primary["leap_all"] / primary["wall"]
        ess = primary["ess"]
        out = {
            "metric": "leapfrog-steps/sec (all chains)", "value": value, "unit": "leapfrog-steps/s",
            "n_gpus": n_units, "steps": K, "warmup": W, "ms_per_step": primary["wall"] * 1e3 / K,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": primary["workload"], "workload_short": primary["workload_short"] + (", %s mass" % args.mass),
                "chains_total": chains_total, "chains_this_gpu": chains,
                "dim": args.dim, "target": args.target, "tune": n_tune, "draws": n_total - n_tune,
                "rng": primary["rng"],
                "trace_in_hbm": not args.no_trace, "parallelism": "chain-block x%d (%s scaling)" % (n_units, args.scaling),
            },
            "leapfrogs": primary["leap_all"], "wall_s": primary["wall"], "mean_depth_draws": primary["depth_mean"],
            "ess_per_sec": None if ess is None else {
                "min": ess["min"] / ess["draw_seconds"], "median": ess["median"] / ess["draw_seconds"],
                "min_including_diagnostics": ess["min"] / (ess["draw_seconds"] + ess["diagnostics_seconds"]),
                "median_including_diagnostics": ess["median"] / (ess["draw_seconds"] + ess["diagnostics_seconds"]),
                "ess_min": ess["min"], "ess_median": ess["median"], "rhat_max": ess["rhat_max"],
                "definition": "%s over all %d chains x %d post-warm-up draws; per second of post-warm-up kernel time "
                              "(slowest rank), and per second of kernel + diagnostics time. The definition is this build's own: "
                              "the reference has no ESS / R-hat (ArviZ appears in a docs recipe only), so the figure is checked "
                              "against a numpy restatement and an analytic AR(1) autocorrelation time, not against a reference "
                              "implementation (parity unpinned)" % (
                                  ess["definition"], ess["chains_total"], ess["draws"]),
                "draw_seconds": ess["draw_seconds"], "diagnostics_seconds": ess["diagnostics_seconds"],
                "lag_passes": ess["lag_passes"],
                "diagnostics_process_warmup_seconds": ess["diagnostics_process_warmup_seconds"]},
            "divergences_after_tune": primary["div_after"],
            "roofline": roofline(primary),
            "tail": primary["tail"],
            "per_rank": primary["per_rank"],
            "rccl_ranks": rccl_ranks, "rccl_error": rccl_error, "backend": None if inproc else args.backend,
            "launcher": ("inproc: one process, one engine per GPU (littlemcmc_amd.sample(devices=...)'s path), no process group"
                         if inproc else "ranks: one process per GPU under torch.distributed.run, %s" % args.backend),
            "launcher_fallback": launcher_fallback,
            "source_hash": src_hash, "source_tree_hash": _build.source_hash(),
        }
        if e2e is not None:
            out["sample_e2e"] = e2e
        if secondaries:
            out["secondary"] = []
            for job in secondaries:   # (a counter-based line is separately labelled, never the headline: not the reference's random stream)
                out["secondary"].append({
                    "workload": job["workload"] + ("; COUNTER-BASED MOMENTUM STREAM" if job["rng_mode"] == "philox" else ""),
                    "workload_short": job["workload_short"] + (" [PHILOX momentum stream]" if job["rng_mode"] == "philox" else ""),
                    "rng": job["rng"], "value": job["leap_all"] / job["wall"], "unit": "leapfrog-steps/s",
                    "steps": job["K"], "iters_per_step": job["ips"], "ms_per_step": job["wall"] * 1e3 / job["K"],
                    "chains_total": job["chains_total"], "chains_this_gpu": job["chains_this_gpu"], "dim": job["dim"],
                    "leapfrogs": job["leap_all"], "wall_s": job["wall"], "mean_depth_draws": job["depth_mean"],
                    "divergences_after_tune": job["div_after"], "roofline": roofline(job), "tail": job["tail"],
                    "per_rank": job["per_rank"]})
        if not args.no_cpu_baseline and n_units == 1:   # reported baseline: rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(args.target, args.dim, seeds_all[:64], primary["start"], args.cpu_iters, args.mass)
        # stdout carries ONE compact line (<= LINE_LIMIT bytes); the verbose object -- every note, the full secondary blocks --
        # goes to a side file, never to stderr (the driver's record tails stdout and stderr together)
        detail_path = args.detail_out or os.path.join(ROOT, "bench_detail.json")
        try:
            with open(detail_path, "w") as fh:
                json.dump(out, fh, indent=1)
            detail_rel = os.path.relpath(detail_path, ROOT)
        except OSError as err:
            detail_rel = "not written: %s" % err
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(compact_line(out, detail_rel)) + "\n").encode())
    if group_up:
        if world > 1:
            dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
