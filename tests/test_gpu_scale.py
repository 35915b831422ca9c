"""-m gpu: BASELINE.json-size runs checked through size-independent properties: posterior moments,
tree invariants, chain-block prefix stability (the multi-GPU partition), checkpoint/resume idempotence,
launch-slicing invariance, divergence-heavy ragged trees."""
import numpy as np
import pytest

import littlemcmc_amd as lmc
from littlemcmc_amd import _abi
from littlemcmc_amd import targets as T

pytestmark = pytest.mark.gpu


def tree_invariants(stats, max_treedepth):
    depth = stats["depth"][..., 0]
    size = stats["tree_size"][..., 0]
    div = stats["diverging"][..., 0]
    assert depth.min() >= 1 and depth.max() <= max_treedepth
    # a tree of depth k holds 2^k - 1 leapfrogs unless its last doubling stopped early (turn/divergence inside)
    assert np.all(size <= 2.0 ** depth - 1)
    assert np.all(size >= 2.0 ** (depth - 1))
    assert np.all(np.isfinite(stats["energy"])) and np.all(stats["mean_tree_accept"] >= 0)
    assert np.all(stats["mean_tree_accept"][~div[..., None]] <= 1 + 1e-12)
    return depth, size, div


def test_c2_4096_chains_dim64_std_normal_moments():
    """configs[1]: 4096 chains, dim 64 standard normal, NUTS max_treedepth=10."""
    from tests._gpu_util import assert_selected_chains_match_oracle

    d, chains, tune, draws = 64, 4096, 300, 200
    trace_all, stats_all = lmc.sample(T.StdNormal(d), d, draws=draws, tune=tune, chains=chains, random_seed=20260928,
                                      max_treedepth=10, discard_tuned_samples=False)
    # first / middle / last chains of the full-size job (both sub-blocks, highest chain_begin) ARE the oracle's chains on
    # the same global seeds: 12 iterations, integer statistics exact, positions to 1e-7
    seeds = lmc.distributed.global_seeds(20260928, chains)
    start = lmc.init_nuts(T.StdNormal(d), d, random_seed=seeds)[0]
    sel = [0, 1, chains // 2 - 1, chains // 2, chains - 2, chains - 1]
    done = assert_selected_chains_match_oracle("std_normal", d, seeds, start, sel, 12, trace_all, stats_all,
                                               okw={"max_treedepth": 10}, label="C2")
    print("C2 full size: oracle-identical iterations of chains %s: %s" % (sel, done))
    assert min(done) >= 11 and sum(done) >= 12 * len(sel) - 2, done
    trace = trace_all[:, tune:]
    stats = {k: v[:, tune:] for k, v in stats_all.items()}
    assert trace.shape == (chains, draws, d)
    tree_invariants(stats, 10)
    # 4096 x 200 draws per dimension: moments to ~3e-3; per-chain statistics pooled
    assert np.abs(trace.mean(axis=(0, 1))).max() < 5e-3
    assert np.abs(trace.var(axis=(0, 1)) - 1).max() < 1e-2
    assert abs(stats["mean_tree_accept"].mean() - 0.8) < 0.05        # dual averaging hit its target
    assert stats["diverging"].sum() == 0
    assert np.median(stats["depth"]) == 3                           # depth 3, 7 leapfrogs per draw


def test_chain_block_prefix_stability_and_launch_slicing():
    """The first K chains of a big run equal a K-chain run (what makes chain-block partitioning across
    GPUs exact), and cutting a run into launches of any length does not change a single bit."""
    d, tune, draws = 16, 60, 20
    tgt = T.AR1(d, 0.9)
    seeds = lmc.distributed.global_seeds(7, 96)
    start, step = lmc.init_nuts(tgt, d, random_seed=seeds)
    full, sfull = lmc.sample(tgt, d, draws=draws, tune=tune, chains=96, random_seed=seeds, start=start, step=step,
                             launch_iters=1000)
    lo, hi = lmc.distributed.chain_block(96, 1, 3)
    _s, step2 = lmc.init_nuts(tgt, d, random_seed=seeds)
    part, spart = lmc.sample(tgt, d, draws=draws, tune=tune, chains=hi - lo, random_seed=seeds[lo:hi], start=start,
                             step=step2, launch_iters=7)
    np.testing.assert_array_equal(part, full[lo:hi])
    for k in sfull:
        np.testing.assert_array_equal(spart[k], sfull[k][lo:hi])


def test_sub_block_streams_do_not_change_a_bit(monkeypatch):
    """lmc_engine_run launches the chains as four contiguous sub-blocks on four internal streams (two until round 5; the
    tail of one sub-block's launch is covered by the others' next launches). Same job, launched as ONE block and as TWO:
    identical draws, statistics and adaptation
    state -- with reads, position pushes and other entry points interleaved between the launches."""
    d, tune, draws, chains = 48, 70, 30, 513
    tgt = T.AR1(d, 0.9)
    seeds = lmc.distributed.global_seeds(11, chains)

    def job(sub_blocks):
        if sub_blocks is None:
            monkeypatch.delenv("LMC_SUB_BLOCKS", raising=False)
        else:
            monkeypatch.setenv("LMC_SUB_BLOCKS", str(sub_blocks))
        start, step = lmc.init_nuts(tgt, d, random_seed=seeds)
        eng = step._make_engine(chains)
        assert len(eng.run_streams()) == (4 if sub_blocks is None else sub_blocks)
        eng.seed(seeds); eng.set_position(start); eng.reset_tuning()
        eng.reserve(tune + draws, keep_trace=True, trace_begin=0)
        eng.run(tune, 0, 13)
        eng.run(tune, 13, 40)                 # back to back: chained per sub-block, no host sync
        mid = eng.trace(20, 10)               # a read between launches is ordered after both halves
        eng.run(tune, 53, 17)
        eng.run(tune, 70, 30)
        out = (eng.trace(0, tune + draws), eng.stat_i32(_abi.STAT_DEPTH, 0, tune + draws), eng.counters(),
               eng.stat_f64(_abi.STAT_STEP_SIZE, 0, tune + draws), mid)
        eng.close()
        return out

    a = job(None)
    for b in (job(1), job(2)):
        for x, y in zip(a, b):
            if x.ndim == 2 and x.shape[1] == _abi.NUM_COUNTERS:     # the residency clock (CT_WAVE_TICKS) is a timing, not a result
                x, y = x[:, :_abi.CT_WAVE_TICKS], y[:, :_abi.CT_WAVE_TICKS]
            np.testing.assert_array_equal(x, y)
    np.testing.assert_array_equal(a[4], a[0][:, 20:30])


def test_sub_block_streams_dense_mass(monkeypatch):
    """The dense-mass kernels go out as sub-blocks too (with FullAdapt: run kernel + covariance refresh per tuning
    iteration, chained per sub-block): same draws and statistics as launched in one block."""
    d, tune, draws, chains = 10, 40, 12, 300
    tgt = T.AR1(d, 0.9)

    def job(sub_blocks):
        if sub_blocks is None:
            monkeypatch.delenv("LMC_SUB_BLOCKS", raising=False)
        else:
            monkeypatch.setenv("LMC_SUB_BLOCKS", str(sub_blocks))
        return lmc.sample(tgt, d, draws=draws, tune=tune, chains=chains, init="jitter+adapt_full", random_seed=3,
                          discard_tuned_samples=False)

    (ta, sa), (tb, sb) = job(None), job(1)
    np.testing.assert_array_equal(ta, tb)
    for k in sa:
        np.testing.assert_array_equal(sa[k], sb[k])


def test_checkpoint_resume_is_bit_identical():
    d, chains = 32, 64
    tgt = T.StdNormal(d)
    seeds = list(range(100, 100 + chains))

    def fresh():
        step = lmc.NUTS(tgt, d)
        eng = step._make_engine(chains)
        eng.seed(seeds)
        eng.set_position(np.full(d, 0.2))
        eng.reset_tuning()
        return eng

    a = fresh()
    a.reserve(80)
    a.run(50, 0, 80)
    want = a.trace()
    b = fresh()
    b.reserve(80)
    b.run(50, 0, 30)
    ckpt = b.get_chain_state()
    pos = b.get_position()
    rng = [b.get_rng_state(c) for c in range(chains)]
    part1 = b.trace(0, 30)
    b.close()
    c = fresh()
    c.set_chain_state(ckpt)
    c.set_position(pos)
    for i, st in enumerate(rng):
        c.set_rng_state(i, st)
    c.reserve(80)
    c.run(50, 30, 50)
    np.testing.assert_array_equal(np.concatenate([part1, c.trace(30, 50)], axis=1), want)
    a.close()
    c.close()


def test_c5_funnel_divergences_and_ragged_trees():
    """configs[4] shape at test size: Neal's funnel d=256, max_treedepth=12: divergence-heavy, ragged."""
    d, chains, tune, draws = 256, 512, 200, 100
    trace, stats = lmc.sample(T.Funnel(d), d, draws=draws, tune=tune, chains=chains, random_seed=3, max_treedepth=12)
    depth, size, div = tree_invariants(stats, 12)
    assert div.sum() > 0                        # the funnel neck diverges
    assert depth.max() > depth.min()            # ragged trees across chains
    assert np.isfinite(trace).all()
    v = trace[..., 0]
    assert abs(v.mean()) < 1.5 and 0.5 < v.std() < 4.0      # q_0 ~ N(0, 3^2) roughly explored


def test_c4_ill_conditioned_dim1000_mass_adaptation():
    """configs[3] shape at test size: d=1000 diagonal Gaussian, kappa=1e4: the adapted mass matrix must
    learn the scales (var_i ~ sigma_i^2) and the draws must have the right variances."""
    d, chains, tune, draws = 1000, 128, 400, 100
    tgt = T.DiagGaussian.ill_conditioned(d, 1e4)
    trace, stats, eng = lmc.sample(tgt, d, draws=draws, tune=tune, chains=chains, random_seed=11, return_engine=True)
    var = eng.adapt_state()["var"]
    eng.close()
    sigma2 = 1.0 / tgt.params
    ratio = np.median(var / sigma2, axis=0)
    assert 0.5 < np.median(ratio) < 1.5
    pooled = trace.var(axis=(0, 1)) / sigma2
    assert 0.8 < np.median(pooled) < 1.2
    tree_invariants(stats, 10)


def test_hmc_c1_four_chains():
    """configs[0]: 4 chains, dim 10 standard normal, HamiltonianMC path_length=2.0."""
    d = 10
    step = lmc.HamiltonianMC(T.StdNormal(d), d, path_length=2.0)
    trace, stats = lmc.sample(T.StdNormal(d), d, draws=1000, tune=1000, step=step, chains=4, cores=4,
                              random_seed=20260928)
    assert trace.shape == (4, 1000, d)
    assert np.abs(trace.mean(axis=(0, 1))).max() < 0.15 and np.abs(trace.var(axis=(0, 1)) - 1).max() < 0.25
    assert np.all(stats["n_steps"] >= 1) and stats["accepted"].mean() > 0.5


def test_headline_shape_65536_chains_dim128_moments_on_device():
    """north_star shape: 65 536 chains x dim 128 standard normal. Posterior moments pooled over 6.5e6 draws per
    dimension are within 2e-3 (5 sigma of the Monte-Carlo error) of the truth, computed where the draws live
    (HBM, zero-copy torch view); cross-chain R-hat < 1.01; ESS close to the number of draws."""
    import torch

    from littlemcmc_amd import diagnostics as dg

    d, chains, tune, draws = 128, 65536, 200, 100
    tgt = T.StdNormal(d)
    seeds = lmc.distributed.global_seeds(20260928, chains)
    start, step = lmc.init_nuts(tgt, d, random_seed=seeds)
    eng = step._make_engine(chains)
    try:
        eng.seed(seeds)
        eng.set_position(start)
        eng.reset_tuning()
        eng.reserve(tune + draws, keep_trace=True, trace_begin=tune)
        eng.run(tune, 0, tune + draws)
        eng.synchronize()
        assert not eng.status().any()
        x = dg.trace_tensor(eng)                      # [chains, draws, d] in HBM
        assert tuple(x.shape) == (chains, draws, d)
        mean = x.mean(dim=(0, 1))
        var = x.var(dim=(0, 1))
        assert float(mean.abs().max()) < 2e-3
        assert float((var - 1).abs().max()) < 4e-3
        diag = dg.summarize(x, chunk=1024)
        assert float(diag["rhat"].max()) < 1.01
        ess = diag["ess"]
        assert float(ess.min()) > 0.5 * chains * draws
        depth = eng.stat_i32(_abi.STAT_DEPTH, tune, draws)
        assert np.median(depth) == 3
        ct = eng.counters()
        assert ct[:, _abi.CT_DIVS_AFTER_TUNE].sum() == 0
        # per-chain step sizes adapted independently but to the same regime (step_bar ~ 0.61 at d=128)
        sb = np.exp(eng.adapt_state()["log_bar"])
        assert 0.45 < np.median(sb) < 0.8 and sb.std() / sb.mean() < 0.2
        del x
        torch.cuda.synchronize()
    finally:
        eng.close()


def test_north_star_shape_moments_within_1e3():
    """Row N of the scope table, north_star's own tolerance on north_star's own shape: 65 536 chains x d = 128 standard
    normal, NUTS defaults, 1000 post-warm-up draws per chain (6.6e7 per dimension: Monte-Carlo error 1.2e-4 on a mean,
    ~2e-4 on a variance). EVERY dimension's pooled mean and variance within 1e-3 of the CPU reference's stationary
    values, which for a standard normal are the target's (0, 1) -- the reference leaves it invariant (DESIGN.md section
    5). Moments come from the kernel's running per-chain accumulators (sampling.py:207-220 is the layout pooled)."""
    d, chains, tune, draws = 128, 65536, 400, 1000
    gmean, gvar = _many_chain_moments(T.StdNormal(d), d, chains, tune, draws)
    print("north_star shape: max |mean| %.2e, max |var - 1| %.2e" % (np.abs(gmean).max(), np.abs(gvar - 1.0).max()))
    assert np.abs(gmean).max() < 1e-3, np.abs(gmean).max()
    assert np.abs(gvar - 1.0).max() < 1e-3, np.abs(gvar - 1.0).max()


class _InterruptAt:
    """callback for sample(): raises KeyboardInterrupt once the DEVICE says the job has reached iteration `at` (the
    engine's progress word, written by the sampling kernel) -- no wall-clock assumption about how fast the job runs."""

    def __init__(self, at):
        self.at, self.fired_at, self.calls = at, None, 0

    def __call__(self, trace, draw):
        self.calls += 1
        if self.fired_at is None and draw.iteration >= self.at:
            self.fired_at = draw.iteration
            raise KeyboardInterrupt


def test_keyboard_interrupt_returns_the_draws_so_far():
    """sampling.py:324-328 / :470-471: Ctrl-C ends sampling and the draws so far are returned. On the device every chain
    leaves its launch at its next iteration boundary (lmc_engine_request_stop); the iterations EVERY chain completed are
    returned and are, bit for bit, the prefix of the uninterrupted job -- every returned row, not a sample of them. The
    interrupt is raised where the reference allows it (sampling.py:277: a KeyboardInterrupt thrown in the callback), when
    the device reports iteration 48 of 100 050: the job is ~1 s long, the stop takes ~16 iterations to arrive."""
    d, chains, tune, draws = 32, 2048, 50, 100000
    tgt = T.StdNormal(d)
    cb = _InterruptAt(48)
    # (stream_results=False: the job is 100 050 iterations long only so that it is still running when the interrupt comes --
    #  streamed, sample() would pin the 69 GB its full length returns before the first launch; the streamed form of an
    #  interrupted job is tests/test_gpu_round6.py's, at a size that is actually returned)
    trace, stats = lmc.sample(tgt, d, draws=draws, tune=tune, chains=chains, random_seed=12, discard_tuned_samples=False,
                              callback=cb, progressbar=False, stream_results=False)
    n = trace.shape[1]
    print("interrupt raised at device iteration %d; %d of %d iterations completed by every chain" % (cb.fired_at, n, tune + draws))
    assert cb.fired_at is not None and 0 < n < tune + draws   # (the hint is where the fastest relay chain is; n what EVERY chain completed)
    assert trace.shape == (chains, n, d) and stats["depth"].shape == (chains, n, 1)
    full, fstats = lmc.sample(tgt, d, draws=max(n - tune, 0), tune=min(n, tune), chains=chains, random_seed=12,
                              discard_tuned_samples=False, progressbar=False)
    # the engine is re-armed by the next job: a fresh sample() after an interrupted one runs to the end
    assert full.shape == (chains, n, d)
    np.testing.assert_array_equal(trace, full)
    for name in fstats:
        np.testing.assert_array_equal(stats[name], fstats[name], err_msg=name)


def test_keyboard_interrupt_with_many_queued_launches_returns_only_written_rows():
    """sample() enqueues every launch of a job up front. The launches still queued when Ctrl-C arrives must do NOTHING:
    a launch that ran even one iteration at its own iter_begin would raise iter_count past rows nobody wrote and pollute
    the tuning counters (found by review in round 3: the stop word was only tested at the END of an iteration). Here the
    job is 500 launches of 8 iterations; after the interrupt every returned row equals the uninterrupted job's, no
    chain's iteration count ran ahead into another launch, and the per-chain counters are those of the rows returned."""
    from littlemcmc_amd import _abi

    d, chains, tune, draws = 16, 1024, 30, 3970
    tgt = T.StdNormal(d)
    cb = _InterruptAt(40)
    trace, stats, eng = lmc.sample(tgt, d, draws=draws, tune=tune, chains=chains, random_seed=5, discard_tuned_samples=False,
                                   callback=cb, progressbar=False, launch_iters=8, return_engine=True)
    try:
        n = trace.shape[1]
        it_c = eng.get_chain_state(fields=("iter_count",))["iter_count"]
        ct = eng.counters()
    finally:
        eng.close()
    print("interrupted at %d: %d rows; iter_count %d..%d" % (cb.fired_at, n, it_c.min(), it_c.max()))
    assert cb.fired_at is not None and 0 < n < tune + draws
    assert it_c.min() == n
    # a chain stops at the end of the iteration in which it saw the request -- inside ONE launch of 8 iterations; the
    # launches behind it did not run, so nobody is more than a launch ahead of the slowest chain
    assert it_c.max() - it_c.min() <= 8 + 8, (it_c.min(), it_c.max())
    np.testing.assert_array_equal(ct[:, _abi.CT_SAMPLES_AFTER_TUNE], np.maximum(it_c - tune, 0))
    full, fstats = lmc.sample(tgt, d, draws=max(n - tune, 0), tune=min(n, tune), chains=chains, random_seed=5,
                              discard_tuned_samples=False, progressbar=False, launch_iters=8)
    np.testing.assert_array_equal(trace, full)
    for name in fstats:
        np.testing.assert_array_equal(stats[name], fstats[name], err_msg=name)


def test_callback_sees_the_job_advance():
    """`callback` (sampling.py:49, :272-277 of the reference) is called as the job advances, with where it is."""
    seen = []
    d = 8
    lmc.sample(T.StdNormal(d), d, draws=3000, tune=1000, chains=256, random_seed=2, progressbar=False,
               callback=lambda trace, draw: seen.append((draw.iteration, draw.tuning, draw.total, draw.chains)))
    its = [s[0] for s in seen]
    assert len(seen) >= 2 and max(its) > 1000 and all(s[2] == 4000 and s[3] == 256 for s in seen)
    assert all((i < 1000) == t for i, t, _n, _c in seen)


def _pooled_moments(mean, m2, n):
    """Pooled mean / variance per dimension from per-chain running moments (mean[c, d], M2[c, d], n[c])."""
    n = np.asarray(n, dtype="d")[:, None]
    tot = n.sum()
    grand = (mean * n).sum(axis=0) / tot
    ss = m2.sum(axis=0) + (n * (mean - grand) ** 2).sum(axis=0)
    return grand, ss / (tot - 1.0)


def _many_chain_moments(tgt, d, chains, tune, draws):
    """Pooled moments of a many-chain NUTS run from the kernel's running per-chain moments (no trace in HBM)."""
    seeds = lmc.distributed.global_seeds(20260928, chains)
    start, step = lmc.init_nuts(tgt, d, random_seed=seeds)
    eng = step._make_engine(chains)
    try:
        eng.seed(seeds)
        eng.set_position(start)
        eng.reset_tuning()
        eng.keep_moments(True)
        eng.reserve(tune + draws, keep_trace=False)
        eng.run(tune, 0, tune + draws)
        eng.synchronize()
        assert not eng.status().any()
        mean, m2, n = eng.moments()
        return _pooled_moments(np.asarray(mean), np.asarray(m2), n)
    finally:
        eng.close()


def test_stationary_moments_are_the_references_not_the_targets(golden_dir):
    """north_star: "posterior moments within 1e-3 of the CPU reference". The reference's NUTS does not leave a
    correlated Gaussian exactly invariant (``_Tree.extend`` aliases p_sum): on AR(1) rho = 0.9 at
    d = 32 its pooled marginal variance is 1.0195 +- 0.0007 (tests/golden/stationary_moments.npz, 3.6e6 draws of the
    imported reference), not 1. The device reproduces THAT number -- the same algorithm, quirk included -- which a
    sampler that merely targets N(0, Sigma) correctly would not."""
    import os

    g = np.load(os.path.join(golden_dir, "stationary_moments.npz"))
    gmean, gvar = _many_chain_moments(T.AR1(32, 0.9), 32, 65536, 400, 600)
    ref, se = float(g["ar1_32_var_avg"]), float(g["ar1_32_var_avg_se"])
    assert abs(gvar.mean() - ref) < 1e-3 + 3 * se, (gvar.mean(), ref, se)
    assert gvar.mean() - 1.0 > 0.015                       # far outside the Monte-Carlo error of an unbiased sampler
    assert np.abs(gmean).max() < 1e-3 + 3 * float(np.abs(g["ar1_32_mean"]).max())


def test_c3_full_size_65536_chains_dim128_ar1(golden_dir):
    """configs[2] at its one-GPU size, through the kernel the benchmark times (run_kernel<2, 1, AR1Target>):
    65 536 chains x d = 128 AR(1) rho = 0.9, NUTS defaults, diagonal mass adaptation. Pooled posterior mean within
    1e-3 of 0 and dimension-averaged marginal variance within 1e-3 (north_star; plus the golden value's own Monte-Carlo
    error) of the CPU reference's stationary value -- which is 1.003, not 1 (tests/golden/capture_moments.py) --,
    cross-chain R-hat < 1.01, no divergences, trees as deep as the reference's (mean depth 6.3)."""
    import os

    from littlemcmc_amd import diagnostics as dg

    g = np.load(os.path.join(golden_dir, "stationary_moments.npz"))
    ref_var, ref_se = float(g["ar1_128_var_avg"]), float(g["ar1_128_var_avg_se"])
    d, chains, tune, draws = 128, 65536, 400, 1000
    tgt = T.AR1(d, 0.9)
    seeds = lmc.distributed.global_seeds(20260928, chains)
    start, step = lmc.init_nuts(tgt, d, random_seed=seeds)
    eng = step._make_engine(chains)
    try:
        eng.seed(seeds)
        eng.set_position(start)
        eng.reset_tuning()
        eng.keep_moments(True)                         # running per-chain moments in the kernel
        # the draws of ALL iterations stay in HBM (94 GB of the 288): the first iterations of the first / middle / last
        # chains of THIS job are compared with the oracle below
        eng.reserve(tune + draws, keep_trace=True, trace_begin=0)
        eng.run(tune, 0, tune + draws)
        eng.synchronize()
        assert not eng.status().any()
        from tests._gpu_util import assert_selected_chains_match_oracle

        n_it = 12
        sel = [0, 1, chains // 2 - 1, chains // 2, chains - 2, chains - 1]
        first = eng.trace(0, n_it)
        fstats = {k: v[:, :, None] for k, v in step._stats_from_engine(eng, 0, n_it).items()}
        done = assert_selected_chains_match_oracle("ar1", d, seeds, start, sel, n_it, first, fstats, label="C3")
        print("C3 full size: oracle-identical iterations of chains %s: %s" % (sel, done))
        assert min(done) >= 11 and sum(done) >= n_it * len(sel) - 2, done    # measured (round 3): 12 of 12 for all six chains
        del first
        mean, m2, n = eng.moments()
        assert (np.asarray(n) == draws).all()
        gmean, gvar = _pooled_moments(np.asarray(mean), np.asarray(m2), n)
        assert np.abs(gmean).max() < 1e-3, np.abs(gmean).max()
        assert abs(gvar.mean() - ref_var) < 1e-3 + 3 * ref_se, (gvar.mean(), ref_var, ref_se)
        assert np.abs(gvar - gvar.mean()).max() < 1.5e-3          # every dimension sits at that level (edges slightly lower)
        rhat = dg.rhat_from_moments(mean, m2, n).cpu().numpy()
        assert rhat.max() < 1.01
        depth = eng.stat_i32(_abi.STAT_DEPTH, tune, draws)
        size = eng.stat_i32(_abi.STAT_TREE_SIZE, tune, draws)
        assert 5.8 < depth.mean() < 6.6, depth.mean()
        assert np.all(size <= 2 ** depth.astype(np.int64) - 1) and np.all(size >= 2 ** (depth.astype(np.int64) - 1))
        ct = eng.counters()
        assert ct[:, _abi.CT_DIVS_AFTER_TUNE].sum() == 0
        assert ct[:, _abi.CT_LEAPFROGS].sum() == eng.stat_i32(_abi.STAT_TREE_SIZE, 0, tune + draws).astype(np.int64).sum()
    finally:
        eng.close()


def test_c4_per_gpu_size_1024_chains_dim1000():
    """configs[3] at its per-GPU size (8192 chains over 8 GPUs): 1024 chains x d = 1000, kappa = 1e4 diagonal
    Gaussian, QuadPotentialDiagAdapt warm-up -- the team-of-4-wavefronts kernel."""
    d, chains, tune, draws = 1000, 1024, 500, 200
    tgt = T.DiagGaussian.ill_conditioned(d, 1e4)
    trace, stats, eng = lmc.sample(tgt, d, draws=draws, tune=tune, chains=chains, random_seed=20260928, return_engine=True)
    var = eng.adapt_state()["var"]
    eng.close()
    tree_invariants(stats, 10)
    sigma2 = 1.0 / tgt.params
    # the adapted mass matrix learnt every scale over four orders of magnitude (500 tuning draws: ~9 % noise per entry)
    assert np.abs(np.median(var / sigma2, axis=0) - 1.0).max() < 0.15
    pooled = trace.var(axis=(0, 1)) / sigma2                      # 2e5 draws per dimension
    assert np.abs(pooled - 1.0).max() < 0.03, np.abs(pooled - 1.0).max()
    assert np.abs(trace.mean(axis=(0, 1)) / np.sqrt(sigma2)).max() < 0.02
    assert stats["diverging"].sum() == 0
    assert abs(stats["mean_tree_accept"].mean() - 0.8) < 0.05


def test_c5_per_gpu_size_2048_chains_dim256_funnel():
    """configs[4] at its per-GPU size (16 384 chains over 8 GPUs): 2048 chains x d = 256 Neal's funnel,
    max_treedepth = 12 -- divergence-heavy, ragged trees of 1 ... 4095 leapfrogs side by side."""
    d, chains, tune, draws = 256, 2048, 300, 200
    trace, stats = lmc.sample(T.Funnel(d), d, draws=draws, tune=tune, chains=chains, random_seed=20260928,
                              max_treedepth=12)
    depth, size, div = tree_invariants(stats, 12)
    assert div.sum() > 0 and depth.max() >= depth.min() + 3
    assert np.isfinite(trace).all()
    # every divergent transition reports an energy error beyond Emax or a non-finite one (nuts.py:358)
    mee = np.abs(stats["max_energy_error"][..., 0])
    assert np.all((mee[div] >= 1000.0) | ~np.isfinite(mee[div]))
    assert np.all(mee[~div] < 1000.0)
    v = trace[..., 0]
    assert abs(v.mean()) < 1.0 and 1.0 < v.std() < 4.0          # q_0 ~ N(0, 3^2): the mouth is explored, the neck under-sampled


def test_same_seed_chains_agree_statistically_with_the_oracle():
    """Whole tuned chains decorrelate from the reference after a few dozen iterations, so
    long same-seed runs are compared as samples: per-chain means of the device and of the oracle (32 chains,
    identical seeds) must be draws from the same distribution."""
    from oracle import lmc_oracle as orc
    from oracle import targets as OT

    d, chains, tune, draws = 8, 32, 200, 300
    gt_all, gs_all = lmc.sample(T.AR1(d, 0.9), d, draws=draws, tune=tune, chains=chains, random_seed=99,
                                discard_tuned_samples=False)
    ot_all, os_all = orc.sample(OT.AR1(d, 0.9), d, draws=draws, tune=tune, chains=chains, random_seed=99,
                                discard_tuned_samples=False)
    gt, ot = gt_all[:, tune:], ot_all[:, tune:]
    gs = {k: v[:, tune:] for k, v in gs_all.items()}
    os_ = {k: v[:, tune:] for k, v in os_all.items()}
    gm, om = gt.mean(axis=1), ot.mean(axis=1)               # [chains, d]
    se = np.sqrt(gm.var(axis=0) / chains + om.var(axis=0) / chains)
    assert np.all(np.abs(gm.mean(axis=0) - om.mean(axis=0)) < 5 * se)
    assert np.all(np.abs(gt.var(axis=(0, 1)) / ot.var(axis=(0, 1)) - 1) < 0.25)
    assert abs(gs["depth"].mean() - os_["depth"].mean()) < 0.3
    assert abs(gs["mean_tree_accept"].mean() - os_["mean_tree_accept"].mean()) < 0.03
    # and the first iterations are the very same chain
    np.testing.assert_array_equal(gs_all["depth"][:, :10, 0], os_all["depth"][:, :10, 0])
    np.testing.assert_allclose(gt_all[:, :10], ot_all[:, :10], rtol=1e-7, atol=1e-9)


def test_dense_adapt_and_tick_paths_sample_the_target_at_scale():
    """Posterior moments at scale for the two widened paths: per-chain dense mass adaptation (4096 chains: 4096
    matrices, estimators and Cholesky factors) and a torch-callable density through ticks (16 384 chains).

    d = 24 for the dense case: the reference's FullAdapt has no regularisation, so a 128 x 128 covariance learnt from
    a few hundred tuning draws is nearly singular and the chains barely leave the jittered start (pooled marginal
    variance 0.41 in the oracle, 0.39 on the device at d = 128, tune = 450) -- faithful, but not a moments test."""
    torch = pytest.importorskip("torch")
    from littlemcmc_amd.targets import TorchTarget

    d, chains = 24, 4096
    idx = np.arange(d)
    cov = 0.9 ** np.abs(idx[:, None] - idx[None, :])
    trace, stats = lmc.sample(lmc.targets.AR1(d, 0.9), d, draws=150, tune=600, chains=chains, init="jitter+adapt_full",
                              random_seed=99)
    x = trace.reshape(-1, d)
    assert np.abs(x.mean(axis=0)).max() < 0.02
    assert np.abs(np.cov(x.T) - cov).max() < 0.03
    assert stats["diverging"].mean() < 1e-3 and np.isfinite(trace).all()
    assert stats["depth"].mean() < 3.6      # the learnt dense metric decorrelates AR(1): short trees

    d2, chains2 = 64, 16384
    tgt = TorchTarget(d2, lambda q: (-0.5 * (q * q).sum(dim=1), -q))
    tr2, st2 = lmc.sample(tgt, d2, draws=40, tune=120, chains=chains2, random_seed=5)
    y = tr2.reshape(-1, d2)
    assert np.abs(y.mean(axis=0)).max() < 0.01 and np.abs(y.var(axis=0) - 1.0).max() < 0.02
    assert st2["depth"].mean() > 1.5 and not st2["diverging"].any()


def test_counter_based_momentum_stream_samples_the_target_and_is_partition_invariant():
    """LMC_RNG_PHILOX (NUTS(momentum_rng="philox")): the throughput mode's momentum draw is a pure function of (chain seed,
    iteration, element). It is NOT the reference's stream -- so no oracle chain to compare with -- and is held to what it
    promises: the draws have the target's moments at scale (65 536 chains x d = 128 standard normal, every dimension within
    2e-3 after 600 draws per chain -- Monte-Carlo error of a variance ~3e-4; d = 200 exercises NS = 4), a chain does not depend
    on how the job is cut into launches or chain blocks, and the parity stream of the same job is untouched by the mode's
    existence."""
    d, chains, tune, draws = 128, 65536, 300, 600
    tgt = T.StdNormal(d)
    seeds = lmc.distributed.global_seeds(20260928, chains)
    start, step = lmc.init_nuts(tgt, d, random_seed=seeds, momentum_rng="philox")
    eng = step._make_engine(chains)
    try:
        eng.seed(seeds); eng.set_position(start); eng.reset_tuning(); eng.keep_moments(True)
        eng.reserve(tune + draws, keep_trace=False)
        eng.run(tune, 0, tune + draws)
        eng.synchronize()
        assert not eng.status().any()
        mean, m2, n = eng.moments()
        gmean, gvar = _pooled_moments(np.asarray(mean), np.asarray(m2), n)
        depth = eng.stat_i32(_abi.STAT_DEPTH, tune, draws)
    finally:
        eng.close()
    print("philox momentum stream: max |mean| %.2e, max |var - 1| %.2e, median depth %g" % (
        np.abs(gmean).max(), np.abs(gvar - 1.0).max(), np.median(depth)))
    assert np.abs(gmean).max() < 2e-3 and np.abs(gvar - 1.0).max() < 2e-3
    assert np.median(depth) == 3

    # launch slicing and chain blocks do not change a chain; wider vectors (NS = 4) and a correlated target
    d2, c2 = 200, 96
    tgt2 = T.AR1(d2, 0.9)
    seeds2 = lmc.distributed.global_seeds(7, c2)
    start2, step2 = lmc.init_nuts(tgt2, d2, random_seed=seeds2, momentum_rng="philox")
    full, sfull = lmc.sample(tgt2, d2, draws=15, tune=40, chains=c2, random_seed=seeds2, start=start2, step=step2, launch_iters=1000)
    lo, hi = lmc.distributed.chain_block(c2, 1, 3)
    _s, step3 = lmc.init_nuts(tgt2, d2, random_seed=seeds2, momentum_rng="philox")
    part, spart = lmc.sample(tgt2, d2, draws=15, tune=40, chains=hi - lo, random_seed=seeds2[lo:hi], start=start2, step=step3, launch_iters=7)
    np.testing.assert_array_equal(part, full[lo:hi])
    np.testing.assert_array_equal(spart["tree_size"], sfull["tree_size"][lo:hi])
    # the two modes draw different momenta (the parity stream is the default and is what every oracle test runs on)
    _s, step4 = lmc.init_nuts(tgt2, d2, random_seed=seeds2)
    ref, _ = lmc.sample(tgt2, d2, draws=15, tune=40, chains=8, random_seed=seeds2[:8], start=start2, step=step4)
    assert not np.allclose(ref, full[:8])
    assert np.isfinite(full).all() and abs(full.var() - 1.0) < 0.5


@pytest.mark.parametrize("kind,d", [("hmc", 10), ("nuts_fixed_diag", 70), ("nuts_team", 600)])
def test_counter_based_momentum_stream_other_paths(kind, d):
    """LMC_RNG_PHILOX through the other instantiations: HamiltonianMC, a fixed diagonal potential (float64 momentum,
    quadpotential.py:374-376) and a team of wavefronts (every thread draws its own elements, no barrier): the draws have the
    target's moments."""
    chains = 2048 if d < 100 else 512
    tgt = T.StdNormal(d)
    if kind == "hmc":
        step = lmc.HamiltonianMC(tgt, d, path_length=2.0, momentum_rng="philox")
    elif kind == "nuts_fixed_diag":
        step = lmc.NUTS(tgt, d, scaling=np.full(d, 1.3), is_cov=True, momentum_rng="philox")
    else:
        step = lmc.NUTS(tgt, d, momentum_rng="philox")
    trace, stats = lmc.sample(tgt, d, draws=300, tune=300, step=step, chains=chains, random_seed=5)
    assert np.isfinite(trace).all() and not stats["diverging"].any()
    n = chains * 300
    assert np.abs(trace.mean(axis=(0, 1))).max() < 6.0 / np.sqrt(n) + 5e-3
    assert np.abs(trace.var(axis=(0, 1)) - 1.0).max() < 0.03
