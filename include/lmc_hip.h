/*
 * lmc_hip.h -- C ABI of the MI355X-native many-chain HMC/NUTS engine (liblmc_hip.so).
 *
 * This is the drop-in boundary for the hot path of eigenfoo/littlemcmc 0.2.2. The reference has
 * no FFI: its boundary is a set of duck-typed Python contracts. Each entry
 * point below names the reference interface it replaces (paths relative to
 * /root/reference/littlemcmc/); the Python mirror in littlemcmc_amd/ binds them with ctypes
 * (see INTEGRATION.md for the stub a maintainer of the reference would add).
 *
 * Conventions
 *   - plain C: opaque handle, pointers + sizes, int status returns (0 = LMC_OK); no C++ exceptions
 *     and no torch types cross this boundary;
 *   - array arguments may be HOST or DEVICE pointers (copies use hipMemcpyDefault); shapes are
 *     row-major and UNPADDED: positions [chains][dim], per-chain scalars [chains];
 *   - one engine == one GPU == one HIP stream; a handle is not thread-safe;
 *   - all work is enqueued on the engine's stream; calls that return data synchronise it;
 *   - per-chain numerical failures are reported through status bits, never by aborting the launch.
 */
#ifndef LMC_HIP_H_
#define LMC_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LMC_ABI_VERSION 8

/* status codes */
#define LMC_OK 0
#define LMC_ERR_INVALID 1      /* bad argument / unsupported configuration */
#define LMC_ERR_HIP 2          /* a HIP runtime call failed; see lmc_last_error() */
#define LMC_ERR_STATE 3        /* call sequence error (e.g. run before reserve) */

/* step method: nuts.py:32 (NUTS) / hmc.py:30 (HamiltonianMC) */
#define LMC_KIND_NUTS 0
#define LMC_KIND_HMC 1

/* mass matrix: quadpotential.py:148 (QuadPotentialDiagAdapt, float32 momentum draw, adapted during
 * tuning) / quadpotential.py:346 (QuadPotentialDiag, fixed diagonal, float64 momentum draw) */
#define LMC_POT_DIAG_ADAPT 0
#define LMC_POT_DIAG 1
/* dense mass matrices ("Which kernels an engine runs" below for the sizes): quadpotential.py:428 (QuadPotentialFull: float32 covariance, float32
 * momentum by a triangular solve), :388 (QuadPotentialFullInv: mass matrix A given, float64 momentum L n),
 * :471 (QuadPotentialFullAdapt: covariance + Cholesky factor re-estimated while tuning, one matrix per chain) */
#define LMC_POT_FULL 2
#define LMC_POT_FULL_INV 3
#define LMC_POT_FULL_ADAPT 4
/* QuadPotentialFull(cov, dtype="float64") (quadpotential.py:431-444, the dtype argument): float64 covariance, float64
 * velocity (dgemv), float64 momentum solve_triangular(chol.T, float64 normals). Same kernels as FULL_INV (float64 matrix
 * sweeps); the momentum is the sweep of L^-1, formed once on the host in extended precision. */
#define LMC_POT_FULL_F64 5

/* built-in device log-densities (littlemcmc_amd/csrc/lmc_targets.hpp); LMC_TARGET_USER exists only in
 * libraries built with a user target header (littlemcmc_amd/targets.py: UserTarget) */
#define LMC_TARGET_STD_NORMAL 0
#define LMC_TARGET_DIAG_GAUSSIAN 1   /* params: prec[dim] */
#define LMC_TARGET_AR1 2             /* params: {c_end, c_mid, off} */
#define LMC_TARGET_FUNNEL 3
#define LMC_TARGET_NORMAL1D 4        /* params: {loc, scale}; dim must be 1 */
#define LMC_TARGET_USER 5
/* no device functor: the caller evaluates logp_dlogp_func for all chains between two lmc_engine_tick() calls
 * (a batched callable on device memory, e.g. torch-ROCm); all potentials (dense ones up to dim 256) */
#define LMC_TARGET_EXTERNAL 6

/* Summation order of the float32 kinetic energy of the start state, 0.5f * sdot(p, v)
 * (integration.py:63-64 with float32 operands -> numpy -> OpenBLAS cblas_sdot). The value feeds the
 * accept statistic and, through dual averaging, every later step size, so same-seed parity beyond
 * ~1e-6 needs the host BLAS's rounding. NATIVE: exact products summed in float64, rounded once. */
#define LMC_SDOT_NATIVE 0
#define LMC_SDOT_OPENBLAS_SKYLAKEX 1   /* OpenBLAS 0.3.29 sdot_k_SKYLAKEX (AVX-512 hosts); default */
#define LMC_SDOT_OPENBLAS_HASWELL 2    /* OpenBLAS 0.3.29 sdot_k_HASWELL (AVX2 hosts) */

/* Where the momentum draw's normals come from (potential.random(), quadpotential.py:221-224 / :374-376).
 * NUMPY: the reference's own stream -- per-chain legacy MT19937 + polar method in numpy's consumption order (same-seed
 * parity; the default). PHILOX: a counter-based stream, Philox4x32-10 keyed by the chain's seed and counted by
 * (iteration, element), float32 Box-Muller: the throughput mode for callers that do not need the reference's draws
 * (the momentum draw is ~20 % of a depth-3 iteration on the parity stream). Tree uniforms stay on MT19937 in both.
 * Fused diagonal-mass kernels with the built-in densities (in a team of wavefronts every thread draws its own
 * elements: the draw needs no barrier, where the parity stream is produced by wave 0 alone). */
#define LMC_RNG_NUMPY 0
#define LMC_RNG_PHILOX 1

/* LDS plan of the one-wave fused sampling kernels (csrc/lmc_sampler.hpp: PairLds<NS, 1, PL>). SHALLOW keeps the chain's
 * MT19937 state and the cold slots of the trajectory in LDS and subtree-stack level 1; DEEP uses the generator in place
 * (HBM/L2), one cold slot, and the room that frees holds stack level 2 -- faster once trees are deep (C3: +6 %). AUTO (the
 * default): the engine picks the plan of every launch when it is enqueued, from the mean tree size the running chains report
 * (plan SHALLOW for launches that start before iteration 200). Same statements, same results bit for bit under either plan
 * (tests/test_gpu_round5.py); the pinned values exist for A/B runs and for putting either kernel under the oracle. Ignored
 * by team (dim > 256), dense, general and tick kernels. */
#define LMC_LDS_PLAN_AUTO 0
#define LMC_LDS_PLAN_SHALLOW 1
#define LMC_LDS_PLAN_DEEP 2

/* per-chain status bits (lmc_engine_get_status) */
#define LMC_STATUS_BAD_INITIAL_ENERGY 1   /* base_hmc.py:145-148 (ValueError in the reference) */
/* (No bit for math.py:23-24's FloatingPointError: logbern() only sees log-weights of leaves that passed the divergence
 * test |dE| < Emax (nuts.py:358; a NaN dE becomes +inf first, and inf < Emax is false even for Emax = inf), so every
 * log_size entering a merge is finite and log_p cannot be NaN -- the error is unreachable through this path.) */

/* per-draw statistics. f64 slots */
#define LMC_STAT_STEP_SIZE 0        /* "step_size": exp(log_step) after the update (step_sizes.py:94-99) */
#define LMC_STAT_STEP_SIZE_BAR 1    /* "step_size_bar" */
#define LMC_STAT_ACCEPT 2           /* NUTS "mean_tree_accept" (nuts.py:421-425) / HMC "accept" (hmc.py:164) */
#define LMC_STAT_ENERGY_ERROR 3     /* "energy_error" */
#define LMC_STAT_ENERGY 4           /* "energy" */
#define LMC_STAT_MAX_ENERGY_ERROR 5 /* NUTS "max_energy_error" / HMC "path_length" */
#define LMC_STAT_MODEL_LOGP 6       /* "model_logp" */
#define LMC_NUM_STAT_F64 7
/* i32 slots */
#define LMC_STAT_DEPTH 0            /* NUTS "depth" / HMC "n_steps" */
#define LMC_STAT_TREE_SIZE 1        /* NUTS "tree_size" (= leapfrog steps, nuts.py:432) / HMC n_steps */
#define LMC_NUM_STAT_I32 2
/* u8 slots */
#define LMC_STAT_DIVERGING 0        /* "diverging" */
#define LMC_STAT_TUNE 1             /* "tune" */
#define LMC_STAT_ACCEPTED 2         /* HMC "accepted" */
#define LMC_NUM_STAT_U8 3

/* per-chain counters (lmc_engine_get_counters), int64 [chains][LMC_NUM_COUNTERS] */
#define LMC_CT_REACHED_MAX_TREEDEPTH 0   /* nuts.py:218-220 */
#define LMC_CT_DIVS_AFTER_TUNE 1         /* base_hmc.py:171 */
#define LMC_CT_SAMPLES_AFTER_TUNE 2      /* base_hmc.py:183 */
#define LMC_CT_LEAPFROGS 3               /* sum of tree_size / n_steps: the bench metric's numerator */
#define LMC_CT_WAVE_TICKS 4              /* wall-clock ticks (lmc_engine_occupancy: wall_clock_hz) the chain's wavefronts were
                                          * resident in lmc_engine_run() launches: sum over chains / (resident slots x launch
                                          * time) = mean wave-slot occupancy; the reference's analogue is a chain's process
                                          * time (parallel_sampling.py:377-496). Fused diagonal-mass kernels only. */
#define LMC_NUM_COUNTERS 5

typedef struct lmc_engine lmc_engine;

/* Kernel-selection and layout knobs, every one 0 = the engine decides. They exist for tests (replaying the small goldens
 * through the general kernels, the large team, the HBM factorisation) and for A/B measurements; a result never depends on
 * them beyond the tolerance the selected kernel documents. Until ABI 7 these were environment variables read inside the
 * library -- hidden inputs to an ABI that takes everything else through lmc_config; the Python host still honours the
 * variables of the same names (littlemcmc_amd/engine.py: tuning_from_env), but it is the host that reads them. */
typedef struct lmc_tuning {
    int32_t sub_blocks;           /* sub-block streams lmc_engine_run() deals the chains to: 1 .. 4 (LMC_SUB_BLOCKS) */
    int32_t force_general;        /* 1: every shape the general kernels support runs in them (LMC_FORCE_WIDE) */
    int32_t general_team;         /* 16: the general kernels' 16-wavefront team at every shape (LMC_WIDE_TEAM) */
    int32_t run_ns, run_w;        /* shape of the fused sampling kernel, 64 * run_ns * run_w = padded dim (LMC_RUN_SHAPE="ns,w") */
    int32_t dense_coop_off;       /* 1: QuadPotentialFull never takes the cooperative shared-matrix kernel (LMC_DENSE_COOP=0) */
    int32_t dense_cache_rows_p1;  /* rows of the matrix a dense kernel caches in LDS, PLUS ONE (LMC_DENSE_CACHE_ROWS) */
    int32_t dense_lds_slots_p1;   /* leading tree slots a dense kernel keeps in LDS, PLUS ONE (LMC_DENSE_LDS_SLOTS) */
    int32_t chol_hbm;             /* 1: FullAdapt's refresh factorises through HBM at every size (LMC_CHOL_HBM) */
    int32_t reserved[3];          /* must be 0 */
} lmc_tuning;

/* Constructor arguments of the step method: BaseHMC.__init__ (base_hmc.py:32-126) +
 * NUTS.__init__ (nuts.py:103-202) + HamiltonianMC.__init__ (hmc.py:52-138). */
typedef struct lmc_config {
    int32_t abi_version;          /* LMC_ABI_VERSION */
    int32_t device;               /* HIP device ordinal */
    int32_t chains;               /* chains owned by this engine (one wavefront each) */
    int32_t dim;                  /* model_ndim */
    int32_t kind;                 /* LMC_KIND_* */
    int32_t target_family;        /* LMC_TARGET_* */
    int32_t potential;            /* LMC_POT_* */
    int32_t adapt_step_size;      /* bool */
    double target_accept;         /* 0.8 */
    double emax;                  /* 1000 */
    double step_scale;            /* 0.25; initial step = step_scale / dim**0.25 (base_hmc.py:102) */
    double gamma;                 /* 0.05 */
    double k;                     /* 0.75 */
    double t0;                    /* 10 */
    int32_t max_treedepth;        /* 10 */
    int32_t early_max_treedepth;  /* 8 */
    double path_length;           /* 2.0 (HMC) */
    int32_t max_steps;            /* 1024 (HMC) */
    int32_t adaptation_window;    /* 101 (quadpotential.py:156) */
    int32_t lds_levels;           /* subtree-stack levels kept in LDS; 0 = choose automatically */
    int32_t start_energy_sdot;    /* LMC_SDOT_*: summation order of the float32 start-state kinetic energy */
    double adaptation_window_multiplier; /* 1.0: QuadPotentialDiagAdapt's window grows by this factor at every switch (quadpotential.py:243) */
    int32_t rng_mode;             /* LMC_RNG_*; 0 = the reference's stream */
    int32_t mass_f64;             /* QuadPotentialDiagAdapt(dtype="float64") (quadpotential.py:159,175-184): variance, standard deviations
                                   * and the momentum draw stay float64; with LMC_POT_FULL_ADAPT: QuadPotentialFullAdapt(dtype="float64")
                                   * (:484,497-509): float64 covariance, Cholesky factor and momentum solve. 0 = the reference's
                                   * default float32. General kernels (below). */
    int32_t lds_plan;             /* LMC_LDS_PLAN_*: which LDS plan the one-wave sampling kernels run under (results do not
                                   * depend on it). 0 = the engine chooses per launch. */
    int32_t reserved0;            /* must be 0 */
    lmc_tuning tuning;            /* all zero = the engine decides */
} lmc_config;

/* Which kernels an engine runs. The FUSED kernels (one chain = one wavefront or a team of 2 / 4, the tree in registers and
 * LDS) cover dim <= 1024 with diagonal and dim <= 256 with dense mass matrices, float32 adaptive masses. Everything else the
 * reference accepts -- dim up to 16 384 (base_hmc.py:102 has no limit), QuadPotentialFull / FullInv up to dim 2048
 * (quadpotential.py:388-468), QuadPotentialFullAdapt up to dim 1024 (:470-560; the refresh then factorises through HBM),
 * QuadPotentialDiagAdapt / QuadPotentialFullAdapt(dtype="float64") -- runs in the GENERAL kernels (csrc/lmc_wide.hpp: one
 * chain = one wavefront up to dim 512, a workgroup of 16 wavefronts beyond, the tree in the chain's HBM row): the same
 * algorithm, statement for statement, several times slower per leapfrog. Not in the general kernels: LMC_RNG_PHILOX, and LMC_TARGET_EXTERNAL with
 * anything but a float32 diagonal. */
/* Fill *cfg with the reference's defaults for the given shape. */
void lmc_config_defaults(lmc_config* cfg, int32_t chains, int32_t dim);

/* Library-wide: message of the last failure on this thread (engine may be NULL). */
const char* lmc_last_error(const lmc_engine* e);
int32_t lmc_abi_version(void);
/* Identity of THIS binary: the hash of the sources, header and compiler flags it was compiled from
 * (littlemcmc_amd/_build.py: source_hash(), passed as -DLMC_SOURCE_HASH at build time; "unstamped" for a build made
 * any other way). Measurements under profiles/ are only quoted for the binary whose hash they carry. */
const char* lmc_build_hash(void);
/* HIP devices visible to this process (0: none, or no usable HIP runtime). One engine runs on one of them
 * (lmc_config.device); a job on several is one engine per device on a contiguous chain block each -- the reference's
 * `cores` (sampling.py:117-129) with GPUs in the place of worker processes. */
int32_t lmc_device_count(void);
/* 1 if this library was built with the given LMC_TARGET_* family. */
int32_t lmc_has_target(int32_t family);

/* ---- lifecycle: NUTS(...) / HamiltonianMC(...) construction ------------------------------------ */
int lmc_engine_create(const lmc_config* cfg, lmc_engine** out);
void lmc_engine_destroy(lmc_engine* e);
/* Launch on an externally owned hipStream_t (e.g. torch's current stream). NULL = engine's own. On an external stream
 * every lmc_engine_run() is ordered after whatever the caller enqueued on that stream before the call, and the stream is
 * ordered after the run's kernels on return (event waits on the device): the caller's own work on the stream needs no
 * lmc_engine_synchronize(). (With the engine's own stream consecutive runs overlap per sub-block instead.) */
int lmc_engine_set_stream(lmc_engine* e, void* hip_stream);
int lmc_engine_synchronize(lmc_engine* e);

/* ---- plug-in parameters: the closure of the user's logp_dlogp_func (integration.py:40) ----------- */
int lmc_engine_set_target_params(lmc_engine* e, const double* params, int64_t n);

/* ---- potential: QuadPotentialDiagAdapt(n, initial_mean, initial_diag, initial_weight)
 *      (quadpotential.py:151-204) or QuadPotentialDiag(v) (quadpotential.py:349-365).
 *      initial_mean / initial_diag: [dim] (per_chain = 0, shared) or [chains][dim].
 *      Also performs reset(). initial_mean may be NULL for LMC_POT_DIAG. */
int lmc_engine_set_potential(lmc_engine* e, const double* initial_mean, const double* initial_diag,
                             double initial_weight, int32_t per_chain);

/* ---- dense potentials (cfg.potential = LMC_POT_FULL / FULL_INV / FULL_ADAPT). matrix: [dim][dim] row-major,
 *      the same for every chain.
 *      FULL:       QuadPotentialFull(cov) (quadpotential.py:431-444): matrix = covariance, cast to float32 and
 *                  factorised (failure -> LMC_ERR_INVALID, as scipy.linalg.cholesky raises).
 *      FULL_INV:   QuadPotentialFullInv(A) (quadpotential.py:391-402): matrix = A (inverse covariance).
 *      FULL_F64:   QuadPotentialFull(cov, dtype="float64"): matrix = covariance, kept in float64.
 *      FULL_ADAPT: QuadPotentialFullAdapt(n, initial_mean, initial_cov, initial_weight, adaptation_window,
 *                  adaptation_window_multiplier, update_window) (quadpotential.py:474-519); matrix = initial_cov.
 *      initial_mean .. update_window are read for FULL_ADAPT only. Also the state reset_tuning() restores. */
int lmc_engine_set_dense_potential(lmc_engine* e, const double* matrix, const double* initial_mean,
                                   double initial_weight, int32_t adaptation_window,
                                   double adaptation_window_multiplier, int32_t update_window);

/* ---- RNG: np.random.seed(seeds[c]) for every chain (sampling.py:496-497) ------------------------- */
int lmc_engine_seed(lmc_engine* e, const uint32_t* seeds);
/* np.random.get_state()/set_state() of one chain: key[624], pos, has_gauss, cached_gaussian */
int lmc_engine_set_rng_state(lmc_engine* e, int32_t chain, const uint32_t* key, int32_t pos,
                             int32_t has_gauss, double gauss);
int lmc_engine_get_rng_state(lmc_engine* e, int32_t chain, uint32_t* key, int32_t* pos,
                             int32_t* has_gauss, double* gauss);

/* ---- positions: the `q` threaded through step._astep(q) (sampling.py:498,512) ------------------- */
int lmc_engine_set_position(lmc_engine* e, const double* q, int32_t per_chain);
int lmc_engine_get_position(lmc_engine* e, double* q);

/* ---- tuning lifecycle: step.reset_tuning() + iter_count = 0 (base_hmc.py:192-200, sampling.py:503-509) */
int lmc_engine_reset_tuning(lmc_engine* e);
/* Overwrite the dual-averaging state of every chain (step_sizes.py:49-56 fields). */
int lmc_engine_set_dual_average(lmc_engine* e, double log_step, double log_bar, double hbar,
                                int32_t count);

/* ---- sampling: the body of _iter_sample (sampling.py:507-521) for ALL chains ---------------------
 * reserve(): allocate output storage for `capacity` iterations per chain. Draws are stored for iterations
 *        >= trace_begin (0 = all, tune = discard_tuned_samples, < 0 = keep no trace: statistics only).
 * run(): iterations [iter_begin, iter_begin + n_iters) of every chain; iterations with global index
 *        < n_tune are tuning iterations (stop_tuning happens at index n_tune, sampling.py:510-511).
 *        Asynchronous. Chains are independent (sampling.py:131-136: one process per chain in the reference), so
 *        the engine launches them as contiguous sub-blocks on internal streams of their own (FOUR for the fused diagonal-mass
 *        kernels, two for the dense ones, one for the general kernels; lmc_engine_run_streams() returns the number --
 *        size the array for LMC_MAX_RUN_STREAMS): consecutive run() calls chain up per sub-block, and the tail of one
 *        sub-block's launch is covered by the others' next launch. Every other
 *        entry point (and lmc_engine_synchronize) is ordered after all launched sub-blocks; work on the engine's
 *        stream that precedes a run() is ordered before it.
 * run_streams(): the streams run() launches its kernels on (returns their number, at most `capacity` written) -- for
 *        callers that bracket launches with their own timing events. */
/* step_rand (base_hmc.py:46,123,154-155) in the one form that keeps same-seed parity on the device:
 * step_rand = lambda s: s * np.random.uniform(lo, hi) -- one double of the chain's own stream per iteration, drawn
 * between the start state and the trajectory. enable = 0 switches it off (the default). */
int lmc_engine_set_step_jitter(lmc_engine* e, int32_t enable, double lo, double hi);
/* step_rand as an ARBITRARY host function (base_hmc.py:46,123,154-155: `step_size = self._step_rand(step_size)` once per
 * iteration): the caller evaluates it for every chain and hands the results over; the next lmc_engine_run() integrates
 * with step_sizes[chain] (HOST or DEVICE pointer, [chains]) instead of exp(log_step) / exp(log_bar) -- run ONE iteration
 * per call and refresh the values in between. NULL switches back -- to the adapted step sizes, or to the device's own jitter if
 * lmc_engine_set_step_jitter() enabled one (before or while the override was in force). Dual averaging is untouched (the reference adapts the
 * un-jittered step size too). */
int lmc_engine_set_step_sizes(lmc_engine* e, const double* step_sizes);
/* potential.update(sample = the chain's current position, grad, tune) of QuadPotentialDiagAdapt for every chain as a call
 * of its own (quadpotential.py:231-245; the fixed potentials' update is the base class's `pass`, :112-118): the device
 * function lmc_engine_run() applies after every tuning iteration. The dense counterpart is lmc_engine_dense_update(). */
int lmc_engine_diag_update(lmc_engine* e, int32_t tune);
int lmc_engine_reserve(lmc_engine* e, int64_t capacity, int64_t trace_begin);
/* Where the draws go, decided after reserve(capacity, trace_begin < 0) and possibly while the job is already running (the
 * kernel arguments of a launch are fixed when it is enqueued: launches enqueued AFTER this call store the draws of iterations
 * >= trace_begin, so call it before enqueueing the first launch that reaches trace_begin).
 *   dst != NULL: the caller's own array [chains][capacity - trace_begin][dim] in device-accessible memory (page-locked host
 *        memory: lmc_host_alloc / lmc_host_register; or device memory). The sampling kernel stores every draw THERE, one
 *        coalesced row per chain per iteration, straight over the host link as it is produced -- the array sample() returns
 *        (sampling.py:207-222) is complete when the last launch is, with no copy and no trace in HBM. The engine never frees
 *        it; lmc_engine_get_trace() / trace_device_ptr() read through it.
 *   dst == NULL: the engine allocates the trace in HBM (what reserve(capacity, trace_begin) does). */
int lmc_engine_attach_trace(lmc_engine* e, double* dst, int64_t trace_begin);
/* KeyboardInterrupt (sampling.py:324-328, :470-471: the reference keeps what has been drawn so far). stop = 1: every
 * chain leaves the launch it is in at the end of an iteration within ~16 iterations, and a workgroup that STARTS under the
 * request does nothing at all -- launches still queued neither run an iteration nor touch iter_count, and chains of the
 * current launch whose wavefronts had not started yet stay where the previous launch left them. The stop word is pinned
 * host memory that one chain in at most 256 of a launch (taking turns) polls and copies to a device word, so the request
 * needs nothing scheduled on the device. Chains end at different iterations: lmc_chain_state.iter_count says where; draws
 * and statistics below its MINIMUM are complete for every chain. Consequence for callers: with many more chains than
 * resident wavefront slots (lmc_engine_occupancy), a launch is the granularity at which the not-yet-started chains are
 * cut off -- littlemcmc_amd.sample() therefore cuts such jobs into launches of at most 500 iterations. stop = 0 re-arms
 * the engine, ordered after everything launched so far. Fused, dense and general kernels; a tick-driven job stops by not
 * ticking. */
int lmc_engine_request_stop(lmc_engine* e, int32_t stop);
/* The `callback` of the reference's drivers (sampling.py:272-277, :307-308: called per draw, "sampling can be interrupted
 * by throwing a KeyboardInterrupt in the callback") needs to know where a running job is WITHOUT waiting for it:
 * progress() is the iteration index a relay chain of the launch in flight last started, read from pinned host memory (no
 * stream is touched; updated every 16th iteration). A hint: chains advance at their own pace; what every chain has
 * completed is lmc_chain_state.iter_count. */
int64_t lmc_engine_progress(lmc_engine* e);
int lmc_engine_run(lmc_engine* e, int64_t n_tune, int64_t iter_begin, int32_t n_iters);
#define LMC_MAX_RUN_STREAMS 8
int lmc_engine_run_streams(lmc_engine* e, void** streams, int32_t capacity);
/* The LDS plan (LMC_LDS_PLAN_SHALLOW / _DEEP) of the most recent lmc_engine_run() launch; 0 before the first launch and for
 * engines whose kernels have one plan only. lmc_engine_run_lds_bytes() reports the bytes of that launch. */
int32_t lmc_engine_last_run_plan(lmc_engine* e);

/* ---- results (synchronise the stream). dst shapes: trace [chains][n_iters][dim]; stats [chains][n_iters] */
int lmc_engine_get_trace(lmc_engine* e, double* dst, int64_t iter_begin, int64_t n_iters);
int lmc_engine_get_stat_f64(lmc_engine* e, int32_t stat, double* dst, int64_t iter_begin, int64_t n_iters);
int lmc_engine_get_stat_i32(lmc_engine* e, int32_t stat, int32_t* dst, int64_t iter_begin, int64_t n_iters);
int lmc_engine_get_stat_u8(lmc_engine* e, int32_t stat, uint8_t* dst, int64_t iter_begin, int64_t n_iters);
/* ---- results, streamed: sampling.py:207-222 hands the caller host arrays; a job's draws are tens of GiB (C3: 64 GiB), so the
 * copy must not wait for the job. copy_window_async() enqueues, on a high-priority copy stream of the engine's own, the device->host
 * copy of iterations [iter_begin, iter_begin + n_iters) of EVERY chain into the caller's final arrays, ordered after every
 * lmc_engine_run() enqueued so far and asynchronous to the host and to later launches (the D2H of launch k runs under
 * launch k + 1). Destinations are laid out like the reference's results -- trace [chains][n_out][dim], planes [chains][n_out] --
 * and iteration `first` lands in row 0; a plane is one statistic out of the per-draw records, converted on the device to the
 * dtype the reference's stats dict carries (nuts.py:87-101, hmc.py:36-50). The copies are kernels that write the
 * destinations themselves (coalesced stores over the host link, no per-row DMA command), so the destinations must be
 * DEVICE-ACCESSIBLE: page-locked host memory (lmc_host_alloc, or memory registered with hipHostRegister) or device
 * memory; pageable memory is refused with LMC_ERR_INVALID. copy_wait() waits for the copies enqueued so far (and
 * nothing else). */
#define LMC_PLANE_F64 0      /* idx = LMC_STAT_* f64 slot */
#define LMC_PLANE_I32 1      /* idx = LMC_STAT_DEPTH / LMC_STAT_TREE_SIZE */
#define LMC_PLANE_U8 2       /* idx = LMC_STAT_DIVERGING / _TUNE / _ACCEPTED */
#define LMC_AS_NATIVE 0      /* float64 / int32 / uint8 as stored */
#define LMC_AS_F64 1         /* written as float64 (the reference's "tree_size" is float64) */
#define LMC_AS_I64 2         /* written as int64 (the reference's "depth", HMC's "n_steps") */
#define LMC_MAX_PLANES 16
typedef struct lmc_window_plane {
    void* dst;               /* [chains][n_out] of the output dtype */
    int32_t kind;            /* LMC_PLANE_* */
    int32_t idx;
    int32_t as;              /* LMC_AS_* */
    int32_t reserved;
} lmc_window_plane;
typedef struct lmc_window_dst {
    int64_t n_out;           /* iterations per chain the destination arrays hold */
    int64_t first;           /* iteration index that lands in destination row 0 */
    double* trace;           /* [chains][n_out][dim], or NULL */
    int32_t n_planes;
    int32_t copy_workgroups; /* single-wavefront workgroups a window copy may use; 0 = the default (64): enough to saturate the
                              * host link, few enough to leave the wave slots to the sampling launches it runs under */
    lmc_window_plane plane[LMC_MAX_PLANES];
} lmc_window_dst;
int lmc_engine_copy_window_async(lmc_engine* e, const lmc_window_dst* dst, int64_t iter_begin, int64_t n_iters);
int lmc_engine_copy_wait(lmc_engine* e);
/* Page-locked host memory every visible GPU can copy into (hipHostMalloc, portable): what the arrays sample() returns live
 * in. NULL on failure (lmc_last_error(NULL)); free with lmc_host_free. */
void* lmc_host_alloc(uint64_t bytes);
void lmc_host_free(void* p);
/* Page-lock (and map into every GPU) a range of memory the caller already owns -- e.g. an anonymous mapping it has
 * pre-faulted from several threads: pinning is dominated by the kernel zeroing fresh pages, which one thread does at
 * ~12 GiB/s; disjoint ranges may be registered concurrently from several threads. Unregister before unmapping. */
int lmc_host_register(void* p, uint64_t bytes);
int lmc_host_unregister(void* p);

/* Device pointers of the engine-owned outputs for zero-copy consumers; valid until the next reserve()/destroy().
 * Trace: [chains][capacity - trace_begin][dim] float64. Statistics: [chains][capacity] records of LMC_STAT_RECORD_BYTES
 * = 64 bytes, one per draw, written by the sampling kernel in one coalesced store:
 *     bytes  0..55  the seven float64 statistics in LMC_STAT_* (f64) order
 *     bytes 56..59  int32  tree_size (NUTS) / n_steps (HMC): the leapfrog steps of the draw
 *     bytes 60..63  uint32 bits 0-15 NUTS depth, bit 16 diverging, bit 17 tune, bit 18 accepted
 * (lmc_engine_get_stat_*() gather one statistic out of the records into a dense [chains][n_iters] array.) */
#define LMC_STAT_RECORD_BYTES 64
void* lmc_engine_trace_device_ptr(lmc_engine* e);
void* lmc_engine_stat_records_device_ptr(lmc_engine* e);
int64_t lmc_engine_trace_begin(lmc_engine* e);
int64_t lmc_engine_capacity(lmc_engine* e);

/* adaptation state: potential._var [chains][dim] (float32), step_adapt fields [chains][4] =
 * {log_step, log_bar, hbar, mu}, step_adapt._count [chains], potential._n_samples [chains].
 * Any pointer may be NULL. */
int lmc_engine_get_adapt_state(lmc_engine* e, float* var, double* dual_avg, int32_t* da_count,
                               int32_t* n_samples);
/* Full adaptation state of every chain, for checkpoint/resume and for per-iteration parity tests
 * (the reference has no such call: its state is the Python step object, base_hmc.py:192-200).
 * Every pointer may be NULL (skipped). Shapes: vectors [chains][dim], scalars [chains].
 * set(): inv_std is re-derived from var exactly as quadpotential.py:226-229 does (float32 sqrt, 1/x). */
typedef struct lmc_chain_state {
    float* var;             /* potential._var */
    double* fore_mean;      /* potential._foreground_var.mean */
    double* fore_raw_var;   /* potential._foreground_var.raw_var */
    double* back_mean;      /* potential._background_var.mean */
    double* back_raw_var;   /* potential._background_var.raw_var */
    double* fore_w_sum;     /* potential._foreground_var.w_sum */
    double* back_w_sum;     /* potential._background_var.w_sum */
    int32_t* n_samples;     /* potential._n_samples */
    double* log_step;       /* step_adapt._log_step */
    double* log_bar;        /* step_adapt._log_bar */
    double* hbar;           /* step_adapt._hbar */
    int32_t* da_count;      /* step_adapt._count */
    int32_t* iter_count;    /* step.iter_count */
    int32_t* window;        /* potential.adaptation_window (grows by the multiplier at every switch) */
    double* var64;          /* potential._var of a float64 QuadPotentialDiagAdapt (general kernels only; LMC_ERR_STATE otherwise) */
} lmc_chain_state;
int lmc_engine_get_chain_state(lmc_engine* e, const lmc_chain_state* dst);
int lmc_engine_set_chain_state(lmc_engine* e, const lmc_chain_state* src);
/* Dense-potential state of every chain (FULL_ADAPT: per chain; FULL / FULL_INV: get() replicates the shared
 * matrices). Matrices [chains][dim][dim] row-major in the reference's orientation, vectors [chains][dim],
 * scalars [chains]; NULL = skipped. set() is for checkpoint / resume and per-iteration parity tests. */
typedef struct lmc_dense_state {
    float* cov;              /* potential._cov (FULL_INV: float32 of A^-1, get only) */
    float* chol;             /* potential._chol, lower (FULL_INV: float32 of L, get only) */
    double* fore_mean;       /* potential._foreground_cov.mean */
    double* fore_raw_cov;    /* potential._foreground_cov.raw_cov */
    double* fore_n;          /* potential._foreground_cov.n_samples */
    double* back_mean;
    double* back_raw_cov;
    double* back_n;
    int32_t* window;         /* potential._adaptation_window */
    int32_t* previous_update;/* potential._previous_update */
    int32_t* chol_failures;  /* refreshes whose factorisation failed: potential._chol_error is not None */
    double* cov64;           /* cov / chol as float64: what QuadPotentialFullAdapt(dtype="float64") (cfg.mass_f64) holds; the */
    double* chol64;          /* float32 potentials' values widened. set(): either form of a matrix, not both */
} lmc_dense_state;
int lmc_engine_get_dense_state(lmc_engine* e, const lmc_dense_state* dst);
int lmc_engine_set_dense_state(lmc_engine* e, const lmc_dense_state* src);
/* cov / chol [dim][dim] of ONE chain (the host mirror of step.potential after sample()). */
int lmc_engine_get_dense_chain(lmc_engine* e, int32_t chain, float* cov, float* chol);
int lmc_engine_get_dense_chain_f64(lmc_engine* e, int32_t chain, double* cov, double* chol);   /* (float64 potentials: unrounded) */
/* The lower Cholesky factor [dim][dim] in float64 of the float64 potentials: QuadPotentialFull(cov, dtype="float64")._chol
 * (quadpotential.py:441-443) / QuadPotentialFullInv.L (:402), as the device holds it. */
int lmc_engine_get_dense_factor_f64(lmc_engine* e, double* chol);
/* Test entry: potential.update(sample = current position, grad, tune) for every chain (quadpotential.py:528-552).
 * During lmc_engine_run() the same kernel runs after every tuning iteration. */
int lmc_engine_dense_update(lmc_engine* e, int32_t tune);

/* ---- externally evaluated density (cfg.target_family = LMC_TARGET_EXTERNAL): the iteration loop of
 * sampling.py:507-521 / base_hmc.py:140-190 cut at the two places the reference calls logp_dlogp_func
 * (integration.py:62 compute_state, :115 _step). Protocol:
 *     lmc_engine_reserve(); lmc_engine_tick_begin(n_tune, iter_begin, n_iters);
 *     do { evaluate (logp[chains], grad[chains][dim]) at lmc_engine_tick_positions() [chains][dim];
 *          lmc_engine_tick(logp, grad, &n_active); } while (n_active > 0);
 * All three arrays are DEVICE memory. Every tick advances every unfinished chain by exactly one density
 * evaluation, whatever iteration or tree depth it is in; finished chains ignore their inputs. Passing
 * n_active = NULL skips the host synchronisation (poll every few ticks instead). */
int lmc_engine_tick_begin(lmc_engine* e, int64_t n_tune, int64_t iter_begin, int32_t n_iters);
void* lmc_engine_tick_positions(lmc_engine* e);
int lmc_engine_tick(lmc_engine* e, const double* logp, const double* grad, int32_t* n_active);

/* Running per-chain moments of the post-warm-up draws, kept on the device so that cross-chain R-hat needs no
 * trace (the only quantities the multi-GPU gather moves): mean [chains][dim], m2 = sum of squared
 * deviations [chains][dim], n [chains]. Enable before reset_tuning(); reset with it. */
int lmc_engine_keep_moments(lmc_engine* e, int32_t enable);
int lmc_engine_get_moments(lmc_engine* e, double* mean, double* m2, int32_t* n);
int lmc_engine_get_status(lmc_engine* e, int32_t* status);
int lmc_engine_get_counters(lmc_engine* e, int64_t* counters);

/* ---- unit entry points --------------------------------------------------------------------------
 * trajectory(): integrator.compute_state(q0, p0) followed by n_fwd steps of +eps and n_back steps of
 * -eps (integration.py:52-121), every chain from its own (q0, p0) [chains][dim]. Outputs hold
 * n_fwd + n_back + 1 states: q/p/v/g [chains][n_states][dim], energy/logp [chains][n_states].
 * p0_is_f32 selects the float32 start-state dtype flow of QuadPotentialDiagAdapt. */
int lmc_engine_trajectory(lmc_engine* e, const double* q0, const double* p0, int32_t p0_is_f32,
                          double eps, int32_t n_fwd, int32_t n_back, double* out_q, double* out_p,
                          double* out_v, double* out_g, double* out_energy, double* out_logp);
/* logp_dlogp_func(q) for every chain: q [chains][dim] -> logp [chains], grad [chains][dim]. */
int lmc_engine_logp_dlogp(lmc_engine* e, const double* q, double* logp, double* grad);
/* Draw from every chain's stream: ops[i] > 0 -> normal(size=ops[i]); ops[i] < 0 -> -ops[i] uniforms.
 * out [chains][sum |ops|]. */
int lmc_engine_rng_draw(lmc_engine* e, const int32_t* ops, int32_t n_ops, double* out);
/* potential.random() for every chain (quadpotential.py:221-224 / :374-376): out [chains][dim]. */
int lmc_engine_draw_momentum(lmc_engine* e, double* out);

/* ---- a user-written density compiled at run time (cfg.target_family = LMC_TARGET_USER in the stock library) ------------
 * The kernels that depend on the density functor -- run_kernel<run_ns, run_w, UserTarget>, trajectory_kernel<unit_ns,
 * UserTarget>, logp_kernel<unit_ns, UserTarget> (csrc/lmc_sampler.hpp, csrc/lmc_unit_kernels.hpp) -- are compiled by the
 * caller with hiprtc for the shape lmc_engine_kernel_shape() reports, and handed over as a code object plus the three
 * (lowered) kernel names; everything else stays the prebuilt library. This is how `logp_dlogp_func` (integration.py:
 * 40,62,115) becomes a device function linked into the leapfrog kernel without hipcc on the machine. Diagonal mass
 * matrices only; littlemcmc_amd.targets.UserTarget drives it. */
int lmc_engine_kernel_shape(lmc_engine* e, int32_t* unit_ns, int32_t* run_ns, int32_t* run_w);
/* 1 if this engine runs the general kernels ("Which kernels an engine runs", above): one wavefront per chain up to dim 512
 * (run_w = 1), sixteen beyond (run_w = 16); a run-time compiled density then instantiates run_wide_kernel<run_ns, run_w,
 * UserTarget>, wide_trajectory_kernel and wide_logp_kernel instead of the fused kernels. */
int32_t lmc_engine_uses_general_kernels(lmc_engine* e);
/* How the sampling kernel of this engine occupies the GPU (asked of the HIP runtime for the very kernel, block size and
 * dynamic LDS lmc_engine_run() launches with): resident_chains = chains that run concurrently (compute units x
 * workgroups per unit; 0 for engines without a fused sampling kernel), waves_per_chain, and the rate of the constant
 * wall clock LMC_CT_WAVE_TICKS counts in. The host uses it to size launches (the reference's analogue is `cores`,
 * sampling.py:117-129). Any pointer may be NULL. */
int lmc_engine_occupancy(lmc_engine* e, int32_t* resident_chains, int32_t* waves_per_chain, double* wall_clock_hz);
/* Dynamic LDS bytes per workgroup (= per chain) of the sampling kernel lmc_engine_run() launches; rocprofv3's kernel trace
 * lists only the static part (0). Negative: no fused diagonal-mass kernel. */
int32_t lmc_engine_run_lds_bytes(lmc_engine* e);
int lmc_engine_load_user_kernels(lmc_engine* e, const void* code_object, const char* run_name, const char* trajectory_name,
                                 const char* logp_name);
/* Optional, after lmc_engine_load_user_kernels(): the name of run_kernel<NS, 1, UserTarget, 0, 1> in the same code object -- the
 * one-wave sampling kernel under the deep-tree LDS plan (stack level 2 in LDS, MT19937 used in place; csrc/lmc_sampler.hpp).
 * With it the engine chooses the plan of every launch from the tree sizes the running chains report, as it does for the
 * built-in densities (results do not depend on the choice); without it a run-time compiled density runs under plan 0. */
int lmc_engine_load_user_run_plan1(lmc_engine* e, const char* run_name_plan1);

/* ---- cross-chain diagnostics on the draws in HBM (the reference has none: ArviZ recipe only,
 * docs/tutorials/framework_cookbook.rst:201-213) -------------------------------------------------------------------
 * Per-dimension sufficient statistics of the sub-series [t0, t0 + n) of every chain of a trace block
 * x[chains][draws_stride][dim] (DEVICE pointer, e.g. lmc_engine_trace_device_ptr()):
 *   out[0][dim] = sum over chains of the chain mean          out[1][dim] = sum of squared chain means
 *   out[2][dim] = sum of unbiased within-chain variances (lag0 == 0 only, else 0)
 *   out[3 + k][dim] = sum of biased (1/n) autocovariances at lag lag0 + k, k < lmc_diag_lags_per_pass()
 * out is a DEVICE pointer to (3 + lags_per_pass) * dim doubles. The work is enqueued on `stream` (a hipStream_t, NULL =
 * default stream) of the current device; the result is bit-reproducible (no floating-point atomics). The statistics add
 * over chains, chain halves and ranks: split R-hat and the Geyer ESS follow from their sums (littlemcmc_amd/diagnostics.py). */
int lmc_diag_lags_per_pass(void);
int lmc_diag_chain_stats(const double* x, int64_t chains, int64_t draws_stride, int32_t dim, int64_t t0, int64_t n,
                         int32_t lag0, double* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LMC_HIP_H_ */
