// The hot path as device code: one chain per wavefront (or per team of wavefronts), whole iterations on the GPU.
//
//   leapfrog        <- /root/reference/littlemcmc/integration.py:52-66,100-121  (+ quadpotential.py:206-219)
//   momentum draw   <- quadpotential.py:221-224 (float32) / :374-376 (float64)
//   NUTS transition <- nuts.py:204-224 (_hamiltonian_step), :251-435 (_Tree), iterative post-order
//   HMC transition  <- hmc.py:140-182
//   dual averaging  <- step_sizes.py:49-99
//   mass adaptation <- quadpotential.py:226-245, :294-340
//   iteration body  <- base_hmc.py:140-190 (_astep)
//
// Storage plan per chain:
//   registers : current state (q,p,g), trajectory ends L/R (q,p,g), p_sum, proposal q, the in-flight subtree
//               node (lp, rp, psum, prop q + weights), float32 mass (var, inv_std) and its float64 promotion,
//               per-level subtree scalars (lane j holds level j), the uniform look-ahead window
//   LDS       : subtree stack levels [0, nlds) (level 0 = {p, q}; level j>0 = {lp, rp, psum, prop q}), the chain's
//               MT19937 state for the duration of the launch, the team exchange area (W > 1)
//   HBM       : subtree stack levels >= nlds (per-chain scratch rows, L2 resident while hot), Welford
//               accumulators, persistent chain state between launches, trace / stat outputs
// Velocities are never stored: v = var (.) p is recomputed (bit-identical, it is one rounded product), except
// for the start state whose v is the float32 product (dtype flow).
#pragma once
#include "lmc_rng.hpp"
#include "lmc_targets.hpp"
#include "lmc_team.hpp"

namespace lmc {

// ---- status bits (per chain) ------------------------------------------------------------------
constexpr int kStatusBadInitialEnergy = 1;   // base_hmc.py:145-148 -> ValueError on the host

// ---- per-draw statistic slots -------------------------------------------------------------------
enum StatF64 : int {
    kSfStepSize = 0,      // exp(log_step) AFTER the update (nuts.py/hmc.py "step_size")
    kSfStepSizeBar = 1,
    kSfAccept = 2,        // mean_tree_accept (NUTS) / accept (HMC)
    kSfEnergyError = 3,
    kSfEnergy = 4,
    kSfMaxEnergyError = 5,  // NUTS max_energy_error / HMC path_length
    kSfModelLogp = 6,
    kNumStatF64 = 7
};
enum StatI32 : int { kSiDepth = 0 /* HMC: n_steps */, kSiTreeSize = 1 /* leapfrogs */, kNumStatI32 = 2 };
enum StatU8 : int { kSbDiverging = 0, kSbTune = 1, kSbAccepted = 2, kNumStatU8 = 3 };
enum Counter : int { kCtMaxTreedepth = 0, kCtDivsSample = 1, kCtSamplesAfterTune = 2, kCtLeapfrogs = 3, kCtWaveTicks = 4, kNumCounters = 5 };

// The sampler statistics of ONE draw (nuts.py:87-101 / hmc.py:36-50: eleven named values) as one 64-byte record, written
// by eight lanes in ONE coalesced store. (Until round 4 lane 0 issued twelve scattered 1-8-byte stores into per-statistic
// planes: 14 memory instructions and ~95 scalar address instructions per draw, each store dirtying its own cache line.)
struct StatRecord {
    double f64[kNumStatF64];   // kSf* order
    int tree_size;             // NUTS tree_size / HMC n_steps (= leapfrogs of the draw)
    unsigned depth_flags;      // bits 0-15 NUTS depth (HMC: n_steps is tree_size), bit 16 diverging, 17 tune, 18 accepted
};
static_assert(sizeof(StatRecord) == 64, "one cache-line-sized record per draw");
constexpr unsigned kRecDiverging = 1u << 16, kRecTune = 1u << 17, kRecAccepted = 1u << 18;

struct ChainArrays {
    int chains, d, dpad;
    // persistent state
    double* q;            // [C][dpad]
    float* var;           // [C][dpad]
    float* inv_std;       // [C][dpad]
    double* var64;        // [C][dpad] wide kernels (lmc_wide.hpp) only, else nullptr: the diagonal as float64 -- float32-valued
    double* inv_std64;    // [C][dpad]   unless the potential's dtype is float64
    double* wmean;        // [2][C][dpad]   Welford means (slot wsel = foreground)
    double* wraw;         // [2][C][dpad]
    double* wsum;         // [C][2]
    int* wsel;            // [C]
    int* n_samples;       // [C]
    int* awindow;         // [C] current adaptation window (grows by P.window_multiplier at every switch)
    double* da;           // [C][4] log_step, log_bar, hbar, mu
    const double* da_sqrt; // [da_table_len] sqrt(count)        (host libm)
    const double* da_mk;   // [da_table_len] count ** -k        (host libm)
    int da_table_len;
    int* da_count;        // [C]
    int* iter_count;      // [C]
    uint32_t* mt;         // [C][624]
    int* rng_pos;         // [C]
    int* rng_has_gauss;   // [C]
    double* rng_gauss;    // [C]
    int* status;          // [C]
    long long* counters;  // [C][kNumCounters]
    const int* stop;      // [1] pinned host word, != 0: stop requested (lmc_engine_request_stop); read by a few relay chains only
    int* stop_dev;        // [1] device word the relay chains copy it to: what every chain looks at, once per iteration
    const double* step_override;   // [C] step sizes chosen by the host for the next iteration (P.step_jitter == 2), else nullptr
    int* progress;        // [1] pinned host word: the iteration index a relay chain last started (lmc_engine_progress: a hint the
                          //     host reads without touching a stream)
    int* tree_hint;       // [1] pinned host word (or nullptr): (iteration << 12) | a relay chain's mean tree size in its launch so far (LDS plan choice)
    const uint32_t* seed; // [C] the seeds of lmc_engine_seed (key of the counter-based momentum stream, LMC_RNG_PHILOX)
    double* mom_mean;     // [C][dpad] running mean of the post-warm-up draws (nullptr = not kept)
    double* mom_m2;       // [C][dpad] running sum of squared deviations (Welford)
    int* mom_n;           // [C] number of draws accumulated
    double* scratch;      // [C][scratch_stride]
    long long scratch_stride;
    // outputs (row = iteration index relative to the engine's reserved capacity)
    double* trace;        // [C][cap - trace_begin][d] or nullptr
    long long trace_begin; // first iteration whose draw is stored
    StatRecord* stat_rec; // [C][cap]: one 64-byte record of sampler statistics per draw
    long long cap;
};

struct SamplerParams {
    int kind;             // 0 NUTS, 1 HMC
    int momentum_f32;     // 1: QuadPotentialDiagAdapt (float32 draw), 0: QuadPotentialDiag (float64 draw)
    int adapt_mass;       // 1: Welford updates during tuning (DiagAdapt)
    int adapt_step_size;
    double target_accept, emax, gamma, k, t0;
    int max_treedepth, early_max_treedepth;
    double path_length;
    int max_steps;
    int window;           // adaptation_window (101): initial value, the current one is per chain (A.awindow)
    double window_multiplier;   // adaptation_window_multiplier (quadpotential.py:243)
    long long n_tune;     // iterations with index < n_tune are tuning iterations
    long long iter_begin; // global index of the first iteration of this launch
    int n_iters;
    int nlds;             // subtree levels kept in LDS (>= 1)
    int lds_doubles;      // LDS doubles used by the subtree stack; the MT19937 state (624 words) follows
    int sdot_mode;        // SdotMode for the float32 start-state kinetic energy
    int chain_begin;      // run_kernel: first chain of this launch (the engine launches its chains as sub-blocks)
    int rng_mode;         // LMC_RNG_*: which run_kernel instantiation the host launches (informational on the device)
    int step_jitter;      // step_rand (base_hmc.py:154-155) in its one device form: step * uniform(jitter_lo, jitter_hi)
    double jitter_lo, jitter_hi;
    int relay_mask;       // stop word: one chain in (relay_mask + 1) of a launch reads the host's word (stop_request_load)
    int mass_f64;         // QuadPotentialDiagAdapt(dtype="float64"): the adapted diagonal is NOT rounded to float32 (wide kernels)
};

// ---- kernel arguments, re-read where they are used ---------------------------------------------------------------
// ChainArrays + SamplerParams are ~100 SGPRs of loop-invariant values. Taken by value they are all loaded at kernel entry
// and stay live across the tree build; the register allocator then parks them in lanes of VGPRs and every use outside the
// tree costs a v_readlane -- a VALU issue slot (~1.8 of them, tools/ubench/valu_cost.hip): ~200 per iteration of the
// sampling kernel, 12 % of a depth-3 iteration. The sampling kernel therefore reads them from the kernarg segment itself
// (constant address space: s_load through the scalar cache, no VALU slot) through a pointer that is made opaque once per
// REGION of the iteration body, so the loads are issued where the values are used -- adjacent fields in one
// s_load_dwordx4/x8 -- and nothing but the two segment pointers lives across the transition.
typedef const __attribute__((address_space(4))) ChainArrays KChainArrays;
typedef const __attribute__((address_space(4))) SamplerParams KSamplerParams;
struct KernArgs {   // run_kernel(ChainArrays, SamplerParams, const double*): the by-value structs as they lie in the kernarg segment
    KChainArrays* a;
    KSamplerParams* p;
    __device__ __forceinline__ static KernArgs get() {
        typedef const __attribute__((address_space(4))) char kchar;
        static_assert(alignof(ChainArrays) == 8 && alignof(SamplerParams) == 8 && sizeof(ChainArrays) % 8 == 0, "kernarg layout");
        kchar* base = (kchar*)__builtin_amdgcn_kernarg_segment_ptr();
        KernArgs k;
        k.a = (KChainArrays*)base;
        k.p = (KSamplerParams*)(base + sizeof(ChainArrays));
        return k;
    }
    // a fresh opaque copy of the pointer: loads through the returned reference are issued after this point
    __device__ __forceinline__ KChainArrays& A() const { KChainArrays* q = a; asm volatile("" : "+s"(q)); return *q; }
    __device__ __forceinline__ KSamplerParams& P() const { KSamplerParams* q = p; asm volatile("" : "+s"(q)); return *q; }
};

// ---- vector <-> memory (blocked layout, 8*NS contiguous bytes per thread of the team) -----------------
template <int NS>
__device__ __forceinline__ void vload(const double* base, double (&x)[NS]) {
    const double* p = base + LMC_CHAIN_THREAD * NS;
#pragma unroll
    for (int s = 0; s < NS; ++s) x[s] = p[s];
}
template <int NS>
__device__ __forceinline__ void vstore(double* base, const double (&x)[NS]) {
    double* p = base + LMC_CHAIN_THREAD * NS;
#pragma unroll
    for (int s = 0; s < NS; ++s) p[s] = x[s];
}
template <int NS>
__device__ __forceinline__ void vcopy(double (&dst)[NS], const double (&src)[NS]) {
#pragma unroll
    for (int s = 0; s < NS; ++s) dst[s] = src[s];
}

// partial (per-lane) dot of a with var (.) b, i.e. numpy's a.dot(velocity(b))
template <int NS>
__device__ __forceinline__ double pdot_v(const double (&a)[NS], const double (&var)[NS], const double (&b)[NS]) {
    double acc = 0.0;
#pragma unroll
    for (int s = 0; s < NS; ++s) acc = __builtin_fma(a[s], var[s] * b[s], acc);
    return acc;
}
// per-lane partial of a . v for an already formed velocity v
template <int NS>
__device__ __forceinline__ double pdot(const double (&a)[NS], const double (&v)[NS]) {
    double acc = 0.0;
#pragma unroll
    for (int s = 0; s < NS; ++s) acc = __builtin_fma(a[s], v[s], acc);
    return acc;
}
// velocity of a trajectory end: var (.) p, or its float32 product while the end still is the float32 start state
template <int NS>
__device__ __forceinline__ void end_velocity(double (&v)[NS], const double (&var)[NS], const double (&p)[NS], bool f32_start) {
    if (f32_start) {   // wave-uniform branch
#pragma unroll
        for (int s = 0; s < NS; ++s) v[s] = static_cast<double>(static_cast<float>(var[s]) * static_cast<float>(p[s]));
    } else {
#pragma unroll
        for (int s = 0; s < NS; ++s) v[s] = var[s] * p[s];
    }
}

// ---- float32 start-state kinetic energy: 0.5f * sdot(p, v) --------------------------------------------
// The reference computes the start state's kinetic energy with numpy's float32 dot, i.e. OpenBLAS
// cblas_sdot (integration.py:63-64 -> quadpotential.py:210-214). Its value feeds every energy error of
// the iteration, the accept statistic and through dual averaging the next step size, so a chain only
// tracks the reference beyond ~1e-6 if this one number is rounded the same way. The two modes below
// restate the summation order of OpenBLAS 0.3.29's sdot_k_SKYLAKEX / sdot_k_HASWELL (unit stride; read
// off the shipped binary numpy 2.2.6 links against -- see DESIGN.md "float32 start energy"):
//   n1 = n & ~31 elements in SIMD accumulators with fused multiply-add, consolidated in a fixed order,
//   the n - n1 tail as float32 products accumulated sequentially in DOUBLE, result = f32(tail + f64(simd)).
// kSdotNative sums the exact float32 products in float64 (wave butterfly) and rounds once.
enum SdotMode : int { kSdotNative = 0, kSdotOpenblasSkylakeX = 1, kSdotOpenblasHaswell = 2 };

// x, y: float32 vectors staged in LDS (n elements each). Wave-uniform result.
__device__ inline float sdot_openblas(const float* x, const float* y, int n, int mode) {
    const int lane = lane_id();
    const int n1 = n & ~31;
    float simd = 0.0f;
    if (n1) {
        if (mode == kSdotOpenblasSkylakeX) {
            const int n64 = n1 & ~63;
            float acc = 0.0f;                                  // zmm k = lane/16, element lane%16
            for (int b = 0; b < n64; b += 64) acc = __builtin_fmaf(x[b + lane], y[b + lane], acc);
            // The cross-lane moves are VALU (DPP row shifts, gfx950 permlane swaps, v_readlane): the ds_bpermute form
            // cost a dozen dependent LDS round trips per iteration (~2 k cycles of a depth-3 iteration's 30 k).
            float a = acc + dpp_f32<0x108>(acc);               // row_shl:8: fold 512 -> 256, valid on lanes 16k+m, m<8
            if (n1 > n64) {                                    // one trailing 32-block on ymm accumulators
                const int k = lane >> 4, m = lane & 15;
                if (m < 8) a = __builtin_fmaf(x[n64 + 8 * k + m], y[n64 + 8 * k + m], a);
            }
            // rows of a: [A0, A1, A2, A3]; bring A1, A2, A3 to row 0: swap16(a, a) -> [A0,A0,A2,A2] / [A1,A1,A3,A3],
            // swap32 of the second with itself -> [A1,A1,A1,A1] / [A3,A3,A3,A3]; swap32(a, a) -> [A0,A1,A0,A1] / [A2,A3,A2,A3]
            // (written as asm: with the builtins hipcc 7.2 picked the first result of each swap where the second was
            // asked for -- tools/ubench/swap_probe.hip shows the hardware semantics used here; "s_nop 1" covers the
            // VALU-write -> permlane-read hazard)
            const unsigned au = __builtin_bit_cast(unsigned, a);
            unsigned x16 = au, y16 = au, x32 = au, y32 = au;
            asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x16), "+v"(y16));   // y16 rows: [A1, A1, A3, A3]
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x32), "+v"(y32));   // y32 rows: [A2, A3, A2, A3]
            unsigned x3 = y16, y3 = y16;
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x3), "+v"(y3));     // y3 rows: [A3, A3, A3, A3]
            const float a1 = __builtin_bit_cast(float, y16);   // row 0 = A1
            const float a2 = __builtin_bit_cast(float, y32);   // row 0 = A2
            const float a3 = __builtin_bit_cast(float, y3);    // row 0 = A3
            float sv = a + a1;                                 // ((A0 + A1) + A2) + A3 on lanes m<8
            sv = sv + a2;
            sv = sv + a3;
            const float h = sv + dpp_f32<0x104>(sv);           // row_shl:4: low half + high half, lanes m<4
            const float h0 = readlane_f32(h, 0), h1 = readlane_f32(h, 1), h2 = readlane_f32(h, 2), h3 = readlane_f32(h, 3);
            simd = (h0 + h1) + (h2 + h3);                      // two vhaddps
        } else {
            float acc = 0.0f;                                  // ymm k = lane/8 (lanes < 32)
            if (lane < 32)
                for (int b = 0; b < n1; b += 32) acc = __builtin_fmaf(x[b + lane], y[b + lane], acc);
            const float hk = acc + __shfl_down(acc, 4, 64);    // H_k[m] on lanes 8k+m, m<4
            const int m = lane & 3;
            const float s01 = __shfl(hk, m, 64) + __shfl(hk, 8 + m, 64);
            const float s23 = __shfl(hk, 16 + m, 64) + __shfl(hk, 24 + m, 64);
            const float sv = s01 + s23;
            const float h0 = __shfl(sv, 0, 64), h1 = __shfl(sv, 1, 64), h2 = __shfl(sv, 2, 64), h3 = __shfl(sv, 3, 64);
            simd = (h0 + h1) + (h2 + h3);
        }
    }
    // tail: float32 products, sequential double accumulation from 0, then + simd
    const int nt = n - n1;                                     // < 32
    double prod = 0.0;
    if (lane < nt) prod = static_cast<double>(x[n1 + lane] * y[n1 + lane]);
    double tail = 0.0;
    for (int t = 0; t < nt; ++t) tail = tail + readlane_f64(prod, t);
    const double tot = tail + static_cast<double>(simd);
    return __builtin_bit_cast(float, first_u32(__builtin_bit_cast(uint32_t, static_cast<float>(tot))));
}

// 0.5f * p.dot(v) for the float32 start state; p, v float32-valued. scratch: >= 2*dpad floats of LDS.
// With W > 1 every wave evaluates the same staged vectors redundantly (same value everywhere).
template <int NS, class TeamT>
__device__ inline float start_kinetic_f32(TeamT& tm, const double (&p0)[NS], const float (&var)[NS], int d, int mode,
                                          float* scratch, int dpad) {
    if (mode == kSdotNative) {
        double part = 0.0;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const float pf = static_cast<float>(p0[s]);
            part = __builtin_fma(static_cast<double>(pf), static_cast<double>(var[s] * pf), part);
        }
        return 0.5f * static_cast<float>(tm.sum(part));
    }
    const int t = tm.tid();
    float* x = scratch;
    float* y = scratch + dpad;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const float pf = static_cast<float>(p0[s]);
        x[t * NS + s] = pf;
        y[t * NS + s] = var[s] * pf;
    }
    tm.sync();
    const float dot = sdot_openblas(x, y, d, mode);
    tm.sync();
    return 0.5f * dot;
}

// ---- team-wide RNG access ---------------------------------------------------------------------------------
// The MT19937 state is shared by the team (LDS); every wave keeps identical copies of the scalar stream state.
// A twist is done by wave 0 between two barriers; normal(size=d) is produced by wave 0 and its stream state
// re-broadcast through LDS.
template <class TeamT>
__device__ inline double team_uniform(TeamT& tm, RngState& r, UniformWindow& w) {
    if constexpr (TeamT::kWaves == 1) {
        return window_next(r, w);
    } else {
        if (w.idx == w.n && r.pos >= kMtN) {
            tm.sync();                       // every wave has finished reading the old generation
            if (tm.wave() == 0) mt_regen(r);
            tm.sync();
            r.pos = 0;
        }
        return window_next(r, w);
    }
}
// step_rand (base_hmc.py:46,123,154-155) for  lambda s: s * np.random.uniform(lo, hi): ONE double of the chain's own
// stream, drawn where the reference calls it -- after the momentum draw and the start state, before the trajectory
// (np.random.uniform(lo, hi) = lo + (hi - lo) * random_sample())
// step_jitter == 2: an arbitrary Python step_rand callable, evaluated by the HOST for every chain before the (one-iteration)
// launch; the kernel takes the value it left in A.step_override (lmc_engine_set_step_sizes).
template <class TeamT, class CA, class PT>
__device__ __forceinline__ double jitter_step_size(TeamT& tm, RngState& rng, const CA& A, const PT& P, int c, double step_size) {
    if (!P.step_jitter) return step_size;
    if (P.step_jitter == 2) return first_f64(A.step_override[c]);
    UniformWindow jw;
    window_reset(jw);
    const double u = team_uniform(tm, rng, jw);
    return first_f64(step_size * (P.jitter_lo + (P.jitter_hi - P.jitter_lo) * u));
}
// normal(size=d) of the chain's numpy-legacy stream by ALL four waves of a team (d > 512). Drawn by wave 0 alone (round 2:
// ten rounds of 64 polar attempts, eight passes of log / divide / sqrt and four twists per iteration at d = 1000, every
// instruction at a lone wave's issue latency while three waves wait at a barrier) it is 36 % of C4's iteration; with the
// draw off the critical path altogether (LMC_RNG_PHILOX) C4 runs 3.05e8 instead of 2.1e8. Here
//   * a round evaluates one attempt per thread of waves 0..2 in stream order -- a generation of 624 words holds 156;
//   * in the same round the team twists the generation OUT OF PLACE into a second LDS buffer, one thread per strand
//     (mt_twist_strand: three dependent word steps, no synchronisation), so the next generation is ready when the round's
//     barrier falls (its last word, which needs two other threads' results, right behind the barrier); the (at most two)
//     words of the old generation that no whole attempt used are carried in registers, the buffers swap, and the old
//     buffer is free to receive the generation after;
//   * the waves' acceptance masks travel through the team's exchange area (the round's one barrier) and every wave then
//     knows every wave's mask: ranks, pairs taken and, in the last round, the attempt that ends the call are scalar
//     arithmetic on the masks, identical in every wave -- the stream position is never broadcast;
//   * the accepted pairs are compacted into `stage` in stream order and the expensive part runs over 256 pairs at a time.
// Consumption order, accepted set and arithmetic are rng_normals' (lmc_rng.hpp), hence numpy's. On return r.mt points to
// the buffer that holds the current generation (either of the two).
// History (C4, tools/phase_timing.py: ticks of the draw per iteration / leapfrogs per second): wave 0 alone 27.9 k /
// 2.06e8; attempts on all waves but twist and generation crossing still wave 0's: no gain; the fourth wave twisting out
// of place (three dependent batches) beside the attempts: 22.0 k / 2.20e8; the twist by strands on all waves: 20.4 k /
// 2.27e8. What is left is one team barrier per generation (five per draw at d = 1000) plus three around the second phase.
template <class TeamT>
__device__ inline void team_normals_parallel(TeamT& tm, RngState& r, int d, double* out, double* stage, uint32_t* buf_a,
                                             uint32_t* buf_b) {
    constexpr int W = TeamT::kWaves;
    static_assert(W >= 4, "three attempt waves cover a generation's 156 attempts, 227 threads twist");
    constexpr int AW = 3;   // waves that evaluate attempts (a generation of 624 words holds 156: three waves cover it)
    const int tid = tm.tid(), lane = lane_id(), wave = tm.wave();
    int produced = 0;
    tm.sync();   // earlier readers of out / stage / the stream state are done
    if (r.has_gauss && d > 0) {
        if (tid == 0) out[0] = r.gauss;
        r.has_gauss = 0;
        r.gauss = 0.0;
        produced = 1;
    }
    const int need_pairs = first_i32((d - produced + 1) >> 1);
    int have = 0;
    int pos = first_i32(r.pos);
    uint32_t* cur = r.mt;
    uint32_t carry0 = 0u, carry1 = 0u;
    int ncarry = 0;          // words of the previous generation ahead of cur[pos] (0 or 2)
    bool next_ready = false; // the other buffer holds genrand(cur)
    uint32_t gen623 = 0u;    // word 623 of cur once cur is a generation twisted in this call (thread 0 stores it behind a
    bool have623 = false;    // barrier; the waves that need it before the next barrier use this copy)
    uint32_t next623 = 0u;   // word 623 of the generation in the other buffer
    while (have < need_pairs) {
        uint32_t* alt = (cur == buf_a) ? buf_b : buf_a;
        const bool twisting = !next_ready;
        if (twisting && tid < 227) mt_twist_strand(cur, alt, tid);
        // attempts of this round: whole attempts in carry ++ cur[pos, 624), one per thread of waves 0 .. AW-1
        const int navail = (ncarry + kMtN - pos) >> 2;
        const int n_att = navail < 64 * AW ? navail : 64 * AW;
        double x1 = 0.0, x2 = 0.0;
        unsigned long long mask = 0ull;
        if (wave < AW && n_att > 0) {
            const int at = tid < n_att ? tid : n_att - 1;   // a thread beyond n_att re-reads the last whole attempt and is masked out
            const int base = pos + 4 * at - ncarry;          // -2 for the attempt that starts in the carried words
            const bool in_carry = base < pos;
            const uint32_t w0 = in_carry ? carry0 : cur[base < 0 ? 0 : base];
            const uint32_t w1 = in_carry ? carry1 : cur[base < 0 ? 0 : base + 1];
            x1 = 2.0 * mt_words_to_double(w0, w1) - 1.0;
            x2 = 2.0 * mt_words_to_double(cur[base + 2], cur[base + 3]) - 1.0;
            const double r2 = x1 * x1 + x2 * x2;
            const int n_here = n_att - 64 * wave;
            const unsigned long long lanes = n_here >= 64 ? ~0ull : (n_here <= 0 ? 0ull : ((1ull << n_here) - 1ull));
            mask = ballot64(r2 > 0.0) & ballot64(r2 < 1.0) & lanes;
        }
        // the words no whole attempt of this generation can use (read before the barrier: the buffer is twisted over after it)
        const uint32_t tail0 = first_u32(cur[kMtN - 2]), tail1 = have623 ? gen623 : first_u32(cur[kMtN - 1]);
        double v[2 * AW];
#pragma unroll
        for (int k = 0; k < AW; ++k) {   // exact: a 32-bit integer in a double, the other waves' slots contribute 0
            v[2 * k] = (k == wave) ? static_cast<double>(static_cast<uint32_t>(mask)) : 0.0;
            v[2 * k + 1] = (k == wave) ? static_cast<double>(static_cast<uint32_t>(mask >> 32)) : 0.0;
        }
        tm.template exchange<2 * AW>(v);   // the round's one barrier: all reads of cur and the strands of the twist are behind it
        if (twisting) {   // the generation's last word: every wave forms it for itself, thread 0 stores it
            next623 = first_u32(mt_twist(tail1, alt[0], alt[396]));
            if (tid == 0) alt[kMtN - 1] = next623;
            next_ready = true;
        }
        int before = 0, total = 0;
        unsigned long long m[AW];
#pragma unroll
        for (int k = 0; k < AW; ++k) {
            m[k] = static_cast<unsigned long long>(static_cast<uint32_t>(v[2 * k])) |
                   (static_cast<unsigned long long>(static_cast<uint32_t>(v[2 * k + 1])) << 32);
            const int cnt = __popcll(m[k]);
            if (k < wave) before += cnt;
            total += cnt;
        }
        const int want = need_pairs - have;
        int consumed = n_att, taken = total;
        if (total >= want) {   // the want-th accepted attempt of the round ends the call
            int run = 0;
#pragma unroll
            for (int k = 0; k < AW; ++k) {
                const int cnt = __popcll(m[k]);
                if (run < want && run + cnt >= want) {
                    unsigned long long mm = m[k];
                    for (int i = want - run - 1; i > 0; --i) mm &= mm - 1ull;   // drop the accepted attempts before it
                    consumed = 64 * k + __ffsll(static_cast<long long>(mm));
                }
                run += cnt;
            }
            taken = want;
        }
        if (wave < AW) {
            const int rank = before + static_cast<int>(__builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(mask >> 32),
                                                       __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(mask), 0u)));
            const bool acc = ((mask >> lane) & 1ull) != 0ull;
            if (acc && rank < want) {
                stage[2 * (have + rank)] = x1;
                stage[2 * (have + rank) + 1] = x2;
            }
        }
        have += taken;
        if (consumed > 0) { pos += 4 * consumed - ncarry; ncarry = 0; }
        if (have < need_pairs && kMtN - pos < 4 && ncarry == 0) {   // generation used up: carry its last words, move to the next
            ncarry = kMtN - pos;                      // 0 or 2 (the stream only ever sits at even positions here)
            carry0 = tail0; carry1 = tail1;
            cur = alt;
            pos = 0;
            next_ready = false;
            gen623 = next623; have623 = true;
        }
    }
    r.mt = cur;
    r.pos = pos;
    tm.sync();
    double tail[1] = {0.0};
    const bool odd_tail = produced + 2 * need_pairs > d;
    for (int base = 0; base < need_pairs; base += 64 * W) {
        const int pi = base + tid;
        if (pi < need_pairs) {
            const double px1 = stage[2 * pi], px2 = stage[2 * pi + 1];
            const double r2 = px1 * px1 + px2 * px2;
            const double f = sqrt(-2.0 * log_unit(r2) / r2);
            const int idx = produced + 2 * pi;
            out[idx] = f * px2;
            const double g1 = f * px1;
            if (idx + 1 < d) out[idx + 1] = g1;
            if (odd_tail && pi == need_pairs - 1) tail[0] = g1;   // the last second variate stays in the cache (numpy legacy_gauss)
        }
    }
    if (odd_tail) {   // one thread holds it: its wave's value, the others contribute 0
        tail[0] = readlane_f64(tail[0], (need_pairs - 1) & 63);
        if (wave != (((need_pairs - 1) >> 6) % W)) tail[0] = 0.0;
        tm.template exchange<1>(tail);
        r.gauss = first_f64(tail[0]);
        r.has_gauss = 1;
    }
    tm.sync();
}
template <class TeamT>
__device__ inline void team_normals(TeamT& tm, RngState& r, int d, double* out, double* stage, double* bcast,
                                    uint32_t* mt_a, uint32_t* mt_b) {
    if constexpr (TeamT::kWaves == 1) {
        rng_normals(r, d, out, stage);
    } else if (TeamT::kWaves >= 4 && (first_i32(r.pos) & 1) == 0) {   // (an odd position -- a state handed over from the host after a
        if constexpr (TeamT::kWaves >= 4) team_normals_parallel(tm, r, d, out, stage, mt_a, mt_b);   // 32-bit draw -- takes wave 0's word-granular path)
    } else {
        tm.sync();
        if (tm.wave() == 0) {
            rng_normals(r, d, out, stage);
            if (lane_id() == 0) {
                bcast[0] = static_cast<double>(r.pos);
                bcast[1] = static_cast<double>(r.has_gauss);
                bcast[2] = r.gauss;
            }
        }
        tm.sync();
        r.pos = first_i32(static_cast<int>(bcast[0]));
        r.has_gauss = first_i32(static_cast<int>(bcast[1]));
        r.gauss = first_f64(bcast[2]);
    }
}

// ---- leapfrog (integration.py:100-121) -------------------------------------------------------------
// In/out: q, p, g. Out: energy, logp. Every elementwise operation is the same rounded IEEE
// operation as numpy's (TU built with -ffp-contract=off); only the two reductions differ in order.
// Targets whose log-density is a plain lane sum (kLanePartial) hand back the per-lane partial, so
// logp and the kinetic energy share ONE interleaved butterfly instead of two dependent ones.
template <int NS, class Target, class TeamT>
__device__ __forceinline__ void leapfrog(TeamT& tm, const Target& tgt, const double (&var)[NS], double eps,
                                         double (&q)[NS], double (&p)[NS], double (&g)[NS],
                                         double& energy, double& logp) {
    const double dt = 0.5 * eps;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        p[s] = p[s] + dt * g[s];
        const double v = var[s] * p[s];
        q[s] = q[s] + eps * v;
    }
    if constexpr (Target::kLanePartial) {
        double lp = tgt.logp_grad_partial(tm, q, g);
        double kin = 0.0;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            p[s] = p[s] + dt * g[s];
            kin = __builtin_fma(p[s], var[s] * p[s], kin);
        }
        tm.sum2(lp, kin);
        logp = lp;
        energy = 0.5 * kin - logp;   // wave-uniform, VGPR resident
    } else {
        logp = first_f64(tgt.logp_grad(tm, q, g));
        double kin = 0.0;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            p[s] = p[s] + dt * g[s];
            kin = __builtin_fma(p[s], var[s] * p[s], kin);
        }
        energy = 0.5 * tm.sum(kin) - logp;
    }
}

// ---- vector slots in LDS / the scratch row --------------------------------------------------------------
// Stack traffic with explicit address spaces: ds_read/ds_write_b128 for the LDS levels, global_load/store for
// the spilled ones (a generic pointer would make every access a flat_* instruction that checks the aperture
// and ties up both the LDS and the vector-memory counters).
typedef __attribute__((address_space(3))) double lds_double;
typedef __attribute__((address_space(1))) double glb_double;
// (Round 4, measured and dropped: the two halves of a four-element slice in two LDS planes -- lane stride 16 bytes, no 2-way
// bank conflict of the ds_read_b128 pairs -- C4 2.09e8 either way, C5 -2.7 %: the conflicts the counters show are not on this
// path's critical path, and the second address costs an SGPR.)
template <int NS, class PTR>
__device__ __forceinline__ void vload_as(PTR base, double (&x)[NS]) {
    PTR p = base + LMC_CHAIN_THREAD * NS;
#pragma unroll
    for (int s = 0; s < NS; ++s) x[s] = p[s];
}
template <int NS, class PTR>
__device__ __forceinline__ void vstore_as(PTR base, const double (&x)[NS]) {
    PTR p = base + LMC_CHAIN_THREAD * NS;
#pragma unroll
    for (int s = 0; s < NS; ++s) p[s] = x[s];
}
// Tree weights are kept in the LINEAR domain: w = exp(-dE - c) with one offset c per transition.
// The reference carries log-weights and pays logaddexp (exp + log1p) twice plus log(U) per merge
// (nuts.py:322-328, :399-404); in the linear domain a merge is two additions and the multinomial
// choice is  U * (w_a + w_b) < w_b  -- the same event as  log U < ls_b - logaddexp(ls_a, ls_b).
// One exp per LEAF replaces five transcendental evaluations per MERGE. c starts at 0 (start state
// weight 1) and is only moved (cold path) when a leaf's -dE exceeds it by 600, so nothing overflows
// for |dE| < Emax; results differ from the log-domain form by a few ulps of the weights.
struct LevelScalars {   // lane j holds level j
    double w, a, pe, plogp;
    __device__ __forceinline__ void put(int j, double w_, double a_, double c, double d) {
        if (lane_id() == j) { w = w_; a = a_; pe = c; plogp = d; }
    }
    __device__ __forceinline__ void get(int j, double& w_, double& a_, double& c, double& d) const {
        w_ = readlane_f64(w, j); a_ = readlane_f64(a, j); c = readlane_f64(pe, j); d = readlane_f64(plogp, j);
    }
};

struct TransitionOut {
    double accept;         // mean_tree_accept / HMC accept
    double energy, energy_error, max_energy_error, model_logp;
    int depth;             // NUTS depth / HMC n_steps
    int n_leapfrog;        // NUTS tree_size / HMC n_steps
    int diverging;
    int exhausted;         // NUTS: loop ran to max_treedepth without turning/diverging
    int accepted;          // HMC
};

// ---- HMC transition (hmc.py:140-182): ONE statement for every kernel family -------------------------------------------------
// What differs between the families -- how a state is integrated and where the accepted position goes -- is a policy:
//   typename P::End              the state being integrated (members q, p, g, v [NS] + what the integrator carries along)
//   double uniform()             next uniform of the chain's stream
//   void start_state(End&)       the start State (integration.py:52-66)
//   void leapfrog(eps, End&, energy&, logp&)      integration.py:100-121
//   void accept_state(const End&)                 the chain's position becomes this state's
// (FusedHmcPolicy below; DenseTreePolicy in lmc_dense.hpp; WideTreePolicy in lmc_wide.hpp.)
template <class P>
__device__ inline void hmc_transition_any(P& pol, double e0, double logp0, double step_size, double emax, double path_length,
                                          int max_steps, TransitionOut& out) {
    const double plen = first_f64(pol.uniform() * path_length);
    int n_steps = static_cast<int>(plen / step_size);
    n_steps = n_steps < 1 ? 1 : n_steps;
    n_steps = n_steps > max_steps ? max_steps : n_steps;
    typename P::End c;
    pol.start_state(c);
    double energy = e0, logp = logp0;
    for (int i = 0; i < n_steps; ++i) pol.leapfrog(step_size, c, energy, logp);
    bool diverging = !isfinite(energy);
    double de = first_f64(e0 - energy);
    if (isnan(de)) de = -__builtin_inf();
    if (fabs(de) > emax) diverging = true;
    const double accept = first_f64(fmin(1.0, exp_uniform(de)));
    bool accepted = false;
    if (!diverging) {
        const double u = pol.uniform();
        if (!(u >= accept)) { accepted = true; pol.accept_state(c); }
    }
    out.accept = accept;
    out.energy = energy;
    out.energy_error = de;
    out.max_energy_error = plen;   // slot shared with path_length
    out.model_logp = logp;
    out.depth = n_steps;
    out.n_leapfrog = n_steps;
    out.diverging = diverging;
    out.exhausted = 0;
    out.accepted = accepted;
}

// the fused diagonal-mass kernels: the state lives in registers, the velocity is recomputed inside leapfrog<>
template <int NS, class Target, class TeamT>
struct FusedHmcPolicy {
    static constexpr int kNS = NS;
    struct End { double q[NS], p[NS], g[NS]; };
    TeamT& tm; const Target& tgt; const double (&var)[NS]; RngState& rng;
    double (&q)[NS]; const double (&p0)[NS]; const double (&g0)[NS];
    UniformWindow win;
    __device__ __forceinline__ double uniform() { return team_uniform(tm, rng, win); }
    __device__ __forceinline__ void start_state(End& c) const { vcopy(c.q, q); vcopy(c.p, p0); vcopy(c.g, g0); }
    __device__ __forceinline__ void leapfrog(double eps, End& c, double& energy, double& logp) {
        lmc::leapfrog<NS>(tm, tgt, var, eps, c.q, c.p, c.g, energy, logp);
    }
    __device__ __forceinline__ void accept_state(const End& c) { vcopy(q, c.q); }
};
template <int NS, class Target, class TeamT>
__device__ inline void hmc_transition(TeamT& tm, const Target& tgt, const double (&var)[NS], RngState& rng,
                                      double (&q)[NS], const double (&p0)[NS], const double (&g0)[NS],
                                      double e0, double logp0, double step_size, double emax,
                                      double path_length, int max_steps, TransitionOut& out) {
    FusedHmcPolicy<NS, Target, TeamT> pol{tm, tgt, var, rng, q, p0, g0, UniformWindow{0.0, 0, 0}};
    hmc_transition_any(pol, e0, logp0, step_size, emax, path_length, max_steps, out);
}

// ---- NUTS transition (nuts.py:204-224, _Tree :251-435; iterative post-order), pair form ----------------
// Batched LDS-transposed reductions, leaf pairs. The straightforward statement of the same algorithm -- one leaf at a
// time, a DPP/permlane reduction per dot product group, level scalars in VGPR lanes -- is what lmc_dense.hpp and
// lmc_tick.hpp still use; here it was replaced because of where the issue slots go. Measured on gfx950
// (tools/ubench/valu_cost.hip): a DPP move costs as much VALU time as a float64 operation, a permlane swap ~1.9x, a
// v_readlane with an SGPR index ~1.8x, and at 2-3 waves per SIMD every instruction of the per-leaf path counts.
//   * leaves are processed in PAIRS (2k, 2k+1): the even leaf's {p, q} stays in registers, so subtree-stack
//     level 0 is never written to or read from LDS;
//   * every length-d reduction of a pair -- two kinetic energies, two log-densities, the two level-0 U-turn dots --
//     goes through ONE transposed reduction: each lane drops its partials into LDS columns, lane l adds the eight
//     partials [8l, 8l+8) and three DPP steps finish the sums (16 VALU instead of 22 + 22 + 20); every further
//     cascade level is one more such flush of its six dots (16 VALU instead of 42);
//   * the per-leaf scalars are evaluated on lanes: energies of both leaves with one DPP shift, and the four
//     weights (w_A, w_A min(1,e^-dE_A), w_B, ...) with ONE table-driven exp on four lanes instead of two to four
//     wave-uniform exps (each ~25 VALU + ~20 s_mov + a scalar-cache round trip);
//   * the second leaf of a pair is integrated before the first one's energy is known (speculation): a diverging
//     first leaf discards it; leapfrog count and random stream are those of the sequential algorithm;
//   * per-level subtree scalars live in LDS (lane 0 writes, broadcast reads) instead of VGPR lanes read with
//     v_readlane; the proposal position of a merged node is tracked by its SOURCE (this pair or a stack level)
//     and copied once when the node is parked; a level-1 node stores {lp, rp, q} only (its psum is lp + rp);
//   * cold per-transition vectors (other trajectory end, running momentum sum, ...) live in LDS slots and the
//     trajectory proposal in the chain's row of A.q, so nothing but the hot set occupies registers in the pair loop;
//   * the whole LDS plan is compile-time (dpad = 64 * NS): every LDS access is one lane-offset register plus an
//     immediate, no per-slot address registers.
#ifndef LMC_EXP_LANES_SLOAD
#define LMC_EXP_LANES_SLOAD 2
#endif
constexpr int kRedValues = 6;
constexpr int kExpTableDoubles = 32;    // 2^(j/32): the even entries of kExp2Table
constexpr int kLevelScalDoubles = 96;   // 4 per parked level, levels < 24 (max_treedepth <= 20)
#ifndef LMC_PAIR_COLD_LDS
#define LMC_PAIR_COLD_LDS 3   // cold slots kept in LDS: {aold, psum, op}; {oq, og} (touched only when the direction flips) head
#endif                        // the scratch row -- measured: with the MT19937 state in LDS instead, C3 is unchanged and depth-3 trees gain 9 %
enum ColdSlot : int { kColdAold = 0, kColdPsum = 1, kColdOp = 2, kColdOq = 3, kColdOg = 4, kNumCold = 5 };
constexpr int kNumColdSlots = kNumCold;

template <int NS, int W = 1, int PL = 0>
struct PairLds {   // offsets in doubles from the start of the block's dynamic LDS; W waves share one plan
    static constexpr int DP = 64 * NS * W;
    static constexpr int kRedWave = 64 * kRedValues;              // one reduction buffer per wave
    static constexpr int kRedSize = (2 * DP > W * kRedWave) ? 2 * DP : W * kRedWave;   // also normals / sdot staging
    static constexpr int kExp = kRedSize;
    static constexpr int kXsum = kExp + kExpTableDoubles;         // team combine area (W > 1): 2 buffers x 8 sums x W waves
    static constexpr int kXsumSize = W > 1 ? 2 * 8 * W : 0;
    static constexpr int kScal = kXsum + kXsumSize;               // per-level scalars, one copy per wave
    static constexpr int kCold = kScal + W * kLevelScalDoubles;
    // cold slots in LDS; the others head the scratch row. Eight waves per chain (NS = 2, d <= 1024): eight reduction buffers
    // and eight copies of the level scalars leave room for two (79 KB of the 80 KB a team may use at two teams per CU)
    // PL = 1 (one-wave kernels only): the DEEP-TREE plan -- the MT19937 state is used in place (L2) and only the first cold slot
    // (NS <= 2; none at NS = 4) stays in LDS, which makes room for stack level 2. Measured (profiles/r05_c3_level2_lds_ab.txt,
    // r05_level2_lds_ab_others.txt): C3 +7.5 %, C5 +2.4 %, but depth-3 trees -8 ... -11 % (the momentum draw reads the
    // generator through L2) -- so the engine picks the plan per launch from the tree sizes the chains report (run_kernel<.., PL>).
    static_assert(PL == 0 || W == 1, "the deep-tree plan exists for one-wave kernels");
    static constexpr int kColdLds = PL == 1 ? (NS >= 4 ? 0 : (LMC_PAIR_COLD_LDS < 1 ? LMC_PAIR_COLD_LDS : 1))
                                            : (W >= 8 ? (LMC_PAIR_COLD_LDS < 2 ? LMC_PAIR_COLD_LDS : 2) : LMC_PAIR_COLD_LDS);
    static constexpr int kL1 = kCold + kColdLds * DP;             // level 1: {lp, rp, q}
    static constexpr int kL2 = kL1 + 3 * DP;                      // levels 2..nlds: {lp, rp, psum, q}
    static constexpr int kMinDoubles = kL2;                       // head + cold + level 1: what the form needs at least
    static constexpr int kGlbLevels = (kNumCold - kColdLds) * DP; // scratch row: cold slots not in LDS, then levels > nlds
    __host__ __device__ static constexpr int total_doubles(int nlds) { return kL2 + (nlds > 1 ? (nlds - 1) * 4 * DP : 0); }
};
// (ns, w) -> plan sizes for the host (every shape run_kernel is instantiated for)
// LMC_EXPERIMENTAL_SHAPES (a variant build, tools/c4_shape_ab.sh / tools/c5_team_latency.sh): shapes that were measured and
// lost -- eight waves of two elements (C4: 1.48e8 against <4,4>'s 2.11e8, profiles/r05_c4_shape_ab.txt), and the d = 256
// teams <2,2> / <1,4> (a lone chain's leapfrog is slower on them than on <4,1>, profiles/r05_c5_team_latency.txt). Selected
// per engine with LMC_RUN_SHAPE="ns,w".
#ifdef LMC_EXPERIMENTAL_SHAPES
#define LMC_PAIR_SHAPES(X) X(1, 1) X(2, 1) X(4, 1) X(4, 2) X(4, 4) X(2, 8) X(2, 2) X(1, 4)
#else
#define LMC_PAIR_SHAPES(X) X(1, 1) X(2, 1) X(4, 1) X(4, 2) X(4, 4)
#endif
constexpr int pair_min_doubles(int ns, int w, int plan = 0) {
#define X(NSV, WV) if (ns == NSV && w == WV) return (plan == 1 && WV == 1) ? PairLds<NSV, 1, 1>::kMinDoubles : PairLds<NSV, WV, 0>::kMinDoubles;
    LMC_PAIR_SHAPES(X)
#undef X
    return 1 << 30;
}
constexpr int pair_total_doubles(int ns, int w, int nlds, int plan = 0) {
#define X(NSV, WV) if (ns == NSV && w == WV) return (plan == 1 && WV == 1) ? PairLds<NSV, 1, 1>::total_doubles(nlds) : PairLds<NSV, WV, 0>::total_doubles(nlds);
    LMC_PAIR_SHAPES(X)
#undef X
    return 1 << 30;
}

struct PairCtx {
    double* lds;        // block's dynamic LDS
    double* glb;        // this chain's scratch row
    int nlds;           // stack levels 1..nlds live in LDS (>= 1)
    int red_lane;       // LDS address (bytes) of this lane's 64-byte row of its wave's reduction buffer, rotation included
    int wave_red;       // this wave's reduction buffer (doubles; 0 for W = 1)
    int wave_scal;      // this wave's copy of the level scalars (doubles; 0 for W = 1)
    int wave;           // wave index in the team
    int xpar;           // which team combine buffer the next reduction uses
};

// ---- batched lane reductions through LDS ("transposed" reduction)
template <int NS, int W = 1, int PL = 0>
__device__ __forceinline__ void red_put(const PairCtx& cx, int v, double x) {
    if constexpr (W == 1) ((lds_double*)cx.lds)[v * 64 + lane_id()] = x;
    else ((lds_double*)cx.lds)[v * 64 + lane_id() + cx.wave_red] = x;
}
// lane 8k+7 of the result holds the sum of value k (k < kRedValues) over the whole team; other lanes hold partial scans
// (W = 1) or the same totals (W > 1).
// Team (W > 1): every wave reduces its own buffer, lanes 8k+7 drop the wave's six sums into a double-buffered combine
// area, ONE barrier, every wave adds the W partials in wave order (the same value in every wave). A wave can reach
// the next-but-one reduction (same buffer) only after every wave has passed the barrier in between.
template <int NS, int W = 1, int PL = 0>
__device__ __forceinline__ double red_gather(PairCtx& cx) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) d2 lds_d2;
    asm volatile("" ::: "memory");   // DS operations of one wave execute in issue order: no wait, only no reordering
    unsigned r0 = static_cast<unsigned>(cx.red_lane);   // LDS address of this lane's row (rotation included)
    asm volatile("" : "+v"(r0));     // keep ONE address register: the other three pieces are re-derived here, not kept live
    lds_double* L = (lds_double*)cx.lds;
    // the four 16-byte pieces of this lane's row in a rotated order (r0 already holds the rotation), so that a
    // ds_read_b128 of 16 lanes covers all 64 banks (lane stride 64 B = 16 banks); plain LDS addresses, so that a piece's
    // address is one v_xor away from r0 (an element index would cost a shift-add per piece on top; the row is 64-byte
    // aligned, so the xor stays inside it)
    const d2 a = *(const lds_d2*)(size_t)(r0), b = *(const lds_d2*)(size_t)(r0 ^ 16u);
    const d2 c = *(const lds_d2*)(size_t)(r0 ^ 32u), d = *(const lds_d2*)(size_t)(r0 ^ 48u);
    asm volatile("" ::: "memory");
    double s = ((a.x + a.y) + (b.x + b.y)) + ((c.x + c.y) + (d.x + d.y));
    s += dpp_f64<0x111>(s);
    s += dpp_f64<0x112>(s);
    s += dpp_f64<0x114>(s);
    if constexpr (W > 1) {
        const int lane = lane_id();
        lds_double* xs = L + (PairLds<NS, W, PL>::kXsum + cx.xpar * (8 * W)) + (lane >> 3) * W;
        if ((lane & 7) == 7) xs[cx.wave] = s;
        __syncthreads();
        double t = xs[0];
#pragma unroll
        for (int w = 1; w < W; ++w) t += xs[w];
        s = t;
        cx.xpar ^= 1;
    }
    return s;
}
__device__ __forceinline__ int red_lane_init() {
    const int lane = lane_id();
    const int row = lane < 8 * kRedValues ? lane : 0;   // rows beyond the buffer re-read row 0 (their sums are never used)
    return row * 8 + 2 * ((lane >> 2) & 3);
}
// any of the sums k in [k0, k0 + n) <= 0 ?
__device__ __forceinline__ bool red_any_nonpositive(double s, int k0, int n) {
    const unsigned long long m = ballot64(s <= 0.0);
    unsigned long long sel = 0ull;
    for (int k = k0; k < k0 + n; ++k) sel |= 1ull << (8 * k + 7);
    return (m & sel) != 0ull;
}

// exp on lanes (arguments <= ~700; very negative ones underflow to 0): exp_uniform_fast's algorithm with a 32-entry
// 2^(j/32) table read from LDS by every lane: x = (32 e + j) ln2/32 + r, |r| <= ln2/64, degree-6 polynomial
// (truncation r^7/5040 < 4e-18 relative). |error| < ~1 ulp.
// The reduction and polynomial constants come from constant memory in ONE scalar load (s_load_dwordx16, scalar cache):
// as s_mov literals they cost two SALU issue slots each plus a wait state, 22 slots per call on a wave that can issue
// one instruction of any kind at a time.
__constant__ double kExpLanesConst[8] = {46.166241308446828,            // 32 / ln 2
                                         2.16608493865351192654e-02,    // ln2/32 hi
                                         5.96317165397058656256e-12,    // ln2/32 lo
                                         1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0, 0.0};
struct ExpLanesConst { double inv, hi, lo, c6, c5, c4, c3; };
__device__ __forceinline__ ExpLanesConst exp_lanes_const() {
    ExpLanesConst c;
#if LMC_EXP_LANES_SLOAD == 2
    // re-loaded where it is used (the address is made opaque, so the load is not hoisted and the seven values do not
    // occupy fourteen SGPRs for the whole kernel)
    typedef const __attribute__((address_space(4))) double cst_double;
    cst_double* p = (cst_double*)kExpLanesConst;
    asm volatile("" : "+s"(p));
    c.inv = p[0]; c.hi = p[1]; c.lo = p[2]; c.c6 = p[3]; c.c5 = p[4]; c.c4 = p[5]; c.c3 = p[6];
#elif LMC_EXP_LANES_SLOAD == 1
    c.inv = kExpLanesConst[0]; c.hi = kExpLanesConst[1]; c.lo = kExpLanesConst[2];
    c.c6 = kExpLanesConst[3]; c.c5 = kExpLanesConst[4]; c.c4 = kExpLanesConst[5]; c.c3 = kExpLanesConst[6];
#else
    c.inv = LMC_SC(46.166241308446828); c.hi = LMC_SC(2.16608493865351192654e-02); c.lo = LMC_SC(5.96317165397058656256e-12);
    c.c6 = LMC_SC(1.0 / 720.0); c.c5 = LMC_SC(1.0 / 120.0); c.c4 = LMC_SC(1.0 / 24.0); c.c3 = LMC_SC(1.0 / 6.0);
#endif
    return c;
}
template <int NS, int W = 1, int PL = 0>
__device__ __forceinline__ double exp_lanes(const PairCtx& cx, const ExpLanesConst& c, double x) {
    const double kf = rint(x * c.inv);
    double r = __builtin_fma(-kf, c.hi, x);
    r = __builtin_fma(-kf, c.lo, r);
    const int ki = static_cast<int>(kf);
    const double t = ((const lds_double*)cx.lds)[PairLds<NS, W, PL>::kExp + (ki & 31)];
    double p = fma_sgpr_addend(r, c.c6, c.c5);
    p = fma_sgpr_addend(p, r, c.c4);
    p = fma_sgpr_addend(p, r, c.c3);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return ldexp(p * t, ki >> 5);
}

// ---- cold slots and subtree-stack levels: LDS offsets are immediates, the scratch row takes what does not fit
template <int NS, int W, int PL, int SLOT>
__device__ __forceinline__ void cold_load(const PairCtx& cx, double (&x)[NS]) {
    using L = PairLds<NS, W, PL>;
    if constexpr (SLOT < L::kColdLds) vload_as<NS>((lds_double*)cx.lds + (L::kCold + SLOT * L::DP), x);
    else vload_as<NS>((glb_double*)cx.glb + (SLOT - L::kColdLds) * L::DP, x);
}
template <int NS, int W, int PL, int SLOT>
__device__ __forceinline__ void cold_store(const PairCtx& cx, const double (&x)[NS]) {
    using L = PairLds<NS, W, PL>;
    if constexpr (SLOT < L::kColdLds) vstore_as<NS>((lds_double*)cx.lds + (L::kCold + SLOT * L::DP), x);
    else vstore_as<NS>((glb_double*)cx.glb + (SLOT - L::kColdLds) * L::DP, x);
}
// level 1 (always LDS): {lp, rp, q}
template <int NS, int W = 1, int PL = 0>
__device__ __forceinline__ void level1_load(const PairCtx& cx, double (&lp)[NS], double (&rp)[NS], double (&ps)[NS]) {
    using L = PairLds<NS, W, PL>;
    lds_double* b = (lds_double*)cx.lds + L::kL1;
    vload_as<NS>(b, lp); vload_as<NS>(b + L::DP, rp);
#pragma unroll
    for (int s = 0; s < NS; ++s) ps[s] = lp[s] + rp[s];   // the very sum the pair formed (nuts.py:386)
}
template <int NS, int W = 1, int PL = 0>
__device__ __forceinline__ void level1_store(const PairCtx& cx, const double (&lp)[NS], const double (&rp)[NS], const double (&pq)[NS]) {
    using L = PairLds<NS, W, PL>;
    lds_double* b = (lds_double*)cx.lds + L::kL1;
    vstore_as<NS>(b, lp); vstore_as<NS>(b + L::DP, rp); vstore_as<NS>(b + 2 * L::DP, pq);
}
// offset (doubles) of vector v of level j > nlds in the scratch row. Opaque to the optimiser on purpose: a
// loop-invariant level (the peeled j = 2) would otherwise get its per-lane 64-bit address precomputed outside the pair
// loop and kept in (spilled) registers; this way the access is scalar base + uniform offset + the lane-offset register
template <int NS, int W = 1, int PL = 0>
__device__ __forceinline__ unsigned glb_level_offset(const PairCtx& cx, int j, int v) {
    using L = PairLds<NS, W, PL>;
    unsigned off = L::kGlbLevels + static_cast<unsigned>(j - cx.nlds - 1) * (4u * L::DP) + static_cast<unsigned>(v) * L::DP;
    asm volatile("" : "+s"(off));
    return off;
}
// levels j >= 2: {lp, rp, psum, q}; vector index v in 0..3
template <int NS, int W = 1, int PL = 0>
__device__ __forceinline__ void levelN_load(const PairCtx& cx, int j, int v, double (&x)[NS]) {
    using L = PairLds<NS, W, PL>;
    int nl = cx.nlds;
    asm volatile("" : "+s"(nl));   // compared afresh (one s_cmp): hoisted, the loop-invariant test lives in a spilled lane mask
    if (j <= nl) vload_as<NS>((lds_double*)cx.lds + (L::kL2 + (j - 2) * 4 * L::DP + v * L::DP), x);
    else vload_as<NS>((glb_double*)cx.glb + glb_level_offset<NS, W, PL>(cx, j, v), x);
}
template <int NS, int W = 1, int PL = 0>
__device__ __forceinline__ void levelN_store(const PairCtx& cx, int j, int v, const double (&x)[NS]) {
    using L = PairLds<NS, W, PL>;
    int nl = cx.nlds;
    asm volatile("" : "+s"(nl));
    if (j <= nl) vstore_as<NS>((lds_double*)cx.lds + (L::kL2 + (j - 2) * 4 * L::DP + v * L::DP), x);
    else vstore_as<NS>((glb_double*)cx.glb + glb_level_offset<NS, W, PL>(cx, j, v), x);
}
// left-end momentum / proposal position of any level j >= 1
template <int NS, int W = 1, int PL = 0>
__device__ __forceinline__ void level_load_lp(const PairCtx& cx, int j, double (&x)[NS]) {
    if (j == 1) vload_as<NS>((lds_double*)cx.lds + PairLds<NS, W, PL>::kL1, x); else levelN_load<NS, W, PL>(cx, j, 0, x);
}
template <int NS, int W = 1, int PL = 0>
__device__ __forceinline__ void level_load_q(const PairCtx& cx, int j, double (&x)[NS]) {
    if (j == 1) vload_as<NS>((lds_double*)cx.lds + (PairLds<NS, W, PL>::kL1 + 2 * PairLds<NS, W, PL>::DP), x); else levelN_load<NS, W, PL>(cx, j, 3, x);
}
template <int NS, int W = 1, int PL = 0>
__device__ __forceinline__ void level_scal_put(const PairCtx& cx, int j, double w, double a, double pe, double plogp) {
    if (lane_id() == 0) {
        lds_double* s = (lds_double*)cx.lds + (PairLds<NS, W, PL>::kScal + 4 * j) + (W > 1 ? cx.wave_scal : 0);
        s[0] = w; s[1] = a; s[2] = pe; s[3] = plogp;
    }
}
// weights of level j only (the proposal's energy / log-density stay where they are until a node is parked or accepted)
template <int NS, int W = 1, int PL = 0>
__device__ __forceinline__ void level_scal_get_wa(const PairCtx& cx, int j, double& w, double& a) {
    const lds_double* s = (const lds_double*)cx.lds + (PairLds<NS, W, PL>::kScal + 4 * j) + (W > 1 ? cx.wave_scal : 0);
    w = s[0]; a = s[1];
}
// scalars of a node parked at level j: weights from the caller; the proposal's {energy, log-density} from their source --
// lane `elane` of (en, lp) when the proposal is a leaf of the current pair (src < 0: that lane stores them itself, no
// cross-lane read), else copied from level src
template <int NS, int W = 1, int PL = 0>
__device__ __forceinline__ void level_scal_park(const PairCtx& cx, int j, double w, double a, int src, int elane, double en, double lp) {
    lds_double* s = (lds_double*)cx.lds + (PairLds<NS, W, PL>::kScal + 4 * j) + (W > 1 ? cx.wave_scal : 0);
    const int lane = lane_id();
    if (src < 0) {
        if (lane == elane) { s[2] = en; s[3] = lp; }
    } else {
        const lds_double* f = (const lds_double*)cx.lds + (PairLds<NS, W, PL>::kScal + 4 * src) + (W > 1 ? cx.wave_scal : 0);
        const double pe = f[2], pl = f[3];
        if (lane == 0) { s[2] = pe; s[3] = pl; }
    }
    if (lane == 0) { s[0] = w; s[1] = a; }
}
template <int NS, int W = 1, int PL = 0>
__device__ __forceinline__ void level_scal_get(const PairCtx& cx, int j, double& w, double& a, double& pe, double& plogp) {
    const lds_double* s = (const lds_double*)cx.lds + (PairLds<NS, W, PL>::kScal + 4 * j) + (W > 1 ? cx.wave_scal : 0);   // same address in every lane: LDS broadcast
    w = s[0]; a = s[1]; pe = s[2]; plogp = s[3];
}

// first half of integration.py:100-121 plus the elementwise part of the second: leaves the per-lane partials of the
// kinetic energy and of the log-density for a batched reduction (a target that reduces its log-density itself
// contributes it from lane 0). v = var (.) p' on return.
template <int NS, class Target, class TeamT>
__device__ __forceinline__ void leapfrog_partial(TeamT& tm, const Target& tgt, const double (&var)[NS], double eps,
                                                 double (&q)[NS], double (&p)[NS], double (&g)[NS], double (&v)[NS],
                                                 double& kin_part, double& lp_part) {
    const double dt = 0.5 * eps;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        p[s] = p[s] + dt * g[s];
        const double vh = var[s] * p[s];
        q[s] = q[s] + eps * vh;
    }
    if constexpr (Target::kLanePartial) {
        lp_part = tgt.logp_grad_partial(tm, q, g);
    } else {
        const double lp = first_f64(tgt.logp_grad(tm, q, g));
        lp_part = (tm.tid() == 0) ? lp : 0.0;
    }
    double kin = 0.0;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        p[s] = p[s] + dt * g[s];
        v[s] = var[s] * p[s];
        kin = __builtin_fma(p[s], v[s], kin);
    }
    kin_part = kin;
}

// one cascade level: node a = {alp, arp, aps} (earlier), in-flight node {tl (left end), tps, right end velocity v}
template <int NS, int W = 1, int PL = 0>
__device__ __forceinline__ double cascade_dots(PairCtx& cx, const double (&var)[NS], const double (&alp)[NS],
                                               const double (&arp)[NS], const double (&aps)[NS], const double (&tl)[NS],
                                               double (&tps)[NS], const double (&v)[NS]) {
    double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0, d4 = 0.0, d5 = 0.0;
#pragma unroll
    for (int s = 0; s < NS; ++s) {   // nuts.py:386-396
        const double p1 = aps[s] + tl[s];
        const double p2 = arp[s] + tps[s];
        const double ps = aps[s] + tps[s];
        const double valp = var[s] * alp[s], vtlp = var[s] * tl[s], varp = var[s] * arp[s];
        d0 = __builtin_fma(ps, valp, d0); d1 = __builtin_fma(ps, v[s], d1);
        d2 = __builtin_fma(p1, valp, d2); d3 = __builtin_fma(p1, vtlp, d3);
        d4 = __builtin_fma(p2, varp, d4); d5 = __builtin_fma(p2, v[s], d5);
        tps[s] = ps;
    }
    red_put<NS, W, PL>(cx, 0, d0); red_put<NS, W, PL>(cx, 1, d1); red_put<NS, W, PL>(cx, 2, d2);
    red_put<NS, W, PL>(cx, 3, d3); red_put<NS, W, PL>(cx, 4, d4); red_put<NS, W, PL>(cx, 5, d5);
    return red_gather<NS, W, PL>(cx);
}

// On return the chain's row of A.q (qrow) holds the proposal; q is NOT updated (the caller reloads it).
template <int NS, int PL, class Target, class TeamT>
__device__ inline void nuts_transition2(TeamT& tm, const Target& tgt, const double (&var)[NS], RngState& rng,
                                        PairCtx& cx, double* qrow, const double (&q)[NS],
                                        const double (&p0)[NS], const double (&g0)[NS], double e0, double logp0,
                                        double step_size, double emax, int max_depth, bool momentum_f32,
                                        TransitionOut& out) {
    constexpr int W = TeamT::kWaves;
    // {cq, cp, cg}: the trajectory end that is being (or was last) extended (registers); the other end, the running
    // momentum sum and the extended end's momentum before the doubling are cold slots
    double cq[NS], cp[NS], cg[NS];
    vcopy(cq, q); vcopy(cp, p0); vcopy(cg, g0);
    cold_store<NS, W, PL, kColdOq>(cx, q); cold_store<NS, W, PL, kColdOp>(cx, p0); cold_store<NS, W, PL, kColdOg>(cx, g0);
    cold_store<NS, W, PL, kColdPsum>(cx, p0);
    bool c_right = true;                                    // which end {c*} is
    bool c_start = momentum_f32, o_start = momentum_f32;    // end still is the float32 start state
    double prop_e = e0, prop_logp = logp0;
    double coff = 0.0;
    // Sum of accepted subtree weights wn, of weight * min(1, e^{-dE}) an, and the start state's weight w_start are touched
    // once per doubling but would otherwise be loop-carried through the pair loop (and re-copied there every pair, because
    // the rare rescale path would write them): they live in slot 0 of this wave's level scalars {wn, an, w_start,
    // c_tot}. They are expressed in the weight offset
    // c_tot of the moment they were last merged and are brought to the current one only when a subtree is ACCEPTED: a
    // rejected subtree whose leaf moved the offset by more than ~745 would otherwise flush them to zero and the
    // acceptance statistic with them (0/0).
    lds_double* tot = (lds_double*)cx.lds + PairLds<NS, W, PL>::kScal + (W > 1 ? cx.wave_scal : 0);
    if (lane_id() == 0) { tot[0] = 0.0; tot[1] = 0.0; tot[2] = 1.0; tot[3] = 0.0; }
    int depth = 0, n_leap = 0;
    bool diverging = false, turning = false, exhausted = true;
    const bool odd_lane = (lane_id() & 1) != 0;
    UniformWindow win;
    window_reset(win);

    // Scalars of the n (1 or 2) leaves whose kinetic energy / log-density sums sit in lanes 7 / 15 (first leaf) and
    // 23 / 31 (second) of s (nuts.py:344-375): energy errors, divergence, linear-domain weights. On return lanes
    // 15 / 31 of en hold the energies, of ev the weights w = e^{-dE - c}, lanes 14 / 30 of ev w * min(1, e^{-dE}).
    // Returns the number of leaves accepted into the subtree (a diverging leaf stops the count).
    // The running signed max |dE| (nuts.py:356-357) is kept on lanes: lane 15 follows the first leaves of the pairs,
    // lane 31 the second ones; the two are combined when the transition ends (and whenever the slow path runs).
    double mde = 0.0;
    const bool lane15 = lane_id() == 15, lane31 = lane_id() == 31;
    auto mde_combined = [&]() -> double {
        const double a = readlane_f64(mde, 15), b = readlane_f64(mde, 31);
        return (fabs(b) > fabs(a)) ? b : a;
    };
    auto leaf_scalars = [&](double s, int n, double& en, double& ev) -> int {
        const ExpLanesConst ec = exp_lanes_const();            // requested first: the scalar load runs under the checks below
        en = 0.5 * dpp_f64<0x118>(s) - s;                      // row_shr:8 brings the kinetic sum next to the log-density
        double de = en - e0;
        int ok = 0;
        // Both leaves at once on their lanes: neither diverges (or is NaN) and no weight offset has to move -- the
        // usual case -- costs three compares and one ballot; anything else takes the sequential path below.
        const bool leaf_lane = lane15 | (lane31 & (n == 2));
        const unsigned long long leaf_bits = (n == 2) ? ((1ull << 15) | (1ull << 31)) : (1ull << 15);
        // (one ballot per compare: the mask of a compare IS its SGPR result, a ballot of an AND/OR of lane predicates is
        //  materialised as 0/1 in a VGPR and compared again)
        const unsigned long long rare = ballot64(!(fabs(de) < emax)) | ballot64((-de) - coff > 600.0);
        bool examined = leaf_lane;   // lanes whose leaf counts for the running max |dE|
        double de_x = de;            // ... and its energy error (NaN -> inf on the sequential path)
        if ((rare & leaf_bits) == 0ull) {
            n_leap += n;
            ok = n;
        } else {
            int seen = 0;
            for (int i = 0; i < n; ++i) {
                double dei = readlane_f64(de, 15 + 16 * i);
                ++n_leap; ++seen;
                if (isnan(dei)) dei = __builtin_inf();
                if (!(fabs(dei) < emax)) { diverging = true; break; }   // nuts.py:358,370-375
                const double x = -dei;
                if (x - coff > 600.0) {   // cold path: move the offset, rescale every stored weight
                    const double f = exp_uniform(coff - x);
                    const int lane = lane_id();
                    if (lane >= 1 && lane < kLevelScalDoubles / 4) {   // the parked levels {w, a, ..}; slot 0 (totals) keeps its own offset
                        lds_double* sc = (lds_double*)cx.lds + (PairLds<NS, W, PL>::kScal + 4 * lane) + (W > 1 ? cx.wave_scal : 0);
                        sc[0] = sc[0] * f; sc[1] = sc[1] * f;
                    }
                    coff = x;
                }
                ++ok;
            }
            de_x = isnan(de) ? __builtin_inf() : de;
            examined = lane15 | (lane31 & (seen == 2));
        }
        // running signed max |dE| (nuts.py:356-357), one definition for both paths (a second one would make it a
        // two-register loop-carried value that is copied back and forth every pair)
        const bool upd = (fabs(de_x) > fabs(mde)) & examined;
        mde = upd ? de_x : mde;
        // lanes 15 / 31: x - c; lanes 14 / 30: (x - c) + min(x, 0), i.e. log of w * min(1, e^{-dE})
        const double x = -de;
        const double xn = dpp_f64<0x101>(x);                   // row_shl:1: lane l <- lane l+1
        const double arg = odd_lane ? (x - coff) : ((xn - coff) + fmin(xn, 0.0));
        ev = exp_lanes<NS, W, PL>(cx, ec, arg);
        return ok;
    };

    for (int dd = 0; dd < max_depth; ++dd) {
        const bool right = team_uniform(tm, rng, win) < 0.5;   // nuts.py:213
        const double eps = right ? step_size : -step_size;
        if (right != c_right) {   // the other end becomes the one that is extended
            double t[NS];
            cold_load<NS, W, PL, kColdOq>(cx, t); cold_store<NS, W, PL, kColdOq>(cx, cq); vcopy(cq, t);
            cold_load<NS, W, PL, kColdOp>(cx, t); cold_store<NS, W, PL, kColdOp>(cx, cp); vcopy(cp, t);
            cold_load<NS, W, PL, kColdOg>(cx, t); cold_store<NS, W, PL, kColdOg>(cx, cg); vcopy(cg, t);
            const bool tb = c_start; c_start = o_start; o_start = tb;
            c_right = right;
        }
        cold_store<NS, W, PL, kColdAold>(cx, cp);   // momentum of the extended end before this doubling (nuts.py:332-338 operands)
        const bool aold_start = c_start;

        // subtree node under construction: momentum sum tps, weights; its right-end momentum is always the current cp,
        // its left-end momentum and proposal position are identified by their sources
        double tps[NS], eq[NS], ep[NS];
        double tw = 0.0, ta = 0.0;
        int qsrc = -1;   // proposal position: -2 first leaf of the last pair (eq), -1 the current state (cq), j >= 1 stack level j
        // The proposal's energy and log-density are not carried through the merges: for qsrc < 0 they sit on lane `elane`
        // of the last pair's (en_last, lp_last), for qsrc >= 1 in that level's scalars; they are fetched when the node is
        // parked or accepted. (The sum of w * min(1, e^{-dE}) could leave the merges the same way -- accumulated on lanes
        // 14 / 30 -- but the extra loop-carried register measured -1.6 % on depth-3 trees, 0 on C3.)
        int elane = 15;
        double en_last = 0.0, lp_last = 0.0;
        const int D = depth;
        if (D == 0) {
            double v[NS], kinp, lp, en, ev;
            leapfrog_partial<NS>(tm, tgt, var, eps, cq, cp, cg, v, kinp, lp);
            red_put<NS, W, PL>(cx, 0, kinp); red_put<NS, W, PL>(cx, 1, lp);   // (DPP butterflies for the lone leaf and the trajectory-level
            const double s0 = red_gather<NS, W, PL>(cx);                   //  test measured 1 % slower on depth-3 trees, equal on C3)
            if (leaf_scalars(s0, 1, en, ev) == 1) {
                tw = readlane_f64(ev, 15); ta = readlane_f64(ev, 14);
                en_last = en; lp_last = s0;
                vcopy(tps, cp);
            }
        } else {
            const int n_pairs = 1 << (D - 1);
            for (int k = 0; k < n_pairs; ++k) {
                double v[NS], kinA, lpA, kinB, lpB;
                leapfrog_partial<NS>(tm, tgt, var, eps, cq, cp, cg, v, kinA, lpA);
                red_put<NS, W, PL>(cx, 0, kinA); red_put<NS, W, PL>(cx, 1, lpA);
                vcopy(eq, cq); vcopy(ep, cp);
                leapfrog_partial<NS>(tm, tgt, var, eps, cq, cp, cg, v, kinB, lpB);   // speculative w.r.t. the first leaf's divergence test
                red_put<NS, W, PL>(cx, 2, kinB); red_put<NS, W, PL>(cx, 3, lpB);
#pragma unroll
                for (int s = 0; s < NS; ++s) tps[s] = ep[s] + cp[s];
                red_put<NS, W, PL>(cx, 4, pdot_v<NS>(tps, var, ep));   // var (.) ep is the first leaf's velocity, re-formed (same rounded product)
                red_put<NS, W, PL>(cx, 5, pdot<NS>(tps, v));
                const double s0 = red_gather<NS, W, PL>(cx);
                // this pair closes m right children (levels 1..m)
                const int m = __builtin_ctz(~static_cast<unsigned>(k) | (1u << (D - 1)));
                double en, ev;
                if (leaf_scalars(s0, 2, en, ev) != 2) break;
                // ---- level-0 merge (nuts.py:384-417 with depth 1: only the span check)
                {
                    const double wA = readlane_f64(ev, 15), aA = readlane_f64(ev, 14);
                    const double wB = readlane_f64(ev, 31), aB = readlane_f64(ev, 30);
                    const bool turn = red_any_nonpositive(s0, 4, 2);
                    const double wsum = wA + wB;
                    const bool take_b = uniform_true(team_uniform(tm, rng, win) * wsum < wB);   // drawn even if turning
                    qsrc = take_b ? -1 : -2;
                    elane = take_b ? 31 : 15;
                    en_last = en; lp_last = s0;
                    tw = wsum; ta = aA + aB;
                    if (turn) { turning = true; break; }
                }
                // ---- cascade level 1 (left end of the in-flight pair node = ep)
                if (m >= 1) {
                    double alp[NS], arp[NS], aps[NS];
                    double aw, aa;
                    level1_load<NS, W, PL>(cx, alp, arp, aps);   // (requesting it before the leaf scalars measured -8 %: one more live address)
                    level_scal_get_wa<NS, W, PL>(cx, 1, aw, aa);
                    const double sj = cascade_dots<NS, W, PL>(cx, var, alp, arp, aps, ep, tps, v);
                    const bool turn = red_any_nonpositive(sj, 0, 6);
                    const double wsum = aw + tw;
                    const bool take_b = uniform_true(team_uniform(tm, rng, win) * wsum < tw);
                    if (!take_b) qsrc = 1;
                    tw = wsum; ta = aa + ta;
                    if (turn) { turning = true; break; }
                }
                // ---- cascade levels 2..m: node a = stack[j]; the in-flight node's left end is stack[j-1]'s
                for (int j = 2; j <= m; ++j) {
                    double blp[NS], brp[NS], bps[NS], tl[NS];
                    levelN_load<NS, W, PL>(cx, j, 0, blp); levelN_load<NS, W, PL>(cx, j, 1, brp); levelN_load<NS, W, PL>(cx, j, 2, bps);
                    level_load_lp<NS, W, PL>(cx, j - 1, tl);
                    double bw, ba;
                    level_scal_get_wa<NS, W, PL>(cx, j, bw, ba);
                    const double sj = cascade_dots<NS, W, PL>(cx, var, blp, brp, bps, tl, tps, v);
                    const bool turn = red_any_nonpositive(sj, 0, 6);
                    const double wsum = bw + tw;
                    const bool take_b = uniform_true(team_uniform(tm, rng, win) * wsum < tw);
                    if (!take_b) qsrc = j;
                    tw = wsum; ta = ba + ta;
                    if (turn) { turning = true; break; }
                }
                if (turning) break;
                if (k + 1 < n_pairs) {   // park the node at level m + 1 (the last pair's cascade result stays in flight)
                    double tl[NS], tqv[NS];
                    if (m == 0) vcopy(tl, ep); else level_load_lp<NS, W, PL>(cx, m, tl);
                    if (qsrc == -1) vcopy(tqv, cq);
                    else if (qsrc == -2) vcopy(tqv, eq);
                    else level_load_q<NS, W, PL>(cx, qsrc, tqv);
#ifndef LMC_PARK_REORDER_MIN_NS
#define LMC_PARK_REORDER_MIN_NS 4
#endif
                    // Four-element slices (C4, C5: trees that live in the scratch row from level 2 on, and in C5 single chains
                    // whose 4 095-leapfrog trees are what a launch ends with): the node's scalars (LDS) go first and both loaded
                    // vectors are in registers before the node's FIRST scratch-row store. Left to itself the compiler waits for
                    // tqv's load right before the fourth store and again at the scalars, and at those joins of LDS / scratch-row
                    // paths the wait is "everything outstanding" -- the acknowledgement of the stores just issued (found reading
                    // the ISA, round 4). Measured, alternating runs on one box (profiles/r04_iteration_tail_ab.txt, box 3): C5
                    // +1 ... +2 %, C4 equal; C3 (two-element slices) -0.1 ... -0.4 %, hence the condition.
                    constexpr bool kParkReorder = NS >= LMC_PARK_REORDER_MIN_NS;
                    if constexpr (kParkReorder) level_scal_park<NS, W, PL>(cx, m + 1, tw, ta, qsrc, elane, en_last, lp_last);
                    if (m == 0) {
                        level1_store<NS, W, PL>(cx, tl, cp, tqv);
                    } else {
                        if constexpr (kParkReorder) {
#pragma unroll
                            for (int s = 0; s < NS; ++s) asm volatile("" : "+v"(tl[s]), "+v"(tqv[s]));
                        }
                        levelN_store<NS, W, PL>(cx, m + 1, 0, tl); levelN_store<NS, W, PL>(cx, m + 1, 1, cp);
                        levelN_store<NS, W, PL>(cx, m + 1, 2, tps); levelN_store<NS, W, PL>(cx, m + 1, 3, tqv);
                    }
                    if constexpr (!kParkReorder) level_scal_park<NS, W, PL>(cx, m + 1, tw, ta, qsrc, elane, en_last, lp_last);
                }
            }
        }
        ++depth;   // nuts.py:315
        if (diverging || turning) { exhausted = false; break; }

        // ---- accepted subtree: merge into the trajectory (nuts.py:321-340)
        double wn = tot[0], an = tot[1], w_start = tot[2];   // same address in every lane: LDS broadcast
        const double c_tot = tot[3];
        if (uniform_true(c_tot != coff)) {   // the offset moved inside this subtree: bring the accepted totals to it (rare)
            const double f = exp_uniform(first_f64(c_tot) - coff);
            wn = wn * f; an = an * f; w_start = w_start * f;
            if (lane_id() == 0) { tot[2] = w_start; tot[3] = coff; }
        }
        if (uniform_true(team_uniform(tm, rng, win) * (w_start + wn) < tw)) {   // biased progressive
            double tqv[NS];
            if (qsrc == -1) vcopy(tqv, cq);
            else if (qsrc == -2) vcopy(tqv, eq);
            else level_load_q<NS, W, PL>(cx, qsrc, tqv);
            vstore_as<NS>((glb_double*)qrow, tqv);
            if (qsrc < 0) {
                prop_e = readlane_f64(en_last, elane); prop_logp = readlane_f64(lp_last, elane);
            } else {
                double w_, a_;
                level_scal_get<NS, W, PL>(cx, qsrc, w_, a_, prop_e, prop_logp);
                prop_e = first_f64(prop_e); prop_logp = first_f64(prop_logp);
            }
        }
        if (lane_id() == 0) { tot[0] = wn + tw; tot[1] = an + ta; }
        double tlp[NS], psum[NS], op[NS], aold[NS];
        if (D == 0) vcopy(tlp, cp);
        else if (D == 1) vcopy(tlp, ep);
        else level_load_lp<NS, W, PL>(cx, D - 1, tlp);
        cold_load<NS, W, PL, kColdPsum>(cx, psum); cold_load<NS, W, PL, kColdOp>(cx, op); cold_load<NS, W, PL, kColdAold>(cx, aold);
#pragma unroll
        for (int s = 0; s < NS; ++s) {   // in place; float32 storage when the start momentum is float32
            const double t = psum[s] + tps[s];
            psum[s] = momentum_f32 ? static_cast<double>(static_cast<float>(t)) : t;
        }
        cold_store<NS, W, PL, kColdPsum>(cx, psum);
        double ov[NS], av[NS];   // velocities of the untouched end and of the extended end as it was before this doubling
        end_velocity<NS>(ov, var, op, o_start);
        end_velocity<NS>(av, var, aold, aold_start);
        double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0, d4 = 0.0, d5 = 0.0;
        if (right) {   // L = other end, old R = aold, subtree left end = tlp (adjacent to old R), right end = cp
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const double vtl = var[s] * tlp[s], vtr = var[s] * cp[s];
                const double p1 = psum[s] + tlp[s], p2 = aold[s] + tps[s];
                d0 = __builtin_fma(psum[s], ov[s], d0); d1 = __builtin_fma(psum[s], vtr, d1);
                d2 = __builtin_fma(p1, ov[s], d2);      d3 = __builtin_fma(p1, vtl, d3);
                d4 = __builtin_fma(p2, av[s], d4);      d5 = __builtin_fma(p2, vtr, d5);
            }
        } else {       // R = other end, old L = aold, subtree "left" end tlp is adjacent to old L, far end = cp
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const double vtl = var[s] * tlp[s], vtr = var[s] * cp[s];
                const double p1 = tps[s] + aold[s], p2 = tlp[s] + psum[s];
                d0 = __builtin_fma(psum[s], vtr, d0);   d1 = __builtin_fma(psum[s], ov[s], d1);
                d2 = __builtin_fma(p1, vtr, d2);        d3 = __builtin_fma(p1, av[s], d3);
                d4 = __builtin_fma(p2, vtl, d4);        d5 = __builtin_fma(p2, ov[s], d5);
            }
        }
        c_start = false;
        red_put<NS, W, PL>(cx, 0, d0); red_put<NS, W, PL>(cx, 1, d1); red_put<NS, W, PL>(cx, 2, d2);
        red_put<NS, W, PL>(cx, 3, d3); red_put<NS, W, PL>(cx, 4, d4); red_put<NS, W, PL>(cx, 5, d5);
        if (red_any_nonpositive(red_gather<NS, W, PL>(cx), 0, 6)) { turning = true; exhausted = false; break; }
    }

    const double wn_end = first_f64(tot[0]), an_end = first_f64(tot[1]);
    const double mean_accept = (wn_end > 0.0) ? first_f64(an_end / wn_end) : 0.0;
    out.accept = mean_accept;
    out.energy = prop_e;
    out.energy_error = first_f64(prop_e - e0);
    out.max_energy_error = mde_combined();
    out.model_logp = prop_logp;
    out.depth = depth;
    out.n_leapfrog = n_leap;
    out.diverging = diverging;
    out.exhausted = exhausted;
    out.accepted = 0;
}

// ---- the iteration kernel: n_iters x _astep for every chain, no host round trips ------------------------
// Occupancy target per vector width (waves per SIMD; the VGPR budget is 512 / waves): the per-chain state
// is register resident, so wider per-thread slices trade occupancy for registers. Chains longer than
// 128 elements are spread over W waves (dpad = 64 * NS * W) instead of growing NS further.
#ifndef LMC_NORMALS_IN_REGS
#define LMC_NORMALS_IN_REGS 1
#endif
#ifndef LMC_WAVES_NS1
#define LMC_WAVES_NS1 4
#endif
#ifndef LMC_WAVES_NS2
#define LMC_WAVES_NS2 3
#endif
#ifndef LMC_WAVES_NS4
#define LMC_WAVES_NS4 2
#endif
#ifndef LMC_WAVES_NS2_W8
#define LMC_WAVES_NS2_W8 4   // eight waves per chain = two per SIMD: two chains per CU need four waves per SIMD (128 VGPRs)
#endif
constexpr int run_waves_per_simd(int ns, int w = 1) {
    return (ns == 2 && w == 8) ? LMC_WAVES_NS2_W8 : ns <= 1 ? LMC_WAVES_NS1 : ns == 2 ? LMC_WAVES_NS2 : ns == 4 ? LMC_WAVES_NS4 : 1;
}
// LDS carve (doubles) behind the subtree stack: MT19937 state (624 words), team exchange area, RNG re-broadcast
constexpr int kLdsMtDoubles = 320;
#ifndef LMC_MT_IN_LDS_W1
#define LMC_MT_IN_LDS_W1 1   // one-wave kernels: 1 keeps the MT19937 state in LDS for the launch, 0 uses it in place (L2)
#endif
constexpr bool run_mt_in_lds(int w) { return w > 1 || LMC_MT_IN_LDS_W1; }
constexpr int lds_tail_doubles(int w) {   // W >= 4: a second MT19937 buffer behind everything (team_normals_parallel)
    return w == 1 ? (run_mt_in_lds(1) ? kLdsMtDoubles : 0) : kLdsMtDoubles + 2 * w * kTeamSlots + 4 + (w >= 4 ? kLdsMtDoubles : 0);
}
// The engine switches a launch to the deep-tree LDS plan when the chains report at least kPlanUp leapfrogs per iteration and back
// when they report at most kPlanDown (hysteresis; the measured break-even is ~20: the plan costs ~2 800 cycles per iteration --
// the momentum draw reads MT19937 through L2 -- and saves ~150 per leapfrog of a deep tree).
constexpr int kPlanUp = 24, kPlanDown = 14;

// ---- pieces of the iteration body shared by the diagonal and the dense-mass kernels ---------------------------
// lmc_engine_request_stop(): the host's Ctrl-C (sampling.py:324-328, :470-471 in the reference: keep what has been drawn).
// The request is a host store into pinned memory -- nothing on the device has to be scheduled for it to arrive (a device
// word set through a stream reached whole-job launches only after the job, tools/ubench/stop_probe.hip). Reading host
// memory is slow, though (an uncached dword over the host link: every chain doing it every 16th iteration halved the rate
// of short iterations), so only one chain in (relay_mask + 1) <= 256 of a launch reads it, at the first iteration of the
// launch and then every 16th iteration, and RELAYS a set word to a device word; that one every chain looks at once per
// iteration -- requested when the iteration starts, looked at when it ends (its latency hides behind the whole iteration).
// WHICH chains relay rotates with the iteration index (chain + git / 16 = 0 mod the mask), so that no particular chain
// has to be alive for the word to arrive (a chain that stopped with "Bad initial energy" used to be able to take the
// relay with it). A team agrees on ONE value (thread 0's) so that no wave leaves a barrier behind.
// A launch that STARTS under a request does nothing at all (stop_at_entry): sample() keeps two launches of a job in flight
// (bench.py likewise), and the one still queued when Ctrl-C arrives must neither run an iteration nor touch iter_count.
// The relay also leaves ITS OWN mean tree size of this launch so far (`leaps` leapfrogs in `it` iterations; leaps < 0: nothing to
// report) in A.tree_hint, a pinned host word: what the engine picks the next launch's LDS plan from, without touching a stream.
// Computed inside the relay branch only -- no chain pays for it per iteration.
template <class CA, class PT>
__device__ __forceinline__ int stop_request_load(const CA& A, const PT& P, int chain_in_launch, int it, long long git, long long leaps = -1) {
    int* dev = A.stop_dev;
    const int mask = P.relay_mask;
    const bool relay = (it == 0) ? ((chain_in_launch & mask) == 0)
                                 : ((git & 15) == 0 && ((chain_in_launch + static_cast<int>(git >> 4)) & mask) == 0);
    if (relay) {   // wave-uniform
        if (__hip_atomic_load(A.stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0)
            __hip_atomic_store(dev, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // ... and leaves word of where the job is (a posted store into the same pinned block)
        __hip_atomic_store(A.progress, static_cast<int>(git), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (leaps >= 0 && it >= 8 && A.tree_hint != nullptr) {
            int mean = static_cast<int>(static_cast<float>(leaps) / static_cast<float>(it));
            mean = mean < 1 ? 1 : (mean > 4095 ? 4095 : mean);
            const int at = git > 0x7ffff ? 0x7ffff : static_cast<int>(git);   // which iteration reports (the engine ignores the first 100)
            __hip_atomic_store(A.tree_hint, (at << 12) | mean, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    return __hip_atomic_load(dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <class TeamT>
__device__ __forceinline__ bool stop_requested(TeamT& tm, int loaded, double* bcast) {
    if constexpr (TeamT::kWaves == 1) {
        return first_i32(loaded) != 0;
    } else {
        tm.sync();
        if (tm.tid() == 0) bcast[3] = static_cast<double>(loaded);
        tm.sync();
        return first_f64(bcast[3]) != 0.0;
    }
}
// the device word as a launch finds it; `word` = one LDS int the workgroup may use for the agreement (W > 1: the waves of
// a team could otherwise read different values and part at the first barrier)
template <int W>
__device__ __forceinline__ bool stop_at_entry(const int* stop_dev, int* word) {
    const int s = __hip_atomic_load(const_cast<int*>(stop_dev), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if constexpr (W == 1) {
        return first_i32(s) != 0;
    } else {
        if (threadIdx.x == 0) *word = s;
        __syncthreads();
        const int agreed = *word;
        __syncthreads();   // the word's LDS may be reused right away
        return first_i32(agreed) != 0;
    }
}
struct DualAverage {   // step_sizes.py:49-99, wave-uniform
    double log_step, log_bar, hbar, mu;
    int count;
    double step_now, step_bar_now;   // exp(log_step), exp(log_bar)
};
template <class CA>
__device__ __forceinline__ void dual_average_load(const CA& A, int c, DualAverage& da) {
    da.log_step = first_f64(A.da[c * 4 + 0]);
    da.log_bar = first_f64(A.da[c * 4 + 1]);
    da.hbar = first_f64(A.da[c * 4 + 2]);
    da.mu = first_f64(A.da[c * 4 + 3]);
    da.count = first_i32(A.da_count[c]);
    da.step_now = exp_uniform(da.log_step);
    da.step_bar_now = exp_uniform(da.log_bar);
}
template <class CA, class PT>
__device__ __forceinline__ void dual_average_update(const CA& A, const PT& P, double accept, DualAverage& da) {
    const double w = 1.0 / (static_cast<double>(da.count) + P.t0);
    da.hbar = first_f64((1.0 - w) * da.hbar + w * (P.target_accept - accept));
    // sqrt(count) and count ** -k come from host-built tables (glibc sqrt/pow: the very values the
    // reference's Python floats get); the device pow is only the fallback beyond the table
    double sq, mk;
    if (da.count < A.da_table_len) {
#ifndef LMC_DA_TABLE_SLOAD
#define LMC_DA_TABLE_SLOAD 1
#endif
#if LMC_DA_TABLE_SLOAD
        // The tables are written once by the host before any launch and count is wave-uniform: read them through the scalar
        // cache (s_load_dwordx2, no vector-memory round trip and no vmcnt wait behind the reload of q; round 4).
        typedef const __attribute__((address_space(4))) double cst_double;
        cst_double* ts = (cst_double*)A.da_sqrt;
        cst_double* tk = (cst_double*)A.da_mk;
        asm volatile("" : "+s"(ts), "+s"(tk));
        sq = ts[da.count];
        mk = tk[da.count];
#else
        sq = first_f64(A.da_sqrt[da.count]);
        mk = first_f64(A.da_mk[da.count]);
#endif
    } else {
        sq = sqrt(static_cast<double>(da.count));
        mk = pow(static_cast<double>(da.count), -P.k);
    }
    da.log_step = first_f64(da.mu - da.hbar * sq / P.gamma);
    da.log_bar = first_f64(mk * da.log_step + (1.0 - mk) * da.log_bar);
    ++da.count;
    da.step_now = exp_uniform(da.log_step);
    da.step_bar_now = exp_uniform(da.log_bar);
}

// running per-chain moments of the post-warm-up draws (optional): enough for R-hat without a trace
template <int NS, class TeamT, class CA>
__device__ __forceinline__ void moments_update(const CA& A, TeamT& tm, int c, long long row, const double (&q)[NS]) {
    const int n_new = first_i32(A.mom_n[c]) + 1;
    const double inv_n = 1.0 / static_cast<double>(n_new);
    double mm[NS], m2[NS];
    vload<NS>(A.mom_mean + row, mm); vload<NS>(A.mom_m2 + row, m2);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const double dlt = q[s] - mm[s];
        mm[s] = mm[s] + dlt * inv_n;
        m2[s] = m2[s] + dlt * (q[s] - mm[s]);
    }
    vstore<NS>(A.mom_mean + row, mm); vstore<NS>(A.mom_m2 + row, m2);
    tm.sync();
    if (tm.tid() == 0) A.mom_n[c] = n_new;
}

// draw row + per-draw statistics of iteration `git`
// The record's value for this lane (lane k < 7: statistic k; lane 7: the integers), pure register arithmetic.
__device__ __forceinline__ double stat_record_value(int tid, const TransitionOut& out, double step_now, double step_bar_now, bool tune) {
    double v = step_now;
    v = (tid == kSfStepSizeBar) ? step_bar_now : v;
    v = (tid == kSfAccept) ? out.accept : v;
    v = (tid == kSfEnergyError) ? out.energy_error : v;
    v = (tid == kSfEnergy) ? out.energy : v;
    v = (tid == kSfMaxEnergyError) ? out.max_energy_error : v;
    v = (tid == kSfModelLogp) ? out.model_logp : v;
    const unsigned flags = (static_cast<unsigned>(out.depth) & 0xffffu) | (out.diverging ? kRecDiverging : 0u) |
                           (tune ? kRecTune : 0u) | (out.accepted ? kRecAccepted : 0u);
    const double ints = __hiloint2double(static_cast<int>(flags), out.n_leapfrog);
    return (tid == 7) ? ints : v;
}
// the two stores of a draw: the trace row and the 64-byte record (eight lanes, ONE store)
template <int NS, class CA>
__device__ __forceinline__ void store_outputs(const CA& A, int c, int tid, long long git, const double (&q)[NS], double rec) {
    const int d = A.d;
    const long long orow = static_cast<long long>(c) * A.cap + git;
    if (A.trace != nullptr && git >= A.trace_begin) {
        double* tr = A.trace + (static_cast<long long>(c) * (A.cap - A.trace_begin) + (git - A.trace_begin)) * d;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int e = tid * NS + s;
            if (e < d) tr[e] = q[s];
        }
    }
    if (tid < 8) reinterpret_cast<double*>(A.stat_rec + orow)[tid] = rec;
}
template <int NS, class CA>
__device__ __forceinline__ void write_outputs(const CA& A, int c, int tid, long long git, const double (&q)[NS],
                                              const TransitionOut& out, double step_now, double step_bar_now, bool tune) {
    store_outputs<NS>(A, c, tid, git, q, stat_record_value(tid, out, step_now, step_bar_now, tune));
}

// diagonal mass adaptation (quadpotential.py:231-245, :324-340): both Welford estimators take the draw, the
// foreground one becomes the float32 variance, the window switches every P.window samples
struct MassScalars { double wsum_f, wsum_b; int wsel, n_samples, window; };
// The four estimator rows of a chain (foreground / background mean and raw variance): all requested before any is used,
// one HBM round trip per update instead of two.
template <int NS, class CA>
__device__ __forceinline__ void diag_mass_prefetch(const CA& A, long long row, const MassScalars& ms,
                                                   double (&m)[NS], double (&r)[NS], double (&mb)[NS], double (&rb)[NS]) {
    const long long plane = static_cast<long long>(A.chains) * A.dpad;
    vload<NS>(A.wmean + ms.wsel * plane + row, m); vload<NS>(A.wraw + ms.wsel * plane + row, r);
    vload<NS>(A.wmean + (1 - ms.wsel) * plane + row, mb); vload<NS>(A.wraw + (1 - ms.wsel) * plane + row, rb);
}
template <int NS, class CA, class PT>
__device__ __forceinline__ void diag_mass_update(const CA& A, const PT& P, long long row, int tid,
                                                 const double (&q)[NS], float (&var)[NS], float (&inv_std)[NS],
                                                 double (&vard)[NS], MassScalars& ms,
                                                 double (&m)[NS], double (&r)[NS], double (&mb)[NS], double (&rb)[NS]) {
    const int d = A.d;
    const long long plane = static_cast<long long>(A.chains) * A.dpad;
    double* fm = A.wmean + ms.wsel * plane + row;
    double* fr = A.wraw + ms.wsel * plane + row;
    double* bm = A.wmean + (1 - ms.wsel) * plane + row;
    double* br = A.wraw + (1 - ms.wsel) * plane + row;
    ms.wsum_f += 1.0;
    ms.wsum_b += 1.0;
    const double prop_f = first_f64(1.0 / ms.wsum_f), prop_b = first_f64(1.0 / ms.wsum_b);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const double od = q[s] - m[s];
        m[s] = m[s] + prop_f * od;
        const double nd = q[s] - m[s];
        r[s] = r[s] + 1.0 * od * nd;
        const int e = tid * NS + s;
        if (e < d) {
            var[s] = static_cast<float>(r[s] / ms.wsum_f);
            const float sd = sqrtf(var[s]);
            inv_std[s] = 1.0f / sd;
            vard[s] = static_cast<double>(var[s]);
        }
    }
    vstore<NS>(fm, m); vstore<NS>(fr, r);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const double od = q[s] - mb[s];
        mb[s] = mb[s] + prop_b * od;
        const double nd = q[s] - mb[s];
        rb[s] = rb[s] + 1.0 * od * nd;
        m[s] = mb[s]; r[s] = rb[s];
    }
    if (ms.n_samples > 0 && ms.n_samples % ms.window == 0) {   // background becomes foreground
        vstore<NS>(bm, m); vstore<NS>(br, r);
#pragma unroll
        for (int s = 0; s < NS; ++s) { m[s] = 0.0; r[s] = 0.0; }
        vstore<NS>(fm, m); vstore<NS>(fr, r);             // old foreground = fresh background
        ms.wsum_f = ms.wsum_b;
        ms.wsum_b = 0.0;
        ms.wsel = 1 - ms.wsel;
        ms.window = static_cast<int>(static_cast<double>(ms.window) * P.window_multiplier);   // quadpotential.py:243
    } else {
        vstore<NS>(bm, m); vstore<NS>(br, r);
    }
    ++ms.n_samples;
}

// RNG = 0: the reference's stream (numpy legacy MT19937 + polar method, same-seed parity); 1: momentum from Philox
// (philox_normals: the throughput mode, its own kernel instantiation so that the parity kernels are untouched by it)
// PL: the LDS plan (PairLds<NS, W, PL>). 0 keeps the MT19937 state and three cold slots in LDS -- best for shallow trees, where the
// momentum draw is a fifth of an iteration; 1 (one-wave kernels) uses the generator in place and keeps stack level 2 in LDS
// instead -- best for deep trees, where every other pair cascades through it. The plans differ in WHERE a chain's private data
// lives, never in arithmetic: results are bit-identical (tests/test_gpu_round5.py::test_lds_plans_are_bit_identical), so the
// engine is free to pick per launch from the tree sizes the chains report (A.tree_hint; lmc_engine.hip: choose_lds_plan).
// (Both plans as two bodies of ONE kernel with every chain choosing for itself were built first and lose 7-20 % under
// either plan -- profiles/r05_lds_plan_dual_body_ab.txt -- the kernel per plan keeps each plan's code as it was measured.)
template <int NS, int W, template <int> class TargetT, int RNG = 0, int PL = 0>
__global__ __launch_bounds__(64 * W, run_waves_per_simd(NS, W)) void run_kernel(ChainArrays, SamplerParams, const double* tparams) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const long long t_resident = wall_clock64();   // constant-rate clock: the chain's residence time (kCtWaveTicks)
    // the two argument structs are read from the kernarg segment region by region (KernArgs above), never held by value
    const KernArgs ka = KernArgs::get();
    KChainArrays& A0 = ka.A();
    KSamplerParams& P0 = ka.P();
    const int c = blockIdx.x + P0.chain_begin;
    const int d = A0.d, dpad = A0.dpad;
    const long long row = static_cast<long long>(c) * dpad;
    const int lds_doubles = P0.lds_doubles;
    Team<W> tm;
    tm.xbuf = lds + lds_doubles + kLdsMtDoubles;
    tm.parity = 0;
    double* rng_bcast = tm.xbuf + 2 * W * kTeamSlots;
    const int tid = tm.tid();

    if (stop_at_entry<W>(A0.stop_dev, reinterpret_cast<int*>(lds))) return;   // queued behind a Ctrl-C: nothing runs, nothing is touched
    if (A0.status[c] & kStatusBadInitialEnergy) return;   // chain already aborted (ValueError on host)

    TargetT<NS> tgt;
    tgt.init(tm, tparams, d);

    // ---- load persistent chain state
    double q[NS];
    float var[NS], inv_std[NS];
    double vard[NS];   // the float32 mass promoted once (every use is a float64 product)
    double* const qrow = A0.q + row;
    vload<NS>(qrow, q);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        var[s] = A0.var[row + tid * NS + s];
        inv_std[s] = A0.inv_std[row + tid * NS + s];
        vard[s] = static_cast<double>(var[s]);
    }
    // The MT19937 state lives in LDS for the whole launch (2.5 KB per wave, behind the subtree stack): the
    // momentum draw, the uniform window and the twist then cost LDS latency instead of HBM/L2 round trips.
    RngState rng;
    uint32_t* mt_glb = A0.mt + static_cast<long long>(c) * kMtN;
    uint32_t* mt_lds = reinterpret_cast<uint32_t*>(lds + lds_doubles);
    uint32_t* mt_lds2 = reinterpret_cast<uint32_t*>(rng_bcast + 4);   // W >= 4 only: the generation being twisted out of place
    constexpr bool kMtInLds = run_mt_in_lds(W) && PL == 0;
    if constexpr (kMtInLds) {
        for (int i = tid; i < kMtN; i += 64 * W) mt_lds[i] = mt_glb[i];
        tm.sync();
        rng.mt = mt_lds;
    } else {
        rng.mt = mt_glb;   // used in place (L2): the LDS it would take holds subtree-stack data instead
    }
    rng.pos = first_i32(A0.rng_pos[c]);
    rng.has_gauss = first_i32(A0.rng_has_gauss[c]);
    rng.gauss = first_f64(A0.rng_gauss[c]);
    const uint32_t chain_seed = RNG == 1 ? first_u32(A0.seed[c]) : 0u;
    DualAverage da;
    dual_average_load(A0, c, da);
    int iter_count = first_i32(A0.iter_count[c]);
    MassScalars ms;
    ms.n_samples = first_i32(A0.n_samples[c]);
    ms.wsel = first_i32(A0.wsel[c]);
    ms.wsum_f = first_f64(A0.wsum[c * 2 + ms.wsel]);
    ms.wsum_b = first_f64(A0.wsum[c * 2 + (1 - ms.wsel)]);
    ms.window = first_i32(A0.awindow[c]);
    long long ct_maxdepth = 0, ct_divs = 0, ct_after = 0, ct_leap = 0;
    int status = 0;

    // compile-time LDS plan (PairLds<NS, W, PL>), the chain's scratch row for what does not fit
    static_assert(NS <= 4, "sampling kernels hold at most four elements per lane");
    PairCtx cx;
    cx.lds = lds;
    cx.glb = A0.scratch + static_cast<long long>(c) * A0.scratch_stride;
    cx.nlds = P0.nlds;
    cx.wave = tm.wave();
    cx.wave_red = W > 1 ? cx.wave * PairLds<NS, W>::kRedWave : 0;
    cx.wave_scal = W > 1 ? cx.wave * kLevelScalDoubles : 0;
    cx.red_lane = static_cast<int>(reinterpret_cast<size_t>((lds_double*)lds + (red_lane_init() + cx.wave_red)));
    cx.xpar = 0;
    if (tid < kExpTableDoubles) lds[PairLds<NS, W>::kExp + tid] = kExp2Table[2 * tid];
    tm.sync();

#ifdef LMC_PHASE_TIMING   // diagnostic build (tools/phase_timing.py): s_memtime ticks per phase replace three counters
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long ph_t = clock64();
#define LMC_PHASE(i) { const unsigned long long now_ = clock64(); ph[i] += now_ - ph_t; ph_t = now_; }
#else
#define LMC_PHASE(i)
#endif
    const int n_iters = P0.n_iters;
    for (int it = 0; it < n_iters; ++it) {
        // ---- region 1 of the arguments: iteration head (momentum draw, start state, step size, transition inputs)
        KSamplerParams& P = ka.P();
        const long long git = P.iter_begin + it;
        const bool tune = git < P.n_tune;
        const bool momentum_f32 = P.momentum_f32 != 0;
        LMC_PHASE(5)
        const int stop_word = stop_request_load(ka.A(), P, static_cast<int>(blockIdx.x), it, git, P.kind == 0 ? ct_leap : -1);

        // ---- momentum draw (quadpotential.py:221-224 / :374-376)
        double p0[NS];
        if constexpr (RNG == 1) {
            double z[NS];
            philox_normals<NS>(chain_seed, git, tid, d, z);
#pragma unroll
            for (int s = 0; s < NS; ++s)
                p0[s] = momentum_f32 ? static_cast<double>(inv_std[s] * static_cast<float>(z[s])) : z[s] * static_cast<double>(inv_std[s]);
        } else {
            double zr[NS];
            if constexpr (W == 1 && LMC_NORMALS_IN_REGS) {
                rng_normals_owned<NS>(rng, d, lds, lds + dpad, zr);   // even d: the variates never travel through LDS
            } else {
                team_normals(tm, rng, d, lds, lds + dpad, rng_bcast, mt_lds, mt_lds2);   // level-0 LDS region (2*dpad doubles) = normals + staging
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const int e = tid * NS + s;
                    zr[s] = (e < d) ? lds[e] : 0.0;
                }
                tm.sync();
            }
#pragma unroll
            for (int s = 0; s < NS; ++s)
                p0[s] = momentum_f32 ? static_cast<double>(inv_std[s] * static_cast<float>(zr[s]))
                                     : zr[s] * static_cast<double>(inv_std[s]);
        }
        LMC_PHASE(0)

        // ---- start state (integration.py:52-66)
        double g0[NS];
        const double logp0 = first_f64(tgt.logp_grad(tm, q, g0));
        double e0;
        if (momentum_f32) {   // float32 velocity, float32 kinetic energy (BLAS-order faithful)
            const float kin = start_kinetic_f32<NS>(tm, p0, var, d, P.sdot_mode, reinterpret_cast<float*>(lds), dpad);
            e0 = first_f64(static_cast<double>(kin) - logp0);
        } else {
            e0 = first_f64(0.5 * tm.sum(pdot_v<NS>(p0, vard, p0)) - logp0);
        }
        if (!isfinite(e0)) {   // base_hmc.py:145-148: the reference raises; the chain stops here
            status |= kStatusBadInitialEnergy;
            break;
        }

        // ---- step size for this iteration (base_hmc.py:151-153)
        const bool adapt_step = tune && P.adapt_step_size;
        const double step_size = jitter_step_size(tm, rng, ka.A(), P, c, adapt_step ? da.step_now : da.step_bar_now);   // exp(log_step) / exp(log_bar)

        LMC_PHASE(1)
        TransitionOut out;
        if (P.kind == 0) {
            const int md = (tune && iter_count < 200) ? P.early_max_treedepth : P.max_treedepth;
            nuts_transition2<NS, PL>(tm, tgt, vard, rng, cx, qrow, q, p0, g0, e0, logp0, step_size, P.emax, md,
                                 momentum_f32, out);
            vload<NS>(qrow, q);   // the proposal was written to the chain's row of A.q
            // (handing it over in registers when the last doubling accepted it measured -4 % on depth-3 trees)
            if (out.exhausted && !tune) ++ct_maxdepth;
        } else {
            hmc_transition<NS>(tm, tgt, vard, rng, q, p0, g0, e0, logp0, step_size, P.emax, P.path_length,
                               P.max_steps, out);
        }
        ct_leap += out.n_leapfrog;
        LMC_PHASE(2)

        // ---- region 2 of the arguments: everything after the transition
        KChainArrays& A = ka.A();
        KSamplerParams& P2 = ka.P();
        // ---- dual averaging (step_sizes.py:71-92)
        if (adapt_step) dual_average_update(A, P2, out.accept, da);
        LMC_PHASE(3)

        // ---- the stop word, looked at HERE (round 4): it was requested when the iteration started, so the wait in front of its
        // first use is a wait for whatever vector-memory operation is still outstanding. At the very end of the iteration
        // (where it used to be looked at) that was the acknowledgement of the stores just issued -- the estimator rows, the
        // trace row, the statistics record: a store round trip per iteration spent doing nothing. Here the only thing
        // outstanding is the reload of q above (needed for the stores below anyway; while tuning the dual-averaging update
        // has already waited for it), and the record's value is formed first so that its arithmetic runs under that wait.
        // The stores of the iteration then drain under the next iteration's momentum draw. Measured, alternating runs on one
        // box (profiles/r04_iteration_tail_ab.txt), together with the dual-averaging tables read through the scalar cache:
        // north_star shape +0.4 ... +1.0 %, C2 +1 ... +1.5 %, C4 +0.4 ... +1.1 %, C3 equal (its iteration is 25 times longer).
#ifndef LMC_EARLY_STOP_CHECK
#define LMC_EARLY_STOP_CHECK 1
#endif
        double rec_value = 0.0;
        bool stop_now = false;
        if constexpr (LMC_EARLY_STOP_CHECK != 0) {
            rec_value = stat_record_value(tid, out, da.step_now, da.step_bar_now, tune);
            int sw = stop_word;
            asm volatile("" : "+v"(rec_value), "+v"(sw));   // the record's value is complete before the word is waited for
            stop_now = stop_requested(tm, sw, rng_bcast);
        }

        // ---- diagonal mass adaptation (quadpotential.py:231-245, :324-340). (Requesting the estimator rows before the
        // dual-averaging update, to take their HBM round trip off the critical path, measured -12 % at d = 128: sixteen
        // more live registers across the update spill other state. Requesting them in front of the record's arithmetic and
        // the stop word above -- only ~50 instructions under the round trip -- still loses: north_star shape -1.4 %, C3
        // -1.5 %, C4 -5 %, profiles/r04_iteration_tail_ab.txt.)
        if (tune && P2.adapt_mass) {
            double wm[NS], wr[NS], wmb[NS], wrb[NS];
            diag_mass_prefetch<NS>(A, row, ms, wm, wr, wmb, wrb);
            diag_mass_update<NS>(A, P2, row, tid, q, var, inv_std, vard, ms, wm, wr, wmb, wrb);
        }

        LMC_PHASE(4)
        // ---- bookkeeping (base_hmc.py:164-190)
        if (out.diverging && !tune) ++ct_divs;
        ++iter_count;
        if (!tune) ++ct_after;

        if (A.mom_mean != nullptr && !tune) moments_update<NS>(A, tm, c, row, q);
        if constexpr (LMC_EARLY_STOP_CHECK != 0) {
            store_outputs<NS>(A, c, tid, git, q, rec_value);
            if (stop_now) break;
        } else {
            write_outputs<NS>(A, c, tid, git, q, out, da.step_now, da.step_bar_now, tune);
            if (stop_requested(tm, stop_word, rng_bcast)) break;
        }
    }

    // ---- store persistent chain state (region 3 of the arguments)
    KChainArrays& A = ka.A();
    tm.sync();
    if constexpr (kMtInLds) {
        for (int i = tid; i < kMtN; i += 64 * W) mt_glb[i] = rng.mt[i];   // (a team's current generation may be in either buffer)
    }
    vstore<NS>(qrow, q);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        A.var[row + tid * NS + s] = var[s];
        A.inv_std[row + tid * NS + s] = inv_std[s];
    }
    if (tid == 0) {
        A.rng_pos[c] = rng.pos;
        A.rng_has_gauss[c] = rng.has_gauss;
        A.rng_gauss[c] = rng.gauss;
        A.da[c * 4 + 0] = da.log_step;
        A.da[c * 4 + 1] = da.log_bar;
        A.da[c * 4 + 2] = da.hbar;
        A.da_count[c] = da.count;
        A.iter_count[c] = iter_count;
        A.n_samples[c] = ms.n_samples;
        A.wsel[c] = ms.wsel;
        A.awindow[c] = ms.window;
        A.wsum[c * 2 + ms.wsel] = ms.wsum_f;
        A.wsum[c * 2 + (1 - ms.wsel)] = ms.wsum_b;
        A.status[c] |= status;
        A.counters[c * kNumCounters + kCtMaxTreedepth] += ct_maxdepth;
        A.counters[c * kNumCounters + kCtDivsSample] += ct_divs;
        A.counters[c * kNumCounters + kCtSamplesAfterTune] += ct_after;
        A.counters[c * kNumCounters + kCtLeapfrogs] += ct_leap;
        A.counters[c * kNumCounters + kCtWaveTicks] += static_cast<long long>(W) * (wall_clock64() - t_resident);
#ifdef LMC_PHASE_TIMING
        LMC_PHASE(5)
        A.counters[c * kNumCounters + 0] += static_cast<long long>(((ph[0] & 0xffffffffull) << 32) | (ph[1] & 0xffffffffull)) - ct_maxdepth;
        A.counters[c * kNumCounters + 1] += static_cast<long long>(((ph[2] & 0xffffffffull) << 32) | (ph[3] & 0xffffffffull)) - ct_divs;
        A.counters[c * kNumCounters + 2] += static_cast<long long>(((ph[4] & 0xffffffffull) << 32) | (ph[5] & 0xffffffffull)) - ct_after;
#endif
    }
}

}  // namespace lmc
