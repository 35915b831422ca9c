// The general ("wide") sampling kernel: every configuration of the hot path the fused kernels do not instantiate.
//
//   model_ndim > 1024 (diagonal mass; the reference has no limit: base_hmc.py:102 takes any model_ndim)
//   dense mass matrices beyond 256 dimensions (QuadPotentialFull / FullInv, quadpotential.py:388-468)
//   QuadPotentialDiagAdapt(dtype="float64") (quadpotential.py:159,175-184)
//   a run-time compiled user density with a dense mass matrix
//
// One chain = one team of W wavefronts (W = 16: 1024 threads, or W = 1, below); thread t owns elements t*NS .. t*NS+NS-1
// (dpad = 64 * W * NS, NS <= 16: model_ndim <= 16 384), so the density functors of lmc_targets.hpp -- and a user's -- run
// unchanged. It is the plain statement of the algorithm, leaf by leaf, with EVERY vector of the
// tree in the chain's HBM scratch row (L2 resident) and only the operands of the operation at hand in registers: nothing
// here depends on what fits a register file or an LDS budget. Slow next to the fused kernels (a barrier per reduction,
// two velocity evaluations per leapfrog like the reference's integration.py:111,118) -- it is the path that never
// refuses; same arithmetic statements as lmc_sampler.hpp / lmc_dense.hpp, checked against the same oracle.
//
//   leapfrog  <- /root/reference/littlemcmc/integration.py:52-66,100-121     momentum <- quadpotential.py:221-224, :374-376,
//   NUTS      <- nuts.py:204-435    HMC <- hmc.py:140-182                                :411-414, :450-453
//   adaptation <- step_sizes.py:49-99, quadpotential.py:226-245, :294-340    iteration <- base_hmc.py:140-190
#pragma once
#include "lmc_dense_types.hpp"
#include "lmc_tree_leaf.hpp"

namespace lmc {

// Two team sizes: W = 16 (1024 threads, dpad = 1024 * NS: model_ndim up to 16 384) and, for model_ndim <= 512 (a float64
// diagonal, a dense matrix beyond the fused kernels' 256 dimensions, a run-time compiled density with a dense matrix),
// W = 1 (dpad = 64 * NS, NS <= 8): the same code with wave-level reductions and no barriers, and sixteen times as many
// chains resident -- 2-6x the large team's rate at those shapes; at 16 elements per lane the large team wins
// (tools/wide_team_ab.py; lmc_wide_launch.hpp: kWideOneWaveMaxDim).
constexpr int kWideWaves = 16;
constexpr int kWideThreads = 64 * kWideWaves;
constexpr int kWideChunk = 1024;   // normals per rng_normals() call (the stream semantics do not depend on the chunking)
typedef Team<kWideWaves> WideTeam;   // the tick kernel of the wide shapes (lmc_wide.hip: TickWideShape) always uses the large team

// vectors of the chain's scratch row (dpad doubles each)
enum WideSlot : int {
    kWLq = 0, kWLp, kWLg, kWLv,        // left end of the trajectory {q, p, grad, velocity}
    kWRq, kWRp, kWRg, kWRv,            // right end
    kWPsum, kWProp,                    // running momentum sum (nuts.py:329), proposal position
    kWZ,                               // normal(size=d) of the momentum draw
    kWV0s,                             // velocity stored in the start State (float32 for the float32 potentials)
    kWTlp, kWTlv, kWTps, kWTq,         // subtree node under construction: left-end momentum / velocity, momentum sum, proposal
    kWNumFixed
};
__host__ __device__ constexpr int wide_level(int j, int k) { return kWNumFixed + 6 * j + k; }   // {lp, lv, rp, rv, psum, q}
__host__ __device__ constexpr int wide_scratch_vectors(int max_levels) { return kWNumFixed + 6 * max_levels; }
// LDS (doubles): operand / staging area, MT19937 state, team exchange, broadcast words
__host__ __device__ constexpr int wide_stage_doubles(int dpad) { return dpad > 2 * kWideChunk + 8 ? dpad : 2 * kWideChunk + 8; }
__host__ __device__ constexpr int wide_lds_doubles(int dpad) { return wide_stage_doubles(dpad) + kLdsMtDoubles + 2 * kWideWaves * kTeamSlots + 8; }

// the mass matrix as the kernel sees it
struct WideMass {
    int kind;            // 0 diagonal; 1 dense, float32 matrix; 2 dense, float64 matrix
    const void* covT;    // dense: [sweep_rows(d)][dpad], covT[j][i] = cov[i][j]
    const void* fac;     // dense: the chain's factor for the momentum draw (DenseArrays::fac; per chain for FullAdapt)
    int d, dpad;
};

template <int NS>
struct WideVec {   // the chain's scratch row
    glb_double* base;
    int dpad;
    __device__ __forceinline__ void ld(int slot, double (&x)[NS]) const { vload_as<NS>(base + static_cast<long long>(slot) * dpad, x); }
    __device__ __forceinline__ void st(int slot, const double (&x)[NS]) const { vstore_as<NS>(base + static_cast<long long>(slot) * dpad, x); }
    __device__ __forceinline__ void cp(int dst, int src) const { double t[NS]; ld(src, t); st(dst, t); }
};

// ---- sum_j M[j][i] x[j] for this thread's elements i (matrix rows contiguous over i: coalesced; operand from LDS) ----
template <int NS, class MatT, class TeamT>
__device__ __forceinline__ void wide_matvec(TeamT& tm, const MatT* M, int d, int dpad, lds_double* xop,
                                            const double (&x)[NS], double (&out)[NS]) {
    const int t = tm.tid();
    tm.sync();   // earlier readers of the operand area are done
#pragma unroll
    for (int s = 0; s < NS; ++s) xop[t * NS + s] = x[s];   // (padding elements are zero, and so are the matrix rows beyond d)
    tm.sync();
    double acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = 0.0;
    if (t * NS < d) {
        const MatT* col = M + t * NS;
        const int rows = sweep_rows(d);
        for (int jb = 0; jb < rows; jb += 8) {
            MatT m[8][NS];
#pragma unroll
            for (int b = 0; b < 8; ++b)
#pragma unroll
                for (int s = 0; s < NS; ++s) m[b][s] = col[static_cast<long long>(jb + b) * dpad + s];
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const double xj = xop[jb + b];
#pragma unroll
                for (int s = 0; s < NS; ++s) acc[s] = __builtin_fma(static_cast<double>(m[b][s]), xj, acc[s]);
            }
        }
    }
    vcopy(out, acc);
}

// velocity(x) = M^-1 x (quadpotential.py:206-208 diagonal: one rounded product per element; :446-448 / :404-409 dense)
template <int NS, class TeamT>
__device__ __forceinline__ void wide_velocity(TeamT& tm, const WideMass& M, const double (&vard)[NS], lds_double* xop,
                                              const double (&p)[NS], double (&v)[NS]) {
    if (M.kind == 0) {
#pragma unroll
        for (int s = 0; s < NS; ++s) v[s] = vard[s] * p[s];
    } else if (M.kind == 1) {
        wide_matvec<NS, float>(tm, static_cast<const float*>(M.covT), M.d, M.dpad, xop, p, v);
    } else {
        wide_matvec<NS, double>(tm, static_cast<const double*>(M.covT), M.d, M.dpad, xop, p, v);
    }
}

// integration.py:100-121. In/out: q, p, g; out: v = velocity(p'), energy, logp.
template <int NS, class Target, class TeamT>
__device__ __forceinline__ void wide_leapfrog(TeamT& tm, const Target& tgt, const WideMass& M, const double (&vard)[NS],
                                              lds_double* xop, double eps, double (&q)[NS], double (&p)[NS], double (&g)[NS],
                                              double (&v)[NS], double& energy, double& logp) {
    const double dt = 0.5 * eps;
#pragma unroll
    for (int s = 0; s < NS; ++s) p[s] = p[s] + dt * g[s];
    wide_velocity<NS>(tm, M, vard, xop, p, v);
#pragma unroll
    for (int s = 0; s < NS; ++s) q[s] = q[s] + eps * v[s];
    logp = first_f64(tgt.logp_grad(tm, q, g));
#pragma unroll
    for (int s = 0; s < NS; ++s) p[s] = p[s] + dt * g[s];
    wide_velocity<NS>(tm, M, vard, xop, p, v);
    energy = first_f64(0.5 * tm.sum(pdot<NS>(p, v)) - logp);
}

// normal(size=d) of the chain's stream into the scratch slot kWZ, kWideChunk at a time (wave 0 draws: numpy's legacy
// stream is sequential; its state is re-broadcast to the other waves)
template <int NS, class TeamT>
__device__ inline void wide_normals(TeamT& tm, RngState& r, int d, const WideVec<NS>& V, double* stage, double* bcast) {
    glb_double* z = V.base + static_cast<long long>(kWZ) * V.dpad;
    for (int off = 0; off < d; off += kWideChunk) {
        const int n = d - off < kWideChunk ? d - off : kWideChunk;
        tm.sync();
        if (tm.wave() == 0) {
            rng_normals(r, n, stage, stage + kWideChunk);
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
        for (int i = tm.tid(); i < n; i += TeamT::kThreads) z[off + i] = stage[i];
    }
    // padding of the slot: zero
    for (int i = d + tm.tid(); i < V.dpad; i += TeamT::kThreads) z[i] = 0.0;
    __threadfence_block();
    tm.sync();
}

// the same into registers (thread t receives elements t*NS .. t*NS+NS-1): the tick kernel of the wide shapes, which has no
// scratch slot to spare for the normals
template <int NS, class TeamT>
__device__ inline void wide_normals_regs(TeamT& tm, RngState& r, int d, double* stage, double* bcast, double (&z)[NS]) {
    const int t = tm.tid();
#pragma unroll
    for (int s = 0; s < NS; ++s) z[s] = 0.0;
    for (int off = 0; off < d; off += kWideChunk) {
        const int n = d - off < kWideChunk ? d - off : kWideChunk;
        tm.sync();
        if (tm.wave() == 0) {
            rng_normals(r, n, stage, stage + kWideChunk);
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
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int e = t * NS + s;
            if (e >= off && e < off + n) z[s] = stage[e - off];
        }
    }
    tm.sync();
}

// solve_triangular(chol.T, float32(z)) (quadpotential.py:450-453): the column sweep of the reference BLAS strsv over the
// row-major float32 factor (x_j /= L_jj, then x_i -= L_ji x_j for i < j, j descending), one team barrier per column
// (T = double: QuadPotentialFullAdapt(dtype="float64") -- the same sweep in float64 on a float64 factor)
template <int NS, class T, class TeamT>
__device__ inline void wide_momentum_strsv(TeamT& tm, const T* L, int d, int dpad, const double (&z)[NS], double* bc,
                                           double (&p0)[NS]) {
    const int t = tm.tid();
    T x[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) x[s] = (t * NS + s < d) ? static_cast<T>(z[s]) : T(0);
    T* bcf = reinterpret_cast<T*>(bc);   // two broadcast slots, used alternately
    for (int j = d - 1; j >= 0; --j) {
        const int owner = j / NS, sj = j % NS;
        if (t == owner) {
            T xs = T(0);
#pragma unroll
            for (int s = 0; s < NS; ++s) if (s == sj) xs = x[s];
            const T xj = xs / L[static_cast<long long>(j) * dpad + j];
#pragma unroll
            for (int s = 0; s < NS; ++s) if (s == sj) x[s] = xj;
            bcf[j & 1] = xj;
        }
        tm.sync();
        const T xj = bcf[j & 1];
        const T* row = L + static_cast<long long>(j) * dpad + t * NS;
#pragma unroll
        for (int s = 0; s < NS; ++s)
            if (t * NS + s < j) x[s] = x[s] - row[s] * xj;
    }
    tm.sync();
#pragma unroll
    for (int s = 0; s < NS; ++s) p0[s] = static_cast<double>(x[s]);
}

// mass-matrix state of the wide kernels: the diagonal in float64 registers whatever its dtype (a float32 value is
// exact in a double), the dtype deciding where results are rounded
struct WideDiagDtype { bool f32; };

// QuadPotentialDiagAdapt.update (quadpotential.py:231-245, :324-340) with the mass in `dtype`
template <int NS, class CA, class PT>
__device__ __forceinline__ void wide_diag_update(const CA& A, const PT& P, long long row, int tid, bool mass_f32,
                                                 const double (&q)[NS], double (&vard)[NS], double (&invd)[NS], MassScalars& ms) {
    const int d = A.d;
    const long long plane = static_cast<long long>(A.chains) * A.dpad;
    double m[NS], r[NS], mb[NS], rb[NS];
    double* fm = A.wmean + ms.wsel * plane + row;
    double* fr = A.wraw + ms.wsel * plane + row;
    double* bm = A.wmean + (1 - ms.wsel) * plane + row;
    double* br = A.wraw + (1 - ms.wsel) * plane + row;
    vload<NS>(fm, m); vload<NS>(fr, r); vload<NS>(bm, mb); vload<NS>(br, rb);
    ms.wsum_f += 1.0;
    ms.wsum_b += 1.0;
    const double prop_f = first_f64(1.0 / ms.wsum_f), prop_b = first_f64(1.0 / ms.wsum_b);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const double od = q[s] - m[s];
        m[s] = m[s] + prop_f * od;
        const double nd = q[s] - m[s];
        r[s] = r[s] + 1.0 * od * nd;
        if (tid * NS + s < d) {
            if (mass_f32) {
                const float vf = static_cast<float>(r[s] / ms.wsum_f);
                const float sd = sqrtf(vf);
                vard[s] = static_cast<double>(vf);
                invd[s] = static_cast<double>(1.0f / sd);
            } else {
                vard[s] = r[s] / ms.wsum_f;
                invd[s] = 1.0 / sqrt(vard[s]);
            }
        }
        const double odb = q[s] - mb[s];
        mb[s] = mb[s] + prop_b * odb;
        const double ndb = q[s] - mb[s];
        rb[s] = rb[s] + 1.0 * odb * ndb;
    }
    if (ms.n_samples > 0 && ms.n_samples % ms.window == 0) {   // background becomes foreground
        vstore<NS>(bm, mb); vstore<NS>(br, rb);
#pragma unroll
        for (int s = 0; s < NS; ++s) { m[s] = 0.0; r[s] = 0.0; }
        vstore<NS>(fm, m); vstore<NS>(fr, r);
        ms.wsum_f = ms.wsum_b;
        ms.wsum_b = 0.0;
        ms.wsel = 1 - ms.wsel;
        ms.window = static_cast<int>(static_cast<double>(ms.window) * P.window_multiplier);
    } else {
        vstore<NS>(fm, m); vstore<NS>(fr, r); vstore<NS>(bm, mb); vstore<NS>(br, rb);
    }
    ++ms.n_samples;
}

// float32 kinetic energy of the start state, 0.5f * sdot(p, v) in the host BLAS's order (see start_kinetic_f32)
template <int NS, class TeamT>
__device__ inline float wide_start_kinetic_f32(TeamT& tm, const float (&pf)[NS], const float (&vf)[NS], int d, int mode,
                                               float* scratch, int dpad) {
    if (mode == kSdotNative) {
        double part = 0.0;
#pragma unroll
        for (int s = 0; s < NS; ++s) part = __builtin_fma(static_cast<double>(pf[s]), static_cast<double>(vf[s]), part);
        return 0.5f * static_cast<float>(tm.sum(part));
    }
    const int t = tm.tid();
    tm.sync();
#pragma unroll
    for (int s = 0; s < NS; ++s) { scratch[t * NS + s] = pf[s]; scratch[dpad + t * NS + s] = vf[s]; }
    tm.sync();
    const float dot = sdot_openblas(scratch, scratch + dpad, d, mode);
    tm.sync();
    return 0.5f * dot;
}

struct WideCtx {   // what every stage of an iteration needs
    lds_double* xop;      // operand / staging area
    double* stage;        // the same area as a generic pointer
    double* bcast;        // 8 doubles
};

// ---- NUTS / HMC transitions: lmc_tree_leaf.hpp's leaf form / lmc_sampler.hpp's hmc_transition_any with this policy --------
// EVERY vector lives in the chain's scratch row (WideSlot): the trajectory's ends (their velocity slots hold the velocity
// stored with the State -- the start velocity until the end is replaced), the node under construction, the stack, the running
// momentum sum and the proposal; only the state being integrated and the operands of the statement at hand are in registers.
// wide_start() has put the start state at both ends, kWPsum = p0 and kWProp = q; the proposal / accepted position lands in kWProp.
template <int NS, class Target, class TeamT>
struct WideTreePolicy {
    static constexpr int kNS = NS;
    struct End { double q[NS], p[NS], g[NS], v[NS]; };
    TeamT& tm; const Target& tgt; const WideMass& M; const double (&vard)[NS]; const WideCtx& cx; RngState& rng; const WideVec<NS>& V;
    UniformWindow win;

    __device__ __forceinline__ double uniform() { return team_uniform(tm, rng, win); }
    __device__ __forceinline__ bool any_nonpositive2(double a, double b) { return tm.any_nonpositive2(a, b); }
    __device__ __forceinline__ bool any_nonpositive6(double (&d)[6]) { return tm.any_nonpositive6(d); }
    __device__ __forceinline__ void start_state(End& c) const { V.ld(kWRq, c.q); V.ld(kWRp, c.p); V.ld(kWRg, c.g); }
    __device__ __forceinline__ void accept_state(const End& c) { V.st(kWProp, c.q); }
    __device__ __forceinline__ void end_load(int side, End& c) const {   // (the velocity is leapfrog's output: not loaded)
        const int e = side ? kWRq : kWLq;
        V.ld(e, c.q); V.ld(e + 1, c.p); V.ld(e + 2, c.g);
    }
    __device__ __forceinline__ void end_store(int side, const End& c) {
        const int e = side ? kWRq : kWLq;
        V.st(e, c.q); V.st(e + 1, c.p); V.st(e + 2, c.g); V.st(e + 3, c.v);
    }
    __device__ __forceinline__ void end_velocity(int side, double (&v)[NS]) const { V.ld(side ? kWRv : kWLv, v); }
    __device__ __forceinline__ void end_momentum(int side, double (&p)[NS]) const { V.ld(side ? kWRp : kWLp, p); }
    __device__ __forceinline__ void leapfrog(double eps, End& c, double& energy, double& logp) {
        wide_leapfrog<NS>(tm, tgt, M, vard, cx.xop, eps, c.q, c.p, c.g, c.v, energy, logp);
    }
    template <int F> static __device__ __forceinline__ constexpr int node_slot() {
        return F == kNodeLp ? kWTlp : F == kNodeLv ? kWTlv : F == kNodePs ? kWTps : kWTq;
    }
    template <int F> __device__ __forceinline__ void node_ld(double (&x)[NS]) const { V.ld(node_slot<F>(), x); }
    template <int F> __device__ __forceinline__ void node_st(const double (&x)[NS]) { V.st(node_slot<F>(), x); }
    __device__ __forceinline__ void level_ld(int j, int f, double (&x)[NS]) const { V.ld(wide_level(j, f), x); }
    __device__ __forceinline__ void level_st(int j, int f, const double (&x)[NS]) { V.st(wide_level(j, f), x); }
    __device__ __forceinline__ void psum_ld(double (&x)[NS]) const { V.ld(kWPsum, x); }
    __device__ __forceinline__ void psum_st(const double (&x)[NS]) { V.st(kWPsum, x); }
    __device__ __forceinline__ void proposal_from_node() { V.cp(kWProp, kWTq); }
};

// the start of an iteration (base_hmc.py:141-148 / integration.py:52-66): momentum draw, start state -> both ends of the
// trajectory, kWPsum, kWProp. Returns e0 (non-finite: base_hmc.py:145-148).
template <int NS, class Target, class TeamT>
__device__ inline double wide_start(TeamT& tm, const Target& tgt, const WideMass& M, const DenseArrays& D, const double (&vard)[NS],
                                    const double (&invd)[NS], const WideCtx& cx, RngState& rng, const WideVec<NS>& V, int d,
                                    bool momentum_f32, int sdot_mode, const double (&q)[NS], double& logp0) {
    const int t = tm.tid();
    wide_normals<NS>(tm, rng, d, V, cx.stage, cx.bcast);
    double z[NS], p0[NS];
    V.ld(kWZ, z);
    if (M.kind == 0) {   // quadpotential.py:221-224 (dtype of the potential) / :374-376
#pragma unroll
        for (int s = 0; s < NS; ++s)
            p0[s] = momentum_f32 ? static_cast<double>(static_cast<float>(invd[s]) * static_cast<float>(z[s])) : z[s] * invd[s];
    } else if (D.kind == kDenseFullInv) {   // L n (FullInv) / solve_triangular(chol.T, n) as the sweep of L^-1 (Full float64)
        wide_matvec<NS, double>(tm, static_cast<const double*>(M.fac), d, V.dpad, cx.xop, z, p0);
    } else if (D.mat_f64) {                 // FullAdapt(dtype="float64"): the factor changes on the device, so the solve runs here
        wide_momentum_strsv<NS, double>(tm, static_cast<const double*>(M.fac), d, V.dpad, z, cx.bcast + 4, p0);
    } else {
        wide_momentum_strsv<NS, float>(tm, static_cast<const float*>(M.fac), d, V.dpad, z, cx.bcast + 4, p0);
    }
    double g0[NS], v0[NS], v0s[NS];
    logp0 = first_f64(tgt.logp_grad(tm, q, g0));
    wide_velocity<NS>(tm, M, vard, cx.xop, p0, v0);
    double e0;
    if (momentum_f32) {   // float32 velocity and kinetic energy of the start State
        float pf[NS], vf[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            pf[s] = static_cast<float>(p0[s]);
            vf[s] = (M.kind == 0) ? static_cast<float>(vard[s]) * pf[s] : static_cast<float>(v0[s]);
            v0s[s] = static_cast<double>(vf[s]);
        }
        const float kin = wide_start_kinetic_f32<NS>(tm, pf, vf, d, sdot_mode, reinterpret_cast<float*>(cx.stage), V.dpad);
        e0 = first_f64(static_cast<double>(kin) - logp0);
    } else {
        vcopy(v0s, v0);
        e0 = first_f64(0.5 * tm.sum(pdot<NS>(p0, v0)) - logp0);
    }
    (void)t;
    V.st(kWLq, q); V.st(kWLp, p0); V.st(kWLg, g0); V.st(kWLv, v0s);
    V.st(kWRq, q); V.st(kWRp, p0); V.st(kWRg, g0); V.st(kWRv, v0s);
    V.st(kWPsum, p0); V.st(kWProp, q);
    return e0;
}

// ---- the iteration kernel -------------------------------------------------------------------------------------------
template <int NS, int W, template <int> class TargetT>
__global__ __launch_bounds__(64 * W) void run_wide_kernel(ChainArrays A, DenseArrays D, SamplerParams P, const double* tparams) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int c = blockIdx.x + P.chain_begin;
    const int d = A.d, dpad = A.dpad;
    const long long row = static_cast<long long>(c) * dpad;
    Team<W> tm;
    const int stage_doubles = wide_stage_doubles(dpad);
    uint32_t* mt_lds = reinterpret_cast<uint32_t*>(lds + stage_doubles);
    tm.xbuf = lds + stage_doubles + kLdsMtDoubles;
    tm.parity = 0;
    double* bcast = tm.xbuf + 2 * kWideWaves * kTeamSlots;
    const int tid = tm.tid();
    if (stop_at_entry<W>(A.stop_dev, reinterpret_cast<int*>(bcast))) return;
    if (A.status[c] & kStatusBadInitialEnergy) return;

    TargetT<NS> tgt;
    tgt.init(tm, tparams, d);
    WideMass M;
    M.kind = D.covT == nullptr ? 0 : (D.mat_f64 ? 2 : 1);
    M.covT = D.covT == nullptr ? nullptr : static_cast<const char*>(D.covT) + static_cast<long long>(c) * D.mat_stride * (M.kind == 2 ? 8 : 4);
    M.fac = D.covT == nullptr ? nullptr : static_cast<const char*>(D.fac) + static_cast<long long>(c) * D.fac_stride * (M.kind == 2 ? 8 : 4);
    M.d = d; M.dpad = dpad;
    WideVec<NS> V{(glb_double*)(A.scratch + static_cast<long long>(c) * A.scratch_stride), dpad};
    WideCtx cx{(lds_double*)lds, lds, bcast};

    double q[NS], vard[NS], invd[NS];
    vload<NS>(A.q + row, q);
    vload<NS>(A.var64 + row, vard);
    vload<NS>(A.inv_std64 + row, invd);
    RngState rng;
    uint32_t* mt_glb = A.mt + static_cast<long long>(c) * kMtN;
    for (int i = tid; i < kMtN; i += 64 * W) mt_lds[i] = mt_glb[i];
    tm.sync();
    rng.mt = mt_lds;
    rng.pos = first_i32(A.rng_pos[c]);
    rng.has_gauss = first_i32(A.rng_has_gauss[c]);
    rng.gauss = first_f64(A.rng_gauss[c]);
    DualAverage da;
    dual_average_load(A, c, da);
    int iter_count = first_i32(A.iter_count[c]);
    MassScalars ms;
    ms.n_samples = first_i32(A.n_samples[c]);
    ms.wsel = first_i32(A.wsel[c]);
    ms.wsum_f = first_f64(A.wsum[c * 2 + ms.wsel]);
    ms.wsum_b = first_f64(A.wsum[c * 2 + (1 - ms.wsel)]);
    ms.window = first_i32(A.awindow[c]);
    long long ct_maxdepth = 0, ct_divs = 0, ct_after = 0, ct_leap = 0;
    int status = 0;
    const bool momentum_f32 = P.momentum_f32 != 0;
    const bool mass_f32 = P.mass_f64 == 0;

    for (int it = 0; it < P.n_iters; ++it) {
        const long long git = P.iter_begin + it;
        const bool tune = git < P.n_tune;
        const int stop_word = stop_request_load(A, P, static_cast<int>(blockIdx.x), it, git);
        double logp0;
        const double e0 = wide_start<NS>(tm, tgt, M, D, vard, invd, cx, rng, V, d, momentum_f32, P.sdot_mode, q, logp0);
        if (!isfinite(e0)) {   // base_hmc.py:145-148
            status |= kStatusBadInitialEnergy;
            break;
        }
        const bool adapt_step = tune && P.adapt_step_size;
        const double step_size = jitter_step_size(tm, rng, A, P, c, adapt_step ? da.step_now : da.step_bar_now);
        TransitionOut out;
        WideTreePolicy<NS, TargetT<NS>, Team<W>> pol{tm, tgt, M, vard, cx, rng, V, UniformWindow{0.0, 0, 0}};
        if (P.kind == 0) {
            const int md = (tune && iter_count < 200) ? P.early_max_treedepth : P.max_treedepth;
            leaf_nuts_transition(pol, e0, logp0, step_size, P.emax, md, momentum_f32, out);   // lmc_tree_leaf.hpp
            if (out.exhausted && !tune) ++ct_maxdepth;
        } else {
            hmc_transition_any(pol, e0, logp0, step_size, P.emax, P.path_length, P.max_steps, out);   // lmc_sampler.hpp
        }
        __threadfence_block();
        tm.sync();
        V.ld(kWProp, q);
        ct_leap += out.n_leapfrog;
        if (adapt_step) dual_average_update(A, P, out.accept, da);
        if (tune && P.adapt_mass) wide_diag_update<NS>(A, P, row, tid, mass_f32, q, vard, invd, ms);
        if (out.diverging && !tune) ++ct_divs;
        ++iter_count;
        if (!tune) ++ct_after;
        if (A.mom_mean != nullptr && !tune) moments_update<NS>(A, tm, c, row, q);
        write_outputs<NS>(A, c, tid, git, q, out, da.step_now, da.step_bar_now, tune);
        if (stop_requested(tm, stop_word, bcast)) break;
    }

    tm.sync();
    for (int i = tid; i < kMtN; i += 64 * W) mt_glb[i] = rng.mt[i];
    vstore<NS>(A.q + row, q);
    vstore<NS>(A.var64 + row, vard);
    vstore<NS>(A.inv_std64 + row, invd);
#pragma unroll
    for (int s = 0; s < NS; ++s) {   // the float32 views the state getters hand out
        A.var[row + tid * NS + s] = static_cast<float>(vard[s]);
        A.inv_std[row + tid * NS + s] = static_cast<float>(invd[s]);
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
    }
}

// ---- unit entry points of the wide shapes (lmc_engine_logp_dlogp / _trajectory / _draw_momentum / _diag_update) ----------
template <int NS, int W, template <int> class TargetT>
__global__ __launch_bounds__(64 * W) void wide_logp_kernel(ChainArrays A, const double* tparams, const double* qin,
                                                                     double* logp_out, double* grad_out) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int c = blockIdx.x;
    const int d = A.d;
    Team<W> tm;
    tm.xbuf = lds;
    tm.parity = 0;
    const int t = tm.tid();
    TargetT<NS> tgt;
    tgt.init(tm, tparams, d);
    double q[NS], g[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int e = t * NS + s;
        q[s] = (e < d) ? qin[static_cast<long long>(c) * d + e] : 0.0;
    }
    const double logp = tgt.logp_grad(tm, q, g);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int e = t * NS + s;
        if (e < d) grad_out[static_cast<long long>(c) * d + e] = g[s];
    }
    if (t == 0) logp_out[c] = logp;
}

// compute_state + n_fwd steps (+eps) + n_back steps (-eps); all states written out (integration.py:52-121)
template <int NS, int W, template <int> class TargetT>
__global__ __launch_bounds__(64 * W) void wide_trajectory_kernel(ChainArrays A, DenseArrays D, const double* tparams,
                                                                           const double* q0, const double* p0in, int p0_is_f32,
                                                                           int sdot_mode, double eps, int n_fwd, int n_back,
                                                                           double* oq, double* op, double* ov, double* og,
                                                                           double* oe, double* ol) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int c = blockIdx.x;
    const int d = A.d, dpad = A.dpad;
    const long long row = static_cast<long long>(c) * dpad;
    Team<W> tm;
    const int stage_doubles = wide_stage_doubles(dpad);
    tm.xbuf = lds + stage_doubles;
    tm.parity = 0;
    const int t = tm.tid();
    TargetT<NS> tgt;
    tgt.init(tm, tparams, d);
    WideMass M;
    M.kind = D.covT == nullptr ? 0 : (D.mat_f64 ? 2 : 1);
    M.covT = D.covT == nullptr ? nullptr : static_cast<const char*>(D.covT) + static_cast<long long>(c) * D.mat_stride * (M.kind == 2 ? 8 : 4);
    M.fac = D.covT == nullptr ? nullptr : static_cast<const char*>(D.fac) + static_cast<long long>(c) * D.fac_stride * (M.kind == 2 ? 8 : 4);
    M.d = d; M.dpad = dpad;
    double q[NS], p[NS], g[NS], v[NS], vard[NS];
    vload<NS>(A.var64 + row, vard);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int e = t * NS + s;
        q[s] = (e < d) ? q0[static_cast<long long>(c) * d + e] : 0.0;
        p[s] = (e < d) ? p0in[static_cast<long long>(c) * d + e] : 0.0;
        if (p0_is_f32) p[s] = static_cast<double>(static_cast<float>(p[s]));
    }
    const int n_states = n_fwd + n_back + 1;
    double logp = first_f64(tgt.logp_grad(tm, q, g));
    double energy;
    wide_velocity<NS>(tm, M, vard, (lds_double*)lds, p, v);
    if (p0_is_f32) {
        float pf[NS], vf[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            pf[s] = static_cast<float>(p[s]);
            vf[s] = (M.kind == 0) ? static_cast<float>(vard[s]) * pf[s] : static_cast<float>(v[s]);
            v[s] = static_cast<double>(vf[s]);
        }
        const float kin = wide_start_kinetic_f32<NS>(tm, pf, vf, d, sdot_mode, reinterpret_cast<float*>(lds), dpad);
        energy = static_cast<double>(kin) - logp;
    } else {
        energy = 0.5 * tm.sum(pdot<NS>(p, v)) - logp;
    }
    for (int k = 0; k < n_states; ++k) {
        if (k > 0) wide_leapfrog<NS>(tm, tgt, M, vard, (lds_double*)lds, (k <= n_fwd) ? eps : -eps, q, p, g, v, energy, logp);
        const long long base = (static_cast<long long>(c) * n_states + k) * d;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int e = t * NS + s;
            if (e < d) { oq[base + e] = q[s]; op[base + e] = p[s]; ov[base + e] = v[s]; og[base + e] = g[s]; }
        }
        if (t == 0) {
            oe[static_cast<long long>(c) * n_states + k] = energy;
            ol[static_cast<long long>(c) * n_states + k] = logp;
        }
    }
}

// potential.random() for every chain (quadpotential.py:221-224 / :374-376 / :411-414 / :450-453)
template <int NS, int W>
__global__ __launch_bounds__(64 * W) void wide_momentum_kernel(ChainArrays A, DenseArrays D, int momentum_f32, double* out) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int c = blockIdx.x;
    const int d = A.d, dpad = A.dpad;
    const long long row = static_cast<long long>(c) * dpad;
    Team<W> tm;
    const int stage_doubles = wide_stage_doubles(dpad);
    tm.xbuf = lds + stage_doubles;
    tm.parity = 0;
    double* bcast = tm.xbuf + 2 * kWideWaves * kTeamSlots;
    const int t = tm.tid();
    RngState r;
    r.mt = A.mt + static_cast<long long>(c) * kMtN;   // in place (a unit entry point)
    r.pos = first_i32(A.rng_pos[c]);
    r.has_gauss = first_i32(A.rng_has_gauss[c]);
    r.gauss = first_f64(A.rng_gauss[c]);
    WideVec<NS> V{(glb_double*)(A.scratch + static_cast<long long>(c) * A.scratch_stride), dpad};
    wide_normals<NS>(tm, r, d, V, lds, bcast);
    double z[NS], p0[NS], invd[NS];
    V.ld(kWZ, z);
    vload<NS>(A.inv_std64 + row, invd);
    if (D.covT == nullptr) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
            p0[s] = momentum_f32 ? static_cast<double>(static_cast<float>(invd[s]) * static_cast<float>(z[s])) : z[s] * invd[s];
    } else if (D.kind == kDenseFullInv) {
        wide_matvec<NS, double>(tm, static_cast<const double*>(D.fac), d, dpad, (lds_double*)lds, z, p0);
    } else if (D.mat_f64) {
        wide_momentum_strsv<NS, double>(tm, static_cast<const double*>(D.fac) + static_cast<long long>(c) * D.fac_stride, d, dpad, z, bcast + 4, p0);
    } else {
        wide_momentum_strsv<NS, float>(tm, static_cast<const float*>(D.fac) + static_cast<long long>(c) * D.fac_stride, d, dpad, z, bcast + 4, p0);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int e = t * NS + s;
        if (e < d) out[static_cast<long long>(c) * d + e] = p0[s];
    }
    if (t == 0) {
        A.rng_pos[c] = r.pos;
        A.rng_has_gauss[c] = r.has_gauss;
        A.rng_gauss[c] = r.gauss;
    }
}

// QuadPotentialDiagAdapt.update(sample = current position, grad, tune = True) for every chain
template <int NS, int W>
__global__ __launch_bounds__(64 * W) void wide_mass_update_kernel(ChainArrays A, SamplerParams P) {
    const int c = blockIdx.x;
    const int tid = static_cast<int>(threadIdx.x);
    const long long row = static_cast<long long>(c) * A.dpad;
    double q[NS], vard[NS], invd[NS];
    vload<NS>(A.q + row, q);
    vload<NS>(A.var64 + row, vard);
    vload<NS>(A.inv_std64 + row, invd);
    MassScalars ms;
    ms.n_samples = A.n_samples[c];
    ms.wsel = A.wsel[c];
    ms.wsum_f = A.wsum[c * 2 + ms.wsel];
    ms.wsum_b = A.wsum[c * 2 + (1 - ms.wsel)];
    ms.window = A.awindow[c];
    __syncthreads();   // every thread has read the scalars thread 0 rewrites below
    wide_diag_update<NS>(A, P, row, tid, P.mass_f64 == 0, q, vard, invd, ms);
    vstore<NS>(A.var64 + row, vard);
    vstore<NS>(A.inv_std64 + row, invd);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        A.var[row + tid * NS + s] = static_cast<float>(vard[s]);
        A.inv_std[row + tid * NS + s] = static_cast<float>(invd[s]);
    }
    if (tid == 0) {
        A.n_samples[c] = ms.n_samples;
        A.wsel[c] = ms.wsel;
        A.awindow[c] = ms.window;
        A.wsum[c * 2 + ms.wsel] = ms.wsum_f;
        A.wsum[c * 2 + (1 - ms.wsel)] = ms.wsum_b;
    }
}

}  // namespace lmc
