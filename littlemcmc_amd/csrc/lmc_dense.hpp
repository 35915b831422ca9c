// Dense mass matrices on the device: QuadPotentialFull / FullInv / FullAdapt.
//
//   velocity / energy   <- /root/reference/littlemcmc/quadpotential.py:446-464 (Full), :404-425 (FullInv)
//   momentum draw       <- quadpotential.py:450-453 (float32 triangular solve), :411-414 (float64 L n)
//   covariance estimate <- quadpotential.py:521-552 (FullAdapt.update), :560-615 (_WeightedCovariance)
//   leapfrog            <- integration.py:52-66, :100-121
//   NUTS / HMC          <- nuts.py:204-435, hmc.py:140-182 (same iterative tree as lmc_sampler.hpp)
//
// What changes against the diagonal kernel (lmc_sampler.hpp):
//  * A velocity is a d x d matrix-vector product, so velocities are STORED with every tree node instead of being
//    recomputed, and the kernel is bound by streaming the matrix, not by VALU issue. One chain = one wavefront;
//    thread t owns elements t*NS .. t*NS+NS-1 of every vector.
//  * Matrix layout: M^-1 is kept TRANSPOSED, covT[j][i] = cov[i][j], rows padded to dpad. Row j is contiguous
//    over i, so   v_i = sum_j cov[i][j] p_j   is a sweep over rows with lane-contiguous (coalesced) loads and the
//    operand p_j broadcast from LDS; no cross-lane reduction. float32 storage (as the reference), float64 products.
//  * ONE sweep per leapfrog instead of the reference's two: the sweep at the end of a step forms both
//    v' = C p' and w' = C g'; the next half step uses C (p' + dt g') = v' + dt w'. Same value up to the rounding
//    of one float64 addition; halves the bytes that bound the kernel.
//  * Tree nodes and both trajectory ends live in the chain's HBM scratch row (L2 resident while the chain is hot):
//    next to a 64 KiB matrix sweep per leapfrog their traffic is noise, and the register file stays free for
//    the sweep's loads in flight.
//  * FullAdapt's per-iteration covariance refresh + Cholesky is a separate kernel (one workgroup per chain, matrix
//    in LDS) launched between iterations while tuning; see dense_adapt_kernel.
#pragma once
#include "lmc_dense_types.hpp"
#include "lmc_tree_leaf.hpp"

namespace lmc {

// N matrix entries of one thread as one vector register group: loads from HBM become global_load_dwordx{N}
template <class T, int N> struct PackOf { typedef T type __attribute__((ext_vector_type(N))); };
template <class T> struct PackOf<T, 1> { typedef T type; };
template <class T, int N> using Pack = typename PackOf<T, N>::type;
template <int N, class P> __device__ __forceinline__ auto pack_get(const P& p, int s) {
    if constexpr (N == 1) return p; else return p[s];
}

// ---- the matrix as a wave sees it -----------------------------------------------------------------------------
// The leading `cache_rows` rows of the chain's matrix are copied to LDS once per launch (the matrix is constant
// while a launch runs; FullAdapt refreshes it between launches): that share of every sweep costs no HBM / L2
// traffic, and the loads of the remaining rows are already in flight while the cached rows are consumed. Each
// lane copies and later reads only its own columns, so the copy needs no barrier.
// A matrix shared by all chains, met by eight chains at a time (run_dense_coop_kernel below): the whole transposed
// matrix sits in the workgroup's LDS, the chains' operand vectors and the products travel through two LDS panels.
struct DenseCoop {
    __attribute__((address_space(3))) const float* ct;    // [k_rows][ct_stride] transposed matrix, float32 (row k over i)
    lds_double* x;                                        // [16][xs]: column 2w = first operand of wave w's chain, 2w+1 = second
    lds_double* dout;                                     // [16][xs]: the products, same columns
    __attribute__((address_space(3))) int* n_active;      // chains of this workgroup still sampling
    int ct_stride, xs, k_rows, dpad, wave, n_waves;
#ifdef LMC_COOP_TIMING   // diagnostic build (tools/coop_timing.py): clock ticks waiting / multiplying / on the chain's own work
    unsigned long long* tk;
#endif
};

template <class MatT>
struct DenseMat {
    const MatT* glb;                                            // [sweep_rows(d)][dpad]
    __attribute__((address_space(3))) const MatT* cache;        // [cache_rows][dpad] in LDS
    int cache_rows;                                             // multiple of kSweepBatch, <= sweep_rows(d) - 2 * kSweepBatch or == sweep_rows(d)
    int d, dpad;
    const DenseCoop* coop;                                      // != nullptr only in the coop kernel (LMC_DENSE_COOP builds)
};

// ---- one sweep over the transposed matrix: acc[v][s] = sum_j M[j][lane*NS+s] * x[j][v] --------------------------
// x: LDS, NV operands interleaved per row index. float64 fused multiply-adds on the promoted matrix entries
// (numpy promotes the float32 matrix and calls dgemv; only the summation order differs).
template <int NS, int NV, class MatT>
__device__ __forceinline__ void dense_sweep(const DenseMat<MatT>& mm, const lds_double* x, double (&acc)[NV][NS]) {
    typedef __attribute__((address_space(1))) const Pack<MatT, NS> glb_pack;
    typedef __attribute__((address_space(3))) const Pack<MatT, NS> lds_pack;
    glb_pack* col = (glb_pack*)(mm.glb + lane_id() * NS);
    lds_pack* ccol = (lds_pack*)(mm.cache + lane_id() * NS);
    const int stride = mm.dpad / NS;   // in packs
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int s = 0; s < NS; ++s) acc[v][s] = 0.0;
    // Software pipeline, two batches of kSweepBatch rows: the loads of one batch are in flight while the other is
    // consumed, so a wave always has 8..16 row loads outstanding (the sweep is latency / bandwidth bound, not ALU
    // bound). The matrix has sweep_rows(d) rows, the extra ones zero, and the operand area is zero beyond d.
    constexpr int B = kSweepBatch;
    const int rows = sweep_rows(mm.d);
    const int first = mm.cache_rows;
    Pack<MatT, NS> m0[B], m1[B];
    auto fetch = [&](Pack<MatT, NS> (&m)[B], int jb) {
#pragma unroll
        for (int b = 0; b < B; ++b) m[b] = col[static_cast<long long>(jb + b) * stride];
    };
    auto consume = [&](const Pack<MatT, NS> (&m)[B], int jb) {
#pragma unroll
        for (int b = 0; b < B; ++b)
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const double xj = x[(jb + b) * NV + v];
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    acc[v][s] = __builtin_fma(static_cast<double>(pack_get<NS>(m[b], s)), xj, acc[v][s]);
            }
    };
    if (first < rows) fetch(m0, first);
    for (int jb = 0; jb < first; jb += B) {   // LDS-resident rows
        Pack<MatT, NS> c[B];
#pragma unroll
        for (int b = 0; b < B; ++b) c[b] = ccol[(jb + b) * stride];
        consume(c, jb);
    }
    for (int jb = first; jb < rows; jb += 2 * B) {
        fetch(m1, jb + B);
        consume(m0, jb);
        if (jb + 2 * B < rows) fetch(m0, jb + 2 * B);
        consume(m1, jb + B);
    }
}

// stage one or two operand vectors for dense_sweep (thread t writes its own elements, every lane reads all)
template <int NS>
__device__ __forceinline__ void stage2(lds_double* x, const double (&a)[NS], const double (&b)[NS]) {
    wave_sync();   // earlier readers of the area are done
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int e = lane_id() * NS + s;
        x[2 * e] = a[s];
        x[2 * e + 1] = b[s];
    }
    wave_sync();
}

#ifdef LMC_DENSE_COOP
// C [p g] for the EIGHT chains of a workgroup at once, on the matrix cores (quadpotential.py:446-464 for every chain of
// the group). Every chain needs exactly one such product per leapfrog (and one for its start state) wherever it is in its
// tree, so the chains' wavefronts meet here: each drops its two vectors into columns of the operand panel, one barrier,
// wave w forms rows [16 t, 16 t + 16) (t = w, w + 8, ...) of  C (dpad x k_rows) x X (k_rows x 16)  with
// v_mfma_f64_16x16x4_f64 -- the float32 entries promoted on the way in, float64 accumulation, as numpy's dgemv on the
// promoted matrix -- writes them to the product panel, one more barrier, each wave picks its two columns up.
// Operand / result layout of the instruction (guides/cdna_hip_programming.md): A[i][k] on lane i + 16 k, B[k][j] on lane
// j + 16 k, D[row = (lane >> 4) + 4 r][col = lane & 15] in result register r. Row stride of the matrix dpad + 16 floats and
// column stride of the panels dpad + 2 doubles keep the four k-groups of a wave on different LDS banks.
// Returns the number of chains of the workgroup still sampling (read between the two barriers: the same in every wave).
template <int NS>
__device__ __forceinline__ int coop_product(const DenseCoop& cc, const double (&p)[NS], const double (&g)[NS],
                                            double (&v)[NS], double (&w)[NS]) {
    typedef double v4d __attribute__((ext_vector_type(4)));
    const int lane = lane_id();
    lds_double* xp = cc.x + (2 * cc.wave) * cc.xs + lane * NS;
#pragma unroll
    for (int s = 0; s < NS; ++s) { xp[s] = p[s]; xp[cc.xs + s] = g[s]; }
#ifdef LMC_COOP_TIMING
    const unsigned long long t_arrive = clock64();
    cc.tk[3] += t_arrive - cc.tk[4];
#endif
    __syncthreads();
#ifdef LMC_COOP_TIMING
    const unsigned long long t_go = clock64();
    cc.tk[0] += t_go - t_arrive;
#endif
    const int active = first_i32(*cc.n_active);
    const int kk = lane >> 4, jj = lane & 15;
    for (int t = cc.wave; t < cc.dpad / 16; t += cc.n_waves) {
        v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};   // two accumulators: back-to-back issues do not wait on each other
        __attribute__((address_space(3))) const float* a = cc.ct + kk * cc.ct_stride + 16 * t + jj;
        const lds_double* b = cc.x + jj * cc.xs + kk;
        const int nkb = cc.k_rows / 4;   // a multiple of 4: k_rows is a multiple of 16
        // (requesting the next operands before issuing the current products -- a software pipeline over batches of 4 or 8
        // k-blocks -- was measured and is SLOWER: 2.46e8 / 2.22e8 against 2.66e8 leapfrog-steps/s. The product phase is bound by
        // the FP64 matrix pipe itself -- 32 products x 64 cycles per wave, two waves per SIMD -- and the extra live operands
        // cost spills: DESIGN.md section 9)
        for (int kb = 0; kb < nkb; kb += 4) {   // the four tiles' operands are requested before the first product is issued
            float af[4];
            double bf[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { af[u] = a[(4 * (kb + u)) * cc.ct_stride]; bf[u] = b[4 * (kb + u)]; }
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(static_cast<double>(af[0]), bf[0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(static_cast<double>(af[1]), bf[1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(static_cast<double>(af[2]), bf[2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(static_cast<double>(af[3]), bf[3], acc1, 0, 0, 0);
        }
        lds_double* o = cc.dout + jj * cc.xs + 16 * t + kk;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[4 * r] = acc0[r] + acc1[r];
    }
#ifdef LMC_COOP_TIMING
    const unsigned long long t_done = clock64();
    cc.tk[1] += t_done - t_go;
#endif
    __syncthreads();
    const lds_double* dp = cc.dout + (2 * cc.wave) * cc.xs + lane * NS;
#pragma unroll
    for (int s = 0; s < NS; ++s) { v[s] = dp[s]; w[s] = dp[cc.xs + s]; }
#ifdef LMC_COOP_TIMING
    cc.tk[4] = clock64();
    cc.tk[2] += cc.tk[4] - t_done;
#endif
    return active;
}
#endif

// v = C p and w = C g in one sweep
template <int NS, class MatT>
__device__ __forceinline__ void velocity2(const DenseMat<MatT>& mm, lds_double* xop, const double (&p)[NS],
                                          const double (&g)[NS], double (&v)[NS], double (&w)[NS]) {
#ifdef LMC_DENSE_COOP
    coop_product<NS>(*mm.coop, p, g, v, w);
    return;
#endif
    stage2<NS>(xop, p, g);
    double acc[2][NS];
    dense_sweep<NS, 2, MatT>(mm, xop, acc);
    vcopy(v, acc[0]);
    vcopy(w, acc[1]);
}

// ---- momentum draws ----------------------------------------------------------------------------------------
// quadpotential.py:450-453: solve_triangular(chol.T, float32(normals)) -- the column sweep of the reference
// BLAS strsv (upper, no-trans): x_j /= U_jj, then x_i -= x_j U_ij for i < j, j descending. U_ij = L[j][i] is
// row j of the row-major factor: contiguous over i. Rows are fetched 8 at a time one block ahead of use.
template <int NS>
__device__ inline void dense_momentum_full(const float* __restrict__ L, int d, int dpad, const lds_double* z,
                                           double (&p0)[NS]) {
    typedef __attribute__((address_space(1))) const Pack<float, NS> glb_pack;
    constexpr int B = 8;
    const int lane = lane_id();
    float x[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int e = lane * NS + s;
        x[s] = (e < d) ? static_cast<float>(z[e]) : 0.0f;
    }
    glb_pack* col = (glb_pack*)(L + lane * NS);
    const int stride = dpad / NS;
    const int d8 = (d + B - 1) / B * B;
    Pack<float, NS> cur[B], nxt[B];
#pragma unroll
    for (int b = 0; b < B; ++b) nxt[b] = col[static_cast<long long>(d8 - B + b) * stride];
    for (int jb = d8 - B; jb >= 0; jb -= B) {
#pragma unroll
        for (int b = 0; b < B; ++b) cur[b] = nxt[b];
        if (jb >= B) {
#pragma unroll
            for (int b = 0; b < B; ++b) nxt[b] = col[static_cast<long long>(jb - B + b) * stride];
        }
#pragma unroll
        for (int b = B - 1; b >= 0; --b) {
            const int j = jb + b;          // wave-uniform
            constexpr int kNsMask = NS - 1;
            const int sj = b & kNsMask;    // == j % NS (jb is a multiple of 8, NS divides 8)
            const int t = j / NS;          // owner lane
            float xs = 0.0f, ls = 1.0f;
#pragma unroll
            for (int s = 0; s < NS; ++s)
                if (s == sj) { xs = x[s]; ls = pack_get<NS>(cur[b], s); }
            const float xj = readlane_f32(xs, t) / readlane_f32(ls, t);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int i = lane * NS + s;
                if (i < j) x[s] = x[s] - pack_get<NS>(cur[b], s) * xj;
                if (s == sj && lane == t) x[s] = xj;
            }
        }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) p0[s] = static_cast<double>(x[s]);
}

// quadpotential.py:411-414: p = L n in float64 (LT rows = columns of L)
template <int NS>
__device__ inline void dense_momentum_inv(const double* __restrict__ LT, int d, int dpad, lds_double* z,
                                          double (&p0)[NS]) {
    // the sweep runs over sweep_rows(d) rows (the extra matrix rows are zero): give it finite operands there
    for (int e = d + lane_id(); e < sweep_rows(d); e += 64) z[e] = 0.0;
    wave_sync();
    double acc[1][NS];
    DenseMat<double> mm{LT, nullptr, 0, d, dpad};
    dense_sweep<NS, 1, double>(mm, z, acc);
    vcopy(p0, acc[0]);
}

// quadpotential.py:450-453 once more, for the kernel in which eight chains meet at every product (run_dense_coop_kernel):
// there the column sweep above -- 128 steps, each waiting for the one before (a division, two cross-lane reads) -- would
// hold seven other chains at the barrier for ~20 k cycles per iteration. With L^-1 formed once on the host in extended
// precision,  solve_triangular(chol.T, z) = L^-T z = sum_k z_k (row k of L^-1)  is a sweep of independent multiply-adds
// like every other matrix product here. z is the float32 cast of the normals and the result is rounded to float32, as the
// reference's (float32 BLAS strsv); the value is the correctly rounded solution instead of strsv's, i.e. it differs from
// the reference by strsv's own float32 rounding error -- the size of difference the dense parity tests already allow for
// the float32-born momentum (tests/test_gpu_dense.py).
template <int NS>
__device__ inline void dense_momentum_solved(const double* __restrict__ Linv, int d, int dpad, lds_double* z, double (&p0)[NS]) {
    const int lane = lane_id();
    double zf[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int e = lane * NS + s;
        zf[s] = (e < d) ? static_cast<double>(static_cast<float>(z[e])) : 0.0;
    }
    wave_sync();
#pragma unroll
    for (int s = 0; s < NS; ++s) z[lane * NS + s] = zf[s];      // (also zero beyond d: the sweep runs over sweep_rows(d) rows)
    for (int e = 64 * NS + lane; e < sweep_rows(d); e += 64) z[e] = 0.0;
    wave_sync();
    double acc[1][NS];
    DenseMat<double> mm{Linv, nullptr, 0, d, dpad, nullptr};
    dense_sweep<NS, 1, double>(mm, z, acc);
#pragma unroll
    for (int s = 0; s < NS; ++s) p0[s] = static_cast<double>(static_cast<float>(acc[0][s]));
}

// ---- start state (integration.py:52-66) -----------------------------------------------------------------------
// Out: v0/w0 = C p0, C g0 in float64 (feed the first half step), v0s = the velocity the reference STORES in the
// start State (float32 for the float32-momentum potentials: numpy's sgemv; we round the float64 sweep once, which
// differs from sgemv's float32 accumulation by a few float32 ulps), e0.
template <int NS, class MatT>
__device__ inline double dense_start_state(Team<1>& tm, const DenseMat<MatT>& mm, double* lds, bool momentum_f32,
                                           int sdot_mode, const double (&p0)[NS], const double (&g0)[NS], double logp0,
                                           double (&v0)[NS], double (&w0)[NS], double (&v0s)[NS]) {
    const int d = mm.d, dpad = mm.dpad;
    velocity2<NS, MatT>(mm, (lds_double*)lds, p0, g0, v0, w0);
    if (!momentum_f32) {
        vcopy(v0s, v0);
        return first_f64(0.5 * tm.sum(pdot<NS>(p0, v0)) - logp0);
    }
    float vf[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        vf[s] = static_cast<float>(v0[s]);
        v0s[s] = static_cast<double>(vf[s]);
    }
    float kin;
    if (sdot_mode == kSdotNative) {
        double part = 0.0;
#pragma unroll
        for (int s = 0; s < NS; ++s)
            part = __builtin_fma(p0[s], static_cast<double>(vf[s]), part);   // exact float32 products, float64 sum
        kin = 0.5f * static_cast<float>(tm.sum(part));
    } else {   // 0.5f * cblas_sdot(p, v) in OpenBLAS' summation order (quadpotential.py:455-459)
        wave_sync();
        float* x = reinterpret_cast<float*>(lds);
        float* y = x + dpad;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            x[lane_id() * NS + s] = static_cast<float>(p0[s]);
            y[lane_id() * NS + s] = vf[s];
        }
        wave_sync();
        kin = 0.5f * sdot_openblas(x, y, d, sdot_mode);
        wave_sync();
    }
    return first_f64(static_cast<double>(kin) - logp0);
}

// ---- leapfrog (integration.py:100-121) with a dense mass matrix -----------------------------------------------
// State in/out: q, p, g, v = C p, w = C g.
template <int NS, class MatT, class Target>
__device__ __forceinline__ void dense_leapfrog(Team<1>& tm, const Target& tgt, const DenseMat<MatT>& mm,
                                               lds_double* xop, double eps, double (&q)[NS], double (&p)[NS],
                                               double (&g)[NS], double (&v)[NS], double (&w)[NS], double& energy,
                                               double& logp) {
    const double dt = 0.5 * eps;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        p[s] = p[s] + dt * g[s];
        const double vh = v[s] + dt * w[s];   // C (p + dt g)
        q[s] = q[s] + eps * vh;
    }
    logp = first_f64(tgt.logp_grad(tm, q, g));
#pragma unroll
    for (int s = 0; s < NS; ++s) p[s] = p[s] + dt * g[s];
    velocity2<NS, MatT>(mm, xop, p, g, v, w);
    energy = first_f64(0.5 * tm.sum(pdot<NS>(p, v)) - logp);
}

// ---- per-chain HBM rows: trajectory ends and subtree stack -------------------------------------------------------
struct DenseScratch {
    glb_double* base;      // HBM row of the chain: every slot has a home here
    lds_double* lbase;     // the first `nlds` slots live in LDS instead (host-chosen: what is left after the matrix cache)
    int nlds;
    int dpad;
    // slots: 0-4 left end {q, p, g, v, w}, 5-9 right end, then 6 per subtree level {lp, lv, rp, rv, psum, proposal q}.
    // Level j is touched with frequency 2^-j, so the low slots are the hot ones.
    static __device__ __forceinline__ int end(int right, int k) { return right * 5 + k; }
    static __device__ __forceinline__ int level(int j, int k) { return 10 + 6 * j + k; }
    template <int NS>
    __device__ __forceinline__ void ld(int slot, double (&x)[NS]) const {
        if (slot < nlds) vload_as<NS>(lbase + slot * dpad, x); else vload_as<NS>(base + slot * dpad, x);
    }
    template <int NS>
    __device__ __forceinline__ void st(int slot, const double (&x)[NS]) const {
        if (slot < nlds) vstore_as<NS>(lbase + slot * dpad, x); else vstore_as<NS>(base + slot * dpad, x);
    }
};

// ---- NUTS / HMC transitions: lmc_tree_leaf.hpp's leaf form / lmc_sampler.hpp's hmc_transition_any with this policy --------
// One wavefront per chain; a state carries v = C p and w = C g (dense_leapfrog); the node under construction, the running
// momentum sum and the proposal position live in registers, the trajectory's ends and the subtree stack in DenseScratch
// (leading slots in LDS, the rest in the chain's scratch row); an end that still is the start state answers the U-turn
// checks with the velocity STORED in the start State (v0s: float32 for the float32 potentials).
template <int NS, class MatT, class Target>
struct DenseTreePolicy {
    static constexpr int kNS = NS;
    struct End { double q[NS], p[NS], g[NS], v[NS], w[NS]; };
    Team<1>& tm; const Target& tgt; const DenseMat<MatT>& mm; lds_double* xop; RngState& rng; const DenseScratch& scr;
    double (&q)[NS];                                       // in: the chain's position; out: the proposal / accepted position
    const double (&p0)[NS]; const double (&g0)[NS]; const double (&v0)[NS]; const double (&w0)[NS]; const double (&v0s)[NS];
    UniformWindow win;
    bool end_is_start[2];
    double t_lp[NS], t_lv[NS], t_ps[NS], t_q[NS], psum[NS], propq[NS];

    __device__ __forceinline__ double uniform() { return team_uniform(tm, rng, win); }
    __device__ __forceinline__ bool any_nonpositive2(double a, double b) { return tm.any_nonpositive2(a, b); }
    __device__ __forceinline__ bool any_nonpositive6(double (&d)[6]) { return tm.any_nonpositive6(d); }
    __device__ __forceinline__ void start_state(End& c) const {
        vcopy(c.q, q); vcopy(c.p, p0); vcopy(c.g, g0); vcopy(c.v, v0); vcopy(c.w, w0);
    }
    __device__ __forceinline__ void accept_state(const End& c) { vcopy(q, c.q); }
    // NUTS: both ends hold the start state, psum = p0, proposal = q
    __device__ __forceinline__ void begin_tree() {
        End c;
        start_state(c);
#pragma unroll
        for (int r = 0; r < 2; ++r) { end_store(r, c); end_is_start[r] = true; }
        vcopy(psum, p0); vcopy(propq, q);
    }
    __device__ __forceinline__ void end_tree() { vcopy(q, propq); }
    __device__ __forceinline__ void end_load(int side, End& c) const {
        scr.ld<NS>(DenseScratch::end(side, 0), c.q); scr.ld<NS>(DenseScratch::end(side, 1), c.p); scr.ld<NS>(DenseScratch::end(side, 2), c.g);
        scr.ld<NS>(DenseScratch::end(side, 3), c.v); scr.ld<NS>(DenseScratch::end(side, 4), c.w);
    }
    __device__ __forceinline__ void end_store(int side, const End& c) {
        scr.st<NS>(DenseScratch::end(side, 0), c.q); scr.st<NS>(DenseScratch::end(side, 1), c.p); scr.st<NS>(DenseScratch::end(side, 2), c.g);
        scr.st<NS>(DenseScratch::end(side, 3), c.v); scr.st<NS>(DenseScratch::end(side, 4), c.w);
        end_is_start[side] = false;
    }
    __device__ __forceinline__ void end_velocity(int side, double (&v)[NS]) const {
        if (end_is_start[side]) vcopy(v, v0s); else scr.ld<NS>(DenseScratch::end(side, 3), v);
    }
    __device__ __forceinline__ void end_momentum(int side, double (&p)[NS]) const { scr.ld<NS>(DenseScratch::end(side, 1), p); }
    __device__ __forceinline__ void leapfrog(double eps, End& c, double& energy, double& logp) {
        dense_leapfrog<NS, MatT>(tm, tgt, mm, xop, eps, c.q, c.p, c.g, c.v, c.w, energy, logp);
    }
    template <int F> __device__ __forceinline__ void node_ld(double (&x)[NS]) const {
        if constexpr (F == kNodeLp) vcopy(x, t_lp); else if constexpr (F == kNodeLv) vcopy(x, t_lv);
        else if constexpr (F == kNodePs) vcopy(x, t_ps); else vcopy(x, t_q);
    }
    template <int F> __device__ __forceinline__ void node_st(const double (&x)[NS]) {
        if constexpr (F == kNodeLp) vcopy(t_lp, x); else if constexpr (F == kNodeLv) vcopy(t_lv, x);
        else if constexpr (F == kNodePs) vcopy(t_ps, x); else vcopy(t_q, x);
    }
    __device__ __forceinline__ void level_ld(int j, int f, double (&x)[NS]) const { scr.ld<NS>(DenseScratch::level(j, f), x); }
    __device__ __forceinline__ void level_st(int j, int f, const double (&x)[NS]) { scr.st<NS>(DenseScratch::level(j, f), x); }
    __device__ __forceinline__ void psum_ld(double (&x)[NS]) const { vcopy(x, psum); }
    __device__ __forceinline__ void psum_st(const double (&x)[NS]) { vcopy(psum, x); }
    __device__ __forceinline__ void proposal_from_node() { vcopy(propq, t_q); }
};

// ---- the iteration kernel ----------------------------------------------------------------------------------------
// LDS: [0, 2*dpad) doubles = sweep operands / normal(size=d) + its staging / float32 sdot staging; then the chain's
// MT19937 state for the duration of the launch.

// All iterations of one chain (one wavefront): the body of the dense-mass kernels. `lds` = this wavefront's private LDS
// (sweep operands / normals + staging, then the MT19937 state); the tree's leading slots in LDS start at `slot_base`.
template <int NS, class MatT, template <int> class TargetT>
__device__ __forceinline__ void dense_run_chain(const ChainArrays& A, const DenseArrays& D, const SamplerParams& P,
                                                const double* tparams, int c, double* lds, const DenseMat<MatT>& mm,
                                                lds_double* slot_base, int n_lds_slots) {
    const int d = A.d, dpad = A.dpad;
    const long long row = static_cast<long long>(c) * dpad;
    Team<1> tm{nullptr, 0};
    const int tid = tm.tid();
    TargetT<NS> tgt;
    tgt.init(tm, tparams, d);
    lds_double* xop = (lds_double*)lds;
    double q[NS];
    vload<NS>(A.q + row, q);
    RngState rng;
    uint32_t* mt_glb = A.mt + static_cast<long long>(c) * kMtN;
    uint32_t* mt_lds = reinterpret_cast<uint32_t*>(lds + 2 * dpad);
    for (int i = tid; i < kMtN; i += 64) mt_lds[i] = mt_glb[i];
    tm.sync();
    rng.mt = mt_lds;
    rng.pos = first_i32(A.rng_pos[c]);
    rng.has_gauss = first_i32(A.rng_has_gauss[c]);
    rng.gauss = first_f64(A.rng_gauss[c]);
    DualAverage da;
    dual_average_load(A, c, da);
    int iter_count = first_i32(A.iter_count[c]);
    long long ct_maxdepth = 0, ct_divs = 0, ct_after = 0, ct_leap = 0;
    int status = 0;
    DenseScratch scr;
    scr.base = (glb_double*)(A.scratch + static_cast<long long>(c) * A.scratch_stride);
    scr.lbase = slot_base;
    scr.nlds = n_lds_slots;
    scr.dpad = dpad;
    const bool momentum_f32 = P.momentum_f32 != 0;

    for (int it = 0; it < P.n_iters; ++it) {
        const long long git = P.iter_begin + it;
        const bool tune = git < P.n_tune;
        const int stop_word = stop_request_load(A, P, c - P.chain_begin, it, git);   // looked at when the iteration ends (lmc_sampler.hpp)

        // ---- momentum draw
        rng_normals(rng, d, lds, lds + dpad);
        double p0[NS];
        if (D.kind == kDenseFullInv)
            dense_momentum_inv<NS>(static_cast<const double*>(D.fac), d, dpad, xop, p0);
#ifdef LMC_DENSE_COOP
        else if (D.fac_inv != nullptr)   // the meeting point must not wait for a 128-step dependent chain: see dense_momentum_solved
            dense_momentum_solved<NS>(D.fac_inv, d, dpad, xop, p0);
#endif
        else
            dense_momentum_full<NS>(static_cast<const float*>(D.fac) + static_cast<long long>(c) * D.fac_stride, d, dpad, xop, p0);

        // ---- start state
        double g0[NS], v0[NS], w0[NS], v0s[NS];
        const double logp0 = first_f64(tgt.logp_grad(tm, q, g0));
        const double e0 = dense_start_state<NS, MatT>(tm, mm, lds, momentum_f32, P.sdot_mode, p0, g0, logp0, v0, w0, v0s);
        if (!isfinite(e0)) {   // base_hmc.py:145-148
            status |= kStatusBadInitialEnergy;
            break;
        }
        const bool adapt_step = tune && P.adapt_step_size;
        const double step_size = jitter_step_size(tm, rng, A, P, c, adapt_step ? da.step_now : da.step_bar_now);

        TransitionOut out;
        DenseTreePolicy<NS, MatT, TargetT<NS>> pol{tm, tgt, mm, xop, rng, scr, q, p0, g0, v0, w0, v0s, UniformWindow{0.0, 0, 0}, {true, true}};
        if (P.kind == 0) {
            const int md = (tune && iter_count < 200) ? P.early_max_treedepth : P.max_treedepth;
            pol.begin_tree();
            leaf_nuts_transition(pol, e0, logp0, step_size, P.emax, md, momentum_f32, out);   // lmc_tree_leaf.hpp
            pol.end_tree();
            if (out.exhausted && !tune) ++ct_maxdepth;
        } else {
            hmc_transition_any(pol, e0, logp0, step_size, P.emax, P.path_length, P.max_steps, out);   // lmc_sampler.hpp
        }
        ct_leap += out.n_leapfrog;
        if (adapt_step) dual_average_update(A, P, out.accept, da);

        if (out.diverging && !tune) ++ct_divs;
        ++iter_count;
        if (!tune) ++ct_after;
        if (A.mom_mean != nullptr && !tune) moments_update<NS>(A, tm, c, row, q);
        write_outputs<NS>(A, c, tid, git, q, out, da.step_now, da.step_bar_now, tune);
        if (first_i32(stop_word) != 0) break;
    }

    tm.sync();
    for (int i = tid; i < kMtN; i += 64) mt_glb[i] = mt_lds[i];
    vstore<NS>(A.q + row, q);
    if (tid == 0) {
        A.rng_pos[c] = rng.pos;
        A.rng_has_gauss[c] = rng.has_gauss;
        A.rng_gauss[c] = rng.gauss;
        A.da[c * 4 + 0] = da.log_step;
        A.da[c * 4 + 1] = da.log_bar;
        A.da[c * 4 + 2] = da.hbar;
        A.da_count[c] = da.count;
        A.iter_count[c] = iter_count;
        A.status[c] |= status;
        A.counters[c * kNumCounters + kCtMaxTreedepth] += ct_maxdepth;
        A.counters[c * kNumCounters + kCtDivsSample] += ct_divs;
        A.counters[c * kNumCounters + kCtSamplesAfterTune] += ct_after;
        A.counters[c * kNumCounters + kCtLeapfrogs] += ct_leap;
    }
}

template <int NS, class MatT, template <int> class TargetT>
__global__ __launch_bounds__(64, dense_waves_per_simd(NS)) void run_dense_kernel(ChainArrays A, DenseArrays D, SamplerParams P, const double* tparams) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int c = blockIdx.x + P.chain_begin;   // the engine launches its chains as sub-blocks (lmc_engine_run)
    const int dpad = A.dpad;
    const int tid = LMC_CHAIN_THREAD;
    if (stop_at_entry<1>(A.stop_dev, nullptr)) return;   // queued behind a Ctrl-C: nothing runs, nothing is touched
    if (A.status[c] & kStatusBadInitialEnergy) return;
    const MatT* M = static_cast<const MatT*>(D.covT) + static_cast<long long>(c) * D.mat_stride;
    MatT* mcache = reinterpret_cast<MatT*>(lds + dense_lds_doubles(dpad));
    for (int j = 0; j < D.cache_rows; ++j) {   // every lane copies (and later reads) its own columns only
#pragma unroll
        for (int s = 0; s < NS; ++s) mcache[j * dpad + tid * NS + s] = M[static_cast<long long>(j) * dpad + tid * NS + s];
    }
    DenseMat<MatT> mm{M, (__attribute__((address_space(3))) const MatT*)mcache, D.cache_rows, A.d, dpad, nullptr};
    lds_double* slots = (lds_double*)(lds + dense_lds_doubles(dpad)) + (static_cast<long long>(D.cache_rows) * dpad * sizeof(MatT)) / 8;
    dense_run_chain<NS, MatT, TargetT>(A, D, P, tparams, c, lds, mm, slots, D.lds_slots);
}

#ifdef LMC_DENSE_COOP
// ---- a matrix shared by all chains (QuadPotentialFull): eight chains per workgroup, one MFMA product per leapfrog ----
// LDS of the workgroup: the transposed float32 matrix [k_rows][dpad + 16], the operand and product panels
// [16][dpad + 2] doubles each, the count of chains still sampling, then eight private regions of dense_lds_doubles(dpad)
// doubles. The chains run their own control flow (ragged trees) and only meet in coop_product(); a chain that is done
// keeps answering the barriers with zero operands until the whole group is done.
constexpr int kCoopWaves = 8;
__host__ __device__ constexpr int coop_ct_stride(int dpad) { return dpad + 16; }
__host__ __device__ constexpr int coop_xs(int dpad) { return dpad + 2; }
__host__ __device__ constexpr int coop_matrix_bytes(int d, int dpad) { return sweep_rows(d) * coop_ct_stride(dpad) * 4; }
__host__ __device__ constexpr int coop_lds_bytes(int d, int dpad, int slots) {
    return coop_matrix_bytes(d, dpad) + 2 * 16 * coop_xs(dpad) * 8 + 16 + kCoopWaves * (dense_lds_doubles(dpad) + slots * dpad) * 8;
}
template <int NS, template <int> class TargetT>
__global__ __launch_bounds__(64 * kCoopWaves, 2) void run_dense_coop_kernel(ChainArrays A, DenseArrays D, SamplerParams P,
                                                                             const double* tparams, int n_chains) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int d = A.d, dpad = A.dpad;
    const int wave = first_i32(static_cast<int>(threadIdx.x) >> 6);
    const int k_rows = sweep_rows(d), cts = coop_ct_stride(dpad), xs = coop_xs(dpad);
    float* ct = reinterpret_cast<float*>(lds);
    double* x = lds + coop_matrix_bytes(d, dpad) / 8;
    double* dout = x + 16 * xs;
    int* n_active = reinterpret_cast<int*>(dout + 16 * xs);
    const int slots = D.lds_slots;
    double* priv = dout + 16 * xs + 2 + wave * (dense_lds_doubles(dpad) + slots * dpad);
    const float* M = static_cast<const float*>(D.covT);
    if (stop_at_entry<kCoopWaves>(A.stop_dev, n_active)) return;   // queued behind a Ctrl-C (one value for the eight chains)
    DenseCoop cc;
    // the matrix: all 512 threads, rows of the stored transposed matrix are contiguous (coalesced)
    for (int idx = threadIdx.x; idx < k_rows * dpad; idx += 64 * kCoopWaves) {
        const int k = idx / dpad, i = idx - k * dpad;
        ct[k * cts + i] = M[idx];
    }
    for (int idx = threadIdx.x; idx < 2 * 16 * xs; idx += 64 * kCoopWaves) x[idx] = 0.0;   // both panels
    const int c = P.chain_begin + static_cast<int>(blockIdx.x) * kCoopWaves + wave;
    const bool mine = (blockIdx.x * kCoopWaves + wave < n_chains) && !(A.status[c < A.chains ? c : 0] & kStatusBadInitialEnergy);
    if (threadIdx.x == 0) *n_active = 0;
    __syncthreads();
    if (mine && lane_id() == 0) atomicAdd(n_active, 1);
    __syncthreads();
    cc.ct = (__attribute__((address_space(3))) const float*)ct;
    cc.x = (lds_double*)x;
    cc.dout = (lds_double*)dout;
    cc.n_active = (__attribute__((address_space(3))) int*)n_active;
    cc.ct_stride = cts; cc.xs = xs; cc.k_rows = k_rows; cc.dpad = dpad; cc.wave = wave; cc.n_waves = kCoopWaves;
#ifdef LMC_COOP_TIMING
    unsigned long long tk[5] = {0, 0, 0, 0, static_cast<unsigned long long>(clock64())};
    cc.tk = tk;
#endif
    if (mine) {
        DenseMat<float> mm{M, nullptr, 0, d, dpad, &cc};
        dense_run_chain<NS, float, TargetT>(A, D, P, tparams, c, priv, mm, (lds_double*)(priv + dense_lds_doubles(dpad)), slots);
        if (lane_id() == 0) atomicSub(n_active, 1);
#ifdef LMC_COOP_TIMING
        if (lane_id() == 0) {   // [0] = waiting for the group | multiplying, [1] = waiting for the product | own work, [2] = after the last product
            A.counters[c * kNumCounters + 0] += static_cast<long long>(((tk[0] & 0xffffffffull) << 32) | (tk[1] & 0xffffffffull));
            A.counters[c * kNumCounters + 1] += static_cast<long long>(((tk[2] & 0xffffffffull) << 32) | (tk[3] & 0xffffffffull));
            A.counters[c * kNumCounters + 2] += static_cast<long long>(clock64() - tk[4]);
        }
#endif
    }
    // drain: answer the group's barriers until every chain is done
    double z[NS], v[NS], w[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) z[s] = 0.0;
    while (coop_product<NS>(cc, z, z, v, w) > 0) {}
}
#endif

// ---- unit kernels (test entries behind lmc_engine_trajectory / lmc_engine_draw_momentum) -----------------------
template <int NS, class MatT, template <int> class TargetT>
__global__ __launch_bounds__(64) void dense_trajectory_kernel(ChainArrays A, DenseArrays D, const double* tparams,
                                                              const double* q0, const double* p0in, int p0_is_f32,
                                                              int sdot_mode, double eps, int n_fwd, int n_back,
                                                              double* oq, double* op, double* ov, double* og, double* oe,
                                                              double* ol) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int c = blockIdx.x;
    const int lane = lane_id();
    const int d = A.d, dpad = A.dpad;
    Team<1> tm{nullptr, 0};
    TargetT<NS> tgt;
    tgt.init(tm, tparams, d);
    const MatT* M = static_cast<const MatT*>(D.covT) + static_cast<long long>(c) * D.mat_stride;
    DenseMat<MatT> mm{M, nullptr, 0, d, dpad};
    double q[NS], p[NS], g[NS], v[NS], w[NS], vs[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int e = lane * NS + s;
        q[s] = (e < d) ? q0[static_cast<long long>(c) * d + e] : 0.0;
        p[s] = (e < d) ? p0in[static_cast<long long>(c) * d + e] : 0.0;
        if (p0_is_f32) p[s] = static_cast<double>(static_cast<float>(p[s]));
    }
    const int n_states = n_fwd + n_back + 1;
    double logp = first_f64(tgt.logp_grad(tm, q, g));
    double energy = dense_start_state<NS, MatT>(tm, mm, lds, p0_is_f32 != 0, sdot_mode, p, g, logp, v, w, vs);
    for (int k = 0; k < n_states; ++k) {
        if (k > 0) {
            dense_leapfrog<NS, MatT>(tm, tgt, mm, (lds_double*)lds, (k <= n_fwd) ? eps : -eps, q, p, g, v, w, energy, logp);
            vcopy(vs, v);
        }
        const long long base = (static_cast<long long>(c) * n_states + k) * d;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int e = lane * NS + s;
            if (e < d) {
                oq[base + e] = q[s];
                op[base + e] = p[s];
                ov[base + e] = vs[s];
                og[base + e] = g[s];
            }
        }
        if (lane == 0) {
            oe[static_cast<long long>(c) * n_states + k] = energy;
            ol[static_cast<long long>(c) * n_states + k] = logp;
        }
    }
}

template <int NS>
__global__ __launch_bounds__(64) void dense_momentum_kernel(ChainArrays A, DenseArrays D, double* out) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int c = blockIdx.x;
    const int lane = lane_id();
    const int d = A.d, dpad = A.dpad;
    RngState r;
    r.mt = A.mt + static_cast<long long>(c) * kMtN;
    r.pos = first_i32(A.rng_pos[c]);
    r.has_gauss = first_i32(A.rng_has_gauss[c]);
    r.gauss = first_f64(A.rng_gauss[c]);
    rng_normals(r, d, lds, lds + dpad);
    double p0[NS];
    if (D.kind == kDenseFullInv)
        dense_momentum_inv<NS>(static_cast<const double*>(D.fac), d, dpad, (lds_double*)lds, p0);
    else
        dense_momentum_full<NS>(static_cast<const float*>(D.fac) + static_cast<long long>(c) * D.fac_stride, d, dpad,
                                (lds_double*)lds, p0);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int e = lane * NS + s;
        if (e < d) out[static_cast<long long>(c) * d + e] = p0[s];
    }
    if (lane == 0) {
        A.rng_pos[c] = r.pos;
        A.rng_has_gauss[c] = r.has_gauss;
        A.rng_gauss[c] = r.gauss;
    }
}

// ---- FullAdapt.update (quadpotential.py:528-552): one workgroup per chain -----------------------------------------
// Phase 1: both estimators take the new sample (Welford rank-1 updates of the d x d second moments, float64,
// streamed through HBM); when a refresh is due the foreground estimate becomes the float32 covariance.
// Phase 2 (refresh only): Cholesky of that covariance with the matrix held in REGISTERS. The workgroup is a
// T x T grid of threads, thread (tx, ty) owns the entries (i, j) with i = tx (mod T), j = ty (mod T) -- up to
// 8 x 8 of them -- for the whole factorisation. Step k: the threads that own column k (one wavefront; the pivot
// reaches them by a lane shuffle) divide it by sqrt(pivot) and publish it in LDS; after ONE barrier every thread
// reads its 8 + 8 multipliers and applies a_ij -= l_ik l_jk to the blocks that are still active. Same operation
// order per entry as host_cholesky() in lmc_engine.hip. The factor goes back row-major through an LDS row buffer
// (coalesced stores). A failed factorisation (pivot <= 0 or a non-finite entry: scipy.linalg.cholesky raises)
// keeps the previous factor and is counted.
constexpr int kCholLocals = 8;
// T: one wavefront (8 x 8 threads) per chain up to d = 64, 256 threads up to d = 128, else 1024
// (0: beyond 32 x 8 = 256 dimensions the matrix does not fit the register file -- cholesky_hbm below, 1024 threads)
constexpr int dense_adapt_grid(int d) { return d <= 8 * kCholLocals ? 8 : d <= 16 * kCholLocals ? 16 : d <= 32 * kCholLocals ? 32 : 0; }
constexpr int dense_adapt_lds_bytes(int d, int dpad) {
    return 4 * d * 8 + 16 + 2 * (d + 4) * 4 + dense_adapt_grid(d) * dpad * 4;
}
static_assert(32 * kCholLocals == kDenseAdaptRegisterMaxDim, "lmc_dense_types.hpp: kDenseAdaptRegisterMaxDim");
constexpr int kCholHbmThreads = 1024;
constexpr int kCholHbmMaxDim = 2 * kCholHbmThreads;   // two rows of a column per thread

template <int T>
__device__ inline bool cholesky_registers(const float* covT, float* fac, int d, int dpad, float* colbuf, float* rowbuf,
                                          int tid) {
    constexpr int R = kCholLocals;
    const int tx = tid % T, ty = tid / T;
    const int lane = tid & 63;
    float a[R][R];
#pragma unroll
    for (int ai = 0; ai < R; ++ai)
#pragma unroll
        for (int bj = 0; bj < R; ++bj) {
            const int i = tx + T * ai, j = ty + T * bj;
            a[ai][bj] = (ai >= bj && i < d && j <= i) ? covT[static_cast<long long>(j) * dpad + i] : 0.0f;   // cov[i][j]
        }
    bool ok = true;
#pragma unroll
    for (int kb = 0; kb < R; ++kb) {
        const int kr_end = ok ? ((d - kb * T) < T ? (d - kb * T) : T) : 0;   // uniform; <= 0 once past d or after a failure
        for (int kr = 0; kr < kr_end; ++kr) {
            const int k = kb * T + kr;
            float* col = colbuf + (k & 1) * (d + 4);   // double buffered: one barrier per step
            if (ty == kr) {   // owners of column k: T consecutive lanes of one wavefront, the pivot owner is lane-local
                const int pivot_lane = (lane / T) * T + kr;           // the thread with tx == kr in this group
                const float akk = __shfl(a[kb][kb], pivot_lane, 64);
                const bool good = (akk > 0.0f) && (akk < __builtin_inf());
                const float lkk = sqrtf(akk);
#pragma unroll
                for (int ai = kb; ai < R; ++ai) {
                    const int i = tx + T * ai;
                    if (i < d && i >= k) {
                        const float v = (i == k) ? lkk : a[ai][kb] / lkk;
                        a[ai][kb] = v;
                        col[i] = v;
                    }
                }
                if (tx == kr) col[d] = good ? 1.0f : 0.0f;
            }
            __syncthreads();
            if (col[d] == 0.0f) { ok = false; break; }   // uniform
            float li[R], lj[R];
#pragma unroll
            for (int ai = kb; ai < R; ++ai) {
                const int i = tx + T * ai;
                li[ai] = (i > k && i < d) ? col[i] : 0.0f;
            }
#pragma unroll
            for (int bj = kb; bj < R; ++bj) {
                const int j = ty + T * bj;
                lj[bj] = (j > k && j < d) ? col[j] : 0.0f;
            }
#pragma unroll
            for (int bj = kb; bj < R; ++bj)
#pragma unroll
                for (int ai = bj; ai < R; ++ai) {
                    const int i = tx + T * ai, j = ty + T * bj;
                    if (i >= j && j > k && i < d) a[ai][bj] = __builtin_fmaf(-li[ai], lj[bj], a[ai][bj]);
                }
        }
    }
    __syncthreads();
    if (!ok) return false;
    // write back L row-major, T rows at a time through LDS
#pragma unroll
    for (int ai = 0; ai < R; ++ai) {
        if (ai * T < d) {   // uniform
#pragma unroll
            for (int bj = 0; bj < R; ++bj) {
                const int i = tx + T * ai, j = ty + T * bj;
                if (j < dpad) rowbuf[tx * dpad + j] = (ai >= bj && j <= i && i < d) ? a[ai][bj] : 0.0f;
            }
            __syncthreads();
            const int rows = (d - ai * T) < T ? (d - ai * T) : T;
            for (int idx = tid; idx < rows * dpad; idx += T * T) fac[static_cast<long long>(ai * T) * dpad + idx] = rowbuf[idx];
            __syncthreads();
        }
    }
    return true;
}

// The same factorisation for a matrix that does not fit the register file (256 < d <= 2048), column by column
// ("left-looking"): entry (i, j) starts from cov[i][j], takes a_ij = fma(-l_ik, l_jk, a_ij) for k = 0 .. j-1 in that order and is
// divided by l_jj = sqrt(a_jj) -- per entry the very operation sequence of cholesky_registers() / host_cholesky(), hence the
// same factor bit for bit. The factor under construction is kept TRANSPOSED in an HBM work area (wt[k][i] = L[i][k]): the
// thread that owns row i reads its operands coalesced over i, the other operand wt[k][j] is one address for the whole
// workgroup. Every thread also carries the pivot's own accumulation (one more fma on the operand it has already loaded),
// so a column costs ONE barrier. A failed factorisation leaves `fac` untouched (the work area is scratch).
__device__ __forceinline__ float chol_fnma(float a, float b, float c) { return __builtin_fmaf(-a, b, c); }
__device__ __forceinline__ double chol_fnma(double a, double b, double c) { return __builtin_fma(-a, b, c); }
__device__ __forceinline__ float chol_sqrt(float a) { return sqrtf(a); }
__device__ __forceinline__ double chol_sqrt(double a) { return sqrt(a); }
template <int kThreads, class MatT>
__device__ inline bool cholesky_hbm(const MatT* covT, MatT* fac, MatT* wt, int d, int dpad, int tid) {
    constexpr int R = kCholHbmMaxDim / kThreads;
    for (int j = 0; j < d; ++j) {
        int row[R];
        MatT acc[R];
        MatT ajj = covT[static_cast<long long>(j) * dpad + j];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = j + tid + r * kThreads;
            row[r] = i < d ? i : j;                                       // (an idle slot recomputes the pivot's row)
            acc[r] = covT[static_cast<long long>(j) * dpad + row[r]];     // cov[i][j]
        }
        const MatT* wk = wt;
        int k = 0;
        for (; k + 8 <= j; k += 8, wk += 8 * static_cast<long long>(dpad)) {
            MatT wj[8], wi[R][8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                wj[u] = wk[static_cast<long long>(u) * dpad + j];
#pragma unroll
                for (int r = 0; r < R; ++r) wi[r][u] = wk[static_cast<long long>(u) * dpad + row[r]];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                ajj = chol_fnma(wj[u], wj[u], ajj);
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r] = chol_fnma(wi[r][u], wj[u], acc[r]);
            }
        }
        for (; k < j; ++k, wk += dpad) {
            const MatT wjk = wk[j];
            ajj = chol_fnma(wjk, wjk, ajj);
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = chol_fnma(wk[row[r]], wjk, acc[r]);
        }
        if (!((ajj > MatT(0)) && (ajj < static_cast<MatT>(__builtin_inf())))) return false;   // uniform: every thread holds the same pivot
        const MatT ljj = chol_sqrt(ajj);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = j + tid + r * kThreads;
            if (i < d) wt[static_cast<long long>(j) * dpad + i] = (i == j) ? ljj : acc[r] / ljj;
        }
        __threadfence_block();
        __syncthreads();   // row j of the work area is complete before column j + 1 reads it
    }
    for (int idx = tid; idx < d * dpad; idx += kThreads) {   // L row-major, zero above the diagonal and in the padding columns
        const int i = idx / dpad, jc = idx - i * dpad;
        fac[idx] = jc <= i ? wt[static_cast<long long>(jc) * dpad + i] : MatT(0);
    }
    return true;
}

template <int T, class MatT = float>
__global__ __launch_bounds__(T > 0 ? T * T : kCholHbmThreads) void dense_adapt_kernel(ChainArrays A, DenseArrays D, double multiplier,
                                                            int update_window, int* mask, int chain_begin, int expect_iter) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int kThreads = T > 0 ? T * T : kCholHbmThreads;
    const int c = blockIdx.x + chain_begin, tid = threadIdx.x;
    // mask != nullptr (tick path): only the chains that finished a tuning iteration in the last tick take part
    if (mask != nullptr && mask[c] == 0) return;
    // expect_iter >= 0 (sample(): the update that follows iteration expect_iter - 1): under a stop request the chains whose
    // iteration launch was skipped (run_dense_kernel: stop_at_entry) have nothing new to learn from -- every thread of the
    // workgroup reads the same two words, and iter_count is final once the iteration's launch is over
    if (expect_iter >= 0 && __hip_atomic_load(A.stop_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 &&
        A.iter_count[c] != expect_iter)
        return;
    const int d = A.d, dpad = A.dpad;
    double* oldf = lds;            // [d] x - mean_before (foreground)
    double* newf = lds + d;        // [d] x - mean_after
    double* oldb = lds + 2 * d;
    double* newb = lds + 3 * d;
    int* flag = reinterpret_cast<int*>(lds + 4 * d);
    float* colbuf = reinterpret_cast<float*>(lds + 4 * d + 2);
    float* rowbuf = colbuf + 2 * (d + 4);
    const long long plane = static_cast<long long>(A.chains) * dpad;
    const long long mplane = static_cast<long long>(A.chains) * d * dpad;
    const int sel = D.esel[c];
    const int n_samples = A.n_samples[c];
    const int prev = D.prev_update[c];
    const int window = D.window[c];
    const int delta = n_samples - prev;
    const double nf = D.en[c * 2 + sel] + 1.0, nb = D.en[c * 2 + 1 - sel] + 1.0;
    double* meanf = D.emean + sel * plane + static_cast<long long>(c) * dpad;
    double* meanb = D.emean + (1 - sel) * plane + static_cast<long long>(c) * dpad;
    double* rawf = D.rawT + sel * mplane + static_cast<long long>(c) * d * dpad;
    double* rawb = D.rawT + (1 - sel) * mplane + static_cast<long long>(c) * d * dpad;
    MatT* covT = static_cast<MatT*>(D.covT) + static_cast<long long>(c) * D.mat_stride;
    MatT* fac = static_cast<MatT*>(D.fac) + static_cast<long long>(c) * D.fac_stride;
    if (tid == 0) *flag = 0;
    for (int i = tid; i < d; i += kThreads) {   // quadpotential.py:594-599
        const double x = A.q[static_cast<long long>(c) * dpad + i];
        double m = meanf[i];
        double od = x - m;
        m = m + od / nf;
        meanf[i] = m;
        oldf[i] = od; newf[i] = x - m;
        m = meanb[i];
        od = x - m;
        m = m + od / nb;
        meanb[i] = m;
        oldb[i] = od; newb[i] = x - m;
    }
    __syncthreads();
    const bool refresh = ((delta + 1) % update_window) == 0;   // quadpotential.py:542-543
    const double denom = nf - 1.0;
    bool bad = false;
    const int total = d * dpad;
    for (int idx = tid; idx < total; idx += kThreads) {
        const int j = idx / dpad, i = idx - j * dpad;
        if (i >= d) continue;
        const double rf = rawf[idx] + 1.0 * newf[i] * oldf[j];   // raw_cov[i][j] += weight * new_i * old_j
        rawf[idx] = rf;
        rawb[idx] = rawb[idx] + 1.0 * newb[i] * oldb[j];
        if (refresh) {
            const MatT cv = static_cast<MatT>(rf / denom);   // np.divide(raw, n - 1, out=cov): the quotient in the potential's dtype
            covT[idx] = cv;
            bad |= !isfinite(cv);
        }
    }
    if (bad) *flag = 1;   // benign race: every writer stores 1
    __threadfence_block();
    __syncthreads();     // covT of this chain is complete and visible to the whole workgroup
    if (refresh) {
        bool ok = (*flag == 0);
        if constexpr (T > 0) {
            static_assert(sizeof(MatT) == 4, "the register-resident factorisation is float32");
            if (ok) ok = cholesky_registers<T>(covT, fac, d, dpad, colbuf, rowbuf, tid);
        } else {
            (void)colbuf; (void)rowbuf;
            if (ok) ok = cholesky_hbm<kThreads, MatT>(covT, fac, static_cast<MatT*>(D.chol_work) + static_cast<long long>(c) * D.mat_stride, d, dpad, tid);
        }
        if (!ok && tid == 0) D.chol_failed[c] += 1;
    }
    __syncthreads();
    const bool switch_window = delta >= window;   // quadpotential.py:547-552
    if (switch_window) {
        for (int idx = tid; idx < total; idx += kThreads) rawf[idx] = 0.0;   // fresh background: eye * 0
        for (int i = tid; i < d; i += kThreads) meanf[i] = 0.0;
    }
    if (tid == 0) {
        if (switch_window) {
            D.en[c * 2 + sel] = 0.0;
            D.en[c * 2 + 1 - sel] = nb;
            D.esel[c] = 1 - sel;
            D.prev_update[c] = n_samples;
            D.window[c] = static_cast<int>(static_cast<double>(window) * multiplier);
        } else {
            D.en[c * 2 + sel] = nf;
            D.en[c * 2 + 1 - sel] = nb;
        }
        A.n_samples[c] = n_samples + 1;
        if (mask != nullptr) mask[c] = 0;   // every thread read it before the first barrier
    }
}

// QuadPotentialFullAdapt.__init__ for every chain (quadpotential.py:474-519): replicate the initial covariance,
// its factor and the foreground estimator (mean, raw = weight * cov, n = weight); empty background.
template <class MatT>
static __global__ __launch_bounds__(256) void dense_reset_kernel(ChainArrays A, DenseArrays D, const MatT* cov1T,
                                                          const MatT* fac1, const double* raw1T, const double* mean1,
                                                          double weight, int window, int d8) {
    const int c = blockIdx.x;
    const int d = A.d, dpad = A.dpad;
    const long long plane = static_cast<long long>(A.chains) * dpad;
    const long long mplane = static_cast<long long>(A.chains) * d * dpad;
    MatT* covT = static_cast<MatT*>(D.covT) + static_cast<long long>(c) * D.mat_stride;
    MatT* fac = static_cast<MatT*>(D.fac) + static_cast<long long>(c) * D.fac_stride;
    double* raw0 = D.rawT + static_cast<long long>(c) * d * dpad;
    double* mean0 = D.emean + static_cast<long long>(c) * dpad;
    const int total = d8 * dpad;
    for (int idx = blockIdx.y * blockDim.x + threadIdx.x; idx < total; idx += gridDim.y * blockDim.x) {
        fac[idx] = fac1[idx];
        if (idx < d * dpad) {
            covT[idx] = cov1T[idx];
            raw0[idx] = raw1T[idx] * weight;
            raw0[mplane + idx] = 0.0;
        }
        if (idx < dpad) {
            mean0[idx] = mean1[idx];
            mean0[plane + idx] = 0.0;
        }
    }
    if (blockIdx.y == 0 && threadIdx.x == 0) {
        D.en[c * 2 + 0] = weight;
        D.en[c * 2 + 1] = 0.0;
        D.esel[c] = 0;
        D.prev_update[c] = 0;
        D.window[c] = window;
        D.chol_failed[c] = 0;
        A.n_samples[c] = 0;
    }
}

}  // namespace lmc
