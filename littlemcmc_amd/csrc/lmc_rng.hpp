// Per-chain numpy-legacy random stream on the device: MT19937 + 53-bit doubles + polar gaussian.
//
// The reference draws everything from the global legacy np.random generator, re-seeded per chain
// (/root/reference/littlemcmc/sampling.py:496-497); same-seed parity therefore needs the same
// stream. This is the wave-parallel form of that stream (scalar statement: oracle/mt19937.py):
//   * state = 624 words per chain in HBM (row of the [chains x 624] array) + pos/has_gauss/gauss;
//   * the twist runs as three dependent batches (i < 227, 227 <= i < 454, 454 <= i < 623, then 623);
//   * normal(size=d) evaluates up to 64 polar attempts at once (one per lane), compacts the accepted
//     pairs with ballot/popcount into the consumer's order, and advances pos by exactly the number
//     of words the sequential algorithm would have consumed;
//   * uniform() is a wave-uniform read of two words.
// The translation unit is compiled with -ffp-contract=off, so every variate is the same IEEE
// operation sequence as numpy's C code (no fused multiply-add).
#pragma once
#include "lmc_wave.hpp"

namespace lmc {

constexpr int kMtN = 624;
constexpr int kMtM = 397;

struct RngState {   // wave-uniform registers mirroring numpy's rk_state / legacy gauss cache
    uint32_t* mt;   // this chain's 624 words (global memory)
    int pos;
    int has_gauss;
    double gauss;
};

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9D2C5680u;
    y ^= (y << 15) & 0xEFC60000u;
    y ^= y >> 18;
    return y;
}

__device__ __forceinline__ uint32_t mt_twist(uint32_t a, uint32_t b, uint32_t c) {
    const uint32_t y = (a & 0x80000000u) | (b & 0x7FFFFFFFu);
    return c ^ (y >> 1) ^ ((y & 1u) ? 0x9908B0DFu : 0u);
}

// np.random.seed(s): Knuth LCG fill. Serial by nature; every lane runs it redundantly in
// registers and lanes store disjoint words (executed once per chain).
__device__ inline void mt_seed(RngState& r, uint32_t s) {
    const int lane = lane_id();
    uint32_t x = s;
    if (lane == 0) r.mt[0] = x;
    for (int i = 1; i < kMtN; ++i) {
        x = 1812433253u * (x ^ (x >> 30)) + static_cast<uint32_t>(i);
        if ((i & 63) == lane) r.mt[i] = x;
    }
    r.pos = kMtN;
    r.has_gauss = 0;
    r.gauss = 0.0;
    wave_sync();
}

// genrand twist, in place, 64 words per pass. Batch A/B/C only read words that are either still
// old or were written in an earlier batch; wave_sync() between batches makes those writes visible.
__device__ inline void mt_regen(RngState& r) {
    const int lane = lane_id();
    uint32_t* mt = r.mt;
    // batch A: i in [0, 227): reads old mt[i], mt[i+1], mt[i+397]
    {
        uint32_t nv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = lane + 64 * k;
            nv[k] = (i < 227) ? mt_twist(mt[i], mt[i + 1], mt[i + kMtM]) : 0u;
        }
        wave_sync();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = lane + 64 * k;
            if (i < 227) mt[i] = nv[k];
        }
        wave_sync();
    }
    // batch B: i in [227, 454): reads old mt[i], mt[i+1], new mt[i-227]
    {
        uint32_t nv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = 227 + lane + 64 * k;
            nv[k] = (i < 454) ? mt_twist(mt[i], mt[i + 1], mt[i - 227]) : 0u;
        }
        wave_sync();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = 227 + lane + 64 * k;
            if (i < 454) mt[i] = nv[k];
        }
        wave_sync();
    }
    // batch C: i in [454, 623): reads old mt[i], mt[i+1], new mt[i-227]
    {
        uint32_t nv[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int i = 454 + lane + 64 * k;
            nv[k] = (i < 623) ? mt_twist(mt[i], mt[i + 1], mt[i - 227]) : 0u;
        }
        wave_sync();
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int i = 454 + lane + 64 * k;
            if (i < 623) mt[i] = nv[k];
        }
        wave_sync();
    }
    // i = 623: old mt[623], NEW mt[0], new mt[396]
    if (lane == 0) mt[623] = mt_twist(mt[623], mt[0], mt[396]);
    wave_sync();
    r.pos = 0;
}

// The twist decomposes into 227 independent STRANDS: new[s], new[s + 227], new[s + 454] depend on old words and on each
// other (new[i] needs new[i - 227], and 227 is exactly the batch length), never on another strand; only the last word
// needs new[0] and new[396]. Out of place, one thread per strand therefore twists a whole generation in three dependent
// word steps with no synchronisation in between (team_normals_parallel: all four waves of a team, 227 of 256 threads).
__device__ __forceinline__ void mt_twist_strand(const uint32_t* cur, uint32_t* nxt, int s) {   // s < 227
    const uint32_t n0 = mt_twist(cur[s], cur[s + 1], cur[s + kMtM]);
    nxt[s] = n0;
    const uint32_t n1 = mt_twist(cur[s + 227], cur[s + 228], n0);
    nxt[s + 227] = n1;
    if (s + 454 < kMtN - 1) nxt[s + 454] = mt_twist(cur[s + 454], cur[s + 455], n1);
}

// rk_double: (a >> 5, b >> 6) -> 53-bit fraction. Wave-uniform.
__device__ __forceinline__ double mt_words_to_double(uint32_t w0, uint32_t w1) {
    const uint32_t a = mt_temper(w0) >> 5;
    const uint32_t b = mt_temper(w1) >> 6;
    return (static_cast<double>(a) * 67108864.0 + static_cast<double>(b)) / 9007199254740992.0;
}

// Word-granular like numpy's rk_double (two rk_random calls, each of which twists when the state is exhausted):
// a chain stream of this library only ever sits at even positions, but a state handed over from the host
// (lmc_engine_set_rng_state <- np.random.get_state()) is odd after any 32-bit legacy draw (np.random.randint),
// and then one double straddles the twist.
__device__ inline double rng_uniform(RngState& r) {
    if (r.pos >= kMtN) mt_regen(r);
    const uint32_t w0 = first_u32(r.mt[r.pos]);
    r.pos += 1;
    if (r.pos >= kMtN) mt_regen(r);
    const uint32_t w1 = first_u32(r.mt[r.pos]);
    r.pos += 1;
    return first_f64(mt_words_to_double(w0, w1));
}

// Look-ahead window over the stream: lane l holds the l-th upcoming 53-bit double. A tree transition
// draws one uniform per merge; reading them from registers (v_readlane with a uniform index) keeps the
// two dependent HBM/L2 word loads off every merge's critical path. The window is only valid while
// nothing else advances the stream (empty it with window_reset() around rng_normals()).
struct UniformWindow {
    double val;
    int n;      // doubles available in the window
    int idx;    // next one to hand out
};
__device__ __forceinline__ void window_reset(UniformWindow& w) { w.val = 0.0; w.n = 0; w.idx = 0; }

__device__ inline double window_next(RngState& r, UniformWindow& w) {
    // every quantity that steers this function is wave-uniform and kept scalar on purpose (first_i32): the window test,
    // the index and the stream position are then SALU work and scalar branches instead of VALU compares under exec masks
    int idx = first_i32(w.idx);
    if (idx == first_i32(w.n)) {   // (re)fill from the current stream position
        if (first_i32(r.pos) >= kMtN) mt_regen(r);
        const int pos = first_i32(r.pos);
        int n = (kMtN - pos) >> 1;
        double v = 0.0;
        if (n == 0) {          // one word left (odd position inherited from the host): this double straddles the twist;
            v = rng_uniform(r);   // it becomes a window of one (the position advance below is taken back here)
            r.pos = first_i32(r.pos) - 2;
            n = 1;
        } else {
            n = n > 64 ? 64 : n;
            const int lane = lane_id();
            if (lane < n) v = mt_words_to_double(r.mt[pos + 2 * lane], r.mt[pos + 2 * lane + 1]);
        }
        w.val = v;
        w.n = n;
        idx = 0;
    }
    const double u = readlane_f64(w.val, idx);
    w.idx = idx + 1;
    r.pos = first_i32(r.pos) + 2;
    return u;
}

// log(x) for x in (0, 1) (the polar method's r2), per lane. The generic device log costs ~100 VALU instructions; this is
// fdlibm's scheme without its generality: x = m 2^e with m in [sqrt(1/2), sqrt(2)), f = m - 1, s = f / (2 + f), and
// log m = f - (f^2/2 - s (f^2/2 + R)) where R = s^2 (Lg1 + s^2 (Lg2 + ... s^2 Lg7)) is the minimax tail of
// 2 atanh(s) - 2 s; e ln2 is added in two pieces. The quotient comes from v_rcp_f64 and two Newton steps (2 + f lies in
// [1.7, 2.42]: no scaling). ~38 VALU. Checked on the host with the same IEEE operations against glibc's log over 2e7
// arguments (bulk, near 1, near 0; tools/ubench/log_unit_check.c): never more than 1 ulp apart, and
// sqrt(-2 log(r2) / r2) within 3.1e-16 relative of numpy's value (tests/test_gpu_units.py holds the normals to 5e-16).
// The ten constants arrive in one scalar load.
__constant__ double kLogUnitConst[12] = {
    6.666666666666735130e-01, 3.999999999940941908e-01, 2.857142874366239149e-01, 2.222219843214978396e-01,
    1.818357216161805012e-01, 1.531383769920937332e-01, 1.479819860511658591e-01,
    6.93147180369123816490e-01,   // ln2 hi
    1.90821492927058770002e-10,   // ln2 lo
    0.70710678118654752440, 0.0, 0.0};
__device__ __forceinline__ double log_unit(double x) {
    typedef const __attribute__((address_space(4))) double cst_double;
    cst_double* k = (cst_double*)kLogUnitConst;
    asm volatile("" : "+s"(k));       // loaded where it is used: the constants do not occupy SGPRs across the kernel
    const double Lg1 = k[0], Lg2 = k[1], Lg3 = k[2], Lg4 = k[3], Lg5 = k[4], Lg6 = k[5], Lg7 = k[6];
    const double ln2_hi = k[7], ln2_lo = k[8], sqrt_half = k[9];
    double m = __builtin_amdgcn_frexp_mant(x);          // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(x);
    const bool low = m < sqrt_half;
    m = low ? m + m : m;                                 // [sqrt(1/2), sqrt(2))
    e = low ? e - 1 : e;
    const double f = m - 1.0;
    const double y = 2.0 + f;
    double r = __builtin_amdgcn_rcp(y);
    double t = __builtin_fma(-y, r, 1.0);
    r = __builtin_fma(r, t, r);
    t = __builtin_fma(-y, r, 1.0);
    r = __builtin_fma(r, t, r);
    const double s = f * r;
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * __builtin_fma(w, __builtin_fma(w, Lg6, Lg4), Lg2);
    const double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, Lg7, Lg5), Lg3), Lg1);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = static_cast<double>(e);
    return dk * ln2_hi - ((hfsq - __builtin_fma(s, hfsq + R, dk * ln2_lo)) - f);
}

// normal(size=d) -> out[0..d) (LDS or global scratch, any lane may write any slot).
// Consumer order: attempt k accepted => normals (f*x2, f*x1) in that order; an odd tail leaves
// f*x1 in the cache for the next call (numpy legacy_gauss).
// Two phases so that the expensive part (log, divide, sqrt: ~80 VALU with log_unit) runs once per 64 ACCEPTED pairs
// instead of once per 64 attempts: (1) scan attempts 64 at a time -- temper four words, form (x1, x2),
// test r2 -- and compact the accepted (x1, x2) pairs in stream order into `stage` (room for d doubles);
// (2) one lane per accepted pair computes f = sqrt(-2 log(r2) / r2) and writes both variates.
// The stream position advances by exactly the words the sequential algorithm consumes.
// phase 1: the first need_pairs ACCEPTED polar attempts (x1, x2) of the stream, in stream order, into stage[2 i], stage[2 i + 1]
__device__ inline void rng_polar_pairs(RngState& r, int need_pairs, double* stage) {
    const int lane = lane_id();
    // Loop control is wave-uniform and kept scalar on purpose (first_i32, no `continue`): counts and the stream position
    // then live in SGPRs and the loop is a scalar branch -- otherwise the compiler runs it as an exec-masked loop with
    // VGPR counters.
    int have = 0;
    int pos = first_i32(r.pos);
    while (have < need_pairs) {
        const int avail = (kMtN - pos) >> 2;   // whole attempts left in this generation
        if (avail == 0) {                       // fewer than 4 words left: one attempt across the twist
            r.pos = pos;
            const double x1 = 2.0 * rng_uniform(r) - 1.0;
            const double x2 = 2.0 * rng_uniform(r) - 1.0;
            pos = first_i32(r.pos);
            const double r2 = x1 * x1 + x2 * x2;
            if (ballot64(r2 > 0.0) & ballot64(r2 < 1.0) & 1ull) {
                if (lane == 0) { stage[2 * have] = x1; stage[2 * have + 1] = x2; }
                ++have;
            }
        } else {
            const int n_att = avail < 64 ? avail : 64;
            // every lane forms an attempt (a lane beyond n_att re-reads the last whole one and is masked out below)
            const int at = lane < n_att ? lane : n_att - 1;
            const uint32_t* w = r.mt + pos + 4 * at;
            const double x1 = 2.0 * mt_words_to_double(w[0], w[1]) - 1.0;
            const double x2 = 2.0 * mt_words_to_double(w[2], w[3]) - 1.0;
            const double r2 = x1 * x1 + x2 * x2;
            const bool acc = (r2 > 0.0) & (r2 < 1.0) & (lane < n_att);
            // (one ballot per compare: a ballot of a compound predicate is materialised in a VGPR first)
            const unsigned long long lanes = n_att == 64 ? ~0ull : ((1ull << n_att) - 1ull);
            const unsigned long long mask = ballot64(r2 > 0.0) & ballot64(r2 < 1.0) & lanes;
            const int rank = static_cast<int>(__builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(mask >> 32),
                                              __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(mask), 0u)));
            const int want = need_pairs - have;
            int consumed = n_att, taken = __popcll(mask);
            if (taken >= want) {                  // the want-th accepted attempt ends the call
                const unsigned long long lastm = mask & ballot64(rank == want - 1);
                consumed = __ffsll(static_cast<long long>(lastm));
                taken = want;
            }
            if (acc & (rank < want)) {
                stage[2 * (have + rank)] = x1;
                stage[2 * (have + rank) + 1] = x2;
            }
            have += taken;
            pos += 4 * consumed;
        }
    }
    r.pos = pos;
    wave_sync();
}

__device__ inline void rng_normals(RngState& r, int d, double* out, double* stage) {
    const int lane = lane_id();
    int produced = 0;
    if (r.has_gauss && d > 0) {
        if (lane == 0) out[0] = r.gauss;
        r.has_gauss = 0;
        r.gauss = 0.0;
        produced = 1;
    }
    const int need_pairs = first_i32((d - produced + 1) >> 1);
    rng_polar_pairs(r, need_pairs, stage);
    for (int base = 0; base < need_pairs; base += 64) {
        const int pi = base + lane;
        double g1 = 0.0;
        if (pi < need_pairs) {
            const double x1 = stage[2 * pi], x2 = stage[2 * pi + 1];
            const double r2 = x1 * x1 + x2 * x2;
            const double f = sqrt(-2.0 * log_unit(r2) / r2);
            const int idx = produced + 2 * pi;
            out[idx] = f * x2;
            g1 = f * x1;
            if (idx + 1 < d) out[idx + 1] = g1;
        }
        if (base + 64 >= need_pairs && produced + 2 * need_pairs > d) {   // odd tail: cache the last second variate
            r.gauss = readlane_f64(g1, need_pairs - 1 - base);
            r.has_gauss = 1;
        }
    }
    wave_sync();
}

// normal(size=d) delivered to the registers of the lanes that own the elements (lane l: elements l*NS .. l*NS+NS-1; 0
// beyond d). For an EVEN d with no cached gaussian -- every draw of an even-dimensional chain -- pair i of the stream IS
// elements 2i, 2i+1, so the lane that owns them evaluates f = sqrt(-2 log r2 / r2) for its own pairs and the variates never
// travel through LDS (the general form writes them to `out`, waits, and every lane reads its elements back: ~500 cycles of
// a depth-3 iteration's 25 000, measured with a draw probe in round 4). Same pairs, same arithmetic, same stream advance;
// phase 1 is shared. NS = 1: lanes 2i and 2i+1 both evaluate pair i and keep their own variate.
template <int NS>
__device__ __forceinline__ void rng_normals_owned(RngState& r, int d, double* out, double* stage, double (&z)[NS]) {
    const int lane = lane_id();
    int produced = 0;
    if (r.has_gauss && d > 0) {
        if (lane == 0) out[0] = r.gauss;
        r.has_gauss = 0;
        r.gauss = 0.0;
        produced = 1;
    }
    const int need_pairs = first_i32((d - produced + 1) >> 1);
    rng_polar_pairs(r, need_pairs, stage);
    if (produced == 0 && (d & 1) == 0) {   // wave-uniform
        if constexpr (NS == 1) {
            const int pi = lane >> 1;
            double v = 0.0;
            if (pi < need_pairs) {
                const double x1 = stage[2 * pi], x2 = stage[2 * pi + 1];
                const double r2 = x1 * x1 + x2 * x2;
                const double f = sqrt(-2.0 * log_unit(r2) / r2);
                v = (lane & 1) ? f * x1 : f * x2;
            }
            z[0] = v;
        } else {
            static_assert(NS % 2 == 0, "a lane owns whole pairs");
#pragma unroll
            for (int k = 0; k < NS / 2; ++k) {
                const int pi = (NS / 2) * lane + k;
                double a = 0.0, b = 0.0;
                if (pi < need_pairs) {
                    const double x1 = stage[2 * pi], x2 = stage[2 * pi + 1];
                    const double r2 = x1 * x1 + x2 * x2;
                    const double f = sqrt(-2.0 * log_unit(r2) / r2);
                    a = f * x2;
                    b = f * x1;
                }
                z[2 * k] = a;
                z[2 * k + 1] = b;
            }
        }
        wave_sync();   // the staging area may be reused
        return;
    }
    for (int base = 0; base < need_pairs; base += 64) {
        const int pi = base + lane;
        double g1 = 0.0;
        if (pi < need_pairs) {
            const double x1 = stage[2 * pi], x2 = stage[2 * pi + 1];
            const double r2 = x1 * x1 + x2 * x2;
            const double f = sqrt(-2.0 * log_unit(r2) / r2);
            const int idx = produced + 2 * pi;
            out[idx] = f * x2;
            g1 = f * x1;
            if (idx + 1 < d) out[idx + 1] = g1;
        }
        if (base + 64 >= need_pairs && produced + 2 * need_pairs > d) {   // odd tail: cache the last second variate
            r.gauss = readlane_f64(g1, need_pairs - 1 - base);
            r.has_gauss = 1;
        }
    }
    wave_sync();
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int e = lane * NS + s;
        z[s] = (e < d) ? out[e] : 0.0;
    }
    wave_sync();
}

// ---- counter-based momentum draw (LMC_RNG_PHILOX: the throughput mode, NOT the reference's stream) -----------------
// With the numpy-legacy stream the momentum draw is ~20 % of a depth-3 iteration: the polar method's rejection loop, the
// compaction of accepted pairs through LDS, the MT19937 twist. When same-seed parity with the reference is not needed,
// normal(size=d) can instead be a pure function of (chain seed, iteration, element): Philox4x32-10 (Salmon et al., SC11;
// key = the chain's seed, counter = iteration and lane) and a float32 Box-Muller -- the momentum of QuadPotentialDiagAdapt
// is float32 in the reference too (quadpotential.py:221-224) -- with the hardware's single-instruction log2 / sin / cos.
// No LDS, no cross-lane traffic, no barrier; every thread of a team draws its own elements. Results are independent of
// launch slicing and of the chain-block partition, like the parity stream's. Tree uniforms stay on the chain's MT19937.
struct Philox4 { uint32_t c[4]; };
__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    Philox4 o;
    o.c[0] = c0; o.c[1] = c1; o.c[2] = c2; o.c[3] = c3;
    return o;
}
// two standard normals (float32 precision) from two 32-bit words
__device__ __forceinline__ void box_muller_f32(uint32_t a, uint32_t b, float& z0, float& z1) {
    // (0, 1]: (k + 1) 2^-24, exact in float32 for every 24-bit k (the half-offset form (k + 1/2) 2^-24 rounded to 1.0 at
    // k = 2^24 - 1, i.e. a (0, 0) pair, and lost its offset above 1/2: round-3 review)
    const float u1 = static_cast<float>((a >> 8) + 1u) * 5.9604645e-08f;
    const float u2 = static_cast<float>(b >> 8) * 5.9604645e-08f;                    // [0, 1) of a revolution
    const float r = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));   // sqrt(-2 ln u1), v_log_f32 is log2
    z0 = r * __builtin_amdgcn_cosf(u2);   // v_cos_f32 / v_sin_f32 take revolutions
    z1 = r * __builtin_amdgcn_sinf(u2);
}
// z[s] = the normal of element thread * NS + s (0 beyond d) for iteration `git` of the chain seeded `seed`
template <int NS>
__device__ __forceinline__ void philox_normals(uint32_t seed, long long git, int thread, int d, double (&z)[NS]) {
    const Philox4 w = philox4x32_10(static_cast<uint32_t>(git), static_cast<uint32_t>(git >> 32), static_cast<uint32_t>(thread),
                                    0x6c6d636du /* "lmcm": the momentum stream */, seed, 0x4d4f4d31u);
    float n[4];
    box_muller_f32(w.c[0], w.c[1], n[0], n[1]);
    if constexpr (NS > 2) box_muller_f32(w.c[2], w.c[3], n[2], n[3]);
    static_assert(NS <= 4, "one Philox call per thread");
#pragma unroll
    for (int s = 0; s < NS; ++s) z[s] = (thread * NS + s < d) ? static_cast<double>(n[s]) : 0.0;
}

}  // namespace lmc
