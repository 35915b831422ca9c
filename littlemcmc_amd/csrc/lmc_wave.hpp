// Wavefront-level primitives for gfx950 (CDNA4): one chain == one 64-lane wavefront.
//
// Every per-chain scalar (energies, tree weights, U-turn dot products, RNG position) is
// wave-uniform; length-d vectors are spread over the 64 lanes in a *blocked* layout
// (lane l owns elements l*NS .. l*NS+NS-1, NS = ceil(d/64)) so that one lane's slice is a
// contiguous 8*NS-byte run: global and LDS accesses become dwordx2/x4 per lane and fully
// coalesced per wave.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lmc {

constexpr int kWave = 64;

__device__ __forceinline__ int lane_id() { return static_cast<int>(threadIdx.x) & 63; }
// Index of a thread within the team that owns its chain: the thread index of the block when a block IS one chain
// (every kernel but one); a translation unit whose blocks hold several one-wave chains side by side (lmc_dense_coop.hip:
// eight chains meet per leapfrog for one MFMA product) defines it as the lane.
#ifndef LMC_CHAIN_THREAD
#define LMC_CHAIN_THREAD static_cast<int>(threadIdx.x)
#endif
// lane predicate -> 64-bit mask. The builtin takes the i1 as it is (a v_cmp result already IS the mask in an SGPR pair);
// HIP's __ballot() widens it to an int first, which costs a v_cndmask + v_cmp_ne per call.
__device__ __forceinline__ unsigned long long ballot64(bool pred) { return __builtin_amdgcn_ballot_w64(pred); }

// ---- uniform <-> per-lane moves -------------------------------------------------------------
__device__ __forceinline__ double readlane_f64(double x, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double first_f64(double x) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(x));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(x));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int first_i32(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ uint32_t first_u32(uint32_t x) {
    return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(x)));
}

// ---- DPP data movement (VALU cross-lane, no LDS traffic) ---------------------------------------
// dpp_ctrl encodings (GFX9): row_shr:n = 0x110+n, row_bcast:15 = 0x142, row_bcast:31 = 0x143.
// bound_ctrl:1 with full row/bank masks: lanes without a source lane read 0, so no "old" register has
// to be zeroed first (that would be one extra VALU move per DPP move).
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float readlane_f32(float x, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), lane));
}

// Sum over the 64 lanes; result is wave-uniform (read from lane 63 into SGPRs).
// Hillis-Steele inside each 16-lane row (row_shr 1,2,4,8), then row_bcast 15 / 31 across rows.
__device__ __forceinline__ double wave_sum(double x) {
    // rows get extra cross terms from the unmasked broadcasts, but lane 63 ends up with exactly
    // R3 + R2 + (R1 + R0): row_bcast:15 adds R_{k-1} to row k, row_bcast:31 adds lane 31 (R1 + R0) to rows 2,3
    x += dpp_f64<0x111>(x);
    x += dpp_f64<0x112>(x);
    x += dpp_f64<0x114>(x);
    x += dpp_f64<0x118>(x);
    x += dpp_f64<0x142>(x);
    x += dpp_f64<0x143>(x);
    return readlane_f64(x, 63);
}

// ---- transposed multi-value reductions (gfx950 v_permlane32_swap / v_permlane16_swap) ------------------
// Reducing K values with K independent butterflies costs K * 6 stages. Swapping halves instead lets ONE
// add serve two values: after v_permlane32_swap(a, b) the registers hold [a_lo | b_lo] and [a_hi | b_hi],
// so their sum carries a's partial sums in lanes 0-31 and b's in lanes 32-63; v_permlane16_swap does the
// same for 16-lane rows. Each step halves the number of live registers; the remaining intra-row scan
// (row_shr 1,2,4,8) runs once for up to four values.
__device__ __forceinline__ double swap32_add(double a, double b) {   // [a_lo + a_hi | b_lo + b_hi]
    const auto lo = __builtin_amdgcn_permlane32_swap(static_cast<unsigned>(__double2loint(a)),
                                                     static_cast<unsigned>(__double2loint(b)), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(static_cast<unsigned>(__double2hiint(a)),
                                                     static_cast<unsigned>(__double2hiint(b)), false, false);
    return __hiloint2double(static_cast<int>(hi[0]), static_cast<int>(lo[0])) +
           __hiloint2double(static_cast<int>(hi[1]), static_cast<int>(lo[1]));
}
__device__ __forceinline__ double swap16_add(double a, double b) {   // rows [a0+a1, b0+b1, a2+a3, b2+b3]
    const auto lo = __builtin_amdgcn_permlane16_swap(static_cast<unsigned>(__double2loint(a)),
                                                     static_cast<unsigned>(__double2loint(b)), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(static_cast<unsigned>(__double2hiint(a)),
                                                     static_cast<unsigned>(__double2hiint(b)), false, false);
    return __hiloint2double(static_cast<int>(hi[0]), static_cast<int>(lo[0])) +
           __hiloint2double(static_cast<int>(hi[1]), static_cast<int>(lo[1]));
}
__device__ __forceinline__ double row_scan(double x) {   // lane 15 of every 16-lane row = row total
    x += dpp_f64<0x111>(x);
    x += dpp_f64<0x112>(x);
    x += dpp_f64<0x114>(x);
    x += dpp_f64<0x118>(x);
    return x;
}

// Two sums for the price of ~one: a <- sum(a), b <- sum(b), both wave-uniform.
__device__ __forceinline__ void wave_sum2(double& a, double& b) {
    double x = swap32_add(a, b);           // lanes 0-31: a partials, lanes 32-63: b partials
    x = row_scan(x);
    x += dpp_f64<0x142>(x);                // row_bcast:15: lane 31 = R0 + R1 = sum(a), lane 63 = R2 + R3 = sum(b)
    a = readlane_f64(x, 31);
    b = readlane_f64(x, 63);
}

// U-turn tests: true if ANY of the given dot products (per-lane partials in d[]) is <= 0.
__device__ __forceinline__ bool any_sum_nonpositive2(double d0, double d1) {
    double x = swap32_add(d0, d1);
    x = row_scan(x);
    x += dpp_f64<0x142>(x);
    const unsigned long long m = ballot64(x <= 0.0);
    return (m & ((1ull << 31) | (1ull << 63))) != 0ull;
}
__device__ __forceinline__ bool any_sum_nonpositive6(const double (&d)[6]) {
    const double x01 = swap32_add(d[0], d[1]);
    const double x23 = swap32_add(d[2], d[3]);
    const double x45 = swap32_add(d[4], d[5]);
    double y = swap16_add(x01, x23);       // rows: d0, d2, d1, d3 (16 partials each)
    double z = swap16_add(x45, 0.0);       // rows: d4, 0, d5, 0
    y = row_scan(y);
    z = row_scan(z);
    const unsigned long long my = ballot64(y <= 0.0);
    const unsigned long long mz = ballot64(z <= 0.0);
    const unsigned long long rows = (1ull << 15) | (1ull << 31) | (1ull << 47) | (1ull << 63);
    return ((my & rows) | (mz & ((1ull << 15) | (1ull << 47)))) != 0ull;
}

// ---- cheap exp for wave-uniform arguments ---------------------------------------------------------------
// The OCML exp spends half of its ~50 VALU instructions moving polynomial constants into VGPRs. Here the
// constants are produced by the scalar unit (s_mov, free next to a saturated vector pipe) and feed the FMAs
// as SGPR operands. Cody-Waite reduction by ln2 (hi/lo), degree-13 Taylor polynomial on |r| <= ln2/2
// (truncation 4e-18 relative), ldexp. |error| < ~1.5 ulp; arguments are clamped to [-800, 800].
template <unsigned long long BITS>
__device__ __forceinline__ double sgpr_const() {
    unsigned lo, hi;
    asm("s_mov_b32 %0, %1" : "=s"(lo) : "i"(static_cast<unsigned>(BITS & 0xffffffffull)));
    asm("s_mov_b32 %0, %1" : "=s"(hi) : "i"(static_cast<unsigned>(BITS >> 32)));
    return __hiloint2double(static_cast<int>(hi), static_cast<int>(lo));
}
#define LMC_SC(x) (::lmc::sgpr_const<__builtin_bit_cast(unsigned long long, static_cast<double>(x))>())

// p * r + c with c taken straight from an SGPR pair (VOP3 form; hipcc would pick v_fmac and copy c first)
__device__ __forceinline__ double fma_sgpr_addend(double p, double r, double c) {
    double out;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(out) : "v"(p), "v"(r), "s"(c));
    return out;
}

__device__ __forceinline__ double exp_uniform(double x) {
    double xv = fmin(fmax(x, -800.0), 800.0);
    asm volatile("" : "+v"(xv));                       // keep the argument in a VGPR: one SGPR operand per VALU op
    const double kf = rint(xv * LMC_SC(1.4426950408889634074));
    double r = __builtin_fma(-kf, LMC_SC(6.93147180369123816490e-01), xv);
    r = __builtin_fma(-kf, LMC_SC(1.90821492927058770002e-10), r);
    double p = fma_sgpr_addend(r, LMC_SC(1.0 / 6227020800.0), LMC_SC(1.0 / 479001600.0));
    p = fma_sgpr_addend(p, r, LMC_SC(1.0 / 39916800.0));
    p = fma_sgpr_addend(p, r, LMC_SC(1.0 / 3628800.0));
    p = fma_sgpr_addend(p, r, LMC_SC(1.0 / 362880.0));
    p = fma_sgpr_addend(p, r, LMC_SC(1.0 / 40320.0));
    p = fma_sgpr_addend(p, r, LMC_SC(1.0 / 5040.0));
    p = fma_sgpr_addend(p, r, LMC_SC(1.0 / 720.0));
    p = fma_sgpr_addend(p, r, LMC_SC(1.0 / 120.0));
    p = fma_sgpr_addend(p, r, LMC_SC(1.0 / 24.0));
    p = fma_sgpr_addend(p, r, LMC_SC(1.0 / 6.0));
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return first_f64(ldexp(p, static_cast<int>(kf)));
}

// exp for a wave-uniform argument that is known to be <= ~700 (tree weights: x - c <= 600 by construction,
// very negative arguments underflow to 0 through ldexp). Table-driven: x = (64 e + j) ln2/64 + r, |r| <= ln2/128,
// exp(x) = 2^e * 2^(j/64) * (1 + r + ... + r^5/120)   (truncation 4e-17 relative). The 64-entry table sits in
// constant memory and is fetched with ONE scalar load (the index is wave-uniform), so the VALU cost is ~15
// instructions instead of ~26 for the polynomial-only form. |error| < ~1 ulp.
__constant__ double kExp2Table[64] = {
    1.0, 1.0108892860517005, 1.0218971486541166, 1.0330248790212284,
    1.0442737824274138, 1.0556451783605572, 1.0671404006768237, 1.0787607977571199,
    1.0905077326652577, 1.102382583307841, 1.1143867425958924, 1.1265216186082418,
    1.1387886347566916, 1.1511892299529827, 1.1637248587775775, 1.1763969916502812,
    1.189207115002721, 1.202156731452703, 1.215247359980469, 1.22848053610687,
    1.241857812073484, 1.255380757024691, 1.2690509571917332, 1.2828700160787783,
    1.2968395546510096, 1.3109612115247644, 1.3252366431597413, 1.339667524053303,
    1.3542555469368927, 1.3690024229745905, 1.383909881963832, 1.3989796725383112,
    1.4142135623730951, 1.42961333839197, 1.4451808069770467, 1.460917794180647,
    1.4768261459394993, 1.4929077282912648, 1.5091644275934228, 1.5255981507445384,
    1.5422108254079407, 1.559004400237837, 1.5759808451078865, 1.593142151342267,
    1.6104903319492543, 1.6280274218573478, 1.645755478153965, 1.6636765803267364,
    1.681792830507429, 1.7001063537185235, 1.718619298122478, 1.7373338352737062,
    1.7562521603732995, 1.7753764925265212, 1.7947090750031072, 1.8142521755003989,
    1.8340080864093424, 1.8539791250833855, 1.8741676341103, 1.8945759815869656,
    1.9152065613971474, 1.9360617934922943, 1.9571441241754002, 1.978456026387951};

__device__ __forceinline__ double exp_uniform_fast(double x) {
    double xv = x;
    asm volatile("" : "+v"(xv));
    const double kf = rint(xv * LMC_SC(92.332482616893656));            // 64 / ln 2
    double r = __builtin_fma(-kf, LMC_SC(1.08304246932675596327e-02), xv);   // ln2/64 hi
    r = __builtin_fma(-kf, LMC_SC(2.98158582698529328128e-12), r);           // ln2/64 lo
    const int ki = first_i32(static_cast<int>(kf));
    const double t = kExp2Table[ki & 63];
    double p = fma_sgpr_addend(r, LMC_SC(1.0 / 120.0), LMC_SC(1.0 / 24.0));
    p = fma_sgpr_addend(p, r, LMC_SC(1.0 / 6.0));
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return ldexp(p * t, ki >> 6);   // wave-uniform value left in a VGPR pair (callers add to / compare with it)
}

// Uniform predicate from a comparison whose operands are wave-uniform but VGPR-resident: all lanes agree, so
// the ballot is either 0 or exec. One v_cmp + scalar test; tells the compiler the branch is uniform.
__device__ __forceinline__ bool uniform_true(bool lane_pred) { return ballot64(lane_pred) != 0ull; }

// Neighbour exchange for banded targets: value held by lane-1 / lane+1 (0 at the wave edge).
// DPP wave_shr:1 / wave_shl:1 (GFX9 whole-wave shifts, 0x138 / 0x130) with bound_ctrl: two VALU moves per
// double, no LDS crossbar round trip.
__device__ __forceinline__ double from_lane_below(double x) { return dpp_f64<0x138>(x); }  // lane l <- lane l-1
__device__ __forceinline__ double from_lane_above(double x) { return dpp_f64<0x130>(x); }  // lane l <- lane l+1

// Order same-wave accesses to LDS / global scratch that cross lanes (writer lane != reader lane).
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

}  // namespace lmc
