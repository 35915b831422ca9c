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
// dpp_ctrl encodings (GFX9): row_shr:n = 0x110+n, wave_rol:1 = 0x134, wave_ror:1 = 0x13C,
// row_bcast:15 = 0x142, row_bcast:31 = 0x143. Lanes with no source (or masked rows) receive 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}

// Sum over the 64 lanes; result is wave-uniform (read from lane 63 into SGPRs).
// Hillis-Steele inside each 16-lane row (row_shr 1,2,4,8), then row_bcast 15 / 31 across rows.
__device__ __forceinline__ double wave_sum(double x) {
    x += dpp_f64<0x111, 0xf>(x);
    x += dpp_f64<0x112, 0xf>(x);
    x += dpp_f64<0x114, 0xf>(x);
    x += dpp_f64<0x118, 0xf>(x);
    x += dpp_f64<0x142, 0xa>(x);
    x += dpp_f64<0x143, 0xc>(x);
    return readlane_f64(x, 63);
}

// N independent sums in one pass (the six U-turn dot products of a tree merge): the DPP
// chains are independent, so the scheduler interleaves them and hides the DPP latency.
template <int N>
__device__ __forceinline__ void wave_sum_n(double (&x)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] += dpp_f64<0x111, 0xf>(x[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] += dpp_f64<0x112, 0xf>(x[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] += dpp_f64<0x114, 0xf>(x[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] += dpp_f64<0x118, 0xf>(x[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] += dpp_f64<0x142, 0xa>(x[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] += dpp_f64<0x143, 0xc>(x[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = readlane_f64(x[i], 63);
}

// Neighbour exchange for banded targets: value held by lane-1 / lane+1 (0 at the wave edge).
__device__ __forceinline__ double from_lane_below(double x) {  // lane l receives lane l-1
    const double y = __shfl_up(x, 1, 64);
    return lane_id() == 0 ? 0.0 : y;
}
__device__ __forceinline__ double from_lane_above(double x) {  // lane l receives lane l+1
    const double y = __shfl_down(x, 1, 64);
    return lane_id() == 63 ? 0.0 : y;
}

// Order same-wave accesses to LDS / global scratch that cross lanes (writer lane != reader lane).
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

}  // namespace lmc
