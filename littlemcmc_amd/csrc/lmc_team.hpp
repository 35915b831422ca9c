// A "team" is the set of wavefronts that cooperate on ONE chain: W = 1 (d <= 128: the whole chain lives in
// one wave, every scalar is wave-uniform, no barriers) or W = 2, 4 (larger d: the length-d vectors are
// spread over 64*W threads, 8*NS contiguous bytes per thread; all waves of the block run the same
// control flow on block-uniform scalars and meet in LDS for reductions).
//
// Cross-wave reduction: every wave reduces its 64 lanes (DPP / permlane-swap, lmc_wave.hpp), lane 0 drops the
// wave totals into a double-buffered LDS exchange area, ONE __syncthreads(), then every wave adds the W totals
// in wave order (deterministic, identical in all waves). Double buffering makes the single barrier enough:
// a wave can only reach the next-but-one reduction (same buffer) after all waves passed the barrier in between.
#pragma once
#include "lmc_wave.hpp"

namespace lmc {

// Six sums at once, totals returned (the ballot form in lmc_wave.hpp only answers "any <= 0").
__device__ __forceinline__ void wave_sum6_totals(double (&d)[6]) {
    const double x01 = swap32_add(d[0], d[1]);
    const double x23 = swap32_add(d[2], d[3]);
    const double x45 = swap32_add(d[4], d[5]);
    double y = swap16_add(x01, x23);       // rows: d0, d2, d1, d3
    double z = swap16_add(x45, 0.0);       // rows: d4, 0, d5, 0
    y = row_scan(y);
    z = row_scan(z);
    d[0] = readlane_f64(y, 15); d[2] = readlane_f64(y, 31); d[1] = readlane_f64(y, 47); d[3] = readlane_f64(y, 63);
    d[4] = readlane_f64(z, 15); d[5] = readlane_f64(z, 47);
}

constexpr int kTeamSlots = 8;   // doubles per wave per exchange buffer

template <int W>
struct Team {
    static constexpr int kWaves = W;
    static constexpr int kThreads = 64 * W;
    double* xbuf;   // LDS exchange area, 2 * W * kTeamSlots doubles (unused for W == 1)
    int parity;

    __device__ __forceinline__ int tid() const { return LMC_CHAIN_THREAD; }
    __device__ __forceinline__ int wave() const { return first_i32(static_cast<int>(threadIdx.x) >> 6); }
    __device__ __forceinline__ void sync() const {
        if constexpr (W == 1) wave_sync(); else __syncthreads();
    }

    // v[]: wave-uniform partial totals in, team totals out (same in every wave)
    template <int N>
    __device__ __forceinline__ void exchange(double (&v)[N]) {
        if constexpr (W > 1) {
            static_assert(N <= kTeamSlots, "exchange buffer too small");
            double* buf = xbuf + parity * (W * kTeamSlots);
            if (lane_id() == 0) {
                double* mine = buf + wave() * kTeamSlots;
#pragma unroll
                for (int n = 0; n < N; ++n) mine[n] = v[n];
            }
            __syncthreads();
#pragma unroll
            for (int n = 0; n < N; ++n) {
                double acc = buf[n];
#pragma unroll
                for (int w = 1; w < W; ++w) acc += buf[w * kTeamSlots + n];
                v[n] = first_f64(acc);
            }
            parity ^= 1;
        }
    }

    __device__ __forceinline__ double sum(double x) {
        double v[1] = {wave_sum(x)};
        exchange<1>(v);
        return v[0];
    }
    __device__ __forceinline__ void sum2(double& a, double& b) {
        wave_sum2(a, b);
        if constexpr (W > 1) {
            double v[2] = {a, b};
            exchange<2>(v);
            a = v[0]; b = v[1];
        }
    }
    __device__ __forceinline__ bool any_nonpositive2(double d0, double d1) {
        if constexpr (W == 1) {
            return any_sum_nonpositive2(d0, d1);
        } else {
            sum2(d0, d1);
            return (d0 <= 0.0) | (d1 <= 0.0);
        }
    }
    __device__ __forceinline__ bool any_nonpositive6(double (&d)[6]) {
        if constexpr (W == 1) {
            return any_sum_nonpositive6(d);
        } else {
            wave_sum6_totals(d);
            exchange<6>(d);
            return (d[0] <= 0.0) | (d[1] <= 0.0) | (d[2] <= 0.0) | (d[3] <= 0.0) | (d[4] <= 0.0) | (d[5] <= 0.0);
        }
    }
    // value held by thread 0 of the team, delivered to every thread
    __device__ __forceinline__ double bcast0(double x) {
        if constexpr (W == 1) {
            return readlane_f64(x, 0);
        } else {
            double v[1] = {readlane_f64(x, 0)};
            if (wave() != 0) v[0] = 0.0;
            exchange<1>(v);   // sum of {x0, 0, 0, ...}
            return v[0];
        }
    }
    // banded targets: thread t receives `lo_src` of thread t-1 and `hi_src` of thread t+1 (0 at the team's edges)
    __device__ __forceinline__ void neighbours(double lo_src, double hi_src, double& below, double& above) {
        below = dpp_f64<0x138>(lo_src);   // wave_shr:1
        above = dpp_f64<0x130>(hi_src);   // wave_shl:1
        if constexpr (W > 1) {
            double* buf = xbuf + parity * (W * kTeamSlots);
            const int w = wave();
            const double last = readlane_f64(lo_src, 63), first = readlane_f64(hi_src, 0);
            if (lane_id() == 0) { buf[w * kTeamSlots] = last; buf[w * kTeamSlots + 1] = first; }
            __syncthreads();
            const double from_prev = (w > 0) ? buf[(w - 1) * kTeamSlots] : 0.0;
            const double from_next = (w < W - 1) ? buf[(w + 1) * kTeamSlots + 1] : 0.0;
            if (lane_id() == 0) below = from_prev;
            if (lane_id() == 63) above = from_next;
            parity ^= 1;
        }
    }
};

}  // namespace lmc
