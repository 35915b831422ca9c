#!/usr/bin/env python3
"""Generate littlemcmc_amd/csrc/lmc_tick_wide.hpp from lmc_tick.hpp by targeted text transformations: the tick state
machine (externally evaluated densities: Python / torch callables) for model_ndim > 1024, one chain = the general kernels'
workgroup of 16 wavefronts (lmc_wide.hpp), statement-parallel to the one-wavefront tick kernel. Every transformation asserts
its anchor count: if lmc_tick.hpp changes shape this script fails loudly instead of producing a silently different kernel.
Run from the repo root: python tools/gen_tick_wide.py"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "littlemcmc_amd/csrc/lmc_tick_wide.hpp")
src = open(os.path.join(ROOT, "littlemcmc_amd/csrc/lmc_tick.hpp")).read()
body = src[src.index("// register budget per vector width"):src.index("// chains that still want evaluations")]
begin = src[src.index("// lmc_engine_tick_begin(): every chain asks for the density at its current position"):src.rindex("}  // namespace lmc")]


def rep(a, b, count=1, text=None):
    global body
    t = body if text is None else text
    assert t.count(a) == count, (t.count(a), a[:80])
    t = t.replace(a, b)
    if text is None:
        body = t
    return t


rep('''// register budget per vector width (waves per SIMD): the tick kernel is latency / bandwidth bound and insensitive to
// occupancy (4 / 6 / 8 waves measured equal at NS = 2), so wide vectors simply get the registers they need
constexpr int tick_waves_per_simd(int ns) { return ns <= 2 ? 4 : ns == 4 ? 2 : 1; }
template <int NS>
__global__ __launch_bounds__(64, tick_waves_per_simd(NS)) void tick_kernel(ChainArrays A, TickArrays K, SamplerParams P, const double* logp_in,
                                                  const double* grad_in) {''', '''template <int NS>
__global__ __launch_bounds__(kWideThreads, 1) void tick_wide_kernel(ChainArrays A, TickArrays K, SamplerParams P, const double* logp_in,
                                                                     const double* grad_in) {''')
rep("    extern __shared__ __attribute__((aligned(16))) double lds[];   // 2 * dpad doubles: normals + staging / sdot staging",
    "    extern __shared__ __attribute__((aligned(16))) double lds[];   // wide_stage_doubles(dpad): normals chunk + staging / sdot staging; team exchange; broadcast words")
rep("    const int lane = lane_id();",
    "    const int lane = static_cast<int>(threadIdx.x);   // the thread's index in its chain (the name is the one-wavefront kernel's)")
rep("    Team<1> tm{nullptr, 0};",
    "    WideTeam tm;\n    tm.xbuf = lds + wide_stage_doubles(dpad);\n    tm.parity = 0;\n    double* bcast = tm.xbuf + 2 * kWideWaves * kTeamSlots;")
# per-level scalars: every wave keeps its own copy on its lanes
rep('''    if (lane < kTickLevels) {
        lsc.w = lvl[lane]; lsc.a = lvl[kTickLevels + lane]; lsc.pe = lvl[2 * kTickLevels + lane];
        lsc.plogp = lvl[3 * kTickLevels + lane];''', '''    if (lane_id() < kTickLevels) {   // every wave holds the level scalars on its own lanes
        lsc.w = lvl[lane_id()]; lsc.a = lvl[kTickLevels + lane_id()]; lsc.pe = lvl[2 * kTickLevels + lane_id()];
        lsc.plogp = lvl[3 * kTickLevels + lane_id()];''')
rep('''        rng_normals(rng, d, lds, lds + dpad);
        double p0[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int e = lane * NS + s;
            const double z = (e < d) ? lds[e] : 0.0;''', '''        double zz[NS];
        wide_normals_regs<NS>(tm, rng, d, lds, bcast, zz);   // wave 0 draws (numpy's stream is sequential), 1024 at a time
        double p0[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const double z = zz[s];''')
rep("        wave_sync();\n        logp0 = logp_new;", "        tm.sync();\n        logp0 = logp_new;")
rep("window_next(rng, win)", "team_uniform(tm, rng, win)", count=5)
# the Welford scalars are read by every wave and rewritten by thread 0: all reads first
rep("            if (lane == 0) {\n                A.n_samples[c] = ms.n_samples;",
    "            tm.sync();   // every wave has read the estimator scalars thread 0 rewrites\n"
    "            if (lane == 0) {\n                A.n_samples[c] = ms.n_samples;")

b2 = begin
b2 = rep('''template <int NS>
__global__ __launch_bounds__(64) void tick_begin_kernel(ChainArrays A, TickArrays K, long long iter_begin) {''', '''template <int NS>
__global__ __launch_bounds__(kWideThreads, 1) void tick_wide_begin_kernel(ChainArrays A, TickArrays K, long long iter_begin) {''', text=b2)
b2 = rep("    const int lane = lane_id();", "    const int lane = static_cast<int>(threadIdx.x);", text=b2)

header = '''// GENERATED from lmc_tick.hpp by tools/gen_tick_wide.py -- do not edit; edit lmc_tick.hpp / the generator and re-run it.
// The tick state machine (externally evaluated log-densities: a Python callable, a batched torch callable) for the shapes
// of the general kernels: one chain = a workgroup of 16 wavefronts (lmc_wide.hpp), model_ndim up to 16 384, diagonal mass
// matrices. Statement for statement the one-wavefront tick kernel; what differs is who "lane" is (the thread's index in its
// chain), the team's reductions and barriers, the normals drawn 1024 at a time by wave 0, and the uniform stream shared by
// the team.
#pragma once
#include "lmc_tick_launch.hpp"
#include "lmc_wide.hpp"

namespace lmc {

'''
TEXT = header + body + b2 + "}  // namespace lmc\n"
if __name__ == "__main__":
    with open(OUT, "w") as fh:
        fh.write(TEXT)
    print("wrote", OUT)
