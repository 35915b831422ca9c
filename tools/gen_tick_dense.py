#!/usr/bin/env python3
"""Generate littlemcmc_amd/csrc/lmc_tick_dense.hpp from lmc_tick.hpp by targeted text transformations, so that the
tick state machine with a dense mass matrix stays statement-parallel to the diagonal one. Every transformation asserts
that its anchor occurs exactly once: if lmc_tick.hpp changes shape this script fails loudly instead of producing a
silently different kernel. Run from the repo root: python tools/gen_tick_dense.py"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "littlemcmc_amd/csrc/lmc_tick_dense.hpp")
src = open(os.path.join(ROOT, "littlemcmc_amd/csrc/lmc_tick.hpp")).read()
body = src[src.index("// register budget per vector width"):src.index("// chains that still want evaluations")]


def rep(a, b):
    global body
    assert body.count(a) == 1, (body.count(a), a[:80])
    body = body.replace(a, b)


rep('''// register budget per vector width (waves per SIMD): the tick kernel is latency / bandwidth bound and insensitive to
// occupancy (4 / 6 / 8 waves measured equal at NS = 2), so wide vectors simply get the registers they need
constexpr int tick_waves_per_simd(int ns) { return ns <= 2 ? 4 : ns == 4 ? 2 : 1; }
template <int NS>
__global__ __launch_bounds__(64, tick_waves_per_simd(NS)) void tick_kernel(ChainArrays A, TickArrays K, SamplerParams P, const double* logp_in,
                                                  const double* grad_in) {''', '''template <int NS, class MatT>
__global__ __launch_bounds__(64, dense_waves_per_simd(NS)) void tick_dense_kernel(ChainArrays A, DenseArrays D, TickArrays K,
                                                                                  SamplerParams P, const double* logp_in,
                                                                                  const double* grad_in, int* adapt_mask) {''')
rep("    auto level = [&](int j, int k) { return scr + (9 + 4 * j + k) * dpad; };",
    "    auto level = [&](int j, int k) { return scr + (kTickDenseFixedSlots + 6 * j + k) * dpad; };\n"
    "    const MatT* M = static_cast<const MatT*>(D.covT) + static_cast<long long>(c) * D.mat_stride;\n"
    "    DenseMat<MatT> mm{M, nullptr, 0, d, dpad};\n    lds_double* xop = (lds_double*)lds;")
rep('''    float var[NS], inv_std[NS];
    double vard[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        var[s] = A.var[row + lane * NS + s];
        inv_std[s] = A.inv_std[row + lane * NS + s];
        vard[s] = static_cast<double>(var[s]);
    }
''', '')
rep("    double cq[NS], cp[NS], cg[NS];          // the state the next leapfrog starts from",
    "    double cq[NS], cp[NS], cg[NS], cv[NS], cw[NS];   // the state the next leapfrog starts from: q, p, g, v = C p, w = C g")
rep("    double tlp[NS], trp[NS], tps[NS], tq[NS];", "    double tlp[NS], tlv[NS], trp[NS], trv[NS], tps[NS], tq[NS];")
rep('''        rng_normals(rng, d, lds, lds + dpad);
        double p0[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int e = lane * NS + s;
            const double z = (e < d) ? lds[e] : 0.0;
            p0[s] = momentum_f32 ? static_cast<double>(inv_std[s] * static_cast<float>(z)) : z * static_cast<double>(inv_std[s]);
        }
        wave_sync();
        logp0 = logp_new;
        if (momentum_f32) {
            const float kin = start_kinetic_f32<NS>(tm, p0, var, d, P.sdot_mode, reinterpret_cast<float*>(lds), dpad);
            e0 = first_f64(static_cast<double>(kin) - logp0);
        } else {
            e0 = first_f64(0.5 * tm.sum(pdot_v<NS>(p0, vard, p0)) - logp0);
        }''', '''        rng_normals(rng, d, lds, lds + dpad);
        double p0[NS];
        if (D.kind == kDenseFullInv)
            dense_momentum_inv<NS>(static_cast<const double*>(D.fac), d, dpad, xop, p0);
        else
            dense_momentum_full<NS>(static_cast<const float*>(D.fac) + static_cast<long long>(c) * D.fac_stride, d, dpad, xop, p0);
        logp0 = logp_new;
        double v0[NS], w0[NS], v0s[NS];
        e0 = dense_start_state<NS, MatT>(tm, mm, lds, momentum_f32, P.sdot_mode, p0, g0, logp0, v0, w0, v0s);''')
rep('''#pragma unroll
            for (int r = 0; r < 2; ++r) {
                vstore_as<NS>(slot(3 * r + 0), q); vstore_as<NS>(slot(3 * r + 1), p0); vstore_as<NS>(slot(3 * r + 2), g0);
            }
            vstore_as<NS>(slot(6), p0); vstore_as<NS>(slot(7), q);
            l_start = momentum_f32; r_start = momentum_f32;''', '''#pragma unroll
            for (int r = 0; r < 2; ++r) {
                vstore_as<NS>(slot(5 * r + 0), q); vstore_as<NS>(slot(5 * r + 1), p0); vstore_as<NS>(slot(5 * r + 2), g0);
                vstore_as<NS>(slot(5 * r + 3), v0); vstore_as<NS>(slot(5 * r + 4), w0);
            }
            vstore_as<NS>(slot(kSlotPsum), p0); vstore_as<NS>(slot(kSlotProp), q); vstore_as<NS>(slot(kSlotV0s), v0s);
            l_start = true; r_start = true;   // the end still is the start state: its stored velocity is v0s''')
rep("            vcopy(cq, q); vcopy(cp, p0); vcopy(cg, g0);\n            need_leap = true;",
    "            vcopy(cq, q); vcopy(cp, p0); vcopy(cg, g0); vcopy(cv, v0); vcopy(cw, w0);\n            need_leap = true;")
rep('''        vload_as<NS>(slot(8), half);
        const double dt = 0.5 * eps;
        double kin = 0.0;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            cp[s] = half[s] + dt * cg[s];
            kin = __builtin_fma(cp[s], vard[s] * cp[s], kin);
        }
        const double energy = first_f64(0.5 * tm.sum(kin) - logp_new);''', '''        vload_as<NS>(slot(kSlotHalf), half);
        const double dt = 0.5 * eps;
#pragma unroll
        for (int s = 0; s < NS; ++s) cp[s] = half[s] + dt * cg[s];
        velocity2<NS, MatT>(mm, xop, cp, cg, cv, cw);   // the one matrix sweep of this leapfrog: v = C p, w = C g
        const double energy = first_f64(0.5 * tm.sum(pdot<NS>(cp, cv)) - logp_new);''')
rep("                vcopy(tlp, cp); vcopy(trp, cp); vcopy(tps, cp); vcopy(tq, cq);",
    "                vcopy(tlp, cp); vcopy(trp, cp); vcopy(tps, cp); vcopy(tlv, cv); vcopy(trv, cv); vcopy(tq, cq);")
rep('''                    double alp[NS], arp[NS], aps[NS], aq[NS];
                    double aw, aa, ape, aplogp;
                    vload_as<NS>(level(j, 0), alp); vload_as<NS>(level(j, 1), arp);
                    vload_as<NS>(level(j, 2), aps); vload_as<NS>(level(j, 3), aq);''', '''                    double alp[NS], alv[NS], arp[NS], arv[NS], aps[NS], aq[NS];
                    double aw, aa, ape, aplogp;
                    vload_as<NS>(level(j, 0), alp); vload_as<NS>(level(j, 1), alv); vload_as<NS>(level(j, 2), arp);
                    vload_as<NS>(level(j, 3), arv); vload_as<NS>(level(j, 4), aps); vload_as<NS>(level(j, 5), aq);''')
rep('''                        double dots[6] = {pdot_v<NS>(ps, vard, alp), pdot_v<NS>(ps, vard, trp), pdot_v<NS>(p1, vard, alp),
                                          pdot_v<NS>(p1, vard, tlp), pdot_v<NS>(p2, vard, arp), pdot_v<NS>(p2, vard, trp)};''', '''                        double dots[6] = {pdot<NS>(ps, alv), pdot<NS>(ps, trv), pdot<NS>(p1, alv),
                                          pdot<NS>(p1, tlv), pdot<NS>(p2, arv), pdot<NS>(p2, trv)};''')
rep("                        turn = tm.any_nonpositive2(pdot_v<NS>(ps, vard, alp), pdot_v<NS>(ps, vard, trp));",
    "                        turn = tm.any_nonpositive2(pdot<NS>(ps, alv), pdot<NS>(ps, trv));")
rep("                    vcopy(tlp, alp); vcopy(tps, ps);\n", "                    vcopy(tlp, alp); vcopy(tlv, alv); vcopy(tps, ps);\n")
rep('''                        vstore_as<NS>(level(j, 0), tlp); vstore_as<NS>(level(j, 1), trp);
                        vstore_as<NS>(level(j, 2), tps); vstore_as<NS>(level(j, 3), tq);''', '''                        vstore_as<NS>(level(j, 0), tlp); vstore_as<NS>(level(j, 1), tlv); vstore_as<NS>(level(j, 2), trp);
                        vstore_as<NS>(level(j, 3), trv); vstore_as<NS>(level(j, 4), tps); vstore_as<NS>(level(j, 5), tq);''')
rep("                vload_as<NS>(slot(6), psum); vload_as<NS>(slot(7), propq);",
    "                vload_as<NS>(slot(kSlotPsum), psum); vload_as<NS>(slot(kSlotProp), propq);")
rep("                    vstore_as<NS>(slot(7), propq);", "                    vstore_as<NS>(slot(kSlotProp), propq);")
rep("                vstore_as<NS>(slot(6), psum);", "                vstore_as<NS>(slot(kSlotPsum), psum);")
rep('''                double Lp[NS], Rp[NS], oLv[NS], oRv[NS], vtl[NS], vtr[NS];
                vload_as<NS>(slot(1), Lp); vload_as<NS>(slot(4), Rp);
                end_velocity<NS>(oLv, vard, Lp, l_start);
                end_velocity<NS>(oRv, vard, Rp, r_start);
#pragma unroll
                for (int s = 0; s < NS; ++s) { vtl[s] = vard[s] * tlp[s]; vtr[s] = vard[s] * trp[s]; }''', '''                double Lp[NS], Rp[NS], oLv[NS], oRv[NS], vtl[NS], vtr[NS];
                vload_as<NS>(slot(1), Lp); vload_as<NS>(slot(6), Rp);
                vload_as<NS>(slot(l_start ? kSlotV0s : 3), oLv);
                vload_as<NS>(slot(r_start ? kSlotV0s : 8), oRv);
                vcopy(vtl, tlv); vcopy(vtr, trv);''')
rep("                vstore_as<NS>(slot(3 * side + 0), cq); vstore_as<NS>(slot(3 * side + 1), cp); vstore_as<NS>(slot(3 * side + 2), cg);\n                if (tm.any_nonpositive6(dots))",
    "                vstore_as<NS>(slot(5 * side + 0), cq); vstore_as<NS>(slot(5 * side + 1), cp); vstore_as<NS>(slot(5 * side + 2), cg);\n"
    "                vstore_as<NS>(slot(5 * side + 3), cv); vstore_as<NS>(slot(5 * side + 4), cw);\n                if (tm.any_nonpositive6(dots))")
rep("        vload_as<NS>(slot(3 * side + 0), cq); vload_as<NS>(slot(3 * side + 1), cp); vload_as<NS>(slot(3 * side + 2), cg);\n        leaf = 0;",
    "        vload_as<NS>(slot(5 * side + 0), cq); vload_as<NS>(slot(5 * side + 1), cp); vload_as<NS>(slot(5 * side + 2), cg);\n"
    "        vload_as<NS>(slot(5 * side + 3), cv); vload_as<NS>(slot(5 * side + 4), cw);\n        leaf = 0;")
rep('''            half[s] = cp[s] + dt * cg[s];
            const double v = vard[s] * half[s];
            qn[s] = cq[s] + eps * v;
        }
        vstore_as<NS>(slot(8), half);''', '''            half[s] = cp[s] + dt * cg[s];
            const double v = cv[s] + dt * cw[s];   // C (p + dt g)
            qn[s] = cq[s] + eps * v;
        }
        vstore_as<NS>(slot(kSlotHalf), half);''')
rep("            vload_as<NS>(slot(7), q);\n            out.accept = (wn > 0.0)", "            vload_as<NS>(slot(kSlotProp), q);\n            out.accept = (wn > 0.0)")
k0 = body.index("        if (tune && P.adapt_mass) {\n            MassScalars ms;")
k1 = body.index("        ++iter_count;\n        if (A.mom_mean != nullptr && !tune)")
body = body[:k0] + ("        // FullAdapt.update for this chain runs in dense_adapt_kernel right after this tick (the host launches it masked)\n"
                    "        if (tune && D.kind == kDenseFullAdapt && lane == 0) adapt_mask[c] = 1;\n") + body[k1:]
hdr = '''// The tick kernel (lmc_tick.hpp) with a dense mass matrix: densities evaluated by the caller (targets.TorchTarget)
// sampled with QuadPotentialFull / FullInv / FullAdapt. Same state machine, cut at the density evaluation; the
// differences are the ones between lmc_sampler.hpp and lmc_dense.hpp: velocities are matrix sweeps and therefore
// stored with the trajectory ends and tree nodes, one sweep per leapfrog forms v = C p and w = C g, the momentum is
// a triangular solve (or L n), and FullAdapt's update of a chain that finished a tuning iteration in this tick runs
// in dense_adapt_kernel, launched masked by the host between two ticks. GENERATED from lmc_tick.hpp by
// tools/gen_tick_dense.py so that the two state machines stay statement-parallel; do not edit by hand.
#pragma once
#include "lmc_dense.hpp"
#include "lmc_tick_launch.hpp"

namespace lmc {

'''
TEXT = hdr + body + "}  // namespace lmc\n"

if __name__ == "__main__":
    open(OUT, "w").write(TEXT)
    print("wrote", OUT)
