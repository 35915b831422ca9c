#!/usr/bin/env python3
"""What a leapfrog of the NUTS pair loop costs in issued instructions, by PURPOSE (static census of the loop's own basic blocks).

    hipcc ... -gline-tables-only -S --cuda-device-only -o /tmp/isa/eng_g.s littlemcmc_amd/csrc/lmc_engine.hip
    python tools/isa_pair_loop.py /tmp/isa/eng_g.s _ZN3lmc10run_kernelILi4ELi1ENS_12FunnelTargetELi0ELi1EEE

The compiler's own loop annotation is of no use here (the pair loop of nuts_transition2 is not a natural loop after
structurisation: its blocks are labelled with the depth of the iteration loop), so the loop is recovered from the control-flow
graph: the strongly connected component that contains the second leapfrog of a pair once the blocks that close a DOUBLING
(`++depth`, lmc_sampler.hpp) are removed. Every instruction of that component is attributed to the innermost source location
in force (.loc) and put in a bucket by function / line range:

  integrate     leapfrog_partial + the density functor (the algorithm's own arithmetic: integration.py:100-121)
  reduce        red_put / red_gather: all six sums of a pair through LDS
  leaf scalars  energies, divergence / weight-offset checks, the exponential on lanes (nuts.py:344-375)
  merge         level-0 merge, cascade levels: loads of parked nodes, the six U-turn dots of a level, weights (nuts.py:384-417)
  park          storing the node that stays on the subtree stack (vectors + level scalars)
  uniforms      team_uniform / window_next / MT19937 regeneration for the merges' uniforms (math.py:21-25)
  rare          weight-offset moves, the sequential divergence path
  control       loop counters, branches, waits that belong to no statement above

Blocks are weighted by how often a steady-state pair executes them: 1 for the straight-line body, 1/2^j for cascade level j
(level 1 every second pair, ...), ~0 for the rare paths and for the generator's regeneration (once per 624 words); the weights
are heuristics read off the source structure, printed next to every bucket so that they can be disputed."""
import collections
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "littlemcmc_amd", "csrc")
sys.path.insert(0, os.path.join(ROOT, "tools"))
from isa_by_line import function_ranges  # noqa: E402


def parse(asm, kernel):
    files, blocks, order = {}, {}, []
    cur_loc, cur = None, None
    on = False
    for l in open(asm):
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', l)
        if m:
            files[int(m.group(1))] = m.group(3)
            continue
        if "Begin function " + kernel in l:
            on = True
            cur = "entry"
            blocks[cur] = []
            order.append(cur)
            continue
        if not on:
            continue
        if ".Lfunc_end" in l:
            break
        m = re.match(r"^(\.LBB\d+_\d+):", l) or re.match(r"^; %bb\.(\d+):", l)
        if m:
            cur = m.group(1) if m.group(1).startswith(".LBB") else "bb" + m.group(1)
            blocks[cur] = []
            order.append(cur)
            continue
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            cur_loc = (os.path.basename(files.get(int(m.group(1)), "?")), int(m.group(2)))
            continue
        t = l.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        blocks[cur].append((t.split()[0], t, cur_loc))
    return blocks, order


def edges(blocks, order):
    succ = collections.defaultdict(set)
    for i, b in enumerate(order):
        fall = True
        for op, t, _ in blocks[b]:
            m = re.search(r"(\.LBB\d+_\d+)", t)
            if op.startswith("s_cbranch") and m:
                succ[b].add(m.group(1))
            elif op == "s_branch" and m:
                succ[b].add(m.group(1))
                fall = False
            elif op in ("s_endpgm", "s_setpc_b64"):
                fall = False
        if fall and i + 1 < len(order):
            succ[b].add(order[i + 1])
    return succ


def scc_of(start, succ, banned):
    fwd, stack = {start}, [start]
    while stack:
        for y in succ[stack.pop()]:
            if y not in fwd and y not in banned:
                fwd.add(y)
                stack.append(y)
    pred = collections.defaultdict(set)
    for x, ys in succ.items():
        for y in ys:
            pred[y].add(x)
    bwd, stack = {start}, [start]
    while stack:
        for y in pred[stack.pop()]:
            if y not in bwd and y not in banned:
                bwd.add(y)
                stack.append(y)
    return fwd & bwd


def kind_of(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "scratch_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith(("s_waitcnt", "s_nop")):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    asm, kernel = sys.argv[1], sys.argv[2]
    src = open(os.path.join(CSRC, "lmc_sampler.hpp")).read().split("\n")
    line_of = lambda needle: next(i for i, l in enumerate(src, 1) if needle in l)  # noqa: E731
    L_second = line_of("for (int s = 0; s < NS; ++s) tps[s] = ep[s] + cp[s];")   # (arithmetic of the pair body itself: inlined calls carry the callee's lines)
    L_depth = line_of("++depth;   // nuts.py:315")
    L_pairs = line_of("for (int k = 0; k < n_pairs; ++k)")
    L_level0 = line_of("---- level-0 merge")
    L_casc1 = line_of("---- cascade level 1")
    L_cascN = line_of("---- cascade levels 2..m")
    L_park = line_of("if (k + 1 < n_pairs) {   // park the node")
    L_leafsc = line_of("auto leaf_scalars = [&]")
    L_leafsc_end = line_of("for (int dd = 0; dd < max_depth; ++dd)")
    L_rare0 = line_of("int seen = 0;")
    L_rare1 = line_of("de_x = isnan(de)")
    blocks, order = parse(asm, kernel)
    succ = edges(blocks, order)
    has = lambda b, line: any(loc == ("lmc_sampler.hpp", line) for _o, _t, loc in blocks[b])  # noqa: E731
    banned = {b for b in order if has(b, L_depth)}
    starts = [b for b in order if has(b, L_second)]
    if not starts:
        raise SystemExit("no block carries line %d (the pair's momentum sum)" % L_second)
    loop = set()
    for s_ in starts:
        loop |= scc_of(s_, succ, banned)
    ranges = {fn: function_ranges(os.path.join(CSRC, fn)) for fn in os.listdir(CSRC) if fn.endswith(".hpp")}

    def func_of(loc):
        if loc is None or loc[0] not in ranges:
            return "?"
        name = "?"
        for first, nm in ranges[loc[0]]:
            if first <= loc[1]:
                name = nm
        return name

    def bucket(loc):
        fn = func_of(loc)
        f = loc[0] if loc else "?"
        ln = loc[1] if loc else 0
        if f == "lmc_targets.hpp" or fn in ("leapfrog_partial", "leapfrog", "pdot", "pdot_v", "vcopy"):
            return "integrate", 1.0
        if f == "lmc_rng.hpp":
            if fn in ("mt_twist", "mt_regen"):
                return "uniforms: MT19937 regeneration (once per 624 words)", 2.0 / 624.0
            return "uniforms", 1.0
        if fn in ("team_uniform", "uniform_true"):
            return "uniforms", 1.0
        if fn in ("red_put", "red_gather", "red_any_nonpositive", "red_lane_init"):
            return "reduce", 1.0
        if fn in ("exp_lanes", "exp_lanes_const", "exp_uniform", "exp_uniform_fast", "fma_sgpr_addend", "sgpr_const") or f == "__clang_hip_math.h":
            return ("rare", 0.0) if fn in ("exp_uniform", "exp_uniform_fast") else ("leaf scalars", 1.0)
        if fn in ("cascade_dots", "level1_load", "levelN_load", "level_load_lp", "level_load_q", "level_scal_get_wa", "level_scal_get",
                  "glb_level_offset", "vload_as", "cold_load"):
            return "merge: cascade levels (dots, parked-node loads)", 1.0
        if fn in ("level1_store", "levelN_store", "level_scal_park", "level_scal_put", "vstore_as", "cold_store"):
            return "park", 0.5
        if f == "lmc_sampler.hpp" and fn == "nuts_transition2":
            if L_rare0 <= ln < L_rare1:
                return "rare", 0.0
            if L_leafsc <= ln < L_leafsc_end:
                return "leaf scalars", 1.0
            if L_pairs <= ln < L_level0:
                return "integrate", 1.0
            if L_level0 <= ln < L_casc1:
                return "merge: level 0", 1.0
            if L_casc1 <= ln < L_park:
                return "merge: cascade levels (dots, parked-node loads)", 1.0
            if L_park <= ln < L_depth:
                return "park", 0.5
            return "control", 1.0
        if f == "lmc_wave.hpp":
            return "lane plumbing (readlane / DPP / first_*) of the scalars above", 1.0
        return "control", 1.0

    # How often a steady-state pair executes a block: by the REGION of nuts_transition2 the block belongs to, read off the
    # function's own (non-inlined) lines in it; a block without any inherits the region of the block laid out before it.
    def region_of_line(ln):
        if L_rare0 <= ln < L_rare1:
            return "rare", 0.0
        if L_leafsc <= ln < L_leafsc_end:
            return "body", 1.0
        if L_pairs <= ln < L_casc1:
            return "body", 1.0
        if L_casc1 <= ln < L_cascN:
            return "level 1", 0.5            # every second pair closes a level-1 node
        if L_cascN <= ln < L_park:
            return "levels >= 2", 0.5        # sum over j >= 2 of 2^-j executions of the loop body per pair
        if L_park <= ln < L_depth:
            return "park", 0.5               # every second pair's node stays on the stack (the others were merged)
        return None

    region = {}
    last = ("body", 1.0)
    for b in order:
        if b not in loop:
            continue
        own = [region_of_line(loc[1]) for _o, _t, loc in blocks[b] if loc and loc[0] == "lmc_sampler.hpp" and func_of(loc) == "nuts_transition2"]
        own = [r for r in own if r is not None]
        if own:
            last = min(own, key=lambda r: r[1])   # the least frequent region a block touches is how often it runs
        region[b] = last

    tot = collections.defaultdict(collections.Counter)
    for b in order:
        if b not in loop:
            continue
        for op, _t, loc in blocks[b]:
            name, w = bucket(loc)
            rname, rw = region[b]
            if w == 1.0 or w == 0.5:
                w = rw
            tot[("%s [%s]" % (name, rname), w)][kind_of(op)] += 1
    print("kernel %s" % kernel)
    print("pair loop: %d basic blocks of %d, %d instructions (static)" % (len(loop), len(order), sum(sum(c.values()) for c in tot.values())))
    print("%-86s %6s %6s %6s %5s %5s %5s %5s | %s" % ("bucket", "weight", "valu", "salu", "lds", "vmem", "smem", "wait", "weighted, per LEAPFROG (a pair = 2)"))
    grand = 0.0
    for (name, w), c in sorted(tot.items(), key=lambda kv: -sum(kv[1].values()) * kv[0][1]):
        n = sum(c.values())
        per = 0.5 * w * n
        grand += per
        print("%-86s %6.3f %6d %6d %5d %5d %5d %5d | %6.1f" % (name, w, c["valu"], c["salu"], c["lds"], c["vmem"], c["smem"], c["wait"], per))
    print("%-86s %6s %6s %6s %5s %5s %5s %5s | %6.1f" % ("TOTAL (weighted)", "", "", "", "", "", "", "", grand))


if __name__ == "__main__":
    main()
