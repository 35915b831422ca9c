#!/usr/bin/env python3
"""Static instruction census of ONE kernel by source location.

    hipcc ... -gline-tables-only -S --cuda-device-only -o /tmp/isa/eng_g.s littlemcmc_amd/csrc/lmc_engine.hip
    python tools/isa_by_line.py /tmp/isa/eng_g.s _ZN3lmc10run_kernelILi2ELi1ENS_15StdNormalTargetE [--by func|line]

Every instruction is attributed to the innermost .loc (file:line) in force; lines are grouped into the source functions
of lmc_sampler.hpp / lmc_rng.hpp / lmc_wave.hpp / lmc_targets.hpp by line ranges read from the sources. The counts are
static: code outside the tree loops executes once per iteration, so for that part static ~ dynamic (the polar-method
loops of rng_normals run ~2 rounds at d = 128)."""
import collections
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "littlemcmc_amd", "csrc")


def function_ranges(path):
    """[(first_line, name)] of top-level-ish function definitions (heuristic: a line with '(' ending in '{' at brace depth <= 1)."""
    out = []
    rx = re.compile(r"^\s*(?:template\s*<[^>]*>\s*)?(?:__device__|__global__|static|inline|__forceinline__|constexpr|__host__|\s)*"
                    r"[\w:<>\*&\s]+?\b(\w+)\s*\([^;]*$")
    lines = open(path).read().split("\n")
    for i, l in enumerate(lines, 1):
        if ("__device__" in l or "__global__" in l) and "(" in l:
            m = re.search(r"(\w+)\s*\(", l.split("__device__")[-1] if "__device__" in l else l)
            if m:
                out.append((i, m.group(1)))
    return out


def main():
    asm, kernel = sys.argv[1], sys.argv[2]
    by = "func"
    if "--by" in sys.argv:
        by = sys.argv[sys.argv.index("--by") + 1]
    files = {}
    ranges = {}
    cur = None
    on = False
    depth = 0
    counts = collections.defaultdict(lambda: collections.Counter())
    for l in open(asm):
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', l)
        if m:
            files[int(m.group(1))] = m.group(3)
            continue
        if "Begin function " + kernel in l:
            on = True
            continue
        if not on:
            continue
        if ".Lfunc_end" in l:
            break
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
            continue
        dm = re.search(r"Depth=(\d+)", l)
        if re.match(r"^\.LBB", l) or re.match(r"^; %bb\.\d+:", l):   # (fall-through blocks carry their loop in a comment line only)
            depth = int(dm.group(1)) if dm else 0
        t = l.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        op = t.split()[0]
        if op.startswith(("v_", "ds_", "global_", "scratch_", "buffer_", "flat_")):
            kind = "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else "vmem"
        elif op.startswith("s_"):
            kind = "smem" if op.startswith(("s_load", "s_buffer_load")) else "wait" if op.startswith(("s_waitcnt", "s_nop")) else "salu"
        else:
            kind = "other"
        key = cur
        if by == "func" and cur is not None:
            fn = os.path.basename(cur[0])
            path = os.path.join(CSRC, fn)
            if os.path.exists(path):
                if fn not in ranges:
                    ranges[fn] = function_ranges(path)
                name = "?"
                for first, nm in ranges[fn]:
                    if first <= cur[1]:
                        name = nm
                key = (fn, name)
            else:
                key = (fn, "-")
        counts[(key, min(depth, 9))][kind] += 1
    rows = sorted(counts.items(), key=lambda kv: -sum(kv[1].values()))
    tot = collections.Counter()
    print("%-52s %5s %6s %6s %5s %5s %5s %5s" % ("source", "depth", "valu", "salu", "lds", "vmem", "smem", "wait"))
    for (key, depth), c in rows:
        tot.update(c)
        if sum(c.values()) < 8:
            continue
        print("%-52s %5d %6d %6d %5d %5d %5d %5d" % ("%s:%s" % key if key else "?", depth, c["valu"], c["salu"], c["lds"], c["vmem"], c["smem"], c["wait"]))
    print("%-52s %5s %6d %6d %5d %5d %5d %5d" % ("TOTAL", "", tot["valu"], tot["salu"], tot["lds"], tot["vmem"], tot["smem"], tot["wait"]))


if __name__ == "__main__":
    main()
