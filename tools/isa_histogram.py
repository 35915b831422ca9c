#!/usr/bin/env python3
"""Static ISA census of one kernel of a gfx950 .s file (hipcc -save-temps): per basic block, instruction counts by
class, with the loop depth hipcc annotates. Used for profiles/rNN_run_kernel_isa_histogram.json.

    python tools/isa_histogram.py file.s '<mangled-name-substring>' [--loop LBBx_y] [--json out.json]
"""
import argparse
import collections
import json
import re
import sys


def classify(op, line):
    if op.startswith("v_mov_b32") and ("row_" in line or "wave_" in line or "quad_perm" in line):
        return "v_mov_dpp"
    if op.startswith("v_mov_b"):
        return "v_mov"
    if op.startswith("v_permlane"):
        return "v_permlane_swap"
    if op.startswith("v_readlane") or op.startswith("v_readfirstlane"):
        return "v_readlane"
    if op.startswith("v_writelane"):
        return "v_writelane"
    if op.startswith("v_cndmask"):
        return "v_cndmask"
    if op.startswith("v_cmp"):
        return "v_cmp"
    if re.match(r"v_(fma|fmac|mul|add|min|max|rndne|ldexp|rcp|rsq|sqrt|div_\w+|trig_preop|frexp\w*|cvt\w*)_f64", op) or op.endswith("_f64_e32") or op.endswith("_f64_e64") or "_f64" in op:
        return "v_f64_arith"
    if op.startswith("v_"):
        return "v_other"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_") or op.startswith("flat_") or op.startswith("buffer_"):
        return "vmem"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op == "s_nop":
        return "s_nop"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "s_branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("kernel", help="substring of the mangled kernel name")
    ap.add_argument("--loop", default=None, help="only blocks annotated as inside this loop header (e.g. BB16_196)")
    ap.add_argument("--json", default=None)
    ap.add_argument("--blocks", action="store_true", help="print one line per basic block")
    args = ap.parse_args()

    lines = open(args.asm).read().split("\n")
    start = end = None
    for i, l in enumerate(lines):
        if "-- Begin function" in l and args.kernel in l:
            start = i
        if start is not None and end is None and l.strip().startswith(".Lfunc_end"):
            end = i
            break
    if start is None:
        sys.exit("kernel not found")
    body = lines[start:end]
    blocks = []          # (label, in_loop_header, depth, Counter)
    cur = {"label": "entry", "loop": None, "depth": 0, "count": collections.Counter(), "line": start}
    blocks.append(cur)
    pending_label = None
    for off, l in enumerate(body):
        s = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        m2 = re.match(r"^; %bb\.(\d+):", s)
        if m or m2:
            cur = {"label": m.group(1) if m else "bb." + m2.group(1), "loop": None, "depth": 0,
                   "count": collections.Counter(), "line": start + off}
            blocks.append(cur)
        mm = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", l)
        if mm and not cur["count"]:
            cur["loop"], cur["depth"] = mm.group(1), int(mm.group(2))
        mm = re.search(r"This (?:Inner )?Loop Header: Depth=(\d+)", l)
        if mm and not cur["count"]:
            cur["loop"], cur["depth"] = cur["label"].lstrip(".L"), int(mm.group(1))
        if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
            continue
        op = s.split()[0]
        if not re.match(r"^[a-z]", op):
            continue
        cur["count"][classify(op, s)] += 1
    sel = [b for b in blocks if args.loop is None or b["loop"] == args.loop]
    total = collections.Counter()
    for b in sel:
        total.update(b["count"])
        if args.blocks and b["count"]:
            print("%-14s line %6d depth %d %s" % (b["label"], b["line"] + 1, b["depth"],
                                                 " ".join("%s=%d" % kv for kv in sorted(b["count"].items()))))
    valu = sum(v for k, v in total.items() if k.startswith("v_"))
    print("blocks %d  VALU %d  %s" % (len(sel), valu, dict(sorted(total.items()))))
    # per-loop totals (innermost loop each block belongs to), the kernel's resource metadata and the spill census
    by_loop = collections.OrderedDict()
    for b in blocks:
        key = "%s depth %d" % (b["loop"] or "straight-line", b["depth"])
        by_loop.setdefault(key, collections.Counter()).update(b["count"])
    meta = {}
    text = "\n".join(lines)
    ky = text.find("amdhsa.kernels:")
    if ky >= 0:      # one YAML list entry per kernel; take the one whose .name holds the requested substring
        for entry in re.split(r"\n  - ", text[ky:]):
            mname = re.search(r"\.name:\s*(\S+)", entry)
            if mname and args.kernel in mname.group(1):
                for m in re.finditer(r"\.(vgpr_count|agpr_count|sgpr_count|private_segment_fixed_size|vgpr_spill_count|"
                                     r"sgpr_spill_count|group_segment_fixed_size|max_flat_workgroup_size):\s*(\d+)", entry):
                    meta[m.group(1)] = int(m.group(2))
                break
    if args.json:
        json.dump({"kernel": args.kernel, "loop": args.loop, "static_counts": dict(total), "valu_static": valu,
                   "metadata": meta,
                   "by_loop": {k: dict(sorted(v.items())) for k, v in by_loop.items() if v},
                   "blocks": [{"label": b["label"], "depth": b["depth"], "loop": b["loop"], "counts": dict(b["count"])}
                              for b in sel if b["count"]]},
                  open(args.json, "w"), indent=1)

if __name__ == "__main__":
    main()
