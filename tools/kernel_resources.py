#!/usr/bin/env python3
"""Compile the HIP library with -Rpass-analysis=kernel-resource-usage and print one line per kernel."""
import re
import subprocess
import sys

args = sys.argv[1:]
unit = "lmc_engine"
if args and args[0] in ("lmc_engine", "lmc_dense"):   # translation unit to analyse (default: the diagonal kernels)
    unit = args.pop(0)
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-c", "-mllvm", "-disable-machine-licm",
       "-I", "littlemcmc_amd/csrc", "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/lmc_kres.o",
       "littlemcmc_amd/csrc/%s.hip" % unit] + args
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    for key in ("TotalSGPRs", "VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"):
        m = re.search(re.escape(key) + r": (\d+)", line)
        if m and cur is not None and key not in cur:
            cur[key] = int(m.group(1))
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("void lmc::", "").replace("lmc::", "")
    print("%-46s sgpr %3d vgpr %3d agpr %3d scratch %5d occ %d" % (
        name[:46], r.get("TotalSGPRs", -1), r.get("VGPRs", -1), r.get("AGPRs", -1),
        r.get("ScratchSize [bytes/lane]", -1), r.get("Occupancy [waves/SIMD]", -1)))
