#!/usr/bin/env python3
"""gpurun_out/<dir> (tools/profile_round.sh output) -> profiles/<tag>_* summaries + profiles/pmc_traffic.json"""
import collections
import csv
import json
import shutil
import sys

src, tag, key = sys.argv[1], sys.argv[2], sys.argv[3]        # e.g. gpurun_out/prof_r01b r01 ar1:128
d = src.rstrip("/") + "/"
b = json.loads(open(d + "bench_stats.json").read())
shutil.copy(d + "stats_kernel_stats.csv", "profiles/%s_default_kernel_stats.csv" % tag)
shutil.copy(d + "stats_kernel_trace.csv", "profiles/%s_default_kernel_trace.csv" % tag)
n_disp = None
out = {"command": "rocprofv3 --kernel-trace --stats / --pmc <one group per pass> -- python bench.py --no-cpu-baseline --no-ess "
                  "(default workload; see bench_line_under_profiler.config)",
       "bench_line_under_profiler": {k: b[k] for k in ("value", "leapfrogs", "wall_s", "ms_per_step", "steps", "warmup")},
       "workload": b["config"]["workload"], "counters": {}}
for f in ["pmc_sq", "pmc_fetch", "pmc_write", "pmc_mem"]:
    rows = list(csv.DictReader(open(d + f + "_counter_collection.csv")))
    agg = collections.defaultdict(float)
    disp = set()
    for r in rows:
        if "run_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            disp.add(r["Dispatch_Id"])
            out["kernel"] = r["Kernel_Name"].split("(")[0]
    n_disp = len(disp)
    out["counters"].update(agg)
out["dispatches"] = n_disp
leap_total = b["leapfrogs"] * n_disp / b["steps"]          # warm-up launches are the same size as timed ones
c = out["counters"]
out["per_leapfrog"] = {k: v / leap_total for k, v in c.items()}
fetch_b, write_b = c["FETCH_SIZE"] * 1024, c["WRITE_SIZE"] * 1024
dim = int(key.split(":")[1])
out["hbm"] = {
    "note": "gfx950 rocprofv3: FETCH_SIZE tallies 128-B requests at 64 B for wide (16 B/lane) coalesced reads -> doubled "
            "(guides/MI355X_MICROARCH.md, HBM section); WRITE_SIZE uncalibrated, taken as is; separate --pmc passes",
    "read_bytes_per_leapfrog_corrected": 2 * fetch_b / leap_total, "write_bytes_per_leapfrog": write_b / leap_total,
    "hbm_bytes_per_leapfrog": (2 * fetch_b + write_b) / leap_total, "algorithmic_bytes_per_leapfrog": 60 * dim,
    "hbm_bytes_per_launch": (2 * fetch_b + write_b) / n_disp}
wc = c["SQ_WAVE_CYCLES"]
out["wave_time_split"] = {"valu_active": c["SQ_ACTIVE_INST_VALU"] / wc, "wait_inst_any": c["SQ_WAIT_INST_ANY"] / wc,
                          "wait_any": c["SQ_WAIT_ANY"] / wc}
json.dump(out, open("profiles/%s_default_pmc_summary.json" % tag, "w"), indent=1)
try:
    tr = json.load(open("profiles/pmc_traffic.json"))
except Exception:
    tr = {}
tr[key] = {"hbm_bytes_per_leapfrog": out["hbm"]["hbm_bytes_per_leapfrog"],
           "source": "profiles/%s_default_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; "
                     "FETCH_SIZE doubled per the gfx950 note)" % tag}
json.dump(tr, open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps({"per_leapfrog": out["per_leapfrog"], "hbm": out["hbm"], "split": out["wave_time_split"]}, indent=1))
