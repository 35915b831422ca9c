#!/usr/bin/env python3
"""gpurun_out/<dir> (tools/profile_round.sh output) -> profiles/<tag>_* summaries + profiles/pmc_counters.json

    python tools/summarize_profile.py gpurun_out/<dir> <tag> <key> [kernel-substring]
    e.g.  python tools/summarize_profile.py gpurun_out/prof_r02 r02_c3 ar1:128 AR1Target

The bench line captured under the profiler carries the build's source hash; it is stored with the counters, and
bench.py quotes counter-derived figures only for the build they were measured on."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, tag, key = sys.argv[1], sys.argv[2], sys.argv[3]
kern_sub = sys.argv[4] if len(sys.argv) > 4 else "run_kernel"
d = src.rstrip("/") + "/"


def find(pattern):
    hits = sorted(glob.glob(d + "**/" + pattern, recursive=True))
    if not hits:
        raise SystemExit("missing %s under %s" % (pattern, d))
    return hits[0]


b = json.loads(open(d + "bench_stats.json").read())
shutil.copy(find("stats_kernel_stats.csv"), "profiles/%s_kernel_stats.csv" % tag)
# the per-dispatch trace is large for long runs: keep the kernel's own rows only
rows = list(csv.DictReader(open(find("stats_kernel_trace.csv"))))
keep = [r for r in rows if kern_sub in r["Kernel_Name"] and "run_" in r["Kernel_Name"]]
with open("profiles/%s_kernel_trace.csv" % tag, "w", newline="") as fh:
    w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(keep)
# rocprofv3 on gfx950 reports VGPR_Count per 32 lanes (84 for the 168-register wave64 kernel that
# -Rpass-analysis=kernel-resource-usage and the ISA metadata show): double it
vgpr = 2 * (int(keep[0]["VGPR_Count"]) + int(keep[0].get("Accum_VGPR_Count", 0) or 0))
alloc = (vgpr + 7) // 8 * 8
waves_per_simd = min(8, 512 // alloc)
dur_ns = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in keep]

out = {"command": "rocprofv3 --kernel-trace --stats / --pmc <one group per pass> -- python bench.py --no-cpu-baseline --no-ess "
                  "--no-secondary ... (see bench_line_under_profiler.config)",
       "bench_line_under_profiler": {k: b[k] for k in ("value", "leapfrogs", "wall_s", "ms_per_step", "steps", "warmup")},
       "source_hash": b.get("source_hash"), "workload": b["config"]["workload"], "counters": {},
       "kernel_trace": {"dispatches": len(keep), "dispatches_per_step": int((b.get("roofline") or {}).get("dispatches_per_step", 1)), "avg_ms": sum(dur_ns) / len(dur_ns) / 1e6, "vgpr": vgpr,
                        "scratch_bytes_per_lane": int(keep[0]["Scratch_Size"]), "static_lds_bytes_per_block": int(keep[0]["LDS_Block_Size"]),
                        "dynamic_lds_bytes_per_block": (b.get("tail") or {}).get("lds_bytes_per_workgroup"),
                        "resident_chains": (b.get("tail") or {}).get("resident_chains"),
                        "mean_wave_slot_occupancy": (b.get("tail") or {}).get("mean_wave_slot_occupancy"),
                        "waves_per_simd_by_vgpr": waves_per_simd}}
n_disp = None
for f in ["pmc_sq", "pmc_fetch", "pmc_write", "pmc_mem", "pmc_wait"]:
    agg = collections.defaultdict(float)
    disp = set()
    try:
        path = find(f + "_counter_collection.csv")
    except SystemExit:
        if f == "pmc_wait":   # optional pass (counters that may not exist on every rocprofv3 build)
            continue
        raise
    for r in csv.DictReader(open(path)):
        if kern_sub in r["Kernel_Name"] and "run_" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            disp.add(r["Dispatch_Id"])
            out["kernel"] = r["Kernel_Name"].split("(")[0]
    n_disp = len(disp)
    out["counters"].update(agg)
out["dispatches"] = n_disp
dps = int((b.get("roofline") or {}).get("dispatches_per_step", 1))   # the engine launches a step as dps concurrent sub-block dispatches
if b["warmup"] == 0:   # every dispatch of the kernel belongs to the timed job (dense kernels: launches of different lengths)
    leap_total = b["leapfrogs"]
else:
    leap_total = b["leapfrogs"] * n_disp / (b["steps"] * dps)    # warm-up launches are the same size as timed ones
c = out["counters"]
out["per_leapfrog"] = {k: v / leap_total for k, v in c.items()}
fetch_b, write_b = c["FETCH_SIZE"] * 1024, c["WRITE_SIZE"] * 1024
dim = int(key.split(":")[1])
mass = key.split(":")[2] if len(key.split(":")) > 2 and key.split(":")[2] in ("full", "full_adapt") else "diag"
out["hbm"] = {
    "note": "gfx950 rocprofv3: FETCH_SIZE tallies 128-B requests at 64 B for wide (16 B/lane) coalesced reads -> doubled "
            "(guides/MI355X_MICROARCH.md, HBM section); WRITE_SIZE uncalibrated, taken as is; separate --pmc passes",
    "read_bytes_per_leapfrog_corrected": 2 * fetch_b / leap_total, "write_bytes_per_leapfrog": write_b / leap_total,
    "hbm_bytes_per_leapfrog": (2 * fetch_b + write_b) / leap_total, "algorithmic_bytes_per_leapfrog": 60 * dim + (0 if mass == "diag" else 8 * dim * dim),
    "hbm_bytes_per_launch": (2 * fetch_b + write_b) / n_disp * dps}
wc = c["SQ_WAVE_CYCLES"]
out["wave_time_split"] = {"valu_active": c["SQ_ACTIVE_INST_VALU"] / wc, "wait_inst_any": c["SQ_WAIT_INST_ANY"] / wc,
                          "wait_any": c["SQ_WAIT_ANY"] / wc}
out["simd_valu_busy"] = min(1.0, waves_per_simd * out["wave_time_split"]["valu_active"])
json.dump(out, open("profiles/%s_pmc_summary.json" % tag, "w"), indent=1)
path = "profiles/pmc_counters.json"
tr = json.load(open(path)) if os.path.exists(path) else {}
tr[key] = {"hbm_bytes_per_leapfrog": out["hbm"]["hbm_bytes_per_leapfrog"],
           "valu_inst_per_leapfrog": out["per_leapfrog"]["SQ_INSTS_VALU"],
           "salu_inst_per_leapfrog": out["per_leapfrog"]["SQ_INSTS_SALU"],
           "simd_valu_busy": out["simd_valu_busy"], "source_hash": out["source_hash"],
           "source": "profiles/%s_pmc_summary.json (rocprofv3 --pmc in separate passes; FETCH_SIZE doubled per the gfx950 note; "
                     "simd_valu_busy = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES x %d waves per SIMD)" % (tag, waves_per_simd)}
json.dump(tr, open(path, "w"), indent=1)
print(json.dumps({"per_leapfrog": out["per_leapfrog"], "hbm": out["hbm"], "split": out["wave_time_split"],
                  "kernel_trace": out["kernel_trace"], "source_hash": out["source_hash"]}, indent=1))
