#!/bin/bash
# GPU box: kernel-trace stats and HBM counters of the tick path (tools/bench_torch_target.py) -> gpurun_out/<dir>
set -u
out=gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python tools/bench_torch_target.py 65536 128 60"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o stats -- $CMD > $out/bench.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --output-format csv -d $out -o pmc_$c -- $CMD > /dev/null 2>&1
done
python - $out <<'PY'
import csv, glob, sys, json
out = sys.argv[1]
res = {}
for f in glob.glob(out + "/**/stats_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "tick_kernel" in r["Name"]:
            res["tick_kernel_calls"] = int(r["Calls"]); res["tick_kernel_avg_us"] = float(r["AverageNs"]) / 1e3
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    tot = n = 0
    for f in glob.glob(out + "/**/pmc_%s_counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if "tick_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                tot += float(r["Counter_Value"]); n += 1
    res[c + "_KB_per_tick"] = tot / max(n, 1); res[c + "_dispatches"] = n
chains = 65536
res["hbm_bytes_per_chain_per_tick"] = (2 * res["FETCH_SIZE_KB_per_tick"] + res["WRITE_SIZE_KB_per_tick"]) * 1024 / chains
res["hbm_TBps_during_tick_kernel"] = (2 * res["FETCH_SIZE_KB_per_tick"] + res["WRITE_SIZE_KB_per_tick"]) * 1024 / (res["tick_kernel_avg_us"] * 1e-6) / 1e12
res["note"] = "FETCH_SIZE doubled per the gfx950 note (wide coalesced reads); per tick of 65536 chains x d=128 AR(1), NUTS while tuning"
json.dump(res, open(out + "/tick_summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
grep -i "leapfrog" $out/bench.log | tail -4
find $out -name "*.csv" -size +1M -delete
