#!/bin/bash
# Kernel-trace stats of one bench command: tools/quick_stats.sh <outdir-under-gpurun_out> [bench args...]
set -u
out=gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o stats -- python bench.py --no-cpu-baseline --no-ess --no-secondary $* > $out/bench.log 2>&1
grep '"metric"' $out/bench.log > $out/bench.json
f=$(find $out -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print("%-60s calls %6s total %10.2f ms avg %10.1f us  %5s%%" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
find $out -name "*trace.csv" -size +2M -delete
