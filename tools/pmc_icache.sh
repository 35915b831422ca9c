#!/bin/bash
# GPU box: instruction-cache counters of one bench command, summed for the kernel matching $KERNEL (default run_kernel):
#   tools/pmc_icache.sh <outdir-under-gpurun_out> [bench args...]
# (run_kernel<2,1,...> is ~65 KB of code; two CUs share one 64 KB instruction cache)
set -u
out=gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BENCH="python bench.py --no-cpu-baseline --no-ess --no-secondary --no-rccl-check --no-tail $*"
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $set --output-format csv -d $out -o ic$i -- $BENCH > $out/ic$i.log 2>&1
  grep '"metric"' $out/ic$i.log | tail -1 > $out/ic$i.json
done
python - $out "${KERNEL:-run_kernel}" <<'PY'
import csv, glob, json, sys
out, kern = sys.argv[1], sys.argv[2]
tot = {}
for f in sorted(glob.glob(out + "/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
leap = None
for f in sorted(glob.glob(out + "/ic*.json")):
    try:
        leap = json.loads(open(f).read())["leapfrogs"]
    except Exception:
        pass
for k in sorted(tot):
    print("%-28s %.4e%s" % (k, tot[k], ("   per leapfrog %.3f" % (tot[k] / leap)) if leap else ""))
json.dump({"kernel": kern, "leapfrogs": leap, "counters": tot}, open(out + "/icache_totals.json", "w"), indent=1)
PY
rm -f $out/*.log; find $out -name "*.csv" -size +1M -delete
