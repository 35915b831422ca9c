#!/bin/bash
# PMC passes for one bench command, summed for the kernel matching $KERNEL (default run_dense_kernel):
#   tools/pmc_quick.sh <outdir-under-gpurun_out> [bench args...]
set -u
out=gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BENCH="python bench.py --no-cpu-baseline --no-ess --no-secondary $*"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $set --output-format csv -d $out -o pmc$i -- $BENCH > $out/pmc$i.log 2>&1
  grep '"metric"' $out/pmc$i.log | tail -1 > $out/pmc$i.json
done
python - $out "${KERNEL:-run_dense_kernel}" <<'PY'
import csv, glob, json, sys
out, kern = sys.argv[1], sys.argv[2]
tot = {}
for f in sorted(glob.glob(out + "/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
leap = None
for f in sorted(glob.glob(out + "/pmc*.json")):
    try:
        leap = json.loads(open(f).read())["leapfrogs"]
    except Exception:
        pass
for k in sorted(tot):
    print("%-24s %.4e%s" % (k, tot[k], ("   per leapfrog %.2f" % (tot[k] / leap)) if leap else ""))
json.dump({"kernel": kern, "leapfrogs": leap, "counters": tot}, open(out + "/pmc_totals.json", "w"), indent=1)
PY
rm -f $out/*.log; find $out -name "*.csv" -size +1M -delete
