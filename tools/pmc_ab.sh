#!/bin/bash
# SQ + LDS counter passes of run_kernel for several library variants (GPU box):
#   tools/pmc_ab.sh <outdir-under-gpurun_out> "<bench args>" lib1 lib2 ...
set -u
out=gpurun_out/$1; args=$2; shift 2
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  tag=$(basename $lib .so)
  i=0
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
             "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE"; do
    i=$((i+1))
    LMC_HIP_LIB=$lib timeout 900 rocprofv3 --pmc $set --output-format csv -d $out/$tag -o pmc$i -- python bench.py --no-cpu-baseline --no-ess --no-secondary $args > $out/$tag.pmc$i.log 2>&1
    grep '"metric"' $out/$tag.pmc$i.log | tail -1 > $out/$tag.pmc$i.json
  done
  python - $out $tag <<'PY'
import csv, glob, json, sys
out, tag = sys.argv[1], sys.argv[2]
tot = {}
for f in sorted(glob.glob(out + "/" + tag + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "run_kernel" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
leap = val = None
for f in sorted(glob.glob(out + "/" + tag + ".pmc*.json")):
    try:
        d = json.loads(open(f).read()); leap = d["leapfrogs"]; val = d["value"]
    except Exception:
        pass
print("== %s  leapfrogs %.4e  value(under profiler) %.4e" % (tag, leap or 0, val or 0))
for k in sorted(tot):
    print("  %-24s %.4e%s" % (k, tot[k], ("   per leapfrog %.2f" % (tot[k] / leap)) if leap else ""))
json.dump({"lib": tag, "leapfrogs": leap, "value_under_profiler": val, "counters": tot,
           "per_leapfrog": {k: v / leap for k, v in tot.items()} if leap else None}, open(out + "/" + tag + "_pmc.json", "w"), indent=1)
PY
  rm -rf $out/$tag $out/$tag.pmc*.log
done
