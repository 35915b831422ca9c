#!/bin/bash
# GPU box, ONE call: every profiles/rNN_* artifact of the current binary.
#   tools/reprofile.sh <outdir-under-gpurun_out> <tag>      then, here: tools/reprofile_collect.sh <outdir> <tag>
# Order matters: the PMC passes first, summarised ON the box (profiles/pmc_counters.json then carries this binary's hash),
# so that the bench lines taken afterwards quote roofline.traffic from counters of the binary they ran on.
set -u
out=$1; tag=$2
bash tools/profile_all.sh $out > gpurun_out/$out.summary.txt 2>&1
bash tools/summarize_all.sh $out $tag >> gpurun_out/$out.summary.txt 2>&1
L=gpurun_out/$out/lines
mkdir -p $L
line() { grep '"metric"' | head -1; }
timeout 900 python bench.py 2>/dev/null | line > $L/bench_default_line.json
timeout 900 python bench.py --gpus 2 --launcher inproc --no-cpu-baseline 2>/dev/null | line > $L/bench_inproc_2_engines_one_gpu.json
B="--no-cpu-baseline --no-secondary --no-rccl-check"
timeout 600 python bench.py $B --target diag --dim 1000 --chains 8192 2>/dev/null | line > $L/c4_bench.json
timeout 600 python bench.py $B --target diag --dim 1000 --chains 8192 --rng philox 2>/dev/null | line > $L/c4_philox_bench.json
timeout 600 python bench.py $B --target funnel --dim 256 --chains 16384 --max-treedepth 12 --steps 1 --iters-per-step 2000 --warmup 0 2>/dev/null | line > $L/c5_one_launch_bench.json
timeout 900 python bench.py $B --mass full 2>/dev/null | line > $L/dense_full_bench.json
timeout 900 python bench.py $B --mass full_adapt 2>/dev/null | line > $L/dense_full_adapt_bench.json
bash tools/per_gpu_sizes.sh $L/per_gpu_sizes.json > /dev/null 2>&1
cat gpurun_out/$out.summary.txt
for f in $L/*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    if isinstance(d, dict):
        print("%-44s %.4e  traffic %s  hash %s" % (sys.argv[1].split("/")[-1], d["value"], d["roofline"].get("traffic"), d.get("source_hash")))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
