#!/bin/bash
# A/B bench of library variants: tools/ab.sh target chains steps ips lib1 lib2 ...
t=$1; c=$2; k=$3; ips=$4; shift 4
for lib in "$@"; do
  r=$(LMC_HIP_LIB=$lib timeout 600 python bench.py --chains $c --steps $k --warmup 1 --iters-per-step $ips --target $t --no-cpu-baseline --no-secondary $LMC_BENCH_EXTRA 2>&1 | grep metric | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e  kernel_ms %.2f depth %.2f' % (d['value'], d['roofline']['kernel_ms_avg'], d['mean_depth_draws']))")
  echo "$t $(basename $lib) $r"
done
