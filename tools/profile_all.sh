#!/bin/bash
# GPU box: rocprofv3 kernel-trace stats + PMC passes for the headline workload and the secondary configurations.
#   tools/profile_all.sh <outdir-under-gpurun_out>      then, here: tools/summarize_all.sh <outdir> rNN
set -u
out=$1
bash tools/profile_round.sh $out/c3 > /dev/null 2>&1
bash tools/profile_round.sh $out/std128 --target std_normal > /dev/null 2>&1
bash tools/profile_round.sh $out/c2 --target std_normal --dim 64 --chains 4096 > /dev/null 2>&1
bash tools/profile_round.sh $out/c4 --target diag --dim 1000 --chains 8192 --steps 10 > /dev/null 2>&1
bash tools/profile_round.sh $out/c5 --target funnel --dim 256 --chains 16384 --max-treedepth 12 > /dev/null 2>&1
for w in c3 std128 c2 c4 c5; do echo "== $w"; cat gpurun_out/$out/$w/bench_stats.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.4e leapfrog-steps/s  kernel %.1f ms  depth %.2f  frac %.4f' % (d['value'], d['roofline']['kernel_ms_avg'], d['mean_depth_draws'], d['roofline']['frac']))"; done
