#!/bin/bash
# GPU box: rocprofv3 kernel-trace stats + PMC passes for the headline workload, the secondary configurations, the
# counter-based-RNG line and the two dense-mass kernels.
#   tools/profile_all.sh <outdir-under-gpurun_out>      then, here: tools/summarize_all.sh <outdir> rNN
set -u
out=$1
bash tools/profile_round.sh $out/c3 > /dev/null 2>&1
bash tools/profile_round.sh $out/std128 --target std_normal > /dev/null 2>&1
bash tools/profile_round.sh $out/std128_philox --target std_normal --rng philox > /dev/null 2>&1
bash tools/profile_round.sh $out/c2 --target std_normal --dim 64 --chains 4096 > /dev/null 2>&1
bash tools/profile_round.sh $out/c4 --target diag --dim 1000 --chains 8192 > /dev/null 2>&1
bash tools/profile_round.sh $out/c5 --target funnel --dim 256 --chains 16384 --max-treedepth 12 > /dev/null 2>&1
bash tools/profile_round.sh $out/dense_full --mass full --warmup 0 > /dev/null 2>&1
bash tools/profile_round.sh $out/dense_full_adapt --mass full_adapt --chains 16384 --steps 4 --iters-per-step 50 --warmup 0 > /dev/null 2>&1
for w in c3 std128 std128_philox c2 c4 c5 dense_full dense_full_adapt; do echo "== $w"; cat gpurun_out/$out/$w/bench_stats.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.4e leapfrog-steps/s  kernel %.1f ms  depth %.2f  frac %.4f' % (d['value'], d['roofline']['kernel_ms_avg'], d['mean_depth_draws'], d['roofline']['frac']))"; done
