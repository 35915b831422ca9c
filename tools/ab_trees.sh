#!/bin/bash
# GPU box: A/B of two source TREES (each with its own built library) on the same box, alternating runs.
#   tools/ab_trees.sh <tree A> <tree B> <rounds> <bench args...>     e.g. tools/ab_trees.sh build_variants/base_tree . 3 --target std_normal
a=$1; b=$2; n=$3; shift 3
root=$(pwd)
for i in $(seq 1 $n); do
  for t in $a $b; do
    r=$(cd $root/$t && timeout 600 python bench.py --no-cpu-baseline --no-ess --no-secondary --no-rccl-check --no-tail "$@" 2>/dev/null | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e  kernel_ms %.3f depth %.2f' % (d['value'], d['roofline']['kernel_ms_avg'], d['mean_depth_draws']))")
    echo "$t [$*] $r"
  done
done
