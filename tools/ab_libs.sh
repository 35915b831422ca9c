#!/bin/bash
# GPU box: A/B of two builds of the library with the same ABI (LMC_HIP_LIB), alternating runs on one box.
#   tools/ab_libs.sh <lib A> <lib B> <rounds> <bench args...>   e.g. tools/ab_libs.sh build_variants/liblmc_late_stop.so littlemcmc_amd/liblmc_hip.so 3 --target std_normal
a=$1; b=$2; n=$3; shift 3
for i in $(seq 1 $n); do
  for lib in $a $b; do
    r=$(LMC_HIP_LIB=$lib timeout 600 python bench.py --no-cpu-baseline --no-ess --no-secondary --no-rccl-check --no-tail "$@" 2>/dev/null | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e  kernel_ms %.3f depth %.2f' % (d['value'], d['roofline']['kernel_ms_avg'], d['mean_depth_draws']))")
    echo "$(basename $lib) [$*] $r"
  done
done
