#!/bin/bash
# GPU box: like ab_libs.sh for any number of builds: tools/abc_libs.sh "<lib> <lib> ..." <rounds> <bench args...>
libs=$1; n=$2; shift 2
for i in $(seq 1 $n); do
  for lib in $libs; do
    r=$(LMC_HIP_LIB=$lib timeout 600 python bench.py --no-cpu-baseline --no-ess --no-secondary --no-rccl-check --no-tail "$@" 2>/dev/null | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e  kernel_ms %.3f depth %.2f' % (d['value'], d['roofline']['kernel_ms_avg'], d['mean_depth_draws']))")
    echo "$(basename $lib) [$*] $r"
  done
done
