#!/bin/bash
# GPU box: the Welford-row cache prefetch (LMC_WELFORD_PREFETCH) on / off, alternating runs on one box, after a parity check.
L="build_variants/liblmc_wpf0.so build_variants/liblmc_wpf1.so"
LMC_HIP_LIB=build_variants/liblmc_wpf1.so python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
for args in "--target std_normal" "--target std_normal --dim 64 --chains 4096" "--target diag --dim 1000 --chains 8192" "" "--target funnel --dim 256 --chains 16384 --max-treedepth 12"; do for i in 1 2; do for lib in $L; do
r=$(LMC_HIP_LIB=$lib timeout 600 python bench.py $args --no-cpu-baseline --no-ess --no-secondary --no-rccl-check --no-tail 2>/dev/null | grep metric | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e kernel_ms %.2f' % (d['value'], d['roofline']['kernel_ms_avg']))")
echo "$(basename $lib) [$args]: $r"; done; done; done
