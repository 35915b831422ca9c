#!/bin/bash
# GPU box: what FMA-free arithmetic costs (a labelled NON-PARITY experiment, never a headline: numpy rounds every product, so
# the shipped library is built -ffp-contract=off and only the explicit __builtin_fma of the reductions contract).
#   python tools/variant_build.py fast="-ffp-contract=fast"     (here)     then on the box: tools/fp_contract_experiment.sh [rounds]
# Alternating runs of the shipped library and the contracted build on one box: the counter-based momentum stream (the mode
# that does not claim the reference's draws anyway) and the parity stream for comparison, on the north_star shape, C3, and
# C5 as one launch (its wall time is one lone wavefront's dependent instruction chain).
n=${1:-2}
B="--no-cpu-baseline --no-secondary --no-rccl-check --no-ess --no-tail"
run() { LMC_HIP_LIB=$1 timeout 900 python bench.py $B $2 2>/dev/null | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e leapfrog-steps/s  kernel_ms %.3f  depth %.2f' % (d['value'], d['roofline']['kernel_ms_avg'], d['mean_depth_draws']))"; }
for i in $(seq 1 $n); do
  for lib in littlemcmc_amd/liblmc_hip.so build_variants/liblmc_fast.so; do
    for args in "--target std_normal --rng philox" "--target std_normal" "--rng philox" "" \
                "--target funnel --dim 256 --chains 16384 --max-treedepth 12 --steps 1 --iters-per-step 2000 --warmup 0 --rng philox"; do
      echo "$(basename $lib) [${args:-C3 parity stream}] $(run $lib "$args")"
    done
  done
done
