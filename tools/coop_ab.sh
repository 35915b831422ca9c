#!/bin/bash
# GPU box: the staggered (two groups of four chains, half a tick apart) against the lock-step form of the shared-matrix
# dense kernel (csrc/lmc_dense.hpp: coop_product). Needs
#   build_variants/liblmc_lockstep.so        _build.build(out=..., extra_flags=['-DLMC_COOP_STAGGER=0'])
#   build_variants/liblmc_stagger_timing.so  _build.build(out=..., extra_flags=['-DLMC_COOP_TIMING'])
# tools/coop_ab.sh [rounds]
set -u
rounds=${1:-2}
B="--no-cpu-baseline --no-secondary --no-rccl-check --no-ess --mass full"
one() {
  LMC_HIP_LIB=$1 timeout 300 python bench.py $B $2 2>&1 | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e leap/s  kernel_ms %.2f depth %.2f' % (d['value'], d['roofline']['kernel_ms_avg'], d['mean_depth_draws']))"
}
for r in $(seq $rounds); do
  for lib in littlemcmc_amd/liblmc_hip.so build_variants/liblmc_lockstep.so; do
    echo "d=128 65536 chains  $(basename $lib)  $(one $lib '')"
  done
done
for lib in littlemcmc_amd/liblmc_hip.so build_variants/liblmc_lockstep.so; do
  echo "d=64 65536 chains  $(basename $lib)  $(one $lib '--dim 64')"
  echo "d=32 65536 chains  $(basename $lib)  $(one $lib '--dim 32')"
done
if [ -f build_variants/liblmc_stagger_timing.so ]; then
  LMC_HIP_LIB=build_variants/liblmc_stagger_timing.so PYTHONPATH=. timeout 300 python tools/coop_timing.py
fi
