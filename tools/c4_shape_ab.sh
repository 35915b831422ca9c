#!/bin/bash
# GPU box: C4 (8 192 x d = 1000) through run_kernel<4,4> (default) and run_kernel<2,8> (LMC_RUN_SHAPE=2,8), alternating runs on
# one box, both RNG modes; first the d > 512 replay tests through the <2,8> instantiation.   tools/c4_shape_ab.sh [rounds]
n=${1:-3}
LMC_RUN_SHAPE=2,8 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "wide_and_multi_wave and (1000 or 600)" 2>&1 | tail -3
for i in $(seq 1 $n); do
  for shape in 4,4 2,8; do
    for rng in numpy philox; do
      r=$(LMC_RUN_SHAPE=$shape timeout 600 python bench.py --target diag --dim 1000 --chains 8192 --rng $rng --no-cpu-baseline --no-ess --no-secondary --no-rccl-check 2>/dev/null | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); t=d['tail']; print('%.4e  kernel_ms %.3f depth %.2f lone %.3f us occ %.3f resident %d' % (d['value'], d['roofline']['kernel_ms_avg'], d['mean_depth_draws'], t['lone_wave_us_per_leapfrog'], t['mean_wave_slot_occupancy'], t['resident_chains']))")
      echo "shape $shape rng $rng: $r"
    done
  done
done
