#!/bin/bash
# GPU box: the per-GPU halves of C3's strong-scaling split on ONE GPU -- 65 536 / 32 768 / 16 384 / 8 192 chains are what each of
# 1 / 2 / 4 / 8 GPUs runs -- so that a measured SCALE curve can be cross-checked against single-GPU rates of the same build.
#   tools/per_gpu_sizes.sh <out.json> [bench args, e.g. --target std_normal for the north_star shape]
out=$1; shift
echo "[" > $out
first=1
for c in 65536 32768 16384 8192; do
  line=$(timeout 600 python bench.py --chains $c --no-cpu-baseline --no-secondary --no-rccl-check "$@" 2>/dev/null | grep '"metric"')
  [ $first = 1 ] || echo "," >> $out
  first=0
  echo "$line" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'chains_on_this_gpu': d['config']['chains_total'], 'n_gpus_of_the_65536_chain_job': 65536 // d['config']['chains_total'],
  'leapfrog_steps_per_s': d['value'], 'wall_s': d['wall_s'], 'kernel_ms_per_step': d['roofline']['kernel_ms_avg'],
  'mean_wave_slot_occupancy': d['tail']['mean_wave_slot_occupancy'], 'roofline_frac': d['roofline']['frac'],
  'ess_per_sec_min': d['ess_per_sec']['min'] if d['ess_per_sec'] else None, 'source_hash': d['source_hash'], 'workload': d['config']['workload']}))" >> $out
done
echo "]" >> $out
cat $out
