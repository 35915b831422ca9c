#!/bin/bash
# GPU box: one line per BASELINE configuration (+ the north_star shape) with the tail fields. tools/bench_all.sh [extra bench args]
for t in ar1:128:65536:10 std_normal:128:65536:10 std_normal:64:4096:10 funnel:256:16384:12 diag:1000:8192:10; do
  IFS=: read tg dm ch md <<< "$t"
  python bench.py --target $tg --dim $dm --chains $ch --max-treedepth $md --no-cpu-baseline --no-secondary --no-ess --no-trace --no-rccl-check "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().splitlines()[0]); t=d['tail']
print('%-34s %.4e  depth %.2f  lone %.3f us/leapfrog  occupancy %.3f  tail bound %.3f s of %.3f s' % (d['config']['workload'][:34], d['value'], d['mean_depth_draws'], t['lone_wave_us_per_leapfrog'], t['mean_wave_slot_occupancy'], t['implied_wall_lower_bound_s'], d['wall_s']))"
done
