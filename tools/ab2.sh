#!/bin/bash
# tools/ab2.sh "<bench args>" -> one summary line
r=$(timeout 900 python bench.py --no-cpu-baseline --no-ess --no-secondary $1 2>&1 | grep metric | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e leap/s  kernel_ms %.2f depth %.2f div %d  frac60 %.3f' % (d['value'], d['roofline']['kernel_ms_avg'], d['mean_depth_draws'], d['divergences_after_tune'], d['roofline']['frac']))")
echo "$1 => $r"
