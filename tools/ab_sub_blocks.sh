#!/bin/bash
# GPU box: number of sub-block streams of lmc_engine_run (LMC_SUB_BLOCKS = 2 / 4 / 8), alternating runs on one box.
if [ $# -eq 0 ]; then set -- "" "--target std_normal" "--target std_normal --dim 64 --chains 4096" "--target diag --dim 1000 --chains 8192" "--target funnel --dim 256 --chains 16384 --max-treedepth 12"; fi
for args in "$@"; do for i in 1 2; do for sb in 2 4 8; do
r=$(LMC_SUB_BLOCKS=$sb timeout 600 python bench.py $args --no-cpu-baseline --no-ess --no-secondary --no-rccl-check --no-tail 2>/dev/null | grep metric | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e kernel_ms %.2f' % (d['value'], d['roofline']['kernel_ms_avg']))")
echo "LMC_SUB_BLOCKS=$sb [$args]: $r"; done; done; done
for sb in 2 4 8; do echo "LMC_SUB_BLOCKS=$sb sample() loop: $(LMC_SUB_BLOCKS=$sb python tools/sample_path_rate.py funnel s4x100 2>&1 | grep leapfrog)"; done
