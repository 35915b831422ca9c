for args in "--target funnel --dim 256 --chains 16384 --max-treedepth 12" "--target std_normal --dim 64 --chains 4096" ""; do for i in 1 2; do for cfg in "4 4" "4 8" "8 8" "6 8"; do set -- $cfg
r=$(LMC_SUB_BLOCKS=$1 GPU_MAX_HW_QUEUES=$2 timeout 600 python bench.py $args --no-cpu-baseline --no-ess --no-secondary --no-rccl-check --no-tail 2>/dev/null | grep metric | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e kernel_ms %.2f' % (d['value'], d['roofline']['kernel_ms_avg']))")
echo "sub-blocks $1 GPU_MAX_HW_QUEUES=$2 [$args]: $r"; done; done; done
