#!/bin/bash
# GPU box: the per-chain LDS plan choice of the one-wave sampling kernels (lmc_sampler.hpp: kDynPlan) against both pinned plans
# and, if present, the previous build (build_variants/liblmc_base.so), alternating runs on one box.
#   tools/ab_lds_plan.sh ["bench args" ...]     (default: C3, north_star shape, C2, C5)
if [ $# -eq 0 ]; then set -- "" "--target std_normal" "--target std_normal --dim 64 --chains 4096" "--target funnel --dim 256 --chains 16384 --max-treedepth 12"; fi
run() { timeout 600 python bench.py $1 --no-cpu-baseline --no-ess --no-secondary --no-rccl-check 2>/dev/null | grep metric | python -c "import json,sys; d=json.loads(sys.stdin.read()); t=d['tail']; print('%.4e kernel_ms %.2f lds %d lone %.3f us' % (d['value'], d['roofline']['kernel_ms_avg'], t['lds_bytes_per_workgroup'], t['lone_wave_us_per_leapfrog']))"; }
for args in "$@"; do for i in 1 2; do
  [ -f build_variants/liblmc_base.so ] && echo "previous build   [$args]: $(LMC_HIP_LIB=build_variants/liblmc_base.so run "$args")"
  for plan in 0 1 auto; do echo "LMC_LDS_PLAN=$plan [$args]: $(LMC_LDS_PLAN=$plan run "$args")"; done
done; done
