#!/bin/bash
# GPU box: A/B of LDS plans of the one-wave sampling kernels (tools/variant_build.py variants), alternating runs on one box.
#   base   : MT19937 state + three cold slots in LDS, stack level 2 in the scratch row (ships as plan 0)
#   l2lds  : -DLMC_MT_IN_LDS_W1=0 -DLMC_PAIR_COLD_LDS=1 -> MT19937 in place (L2), one cold slot, stack level 2 in LDS
#   l2lds2 : -DLMC_MT_IN_LDS_W1=0 -DLMC_PAIR_COLD_LDS=0 -> no cold slot in LDS
# tools/ab_level2_lds.sh ["bench args" ...]   (default: C3 and the north_star shape)
L=${LIBS:-"build_variants/liblmc_base.so build_variants/liblmc_l2lds.so build_variants/liblmc_l2lds2.so"}
if [ $# -eq 0 ]; then set -- "" "--target std_normal"; fi
for args in "$@"; do for i in 1 2; do for lib in $L; do
r=$(LMC_HIP_LIB=$lib timeout 600 python bench.py $args --no-cpu-baseline --no-ess --no-secondary --no-rccl-check 2>/dev/null | grep metric | python -c "import json,sys; d=json.loads(sys.stdin.read()); t=d['tail']; print('%.4e kernel_ms %.2f lds %d resident %d lone %.3f us' % (d['value'], d['roofline']['kernel_ms_avg'], t['lds_bytes_per_workgroup'], t['resident_chains'], t['lone_wave_us_per_leapfrog']))")
echo "$(basename $lib) [$args]: $r"; done; done; done
