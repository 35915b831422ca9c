#!/bin/bash
# GPU box: WHERE a wavefront of the sampling kernel spends its time, by source line -- program-counter sampling (rocprofv3,
# host-trap method, time based) of a library built with line tables (python tools/variant_build.py g="-gline-tables-only":
# same code, .loc information only), then tools/pc_sample_summary.py buckets the samples by purpose. A sample is a wavefront
# caught at a program counter: issue AND stall time, which is what a lone wavefront's leapfrog consists of.
#   tools/pc_sample.sh <outdir-under-gpurun_out> [lone|c3] [interval_us]
set -u
out=gpurun_out/$1; job=${2:-lone}; iv=${3:-50}
mkdir -p $out
export LMC_HIP_LIB=build_variants/liblmc_g.so
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - > /dev/null
timeout 600 rocprofv3 --pc-sampling-beta-enabled 1 --pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval $iv \
    --output-format csv -d $out/$job -- python tools/pc_sample_job.py $job > $out/$job.log 2>&1
echo "exit $?" >> $out/$job.log
find $out/$job -type f | head -20 >> $out/$job.log
for f in $(find $out/$job -name "*pc_sampling*csv" | head -3); do echo "== $f" >> $out/$job.log; head -5 $f >> $out/$job.log; wc -l $f >> $out/$job.log; done
tail -30 $out/$job.log
