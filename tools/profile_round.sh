#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + PMC passes of the default bench command.
# Usage: tools/profile_round.sh <outdir-under-gpurun_out> [bench args...]
set -u
out=gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BENCH="python bench.py --no-cpu-baseline --no-ess --no-secondary --no-rccl-check --no-tail $*"
echo "== stats"; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o stats -- $BENCH > $out/bench_stats.log 2>&1
grep '"metric"' $out/bench_stats.log > $out/bench_stats.json
echo "== pmc sq"; timeout 900 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $out -o pmc_sq -- $BENCH > $out/bench_pmc_sq.log 2>&1
grep '"metric"' $out/bench_pmc_sq.log > $out/bench_pmc_sq.json
echo "== pmc fetch"; timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out -o pmc_fetch -- $BENCH > $out/bench_pmc_fetch.log 2>&1
grep '"metric"' $out/bench_pmc_fetch.log > $out/bench_pmc_fetch.json
echo "== pmc write"; timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out -o pmc_write -- $BENCH > $out/bench_pmc_write.log 2>&1
grep '"metric"' $out/bench_pmc_write.log > $out/bench_pmc_write.json
echo "== pmc lds/vmem"; timeout 900 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM --output-format csv -d $out -o pmc_mem -- $BENCH > $out/bench_pmc_mem.log 2>&1
echo "== pmc waits"; timeout 900 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out -o pmc_wait -- $BENCH > $out/bench_pmc_wait.log 2>&1
ls -la $out | head -40
rm -f $out/*.log
find $out -name "*.csv" -size +2M -delete
