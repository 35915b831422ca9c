#!/bin/bash
# here, after tools/reprofile.sh on a GPU box: gpurun_out/<outdir> -> profiles/<tag>_*
set -e
out=$1; tag=$2
bash tools/summarize_all.sh $out $tag
L=gpurun_out/$out/lines
for f in bench_default_line bench_inproc_2_engines_one_gpu c4_bench c4_philox_bench c5_one_launch_bench dense_full_bench dense_full_adapt_bench per_gpu_sizes; do
  [ -s $L/$f.json ] && cp $L/$f.json profiles/${tag}_$f.json
done
