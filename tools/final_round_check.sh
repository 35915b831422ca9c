#!/bin/bash
# GPU box, ONE call at the end of a round: the whole -m gpu suite, smoke(), every profiles/rNN_* artifact of the final binary
# (tools/reprofile.sh), the fuzzers on seeds the suite does not use, the end-to-end table.   tools/final_round_check.sh rNN
set -u
tag=$1; out=gpurun_out/${tag}_final
mkdir -p $out
PYTHONFAULTHANDLER=1 python -m pytest tests -q -m gpu > $out/gpu_suite_full.log 2>&1; grep -v "Extension modules" $out/gpu_suite_full.log | tail -40 > $out/gpu_suite.txt
python __graft_entry__.py smoke 2>&1 | tail -2 > $out/smoke.txt
bash tools/reprofile.sh ${tag}prof $tag > $out/reprofile.log 2>&1
( echo "== tools/fuzz_rare.py 2000 777"; python tools/fuzz_rare.py 2000 777 | tail -2
  echo "== tools/fuzz_rare.py 600 778 deep"; python tools/fuzz_rare.py 600 778 deep | tail -2
  echo "== LMC_FORCE_WIDE=1 tools/fuzz_rare.py 400 779   (the general kernels: lmc_tree_leaf.hpp's transition)"; LMC_FORCE_WIDE=1 python tools/fuzz_rare.py 400 779 | tail -2
  echo "== tools/fuzz_paths.py 300 777"; python tools/fuzz_paths.py 300 777 | tail -3 ) > $out/fuzz.txt 2>&1
python tools/sample_e2e.py c3 65536 1000 1000 1 > $out/e2e_c3.txt 2>&1
python tools/sample_e2e.py c2 > $out/e2e_c2.txt 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_form.json 2> $out/bench_driver_form.err
cp bench_detail.json $out/bench_detail_driver_form.json
cat $out/gpu_suite.txt $out/smoke.txt $out/fuzz.txt; grep -h "best\|run " $out/e2e_c3.txt $out/e2e_c2.txt
