#!/bin/bash
# here: gpurun_out/<outdir>/{c3,std128,...} -> profiles/<tag>_<workload>_* + profiles/pmc_counters.json
set -e
out=$1; tag=$2
python tools/summarize_profile.py gpurun_out/$out/c3 ${tag}_c3 ar1:128 AR1Target > /dev/null
python tools/summarize_profile.py gpurun_out/$out/std128 ${tag}_std128 std_normal:128 StdNormalTarget > /dev/null
python tools/summarize_profile.py gpurun_out/$out/std128_philox ${tag}_std128_philox std_normal:128:philox StdNormalTarget > /dev/null
python tools/summarize_profile.py gpurun_out/$out/c2 ${tag}_c2 std_normal:64 StdNormalTarget > /dev/null
python tools/summarize_profile.py gpurun_out/$out/c4 ${tag}_c4 diag:1000 DiagGaussianTarget > /dev/null
python tools/summarize_profile.py gpurun_out/$out/c5 ${tag}_c5 funnel:256 FunnelTarget > /dev/null
python tools/summarize_profile.py gpurun_out/$out/dense_full ${tag}_dense_full ar1:128:full run_dense_coop_kernel > /dev/null
python tools/summarize_profile.py gpurun_out/$out/dense_full_adapt ${tag}_dense_full_adapt ar1:128:full_adapt run_dense_kernel > /dev/null
for w in c3 std128 std128_philox c2 c4 c5 dense_full dense_full_adapt; do cp gpurun_out/$out/$w/bench_stats.json profiles/${tag}_${w}_bench_under_profiler.json; done
python - <<'PY'
import json
d = json.load(open("profiles/pmc_counters.json"))
for k, v in d.items():
    print("%-26s VALU/leapfrog %7.1f  SALU %6.1f  HBM B/leapfrog %9.1f  simd_valu_busy %.2f  hash %s" % (
        k, v.get("valu_inst_per_leapfrog", 0), v.get("salu_inst_per_leapfrog", 0), v["hbm_bytes_per_leapfrog"], v.get("simd_valu_busy", 0), v.get("source_hash")))
PY
