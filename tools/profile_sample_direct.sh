#!/bin/bash
# GPU box: what the device does during ONE streamed lmc.sample() call (C3's chains, tune 300 + draws 200, 13.4 GiB returned):
# rocprofv3 kernel trace + memory-copy trace. Expected: the sampling kernel, a handful of window_gather_kernel dispatches (the
# statistics), set-up kernels -- and NO device-to-host copy of the draws (the sampling kernel wrote them into the returned array).
#   tools/profile_sample_direct.sh <outdir-under-gpurun_out>
set -u
out=gpurun_out/$1
mkdir -p $out
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - > /dev/null
cat > $out/job.py <<'PY'
import sys, time
sys.path.insert(0, ".")
import littlemcmc_amd as lmc
t0 = time.perf_counter()
tr, st = lmc.sample(lmc.targets.AR1(128, 0.9), 128, draws=200, tune=300, chains=65536, random_seed=20260928, progressbar=False)
print("sample(): %.3f s, trace %s %.2f GiB, tree_size sum %.0f" % (time.perf_counter() - t0, tr.shape, tr.nbytes / 2.0 ** 30, st["tree_size"].sum()))
PY
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $out/prof -- python $out/job.py > $out/job.log 2>&1
echo "exit $?" >> $out/job.log
tail -3 $out/job.log
for f in $(find $out/prof -name "*kernel_stats.csv" -o -name "*memory_copy_stats.csv" | sort); do echo "== $f"; head -12 $f | cut -c1-200; done
