#!/bin/bash
# GPU box: kernel durations of tools/lone_leapfrog_probe.py's trajectory launches (host timing there is dominated by copying the
# stored states out): us per leapfrog = (t[8000 steps] - t[2000 steps]) / 6000 per target.
out=gpurun_out/r05_lone_probe; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out -o trace -- python tools/lone_leapfrog_probe.py > $out/log.txt 2>&1
python - $out <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/trace_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "trajectory_kernel" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), r["Kernel_Name"].split("(")[0], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
rows.sort()
# five launches per target: 500, 2000, 8000, 2000, 8000 steps
for i in range(0, len(rows), 5):
    grp = rows[i:i + 5]
    if len(grp) < 5:
        break
    us = (grp[4][2] - grp[3][2]) / 6000.0
    print("%-60s %.3f us per leapfrog (kernel: %.0f us for 8000 steps, %.0f us for 2000)" % (grp[0][1][-60:], us, grp[4][2], grp[3][2]))
PY
