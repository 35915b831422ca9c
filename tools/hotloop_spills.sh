#!/bin/bash
# tools/hotloop_spills.sh [extra hipcc flags]: compile lmc_engine.hip to ISA and report, for run_kernel<2,1,AR1Target>,
# the scratch (spill) instructions inside the innermost (depth >= 3) loops, plus the static instruction census.
set -e
out=/tmp/isa/hl.s
mkdir -p /tmp/isa
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -mllvm -disable-machine-licm "$@" -I littlemcmc_amd/csrc -S --cuda-device-only -o $out littlemcmc_amd/csrc/lmc_engine.hip 2>/dev/null
K=${KERNEL:-_ZN3lmc10run_kernelILi2ELi1ENS_9AR1TargetE}
awk -v k="$K" '$0 ~ "Begin function " k {f=1} f{print} /\.Lfunc_end/{if(f)exit}' $out > /tmp/isa/hlk.s
python3 - <<'PY'
import re
lines = open('/tmp/isa/hlk.s').read().split('\n')
depth = 0; cur = 0
hot = []
for i, l in enumerate(lines):
    m = re.search(r'Depth=(\d+)', l)
    if re.match(r'^\.LBB|^; %bb', l.strip()) :
        cur = 0
    if m: cur = int(m.group(1))
    if 'scratch_' in l and cur >= 3:
        hot.append((i + 1, l.strip()))
print("scratch ops inside depth>=3 loops: %d" % len(hot))
for h in hot: print("  %d: %s" % h)
tot = sum(1 for l in lines if 'scratch_' in l)
print("scratch ops total: %d" % tot)
for l in lines:
    if re.search(r'\.(vgpr_count|sgpr_count|private_segment_fixed_size|vgpr_spill_count):', l): print(l.strip())
PY
