#!/bin/bash
# here, after tools/final_round_check.sh rNN on a GPU box: gpurun_out/rNN_final + rNNprof -> profiles/ for the current source hash
#   tools/collect_final.sh rNN <previous hash quoted in the docs>
set -e
tag=$1; old=${2:-}
H=$(python -c 'from littlemcmc_amd import _build;print(_build.source_hash())')
bash tools/reprofile_collect.sh ${tag}prof $tag > /dev/null 2>&1
cp gpurun_out/${tag}_final/bench_driver_form.json profiles/${tag}_bench_driver_form.json
cp gpurun_out/${tag}_final/bench_detail_driver_form.json profiles/${tag}_bench_detail_driver_form.json
(echo "# on the FINAL build (hash $H), seeds the test suite does not use"; grep -v "Only\|amdgpu" gpurun_out/${tag}_final/fuzz.txt) > profiles/${tag}_fuzz_final_build.txt
python - "$tag" "$H" <<'PY'
import json, sys
tag, H = sys.argv[1], sys.argv[2]
p = "profiles/%s_sample_e2e.txt" % tag
s = open(p).read()
i = s.index("# the same table on the FINAL build")
keep = lambda t: "\n".join(l for l in t.split("\n") if l.startswith(("c3:", "c2:", "run ", "best")))
s = s[:i] + "# the same table on the FINAL build (%s), another box\n" % H + keep(open("gpurun_out/%s_final/e2e_c3.txt" % tag).read()) + "\n" + keep(open("gpurun_out/%s_final/e2e_c2.txt" % tag).read()) + "\n"
open(p, "w").write(s)
d = json.load(open("profiles/%s_bench_driver_form.json" % tag))
print(len(json.dumps(d)), d["value"], d["roofline"]["frac"], d["roofline"]["traffic"], d["source_hash"], d["sample_e2e"]["wall_s"], d["sample_e2e"]["kernel_only_s"], d["cpu_baseline"]["value"])
assert d["source_hash"] == H and d["roofline"]["traffic"] is not None
PY
if [ -n "$old" ]; then sed -i "s/$old/$H/g" profiles/README.md DESIGN.md; fi
grep -c "$H" profiles/README.md profiles/pmc_counters.json
