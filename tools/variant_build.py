#!/usr/bin/env python3
"""Build variants of liblmc_hip.so that differ in the flags of ONE translation unit (default lmc_engine.hip: the sampling
kernels): the other units are compiled once into build_variants/objs/ and re-linked, so a variant costs one hipcc run
(~2 min) instead of a full build. For A/B runs through LMC_HIP_LIB (tools/ab_libs.sh / abc_libs.sh).

    python tools/variant_build.py name1="-mllvm -amdgpu-use-amdgpu-trackers" name2="-DLMC_WAVES_NS2=4" ...
    -> build_variants/liblmc_<name>.so   (name "base" with no flags = the shipped flags)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from littlemcmc_amd import _build  # noqa: E402

UNIT = os.environ.get("LMC_VARIANT_UNIT", "lmc_engine")
OUT = os.path.join(ROOT, "build_variants")
OBJS = os.path.join(OUT, "objs")
os.makedirs(OBJS, exist_ok=True)
hipcc = "/opt/rocm/bin/hipcc"
base_flags = [f for f in _build.HIPCC_FLAGS if f != "-shared"] + ['-DLMC_SOURCE_HASH="%s"' % _build.source_hash(), "-I", _build.CSRC]
units = [f[:-4] for f in sorted(os.listdir(_build.CSRC)) if f.endswith(".hip")]
stamp = os.path.join(OBJS, "hash.txt")
if not os.path.exists(stamp) or open(stamp).read() != _build.source_hash():
    procs = [subprocess.Popen([hipcc] + base_flags + ["-c", os.path.join(_build.CSRC, u + ".hip"), "-o", os.path.join(OBJS, u + ".o")])
             for u in units if u != UNIT]
    assert all(p.wait() == 0 for p in procs)
    open(stamp, "w").write(_build.source_hash())
variants = [a.split("=", 1) for a in sys.argv[1:]]
procs = []
for name, flags in variants:
    obj = os.path.join(OBJS, "%s_%s.o" % (UNIT, name))
    procs.append((name, obj, subprocess.Popen([hipcc] + base_flags + flags.split() + ["-c", os.path.join(_build.CSRC, UNIT + ".hip"), "-o", obj])))
for name, obj, p in procs:
    if p.wait() != 0:
        print("variant %s FAILED to compile" % name)
        continue
    lib = os.path.join(OUT, "liblmc_%s.so" % name)
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj] + [os.path.join(OBJS, u + ".o") for u in units if u != UNIT]
    subprocess.check_call(link)
    print("built", lib)
