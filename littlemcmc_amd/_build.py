"""Build liblmc_hip.so (hipcc, gfx950) in-tree. Used by __graft_entry__.build() and by
littlemcmc_amd.targets.UserTarget (which rebuilds the library around a user-supplied device
log-density header)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_NAME = "liblmc_hip.so"

# -ffp-contract=off: elementwise arithmetic must be the same rounded IEEE operations as numpy's
#   (parity with the reference); explicit __builtin_fma is used only inside reductions.
# -disable-machine-licm: the inlined f64 log/exp/log1p/expm1/pow bodies carry ~70 polynomial
#   constants; hoisting them out of the sampling loops pins ~80 VGPRs for the whole kernel.
HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
    "-mllvm", "-disable-machine-licm",
]


def sources():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".hpp"))]


def lib_path():
    return os.path.join(HERE, LIB_NAME)


def needs_build(out=None):
    out = out or lib_path()
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    deps = sources() + [os.path.join(os.path.dirname(HERE), "include", "lmc_hip.h")]
    return any(os.path.getmtime(s) > t for s in deps)


def build(out=None, extra_flags=(), force=False, verbose=False):
    """Compile csrc/lmc_engine.hip -> liblmc_hip.so. Raises on failure (no fallback)."""
    out = out or lib_path()
    if not force and not needs_build(out):
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    cmd = [hipcc] + HIPCC_FLAGS + ["-I", CSRC] + list(extra_flags) + ["-o", out, os.path.join(CSRC, "lmc_engine.hip")]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed (%d):\n%s\n%s" % (res.returncode, res.stdout, res.stderr))
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
