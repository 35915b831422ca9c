"""Build liblmc_hip.so (hipcc, gfx950) in-tree. Used by __graft_entry__.build() and by
littlemcmc_amd.targets.UserTarget (which rebuilds the library around a user-supplied device
log-density header)."""
import os
import shutil
import subprocess
import tempfile

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


def source_hash(extra_flags=()):
    """Identity of a build: sha256 over the kernel sources, the C header and the compiler flags (16 hex digits).
    It is compiled INTO the library (-DLMC_SOURCE_HASH -> lmc_build_hash()), so what a process reports is the hash of
    the binary it loaded, not of the tree it happens to sit in. Profiles under profiles/ record it, and a measurement is
    only quoted for the build it was taken on."""
    import hashlib

    h = hashlib.sha256()
    for path in sources() + [os.path.join(os.path.dirname(HERE), "include", "lmc_hip.h")]:
        with open(path, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(list(HIPCC_FLAGS) + list(extra_flags)).encode())
    return h.hexdigest()[:16]


def lib_path():
    return os.path.join(HERE, LIB_NAME)


_STAMP = b"LMC_BUILD_HASH="


def binary_hash(path=None):
    """The source hash a built library carries (csrc/lmc_engine.hip: kBuildStamp = -DLMC_SOURCE_HASH at compile time),
    read from the file without loading it; None if the file is missing or carries no stamp. The loaded library reports
    the same string through lmc_build_hash()."""
    path = path or lib_path()
    try:
        with open(path, "rb") as fh:
            blob = fh.read()
    except OSError:
        return None
    i = blob.find(_STAMP)
    if i < 0:
        return None
    j = blob.find(b"\0", i)
    return blob[i + len(_STAMP):j].decode("ascii", "replace")


def needs_build(out=None, extra_flags=()):
    """A library is current when the hash stamped into the BINARY equals the hash of the sources as they are now --
    not when its mtime is newer (a stale .so restored by a checkout or a copy would otherwise pass for the tree's)."""
    return binary_hash(out or lib_path()) != source_hash(extra_flags)


def build(out=None, extra_flags=(), force=False, verbose=False):
    """Compile csrc/*.hip -> liblmc_hip.so (one hipcc process per translation unit, then a link). Raises on
    failure (no fallback)."""
    out = out or lib_path()
    if not force and not needs_build(out, extra_flags):
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    compile_flags = [f for f in HIPCC_FLAGS if f != "-shared"] + ['-DLMC_SOURCE_HASH="%s"' % source_hash(extra_flags)]
    units = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
    objdir = tempfile.mkdtemp(prefix="lmc_build_")
    try:
        procs = []
        for src in units:
            obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
            cmd = [hipcc] + compile_flags + ["-I", CSRC] + list(extra_flags) + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((cmd, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
        objs = []
        for cmd, obj, proc in procs:
            so, se = proc.communicate()
            if proc.returncode != 0:
                raise RuntimeError("hipcc failed (%d): %s\n%s\n%s" % (proc.returncode, " ".join(cmd), so, se))
            objs.append(obj)
        link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
        if verbose:
            print(" ".join(link))
        res = subprocess.run(link, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("link failed (%d):\n%s\n%s" % (res.returncode, res.stdout, res.stderr))
    finally:
        shutil.rmtree(objdir, ignore_errors=True)
    return out


if __name__ == "__main__":
    import sys

    print(build(force="--force" in sys.argv or True, verbose=True))
