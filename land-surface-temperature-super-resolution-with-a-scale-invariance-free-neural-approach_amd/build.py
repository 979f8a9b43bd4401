"""Build libsifsr_hip.so (gfx950) in-tree with hipcc.  Usage: python build.py [--force] [--debug]

Every .hip under csrc/ is compiled to an object (in parallel) and linked into ONE shared library
next to this file.  The library has no torch dependency: it links only libamdhip64, which resolves
to the copy PyTorch already loaded when the package is imported after torch.
"""
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "libsifsr_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra=(), lib=None, obj_dir=None, csrc=None):
    """lib / obj_dir / csrc: build a VARIANT of the library elsewhere, optionally from another source tree (tools/build_ab.sh,
    tools/build_ref.sh: same-device A/B through SIFSR_LIB)."""
    LIB_ = lib or LIB
    OBJ_ = obj_dir or OBJ
    CSRC = csrc or globals()["CSRC"]
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(os.path.dirname(CSRC)), "include", "sifsr_hip.h"))
    os.makedirs(OBJ_, exist_ok=True)
    jobs = []
    for f in srcs:
        src, obj = os.path.join(CSRC, f), os.path.join(OBJ_, f[:-4] + ".o")
        if force or _stale(obj, [src] + hdrs):
            jobs.append([hipcc, *FLAGS, *extra, "-c", src, "-o", obj])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with cf.ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJ_, f[:-4] + ".o") for f in srcs]
    if force or jobs or _stale(LIB_, objs):
        run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_, *objs])
    return LIB_


if __name__ == "__main__":
    extra = ["-Rpass-analysis=kernel-resource-usage"] if "--usage" in sys.argv else []
    print(build(force="--force" in sys.argv, verbose=True, extra=extra))
