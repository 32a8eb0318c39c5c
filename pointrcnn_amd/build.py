"""Build libprcnn_pointops.so (all HIP kernels + the C ABI) for gfx950, in-tree.

    python -m pointrcnn_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with the tree (gpurun snapshot),
so a GPU box never needs to compile.  Flags: -ffp-contract=off is part of the arithmetic contract
(include/prcnn_pointops.h): no implicit FMA anywhere, explicit MFMA/fma only where written.
"""
import fcntl
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
# PRCNN_POINTOPS_LIB selects an alternative build of the library (A/B comparisons on one box)
LIB = os.environ.get("PRCNN_POINTOPS_LIB") or os.path.join(LIBDIR, "libprcnn_pointops.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fPIC", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function"]


# per-file extra flags.  fps.hip: finite-math-only lets fminf/fmaxf lower to bare v_min_f32/v_max_f32 (no NaN
# canonicalisation); it permits no reassociation or contraction, so results are unchanged for finite inputs.
# no-slp-vectorize: packed v_pk_*_f32 have no throughput advantage on gfx950 and cost operand-shuffle movs.
EXTRA_FLAGS = {"fps.hip": ["-ffinite-math-only", "-fno-slp-vectorize"]}


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "prcnn_pointops.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """Compile every .hip translation unit (in parallel) and link the shared library.  Serialised across processes
    by a lock file; the .so is linked to a temporary name and renamed into place, so a concurrent loader never sees
    a half-written library."""
    if not force and not _stale():
        return LIB
    if not os.path.exists(HIPCC):
        raise RuntimeError("hipcc not found at %s and %s is missing or stale" % (HIPCC, LIB))
    os.makedirs(LIBDIR, exist_ok=True)
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not _stale():          # another process built it while we waited
            return LIB
        objdir = os.path.join(LIBDIR, "obj")
        os.makedirs(objdir, exist_ok=True)
        procs = []
        for src in sources():
            obj = os.path.join(objdir, os.path.basename(src).replace(".hip", ".o"))
            procs.append((src, obj, subprocess.Popen([HIPCC] + FLAGS + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj],
                                                     stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs = []
        for src, obj, p in procs:
            out, _ = p.communicate()
            if p.returncode != 0:
                raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode()))
            if verbose and out.strip():
                print(out.decode())
            objs.append(obj)
        tmp = LIB + ".tmp.%d" % os.getpid()
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs, check=True)
        os.replace(tmp, LIB)
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
