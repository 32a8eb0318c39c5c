"""Build libprcnn_pointops.so (all HIP kernels + the C ABI) for gfx950, in-tree.

    python -m pointrcnn_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with the tree (gpurun snapshot),
so a GPU box never needs to compile.  Flags: -ffp-contract=off is part of the arithmetic contract
(include/prcnn_pointops.h): no implicit FMA anywhere, explicit MFMA/fma only where written.
"""
import fcntl
import glob
import hashlib
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
# canonicalisation); it permits no reassociation or contraction, so results are unchanged for FINITE inputs -- which is the
# documented domain of prcnn_fps (include/prcnn_pointops.h; a cloud with NaN/Inf coordinates has no farthest point under
# any rule).  no-slp-vectorize: the distance loops are written on explicit float2 pairs (v_pk_*_f32, the only fp32 VALU
# form that issues at full rate on gfx950); the SLP vectoriser's own pairings on top of that cost
# operand-shuffle movs.
EXTRA_FLAGS = {"fps.hip": ["-ffinite-math-only", "-fno-slp-vectorize"]}
BUILD_ID_TAG = b"PRCNN_BUILD_ID="


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def source_id():
    """hex digest of everything the library is built from: kernel sources, headers, compile flags"""
    h = hashlib.sha1()
    deps = sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "..", "include", "prcnn_pointops.h")]
    for d in deps:
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    h.update(repr((FLAGS, sorted(EXTRA_FLAGS.items()))).encode())
    return h.hexdigest()[:20]


def library_id(path=None):
    """the digest baked into a built library (prcnn_build_id()), read from the file without loading it; None if absent"""
    path = path or LIB
    try:
        with open(path, "rb") as f:
            blob = f.read()
    except OSError:
        return None
    i = blob.find(BUILD_ID_TAG)
    return blob[i + len(BUILD_ID_TAG): i + len(BUILD_ID_TAG) + 20].decode("ascii", "replace") if i >= 0 else None


def have_sources():
    return os.path.isdir(CSRC) and bool(sources())


def _stale():
    """the library is missing, or was built from other sources / flags than the ones next to it"""
    if not os.path.exists(LIB):
        return True
    return have_sources() and library_id() != source_id()


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
        bid = source_id()
        # an object is reused when its own source, every header and its flags are what it was compiled from (digest kept next to it);
        # `force` recompiles everything (what __graft_entry__.build() asks for)
        hdr = hashlib.sha1()
        for d in sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "..", "include", "prcnn_pointops.h")]:
            with open(d, "rb") as f:
                hdr.update(f.read())
        objs = []
        for src in sources():
            obj = os.path.join(objdir, os.path.basename(src).replace(".hip", ".o"))
            objs.append(obj)
            extra = EXTRA_FLAGS.get(os.path.basename(src), [])
            if os.path.basename(src) == "cabi_common.hip":
                extra = extra + ['-DPRCNN_BUILD_ID="%s%s"' % (BUILD_ID_TAG.decode(), bid)]
            with open(src, "rb") as f:
                digest = hashlib.sha1(f.read() + hdr.digest() + repr(FLAGS + extra).encode()).hexdigest()
            stamp = obj + ".digest"
            if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == digest:
                continue
            if os.path.exists(stamp):
                os.remove(stamp)
            procs.append((src, stamp, digest, subprocess.Popen([HIPCC] + FLAGS + extra + ["-c", src, "-o", obj],
                                                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        for src, stamp, digest, p in procs:
            out, _ = p.communicate()
            if p.returncode != 0:
                raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode()))
            if verbose and out.strip():
                print(out.decode())
            with open(stamp, "w") as f:
                f.write(digest)
        tmp = LIB + ".tmp.%d" % os.getpid()
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs, check=True)
        os.replace(tmp, LIB)
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
