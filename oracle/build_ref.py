#!/usr/bin/env python3
"""Build oracle/_ref/libprcnn_ref.so: the reference's OWN native sources, compiled for the host.

TEST INFRASTRUCTURE.  Compiles, from where they lie under /root/reference (read-only, never
copied into this repo), the four files
    lib/utils/iou3d/src/iou3d.cpp, iou3d_kernel.cu
    lib/utils/roipool3d/src/roipool3d.cpp, roipool3d_kernel.cu
with plain g++ against oracle/ref_shim (stub torch/CUDA headers + a sequential grid emulator).
The two .cu files are streamed through a regex that turns `k<<<grid, block>>>(args)` into
`PRCNN_LAUNCH(k, grid, block, args)` and piped to g++ on stdin -- no reference text is written
to disk.  Outputs go ONLY to oracle/_ref/ (git-ignored; it travels to the GPU box).

The reference's own build system (setup.py + CUDAExtension/nvcc) is NOT used: nvcc is absent.
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("PRCNN_REFERENCE", "/root/reference")
OUT = os.path.join(HERE, "_ref")
CXXFLAGS = ["-O2", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-std=c++14", "-w",
            "-I", os.path.join(HERE, "ref_shim")]
LAUNCH = re.compile(r"(\w+)\s*<<<(.*?)>>>\s*\(", re.S)


def have_reference():
    return os.path.isfile(os.path.join(REF, "lib/utils/iou3d/src/iou3d_kernel.cu"))


def build(verbose=True):
    if not have_reference():
        raise RuntimeError("reference checkout not found at %s" % REF)
    os.makedirs(OUT, exist_ok=True)
    objs = []
    for rel in ("lib/utils/iou3d/src/iou3d_kernel.cu", "lib/utils/roipool3d/src/roipool3d_kernel.cu"):
        src = open(os.path.join(REF, rel)).read()
        src = LAUNCH.sub(lambda m: "PRCNN_LAUNCH(%s, %s, " % (m.group(1), m.group(2)), src)
        obj = os.path.join(OUT, os.path.basename(rel).replace(".cu", "_cu.o"))
        cmd = ["g++"] + CXXFLAGS + ["-include", "prcnn_ref_shim.h", "-x", "c++", "-c", "-", "-o", obj]
        subprocess.run(cmd, input=src.encode(), check=True)
        objs.append(obj)
    for rel in ("lib/utils/iou3d/src/iou3d.cpp", "lib/utils/roipool3d/src/roipool3d.cpp"):
        obj = os.path.join(OUT, os.path.basename(rel).replace(".cpp", "_cpp.o"))
        subprocess.run(["g++"] + CXXFLAGS + ["-c", os.path.join(REF, rel), "-o", obj], check=True)
        objs.append(obj)
    wobj = os.path.join(OUT, "ref_wrappers.o")
    subprocess.run(["g++"] + CXXFLAGS + ["-c", os.path.join(HERE, "ref_wrappers.cpp"), "-o", wobj], check=True)
    so = os.path.join(OUT, "libprcnn_ref.so")
    subprocess.run(["g++", "-shared", "-o", so] + objs + [wobj, "-lm"], check=True)
    for o in objs + [wobj]:
        os.remove(o)
    if verbose:
        print("built", so)
    return so


if __name__ == "__main__":
    build()
    sys.exit(0)
