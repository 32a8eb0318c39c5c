"""dev tool: a variant build of the library -- ONE translation unit recompiled with extra -D flags, every other object reused -- into
pointrcnn_amd/lib/libprcnn_<tag>.so; select it at run time with PRCNN_POINTOPS_LIB=<path> (pointrcnn_amd/build.py).
    python tools/build_variant.py iou3d.hip sweeptiming -DSWEEP_TIMING"""
import os
import subprocess
import sys
sys.path.insert(0, ".")
from pointrcnn_amd import build as b

unit, tag, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
b.build(verbose=False)
objdir = os.path.join(b.LIBDIR, "obj")
obj = os.path.join(objdir, "%s_%s.o" % (unit.replace(".hip", ""), tag))
subprocess.run([b.HIPCC] + b.FLAGS + b.EXTRA_FLAGS.get(unit, []) + flags + ["-c", os.path.join(b.CSRC, unit), "-o", obj], check=True,
               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
objs = [os.path.join(objdir, os.path.basename(s).replace(".hip", ".o")) for s in b.sources() if not s.endswith("/" + unit)] + [obj]
out = os.path.join(b.LIBDIR, "libprcnn_%s.so" % tag)
subprocess.run([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, check=True)
print("built", out)
