"""dev tool: ablation builds of the library (mlp.hip recompiled with -DPRCNN_ABL=<mask>, every other object reused) into
pointrcnn_amd/lib/libprcnn_abl<mask>.so; select one at run time with PRCNN_POINTOPS_LIB=<path> (pointrcnn_amd/build.py)."""
import os
import subprocess
import sys
sys.path.insert(0, ".")
from pointrcnn_amd import build as b

b.build(verbose=False)
objdir = os.path.join(b.LIBDIR, "obj")
for mask in [int(a) for a in sys.argv[1:]]:
    obj = os.path.join(objdir, "mlp_abl%d.o" % mask)
    subprocess.run([b.HIPCC] + b.FLAGS + ["-DPRCNN_ABL=%d" % mask, "-c", os.path.join(b.CSRC, "mlp.hip"), "-o", obj], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    objs = [os.path.join(objdir, os.path.basename(s).replace(".hip", ".o")) for s in b.sources() if not s.endswith("mlp.hip")] + [obj]
    out = os.path.join(b.LIBDIR, "libprcnn_abl%d.so" % mask)
    subprocess.run([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, check=True)
    print("built", out)
