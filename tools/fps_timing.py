"""dev tool: cycle accounting of fps_pruned_kernel's sample loop (frame 0, per wave): builds the library with fps.hip compiled
-DPRCNN_FPS_TIMING (s_memtime brackets around update / slot write + barrier / slot read + reduce) and prints the sums."""
import ctypes
import os
import subprocess
import sys
sys.path.insert(0, ".")
from pointrcnn_amd import build as b

b.build(verbose=False)
objdir = os.path.join(b.LIBDIR, "obj")
obj = os.path.join(objdir, "fps_timing.o")
subprocess.run([b.HIPCC] + b.FLAGS + b.EXTRA_FLAGS.get("fps.hip", []) + ["-DPRCNN_FPS_TIMING", "-c", os.path.join(b.CSRC, "fps.hip"), "-o", obj], check=True)
objs = [os.path.join(objdir, os.path.basename(s).replace(".hip", ".o")) for s in b.sources() if not s.endswith("fps.hip")] + [obj]
out = os.path.join(b.LIBDIR, "libprcnn_fpstiming.so")
subprocess.run([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, check=True)
print("built", out)
if "--run" in sys.argv:
    if os.environ.get("PRCNN_POINTOPS_LIB") != out:            # (build.py reads the override at import time)
        os.environ["PRCNN_POINTOPS_LIB"] = out
        os.execv(sys.executable, [sys.executable] + sys.argv)
    import torch
    from pointrcnn_amd import _cabi, ops, rpn
    dev = torch.device("cuda:0")
    xyz = rpn.synthetic_clouds(32, 16384, seed0=100, device=dev)
    for _ in range(2):
        idx = ops.furthest_point_sample(xyz, 4096)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); idx = ops.furthest_point_sample(xyz, 4096); e.record(); torch.cuda.synchronize()
    print("fps 16384 -> 4096, bs32: %.1f us (instrumented)" % (s.elapsed_time(e) * 1e3))
    buf = (ctypes.c_ulonglong * 144)()
    L = ctypes.CDLL(out)
    L.prcnn_fps_timing_read.argtypes = [ctypes.c_void_p]
    assert L.prcnn_fps_timing_read(buf) == 0
    print("wave   update-phase  write+barrier  read+reduce   updates   loop-total   (cycles, sums over 4095 samples)")
    for w in range(16):
        d = buf[w * 8:w * 8 + 5]
        print("%4d %14d %14d %12d %9d %12d   per sample: %6.0f %6.0f %6.0f" % (w, d[0], d[1], d[2], d[3], d[4], d[0] / 4095, d[1] / 4095, d[2] / 4095))
        u = [buf[w * 8 + 5], buf[w * 8 + 6], buf[w * 8 + 7], buf[128 + w]]
        print("       per UPDATE: distances %5.0f  wave-max %5.0f  owner search %5.0f  candidate fetch %5.0f  (not updating: %4.0f per sample)" %
              (u[0] / d[3], u[1] / d[3], u[2] / d[3], u[3] / d[3], (d[0] - sum(u)) / max(1, 4095 - d[3])))
