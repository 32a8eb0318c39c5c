"""dev tool: cycle accounting of fps_pruned_kernel's sample loop (frame 0, per wave): builds the library with fps.hip compiled
-DPRCNN_FPS_TIMING (s_memtime brackets around update / slot write + barrier / slot read + reduce) and prints the sums."""
import ctypes
import os
import subprocess
import sys
sys.path.insert(0, ".")
from pointrcnn_amd import build as b

nobuild = "--no-build" in sys.argv          # the GPU box runs the library built in the container
if not nobuild:
    b.build(verbose=False)
objdir = os.path.join(b.LIBDIR, "obj")
tw = [a.split("=")[1] for a in sys.argv if a.startswith("--wave=")]          # time stamps on one wave only (slot kernel)
obj = os.path.join(objdir, "fps_timing%s.o" % ("_w" + tw[0] if tw else ""))
if not nobuild:
  subprocess.run([b.HIPCC] + b.FLAGS + b.EXTRA_FLAGS.get("fps.hip", []) + ["-DPRCNN_FPS_TIMING"] + (["-DPRCNN_FPS_TIMING_WAVE=" + tw[0]] if tw else []) + ["-c", os.path.join(b.CSRC, "fps.hip"), "-o", obj], check=True)
objs = [os.path.join(objdir, os.path.basename(s).replace(".hip", ".o")) for s in b.sources() if not s.endswith("fps.hip")] + [obj]
out = os.path.join(b.LIBDIR, "libprcnn_fpstiming%s.so" % ("_w" + tw[0] if tw else ""))
if not nobuild:
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
    buf = (ctypes.c_ulonglong * 400)()
    L = ctypes.CDLL(out)
    L.prcnn_fps_timing_read.argtypes = [ctypes.c_void_p]
    assert L.prcnn_fps_timing_read(buf) == 0
    print("wave   update-phase  write+barrier  read+reduce   updates   loop-total   (cycles, sums over 4095 samples)")
    for w in range(16 if buf[3] else 0):          # fps_pruned_kernel<16> ran only with PRCNN_FPS_SLOTS=0
        d = buf[w * 8:w * 8 + 5]
        print("%4d %14d %14d %12d %9d %12d   per sample: %6.0f %6.0f %6.0f" % (w, d[0], d[1], d[2], d[3], d[4], d[0] / 4095, d[1] / 4095, d[2] / 4095))
        u = [buf[w * 8 + 5], buf[w * 8 + 6], buf[w * 8 + 7], buf[128 + w]]
        print("       per UPDATE: distances %5.0f  wave-max %5.0f  owner search %5.0f  candidate fetch %5.0f  (not updating: %4.0f per sample)" %
              (u[0] / d[3], u[1] / d[3], u[2] / d[3], u[3] / d[3], (d[0] - sum(u)) / max(1, 4095 - d[3])))
    print("fps_slot_kernel<16> (the default at this size), frame 0, cycles per sample: bound test | publish+barrier | collect  ;  per UPDATE: distances | max+slot masks | search+selects+candidate ; updates, live pairs per update")
    for w in ([int(tw[0])] if tw else range(16)):
        d = buf[144 + w * 16:144 + w * 16 + 13]
        n = max(1, d[6])
        if d[9]:             # fps_batch_kernel: per exchange ROUND (one or two samples)
            r = d[9]
            print("%4d  batch kernel: rounds %d (%.2f samples per round) | per round: update-phase %5.0f  pub+barrier %5.0f  collect+proof+test %5.0f  total %6.0f | per UPDATE: dist %5.0f  max %5.0f  search+second %5.0f | updates %5d (%.2f of rounds)  live pairs/update %.2f | full updates %d | proofs run by this wave %d, %5.0f cycles each" %
                  (w, r, 4095.0 / r, (d[1] + d[2] + d[3]) / r, d[4] / r, d[5] / r, d[8] / r, d[1] / n, d[2] / max(1, d[12]), d[3] / max(1, d[12]), d[6], d[6] / r, d[7] / n, d[12], d[11], d[10] / max(1, d[11])))
            continue
        print("%4d  test %5.0f  pub+barrier %5.0f  collect %5.0f | dist %5.0f  max %5.0f  search %5.0f | updates %5d (%.2f of samples)  pairs/update %.2f | loop total/sample %6.0f" %
              (w, d[0] / 4095, d[4] / 4095, d[5] / 4095, d[1] / n, d[2] / n, d[3] / n, d[6], d[6] / 4095.0, d[7] / n, d[8] / 4095))
