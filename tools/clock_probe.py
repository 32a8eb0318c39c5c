"""dev probe: the shader clock the GPU sustains while the large plain MLP layer runs back to back (fp32 MFMA on random operands), read
from sysfs / rocm-smi while a second thread keeps the queue full.  The roofline fractions in bench.py are against the NOMINAL 2.4 GHz
peak (157.3 TFLOP/s); this says what the same kernels are worth against the clock the chip actually holds under that load."""
import glob
import os
import re
import subprocess
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pointrcnn_amd import ops  # noqa: E402


def sclk_mhz():
    out = []
    for f in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        try:
            for ln in open(f):
                if "*" in ln:
                    out.append(int(re.search(r"(\d+)Mhz", ln).group(1)))
        except OSError:
            pass
    if out:
        return max(out)
    try:
        t = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
        m = re.findall(r"sclk clock level:.*\((\d+)Mhz\)", t)
        return max(int(x) for x in m) if m else None
    except Exception:
        return None


def main():
    dev = torch.device("cuda:0")
    r = np.random.default_rng(0)
    rows, K, N = 32768, 512, 512
    x = torch.from_numpy(r.normal(size=(rows, K)).astype(np.float32)).to(dev)
    lin = ops.PackedLinear(torch.from_numpy((r.normal(size=(N, K)) * 0.05).astype(np.float32)).to(dev),
                           torch.zeros(N, device=dev), relu=True)
    for _ in range(20):
        ops.mlp_rows(x, lin)
    torch.cuda.synchronize()
    print("idle sclk:", sclk_mhz(), "MHz", flush=True)
    stop = False

    def pump():
        while not stop:
            for _ in range(50):
                ops.mlp_rows(x, lin)
            torch.cuda.synchronize()

    th = threading.Thread(target=pump)
    th.start()
    time.sleep(1.0)
    samples = []
    t0 = time.time()
    while time.time() - t0 < 3.0:
        v = sclk_mhz()
        if v:
            samples.append(v)
        time.sleep(0.05)
    stop = True
    th.join()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(200):
        ops.mlp_rows(x, lin)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / 200 * 1e3
    tf = 2.0 * rows * K * N / (us * 1e-6) / 1e12
    if samples:
        med = float(np.median(samples))
        print("under load: sclk min %d / median %d / max %d MHz over %d samples" % (min(samples), med, max(samples), len(samples)))
        print("layer 32768x512->512: %.1f us = %.1f TFLOP/s = %.3f of the nominal 157.3, %.3f of the peak at the median clock"
              % (us, tf, tf / 157.3, tf / (157.3 * med / 2400.0)))
    else:
        print("no clock reading available; layer %.1f us = %.1f TFLOP/s" % (us, tf))


if __name__ == "__main__":
    main()
