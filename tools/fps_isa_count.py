"""dev tool: static instruction count of the FPS sample loops (gfx950 ISA from hipcc -S), the input of the issue-bound model in
bench.py (`fps_kernel.issue_model`): a lone updating wave issues about one instruction per 4-5 cycles (tools/fps_timing.py), so
the per-sample latency of the serial chain is its instruction count, not its arithmetic.
usage: python tools/fps_isa_count.py   (needs hipcc; prints per kernel: instructions in the outermost loop, by class)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pointrcnn_amd import build  # noqa: E402


def main():
    src = os.path.join(build.CSRC, "fps.hip")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "fps.s")
        subprocess.run([build.HIPCC] + build.FLAGS + build.EXTRA_FLAGS["fps.hip"] + ["--cuda-device-only", "-S", src, "-o", out], check=True,
                       stderr=subprocess.DEVNULL)
        lines = open(out).read().splitlines()
    starts = [(i, m.group(1)) for i, l in enumerate(lines) for m in [re.match(r"^(_Z\w+):", l)] if m]
    for (i0, name), nxt in zip(starts, starts[1:] + [(len(lines), None)]):
        body = lines[i0:nxt[0]]
        labels = {m.group(1): k for k, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        best = None
        for k, l in enumerate(body):
            m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.search(r"s_branch\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < k:
                span = (labels[m.group(1)], k)
                if best is None or span[1] - span[0] > best[1] - best[0]:
                    best = span
        if best is None:
            continue
        ins = [l.strip().split()[0] for l in body[best[0]:best[1] + 1] if l.startswith("\t") and l.strip() and not l.strip().startswith((";", "."))]
        cls = {"valu": 0, "salu": 0, "lds": 0, "vmem": 0, "barrier": 0, "other": 0}
        for op in ins:
            if op.startswith("v_"): cls["valu"] += 1
            elif op == "s_barrier": cls["barrier"] += 1
            elif op.startswith("s_"): cls["salu"] += 1
            elif op.startswith("ds_"): cls["lds"] += 1
            elif op.startswith(("global_", "buffer_", "flat_")): cls["vmem"] += 1
            else: cls["other"] += 1
        print("%-52s loop instructions %5d  %s" % (name, len(ins), cls))


if __name__ == "__main__":
    main()
