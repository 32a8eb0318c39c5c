#!/usr/bin/env python3
"""Recompute bench.py's `roofline.frac` (fused per-point MLP family) from the committed profiler summaries alone.

    python profiles/recompute_roofline.py [round tag, default r05]

Inputs (all under profiles/):
  <tag>_final_kernel_stats.txt   rocprofv3 --kernel-trace --stats of the TIMED configuration (`python bench.py`: 20 batches in flight,
                                 hipGraph replay): per kernel name, calls and total duration
  <tag>_final_kernel_stats_streams1.txt   the same with --streams 1 (one batch on the chip: what bench.py's HIP events time)
  <tag>_mfma_util.txt            rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_{F32,BF16} per launch of one bs32 step: MFMA flops the matrix
                                 pipe was given, by pipe (x 512 flops per counted operation)
  <tag>_final_bench.json         the bench line whose roofline.frac is being checked

Definition (bench.py `roofline.definition`):
  frac = sum over the family's launches of (MFMA flops on the pipe / dense peak of THAT pipe) / sum of the launches' durations
       = (fp32 GFLOP / 157.3 TFLOP/s + bf16 GFLOP / 2500 TFLOP/s) per step  /  family kernel time per step in the timed configuration
"""
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
FP32_PEAK, BF16_PEAK = 157.3, 2500.0          # TFLOP/s dense (MI355X_MICROARCH.md)
FAMILY = ("mlp_", "sa_xyz_chain")


def family(name):
    return name.replace("void ", "").startswith(FAMILY)


def kernel_stats(path):
    """-> {kernel name: (calls, total_us)}"""
    out = {}
    for ln in open(path):
        m = re.match(r"^(.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", ln)
        if m and not ln.startswith(("#", "kernel")):
            out[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)))
    return out


def mfma_flops(path):
    """-> (fp32 GFLOP, bf16 GFLOP) of the family's launches of one step"""
    f32 = bf16 = 0.0
    for ln in open(path):
        if ln.startswith("#") or not family(ln.split("  ")[0].strip()):
            continue
        m = re.match(r"^(\S.*?)\s+[\d.e+]+\s+[\d.e+]+\s+[\d.]+%\s+([\d.]+)(?:\s+bf16 MFMA GFLOP\s+([\d.]+))?", ln)
        if m:
            f32 += float(m.group(2))
            bf16 += float(m.group(3) or 0.0)
    return f32, bf16


def one(ks, f32, bf16, what):
    # steps in the trace = launches of a kernel that runs once per step (level-0 FPS)
    once = [c for n, (c, _) in ks.items() if "fps_slot_kernel<16" in n or "fps_pruned_kernel<16>" in n]
    steps = once[0] if once else None
    fam_us = sum(t for n, (_, t) in ks.items() if family(n))
    if not steps or fam_us <= 0:
        raise SystemExit("kernel stats do not hold the family / the per-step marker kernel")
    us_per_step = fam_us / steps
    at_peak_us = (f32 / FP32_PEAK + bf16 / BF16_PEAK) * 1e3          # GFLOP / (TFLOP/s) = ms -> us
    frac = at_peak_us / us_per_step
    print("%s: %d steps in the trace, family kernel time %.1f us per step, time at the pipes' peaks %.1f us -> frac %.4f" %
          (what, steps, us_per_step, at_peak_us, frac))
    return frac


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
    f32, bf16 = mfma_flops(os.path.join(HERE, tag + "_mfma_util.txt"))
    print("MFMA work per step (PMC counters): %.1f GFLOP on the fp32 pipe + %.1f GFLOP on the bf16 pipe" % (f32, bf16))
    timed = one(kernel_stats(os.path.join(HERE, tag + "_final_kernel_stats.txt")), f32, bf16, "timed configuration (20 batches in flight)")
    alone = None
    p1 = os.path.join(HERE, tag + "_final_kernel_stats_streams1.txt")
    if os.path.exists(p1):
        alone = one(kernel_stats(p1), f32, bf16, "one batch on the chip (--streams 1)            ")
    bj = os.path.join(HERE, tag + "_final_bench.json")
    if os.path.exists(bj):
        line = json.loads(open(bj).read().strip().splitlines()[-1])
        got = line["roofline"]["frac"]
        print("roofline.frac of the bench line (HIP events, one batch on the chip): %.4f" % got)
        if alone:
            print("  ratio to the single-stream trace %.3f, to the timed configuration %.3f" % (got / alone, got / timed))
            return 0 if abs(got / alone - 1.0) <= 0.05 else 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
