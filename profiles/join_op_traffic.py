#!/usr/bin/env python3
"""Add PMC byte counts to the HBM-class lines of the per-operator report.

    python -m pointrcnn_amd.opbench > opbench.jsonl
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <prefix>FETCH_SIZE -- python -m pointrcnn_amd.opbench
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <prefix>WRITE_SIZE -- python -m pointrcnn_amd.opbench
    join_op_traffic.py opbench.jsonl <prefix> out.jsonl

(two SEPARATE counter passes, kernel trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes).  Every line that names
its kernels ("kernels": [[name substring, grid size in work-items or 0], ...]) gets
  pmc_fetch_MB / pmc_write_MB   the counters as reported (KiB x 1024), averaged per call of the operator
  pmc_MB_low  = fetch + write                      (WRITE_SIZE equals the output bytes exactly wherever it can be checked)
  pmc_MB_high = 2 * fetch + write                  (the guide's gfx950 correction: FETCH_SIZE tallies a wide streaming read at
                                                    half its bytes; narrower gathers are uncalibrated, so the truth lies between)
  pmc_GBps_low/high and their fraction of the 6.3 TB/s copy ceiling, over the un-profiled launch time of the same line."""
import collections
import csv
import glob
import json
import sys


def load(prefix, counter):
    f = glob.glob("%s%s/*/*_counter_collection.csv" % (prefix, counter))[0]
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        d = disp.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"], "grid": int(r["Grid_Size"]), "v": 0.0})
        d["v"] += float(r["Counter_Value"])
    return list(disp.values())


def per_call(ds, sub, grid):
    v = [d["v"] for d in ds if sub in d["name"] and (grid == 0 or d["grid"] == grid)]
    return (sum(v) / len(v)) if v else None


def main():
    src, prefix, out = sys.argv[1:4]
    F, W = load(prefix, "FETCH_SIZE"), load(prefix, "WRITE_SIZE")
    with open(out, "w") as fo:
        for ln in open(src):
            ln = ln.strip()
            if not ln.startswith("{"):
                continue
            d = json.loads(ln)
            ks = d.get("kernels")
            if ks:
                f = [per_call(F, k, g) for k, g in ks]
                w = [per_call(W, k, g) for k, g in ks]
                if None not in f and None not in w:
                    fb, wb = sum(f) * 1024, sum(w) * 1024
                    sec = d["avg_launch_us"] * 1e-6
                    lo, hi = fb + wb, 2 * fb + wb
                    d.update(pmc_fetch_MB=round(fb / 1e6, 1), pmc_write_MB=round(wb / 1e6, 1), pmc_MB_low=round(lo / 1e6, 1),
                             pmc_MB_high=round(hi / 1e6, 1), pmc_GBps_low=round(lo / sec / 1e9, 1), pmc_GBps_high=round(hi / sec / 1e9, 1),
                             frac_of_6p3TBps_by_pmc_low=round(lo / sec / 6.3e12, 3), frac_of_6p3TBps_by_pmc_high=round(hi / sec / 6.3e12, 3))
            fo.write(json.dumps(d) + "\n")
            print(json.dumps(d))


if __name__ == "__main__":
    main()
