#!/usr/bin/env python3
"""Derive per-kernel-family HBM bytes per step from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE collected in
SEPARATE runs, as /opt/skills/guides/MI355X_MICROARCH.md prescribes) of
    python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --graph off --streams 1
usage: derive_hbm_traffic.py <dir with *FETCH_SIZE*/ and *WRITE_SIZE*/ results> <out.json>"""
import collections
import csv
import glob
import json
import sys


def load(root, counter):
    f = glob.glob("%s*%s/*/*_counter_collection.csv" % (root, counter))[0]
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        d = disp.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"].split("(")[0].replace("void ", ""), "v": 0.0})
        d["v"] += float(r["Counter_Value"])
    ds = list(disp.values())
    # a step starts with SA1's FPS: fps_sort_kernel + fps_pruned_kernel<16> (default) or fps_reg_kernel<1024, 16>
    firsts = [i - 1 for i, d in enumerate(ds) if ("fps_pruned_kernel<16>" in d["name"] or "fps_slot_kernel<16" in d["name"])] or \
             [i for i, d in enumerate(ds) if "fps_reg_kernel<1024" in d["name"]]
    return ds[firsts[-1]:]        # the dispatches of the last full step


CORRECTION = ("hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: counters are KiB; on gfx950 FETCH_SIZE reads exactly 1/2 of a wide "
              "coalesced stream (MI355X_MICROARCH.md, HBM section); WRITE_SIZE taken as is")


def derive(root):
    """per kernel family of the last full step: launches, FETCH_SIZE / WRITE_SIZE (KiB), corrected HBM bytes per step and per launch"""
    f, w = load(root, "FETCH_SIZE"), load(root, "WRITE_SIZE")
    assert [d["name"] for d in f] == [d["name"] for d in w], "dispatch order differs between the two passes"
    fam = collections.OrderedDict()
    for a, b in zip(f, w):
        key = "mlp" if a["name"].startswith(("mlp_", "sa_xyz_chain")) else a["name"].split("<")[0]
        d = fam.setdefault(key, {"launches": 0, "fetch": 0.0, "write": 0.0})
        d["launches"] += 1
        d["fetch"] += a["v"]
        d["write"] += b["v"]
    res = {}
    for k, d in fam.items():
        hbm = (2 * d["fetch"] + d["write"]) * 1024
        res[k] = {"launches_per_step": d["launches"], "FETCH_SIZE_KiB": round(d["fetch"], 1),
                  "WRITE_SIZE_KiB": round(d["write"], 1), "hbm_bytes_per_step_corrected": int(hbm),
                  "hbm_bytes_per_launch_corrected": int(hbm / d["launches"])}
    return res


def main():
    root, out = sys.argv[1], sys.argv[2]
    res = derive(root)
    json.dump({"command": "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace -- python bench.py --steps 3 --warmup 1 "
                          "--no-cpu-baseline --no-roofline --graph off --streams 1   (two separate passes)",
               "correction": CORRECTION, "per_step_bs32": res}, open(out, "w"), indent=1)
    for k, v in res.items():
        print(k, v)


if __name__ == "__main__":
    main()
