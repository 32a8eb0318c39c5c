#!/usr/bin/env python3
"""dev tool: per-dispatch PMC table of the last RPN step from rocprofv3 --pmc CSVs: pmc_table.py <dir> [<dir2> ...]"""
import collections
import csv
import glob
import sys

disp = collections.OrderedDict()
for root in sys.argv[1:]:
    for f in glob.glob(root + "/**/*_counter_collection.csv", recursive=True):
        per = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            d = per.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"].split("(")[0].replace("void ", ""), "c": collections.Counter()})
            d["c"][r["Counter_Name"]] += float(r["Counter_Value"])
        ds = list(per.values())
        last = ([i - 1 for i, d in enumerate(ds) if ("fps_pruned_kernel<16>" in d["name"] or "fps_slot_kernel<16" in d["name"])] or
                [i for i, d in enumerate(ds) if "fps_reg_kernel<1024" in d["name"]])[-1]
        for i, d in enumerate(ds[last:]):
            e = disp.setdefault(i, {"name": d["name"], "c": collections.Counter()})
            assert e["name"] == d["name"]
            e["c"].update(d["c"])
names = sorted({k for d in disp.values() for k in d["c"]})
print("%-34s " % "kernel" + " ".join("%14s" % n[-14:] for n in names))
for i, d in disp.items():
    print("%-34s " % d["name"][:34] + " ".join("%14.4g" % d["c"][n] for n in names))
