"""dev tool: sum rocprofv3 PMC counters per kernel name over the LAST step of a short eager bench run.
    rocprofv3 --pmc C1 C2 ... --kernel-trace --output-format csv -d DIR -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline \
        --no-variants --graph off --streams 1
    python tools/pmc_by_kernel.py DIR [name-substring ...]"""
import collections
import csv
import glob
import sys

root, pats = sys.argv[1], sys.argv[2:]
f = glob.glob(root + "/**/*_counter_collection.csv", recursive=True)[0]
per = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    d = per.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"].split("(")[0].replace("void ", ""), "c": collections.Counter()})
    d["c"][r["Counter_Name"]] += float(r["Counter_Value"])
ds = list(per.values())
starts = [i for i, d in enumerate(ds) if "fps_sort_kernel" in d["name"]]
first = starts[-2] if len(starts) >= 2 else 0          # two sorts per step (levels 0 and 1): the last step starts at the second-to-last
names = sorted({c for d in ds for c in d["c"]})
print("%-52s %s" % ("kernel", " ".join("%16s" % n[-16:] for n in names)))
for d in ds[first:]:
    if pats and not any(p in d["name"] for p in pats):
        continue
    print("%-52s %s" % (d["name"][:52], " ".join("%16.5g" % d["c"][n] for n in names)))
