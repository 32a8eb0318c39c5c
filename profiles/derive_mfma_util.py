#!/usr/bin/env python3
"""MFMA-pipe utilisation per MLP kernel launch of one RPN step, from a rocprofv3 PMC pass
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv \
        -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --graph off --streams 1
usage: derive_mfma_util.py <dir> > profiles/r01_mfma_util.txt
SQ_VALU_MFMA_BUSY_CYCLES counts per SIMD, SQ_BUSY_CU_CYCLES per CU (4 SIMDs): utilisation = MFMA_BUSY / (4 * BUSY_CU)."""
import collections
import csv
import glob
import sys

root = sys.argv[1]
f = glob.glob(root + "/**/*_counter_collection.csv", recursive=True)[0]
per = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    d = per.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"].split("(")[0].replace("void ", ""), "c": collections.Counter()})
    d["c"][r["Counter_Name"]] += float(r["Counter_Value"])
ds = list(per.values())
first = ([i - 1 for i, d in enumerate(ds) if ("fps_pruned_kernel<16>" in d["name"] or "fps_slot_kernel<16" in d["name"])] or
         [i for i, d in enumerate(ds) if "fps_reg_kernel<1024" in d["name"]])[-1]
print("# MFMA pipe utilisation of the MLP launches of one bs32 RPN step (rocprofv3 PMC, single stream, eager)")
print("# util = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES); MOPS_F32 x 512 = fp32 MFMA FLOPs issued; the split-bf16 kernels issue")
print("# bf16 MFMAs: SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512 (= 6 x their fp32-equivalent FLOPs with six terms), printed when collected")
print("%-46s %14s %14s %8s %12s" % ("kernel", "MFMA_BUSY", "BUSY_CU", "util", "MFMA GFLOP"))
tb = tc = 0.0
for d in ds[first:]:
    if not d["name"].startswith(("mlp_", "sa_xyz")):
        continue
    b, c, m = d["c"]["SQ_VALU_MFMA_BUSY_CYCLES"], d["c"]["SQ_BUSY_CU_CYCLES"], d["c"]["SQ_INSTS_VALU_MFMA_MOPS_F32"]
    mb = d["c"].get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)         # (present when the pass collected it: bf16 MFMA FLOPs issued / 512)
    tb += b
    tc += c
    print("%-46s %14.4g %14.4g %7.1f%% %12.1f%s" % (d["name"][:46], b, c, 100 * b / (4 * c) if c else 0, m * 512 / 1e9,
                                                     "   bf16 MFMA GFLOP %9.1f" % (mb * 512 / 1e9) if mb else ""))
print("%-46s %14.4g %14.4g %7.1f%%" % ("all MLP launches (time-weighted)", tb, tc, 100 * tb / (4 * tc) if tc else 0))
