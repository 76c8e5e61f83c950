#!/usr/bin/env python3
"""Development helper: per-kernel mean of every counter in a rocprofv3 --pmc output directory."""
import csv, glob, os, sys, collections, json

def summarise(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    regs = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            regs[k] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"])
    return {k: {"regs(v,a,s,lds,scratch)": regs[k], **{c: sum(v) / len(v) for c, v in cs.items()}, "launches": max(len(v) for v in cs.values())} for k, cs in acc.items()}

if __name__ == "__main__":
    out = summarise(sys.argv[1])
    for k, v in out.items():
        if len(sys.argv) > 2 and sys.argv[2] not in k:
            continue
        print(k)
        for c, x in v.items():
            print("    %-28s %s" % (c, ("%.4g" % x) if isinstance(x, float) else x))
