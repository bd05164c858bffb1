#!/usr/bin/env python3
"""Per-layer averages of PMC counters from rocprofv3 counter_collection CSVs (layers labelled by launch order).
Tooling only. usage: pmc_layers.py <dir> [<dir> ...]"""
import collections
import csv
import glob
import sys
sys.path.insert(0, __file__.rsplit("/", 1)[0])
from trace_summary import SEQ

for path in sys.argv[1:]:
    f = glob.glob(path + "/**/*counter_collection.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    by_disp = collections.OrderedDict()
    for r in rows:
        by_disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"]})[r["Counter_Name"]] = float(r["Counter_Value"])
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    pos = None
    for d in sorted(by_disp):
        e = by_disp[d]
        n = e["name"]
        if "stem_pool" in n:
            pos, lab = 0, "stem+pool"
        elif "conv3x3_kernel" in n and pos is not None and pos < len(SEQ):
            lab = SEQ[pos]
            pos += 1
        else:
            continue
        for k, v in e.items():
            if k != "name":
                agg[lab][k].append(v)
    names = sorted({k for a in agg.values() for k in a})
    print("layer        " + " ".join(f"{k[-18:]:>18s}" for k in names))
    for lab in ["stem+pool"] + sorted(set(SEQ), key=SEQ.index):
        if lab in agg:
            print(f"{lab:12s} " + " ".join(f"{sum(agg[lab][k]) / max(1, len(agg[lab][k])):18.0f}" for k in names))
