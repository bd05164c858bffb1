#!/usr/bin/env python3
"""Average duration of every conv launch by its POSITION in the forward (stem, then 20 convs) from a rocprofv3 kernel trace of the
bench command; full-size micro-batches only. Tooling only.  usage: trace_positions.py <dir>"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
fw = []
for r in rows:
    n = r["Kernel_Name"]; us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "stem_pool" in n: fw.append([us])
    elif "conv3x3_kernel" in n and fw: fw[-1].append(us)
med = sorted(x[0] for x in fw)[len(fw) // 2]
fw = [x for x in fw if x[0] >= 0.8 * med and len(x) == 17]
names = ["stem+pool", "L1.0.c1", "L1.0.c2", "L1.1.c1", "L1.1.c2(out16)", "L2.0.c1+ds(wide)", "L2.0.c2", "L2.1.c1", "L2.1.c2(out16)", "L3.0.c1+ds(wide)", "L3.0.c2", "L3.1.c1",
         "L3.1.c2", "L4.0.c1+ds", "L4.0.c2", "L4.1.c1", "L4.1.c2"]
tot = sum(sum(x[:17]) for x in fw) / len(fw)
for i, nm in enumerate(names):
    a = sum(x[i] for x in fw) / len(fw)
    print(f"{nm:20s} {a:8.1f} us  {100 * a / tot:5.1f} %")
print(f"forwards {len(fw)}, total per forward {tot / 1e3:.2f} ms")
