#!/usr/bin/env python3
"""Idle gaps between the kernels of the bench command's timed slides (from a rocprofv3 --kernel-trace CSV). Tooling only.
usage: infer_gaps.py <dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
# a slide ends with argmax_kernel
slides, cur = [], []
for k in ks:
    cur.append(k)
    if "argmax" in k[2]:
        slides.append(cur); cur = []
for i, s in enumerate(slides):
    span = (s[-1][1] - s[0][0]) / 1e6
    busy = sum(e - b for b, e, _ in s) / 1e6
    gaps = sorted(((s[j + 1][0] - s[j][1]) / 1e3, s[j][2][:50], s[j + 1][2][:50]) for j in range(len(s) - 1))
    big = [g for g in gaps if g[0] > 20]
    print(f"slide {i}: {len(s)} kernels, span {span:.2f} ms, busy {busy:.2f} ms, idle {span - busy:.2f} ms; gaps > 20 us: {len(big)} totalling {sum(g[0] for g in big) / 1e3:.2f} ms")
    for g in gaps[-6:]:
        print(f"     {g[0]:9.1f} us after {g[1]} before {g[2]}")
if len(slides) > 1:
    for i in range(len(slides) - 1):
        print(f"between slide {i} and {i + 1}: {(slides[i + 1][0][0] - slides[i][-1][1]) / 1e3:.1f} us")
