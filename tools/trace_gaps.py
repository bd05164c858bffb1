#!/usr/bin/env python3
"""Per-queue busy time and idle gaps of the LAST training step in a rocprofv3 kernel trace (tools/train_profile.py run).
usage: trace_gaps.py <dir or kernel_trace.csv> [steps=6].  Tooling only."""
import csv
import sys
from collections import defaultdict
from pathlib import Path

p = Path(sys.argv[1])
f = p if p.is_file() else next(p.rglob("*kernel_trace.csv"))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step starts with the stem convolution of its forward pass
starts = [i for i, r in enumerate(rows) if "stem_kernel" in r["Kernel_Name"]]
assert len(starts) >= steps, (len(starts), steps)
lo, hi = starts[-2], starts[-1]   # the last but one step (the last one has no successor to delimit it)
step = rows[lo:hi]
t0, t1 = int(step[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in step)
print(f"last step: {len(step)} kernels, wall {(t1 - t0) / 1e6:.3f} ms")
byq = defaultdict(list)
for r in step:
    byq[r["Queue_Id"]].append(r)
for q, rs in byq.items():
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rs, rs[1:])]
    pos = [g for g in gaps if g > 0]
    span = int(rs[-1]["End_Timestamp"]) - int(rs[0]["Start_Timestamp"])
    print(f"queue {q}: {len(rs)} kernels, busy {busy / 1e6:.3f} ms, span {span / 1e6:.3f} ms, idle gaps {sum(pos) / 1e6:.3f} ms "
          f"(median {sorted(pos)[len(pos) // 2] / 1e3 if pos else 0:.2f} us, >5us: {sum(1 for g in pos if g > 5000)})")
    big = sorted(((g, a["Kernel_Name"][:50], b["Kernel_Name"][:50]) for g, a, b in zip(gaps, rs, rs[1:])), reverse=True)[:8]
    for g, a, b in big:
        print(f"    {g / 1e3:7.1f} us between {a.replace('(anonymous namespace)::', '')} -> {b.replace('(anonymous namespace)::', '')}")
# tail: what each queue does in the last 600 us of the step
for q, rs in byq.items():
    print(f"queue {q} tail:")
    for r in rs:
        if int(r["End_Timestamp"]) > t1 - 600000:
            print(f"   {(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} .. {(int(r['End_Timestamp']) - t0) / 1e3:9.1f} us  {r['Kernel_Name'].replace('(anonymous namespace)::', '')[:60]}")
