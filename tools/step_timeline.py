#!/usr/bin/env python3
"""Every kernel of ONE training step in launch order, per queue: start offset, duration, gap to the predecessor on its queue, grid size.
Input: a rocprofv3 --kernel-trace of tools/train_profile.py.  usage: step_timeline.py <dir or kernel_trace.csv> [out.txt].  Tooling only.
Ends with a per-kernel-name summary per queue (launches, total us) -- i.e. what the critical (main) queue is made of."""
import csv
import sys
from collections import defaultdict
from pathlib import Path

p = Path(sys.argv[1])
f = p if p.is_file() else next(p.rglob("*kernel_trace.csv"))
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "stem_kernel" in r["Kernel_Name"]]
lo, hi = starts[-2], starts[-1]
step = rows[lo:hi]
t0 = int(step[0]["Start_Timestamp"])
t1 = max(int(r["End_Timestamp"]) for r in step)
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "")[:70]
byq = defaultdict(list)
for r in step:
    byq[r["Queue_Id"]].append(r)
print(f"step: {len(step)} kernels, wall {(t1 - t0) / 1e3:.1f} us, queues {[(q, len(v)) for q, v in byq.items()]}", file=out)
for q, rs in byq.items():
    print(f"\n== queue {q}", file=out)
    prev_end = None
    for i, r in enumerate(rs):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        prev_end = max(e, prev_end or e)
        grid = int(r.get("Grid_Size_X", r.get("Grid_Size", "0")) or 0) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", "1")) or 1))
        print(f"{i:4d} {(s - t0) / 1e3:9.1f} dur {(e - s) / 1e3:7.1f} gap {gap:6.1f} wgs {grid:6d} lds {r.get('LDS_Block_Size', '?'):>6s} {short(r['Kernel_Name'])}", file=out)
    agg = defaultdict(lambda: [0, 0.0])
    for r in rs:
        a = agg[short(r["Kernel_Name"])]
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    busy = sum(v[1] for v in agg.values())
    print(f"-- queue {q}: busy {busy:.1f} us of {(t1 - t0) / 1e3:.1f}", file=out)
    for n, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"   {c:4d} x {us / c:7.1f} = {us:8.1f} us  {n}", file=out)
