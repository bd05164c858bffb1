#!/usr/bin/env python3
"""Fabric-side traffic of ONE training step by kernel, from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel trace only)
of tools/train_profile.py.  Tooling only.  usage: pmc_train_traffic.py <fetch_dir> <write_dir> <arch> [out.txt]
Counters are KiB; on gfx950 FETCH_SIZE reports half of the bytes of 16-byte-per-lane streaming reads (MI355X_MICROARCH.md, HBM) -> x 2 (the
engines read in 16-byte pieces almost everywhere; other widths are uncalibrated, so the per-kernel rows are indicative and the total is good to
a few percent); Infinity-Cache hits are counted (this is L2 <-> fabric traffic, an upper bound of the HBM bytes).  The step is the LAST of the
six the profiled program runs (dispatches between the last two stem launches, as tools/step_timeline.py cuts it)."""
import csv
import glob
import sys
from collections import defaultdict
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def per_kernel(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    starts = [i for i, r in enumerate(rows) if "stem_kernel" in r["Kernel_Name"]]
    step = rows[starts[-2]:starts[-1]]
    agg = defaultdict(lambda: [0, 0.0])
    for r in step:
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        n = n.split("(")[0][:56]
        agg[n][0] += 1
        agg[n][1] += float(r["Counter_Value"]) * 1024
    return agg


def main(fd, wd, arch, out=None):
    rd, wr = per_kernel(fd, "FETCH_SIZE"), per_kernel(wd, "WRITE_SIZE")
    o = open(out, "w") if out else sys.stdout
    names = sorted(set(rd) | set(wr), key=lambda n: -(2 * rd.get(n, [0, 0])[1] + wr.get(n, [0, 0])[1]))
    tr = sum(2 * v[1] for v in rd.values())
    tw = sum(v[1] for v in wr.values())
    print(f"{arch}: one training step (64 x 224^2), L2 <-> fabric traffic by kernel: read = 2 x FETCH_SIZE, written = WRITE_SIZE (MB)", file=o)
    print(f"{'kernel':58s} {'launches':>8s} {'read':>9s} {'written':>9s} {'share':>6s}", file=o)
    for n in names:
        c = max(rd.get(n, [0, 0])[0], wr.get(n, [0, 0])[0])
        r_, w_ = 2 * rd.get(n, [0, 0])[1], wr.get(n, [0, 0])[1]
        print(f"{n:58s} {c:8d} {r_ / 1e6:9.1f} {w_ / 1e6:9.1f} {100 * (r_ + w_) / (tr + tw):5.1f}%", file=o)
    print(f"{'total':58s} {'':8s} {tr / 1e6:9.1f} {tw / 1e6:9.1f}   = {(tr + tw) / 1e9:.2f} GB per step", file=o)
    try:
        import bench
        a, dt = ("resnet18", "f32") if arch == "resnet18" else (arch.replace("bf16", ""), "bf16")
        hb = bench.train_hbm_bytes(a, dt, 64, 224)
        print(f"algorithmic bytes of the engine's passes (bench.py train_hbm_bytes): {hb['total'] / 1e9:.2f} GB  ->  traffic / algorithmic = "
              f"{(tr + tw) / hb['total']:.2f}", file=o)
    except Exception as e:   # the table stands without the comparison
        print(f"(no algorithmic comparison: {e})", file=o)


if __name__ == "__main__":
    main(*sys.argv[1:5])
