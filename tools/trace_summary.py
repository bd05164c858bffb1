#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV of the inference path per kernel and per ResNet layer.

Tooling, not product.  conv3x3 launches are labelled by their position after each stem launch (the
forward's launch order is fixed: stem, 4x layer1, [s2+ds, 3x s1] for layers 2-4, avgpool+fc), because
rocprofv3's demangling garbles the leading template arguments of the kernel name.
"""
import collections
import csv
import glob
import sys

SEQ = ["L1 s1 c64"] * 4 + ["L2 s2+ds"] + ["L2 s1 c128"] * 3 + ["L3 s2+ds"] + ["L3 s1 c256"] * 3 + \
      ["L4 s2+ds"] + ["L4 s1 c512"] * 3
GFLOP = {"L1 s1 c64": 2 * 64 * 64 * 64 * 64 * 9, "L2 s1 c128": 2 * 32 * 32 * 128 * 128 * 9,
         "L3 s1 c256": 2 * 16 * 16 * 256 * 256 * 9, "L4 s1 c512": 2 * 8 * 8 * 512 * 512 * 9,
         "L2 s2+ds": 2 * 32 * 32 * 128 * 64 * 10, "L3 s2+ds": 2 * 16 * 16 * 256 * 128 * 10,
         "L4 s2+ds": 2 * 8 * 8 * 512 * 256 * 10, "stem+pool": 2 * 128 * 128 * 64 * 147}


def main(path, micro_batch=256):
    f = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    agg = collections.defaultdict(list)
    pos = None
    for r in rows:
        n = r["Kernel_Name"]
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        full = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]) >= 256   # persistent grid filled = full micro-batch
        if "stem_pool" in n:
            pos = 0
            lab = "stem+pool"
        elif "conv3x3_kernel" in n and pos is not None and pos < len(SEQ):
            lab = SEQ[pos]
            pos += 1
        else:
            lab = next((v for k, v in (("avgpool", "avgpool+fc"), ("accumulate", "accumulate"), ("argmax", "argmax"),
                                       ("gather", "gather"), ("synth", "synth")) if k in n), None)
            full = True
        if lab and full:
            agg[lab].append(us)
    tot = sum(sum(v) for v in agg.values())
    print(f"{'kernel':12s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'share':>7s} {'TFLOP/s':>8s}   (full micro-batches of {micro_batch} only)")
    for k in ["stem+pool"] + sorted(set(SEQ), key=SEQ.index) + ["avgpool+fc", "accumulate", "argmax", "synth"]:
        v = agg.get(k)
        if not v:
            continue
        a = sum(v) / len(v)
        tf = f"{GFLOP[k] * int(micro_batch) / a / 1e6:8.0f}" if k in GFLOP else " " * 8
        print(f"{k:12s} {len(v):6d} {a:9.1f} {min(v):9.1f} {100 * sum(v) / tot:6.1f}% {tf}")
    print(f"total kernel time {tot / 1e3:.2f} ms")


if __name__ == "__main__":
    main(*sys.argv[1:3])
