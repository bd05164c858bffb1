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
    # group the launches of one forward (stem ... avgpool+fc); short (last, partial) micro-batches are recognised by
    # their stem time and left out of the per-layer averages
    forwards, other = [], collections.defaultdict(list)
    for r in rows:
        n = r["Kernel_Name"]
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if "stem_pool" in n:
            forwards.append([("stem+pool", us)])
        elif "conv3x3_kernel" in n and forwards and len(forwards[-1]) <= len(SEQ):
            forwards[-1].append((SEQ[len(forwards[-1]) - 1], us))
        else:
            lab = next((v for k, v in (("avgpool", "avgpool+fc"), ("accumulate", "accumulate"), ("argmax", "argmax"),
                                       ("gather", "gather"), ("synth", "synth")) if k in n), None)
            if lab:
                other[lab].append(us)
    stems = sorted(fw[0][1] for fw in forwards)
    median = stems[len(stems) // 2] if stems else 0.0
    agg = collections.defaultdict(list)
    for fw in forwards:
        if fw[0][1] >= 0.8 * median:
            for lab, us in fw:
                agg[lab].append(us)
    agg.update(other)
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
