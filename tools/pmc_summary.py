#!/usr/bin/env python3
"""Fold two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into profiles/*_pmc_dominant_kernel.json.

Tooling, not product.  Usage: pmc_summary.py <fetch_dir> <write_dir> <micro_batch> <out.json> [tiles_per_launch]
The dominant kernel is conv3x3_kernel<bf16, stride 1, NT=2, 8 waves> (every stride-1 3x3 conv, layers 1-4, of ResNet-18 at
256x256 tiles).  gfx950 corrections follow MI355X_MICROARCH.md (HBM / rocprofv3 section): counters
are in KiB; FETCH_SIZE under-counts 16-B-per-lane streaming reads by 2x; WRITE_SIZE is taken as is.
"""
import csv
import glob
import json
import sys


def avg_counter(path, name):
    f = glob.glob(path + "/**/*counter_collection.csv", recursive=True)[0]
    vals = []
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if r["Counter_Name"] != name or "conv3x3_kernel" not in k:
            continue
        # template args: <T, STRIDE, NT, WAVES, STAMP, DS, MT, WRES, CLS, HALF>  (round 4: HALF appended; the wide stride-2 variant's
        # name comes out of rocprofv3 mangled and is skipped here: it is not the dominant kernel)
        if "<" not in k:
            continue
        args = k[k.index("<") + 1:k.rindex(">")].replace(" ", "").split(",")
        # rocprofv3 garbles the first two (type, stride) in its demangling; NT=2 exists for stride 1 only and the
        # bench runs bf16 only, so <..., NT=2, WAVES=8, STAMP=false, DS=false, MT=2, WRES, CLS=-1, HALF=false> identifies the variant
        # (the WRES flag distinguishes the layer-1 instantiation: both belong to the variant)
        if args[-8:-3] == ["2", "8", "false", "false", "2"] and args[-2] == "-1" and args[-1] == "false":
            vals.append(float(r["Counter_Value"]))
    return sum(vals) / len(vals), len(vals)


def algorithmic_bytes(mb):
    """input + output (+ residual) of the 13 launches of this variant per forward, bf16, no halo (round 5: layer 4's 8 x 8 maps run on
    this variant too -- eight whole images per 512-slot tile -- so it covers every stride-1 3x3 conv of the network)."""
    tot = 0
    for hw, c, n, n_res in ((64, 64, 4, 2), (32, 128, 3, 2), (16, 256, 3, 2), (8, 512, 3, 2)):
        act = mb * hw * hw * c * 2
        tot += n * 2 * act + n_res * act + n * 9 * c * c * 2
    return tot / 13


def main(fetch_dir, write_dir, mb, out, tiles=None):
    tiles = int(tiles) if tiles else int(mb)   # tiles actually in one launch (38 416 tiles at micro-batch 4096: 10 launches of 3 842)
    fetch, n = avg_counter(fetch_dir, "FETCH_SIZE")
    write, _ = avg_counter(write_dir, "WRITE_SIZE")
    doc = {
        "kernel": "conv3x3_kernel<bf16, stride 1, NT=2, 8 waves>",
        "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, in a separate pass, --pmc WRITE_SIZE) -- "
                   "python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --train-steps 0",
        "micro_batch": int(mb), "tiles_per_launch": tiles, "launches": n,
        "FETCH_SIZE_avg_KB": fetch, "WRITE_SIZE_avg_KB": write,
        "correction": "gfx950: FETCH_SIZE counts 1/2 of the bytes of 16-B-per-lane streaming reads "
                      "(MI355X_MICROARCH.md, HBM) -> x2; WRITE_SIZE taken as is; units KiB",
        "traffic_bytes_per_launch": (2 * fetch + write) * 1024,
        "algorithmic_bytes_per_launch": algorithmic_bytes(tiles),
        "note": "algorithmic = input + output (+ residual on 8 of 13) + weights, no halo, averaged over the 13 "
                "launches per forward of this variant (layer1 x4, layer2 x3, layer3 x3, layer4 x3)",
    }
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:6])
