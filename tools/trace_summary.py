#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV per (kernel, grid) -- tooling, not product."""
import collections
import csv
import glob
import sys


def short(n):
    if 'conv3x3_kernel' in n:
        return 'conv3x3_s1' if ('Li1E' in n or ', 1>' in n) else 'conv3x3_s2'
    for k, v in (('conv_kernel', 'conv1x1'), ('stem', 'stem'), ('maxpool', 'maxpool'), ('avgpool', 'avgfc'),
                 ('accumulate', 'accumulate'), ('argmax', 'argmax'), ('gather', 'gather'), ('synth', 'synth')):
        if k in n:
            return v
    return None


def main(path):
    f = glob.glob(path + '/**/*kernel_trace.csv', recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        s = short(r['Kernel_Name'])
        if s:
            agg[(s, int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), int(r['LDS_Block_Size']),
                 int(r['VGPR_Count']))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    tot = 0.0
    for k, v in sorted(agg.items()):
        tot += sum(v)
    for k, v in sorted(agg.items()):
        print(f"{k[0]:12s} blocks={k[1]:6d} lds={k[2]:6d} vgpr={k[3]:4d} calls={len(v):5d} avg_us={sum(v)/len(v):8.1f} "
              f"min_us={min(v):8.1f} share={100*sum(v)/tot:5.1f}%")
    print(f"total kernel time {tot/1e3:.2f} ms")


if __name__ == '__main__':
    main(sys.argv[1])
