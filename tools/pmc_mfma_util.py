#!/usr/bin/env python3
"""MFMA utilisation per layer from one rocprofv3 --pmc pass (SQ_BUSY_CU_CYCLES, SQ_VALU_MFMA_BUSY_CYCLES, SQ_LDS_IDX_ACTIVE,
SQ_LDS_BANK_CONFLICT) of tools/fwd_once.py.  Tooling only.  usage: pmc_mfma_util.py <dir> <out.json>
util = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES) per dispatch: both counters are in shader cycles summed over the chip,
the MFMA counter over the 4 SIMDs (= 4 matrix pipes) of every CU, the CU counter once per CU (cross-check: a 77.3 GFLOP launch is
2.36 M v_mfma_f32_32x32x16_bf16 x 32 cycles = 75.5 M pipe-cycles against 256 CUs x 67 us x 1.9 GHz = 32.6 M CU-cycles -> ratio 2.3 = 4 x 0.58;
the in-kernel phase stamps give the same share); averaged per layer class;
layers are labelled by launch order inside a forward (trace_summary.SEQ)."""
import collections
import csv
import glob
import json
import sys
sys.path.insert(0, __file__.rsplit("/", 1)[0])
from trace_summary import SEQ

path, out = sys.argv[1], sys.argv[2]
f = glob.glob(path + "/**/*counter_collection.csv", recursive=True)[0]
by_disp = collections.OrderedDict()
for r in csv.DictReader(open(f)):
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
doc = {"command": "rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -- "
                  "python3 tools/fwd_once.py 256 3   (bf16 forwards of 256 tiles of 256x256)",
       "definition": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES); lds_active = SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES "
                     "(ROCm 7.2 ships no gfx950 derived-metric section; raw counters, averaged over the launches of a layer class)",
       "layers": {}}
for lab in ["stem+pool"] + sorted(set(SEQ), key=SEQ.index):
    if lab not in agg:
        continue
    a = {k: sum(v) / len(v) for k, v in agg[lab].items()}
    busy = a.get("SQ_BUSY_CU_CYCLES", 0.0)
    doc["layers"][lab] = {"launches": len(agg[lab]["SQ_BUSY_CU_CYCLES"]), **{k: round(v) for k, v in a.items()},
                          "mfma_util": round(a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * busy), 4) if busy else None,
                          "lds_active": round(a.get("SQ_LDS_IDX_ACTIVE", 0.0) / busy, 4) if busy else None}
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps(doc["layers"], indent=1))
