#!/usr/bin/env python3
"""Compile one .hip for gfx950 with -Rpass-analysis=kernel-resource-usage and print one line per kernel.
Tooling only.  Usage: kernel_resources.py deephisto_amd/csrc/resnet_kernels.hip [name-filter]"""
import re
import subprocess
import sys
import tempfile

src, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
with tempfile.TemporaryDirectory() as d:
    r = subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-c", src, "-o", d + "/x.o",
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
cur = {}
for line in r.stderr.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
    elif ":" in t:
        k, v = t.rsplit(":", 1)
        cur[k.strip()] = v.strip()
        if k.strip().startswith("LDS Size") and flt in cur["name"]:
            n = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
            n = re.sub(r"\(anonymous namespace\)::", "", n)[:90]
            print(f"{n:90s} vgpr={cur.get('VGPRs')} agpr={cur.get('AGPRs')} sgpr_spill={cur.get('SGPRs Spill')} "
                  f"vgpr_spill={cur.get('VGPRs Spill')} occ={cur.get('Occupancy [waves/SIMD]')} scratch={cur.get('ScratchSize [bytes/lane]')}")
