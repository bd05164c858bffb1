#!/usr/bin/env python3
"""Static instruction histogram of one kernel of resnet_kernels.hip (hipcc --save-temps in /tmp/isa). Tooling only.
   python3 tools/isa_stats.py <substring of the mangled kernel name> [more substrings...]"""
import collections, re, subprocess, sys
from pathlib import Path
root = Path(__file__).resolve().parents[1]
out = Path("/tmp/isa"); out.mkdir(exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "--save-temps", "-c",
                str(root / "deephisto_amd/csrc/resnet_kernels.hip"), "-o", "rk.o", f"-I{root}/include"], cwd=out, check=True,
               stderr=subprocess.DEVNULL)
s = (out / "resnet_kernels-hip-amdgcn-amd-amdhsa-gfx950.s").read_text()
def cls(op):
    for pre, c in (("v_mfma", "mfma"), ("v_", "valu"), ("s_waitcnt", "wait"), ("s_barrier", "barrier"), ("s_cbranch", "branch"),
                   ("s_branch", "branch"), ("s_", "salu"), ("ds_", "lds"), ("global_", "vmem"), ("buffer_", "vmem"), ("scratch_", "SCRATCH")):
        if op.startswith(pre): return c
    return op
for m in re.finditer(r'^(_Z\S+):', s, re.M):
    name = m.group(1)
    if not all(k in name for k in sys.argv[1:]): continue
    body = s[m.end():s.index(".end_amdhsa_kernel", m.end())]
    ops = [l.split()[0] for l in (x.strip() for x in body.split("\n")) if l and not l.startswith((".", ";", "//")) and not l.endswith(":")]
    c = collections.Counter(cls(o) for o in ops)
    meta = {k: re.search(rf'\.amdhsa_{k}\s+(\S+)', body) for k in ("next_free_vgpr", "next_free_sgpr", "accum_offset")}
    sp = re.search(r'; ScratchSize: (\d+)', s[m.end():m.end() + len(body) + 4000])
    print(name[:100], len(ops), dict(c), {k: v.group(1) for k, v in meta.items() if v}, "scratch", sp.group(1) if sp else "?")
