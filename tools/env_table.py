#!/usr/bin/env python3
"""Rewrites the knob table of INTEGRATION.md ("## Environment") from deephisto_amd/csrc/env_knobs.h.  Tooling only (tests/test_abi.py
compares the two through dh_debug_env_knobs, so a stale table fails the CPU suite)."""
import re
from pathlib import Path
R = Path(__file__).resolve().parents[1]
src = (R / "deephisto_amd/csrc/env_knobs.h").read_text()
rows = re.findall(r'X\((DH_[A-Z0-9_]+), (.+?), (.+?), (.+?), "(\w+)", "(.*?)"\)', src)
lines = ["| variable | default | range | read | effect |", "|---|---|---|---|---|"]
for n, d, lo, hi, rd, eff in rows:
    lines.append(f"| `{n}` | {eval(d)} | {eval(lo)} .. {eval(hi)} | {rd} | {eff} |")
doc = (R / "INTEGRATION.md").read_text()
new = re.sub(r"\| variable \| default \| range \| read \| effect \|\n(\|.*\n)+", "\n".join(lines) + "\n", doc)
assert new != doc or all(l in doc for l in lines)
(R / "INTEGRATION.md").write_text(new)
print(f"{len(rows)} knobs")
