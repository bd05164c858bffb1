"""The product package must never import or call the oracle (CPU restatement)."""
import re
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]


def test_product_does_not_reference_oracle():
    bad = []
    for f in list((REPO / "deephisto_amd").rglob("*.py")) + list((REPO / "deephisto_amd").rglob("*.hip")) \
            + list((REPO / "deephisto_amd").rglob("*.h")):
        txt = f.read_text()
        if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "oracle/" in txt and f.suffix == ".py":
            bad.append(str(f))
    assert not bad, bad


def test_reference_tree_not_read_at_runtime():
    for f in list((REPO / "deephisto_amd").rglob("*.py")) + [REPO / "bench.py", REPO / "__graft_entry__.py"]:
        if f.exists():
            assert "/root/reference" not in f.read_text().replace("oracle/_ref", ""), f
