#!/usr/bin/env python3
"""Per-parameter relative gradient error of the HIP backward and of the f32 CPU oracle, both against a
float64 CPU run of the same model, for several (B, P).  Tells conditioning (ReLU-mask flips, BN
cancellation at tiny batch) from kernel bugs.  Tooling only."""
import copy
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import torch.nn.functional as F
from oracle import resnet18 as oracle_net
from deephisto_amd.models.patch_cls_simple.model import get_model

dev = torch.device("cuda:0")
cases = [(3, 224), (8, 224), (8, 160), (4, 256), (4, 128), (8, 64), (2, 224), (16, 224)]
if len(sys.argv) > 1:
    cases = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for B, P in cases:
    ref = oracle_net.seeded_model(11, 5, perturb_bn=True).train()
    ref64 = copy.deepcopy(ref).double().train()
    m = get_model(5, "f32")
    m.load_state_dict(ref.state_dict())
    m = m.to(dev).train()
    g = torch.Generator().manual_seed(B * P)
    x = torch.rand(B, 3, P, P, generator=g)
    y = torch.randint(0, 5, (B,), generator=g)
    F.cross_entropy(ref(x), y).backward()
    F.cross_entropy(ref64(x.double()), y).backward()
    F.cross_entropy(m(x.to(dev)), y.to(dev)).backward()
    g64 = {k: p.grad for k, p in ref64.named_parameters()}
    g32 = {k: p.grad for k, p in ref.named_parameters()}
    rows = []
    for k, p in m.named_parameters():
        den = float(g64[k].abs().max()) + 1e-300
        e_hip = float((p.grad.cpu().double() - g64[k]).abs().max()) / den
        e_ref = float((g32[k].double() - g64[k]).abs().max()) / den
        nrm = float(g64[k].norm()) + 1e-300
        l_hip = float((p.grad.cpu().double() - g64[k]).norm()) / nrm
        l_ref = float((g32[k].double() - g64[k]).norm()) / nrm
        rows.append((k, e_hip, e_ref, l_hip, l_ref))
    print(f"   relative L2: worst hip {max(r[3] for r in rows):.2e} ({max(rows, key=lambda r: r[3])[0]})  worst cpu-f32 {max(r[4] for r in rows):.2e}")
    bad = [r[:3] for r in rows if r[1] > 2e-3 or r[2] > 2e-3]
    print(f"B={B} P={P}: worst hip {max(r[1] for r in rows):.2e}  worst cpu-f32 {max(r[2] for r in rows):.2e}; "
          f"head-first offenders (name hip cpu32): " + " ".join(f"{k}:{a:.1e}/{b:.1e}" for k, a, b in bad[::-1][:6]), flush=True)
