#!/usr/bin/env python3
"""Compare the intermediate gradients dumped by DH_TRAIN_DUMP with the CPU oracle's (backward hooks). Tooling only.
Needs a DIAGNOSTIC build of the library (tools/build_variant.sh dump -DDH_TRAIN_DUMP_BUILD=1, copied over libdeephisto_hip.so):
the shipped library compiles the dump to nothing.
usage: grad_trace.py B P"""
import os
import sys
import tempfile
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
d = tempfile.mkdtemp()
os.environ["DH_TRAIN_DUMP"] = d
import numpy as np
import torch
import torch.nn.functional as F
from oracle import resnet18 as oracle_net
from deephisto_amd.models.patch_cls_simple.model import get_model

B, P = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
ref = oracle_net.seeded_model(11, 5, perturb_bn=True).train()
m = get_model(5, "f32")
m.load_state_dict(ref.state_dict())
m = m.to(dev).train()
g = torch.Generator().manual_seed(B * P)
x = torch.rand(B, 3, P, P, generator=g)
y = torch.randint(0, 5, (B,), generator=g)
grads = {}
blocks = [getattr(ref, f"layer{i}")[j] for i in range(1, 5) for j in range(2)]
for bi, blk in enumerate(blocks):
    for nm, mod in (("dZ1", blk.conv1), ("dZ2", blk.conv2), ("dY1_from_conv2_input", blk.conv2)):
        def hook(mod, gin, gout, key=(bi, nm)):
            grads[key] = (gin[0] if "input" in key[1] else gout[0]).detach()
        mod.register_full_backward_hook(hook)
F.cross_entropy(ref(x), y).backward()
F.cross_entropy(m(x.to(dev)), y.to(dev)).backward()
torch.cuda.synchronize()
for bi in range(7, -1, -1):
    for nm, key in (("dZ2", (bi, "dZ2")), ("dY1", (bi, "dY1_from_conv2_input")), ("dZ1", (bi, "dZ1"))):
        want = grads[key].permute(0, 2, 3, 1).contiguous().numpy()
        got = np.fromfile(f"{d}/b{bi}_{nm}.bin", dtype=np.float32).reshape(want.shape)
        diff = np.abs(got - want)
        per_c = diff.reshape(-1, want.shape[-1]).max(0)
        mean_c = (got - want).reshape(-1, want.shape[-1]).mean(0)
        print(f"block {bi} {nm}: max err {diff.max():.3e} / scale {np.abs(want).max():.3e}; channels with err>1e-2*scale: "
              f"{int((per_c > 1e-2 * np.abs(want).max()).sum())}; max |mean offset per channel| {np.abs(mean_c).max():.3e}", flush=True)

# detail: elements of dZ1 that differ, with the oracle's pre-ReLU activation there (a ReLU-mask flip shows as |pre| ~ 1e-7)
pre = {}
for bi, blk in enumerate(blocks):
    blk.bn1.register_forward_hook(lambda mod, inp, out, bi=bi: pre.__setitem__(bi, out.detach()))
ref.zero_grad()
with torch.no_grad():
    ref(x)   # note: updates running stats again; irrelevant here
for bi in range(7, -1, -1):
    want = grads[(bi, "dZ1")].permute(0, 2, 3, 1).contiguous().numpy()
    got = np.fromfile(f"{d}/b{bi}_dZ1.bin", dtype=np.float32).reshape(want.shape)
    diff = np.abs(got - want)
    idx = np.argwhere(diff > 0.02 * np.abs(want).max())
    if len(idx):
        p = pre[bi].permute(0, 2, 3, 1).numpy()
        print(f"block {bi}: {len(idx)} dZ1 elements off by > 2% of scale; first: " +
              "; ".join(f"{tuple(int(v) for v in i)} pre-relu={p[tuple(i)]:+.3e} got={got[tuple(i)]:+.3e} want={want[tuple(i)]:+.3e}" for i in idx[:5]))
        break
