#!/usr/bin/env python3
"""Diagnostic: error magnitudes of the bf16 engine (dh_train2) against the float32 CPU oracle, per conv and per gradient tensor.
Tooling only.  usage: t2_check.py [arch] [B] [P]"""
import ctypes as C
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import torch.nn.functional as F
from deephisto_amd._lib import check, lib
from deephisto_amd.models.patch_cls_simple.model import get_model

arch = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
P = int(sys.argv[3]) if len(sys.argv) > 3 else 64
if arch == "resnet50":
    from oracle import resnet50 as onet
else:
    from oracle import resnet18 as onet
dev = torch.device("cuda:0")
ref = onet.seeded_model(3, 5, perturb_bn=True).train()
GAIN = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0   # gain of every block's last BN (residual branch): < 1 = better conditioned
if GAIN != 1.0:
    with torch.no_grad():
        for name, mod in ref.named_modules():
            if name.endswith("bn3" if arch == "resnet50" else "bn2"):
                mod.weight.mul_(GAIN)
m = get_model(5, "bf16", arch=arch)
m.load_state_dict(ref.state_dict())
m.to(dev).train()
g = torch.Generator().manual_seed(1)
x = torch.rand(B, 3, P, P, generator=g)
y = torch.randint(0, 5, (B,), generator=g)
acts = {}
def hook(name):
    def f(mod, inp, out):
        acts[name] = out.detach()
    return f
for name, mod in ref.named_modules():
    if isinstance(mod, torch.nn.Conv2d):
        mod.register_forward_hook(hook(name))
import copy
from oracle.bf16_emulation import forward_bf16
emu = copy.deepcopy(ref)
acts_e = {}
out_emu = forward_bf16(emu, x, acts_e)
loss_emu = F.cross_entropy(out_emu, y)
loss_emu.backward()
out_ref = ref(x)
loss_ref = F.cross_entropy(out_ref, y)
loss_ref.backward()
out = m(x.to(dev))
loss = F.cross_entropy(out, y.to(dev))
loss.backward()
torch.cuda.synchronize()
# third oracle run: bf16-emulated with the ENGINE's ReLU patterns imposed
eng = m._engine if arch == "resnet50" else m._engine2
masks = {}
for name, a in acts.items():
    if "downsample" in name:
        continue
    n = a.numel()
    buf = torch.empty(n, dtype=torch.float32, device=dev)
    check(lib().dh_train2_debug_act(eng.handle, name.encode(), 1, buf.data_ptr(), n, None), "dbg")
    masks[name] = (buf.cpu().reshape(a.shape[0], a.shape[2], a.shape[3], a.shape[1]).permute(0, 3, 1, 2) > 0)
emm = copy.deepcopy(ref)
emm.zero_grad()
out_m = forward_bf16(emm, x, None, masks)
F.cross_entropy(out_m, y).backward()
gm = {k: p.grad for k, p in emm.named_parameters()}
print("max |dlogit| vs masked-emulated", float((out.detach().cpu() - out_m.detach()).abs().max()))
print("logits ref", out_ref.detach()[0].tolist())
print("logits hip", out.detach().cpu()[0].tolist())
print("logits emu", out_emu.detach()[0].tolist())
print("max |dlogit| vs emulated", float((out.detach().cpu() - out_emu.detach()).abs().max()), "loss emu", float(loss_emu))
print("max |dlogit|", float((out.detach().cpu() - out_ref.detach()).abs().max()), "scale", float(out_ref.abs().max()),
      "loss", float(loss), float(loss_ref))
for name, a in acts.items():
    n = a.numel()
    buf = torch.empty(n, dtype=torch.float32, device=dev)
    check(lib().dh_train2_debug_act(eng.handle, name.encode(), 0, buf.data_ptr(), n, None), "dbg")
    got = buf.cpu().reshape(a.shape[0], a.shape[2], a.shape[3], a.shape[1]).permute(0, 3, 1, 2)
    e = float((got - a).norm() / (a.norm() + 1e-30))
    e2 = float((got - acts_e[name]).norm() / (acts_e[name].norm() + 1e-30))
    e3 = float((acts_e[name] - a).norm() / (a.norm() + 1e-30))
    print(f"Z {name:28s} rel-L2 vs f32 {e:.3e}  vs bf16-emulated {e2:.3e}   (emulated vs f32 {e3:.3e})")
rg = {k: p.grad for k, p in ref.named_parameters()}
for k, p in m.named_parameters():
    e = float((p.grad.cpu() - rg[k]).norm() / (rg[k].norm() + 1e-30))
    ge = dict(emu.named_parameters())[k].grad
    e2 = float((p.grad.cpu() - ge).norm() / (ge.norm() + 1e-30))
    e3 = float((p.grad.cpu() - gm[k]).norm() / (gm[k].norm() + 1e-30))
    print(f"grad {k:36s} rel-L2 vs f32 {e:.3e}  vs emulated {e2:.3e}  vs emulated+engine masks {e3:.3e}  |g| {float(rg[k].norm()):.3e}")
