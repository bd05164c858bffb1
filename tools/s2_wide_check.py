#!/usr/bin/env python3
"""bf16 logits of the fused inference path on random tiles; run with DH_CONV_S2_WIDE=0 and =1 and compare the saved files. Tooling only.
usage: s2_wide_check.py out.npy [P]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
from deephisto_amd import tiles
from deephisto_amd.models.patch_cls_simple.model import get_model
from oracle import resnet18 as oracle_net
P = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
oracle = oracle_net.seeded_model(31, 5, perturb_bn=True).eval()
m = get_model(5, "bf16"); m.load_state_dict(oracle.state_dict()); m.to(dev).eval()
slide = tiles.synth_slide(4096, 4096, 3, dev)
rng = np.random.default_rng(1)
n = 300
o = np.stack([rng.integers(0, 4096 - P, n), rng.integers(0, 4096 - P, n)], 1).astype(np.int32)
lg = m.forward_tiles(slide, torch.from_numpy(o).to(dev), P).cpu().numpy()
np.save(sys.argv[1], lg)
host = slide.cpu().numpy()
from oracle import tiling
with torch.no_grad():
    want = oracle(torch.from_numpy(tiling.features_nchw_predictor(host, o[:24], P))).numpy()
print("max |logit - f32 oracle| over 24 tiles:", float(np.abs(lg[:24] - want).max()), " scale", float(np.abs(want).max()))
