"""Launched by tests/test_gpu_resnet.py::test_wide_stride2_kernel_* in a fresh process (the library reads DH_CONV_S2_WIDE once):
bf16 logits of the fused inference path on 300 random tiles of a closed-form slide -> argv[1] (.npy); argv[2] = patch size."""
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from deephisto_amd import tiles  # noqa: E402
from deephisto_amd.models.patch_cls_simple.model import get_model  # noqa: E402
from oracle import resnet18 as oracle_net  # noqa: E402

P = int(sys.argv[2])
dev = torch.device("cuda:0")
oracle = oracle_net.seeded_model(31, 5, perturb_bn=True).eval()
m = get_model(5, "bf16")
m.load_state_dict(oracle.state_dict())
m.to(dev).eval()
slide = tiles.synth_slide(4096, 4096, 3, dev)
rng = np.random.default_rng(1)
n = 300
o = np.stack([rng.integers(0, 4096 - P, n), rng.integers(0, 4096 - P, n)], 1).astype(np.int32)
runs = [m.forward_tiles(slide, torch.from_numpy(o).to(dev), P).cpu().numpy() for _ in range(3)]
assert all(np.array_equal(runs[0], r) for r in runs[1:]), "the same launch gave different logits (a race)"
np.save(sys.argv[1], runs[0])
