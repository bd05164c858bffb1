"""Launched by tests/test_cli_distributed.py through `python -m torch.distributed.run --nproc-per-node 2` on CPU (gloo).

Runs the REAL `models.patch_cls_simple.train.main()` wiring (process-group setup from the launcher's environment, per-rank
sampler seed, rank-0 checkpoint, metric averaging, destroy at exit) with a stub model / sampler on CPU tensors, since the
HIP model needs a GPU.  Each rank writes a JSON report next to the checkpoint."""
import json
import os
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn as nn  # noqa: E402

from deephisto_amd.models.patch_cls_simple import train as T  # noqa: E402
from deephisto_amd.models.patch_cls_simple import utils  # noqa: E402

out_dir = Path(sys.argv[1])
report = {"rank_env": int(os.environ["RANK"]), "initialized_before_main": dist.is_initialized()}


class StubModel(nn.Module):
    """A linear classifier whose `train_step` averages its gradients over the ranks like the HIP models' does."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(77 + int(os.environ["RANK"]))   # every rank starts from its OWN weights, like the HIP models' unseeded init:
        self.fc = nn.Linear(12, 5)                         # train() must broadcast rank 0's before the first step

    def forward(self, x):
        return self.fc(x.flatten(1))

    def train_step(self, x, labels, lr=1e-4, group=None):
        report["group_up_in_train_step"] = dist.is_initialized() and dist.get_world_size() == 2
        logits = self(x)
        loss = nn.functional.cross_entropy(logits, labels)
        self.zero_grad()
        loss.backward()
        with torch.no_grad():
            for p in self.parameters():
                dist.all_reduce(p.grad)
                p -= lr * p.grad / dist.get_world_size()
        return loss.detach(), logits.detach()


class StubSampler:
    def __init__(self, seed):
        self.g = torch.Generator().manual_seed(1000 + seed)
        report["sampler_seed"] = seed

    def device_batches(self, bs, n, flips=True):
        for k in range(n):
            if os.environ.get("DH_STUB_FAIL_RANK") == os.environ["RANK"] and k == 1:
                raise RuntimeError("stub sampler failure on one rank")   # its peer is inside the next step's all-reduce
            yield torch.rand(bs, 3, 2, 2, generator=self.g), torch.randint(0, 5, (bs,), generator=self.g), None


utils.get_device = lambda: torch.device("cpu")
T.ce_loss = lambda logits, labels: nn.functional.cross_entropy(logits, labels)   # the validation loss kernel is GPU-only
T._synthetic_sampler = lambda cfg, device: StubSampler(T._rank_world()[0])
_real_train = T.train
model = StubModel()
report["initial_weights"] = model.fc.weight.detach().flatten().tolist()
T.train = lambda cfg, **kw: _real_train(cfg, model=model, **kw)

cfg = {"dataset": {"folder": "/nonexistent", "layer": 1, "patch_size": 2, "patches_from_one_region": 1},
       "model": {"n_classes": 5}, "training": {"save_dir": str(out_dir / "save"), "out_dir": str(out_dir), "batch_size": 4,
                                               "lr": 0.1, "n_epochs": 2, "val_steps": 2}}
import yaml  # noqa: E402
cfg_path = out_dir / f"cfg_{os.environ['RANK']}.yaml"
cfg_path.write_text(yaml.safe_dump(cfg))
_, history = T.main(["--config", str(cfg_path), "--steps_per_epoch", "3"])
report["initialized_after_main"] = dist.is_initialized()
report["history"] = history
report["weights"] = model.fc.weight.detach().flatten().tolist()
(out_dir / f"report_{os.environ['RANK']}.json").write_text(json.dumps(report))
