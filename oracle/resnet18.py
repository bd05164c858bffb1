"""Oracle for rows a6/a7: ResNet-18 patch classifier, torch CPU fp32.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  **Parity unpinned** against the
reference: models/patch_cls_simple/model.py:5-11 builds
`torchvision.models.resnet18(weights=DEFAULT)` and swaps `fc` for
`nn.Linear(512, n_classes)`; torchvision is a third-party dependency that is
not vendored in the reference, unpinned in environment.yaml and not installed
here, and the DEFAULT weights are a download.  This file restates the published
torchvision ResNet-18 (He et al. 2015; BasicBlock x [2,2,2,2], stem 7x7/2 +
maxpool 3x3/2, 1x1/2 conv+BN downsample on the first block of stages 2-4,
global average pool, linear head) with `torch.nn` primitives, using the same
`state_dict` key names so checkpoints interchange, and
torchvision's initialisation (Kaiming-normal fan_out for convs, BN weight 1 /
bias 0, nn.Linear default for fc).  Anchors: model.py:5-11 (factory),
train.py:117-118,166-172 (CrossEntropyLoss mean, Adam lr, step order),
examples/predict_full_patched.py:66-78 (inference: raw logits, no softmax).

Cross-check (round 4): tests/test_oracle_crosscheck.py copies this model's parameters and running statistics into Hugging Face
transformers' `ResNetModel` (5.15.0, installed offline; an independently written implementation of the same published
architecture) and requires equal eval logits, training-mode logits, running statistics and parameter gradients to float32
round-off.  That is evidence for the restatement, not the reference's own dependency: "parity unpinned" stands.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

STAGES = ((64, 1), (128, 2), (256, 2), (512, 2))  # (channels, stride of first block)


class _Block(nn.Module):
    def __init__(self, cin: int, cout: int, stride: int):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(
                nn.Conv2d(cin, cout, 1, stride, 0, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = F.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return F.relu(y + idt)


class ResNet18Oracle(nn.Module):
    """state_dict keys == torchvision resnet18 with fc: [n_classes, 512]."""

    def __init__(self, n_classes: int = 5):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        cin = 64
        for i, (c, s) in enumerate(STAGES, start=1):
            setattr(self, f"layer{i}", nn.Sequential(_Block(cin, c, s), _Block(c, c, 1)))
            cin = c
        self.fc = nn.Linear(512, n_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        x = F.relu(self.bn1(self.conv1(x)))
        x = F.max_pool2d(x, 3, 2, 1)
        for i in range(1, 5):
            x = getattr(self, f"layer{i}")(x)
        x = torch.flatten(F.adaptive_avg_pool2d(x, 1), 1)
        return self.fc(x)


def seeded_model(seed: int, n_classes: int = 5, perturb_bn: bool = False) -> ResNet18Oracle:
    """Deterministic random-init model (pretrained weights are unobtainable offline).

    perturb_bn=True also randomises BN affine parameters and running statistics so
    that eval-mode BN is a non-trivial per-channel scale/shift in parity tests."""
    g = torch.Generator().manual_seed(seed)
    state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    try:
        m = ResNet18Oracle(n_classes)
    finally:
        torch.random.set_rng_state(state)
    if perturb_bn:
        with torch.no_grad():
            for mod in m.modules():
                if isinstance(mod, nn.BatchNorm2d):
                    c = mod.num_features
                    mod.weight.copy_(0.5 + torch.rand(c, generator=g))
                    mod.bias.copy_(0.1 * torch.randn(c, generator=g))
                    mod.running_mean.copy_(0.1 * torch.randn(c, generator=g))
                    mod.running_var.copy_(0.5 + torch.rand(c, generator=g))
    return m


def train_step(model: nn.Module, opt: torch.optim.Optimizer, x: torch.Tensor,
               labels: torch.Tensor) -> tuple[float, torch.Tensor]:
    """One step as train.py:166-172: zero_grad, forward, CE(mean), backward, Adam step."""
    model.train()
    opt.zero_grad()
    out = model(x)
    loss = F.cross_entropy(out, labels)
    loss.backward()
    opt.step()
    return float(loss.item()), out.detach()
