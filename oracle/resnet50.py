"""Oracle for BASELINE configs[4]: ResNet-50 patch classifier, torch CPU fp32.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  **Parity unpinned** against the reference: the reference's
factory (models/patch_cls_simple/model.py:5-11) builds torchvision's resnet18; BASELINE.json configs[4] names the
same factory with a ResNet-50 backbone, i.e. `torchvision.models.resnet50` + `fc = nn.Linear(2048, n_classes)`.
torchvision is third-party, unpinned in environment.yaml and not installed here.  This file restates the published
torchvision ResNet-50 (v1.5: Bottleneck x [3,4,6,3], expansion 4, the stride of a stage's first block on its 3x3
conv; stem 7x7/2 + maxpool 3x3/2; 1x1 conv+BN downsample on the first block of EVERY stage, stride 1 in stage 1;
global average pool; linear head) with `torch.nn` primitives, the same `state_dict` key names
(`layer2.0.conv3.weight`, `layer2.0.downsample.1.running_var`, ...) and torchvision's initialisation
(Kaiming-normal fan_out for convs, BN weight 1 / bias 0, nn.Linear default for fc; zero_init_residual off).
23 518 277 parameters at n_classes = 5.  Step semantics: train.py:117-118, 166-172.

Cross-check (round 4): tests/test_oracle_crosscheck.py copies this model's parameters and running statistics into Hugging Face
transformers' `ResNetModel` (5.15.0, installed offline; an independently written implementation of the same published
architecture) and requires equal eval logits, training-mode logits, running statistics and parameter gradients to float32
round-off.  That is evidence for the restatement, not the reference's own dependency: "parity unpinned" stands.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

STAGES = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))  # (width, blocks, stride of first block)


class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin: int, width: int, stride: int):
        super().__init__()
        cout = width * self.expansion
        self.conv1 = nn.Conv2d(cin, width, 1, 1, 0, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, cout, 1, 1, 0, bias=False)
        self.bn3 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, 0, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return F.relu(y + idt)


class ResNet50Oracle(nn.Module):
    """state_dict keys == torchvision resnet50 with fc: [n_classes, 2048]."""

    def __init__(self, n_classes: int = 5):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        cin = 64
        for i, (w, n, s) in enumerate(STAGES, start=1):
            blocks = [_Bottleneck(cin, w, s)] + [_Bottleneck(4 * w, w, 1) for _ in range(n - 1)]
            setattr(self, f"layer{i}", nn.Sequential(*blocks))
            cin = 4 * w
        self.fc = nn.Linear(2048, n_classes)
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


def seeded_model(seed: int, n_classes: int = 5, perturb_bn: bool = False) -> ResNet50Oracle:
    """Deterministic random-init model; perturb_bn also randomises BN affine parameters and running statistics."""
    g = torch.Generator().manual_seed(seed)
    state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    try:
        m = ResNet50Oracle(n_classes)
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
