"""Independent cross-check of the ResNet restatements (oracle/resnet18.py, oracle/resnet50.py).

The reference builds its network with torchvision (models/patch_cls_simple/model.py:5-11), which is not installed here, and holds no vectors of
the network's outputs: by the rules of this build the oracles' parity stays "unpinned".  What CAN be checked on this image is agreement with a
second, independently written implementation of the same published architecture: Hugging Face transformers' `ResNetModel` (installed offline;
`layer_type="basic"` = torchvision's BasicBlock, `"bottleneck"` with `downsample_in_bottleneck=False` = torchvision's v1.5 Bottleneck with the
stride on the 3x3 convolution).  The oracle's parameters and running statistics are copied into it by name; eval-mode logits, training-mode
(batch-statistic) logits, the updated running statistics and every parameter gradient must agree to float32 round-off.  A wrong stride, padding,
block order, shortcut rule or BN setting in the restatement would show here.  CPU only."""
import pytest
import torch
import torch.nn.functional as F

transformers = pytest.importorskip("transformers")


def _hf_name(k: str) -> str | None:
    """torchvision state_dict key -> transformers.ResNetModel state_dict key (None: the classifier head, applied by hand)."""
    parts = k.split(".")
    if parts[0] == "fc":
        return None
    if parts[0] == "conv1":
        return "embedder.embedder.convolution." + parts[1]
    if parts[0] == "bn1":
        return "embedder.embedder.normalization." + parts[1]
    stage, blk = int(parts[0][5:]) - 1, int(parts[1])
    pre = f"encoder.stages.{stage}.layers.{blk}."
    if parts[2] == "downsample":
        return pre + ("shortcut.convolution." if parts[3] == "0" else "shortcut.normalization.") + parts[4]
    idx = int(parts[2][-1]) - 1
    return pre + f"layer.{idx}." + ("convolution." if parts[2].startswith("conv") else "normalization.") + parts[3]


def _pair(arch: str, seed: int):
    from transformers import ResNetConfig, ResNetModel
    torch.manual_seed(seed)
    if arch == "resnet18":
        from oracle.resnet18 import ResNet18Oracle
        ora = ResNet18Oracle(5)
        cfg = ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[64, 128, 256, 512], depths=[2, 2, 2, 2], layer_type="basic",
                           hidden_act="relu", downsample_in_first_stage=False)
    else:
        from oracle.resnet50 import ResNet50Oracle
        ora = ResNet50Oracle(5)
        cfg = ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[256, 512, 1024, 2048], depths=[3, 4, 6, 3], layer_type="bottleneck",
                           hidden_act="relu", downsample_in_first_stage=False, downsample_in_bottleneck=False)
    with torch.no_grad():   # non-trivial BN parameters and running statistics
        for m in ora.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.3, 0.3)
                m.running_mean.uniform_(-0.2, 0.2); m.running_var.uniform_(0.5, 2.0)
    hf = ResNetModel(cfg)
    sd, hsd = ora.state_dict(), hf.state_dict()
    mapped = {}
    for k, v in sd.items():
        hk = _hf_name(k)
        if hk is not None:
            assert hk in hsd and hsd[hk].shape == v.shape, (k, hk)
            mapped[hk] = v.clone()
    assert set(mapped) == set(hsd), sorted(set(hsd) - set(mapped))[:5]   # every tensor of the second implementation is accounted for
    hf.load_state_dict(mapped)
    return ora, hf


def _hf_logits(hf, ora, x):
    return F.linear(torch.flatten(hf(pixel_values=x).pooler_output, 1), ora.fc.weight, ora.fc.bias)


@pytest.mark.parametrize("arch,size", [("resnet18", 96), ("resnet18", 224), ("resnet50", 96)])
def test_eval_logits_agree_with_an_independent_implementation(arch, size):
    ora, hf = _pair(arch, 0)
    ora.eval(); hf.eval()
    x = torch.rand(2, 3, size, size)
    with torch.no_grad():
        a, b = ora(x), _hf_logits(hf, ora, x)
    assert torch.allclose(a, b, rtol=1e-5, atol=1e-5 * float(a.abs().max())), float((a - b).abs().max())


@pytest.mark.parametrize("arch", ["resnet18", "resnet50"])
def test_training_step_agrees_with_an_independent_implementation(arch):
    """Batch-statistic forward, CrossEntropy(mean) backward (train.py:166-172): logits, running statistics and all parameter gradients."""
    ora, hf = _pair(arch, 1)
    ora.train(); hf.train()
    x = torch.rand(4, 3, 64, 64)
    y = torch.tensor([0, 3, 1, 4])
    la = ora(x)
    lb = _hf_logits(hf, ora, x)
    assert torch.allclose(la, lb, rtol=1e-4, atol=1e-5 * float(la.detach().abs().max()))
    ga = torch.autograd.grad(F.cross_entropy(la, y), [p for n, p in ora.named_parameters() if not n.startswith("fc.")])
    hp = dict(hf.named_parameters())
    gb = torch.autograd.grad(F.cross_entropy(lb, y), [hp[_hf_name(n)] for n, _ in ora.named_parameters() if not n.startswith("fc.")])
    for (n, _), u, v in zip([(n, p) for n, p in ora.named_parameters() if not n.startswith("fc.")], ga, gb):
        err = float((u - v).norm() / (u.norm() + 1e-12))
        assert err < 2e-4, (n, err)
    hb = dict(hf.named_buffers())
    for n, buf in ora.named_buffers():
        if n.endswith("num_batches_tracked"):
            continue
        assert torch.allclose(buf, hb[_hf_name(n)], rtol=1e-5, atol=1e-6), n
