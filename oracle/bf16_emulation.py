"""Oracle variant for the bf16 engine: the same torch-CPU network evaluated with bf16 ROUNDING at the points where the
HIP engine (deephisto_amd/csrc/train2.inc) stores bf16 -- conv weights, the network input of the stem, every conv output Z,
every BN(+identity)(+ReLU) output Y, the max-pool output -- and float32 arithmetic in between (products of bf16 values are
exact in float32, accumulation is float32 like the MFMA's).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Why it exists: a randomly initialised 50-layer network with batch-statistic
BN amplifies the 2^-9 relative rounding of bf16 layer by layer (tools/t2_check.py: 0.3 % after the stem, tens of percent at
the logits for tiny batches), so a float32 oracle cannot separate a kernel bug from bf16 arithmetic.  Against THIS oracle the
engine differs only by accumulation order and by the bf16 rounding of activation GRADIENTS, which stays small; the float32
oracle stays the reference for the stated end-to-end bf16 tolerance.  Rounding is a straight-through estimator (identity
gradient), which is what storing a rounded activation and back-propagating through the unrounded formula amounts to.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def rb(t: torch.Tensor) -> torch.Tensor:
    """Round to bf16 (nearest even), keep float32 storage; identity gradient."""
    return t + (t.detach().bfloat16().float() - t.detach())


class _RoundGrad(torch.autograd.Function):
    """Identity in the forward pass; the GRADIENT flowing back through it is rounded to bf16 -- the engine stores every activation
    gradient in bf16: the dX a dgrad kernel writes (a conv's input gradient, both branches of a block already summed in the
    kernel's float32 epilogue) and the dZ a BN backward writes (a conv's output gradient)."""

    @staticmethod
    def forward(ctx, t):
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


_GRAD_ROUNDING = False   # set by forward_bf16(..., grad_rounding=True) for the duration of the call


def rg(t: torch.Tensor) -> torch.Tensor:
    return _RoundGrad.apply(t) if _GRAD_ROUNDING and t.requires_grad else t


def _conv(x, conv, round_in=True):
    # rg(x): the conv's input gradient (dgrad output) is stored in bf16; the outer rg: so is its output gradient (BN backward's dZ).
    # round_in=False: the first conv of a block without a downsample branch -- its dgrad kernel adds the identity gradient in float32
    # and rounds the SUM once (the rg on the block input)
    return rg(rb(F.conv2d(rg(x) if round_in else x, rb(conv.weight), None, conv.stride, conv.padding)))


def _bn(z, bn, training):
    return F.batch_norm(z, bn.running_mean, bn.running_var, bn.weight, bn.bias, training, bn.momentum, bn.eps)


def forward_bf16(model, x: torch.Tensor, record: dict | None = None, masks: dict | None = None, grad_rounding: bool = False,
                 forced: dict | None = None) -> torch.Tensor:
    """Logits of `model` (oracle.resnet18.ResNet18Oracle or oracle.resnet50.ResNet50Oracle) with the engine's rounding
    points.  `model.training` selects batch statistics (running statistics are updated, as in the engine).
    `record[name]` receives the rounded conv outputs Z by conv name.

    `masks[name]` (bool NCHW, optional): use THIS ReLU pattern after conv `name`'s BN instead of the sign of the oracle's own
    pre-activation.  Gradient parity of two bf16 forwards is otherwise dominated by ReLUs whose pre-activation is within the
    forward difference of zero: a 2 % forward difference flips ~2 % of the pattern, and every flipped element moves the
    gradient by 100 % of itself (relative L2 ~ sqrt(fraction) ~ 0.1 per layer; measured with tools/t2_check.py).  With the
    engine's own pattern imposed, what remains is the arithmetic of the backward kernels.

    `grad_rounding` (round 4): also follow the engine's rounding points of the BACKWARD pass -- every conv's input gradient and
    output gradient is rounded to bf16 where the engine stores it (class _RoundGrad).  A block input feeds two branches; autograd sums
    their gradients in float32 before the rounding node, as the dgrad kernel's epilogue does (identity gradient + W^T dZ, one
    rounding).

    `forced[name]` (NCHW float32, optional; round 4): the ENGINE's stored output Y of conv `name`'s BN (+ identity)(+ ReLU).  Its VALUE
    replaces the emulation's own (the gradient still flows through the emulation's formula): every layer of the backward pass then sees
    the engine's own operands -- the forward difference of two bf16 networks (percents at the deepest layers of a random-init net,
    module docstring) no longer enters the weight gradients through their activation operand, and what is compared is the
    composition of the backward pass."""
    global _GRAD_ROUNDING
    _GRAD_ROUNDING = bool(grad_rounding)
    try:
        return _forward_bf16(model, x, record, masks, forced)
    finally:
        _GRAD_ROUNDING = False


def _forward_bf16(model, x, record, masks, forced=None):
    tr = model.training

    def out(h, name):   # the stored activation: the engine's value where given
        h = rb(h)
        if forced is not None and name in forced:
            h = h + (forced[name].to(h.dtype) - h).detach()
        return h

    def relu(h, name):
        if masks is None:
            return F.relu(h)
        return h * masks[name].to(h.dtype)

    def conv(name, mod, inp, round_in=True):
        z = _conv(inp, mod, round_in)
        if record is not None:
            record[name] = z.detach()
        return z

    x = rb(x)   # the stem converts the float input image to bf16
    y = out(relu(_bn(conv("conv1", model.conv1, x), model.bn1, tr), "conv1"), "conv1")
    y = F.max_pool2d(y, 3, 2, 1)
    for li in range(1, 5):
        for bi, blk in enumerate(getattr(model, f"layer{li}")):
            pre = f"layer{li}.{bi}"
            y = rg(y)     # the block's input gradient: both branches summed, rounded once (dgrad epilogue)
            idt = y
            if blk.downsample is not None:
                idt = rb(_bn(conv(pre + ".downsample.0", blk.downsample[0], y), blk.downsample[1], tr))
            h = out(relu(_bn(conv(pre + ".conv1", blk.conv1, y, blk.downsample is not None), blk.bn1, tr), pre + ".conv1"), pre + ".conv1")
            if hasattr(blk, "conv3"):
                h = out(relu(_bn(conv(pre + ".conv2", blk.conv2, h), blk.bn2, tr), pre + ".conv2"), pre + ".conv2")
                h = _bn(conv(pre + ".conv3", blk.conv3, h), blk.bn3, tr)
                last = pre + ".conv3"
            else:
                h = _bn(conv(pre + ".conv2", blk.conv2, h), blk.bn2, tr)
                last = pre + ".conv2"
            y = out(relu(h + idt, last), last)
    y = rg(y)   # the gradient of the last block output (avgpool + fc backward) is stored in bf16 too
    pooled = torch.flatten(F.adaptive_avg_pool2d(y, 1), 1)
    return F.linear(pooled, model.fc.weight, model.fc.bias)
