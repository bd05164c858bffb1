"""`get_model(n_classes)` -- drop-in for models/patch_cls_simple/model.py:5-11.

The reference returns torchvision's ResNet-18 with `fc` swapped for
`nn.Linear(512, n_classes)`.  This module returns an `nn.Module` with the same
`state_dict` keys/shapes (so `best_model.pth` files interchange,
examples/predict_full_patched.py:116-126) whose forward pass runs entirely in the
hand-written HIP kernels of libdeephisto_hip.so.  The torch sub-modules below are
parameter *holders* only -- their own forward() is never called.

Pretrained ImageNet weights (`ResNet18_Weights.DEFAULT` in the reference) are a
download and unobtainable offline: parameters get torchvision's random
initialisation; load a checkpoint with `load_state_dict` for real use.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from ..._lib import DH_DTYPE_BF16, DH_DTYPE_F32, check, lib

_STAGES = ((64, 1), (128, 2), (256, 2), (512, 2))


class _BlockParams(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, 0, bias=False),
                                            nn.BatchNorm2d(cout))


class ResNet18HIP(nn.Module):
    """ResNet-18 patch classifier; forward = dh_resnet18_forward (gfx950 MFMA kernels)."""

    def __init__(self, n_classes: int, compute_dtype: str = "f32"):
        super().__init__()
        if compute_dtype not in ("f32", "bf16"):
            raise ValueError("compute_dtype must be 'f32' or 'bf16'")
        self.n_classes = int(n_classes)
        self.compute_dtype = compute_dtype
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        cin = 64
        for i, (c, s) in enumerate(_STAGES, start=1):
            setattr(self, f"layer{i}", nn.Sequential(_BlockParams(cin, c, s), _BlockParams(c, c, 1)))
            cin = c
        self.fc = nn.Linear(512, n_classes)
        for m in self.modules():  # torchvision's ResNet initialisation
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        self._handle = None
        self._synced = None  # signature of the parameter versions held by the handle
        self._lanes = []     # extra native handles (own workspace each) for multi-stream inference

    # ---- native handle management -------------------------------------------------
    def _signature(self):
        return tuple((k, v.data_ptr(), v._version) for k, v in self.state_dict(keep_vars=True).items())

    def _ensure_handle(self):
        if self._handle is None:
            h = C.c_void_p()
            code = DH_DTYPE_F32 if self.compute_dtype == "f32" else DH_DTYPE_BF16
            check(lib().dh_resnet18_create(C.byref(h), self.n_classes, code), "dh_resnet18_create")
            self._handle = h
        sig = self._signature()
        if sig != self._synced:
            for name, t in self.state_dict().items():
                if name.endswith("num_batches_tracked"):
                    continue
                a = t.detach().to("cpu", torch.float32).contiguous()
                check(lib().dh_resnet18_set_param(self._handle, name.encode(), a.data_ptr(), a.numel()),
                      f"dh_resnet18_set_param({name})")
            check(lib().dh_resnet18_finalize(self._handle, None), "dh_resnet18_finalize")
            self._synced = sig
        return self._handle

    def lane_handles(self, n: int):
        """n native handles holding the current parameters, each with its own activation
        workspace, so that n micro-batches can be in flight on n HIP streams."""
        first = self._ensure_handle()
        sig = self._synced
        while len(self._lanes) < n - 1:
            self._lanes.append([C.c_void_p(), None])
        for lane in self._lanes[:n - 1]:
            if lane[1] != sig:
                if not lane[0]:
                    code = DH_DTYPE_F32 if self.compute_dtype == "f32" else DH_DTYPE_BF16
                    check(lib().dh_resnet18_create(C.byref(lane[0]), self.n_classes, code), "dh_resnet18_create")
                for name, t in self.state_dict().items():
                    if name.endswith("num_batches_tracked"):
                        continue
                    a = t.detach().to("cpu", torch.float32).contiguous()
                    check(lib().dh_resnet18_set_param(lane[0], name.encode(), a.data_ptr(), a.numel()),
                          f"dh_resnet18_set_param({name})")
                check(lib().dh_resnet18_finalize(lane[0], None), "dh_resnet18_finalize")
                lane[1] = sig
        return [first] + [lane[0] for lane in self._lanes[:n - 1]]

    def set_compute_dtype(self, compute_dtype: str):
        if compute_dtype != self.compute_dtype:
            self._release()
            self.compute_dtype = compute_dtype
        return self

    def _release(self):
        if getattr(self, "_handle", None) is not None:
            lib().dh_resnet18_destroy(self._handle)
            self._handle, self._synced = None, None
        for lane in getattr(self, "_lanes", []):
            if lane[0]:
                lib().dh_resnet18_destroy(lane[0])
        self._lanes = []

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    # ---- forward --------------------------------------------------------------------
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x: float32[n, 3, P, P] on the GPU (what batch_predictor builds,
        predict_full_patched.py:67-71) -> float32[n, n_classes] raw logits."""
        if self.training:
            raise NotImplementedError(
                "ResNet18HIP: training-mode forward/backward kernels are not built yet "
                "(SURVEY section 8 row a7); call .eval() for inference")
        if not x.is_cuda:
            raise RuntimeError("ResNet18HIP runs on the GPU only: move the input with .to('cuda')")
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != x.shape[3]:
            raise ValueError(f"expected [n, 3, P, P], got {tuple(x.shape)}")
        x = x.detach().to(torch.float32).contiguous()
        h = self._ensure_handle()
        n, p = int(x.shape[0]), int(x.shape[2])
        out = torch.empty((n, self.n_classes), dtype=torch.float32, device=x.device)
        stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        check(lib().dh_resnet18_forward(h, x.data_ptr(), n, p, out.data_ptr(), stream), "dh_resnet18_forward")
        return out

    def forward_tiles(self, slide: torch.Tensor, origins_dev: torch.Tensor, patch: int) -> torch.Tensor:
        """Fused gather + /255 + forward straight from the uint8 slide in HBM."""
        if self.training:
            raise NotImplementedError("forward_tiles is an inference entry point; call .eval()")
        if not (slide.is_cuda and origins_dev.is_cuda):
            raise RuntimeError("slide and origins must live in GPU memory")
        if slide.dtype != torch.uint8 or slide.dim() != 3 or not slide.is_contiguous():
            raise ValueError("slide must be contiguous uint8[h, w, 3]")
        if origins_dev.dtype != torch.int32 or not origins_dev.is_contiguous():
            raise ValueError("origins must be contiguous int32[n, 2]")
        h = self._ensure_handle()
        n = int(origins_dev.shape[0])
        out = torch.empty((n, self.n_classes), dtype=torch.float32, device=slide.device)
        stream = C.c_void_p(torch.cuda.current_stream(slide.device).cuda_stream)
        check(lib().dh_resnet18_forward_tiles(h, slide.data_ptr(), int(slide.shape[0]), int(slide.shape[1]),
                                              origins_dev.data_ptr(), n, patch, out.data_ptr(), stream),
              "dh_resnet18_forward_tiles")
        return out


def get_model(n_classes: int, compute_dtype: str = "f32") -> ResNet18HIP:
    """Same call as the reference's `get_model(n_classes)` (model.py:5)."""
    return ResNet18HIP(n_classes, compute_dtype)
