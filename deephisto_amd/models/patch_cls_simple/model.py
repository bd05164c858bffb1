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

from ..._lib import BUCKET_CB, DH_DTYPE_BF16, DH_DTYPE_F32, check, lib
from .ddp import DEFAULT_BUCKET_BYTES, BucketReducer, allreduce_mean_  # noqa: F401  (allreduce_mean_ re-exported)

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


class _DevView:
    """float32 device memory owned by the native library, exposed through the CUDA array interface."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}


def ce_loss(logits: torch.Tensor, labels: torch.Tensor, want_grad: bool = False):
    """Mean cross entropy of float32[n, n_cls] logits against int64 labels on the GPU (`dh_ce_loss`:
    nn.CrossEntropyLoss() of train.py:117).  Returns the scalar loss tensor, or (loss, dlogits) with
    dlogits = (softmax - onehot) / n when `want_grad`."""
    if not logits.is_cuda:
        raise RuntimeError("ce_loss runs on the GPU only")
    logits = logits.detach().to(torch.float32).contiguous()
    labels = labels.to(device=logits.device, dtype=torch.int64).contiguous()
    loss = torch.empty((), dtype=torch.float32, device=logits.device)
    dl = torch.empty_like(logits) if want_grad else None
    st = C.c_void_p(torch.cuda.current_stream(logits.device).cuda_stream)
    check(lib().dh_ce_loss(logits.data_ptr(), labels.data_ptr(), logits.shape[0], logits.shape[1], loss.data_ptr(),
                           dl.data_ptr() if want_grad else None, st), "dh_ce_loss")
    return (loss, dl) if want_grad else loss


class _TrainForward(torch.autograd.Function):
    """logits = model(x) in training mode; backward runs the HIP backward kernels and hands
    every parameter its gradient (so `loss.backward(); optimizer.step()` of
    models/patch_cls_simple/train.py:171-172 work unchanged on the nn.Parameters)."""

    @staticmethod
    def forward(ctx, x, model, *params):
        ctx.model, ctx.x = model, x  # x must outlive backward (the stem wgrad reads it)
        return model._native_forward_train(x)

    @staticmethod
    def backward(ctx, dlogits):
        grads = ctx.model._native_backward(dlogits.contiguous())
        return (None, None, *grads)


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
        self._engine2 = None  # bf16 training engine (dh_train2), created on the first bf16 training forward

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
        self._train_shape = None
        if getattr(self, "_handle", None) is not None:
            lib().dh_resnet18_destroy(self._handle)
            self._handle, self._synced = None, None
        for lane in getattr(self, "_lanes", []):
            if lane[0]:
                lib().dh_resnet18_destroy(lane[0])
        self._lanes = []
        if getattr(self, "_engine2", None) is not None:
            self._engine2.release()
            self._engine2 = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    # ---- forward --------------------------------------------------------------------
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x: float32[n, 3, P, P] on the GPU (what batch_predictor builds,
        predict_full_patched.py:67-71) -> float32[n, n_classes] raw logits."""
        if not x.is_cuda:
            raise RuntimeError("ResNet18HIP runs on the GPU only: move the input with .to('cuda')")
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != x.shape[3]:
            raise ValueError(f"expected [n, 3, P, P], got {tuple(x.shape)}")
        x = x.detach().to(torch.float32).contiguous()
        if self.training and self.compute_dtype == "bf16":   # bf16 activations / MFMA, f32 masters: the dh_train2 engine
            e2 = self._bf16_engine()
            e2.pull_parameters()
            if torch.is_grad_enabled():
                return _TrainForward2.apply(x, e2, *self.parameters())
            return e2.forward(x, True)
        if self.training:
            if torch.is_grad_enabled():
                return _TrainForward.apply(x, self, *self.parameters())
            return self._native_forward_train(x)  # e.g. train-mode forward under no_grad
        h = self._ensure_handle()
        n, p = int(x.shape[0]), int(x.shape[2])
        out = torch.empty((n, self.n_classes), dtype=torch.float32, device=x.device)
        stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        check(lib().dh_resnet18_forward(h, x.data_ptr(), n, p, out.data_ptr(), stream), "dh_resnet18_forward")
        return out

    def _bf16_engine(self):
        if self._engine2 is None:
            self._engine2 = Train2Engine(self, "resnet18", self.n_classes)
        return self._engine2

    # ---- training (row a7): HIP forward/backward behind torch autograd ---------------------
    def _stream(self, dev):
        return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def _push_changed_parameters(self, stream):
        """nn.Parameters updated by a torch optimizer -> library masters (device copies)."""
        seen = getattr(self, "_pushed", None)
        if seen is None:
            seen = self._pushed = {}
        dirty = False
        for name, prm in self.named_parameters():
            key = (prm.data_ptr(), prm._version)
            if seen.get(name) != key:
                if name in seen:  # first sight = just uploaded through set_param / train_begin
                    t = prm.detach().to(torch.float32).contiguous()
                    check(lib().dh_resnet18_train_tensor(self._handle, name.encode(), 0, t.data_ptr(), t.numel(), 1,
                                                         stream), f"push {name}")
                    dirty = True
                seen[name] = key
        if dirty:
            check(lib().dh_resnet18_train_repack(self._handle, stream), "dh_resnet18_train_repack")
        if getattr(self, "_buffers_dirty", False):   # load_state_dict: running statistics -> library
            for name, buf in self.named_buffers():
                if not name.endswith("num_batches_tracked"):
                    t = buf.detach().to(torch.float32).contiguous()
                    check(lib().dh_resnet18_train_tensor(self._handle, name.encode(), 2, t.data_ptr(), t.numel(), 1, stream),
                          f"push {name}")
            self._buffers_dirty = False

    def _native_forward_train(self, x, pull_stats=True):
        n, p = int(x.shape[0]), int(x.shape[2])
        st = self._stream(x.device)
        if self._handle is None or getattr(self, "_train_shape", None) is None:
            h = self._ensure_handle()   # uploads the current parameters once; later steps push deltas on-device
            self._pushed = None
        else:
            h = self._handle
        check(lib().dh_resnet18_train_begin(h, n, p, st), "dh_resnet18_train_begin")
        self._train_shape = (n, p)
        self._push_changed_parameters(st)
        out = torch.empty((n, self.n_classes), dtype=torch.float32, device=x.device)
        check(lib().dh_resnet18_forward_train(h, x.data_ptr(), n, p, out.data_ptr(), st), "dh_resnet18_forward_train")
        self._stats_pending = getattr(self, "_stats_pending", 0) + 1
        if pull_stats:
            self._pull_running_stats()
        return out

    def _pull_running_stats(self):
        """Library running statistics -> module buffers (+ the batch counters owed since the last pull)."""
        owed = getattr(self, "_stats_pending", 0)
        if not owed or self._handle is None:
            return
        with torch.no_grad():
            for name, buf in self.named_buffers():
                if name.endswith("num_batches_tracked"):
                    buf += owed
                else:
                    check(lib().dh_resnet18_train_tensor(self._handle, name.encode(), 2, buf.data_ptr(), buf.numel(), 0,
                                                         self._stream(buf.device)), f"pull {name}")
        self._stats_pending = 0

    def _native_backward(self, dlogits):
        h, st = self._handle, self._stream(dlogits.device)
        check(lib().dh_resnet18_backward(h, dlogits.data_ptr(), st), "dh_resnet18_backward")
        grads = []
        for name, prm in self.named_parameters():
            g = torch.empty_like(prm, dtype=torch.float32)
            check(lib().dh_resnet18_train_tensor(h, name.encode(), 1, g.data_ptr(), g.numel(), 0, st), f"grad {name}")
            grads.append(g if prm.requires_grad else None)
        return grads

    def flat_gradients(self, device) -> torch.Tensor:
        """The library's whole gradient arena as one float32 tensor view (no copy)."""
        ptr, n = C.c_void_p(), C.c_int64()
        check(lib().dh_resnet18_train_flat(self._handle, 1, C.byref(ptr), C.byref(n)), "dh_resnet18_train_flat")
        return torch.as_tensor(_DevView(ptr.value, n.value), device=device)

    def train_step(self, x, labels, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, group=None, bucket_bytes=DEFAULT_BUCKET_BYTES):
        """Fused step entirely in HIP: forward, CrossEntropy(mean), backward, Adam.
        Under torch.distributed (one process per GPU) the gradients are averaged over `group`: the arena is cut
        into ~`bucket_bytes` buckets in backward-completion order and each bucket's all-reduce starts on a side
        stream as soon as its last wgrad is enqueued (models/patch_cls_simple/ddp.py); Adam waits for the last
        one (DDP semantics: per-rank batch statistics, replicas stay identical).
        Returns (loss tensor on device, logits).  nn.Parameters are refreshed lazily by
        `pull_parameters()` / state_dict()."""
        if not self.training:
            raise RuntimeError("train_step needs .train() mode")
        if self.compute_dtype == "bf16":
            self._synced = None   # the eval-mode copy of the parameters must be rebuilt after this update
            return self._bf16_engine().train_step(x, labels, lr, betas, eps, group)
        x = x.detach().to(torch.float32).contiguous()
        labels = labels.to(device=x.device, dtype=torch.int64).contiguous()
        logits = self._native_forward_train(x, pull_stats=False)   # 40 small copies per step otherwise; pulled lazily
        st = self._stream(x.device)
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        dl = torch.empty_like(logits)
        check(lib().dh_ce_loss(logits.data_ptr(), labels.data_ptr(), logits.shape[0], self.n_classes, loss.data_ptr(),
                               dl.data_ptr(), st), "dh_ce_loss")
        import torch.distributed as dist
        red = None
        single = not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1)
        if single and getattr(self, "fuse_optimizer", True):   # no gradient exchange: the update rides behind each block's weight gradients
            self._adam_t = getattr(self, "_adam_t", 0) + 1
            check(lib().dh_resnet18_backward_adam(self._handle, dl.data_ptr(), lr, betas[0], betas[1], eps, self._adam_t, st),
                  "dh_resnet18_backward_adam")
            self._native_ahead = True
            return loss, logits
        if not single:
            red = BucketReducer(self.flat_gradients(x.device), group, getattr(self, "ddp_wire", None))   # None: DH_DDP_WIRE (f32 | bf16)
            cb = BUCKET_CB(lambda bucket, off, cnt, _user: red.on_bucket(bucket, off, cnt))
            check(lib().dh_resnet18_set_buckets(self._handle, int(bucket_bytes), cb, None, None), "dh_resnet18_set_buckets")
        try:
            check(lib().dh_resnet18_backward(self._handle, dl.data_ptr(), st), "dh_resnet18_backward")
            if red is not None:
                red.finish()
                self.overlap_log = red.log
        finally:
            if red is not None:   # never leave the library holding a callback into a dead trampoline
                check(lib().dh_resnet18_set_buckets(self._handle, 0, None, None, None), "dh_resnet18_set_buckets")
        self._adam_t = getattr(self, "_adam_t", 0) + 1
        check(lib().dh_resnet18_adam_step(self._handle, lr, betas[0], betas[1], eps, self._adam_t, st), "dh_resnet18_adam_step")
        self._native_ahead = True
        return loss, logits

    def pull_parameters(self):
        """Library masters and running statistics -> nn.Parameters / buffers (after fused train_step calls)."""
        if self._engine2 is not None:
            self._engine2.pull_parameters()
        self._pull_running_stats()
        if getattr(self, "_native_ahead", False):
            with torch.no_grad():
                for name, prm in self.named_parameters():
                    st = self._stream(prm.device)
                    check(lib().dh_resnet18_train_tensor(self._handle, name.encode(), 0, prm.data_ptr(), prm.numel(), 0, st),
                          f"pull {name}")
                    self._pushed[name] = (prm.data_ptr(), prm._version)
            self._native_ahead = False
            self._synced = None  # the raw copies above do not bump torch's version counters: force the
            #                      eval-mode copy of the parameters to be rebuilt on its next use
        return self

    def state_dict(self, *args, **kwargs):
        self.pull_parameters()
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        """The loaded tensors win over whatever the library holds after fused `train_step` calls (which leave the
        library's masters newer than the nn.Parameters): nothing is pulled back over them, the eval-mode copy is
        rebuilt, and the next training forward pushes parameters (version counters) and running statistics."""
        self._native_ahead, self._stats_pending = False, 0
        if self._engine2 is not None:
            self._engine2.native_ahead, self._engine2._stats_pending = False, 0
        out = super().load_state_dict(*args, **kwargs)
        self._synced = None
        self._buffers_dirty = True
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


def get_model(n_classes: int, compute_dtype: str = "f32", arch: str = "resnet18") -> nn.Module:
    """Same call as the reference's `get_model(n_classes)` (model.py:5).  `arch="resnet50"` selects the ResNet-50
    backbone of BASELINE.json configs[4] (bf16 engine, whatever `compute_dtype` says)."""
    if arch == "resnet18":
        return ResNet18HIP(n_classes, compute_dtype)
    if arch == "resnet50":
        return ResNet50HIP(n_classes)
    raise ValueError(f"unknown architecture {arch!r} (resnet18, resnet50)")


from .resnet_bf16 import ResNet50HIP, Train2Engine, _TrainForward2  # noqa: E402  (needs ce_loss above)
