"""bf16 training / evaluation of the patch classifiers: ResNet-50 (BASELINE.json configs[4]) and ResNet-18.

`get_model(n_classes, arch="resnet50")` returns `ResNet50HIP`: an `nn.Module` with torchvision's ResNet-50
`state_dict` layout (Bottleneck x [3,4,6,3], `fc = Linear(2048, n_classes)`) whose forward, backward and optimizer
step run in the bf16 engine of libdeephisto_hip.so (`dh_train2_*`: bf16 activations, bf16 MFMA with f32 accumulation
for forward, dgrad and wgrad; f32 master weights, gradients and Adam).  The step it replaces is
models/patch_cls_simple/train.py:166-172 (`outputs = model(inputs)`, `criterion`, `loss.backward()`,
`optimizer.step()`); the factory is models/patch_cls_simple/model.py:5-11 with a ResNet-50 backbone.

`Train2Engine` is the handle manager shared with `ResNet18HIP` (bf16 training of the reference's own backbone).
Data-parallel training (one process per GPU, RCCL): the gradient arena is laid out in backward-completion order
and cut into ~25 MB buckets; each bucket's all-reduce starts on a side stream as soon as the backward kernels
that complete it are enqueued, Adam waits for the last one.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from ..._lib import BUCKET_CB, check, lib
from .ddp import DEFAULT_BUCKET_BYTES, BucketReducer


class _DevView:
    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}


class Train2Engine:
    """Owns a dh_train2 handle for `module` (whose parameters / buffers carry torchvision names)."""

    def __init__(self, module: nn.Module, arch: str, n_classes: int):
        self.module, self.arch, self.n_classes = module, arch, int(n_classes)
        self.handle = None
        self._pushed = {}          # tensor name -> (data_ptr, version) last copied into the library
        self._stats_pending = 0    # training forwards whose running statistics were not pulled yet
        self.native_ahead = False  # the library's parameters are newer than the nn.Parameters (fused train_step)
        self._cb = None
        self.overlap_log = []      # (bucket, offset, count) in launch order of the last data-parallel backward (tests)
        self.ddp_wire = None        # gradient wire format of data-parallel steps: None = DH_DDP_WIRE (default f32), "f32", "bf16"
        self.fuse_optimizer = True  # single-rank train_step: dh_train2_backward_adam (False: backward, then adam_step; tests)

    # ---- handle and parameter traffic -------------------------------------------------------------------
    def _stream(self, dev):
        return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def ensure(self, dev):
        if self.handle is None:
            h = C.c_void_p()
            check(lib().dh_train2_create(C.byref(h), self.arch.encode(), self.n_classes), "dh_train2_create")
            self.handle = h
            self._pushed = {}
        self.push_changed(dev)
        return self.handle

    def push_changed(self, dev):
        """nn.Parameters / buffers changed since the last push (torch optimizer step, load_state_dict) -> library."""
        if self.native_ahead:
            return
        st = self._stream(dev)
        for kind, items in ((0, self.module.named_parameters()), (2, self.module.named_buffers())):
            for name, t in items:
                if name.endswith("num_batches_tracked"):
                    continue
                key = (t.data_ptr(), t._version)
                if self._pushed.get(name) != key:
                    src = t.detach().to(device=dev, dtype=torch.float32).contiguous()
                    check(lib().dh_train2_tensor(self.handle, name.encode(), kind, src.data_ptr(), src.numel(), 1, st), f"push {name}")
                    self._pushed[name] = key

    def release(self):
        if self.handle is not None:
            lib().dh_train2_destroy(self.handle)
            self.handle = None

    # ---- forward / backward -----------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, training: bool, pull_stats: bool = True) -> torch.Tensor:
        h = self.ensure(x.device)
        n, p = int(x.shape[0]), int(x.shape[2])
        out = torch.empty((n, self.n_classes), dtype=torch.float32, device=x.device)
        check(lib().dh_train2_forward(h, x.data_ptr(), n, p, out.data_ptr(), 1 if training else 0, self._stream(x.device)),
              "dh_train2_forward")
        if training:
            self._stats_pending += 1
            if pull_stats:
                self.pull_running_stats()
        return out

    def backward(self, dlogits: torch.Tensor):
        st = self._stream(dlogits.device)
        check(lib().dh_train2_backward(self.handle, dlogits.data_ptr(), st), "dh_train2_backward")
        grads = []
        for name, prm in self.module.named_parameters():
            g = torch.empty_like(prm, dtype=torch.float32)
            check(lib().dh_train2_tensor(self.handle, name.encode(), 1, g.data_ptr(), g.numel(), 0, st), f"grad {name}")
            grads.append(g if prm.requires_grad else None)
        return grads

    def flat(self, kind: int, dev) -> torch.Tensor:
        ptr, n = C.c_void_p(), C.c_int64()
        check(lib().dh_train2_flat(self.handle, kind, C.byref(ptr), C.byref(n)), "dh_train2_flat")
        return torch.as_tensor(_DevView(ptr.value, n.value), device=dev)

    def pull_running_stats(self):
        owed = self._stats_pending
        if not owed or self.handle is None:
            return
        with torch.no_grad():
            for name, buf in self.module.named_buffers():
                if name.endswith("num_batches_tracked"):
                    buf += owed
                else:
                    check(lib().dh_train2_tensor(self.handle, name.encode(), 2, buf.data_ptr(), buf.numel(), 0, self._stream(buf.device)),
                          f"pull {name}")
                    self._pushed[name] = (buf.data_ptr(), buf._version)
        self._stats_pending = 0

    def pull_parameters(self):
        self.pull_running_stats()
        if self.native_ahead and self.handle is not None:
            with torch.no_grad():
                for name, prm in self.module.named_parameters():
                    check(lib().dh_train2_tensor(self.handle, name.encode(), 0, prm.data_ptr(), prm.numel(), 0, self._stream(prm.device)),
                          f"pull {name}")
                    self._pushed[name] = (prm.data_ptr(), prm._version)
            self.native_ahead = False

    # ---- data-parallel gradient exchange ------------------------------------------------------------------
    def bucket_ranges(self, bucket_bytes: int = DEFAULT_BUCKET_BYTES):
        """[(offset, count)] of the gradient buckets in completion order (fc first, stem last)."""
        n = C.c_int32()
        check(lib().dh_train2_set_buckets(self.handle, int(bucket_bytes), None, None, C.byref(n)), "dh_train2_set_buckets")
        out = []
        for i in range(n.value):
            off, cnt = C.c_int64(), C.c_int64()
            check(lib().dh_train2_bucket(self.handle, i, C.byref(off), C.byref(cnt)), "dh_train2_bucket")
            out.append((off.value, cnt.value))
        return out

    def _arm_overlap(self, dev, group, bucket_bytes):
        red = BucketReducer(self.flat(1, dev), group, self.ddp_wire)
        # called inside dh_train2_backward right after the kernels completing a bucket were enqueued on the current stream
        self._cb = BUCKET_CB(lambda bucket, off, cnt, _user: red.on_bucket(bucket, off, cnt))   # keep the trampoline alive
        n = C.c_int32()
        check(lib().dh_train2_set_buckets(self.handle, int(bucket_bytes), self._cb, None, C.byref(n)), "dh_train2_set_buckets")
        return red

    def _finish_overlap(self, red, ok=True):
        try:
            if ok:
                red.finish()
                self.overlap_log = red.log
        finally:   # never leave the library holding a callback into a dead trampoline
            check(lib().dh_train2_set_buckets(self.handle, 0, None, None, None), "dh_train2_set_buckets")
            self._cb = None

    def train_step(self, x, labels, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, group=None, bucket_bytes=DEFAULT_BUCKET_BYTES):
        from .model import ce_loss
        import torch.distributed as dist

        x = x.detach().to(torch.float32).contiguous()
        logits = self.forward(x, True, pull_stats=False)
        loss, dl = ce_loss(logits, labels, want_grad=True)
        st = self._stream(x.device)
        world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        if world == 1 and self.fuse_optimizer:   # no gradient exchange: the update rides behind each block's weight gradients
            check(lib().dh_train2_backward_adam(self.handle, dl.data_ptr(), lr, betas[0], betas[1], eps, 0, st), "dh_train2_backward_adam")
            self.native_ahead = True
            return loss, logits
        red = self._arm_overlap(x.device, group, bucket_bytes) if world > 1 else None
        try:
            check(lib().dh_train2_backward(self.handle, dl.data_ptr(), st), "dh_train2_backward")
        except Exception:
            if red is not None:
                self._finish_overlap(red, ok=False)
            raise
        if red is not None:
            self._finish_overlap(red)
        check(lib().dh_train2_adam_step(self.handle, lr, betas[0], betas[1], eps, 0, st), "dh_train2_adam_step")
        self.native_ahead = True
        return loss, logits


class _TrainForward2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, engine, *params):
        ctx.engine, ctx.x = engine, x
        return engine.forward(x, True)

    @staticmethod
    def backward(ctx, dlogits):
        return (None, None, *ctx.engine.backward(dlogits.contiguous().to(torch.float32)))


class _BottleneckParams(nn.Module):
    def __init__(self, cin, width, stride):
        super().__init__()
        cout = 4 * width
        self.conv1 = nn.Conv2d(cin, width, 1, 1, 0, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, cout, 1, 1, 0, bias=False)
        self.bn3 = nn.BatchNorm2d(cout)
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, 0, bias=False), nn.BatchNorm2d(cout))


_R50_STAGES = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))


class ResNet50HIP(nn.Module):
    """ResNet-50 patch classifier on the bf16 engine.  The torch sub-modules are parameter holders only."""

    def __init__(self, n_classes: int):
        super().__init__()
        self.n_classes = int(n_classes)
        self.compute_dtype = "bf16"
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        cin = 64
        for i, (w, n, s) in enumerate(_R50_STAGES, start=1):
            blocks = [_BottleneckParams(cin, w, s)] + [_BottleneckParams(4 * w, w, 1) for _ in range(n - 1)]
            setattr(self, f"layer{i}", nn.Sequential(*blocks))
            cin = 4 * w
        self.fc = nn.Linear(2048, n_classes)
        for m in self.modules():  # torchvision's ResNet initialisation
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        self._engine = Train2Engine(self, "resnet50", self.n_classes)

    def __del__(self):
        try:
            self._engine.release()
        except Exception:
            pass

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("ResNet50HIP runs on the GPU only: move the input with .to('cuda')")
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != x.shape[3]:
            raise ValueError(f"expected [n, 3, P, P], got {tuple(x.shape)}")
        x = x.detach().to(torch.float32).contiguous()
        if self.training and torch.is_grad_enabled():
            self._engine.pull_parameters()   # a torch optimizer will step the nn.Parameters: they must hold the newest values
            return _TrainForward2.apply(x, self._engine, *self.parameters())
        return self._engine.forward(x, self.training)

    def train_step(self, x, labels, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, group=None, bucket_bytes=DEFAULT_BUCKET_BYTES):
        """Fused step in HIP: forward (batch-statistic BN), CrossEntropy(mean), backward, [bucketed all-reduce overlapped
        with the backward kernels], Adam.  Returns (loss tensor on device, logits)."""
        if not self.training:
            raise RuntimeError("train_step needs .train() mode")
        return self._engine.train_step(x, labels, lr, betas, eps, group, bucket_bytes)

    def flat_gradients(self, device) -> torch.Tensor:
        return self._engine.flat(1, device)

    def pull_parameters(self):
        self._engine.pull_parameters()
        return self

    def state_dict(self, *args, **kwargs):
        self._engine.pull_parameters()
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self._engine.native_ahead, self._engine._stats_pending = False, 0   # the loaded tensors win over the library's copies
        return super().load_state_dict(*args, **kwargs)
