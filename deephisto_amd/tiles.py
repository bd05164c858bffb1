"""Thin torch-tensor wrappers over the tile-side C ABI (include/deephisto_hip.h).

torch is used only for device memory and streams; every computation below runs
in libdeephisto_hip.so.  No CPU fallbacks: a missing library or GPU raises.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ._lib import (DH_DTYPE_BF16, DH_DTYPE_F32, DH_LAYOUT_NCHW, DH_LAYOUT_NHWC, check, lib)

_TORCH_DTYPE = {DH_DTYPE_F32: torch.float32, DH_DTYPE_BF16: torch.bfloat16}


def _stream(device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _require_cuda(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise ValueError(f"{what} must live in GPU memory (got {t.device})")
    if not t.is_contiguous():
        raise ValueError(f"{what} must be contiguous")


def dtype_code(dtype) -> int:
    if dtype in (torch.float32, "f32", "float32", DH_DTYPE_F32):
        return DH_DTYPE_F32
    if dtype in (torch.bfloat16, "bf16", "bfloat16", DH_DTYPE_BF16):
        return DH_DTYPE_BF16
    raise ValueError(f"unsupported dtype {dtype!r}")


def tile_grid(h: int, w: int, patch: int, stride: int, batch: int) -> tuple[np.ndarray, int]:
    """(int32[n_padded, 2] (y, x) origins in the reference's order, n_unique).

    Host-side integer work done by dh_tile_grid (full_samplers.py:374-404)."""
    nu, npad = C.c_int64(), C.c_int64()
    check(lib().dh_tile_grid_count(h, w, patch, stride, batch, C.byref(nu), C.byref(npad)), "dh_tile_grid_count")
    out = np.empty((npad.value, 2), dtype=np.int32)
    check(lib().dh_tile_grid(h, w, patch, stride, batch, out.ctypes.data_as(C.c_void_p), npad.value), "dh_tile_grid")
    return out, nu.value


def synth_slide(h: int, w: int, seed: int = 0, device="cuda") -> torch.Tensor:
    """uint8[h, w, 3] closed-form synthetic slide generated directly in HBM."""
    dev = torch.device(device)
    out = torch.empty((h, w, 3), dtype=torch.uint8, device=dev)
    check(lib().dh_synth_slide(out.data_ptr(), h, w, seed & 0xFFFFFFFF, _stream(dev)), "dh_synth_slide")
    return out


def gather_tiles(slide: torch.Tensor, origins, patch: int, layout: int = DH_LAYOUT_NCHW,
                 dtype=torch.float32, check_bounds: bool = True) -> torch.Tensor:
    """Gather `patch` x `patch` tiles at int32 (y, x) `origins` from a device-resident
    uint8 HWC slide; returns [n,3,P,P] (NCHW) or [n,P,P,3] (NHWC) values k/255."""
    _require_cuda(slide, "slide")
    if slide.dtype != torch.uint8 or slide.dim() != 3 or slide.shape[2] != 3:
        raise ValueError("slide must be uint8[h, w, 3]")
    h, w = int(slide.shape[0]), int(slide.shape[1])
    host = None
    if isinstance(origins, torch.Tensor):
        yx_dev = origins.to(device=slide.device, dtype=torch.int32).contiguous()
        if check_bounds:
            host = np.ascontiguousarray(origins.detach().cpu().numpy().astype(np.int32))
    else:
        host = np.ascontiguousarray(np.asarray(origins, dtype=np.int32).reshape(-1, 2))
        yx_dev = torch.from_numpy(host).to(slide.device)
    n = int(yx_dev.shape[0])
    code = dtype_code(dtype)
    shape = (n, 3, patch, patch) if layout == DH_LAYOUT_NCHW else (n, patch, patch, 3)
    out = torch.empty(shape, dtype=_TORCH_DTYPE[code], device=slide.device)
    check(lib().dh_tile_gather(slide.data_ptr(), h, w, yx_dev.data_ptr(),
                               host.ctypes.data_as(C.c_void_p) if (check_bounds and host is not None) else None,
                               n, patch, layout, code, out.data_ptr(), _stream(slide.device)), "dh_tile_gather")
    return out


def gather_tiles_aug(slide: torch.Tensor, origins_dev: torch.Tensor, patch: int, layout: int = DH_LAYOUT_NCHW,
                     dtype=torch.float32, flip_h: bool = False, flip_v: bool = False) -> torch.Tensor:
    """gather_tiles with the training pipeline's batch-level flips fused in (dh_tile_gather_aug)."""
    _require_cuda(slide, "slide")
    _require_cuda(origins_dev, "origins")
    n = int(origins_dev.shape[0])
    code = dtype_code(dtype)
    shape = (n, 3, patch, patch) if layout == DH_LAYOUT_NCHW else (n, patch, patch, 3)
    out = torch.empty(shape, dtype=_TORCH_DTYPE[code], device=slide.device)
    check(lib().dh_tile_gather_aug(slide.data_ptr(), int(slide.shape[0]), int(slide.shape[1]), origins_dev.data_ptr(), n,
                                   patch, layout, code, int(flip_h), int(flip_v), out.data_ptr(), _stream(slide.device)),
          "dh_tile_gather_aug")
    return out


class PinnedUploader:
    """Host -> device copies of small per-batch arrays (tile origins, labels) that do not stall the host.

    `torch.from_numpy(a).to(dev)` from pageable memory synchronises the stream: the host cannot prepare batch i+1 while the
    GPU runs step i, and every step starts with an idle GPU.  Here the array is written into one of `depth` pinned staging
    buffers and copied with `non_blocking=True`; an event per slot makes sure a slot is not overwritten before its last
    copy has completed (normally long since)."""

    def __init__(self, device, depth: int = 4):
        self.device, self.depth = torch.device(device), depth
        self._slots: dict = {}   # dtype -> [[pinned byte buffer, event of its last copy] x depth]
        self._next: dict = {}

    def upload(self, arr: "np.ndarray") -> torch.Tensor:
        import numpy as np

        arr = np.ascontiguousarray(arr)
        # one ring per DTYPE, each slot sized to the largest request so far (grown geometrically): arrays whose length changes
        # from batch to batch (per-slide selections of the region samplers) reuse the same pinned memory instead of pinning
        # `depth` new buffers per distinct shape -- pin_memory() is a synchronous hipHostMalloc, the stall this class avoids
        key = arr.dtype.str
        if key not in self._slots:
            self._slots[key] = [[None, None] for _ in range(self.depth)]
            self._next[key] = 0
        k = self._next[key]
        self._next[key] = (k + 1) % self.depth
        slot = self._slots[key][k]
        if slot[1] is not None:
            slot[1].synchronize()
        n = arr.size
        if slot[0] is None or slot[0].numel() < n:
            cap = max(n, 2 * (slot[0].numel() if slot[0] is not None else 0), 64)
            slot[0] = torch.from_numpy(np.empty(cap, dtype=arr.dtype)).pin_memory()
        slot[0].numpy()[:n] = arr.reshape(-1)
        out = slot[0][:n].to(self.device, non_blocking=True).reshape(arr.shape)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        slot[1] = ev
        return out

    def upload_batch(self, origins: "np.ndarray", labels: "np.ndarray") -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """One batch's origins int32[n, 2], labels int64[n] and coordinates float32[n, 2] (= float(origins): exact, |origin| < 2^24,
        what dh_tile_coords_f32 computes) as ONE staged copy: one copy and one event on the stream per batch instead of two copies, two
        events and a kernel.  Returns device views (origins, labels, coords) of the one device buffer."""
        import numpy as np

        n = int(origins.shape[0])
        assert origins.shape == (n, 2) and labels.shape == (n,)
        buf = np.empty(24 * n, np.uint8)
        buf[:8 * n] = np.ascontiguousarray(labels, np.int64).view(np.uint8)                       # 8-byte aligned first
        buf[8 * n:16 * n] = np.ascontiguousarray(origins, np.int32).reshape(-1).view(np.uint8)
        buf[16 * n:] = np.ascontiguousarray(origins, np.float32).reshape(-1).view(np.uint8)
        dev = self.upload(buf)
        return dev[8 * n:16 * n].view(torch.int32).reshape(n, 2), dev[:8 * n].view(torch.int64), dev[16 * n:].view(torch.float32).reshape(n, 2)


def gather_tiles_raw(slide: torch.Tensor, origins_dev: torch.Tensor, patch: int) -> torch.Tensor:
    """float32[n, P, P, 3] of the raw 0..255 pixel values (FullImageRndSampler.generator_torch)."""
    _require_cuda(slide, "slide")
    _require_cuda(origins_dev, "origins")
    n = int(origins_dev.shape[0])
    out = torch.empty((n, patch, patch, 3), dtype=torch.float32, device=slide.device)
    check(lib().dh_tile_gather_raw(slide.data_ptr(), int(slide.shape[0]), int(slide.shape[1]), origins_dev.data_ptr(), n,
                                   patch, out.data_ptr(), _stream(slide.device)), "dh_tile_gather_raw")
    return out


def tile_coords(origins_dev: torch.Tensor) -> torch.Tensor:
    """float32[n, 2] (pos_y, pos_x) from int32 device origins (full_samplers.py:444-451)."""
    _require_cuda(origins_dev, "origins")
    n = int(origins_dev.shape[0])
    out = torch.empty((n, 2), dtype=torch.float32, device=origins_dev.device)
    check(lib().dh_tile_coords_f32(origins_dev.data_ptr(), n, out.data_ptr(), _stream(origins_dev.device)),
          "dh_tile_coords_f32")
    return out


def accumulate_logits(logits: torch.Tensor, origins_host: np.ndarray, patch: int, downscale: int,
                      h: int, w: int, canvas: torch.Tensor | None = None,
                      want_map: bool = True) -> tuple[torch.Tensor, torch.Tensor | None]:
    """Ordered accumulation of per-tile logits into the downscaled canvas and argmax
    (examples/predict_full_patched.py:41-62).  Returns (canvas f32[dh,dw,n], map int64[dh,dw])."""
    _require_cuda(logits, "logits")
    if logits.dtype != torch.float32 or logits.dim() != 2:
        raise ValueError("logits must be float32[n, n_cls]")
    yx = np.ascontiguousarray(np.asarray(origins_host, dtype=np.int32).reshape(-1, 2))
    n, n_cls = int(logits.shape[0]), int(logits.shape[1])
    if yx.shape[0] != n:
        raise ValueError(f"{n} logit rows but {yx.shape[0]} origins")
    dh_, dw_ = h // downscale, w // downscale
    if canvas is None:
        canvas = torch.zeros((dh_, dw_, n_cls), dtype=torch.float32, device=logits.device)
    else:
        _require_cuda(canvas, "canvas")
        if tuple(canvas.shape) != (dh_, dw_, n_cls) or canvas.dtype != torch.float32:
            raise ValueError("canvas must be float32[h//d, w//d, n_cls]")
    cmap = torch.empty((dh_, dw_), dtype=torch.int64, device=logits.device) if want_map else None
    check(lib().dh_accumulate_logits(logits.data_ptr(), yx.ctypes.data_as(C.c_void_p), n, patch, downscale,
                                     n_cls, h, w, canvas.data_ptr(),
                                     cmap.data_ptr() if cmap is not None else None,
                                     _stream(logits.device)), "dh_accumulate_logits")
    return canvas, cmap


def colorize_map(class_map: torch.Tensor, lut: torch.Tensor) -> torch.Tensor:
    """uint8[h, w, 3]: `colored[pred == id] = color` for every class (predict_full_patched.py:89-95);
    `lut` = uint8[n_cls, 3] indexed by class id, ids without an entry stay black."""
    _require_cuda(class_map, "class_map")
    if class_map.dtype != torch.int64 or not class_map.is_contiguous():
        raise ValueError("class_map must be contiguous int64")
    lut = lut.to(device=class_map.device, dtype=torch.uint8).contiguous()
    out = torch.empty(tuple(class_map.shape) + (3,), dtype=torch.uint8, device=class_map.device)
    check(lib().dh_colorize_map(class_map.data_ptr(), class_map.numel(), lut.data_ptr(), int(lut.shape[0]), out.data_ptr(),
                                _stream(class_map.device)), "dh_colorize_map")
    return out


def overlay_blend(img: torch.Tensor, colored: torch.Tensor, alpha: float = 0.6) -> torch.Tensor:
    """`(img * alpha + colored * (1 - alpha)).astype(uint8)` (predict_full_patched.py:108-110), float64 math."""
    _require_cuda(img, "img")
    _require_cuda(colored, "colored")
    if img.dtype != torch.uint8 or colored.dtype != torch.uint8 or img.shape != colored.shape:
        raise ValueError("img and colored must be uint8 tensors of one shape")
    img, colored = img.contiguous(), colored.contiguous()
    out = torch.empty_like(img)
    check(lib().dh_overlay_blend(img.data_ptr(), colored.data_ptr(), img.numel(), float(alpha), out.data_ptr(),
                                 _stream(img.device)), "dh_overlay_blend")
    return out
