"""Dense sampling of one slide -- the caller of examples/sample_full_dense.py (BASELINE configs[0]).

The reference's script builds `FullImageDenseSampler(img_path, layer=2, patch_size=224, batch_size=16, stride=112,
mode=INMEMORY_SINGLEPROC)` on a hard-coded `.psi` path and prints the shapes `generator_torch()` yields
(sample_full_dense.py:14-28).  Same loop here; the slide is a path (psimage when installed, `.npy`), or -- the
default -- the closed-form synthetic 4096 x 4096 slide of BASELINE configs[0], generated in HBM.  Prints the
patches/s of the iteration at the end (the reference's other examples print `items/s` the same way).

    python -m deephisto_amd.examples.sample_full_dense [--slide PATH] [--side 4096] [--patch 256] [--stride 256] [--batch 64]
"""
from __future__ import annotations

import argparse
import time


def main(argv=None):
    import torch

    from .. import tiles
    from ..patch_samplers.full_samplers import FullImageDenseSampler, SamplerExecutionMode

    ap = argparse.ArgumentParser()
    ap.add_argument("--slide", default=None, help="slide path (.psi with psimage installed, or .npy); default: synthetic")
    ap.add_argument("--side", type=int, default=4096)
    ap.add_argument("--layer", type=int, default=1)
    ap.add_argument("--patch", type=int, default=256)
    ap.add_argument("--stride", type=int, default=256)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--quiet", action="store_true")
    args = ap.parse_args(argv)
    src = args.slide if args.slide is not None else tiles.synth_slide(args.side, args.side, 0, "cuda")
    patch_sampler = FullImageDenseSampler(src, layer=args.layer, patch_size=args.patch, batch_size=args.batch, stride=args.stride,
                                          mode=SamplerExecutionMode.INMEMORY_SINGLEPROC)
    n, t0 = 0, time.time()
    for inputs, coords, filled_ratio in patch_sampler.generator_torch():
        if not args.quiet:
            print(inputs.shape, coords.shape, filled_ratio)
        n += int(inputs.shape[0])
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f"{n / dt} items/s")
    return n, dt


if __name__ == "__main__":
    main()
