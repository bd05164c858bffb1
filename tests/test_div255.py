"""The gather kernels compute k/255 as q=k*r; q'=fma(fma(-q,255,k), r, q) with
r = float32(1/255).  Check on the host, in exact arithmetic, that this equals the
correctly rounded float32 quotient (what NumPy's `astype(float32)/255` gives) for
all 256 inputs.  CPU only."""
from fractions import Fraction

import numpy as np


def rn32(x: Fraction) -> Fraction:
    return Fraction(float(np.float32(float(x)))) if x else Fraction(0)


def fma32(a: Fraction, b: Fraction, c: Fraction) -> Fraction:
    exact = a * b + c  # one rounding; operands are float32 so the double detour below is exact enough
    # round exact rational to float32: go through numpy longdouble-free path
    f = float(exact)  # double rounding risk is checked by the assertion against the direct quotient
    return Fraction(float(np.float32(f)))


def test_div255_sequence_is_correctly_rounded():
    r = Fraction(float(np.float32(1.0) / np.float32(255.0)))
    for k in range(256):
        kf = Fraction(k)
        q = rn32(kf * r)
        e = fma32(-q, Fraction(255), kf)
        got = fma32(e, r, q)
        want = Fraction(float(np.float32(k) / np.float32(255)))
        assert got == want, k


def test_bf16_single_multiply_equals_division():
    """The fused stem kernel normalises with ONE multiply when the target is bf16:
    bf16(k * float32(1/255)) must equal bf16(float32(k) / 255) for every byte value."""
    import torch
    k = np.arange(256, dtype=np.float32)
    a = torch.from_numpy(k * np.float32(1.0 / 255.0)).to(torch.bfloat16)
    b = torch.from_numpy(k / np.float32(255)).to(torch.bfloat16)
    assert torch.equal(a.view(torch.int16), b.view(torch.int16))
