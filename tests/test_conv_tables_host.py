"""Host-side schedule tables of the 3x3 convolution kernel under AddressSanitizer + UBSan (CPU only; VERDICT r3 item 6,
SURVEY section 5 "sanitizers" row).

Every LDS-DMA source address and every output address of `conv3x3_kernel` comes out of three host-built integer tables
(`deephisto_amd/csrc/conv3_tables_host.h`, included by the library and by the harness alike).  The harness
`tests/host/conv3_tables_sweep.cpp` builds them for P in {32, 64, 96, 100, 224, 256, 330} x n in {1, 3, 64, 1024, 3842, 4096}
x stride 1 / 2 x all tile candidates x {bf16 channel-blocked, bf16 NHWC, float32 NHWC}, plus (round 4) the wide stride-2 variant on half-chunk stages reading 16-channel planes and the stride-1 convs that write them next to a residual in 32-channel planes, and replays the kernel's address arithmetic:
window pieces inside the tensor and on the right pixel / swizzle slot, in-image pixels never padded, every output pixel x cout
block written exactly once inside the tensor, 32-bit offset words not exceeded (n = 4 096 at P = 256 in bf16 is the 2^31-byte map
the bench's launch size approaches; float32 at that size must be refused)."""
import shutil
import subprocess
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parents[1]


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_conv3_tables_sweep_under_asan_ubsan(tmp_path):
    exe = tmp_path / "conv3_tables_sweep"
    cmd = ["g++", "-std=c++17", "-O2", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           str(REPO / "tests" / "host" / "conv3_tables_sweep.cpp"), "-o", str(exe)]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-4000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-4000:]
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith("OK ") and int(last.split()[1]) >= 1000, r.stdout[-500:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
