"""The multi-GPU entry points' process wiring, on CPU: `train.main()` under
`python -m torch.distributed.run --nproc-per-node 2` with the gloo backend and a stub model (the HIP model needs a GPU;
tests/test_gpu_cli.py runs the real one).  Reference entry: models/patch_cls_simple/train.py:304-315 (single process)."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[1]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_train_main_two_gloo_ranks(tmp_path):
    env = dict(os.environ, DH_DIST_BACKEND="gloo", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(REPO / "tests" / "helpers" / "ddp_cli_stub.py"), str(tmp_path)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    reps = [json.loads((tmp_path / f"report_{k}.json").read_text()) for k in range(2)]
    for k, rep in enumerate(reps):
        assert rep["rank_env"] == k and rep["sampler_seed"] == k          # every rank draws its own stream of patches
        assert not rep["initialized_before_main"] and rep["group_up_in_train_step"]
        assert not rep["initialized_after_main"]                           # destroy_process_group at exit
    assert reps[0]["initial_weights"] != reps[1]["initial_weights"]        # per-rank initialisation ...
    assert reps[0]["weights"] == reps[1]["weights"]                        # ... rank 0's is broadcast: replicas identical bit for bit
    assert reps[0]["history"] == reps[1]["history"]                        # metrics averaged: same LR / checkpoint decisions
    assert len(reps[0]["history"]["val_loss"]) == 2
    sd = torch.load(tmp_path / "best_model.pth", weights_only=True)        # written by rank 0 only
    assert set(sd) == {"fc.weight", "fc.bias"}
    assert "Using device" in r.stdout and r.stdout.count("Using device") == 1   # rank 0 reports


def test_failing_rank_exits_without_a_barrier(tmp_path):
    """a rank leaving through an exception must not enter a barrier its peers will never match -- the job ends
    non-zero with the original exception on stderr, long before the backend's collective timeout (minutes)."""
    import time
    env = dict(os.environ, DH_DIST_BACKEND="gloo", OMP_NUM_THREADS="1", DH_STUB_FAIL_RANK="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(REPO / "tests" / "helpers" / "ddp_cli_stub.py"), str(tmp_path)]
    t0 = time.time()
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode != 0
    assert "stub sampler failure on one rank" in r.stderr
    assert time.time() - t0 < 120


def test_init_from_env_single_process_is_a_noop(monkeypatch):
    from deephisto_amd.distributed import finalize, init_from_env
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    rank, world, _dev, owned = init_from_env()
    assert (rank, world, owned) == (0, 1, False)
    finalize(owned)
