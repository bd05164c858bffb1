"""First execution of the RCCL (backend "nccl") code paths on real hardware.

The builder's box has ONE GPU and RCCL refuses two ranks on one device, so the collectives of N > 1 cannot run here; what CAN
run is every RCCL call this package makes, on a one-rank process group, on the very tensors it makes them on: the flat
all-gather of per-tile logits (`exchange_logits`), and the bucketed asynchronous all-reduce of slices of the native library's
gradient arena (memory hipMalloc'ed OUTSIDE torch's allocator, viewed through __cuda_array_interface__, launched from a side
stream behind an event, waited for on the compute stream).  With one rank a sum is the identity, so the results are checked
exactly; the multi-rank arithmetic is covered on gloo (tests/test_ddp.py, tests/test_distributed_gloo.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from deephisto_amd.examples.predict_full_patched import exchange_logits
    from deephisto_amd.models.patch_cls_simple.ddp import BucketReducer, allreduce_mean_
    from deephisto_amd.models.patch_cls_simple.model import ce_loss, get_model
    from deephisto_amd._lib import check, lib
    out = {}
    # 1. the predict exchange
    local = torch.randn(4802, 5, device=dev)
    full = exchange_logits(local, 4802)
    out["gather"] = bool(torch.equal(full, local))
    # 2. bucketed all-reduce of the native gradient arena (bf16 engine) inside train_step's machinery
    torch.manual_seed(0)
    m = get_model(5, arch="resnet50").to(dev).train()
    x = torch.rand(4, 3, 64, 64, device=dev)
    y = torch.randint(0, 5, (4,), device=dev)
    eng = m._engine
    logits = eng.forward(x, True, pull_stats=False)
    _, dl = ce_loss(logits, y, want_grad=True)
    check(lib().dh_train2_backward(eng.handle, dl.data_ptr(), None), "backward")
    want = m.flat_gradients(dev).clone()
    red = eng._arm_overlap(dev, None, 25 * 1024 * 1024)
    check(lib().dh_train2_backward(eng.handle, dl.data_ptr(), None), "backward")      # callbacks fire -> async all_reduce per bucket
    eng._finish_overlap(red)
    torch.cuda.synchronize()
    out["buckets"] = [b for b, _, _ in eng.overlap_log]
    out["arena"] = bool(torch.equal(m.flat_gradients(dev), want))                     # world 1: sum / 1 = identity
    # 3. the un-bucketed helper and the f32 engine's arena
    m18 = get_model(5, "f32").to(dev).train()
    m18.train_step(x, y, lr=1e-4)
    g18 = m18.flat_gradients(dev)
    before = g18.clone()
    allreduce_mean_(g18)
    torch.cuda.synchronize()
    out["arena18"] = bool(torch.equal(g18, before))
    q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_calls_on_one_rank(built_lib):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(_free_port(), q))
    p.start()
    out = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert out["gather"] and out["arena"] and out["arena18"], out
    assert out["buckets"] == [0, 1, 2, 3]


def _worker_cabi(q):
    """The C-ABI exchange entry the way a host that is not a torch.distributed program calls it: an RCCL communicator of its own."""
    import ctypes as C
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    from deephisto_amd._lib import lib
    out = {}
    rccl = C.CDLL("librccl.so.1")

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    rccl.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    uid = UniqueId()
    out["uid"] = rccl.ncclGetUniqueId(C.byref(uid))
    comm = C.c_void_p()
    out["init"] = rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0)
    send = torch.randn(4802, 5, device=dev)
    recv = torch.zeros_like(send)
    st = torch.cuda.current_stream().cuda_stream
    # a library named in DH_RCCL_LIB that is not loaded in the process is refused (never a second RCCL beside the communicator's own)
    os.environ["DH_RCCL_LIB"] = "/nonexistent/librccl.so.1"
    out["bad_env"] = lib().dh_allgather_logits(comm, send.data_ptr(), recv.data_ptr(), 4802, 5, st)
    out["bad_env_msg"] = lib().dh_last_error().decode()
    del os.environ["DH_RCCL_LIB"]
    # the exact answer: the dlopen handle of the library that made `comm`
    out["set"] = lib().dh_set_rccl(C.c_void_p(rccl._handle))
    out["rc"] = lib().dh_allgather_logits(comm, send.data_ptr(), recv.data_ptr(), 4802, 5, st)
    torch.cuda.synchronize()
    out["equal"] = bool(torch.equal(send, recv))          # one rank: the gathered list is the rank's own
    out["null_comm"] = lib().dh_allgather_logits(None, send.data_ptr(), recv.data_ptr(), 4802, 5, st)
    out["null_ptr"] = lib().dh_allgather_logits(comm, None, recv.data_ptr(), 4802, 5, st)
    out["empty"] = lib().dh_allgather_logits(comm, None, None, 0, 5, st)
    rccl.ncclCommDestroy(comm)
    q.put(out)


def test_c_abi_allgather_on_one_rank(built_lib):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_cabi, args=(q,))
    p.start()
    out = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert out["uid"] == 0 and out["init"] == 0, out
    assert out["bad_env"] == -22 and "not loaded in this process" in out["bad_env_msg"], out
    assert out["set"] == 0 and out["rc"] == 0 and out["equal"], out
    assert out["null_comm"] == -22 and out["null_ptr"] == -22 and out["empty"] == 0, out
