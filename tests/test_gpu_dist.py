"""The collectives of the N > 1 path on the GPU: a ONE-rank `nccl` (= RCCL) process group on cuda:0 runs the same
``gather_score_vectors`` / ``gather_tables`` calls the 8-GPU ranks make, on device buffers -- so that RCCL, its device-buffer
handling and this torch build's nccl backend have executed at least once before the driver's 8-GPU node sees them
(the multi-rank logic itself is covered by the gloo tests in tests/test_dist_cpu.py).
In a spawned process: the pytest process keeps no process group."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _worker(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from proteingym_amd import dist as pdist
    try:
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        assert dist.get_backend() == "nccl"
        sizes = [5, 0, 17, 1000]
        local = {i: np.arange(n, dtype=np.float64) * 1e-3 + i for i, n in enumerate(sizes)}
        got = pdist.gather_score_vectors(local, sizes, [[0, 1, 2, 3]], device="cuda")
        ok = sorted(got) == [0, 1, 2, 3] and all(np.array_equal(got[i], local[i]) for i in got)
        n_toks = [40, 7]
        tabs = {a: np.random.default_rng(a).standard_normal((n, 33)).astype(np.float32) for a, n in enumerate(n_toks)}
        tabs[0][3] = np.nan
        merged = pdist.gather_tables(tabs, n_toks, device="cuda")
        ok = ok and all(np.array_equal(merged[a], tabs[a], equal_nan=True) for a in tabs)
        # the bench's own collective shape: one fixed-stride all_gather of a float64 device buffer + a MAX all_reduce
        buf = torch.arange(1 << 20, dtype=torch.float64, device="cuda")
        out = torch.empty_like(buf)
        dist.all_gather_into_tensor(out, buf)
        t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ok = ok and bool(torch.equal(out, buf)) and float(t.item()) == 1.5
        # the N > 1 bench line's `rccl` object: backend, world size, library version and the per-rank device identities
        ident = pdist.collective_identity(0)
        ok = ok and ident["backend"] == "nccl" and ident["world_size"] == 1 and len(ident["devices"]) == 1
        ok = ok and ident["devices"][0].startswith("cuda:0 ") and ident["distinct_devices"] == 1 and bool(ident["version"])
        dist.barrier()
        dist.destroy_process_group()
        q.put(("ok" if ok else "mismatch", torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else None))
    except Exception as e:                                   # report instead of hanging the parent on an empty queue
        q.put((f"{type(e).__name__}: {e}", None))


def test_one_rank_nccl_group_runs_the_product_collectives_on_device_buffers(lib):
    import torch
    import torch.multiprocessing as mp
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(port, q))
    p.start()
    status, version = q.get(timeout=300)
    p.join(timeout=60)
    print("RCCL (torch nccl backend) version:", version)
    assert status == "ok", status
    assert p.exitcode == 0
