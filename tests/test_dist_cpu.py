"""The N>1 path on CPU: world_size-2 gloo processes run the same partition + fixed-stride
all_gather the GPU ranks run over RCCL."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from proteingym_amd import dist as pdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_scores(item, n):
    return np.arange(n, dtype=np.float64) * 1e-3 + item * 1000.0


def _worker(rank, world, port, sizes, costs, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    r, lr, w = pdist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    assignment = pdist.lpt_partition(costs, world)
    local = {i: _fake_scores(i, sizes[i]) for i in assignment[rank]}
    allv = pdist.gather_score_vectors(local, sizes, assignment, device="cpu")
    ok = sorted(allv) == list(range(len(sizes))) and all(
        np.array_equal(allv[i], _fake_scores(i, sizes[i])) for i in range(len(sizes)))
    q.put((rank, ok, [len(a) for a in assignment]))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_partition_and_gather():
    sizes = [5, 0, 17, 3, 1000, 42, 7]          # includes an empty assay
    costs = [3.0, 0.1, 9.0, 1.0, 50.0, 8.0, 2.0]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, sizes, costs, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert res[0][2] == res[1][2]               # identical assignment on every rank


def test_single_process_gather_is_identity():
    local = {0: np.arange(4.0), 1: np.zeros(0)}
    out = pdist.gather_score_vectors(local, [4, 0], [[0, 1]])
    assert np.array_equal(out[0], np.arange(4.0)) and out[1].size == 0
