"""Multi-GPU sharding of the scoring work: one process per GPU, assays are independent units,
one exchange step (RCCL all_gather of the per-mutant score vectors; ``nccl`` backend == RCCL over
xGMI on ROCm, ``gloo`` in the CPU tests).

The reference scales out only by launching one process per ``--dms_index`` (SLURM-array style,
scripts/scoring_DMS_zero_shot/scoring_ESM1v_substitutions.sh:21-31); the only in-repo precedent
for "shard sequences, gather scores" is ProGen3 (proteingym/baselines/progen3/scorer.py:44-64,
134-167: rank-strided work list, all_gather_object of (scores, indices), re-order by index).
Here the work list is cost-balanced (longest-processing-time first) because assay cost spans
three orders of magnitude (39 .. 3425 forwards of 39 .. 1024 tokens).
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np


def forward_flops(T: int, layers: int = 33, D: int = 1280, F: int = 5120, V: int = 33) -> float:
    """Algorithmic FLOPs of one masked forward of T tokens:
    layers*(8 T D^2 + 4 T D F + 4 T^2 D) + head on one row."""
    return layers * (8.0 * T * D * D + 4.0 * T * D * F + 4.0 * T * T * D) + 2.0 * D * D + 2.0 * D * V


def assay_cost(seq_len: int, n_positions: int = None, window: int = 1024, **model_dims) -> float:
    n_tok = seq_len + 2
    T = min(n_tok, window)
    P = n_tok if n_positions is None else n_positions
    return P * forward_flops(T, **model_dims)


def lpt_partition(costs: Sequence[float], n_ranks: int) -> List[List[int]]:
    """Longest-processing-time-first: deterministic, every rank computes the same assignment."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0.0] * n_ranks
    out: List[List[int]] = [[] for _ in range(n_ranks)]
    for i in order:
        r = min(range(n_ranks), key=lambda k: (loads[k], k))
        out[r].append(i)
        loads[r] += costs[i]
    for r in out:
        r.sort()
    return out


def gather_score_vectors(local: Dict[int, np.ndarray], sizes: Sequence[int], assignment: List[List[int]],
                         device=None) -> Dict[int, np.ndarray]:
    """All ranks call this with their {item index -> float64 scores}.  One fixed-stride
    all_gather (stride = the largest per-rank payload); every rank returns the full
    {item -> scores} map.  ``sizes[i]`` is the score-vector length of item i (known to all)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if world == 1:
        return {i: np.asarray(v, dtype=np.float64) for i, v in local.items()}
    per_rank = [sum(int(sizes[i]) for i in items) for items in assignment]
    stride = max(max(per_rank), 1)
    buf = torch.zeros(stride, dtype=torch.float64, device=device)
    if assignment[rank]:
        mine = np.concatenate([np.asarray(local[i], dtype=np.float64).ravel() for i in assignment[rank]]) \
            if per_rank[rank] else np.zeros(0)
        buf[: mine.size] = torch.from_numpy(mine).to(buf.device)
    out = torch.empty(world * stride, dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(out, buf)
    host = out.cpu().numpy()
    res: Dict[int, np.ndarray] = {}
    for r, items in enumerate(assignment):
        o = r * stride
        for i in items:
            res[i] = host[o:o + int(sizes[i])].copy()
            o += int(sizes[i])
    return res


def init_from_env(backend: str = None):
    """Initialise torch.distributed from the torchrun environment (RANK/LOCAL_RANK/WORLD_SIZE/
    MASTER_*).  Returns (rank, local_rank, world)."""
    import os
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world
