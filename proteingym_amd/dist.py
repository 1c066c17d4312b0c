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
    """Longest-processing-time-first: deterministic, every rank computes the same assignment.
    (heap of (load, rank): ties go to the lower rank; O(n log n) -- the indel pool has ~3e5 items)"""
    import heapq
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    heap = [(0.0, r) for r in range(n_ranks)]
    out: List[List[int]] = [[] for _ in range(n_ranks)]
    for i in order:
        load, r = heapq.heappop(heap)
        out[r].append(i)
        heapq.heappush(heap, (load + costs[i], r))
    for r in out:
        r.sort()
    return out


def plan_position_chunks(seq_lens: Sequence[int], positions: Sequence[np.ndarray], n_ranks: int,
                         chunk_forwards: int = 64, window: int = 1024, **model_dims):
    """Work list of (assay index, positions array) items for sharding INSIDE assays ("chunk = up to B
    positions of one assay at one T"): every assay's masked positions are cut into chunks of <= chunk_forwards
    forwards, the chunks are LPT-balanced over the ranks by algorithmic FLOPs.  Needed when assays are few and
    unequal (config 3: ten assays on eight GPUs); by-assay LPT is enough for the 217-assay benchmark.
    Returns (items, assignment): items[k] = (assay, positions), assignment[r] = item indices of rank r."""
    items, costs = [], []
    for a, (L, pos) in enumerate(zip(seq_lens, positions)):
        pos = np.asarray(pos, dtype=np.int32)
        per = forward_flops(min(L + 2, window), **model_dims)
        for c0 in range(0, len(pos), chunk_forwards):
            chunk = pos[c0:c0 + chunk_forwards]
            items.append((a, chunk))
            costs.append(per * len(chunk))
    return items, lpt_partition(costs, n_ranks)


def merge_tables(parts: Sequence[np.ndarray]) -> np.ndarray:
    """Rows of a log-prob table computed by different ranks (NaN = not computed here) -> one table."""
    out = np.array(parts[0], dtype=np.float32, copy=True)
    for p in parts[1:]:
        fill = np.isnan(out[:, 0]) & ~np.isnan(p[:, 0])
        out[fill] = p[fill]
    return out


def gather_tables(local: Dict[int, np.ndarray], n_toks: Sequence[int], vocab: int = 33, device=None) -> Dict[int, np.ndarray]:
    """Every rank passes {assay -> partial table [n_tok, vocab] with NaN rows it did not compute} for ALL assays
    (all-NaN where it had no chunk); one all_gather of the concatenated tables (sum(n_tok) * vocab floats per rank:
    11 MB for the whole 217-assay benchmark), then a NaN-merge.  Every rank returns the complete tables."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    order = sorted(range(len(n_toks)))
    if not dist.is_initialized():          # (an initialised one-rank group still goes through the collective: tests/test_gpu_dist.py)
        return {a: np.asarray(local[a], dtype=np.float32) for a in order}
    flat = np.concatenate([np.asarray(local[a], dtype=np.float32).ravel() for a in order])
    buf = torch.from_numpy(flat).to(device) if device else torch.from_numpy(flat)
    out = torch.empty(world * flat.size, dtype=torch.float32, device=device)
    dist.all_gather_into_tensor(out, buf)
    host = out.cpu().numpy().reshape(world, flat.size)
    res, o = {}, 0
    for a in order:
        n = int(n_toks[a]) * vocab
        res[a] = merge_tables([host[r, o:o + n].reshape(int(n_toks[a]), vocab) for r in range(world)])
        o += n
    return res


def gather_score_vectors(local: Dict[int, np.ndarray], sizes: Sequence[int], assignment: List[List[int]],
                         device=None) -> Dict[int, np.ndarray]:
    """All ranks call this with their {item index -> float64 scores}.  One fixed-stride
    all_gather (stride = the largest per-rank payload); every rank returns the full
    {item -> scores} map.  ``sizes[i]`` is the score-vector length of item i (known to all)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if not dist.is_initialized():
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
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC handles: what RCCL needs between the ranks of a node on this driver
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


def collective_identity(device=None):
    """What the N > 1 bench line carries so that it proves which collective library saw how many ranks on which devices:
    backend and world size as ``torch.distributed`` reports them, the library version, and one identity string per rank
    (local device index, name, PCI bus id or UUID where this torch build exposes them) moved through the backend itself with
    one fixed-stride all_gather -- ``world_size`` strings that differ pairwise mean ``world_size`` distinct devices took part."""
    import torch
    import torch.distributed as tdist
    if not (tdist.is_available() and tdist.is_initialized()):
        return {"backend": None, "world_size": 1, "devices": [], "version": None}
    backend, world = tdist.get_backend(), tdist.get_world_size()
    on_gpu = backend == "nccl" and torch.cuda.is_available()
    ident = f"host-rank{tdist.get_rank()}"
    if torch.cuda.is_available():
        idx = torch.cuda.current_device() if device is None else int(device)
        prop = torch.cuda.get_device_properties(idx)
        parts = [f"cuda:{idx}", str(prop.name)]
        for attr in ("pci_domain_id", "pci_bus_id", "pci_device_id"):
            if hasattr(prop, attr):
                parts.append(f"{attr}={getattr(prop, attr)}")
        if hasattr(prop, "uuid"):
            parts.append(f"uuid={prop.uuid}")
        ident = " ".join(parts)
    raw = ident.encode("utf-8")[:127]
    mine = torch.zeros(128, dtype=torch.uint8)
    mine[:len(raw)] = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
    dev = "cuda" if on_gpu else "cpu"
    allv = torch.zeros(world * 128, dtype=torch.uint8, device=dev)
    tdist.all_gather_into_tensor(allv, mine.to(dev))
    rows = allv.cpu().numpy().reshape(world, 128)
    devices = [bytes(r[:int((r != 0).sum())]).decode("utf-8", "replace") for r in rows]
    version = None
    if backend == "nccl" and hasattr(torch.cuda, "nccl"):
        try:
            version = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:                                    # noqa: BLE001 -- a missing version string must not take a bench line down
            version = None
    return {"backend": backend, "world_size": world, "devices": devices, "distinct_devices": len(set(devices)), "version": version,
            "library": "RCCL (torch.distributed backend 'nccl' on ROCm)" if backend == "nccl" else backend}
