"""Strong-scaling leg of bench.py (N > 1; the line's `strong_scaling_217` object, timed after the headline's K steps): ONE pass over the 217-assay-shaped substitution benchmark, sharded over the
ranks exactly as the product runner shards it (run_benchmark.plan_assays: LPT on planned seconds), inputs resident.

north_star: "mutants/sec on synthetic 217-assay-shaped input reported at 1/2/4/8 GPUs".  The table is
proteingym_amd/data/dms_substitutions_shapes.csv (the real seq_len / mutant counts of reference_files/DMS_substitutions.csv:
217 assays, 37 .. 3 423 residues, 2 465 767 mutants, one assay with 536 962 rows, 16 proteins beyond the 1 022-residue
window); sequences, mutants and the ESM-1v-650M-shaped checkpoint are synthetic (generator = scripts/bench_217.py).

Total work is fixed, whatever N is (strong scaling).  Every rank generates and uploads ONLY its own assays before the timed
region (pgmi_assay_create: wild type, positions, flattened substitutions resident in HBM), then the timed region runs the
rank's assays once -- cut into K consecutive "steps" of its work list so that the contract's K-steps clock brackets exactly one
pass -- and ends with the ONE fixed-stride all_gather of the per-mutant score vectors (RCCL over xGMI).  value = all mutants /
max-over-ranks time.  No CSV is written here (the end-to-end run with checkpoint read and CSVs is bench.py's N = 1
`secondary.benchmark_217_end_to_end`, whose assay_run_s is the one-GPU baseline of this curve).
"""
import os
import sys
import time

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proteingym_amd import dist as pdist, run_benchmark, synthetic  # noqa: E402
import bench_217  # noqa: E402


def plan(world: int, max_assays: int = 0):
    """(shapes, assignment): the product planner on the 217-assay table, one checkpoint."""
    shapes = synthetic.dms_shapes()
    if max_assays:
        shapes = shapes[:max_assays]
    mapping = pd.DataFrame({"target_seq": ["M" * s["seq_len"] for s in shapes],
                            "DMS_total_number_mutants": [s["n_total"] for s in shapes]})
    return shapes, run_benchmark.plan_assays(mapping, list(range(len(shapes))), world, 1)


def step_slices(n_items: int, steps: int):
    """K consecutive slices of a rank's work list (some empty when the rank has fewer items than steps)."""
    return [(n_items * k // steps, n_items * (k + 1) // steps) for k in range(steps)]


def run(model, rank: int, world: int, steps: int, warmup: int, torch, tdist, max_assays: int = 0, make_assay=None):
    """Returns the dict bench.py turns into its JSON line (times are max over ranks).  ``make_assay`` is a test seam:
    (model, seq, mutants) -> object with run_device_only(ptr), positions, T, close()."""
    from proteingym_amd import esm as pesm
    shapes, assignment = plan(world, max_assays)
    mine = sorted(assignment[rank], key=lambda i: -shapes[i]["n_total"])
    t0 = time.perf_counter()
    assays, n_rows = [], []
    for i in mine:
        s = shapes[i]
        seq, df = bench_217.make_assay(np.random.default_rng(s["DMS_index"]), s["seq_len"], s["n_single"], s["n_total"] - s["n_single"])
        muts = list(df["mutant"])
        assays.append(make_assay(model, seq, muts) if make_assay is not None else pesm.Assay(model, seq, muts))
        n_rows.append(len(muts))
    setup_s = time.perf_counter() - t0
    per_rank = [sum(shapes[i]["n_total"] for i in part) for part in assignment]
    stride = max(max(per_rank), 1)
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    # collectives run on the device buffers over RCCL; with the gloo backend (CPU tests, and the two-ranks-on-one-GPU rehearsal
    # PGMI_BENCH_SHARE_GPU=1) they are staged through host memory
    cdev = dev if (world == 1 or tdist.get_backend() == "nccl") else "cpu"
    buf = torch.zeros(stride, dtype=torch.float64, device=dev)
    gathered = torch.empty(world * stride, dtype=torch.float64, device=cdev) if world > 1 else None
    offs = np.concatenate([[0], np.cumsum(n_rows)]).astype(np.int64)

    def run_items(a, b):
        for j in range(a, b):
            assays[j].run_device_only(buf.data_ptr() + 8 * int(offs[j]))

    def fence():
        if world > 1:
            tdist.barrier()
        if dev == "cuda":
            torch.cuda.synchronize()

    # warm-up: W untimed steps of the same kind (the first W slices of the pass; the pass below repeats them)
    slices = step_slices(len(assays), steps)
    for k in range(min(warmup, steps)):
        run_items(*slices[k])
    if hasattr(model, "profile_reset"):
        model.profile_reset()
        model.profile_enable(True)
    fence()
    t0 = time.perf_counter()
    for a, b in slices:
        run_items(a, b)
    if world > 1:
        tdist.all_gather_into_tensor(gathered, buf if cdev == dev else buf.to(cdev))   # per-mutant score vectors of every rank, one collective
    fence()
    dt = time.perf_counter() - t0
    busy = dt
    if hasattr(model, "profile_enable"):
        model.profile_enable(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        dt = float(t.item())
        tmin = torch.tensor([busy], dtype=torch.float64, device=cdev)
        tdist.all_reduce(tmin, op=tdist.ReduceOp.MIN)
        busy = float(tmin.item())
    positions = sum(int(len(a.positions)) for a in assays)
    flops = sum(len(a.positions) * pdist.forward_flops(a.T) for a in assays)
    tot = torch.tensor([float(positions), float(flops)], dtype=torch.float64, device=cdev)
    if world > 1:
        tdist.all_reduce(tot)
    for a in assays:
        a.close()
    n_mut = sum(s["n_total"] for s in shapes)
    all_pos = sum(s["seq_len"] + 2 for s in shapes)
    planned = np.array([sum(run_benchmark.assay_seconds(shapes[i]["seq_len"], shapes[i]["n_total"], 1) for i in part) for part in assignment])
    return dict(seconds=dt, fastest_rank_seconds=busy, mutants=n_mut, assays=len(shapes), positions_run=int(tot[0].item()),
                positions_reference_runs=all_pos, executed_algorithmic_flops=float(tot[1].item()),
                executed_tflops_per_gpu=float(tot[1].item()) / dt / 1e12 / world, setup_seconds_not_timed=setup_s,
                planned_load_max_over_mean=float(planned.max() / planned.mean()), assays_per_rank=[len(p) for p in assignment],
                all_gather_bytes_per_rank=int(stride * 8))
