"""Configs 4 and 5 of BASELINE.json AT WORKLOAD SCALE, as bounded projections for bench.py's `secondary` object.

The bench line measures Tranception on one BLAT-shaped assay and pseudo-ppl on six 735-residue members; the workloads the
north star names are the whole tables:
  * config 4: Tranception-L over the 217 substitution assays (reference_files/DMS_substitutions.csv: 2 465 767 mutants, 37 .. 3 423
    residues, 72 % multi-mutants; SURVEY 8f: ~2 700 PFLOP for the reference's loop);
  * config 5: ESM2 pseudo-ppl over the 66 indel assays (reference_files/DMS_indels.csv: 287 207 mutants, one masked forward per
    (mutant, residue): ~1.95e8 forwards, ~2e5 PFLOP).
Neither fits a bench run.  What does: a STRATIFIED SAMPLE -- one synthetic assay per protein-length bin of the table, its single
and its multi-mutant rows scored separately (prefix sharing saves very different amounts on the two) -- and the measured seconds
per unit of planned cost, applied to every row of the real table with the product planners' own cost functions
(run_sharded.chunk_cost / plan_mutant_chunks, run_indels.sequence_cost / partition_pool).  The projection says what it is: seconds
of scoring with the model resident, no file I/O, synthetic sequences of the real shapes.

The arithmetic (`project_*`, `*_sample`) is plain python and tested on the CPU (tests/test_host_logic.py); `measure_*` need a GPU.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proteingym_amd import dist as pdist, run_indels, run_sharded, synthetic  # noqa: E402

LENGTH_BINS = [(0, 100), (100, 200), (200, 400), (400, 1023), (1023, 10 ** 9)]       # residues; the last bin: proteins beyond the 1 022-residue context
TR_DIMS = dict(layers=36, D=1280, F=5120, V=25)                                       # Tranception-L


def bin_of(seq_len: int) -> int:
    return next(k for k, (lo, hi) in enumerate(LENGTH_BINS) if lo <= seq_len < hi)


def bin_label(k: int) -> str:
    lo, hi = LENGTH_BINS[k]
    return f"{lo}-{hi - 1}" if hi < 10 ** 9 else f">={lo}"


# ---- config 4 -------------------------------------------------------------------------------------------------------------------
def tranception_cost(seq_len: int, rows: int) -> float:
    """The product planner's price of `rows` mutants of a protein (tokens forwarded, attention share included)."""
    return run_sharded.chunk_cost(min(int(seq_len) + 2, 1024), int(rows))


def tranception_flops(seq_len: int, rows: int) -> float:
    """Algorithmic FLOPs of the REFERENCE's loop for `rows` mutants: every sequence in full, both reading directions."""
    T = min(int(seq_len) + 2, 1024)
    return 2.0 * rows * pdist.forward_flops(T, **TR_DIMS)


def tranception_sample(shapes, rows_per_kind: int = 384):
    """One representative per length bin: the assay that carries the most planned cost of its bin, with at most `rows_per_kind`
    single and `rows_per_kind` multi-mutant rows (bins without multi-mutants sample singles only)."""
    out = []
    for k in range(len(LENGTH_BINS)):
        members = [s for s in shapes if bin_of(s["seq_len"]) == k]
        if not members:
            continue
        rep = max(members, key=lambda s: tranception_cost(s["seq_len"], s["n_total"]))
        any_multi = any(s["n_multi"] > 0 for s in members)
        out.append(dict(bin=k, residues=bin_label(k), DMS_id=rep["DMS_id"], seq_len=rep["seq_len"], seed=rep["DMS_index"],
                        singles=min(rows_per_kind, max(1, max(s["n_single"] for s in members))), multis=rows_per_kind if any_multi else 0))
    return out


def project_tranception(shapes, unit_seconds, worlds=(1, 8)):
    """unit_seconds[(bin, 'single' | 'multi')] = measured seconds per unit of tranception_cost.  Returns the projected seconds of the
    whole table on one GPU, per bin, and on `world` GPUs through the product's mutant-chunk planner (max over ranks of the planned
    chunk costs, priced per bin)."""
    per_assay, by_bin = [], {}
    for s in shapes:
        k = bin_of(s["seq_len"])
        us, um = unit_seconds[(k, "single")], unit_seconds.get((k, "multi"), unit_seconds[(k, "single")])
        t = tranception_cost(s["seq_len"], s["n_single"]) * us + tranception_cost(s["seq_len"], s["n_multi"]) * um if s["n_multi"] else \
            tranception_cost(s["seq_len"], s["n_single"]) * us
        per_assay.append(t)
        b = by_bin.setdefault(bin_label(k), dict(assays=0, mutants=0, seconds=0.0))
        b["assays"] += 1
        b["mutants"] += s["n_total"]
        b["seconds"] += t
    total = float(sum(per_assay))
    out = dict(seconds_1_gpu=total, mutants=int(sum(s["n_total"] for s in shapes)), by_length_bin=by_bin,
               mutants_per_s_1_gpu=sum(s["n_total"] for s in shapes) / max(total, 1e-12))
    for w in worlds:
        if w == 1:
            continue
        items, assignment, costs = run_sharded.plan_mutant_chunks([s["seq_len"] for s in shapes], [s["n_total"] for s in shapes], w)
        # a chunk's seconds: its share of its assay's projected seconds (chunks of one assay are row ranges of equal kind mix)
        sec = [per_assay[k] * (costs[j] / max(tranception_cost(shapes[k]["seq_len"], shapes[k]["n_total"]), 1e-30)) for j, (k, _, _) in enumerate(items)]
        loads = np.array([sum(sec[j] for j in part) for part in assignment])
        out[f"seconds_{w}_gpus_planned"] = float(loads.max())
        out[f"planned_load_max_over_mean_{w}_gpus"] = float(loads.max() / max(loads.mean(), 1e-30))
        out[f"chunks_{w}_gpus"] = len(items)
    return out


def measure_tranception(model, sample, ptr):
    """Scores every sample entry's single rows and multi rows (both directions, prefix-shared with intermediate roots: the product's
    defaults) and returns (unit_seconds, per-entry details)."""
    import pandas as pd
    import bench_217
    unit, detail = {}, []
    for e in sample:
        rng = np.random.default_rng(e["seed"])
        seq, df = bench_217.make_assay(rng, e["seq_len"], e["singles"], e["multis"])
        df = df[["mutant"]].copy()
        df["mutated_sequence"] = ptr.mutated_sequences(seq, df["mutant"])
        df = df.drop_duplicates("mutated_sequence")
        is_multi = df["mutant"].str.contains(":")
        row = dict(residues=e["residues"], DMS_id=e["DMS_id"], seq_len=e["seq_len"])
        for kind, part in (("single", df[~is_multi]), ("multi", df[is_multi])):
            if not len(part):
                continue
            model.score_mutants(DMS_data=part.iloc[:8], target_seq=seq, scoring_mirror=True)           # warm-up of this length
            model.rows_forwarded = model.rows_full = 0
            t0 = time.perf_counter()
            model.score_mutants(DMS_data=part, target_seq=seq, scoring_mirror=True)
            dt = time.perf_counter() - t0
            unit[(e["bin"], kind)] = dt / tranception_cost(e["seq_len"], len(part))
            row[kind] = dict(mutants=int(len(part)), seconds=round(dt, 3), mutants_per_s=round(len(part) / dt, 1),
                             rows_forwarded=int(model.rows_forwarded), rows_of_the_full_forwards=int(model.rows_full),
                             reference_loop_tflops=round(tranception_flops(e["seq_len"], len(part)) / dt / 1e12, 1))
        detail.append(row)
    return unit, detail


# ---- config 5 -------------------------------------------------------------------------------------------------------------------
def indel_forwards(shapes):
    """(masked forwards, algorithmic FLOPs) of compute_pppl over the table: a mutant of L residues costs L - 2 forwards of L + 2 tokens
    (the library members are within a few residues of their wild type: priced at its length)."""
    fw = sum(max(0, s["seq_len"] - 2) * s["n_total"] for s in shapes)
    fl = sum(run_indels.sequence_cost(s["seq_len"]) * s["n_total"] for s in shapes)
    return int(fw), float(fl)


def indel_sample_lengths(shapes, n: int = 4):
    """Lengths at which the forward rate is measured: the cost-weighted quantiles of the table's sequence lengths + its extremes."""
    L = np.array([s["seq_len"] for s in shapes], dtype=np.float64)
    w = np.array([run_indels.sequence_cost(int(s["seq_len"])) * s["n_total"] for s in shapes])
    order = np.argsort(L)
    cw = np.cumsum(w[order]) / w.sum()
    picks = {int(L.min()), int(L.max())}
    for q in np.linspace(0.0, 1.0, n + 2)[1:-1]:
        picks.add(int(L[order][min(int(np.searchsorted(cw, q)), len(L) - 1)]))
    by_count = np.argsort(-np.array([s["n_total"] for s in shapes]))[:2]           # and where most of the mutants are
    picks.update(int(shapes[k]["seq_len"]) for k in by_count)
    picks.update(int(v) for v in np.quantile(L, [0.25, 0.5, 0.75], method="nearest"))   # one assay dominates the cost: the plain quartiles too
    return sorted(picks)


def project_indels(shapes, seconds_per_forward, worlds=(1, 8)):
    """seconds_per_forward: {length: measured seconds per masked forward of a library of that length}.  Other lengths are interpolated
    linearly in the algorithmic FLOPs of a forward (the quantity the time follows), clamped at the measured ends."""
    Ls = sorted(seconds_per_forward)
    x = np.array([pdist.forward_flops(L + 2) for L in Ls])
    y = np.array([seconds_per_forward[L] for L in Ls])
    per_assay = []
    for s in shapes:
        spf = float(np.interp(pdist.forward_flops(s["seq_len"] + 2), x, y))
        per_assay.append(max(0, s["seq_len"] - 2) * s["n_total"] * spf)
    total = float(sum(per_assay))
    fw, fl = indel_forwards(shapes)
    big = max(range(len(shapes)), key=lambda k: per_assay[k])
    out = dict(seconds_1_gpu=total, days_1_gpu=total / 86400.0, masked_forwards=fw, algorithmic_pflop=fl / 1e15,
               mutants=int(sum(s["n_total"] for s in shapes)), mutants_per_s_1_gpu=sum(s["n_total"] for s in shapes) / max(total, 1e-12),
               largest_assay=dict(DMS_id=shapes[big]["DMS_id"], share_of_seconds=per_assay[big] / max(total, 1e-12)))
    for w in worlds:
        if w == 1:
            continue
        # the product shards POOLED sequences (run_indels.partition_pool); with ~3e5 sequences the LPT load is flat -- computed on
        # a 1-in-16 thinning of every assay's rows (same length mix) to keep this a millisecond job
        thin = [s["seq_len"] for s in shapes for _ in range(max(1, s["n_total"] // 16))]
        _, loads = run_indels.partition_pool(thin, w)
        ratio = float(loads.max() / max(loads.mean(), 1e-30))
        out[f"seconds_{w}_gpus_planned"] = total / w * ratio
        out[f"hours_{w}_gpus_planned"] = total / w * ratio / 3600.0
        out[f"planned_load_max_over_mean_{w}_gpus"] = ratio
    return out


def measure_indels(model, lengths, pesm, budget_rows: int = 1500):
    """Masked forwards per second of pseudo-ppl libraries at the given lengths (members within +-3 residues, a few members each:
    ~budget_rows forwards per length, one untimed call first)."""
    spf, detail = {}, []
    for L in lengths:
        n = max(2, min(64, budget_rows // max(L - 2, 1)))
        lib = pesm.SequenceLibrary(model, synthetic.random_indel_library(7 + L, L, n)[1])
        lib.score(first=0, count=1)
        t0 = time.perf_counter()
        lib.score()
        dt = time.perf_counter() - t0
        st = lib.stats()
        lib.close()
        spf[L] = dt / max(st["rows"], 1)
        detail.append(dict(residues=L, members=n, masked_forwards=int(st["rows"]), seconds=round(dt, 3), forwards_per_s=round(st["rows"] / dt, 1),
                           packing_efficiency=round(st["packing_efficiency"], 4),
                           algorithmic_tflops=round(st["rows"] * pdist.forward_flops(L + 2) / dt / 1e12, 1)))
    return spf, detail
