"""The N>1 path on CPU: world_size-2 gloo processes run the same partition + fixed-stride
all_gather the GPU ranks run over RCCL."""
import os
import socket

import numpy as np
import pytest
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


def _sharded_worker(rank, world, port, mapping_csv, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    from proteingym_amd import run_sharded
    mine = run_sharded.main(["tranception", "--dry-run", "--backend", "gloo", "--", "--checkpoint", "x",
                             "--DMS_reference_file_path", mapping_csv, "--DMS_data_folder", "."])
    q.put((rank, mine))


def test_run_sharded_world2_covers_every_assay_once(tmp_path):
    """Tranception / MSA Transformer multi-GPU runner: assays are LPT-sharded over the ranks, no collective on
    the data path; two gloo ranks must take disjoint sets that cover the reference file and balance the cost."""
    import pandas as pd
    from proteingym_amd import run_sharded
    rng = np.random.default_rng(0)
    rows = [{"DMS_id": f"A{i}", "DMS_filename": f"A{i}.csv", "target_seq": "M" * int(rng.integers(40, 1500)),
             "DMS_total_number_mutants": int(rng.integers(100, 20000))} for i in range(11)]
    csv = str(tmp_path / "map.csv")
    pd.DataFrame(rows).to_csv(csv, index=False)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, csv, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res[0] + res[1]) == list(range(11)) and not set(res[0]) & set(res[1])
    mapping = pd.read_csv(csv)
    cost = [sum(run_sharded.assay_cost("tranception", mapping.iloc[i]) for i in res[r]) for r in range(2)]
    assert max(cost) / sum(cost) < 0.6
    with pytest.raises(SystemExit):
        run_sharded.main(["tranception", "--", "--DMS_index", "3", "--DMS_reference_file_path", csv])


class _FakeTables:
    """Stands in for the device: row p of an assay's table is a deterministic function of (sequence, p)."""

    def __init__(self, location):
        self.salt = sum(map(ord, location))

    def table_rows(self, seq, positions, offset):
        return np.stack([self.row(seq, int(p)) for p in positions])

    def row(self, seq, p):
        rng = np.random.default_rng(self.salt * 100003 + len(seq) * 1009 + p)
        x = rng.standard_normal(33).astype(np.float32)
        return (x - np.log(np.exp(x).sum())).astype(np.float32)


def _position_worker(rank, world, port, workdir, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    import pandas as pd
    from proteingym_amd import run_benchmark as rb
    args = rb.create_parser().parse_args([
        "--model-location", "ckA.pt", "ckB.pt", "--model_type", "ESM1v", "--dms_mapping", os.path.join(workdir, "map.csv"),
        "--dms-input", workdir, "--dms-output", os.path.join(workdir, "out"), "--backend", "gloo", "--shard", "positions",
        "--chunk-forwards", "5"])
    r, lr, w = pdist.init_from_env("gloo")
    mapping = pd.read_csv(args.dms_mapping)
    os.makedirs(args.dms_output, exist_ok=True)
    cols, ens = rb.column_names(args.model_location, args.model_type)
    rb.main_position_shards(args, mapping, list(range(len(mapping))), cols, ens, r, lr, w, make_model=_FakeTables)
    q.put(rank)


def test_position_shards_world2_match_unsharded_scores(tmp_path):
    """--shard positions: two gloo ranks each fill the table rows of their position chunks; after the table
    all_gather + NaN-merge rank 0's CSVs equal the scores computed from the complete tables (bit-identical,
    incl. the ensemble mean and multi-mutants whose rows came from different ranks)."""
    import pandas as pd
    from proteingym_amd import esm as pesm, synthetic
    rows = []
    assays = {}
    for k, L in enumerate((23, 61, 40)):
        seq, muts, score = synthetic.random_assay(seed=k, L=L, n_single=30, n_multi=12)
        pd.DataFrame({"mutant": muts, "DMS_score": score}).to_csv(tmp_path / f"A{k}.csv", index=False)
        rows.append({"DMS_id": f"A{k}", "DMS_filename": f"A{k}.csv", "target_seq": seq})
        assays[f"A{k}"] = (seq, muts)
    pd.DataFrame(rows).to_csv(tmp_path / "map.csv", index=False)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_position_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    done = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert done == [0, 1]
    for name, (seq, muts) in assays.items():
        got = pd.read_csv(tmp_path / "out" / f"{name}.csv", float_precision="round_trip")
        cols = {}
        for ck in ("ckA", "ckB"):
            fake = _FakeTables(ck + ".pt")
            table = np.full((len(seq) + 2, 33), np.nan, dtype=np.float32)
            for p in pesm.positions_read(muts, seq, 1):
                table[p] = fake.row(seq, int(p))
            cols[ck] = pesm.score_from_table(table, muts, seq, 1)
            assert np.array_equal(got[ck].to_numpy(), cols[ck])
        assert np.array_equal(got["Ensemble_ESM1v"].to_numpy(), (cols["ckA"] + cols["ckB"]) / 2)


class _FakeScorer:
    def __init__(self, location):
        self.salt = sum(map(ord, location))

    def score(self, seq, mutants, offset):
        return np.array([((self.salt * 31 + len(seq) * 7 + sum(map(ord, m))) % 1000) / 37.0 for m in mutants])

    def close(self):
        pass


def _assay_worker(rank, world, port, workdir, write, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    from proteingym_amd import run_benchmark as rb
    args = rb.create_parser().parse_args([
        "--model-location", "ckA.pt", "ckB.pt", "--model_type", "ESM1v", "--dms_mapping", os.path.join(workdir, "map.csv"),
        "--dms-input", workdir, "--dms-output", os.path.join(workdir, "out_" + write), "--backend", "gloo", "--write", write])
    rb.main(args, make_model=_FakeScorer)
    q.put(rank)


@pytest.mark.parametrize("write", ["owner", "rank0"])
def test_assay_shards_world2_owner_and_rank0_writers(tmp_path, write):
    """--shard assay on two gloo ranks: LPT split, all_gather of the score vectors; with --write owner every rank writes
    its own assays' CSVs (rank 0 adds scores_summary.csv from the gathered vectors), with --write rank0 rank 0 writes
    all of them.  Either way every CSV holds both checkpoint columns and their plain mean."""
    import pandas as pd
    from proteingym_amd import synthetic
    rows, assays = [], {}
    for k, L in enumerate((30, 90, 55, 41, 120)):
        seq, muts, score = synthetic.random_assay(seed=10 + k, L=L, n_single=20 + k, n_multi=5)
        pd.DataFrame({"mutant": muts, "DMS_score": score}).to_csv(tmp_path / f"B{k}.csv", index=False)
        rows.append({"DMS_id": f"B{k}", "DMS_filename": f"B{k}.csv", "target_seq": seq})
        assays[f"B{k}"] = (seq, muts)
    pd.DataFrame(rows).to_csv(tmp_path / "map.csv", index=False)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_assay_worker, args=(r, 2, port, str(tmp_path), write, q)) for r in range(2)]
    for p in procs:
        p.start()
    assert sorted(q.get(timeout=180) for _ in procs) == [0, 1]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    out = tmp_path / ("out_" + write)
    for name, (seq, muts) in assays.items():
        got = pd.read_csv(out / f"{name}.csv", float_precision="round_trip")
        a, b = _FakeScorer("ckA.pt").score(seq, muts, 1), _FakeScorer("ckB.pt").score(seq, muts, 1)
        assert np.array_equal(got["ckA"].to_numpy(), a) and np.array_equal(got["ckB"].to_numpy(), b)
        assert np.array_equal(got["Ensemble_ESM1v"].to_numpy(), (a + b) / 2)
        assert list(got["mutant"]) == muts
    summary = pd.read_csv(out / "scores_summary.csv")
    assert sorted(summary["DMS_id"]) == sorted(assays) and int(summary["mutants"].sum()) == sum(len(m) for _, m in assays.values())


# ---- config 5: pseudo-ppl over pooled indel libraries, mutants sharded over the ranks -------------------
class _FakePppl:
    """Stands in for the device library: a deterministic score per (checkpoint, sequence)."""

    def __init__(self, location):
        self.salt = sum(map(ord, location))

    def score(self, sequences):
        return np.array([-((self.salt * 17 + sum(map(ord, s)) * 13 + len(s)) % 100003) / 97.0 for s in sequences])

    def close(self):
        pass


def _indel_worker(rank, world, port, workdir, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    from proteingym_amd import run_indels as ri
    args = ri.create_parser().parse_args([
        "--model-location", "esm2_a.pt", "esm2_b.pt", "--model_type", "ESM2", "--dms_mapping", os.path.join(workdir, "map.csv"),
        "--dms-input", workdir, "--dms-output", os.path.join(workdir, "out"), "--backend", "gloo"])
    ri.main(args, make_model=_FakePppl)
    q.put(rank)


def test_run_indels_world2_pools_mutants_across_assays(tmp_path):
    """Two gloo ranks: the mutated sequences of three indel assays (one of them holding most of the work, like
    CAPSD_AAV2S) form ONE pool that is cost-balanced over the ranks; after the fixed-stride all_gather rank 0 writes
    every assay's CSV with every sequence's score in its own row."""
    import pandas as pd
    from proteingym_amd import run_indels as ri, synthetic
    rng = np.random.default_rng(3)
    rows, truth = [], {}
    for k, (L, n) in enumerate(((60, 25), (735, 400), (220, 40))):
        wt = synthetic.random_sequence(rng, L)
        seqs = []
        for _ in range(n):
            p, d = int(rng.integers(1, L - 4)), int(rng.integers(-3, 4))
            seqs.append(wt[:p] + (synthetic.random_sequence(rng, d) if d > 0 else "") + wt[p - min(d, 0):])
        pd.DataFrame({"mutant": seqs, "mutated_sequence": seqs, "DMS_score": rng.standard_normal(n)}).to_csv(tmp_path / f"I{k}.csv", index=False)
        rows.append({"DMS_id": f"I{k}", "DMS_filename": f"I{k}.csv", "target_seq": wt})
        truth[f"I{k}"] = seqs
    pd.DataFrame(rows).to_csv(tmp_path / "map.csv", index=False)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_indel_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    assert sorted(q.get(timeout=180) for _ in procs) == [0, 1]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for name, seqs in truth.items():
        got = pd.read_csv(tmp_path / "out" / f"{name}.csv", float_precision="round_trip")
        assert list(got.columns) == ["mutant", "mutated_sequence", "DMS_score", "esm2_a", "esm2_b"]     # ESM2: no ensemble column
        for ck in ("esm2_a", "esm2_b"):
            assert np.array_equal(got[ck].to_numpy(), _FakePppl(ck + ".pt").score(seqs))
    # the planner: identical on every rank, every sequence exactly once, FLOP-balanced although one assay dominates
    lengths = [len(s) for seqs in truth.values() for s in seqs]
    for world in (2, 8):
        a, loads = ri.partition_pool(lengths, world)
        assert sorted(k for part in a for k in part) == list(range(len(lengths)))
        assert loads.max() / loads.mean() < 1.02
