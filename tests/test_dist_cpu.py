"""The N>1 path on CPU: world_size-2 gloo processes run the same partition + fixed-stride
all_gather the GPU ranks run over RCCL."""
import os
import socket
import time

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


def _identity_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    q.put((rank, pdist.collective_identity()))
    dist.barrier()
    dist.destroy_process_group()


def test_collective_identity_reports_backend_world_and_one_string_per_rank():
    """The `rccl` object of the N > 1 bench line: backend / world size from torch.distributed and one identity per rank moved
    through the backend itself (gloo here; tests/test_gpu_dist.py runs it on a one-rank nccl group)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_identity_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == res[1]
    assert res[0]["backend"] == "gloo" and res[0]["world_size"] == 2 and len(res[0]["devices"]) == 2
    assert res[0]["distinct_devices"] == 2 and res[0]["devices"][0].endswith("rank0") and res[0]["devices"][1].endswith("rank1")


def test_single_process_gather_is_identity():
    local = {0: np.arange(4.0), 1: np.zeros(0)}
    out = pdist.gather_score_vectors(local, [4, 0], [[0, 1]])
    assert np.array_equal(out[0], np.arange(4.0)) and out[1].size == 0


def _sharded_worker(rank, world, port, mapping_csv, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    from proteingym_amd import run_sharded
    mine = run_sharded.main(["tranception", "--shard", "assay", "--dry-run", "--backend", "gloo", "--", "--checkpoint", "x",
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


# ---- Tranception: chunks of mutant rows (BASELINE config 4 on N GPUs) -------------------------------------------------
def _fake_tranception(checkpoint, device, scoring_window):
    """Stands in for the device: the log-likelihood of a sliced sequence is a deterministic function of (sequence,
    reading direction, retrieval weight); everything above it (slices, wild-type deltas, mirror average, frames) is
    the product's own TranceptionModel code."""
    import zlib
    from proteingym_amd import tranception as ptr

    class Fake(ptr.TranceptionModel):
        def __init__(self):
            self.n_ctx, self.scoring_window, self.retrieval, self._h, self.cfg = 66, scoring_window, None, None, {}

        def sequence_loglik(self, sliced, window_start=None, window_end=None, reverse=False, **kw):
            salt = 7 if reverse else 3
            return np.array([-(zlib.crc32((s + str(salt)).encode()) % 100003) / 977.0 for s in sliced], dtype=np.float32)

        def close(self):
            pass
    return Fake()


def _make_tranception_assays(workdir, indel):
    import pandas as pd
    from proteingym_amd import synthetic, tranception as ptr
    rng = np.random.default_rng(5)
    rows = []
    os.makedirs(os.path.join(workdir, "dms"), exist_ok=True)
    for a, (L, n) in enumerate([(40, 700), (150, 90), (64, 300), (33, 5)]):          # 150 > the fake context of 64 residues
        seq, muts, _ = synthetic.random_assay(seed=a, L=L, n_single=n // 2, n_multi=n - n // 2)
        if indel:
            wt, lib_ = synthetic.random_indel_library(a, L, n)
            df = pd.DataFrame({"mutant": [f"v{i}" for i in range(n)], "mutated_sequence": lib_})
            df.loc[n // 3, "mutated_sequence"] = wt                                   # the wild type is listed
            df.loc[n // 3, "mutant"] = wt
            df.loc[n - 1, "mutated_sequence"] = df.loc[0, "mutated_sequence"]         # a duplicate sequence under another label
            seq = wt
        else:
            muts[n // 4] = muts[0]                                                    # a duplicated row
            df = pd.DataFrame({"mutant": muts, "mutated_sequence": [ptr.get_mutated_sequence(seq, m) for m in muts],
                               "DMS_score": rng.standard_normal(n)})
        df.to_csv(os.path.join(workdir, "dms", f"T{a}.csv"), index=False)
        rows.append({"DMS_id": f"T{a}", "DMS_filename": f"T{a}.csv", "target_seq": seq, "DMS_total_number_mutants": n})
    pd.DataFrame(rows).to_csv(os.path.join(workdir, "map.csv"), index=False)
    return rows


def _tranception_args(workdir, window, indel):
    return ["tranception", "--backend", "gloo", "--max-chunk-rows", "64", "--", "--checkpoint", "fake",
            "--DMS_reference_file_path", os.path.join(workdir, "map.csv"), "--DMS_data_folder", os.path.join(workdir, "dms"),
            "--output_scores_folder", os.path.join(workdir, "out"), "--scoring_window", window] + (["--indel_mode"] if indel else [])


def _mutant_chunk_worker(rank, world, port, workdir, window, indel, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    from proteingym_amd import run_sharded
    mine = run_sharded.main(_tranception_args(workdir, window, indel), make_model=_fake_tranception)
    q.put((rank, mine))


@pytest.mark.parametrize("window,indel", [("optimal", False), ("sliding", False), ("optimal", True)])
def test_tranception_mutant_chunks_world2_equal_the_unsharded_scorer(tmp_path, window, indel):
    """run_sharded tranception (--shard mutants): two gloo ranks score chunks of <= 64 rows of four assays (one longer than
    the context: several windows / sliding chunks; duplicated rows; in indel mode the wild type among the rows), one
    all_gather, the owners write the CSVs -- which must equal, value for value and row for row, what the single-assay
    scorer returns for the whole file."""
    import pandas as pd
    from proteingym_amd import run_sharded
    workdir = str(tmp_path)
    rows = _make_tranception_assays(workdir, indel)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mutant_chunk_worker, args=(r, 2, port, workdir, window, indel, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    items = sorted(res[0] + res[1])
    assert len(res[0]) and len(res[1]) and len(set(items)) == len(items)
    for k, r in enumerate(rows):                                                      # every row of every assay exactly once
        spans = [(a, b) for kk, a, b in items if kk == k]
        assert spans[0][0] == 0 and spans[-1][1] == r["DMS_total_number_mutants"] and all(x[1] == y[0] for x, y in zip(spans, spans[1:]))
        assert max(b - a for a, b in spans) <= 64
    model = _fake_tranception("fake", 0, window)
    for r in rows:
        df = pd.read_csv(os.path.join(workdir, "dms", r["DMS_filename"]))
        want = model.score_mutants(DMS_data=df, target_seq=r["target_seq"], indel_mode=indel)
        got = pd.read_csv(os.path.join(workdir, "out", r["DMS_id"] + ".csv"), float_precision="round_trip")
        assert list(got.columns) == list(want.columns)
        assert len(got) == len(want)
        for c in want.columns:
            if want[c].dtype == object:
                assert list(got[c].fillna("")) == list(want[c].fillna("")), (r["DMS_id"], c)
            else:
                assert np.array_equal(got[c].to_numpy(dtype=np.float64), want[c].to_numpy(dtype=np.float64), equal_nan=True), (r["DMS_id"], c)
    # one process, no chunk cap: one item per assay, the same CSVs
    argv = _tranception_args(workdir, window, indel)
    cli_part = argv[argv.index("--") + 1:]
    cli_part[cli_part.index("--output_scores_folder") + 1] = os.path.join(workdir, "out1")
    one = run_sharded.main(["tranception", "--", *cli_part], make_model=_fake_tranception)
    assert one == [(k, 0, r["DMS_total_number_mutants"]) for k, r in enumerate(rows)]
    for r in rows:
        a = open(os.path.join(workdir, "out", r["DMS_id"] + ".csv")).read()
        assert a == open(os.path.join(workdir, "out1", r["DMS_id"] + ".csv")).read()


def test_tranception_mutant_chunks_plan_from_the_files_not_the_table(tmp_path):
    """The chunk plan counts the assay FILES' rows: a table whose DMS_total_number_mutants is stale (or missing) and a
    header-only assay file are scored like the single-assay CLI scores them -- the former in full, the latter as a
    header-only CSV without a scorer call -- on one process and on two gloo ranks."""
    import pandas as pd
    from proteingym_amd import run_sharded
    workdir = str(tmp_path)
    rows = _make_tranception_assays(workdir, indel=False)
    table = pd.read_csv(os.path.join(workdir, "map.csv"))
    true_n = [r["DMS_total_number_mutants"] for r in rows]
    table.loc[0, "DMS_total_number_mutants"] = 650                 # the file has 700 rows
    table.loc[1, "DMS_total_number_mutants"] = float("nan")
    empty = pd.read_csv(os.path.join(workdir, "dms", "T3.csv")).iloc[:0]
    empty.to_csv(os.path.join(workdir, "dms", "T3.csv"), index=False)          # header only; the table still says 5
    true_n[3] = 0
    table.to_csv(os.path.join(workdir, "map.csv"), index=False)
    assert run_sharded.rows_per_assay(table, range(4), os.path.join(workdir, "dms")) == true_n
    # counted by the parser that scores: a quoted field with an embedded newline and a blank-looking line are pandas' rows, not lines
    odd = pd.DataFrame({"mutant": ["A1C", "A1D"], "note": ["two\nlines", " "]})
    odd.to_csv(os.path.join(workdir, "dms", "odd.csv"), index=False)
    probe = pd.DataFrame({"DMS_filename": ["odd.csv", "absent.csv", "absent.csv"], "DMS_total_number_mutants": [9, 12, float("nan")]})
    assert run_sharded.rows_per_assay(probe, range(3), os.path.join(workdir, "dms")) == [2, -1, -1]
    assert run_sharded.planned_rows(probe, range(3), os.path.join(workdir, "dms"), 0, 1) == ([2, 12, 0], [True, False, False])
    argv = _tranception_args(workdir, "optimal", False)
    cli_part = argv[argv.index("--") + 1:]
    cli_part[cli_part.index("--output_scores_folder") + 1] = os.path.join(workdir, "out1")
    one = run_sharded.main(["tranception", "--", *cli_part], make_model=_fake_tranception)
    assert one == [(k, 0, n) for k, n in enumerate(true_n)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mutant_chunk_worker, args=(r, 2, port, workdir, "optimal", False, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sum(b - a for part in res.values() for _, a, b in part) == sum(true_n)
    model = _fake_tranception("fake", 0, "optimal")
    for k, r in enumerate(rows):
        a = open(os.path.join(workdir, "out", r["DMS_id"] + ".csv")).read()
        assert a == open(os.path.join(workdir, "out1", r["DMS_id"] + ".csv")).read()
        got = pd.read_csv(os.path.join(workdir, "out", r["DMS_id"] + ".csv"), float_precision="round_trip")
        if true_n[k] == 0:
            assert len(got) == 0 and "avg_score" in got.columns
        else:
            want = model.score_mutants(DMS_data=pd.read_csv(os.path.join(workdir, "dms", r["DMS_filename"])), target_seq=r["target_seq"])
            assert len(got) == len(want) and np.array_equal(got["avg_score"].to_numpy(), want["avg_score"].to_numpy())


def _prior_worker(rank, cache, q):
    from proteingym_amd import run_sharded

    class P:                                                      # counts the builds through a file per call
        @staticmethod
        def build_retrieval(r):
            import time as _t
            open(os.path.join(cache, f"built_by_{rank}"), "w").close()
            _t.sleep(0.5)
            return dict(log_prior=np.arange(6, dtype=np.float32).reshape(2, 3) + 0.25, MSA_start=r["MSA_start"], MSA_end=r["MSA_end"], weight=0.6)
    out = run_sharded.shared_retrieval(P, dict(MSA_start=1, MSA_end=3), cache, "job_T0")
    q.put((rank, out["log_prior"].tolist(), out["MSA_start"], out["MSA_end"], out["weight"]))


def test_retrieval_prior_is_built_once_for_the_ranks_that_share_an_assay(tmp_path):
    """shared_retrieval: of the ranks that hold chunks of one assay exactly one builds the log-prior, the others read its file."""
    cache = str(tmp_path / "cache")
    os.makedirs(cache)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_prior_worker, args=(r, cache, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len([f for f in os.listdir(cache) if f.startswith("built_by_")]) == 1
    assert all(r[1:] == res[0][1:] for r in res) and res[0][1] == [[0.25, 1.25, 2.25], [3.25, 4.25, 5.25]]


def _failing_prior_worker(rank, cache, q):
    from proteingym_amd import run_sharded

    class P:
        @staticmethod
        def build_retrieval(r):
            import time as _t
            open(os.path.join(cache, f"built_by_{rank}"), "w").close()
            _t.sleep(0.5)
            raise FileNotFoundError("no alignment for this assay")
    t0 = time.time()
    try:
        run_sharded.shared_retrieval(P, dict(MSA_start=1, MSA_end=3), cache, "job_T1", wait_s=600.0)
        q.put((rank, "no error", time.time() - t0))
    except BaseException as e:
        q.put((rank, f"{type(e).__name__}: {e}", time.time() - t0))


def test_failing_retrieval_builder_fails_the_waiting_ranks_at_once(tmp_path):
    """A builder that raises publishes <tag>.failed: the ranks waiting for its prior raise the same message within the poll
    interval (not after wait_s, and without rebuilding into the same deterministic error)."""
    cache = str(tmp_path / "cache")
    os.makedirs(cache)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_failing_prior_worker, args=(r, cache, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len([f for f in os.listdir(cache) if f.startswith("built_by_")]) == 1          # nobody rebuilt
    assert all("no alignment for this assay" in r[1] for r in res), res
    assert max(r[2] for r in res) < 30.0


def test_retrieval_waiter_takes_over_from_a_dead_builder(tmp_path):
    """A lock whose pid is gone (builder killed) does not cost the waiter wait_s: it builds the prior itself."""
    from proteingym_amd import run_sharded
    cache = str(tmp_path / "cache")
    os.makedirs(cache)
    dead = mp.get_context("spawn").Process(target=int)
    dead.start()
    dead.join()
    open(os.path.join(cache, "job_T2.lock"), "w").write(f"{run_sharded._lock_host()}:{dead.pid}")

    class P:
        @staticmethod
        def build_retrieval(r):
            return dict(log_prior=np.ones((2, 3), np.float32), MSA_start=1, MSA_end=3, weight=0.6)
    t0 = time.time()
    out = run_sharded.shared_retrieval(P, dict(MSA_start=1, MSA_end=3), cache, "job_T2", wait_s=600.0)
    assert time.time() - t0 < 30.0 and out["log_prior"].shape == (2, 3)
    # a builder on ANOTHER host (or in another pid namespace) cannot be probed: the same pid number there says nothing here, so the
    # waiter keeps polling for the prior (here: until its wait_s is over) instead of rebuilding beside a live builder at once
    open(os.path.join(cache, "job_T3.lock"), "w").write(f"some-other-node/pid:[1]:{dead.pid}")
    t0 = time.time()
    out = run_sharded.shared_retrieval(P, dict(MSA_start=1, MSA_end=3), cache, "job_T3", wait_s=1.5)
    assert 1.4 < time.time() - t0 < 30.0 and out["log_prior"].shape == (2, 3)


def test_tranception_chunk_plan_balances_the_real_table():
    """Config 4 on 8 GPUs: whole assays give max/mean 2.08 on the real substitution table (one assay is 26 % of the
    cost); mutant chunks must bring every rank within 2 % of the mean, and keep the number of assays a rank touches (one
    retrieval prior each) near 217 / world."""
    from proteingym_amd import run_sharded, synthetic
    shapes = synthetic.dms_shapes()
    L, n = [s["seq_len"] for s in shapes], [s["n_total"] for s in shapes]
    for world in (2, 4, 8):
        items, assignment, costs = run_sharded.plan_mutant_chunks(L, n, world)
        loads = np.array([sum(costs[j] for j in part) for part in assignment])
        assert loads.max() / loads.mean() < 1.02
        assert sorted(j for part in assignment for j in part) == list(range(len(items)))
        for k in range(217):                                                          # chunks tile every assay's rows
            spans = [(a, b) for kk, a, b in items if kk == k]
            assert spans[0][0] == 0 and spans[-1][1] == n[k] and all(x[1] == y[0] for x, y in zip(spans, spans[1:]))
        assert max(len({items[j][0] for j in part}) for part in assignment) <= 217 // world + 16
    whole = pdist.lpt_partition([run_sharded.chunk_cost(min(l + 2, 1024), m) for l, m in zip(L, n)], 8)
    wl = np.array([sum(run_sharded.chunk_cost(min(L[k] + 2, 1024), n[k]) for k in part) for part in whole])
    assert wl.max() / wl.mean() > 1.9                                                  # what the plan replaces


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

    def score_group(self, assays):                 # the runner hands a rank's short assays over several at a time
        assert len(assays) > 1
        return [self.score(*a) for a in assays]

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


# ---- the 217-assay benchmark on 8 ranks: planned balance at the real shapes, and a world-8 gloo run of the runner ----
def test_plan_of_the_217_assay_benchmark_on_8_ranks_balances_time_flops_and_rows():
    """The assay-level plan for N = 8 over the REAL table (seq_len 37..3423, 2 465 767 mutants, one assay with 536 962):
    planned seconds (GPU FLOPs for 5 checkpoints + host seconds per row) within 1 % of the mean on every rank; the
    pure-FLOP and the row imbalance are reported and bounded."""
    import pandas as pd
    from proteingym_amd import run_benchmark as rb, synthetic
    shapes = synthetic.dms_shapes()
    assert len(shapes) == 217 and sum(s["n_total"] for s in shapes) == 2465767
    mapping = pd.DataFrame({"target_seq": ["M" * s["seq_len"] for s in shapes], "DMS_total_number_mutants": [s["n_total"] for s in shapes]})
    todo = list(range(217))
    a = rb.plan_assays(mapping, todo, 8, 5)
    assert sorted(k for part in a for k in part) == todo
    secs = np.array([sum(rb.assay_seconds(shapes[k]["seq_len"], shapes[k]["n_total"], 5) for k in part) for part in a])
    flops = np.array([sum(pdist.assay_cost(shapes[k]["seq_len"]) for k in part) for part in a])
    rows = np.array([sum(shapes[k]["n_total"] for k in part) for part in a])
    print("planned s/rank", np.round(secs, 1), "flop max/mean", flops.max() / flops.mean(), "rows/rank", rows)
    assert secs.max() / secs.mean() < 1.01
    assert flops.max() / flops.mean() < 1.05
    # the rank that owns the 537k-row assay carries fewer FLOPs instead: host seconds are part of the plan
    big = max(range(217), key=lambda k: shapes[k]["n_total"])
    r_big = next(r for r, part in enumerate(a) if big in part)
    assert flops[r_big] < flops.mean()


def _world8_worker(rank, world, port, workdir, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    from proteingym_amd import run_benchmark as rb
    args = rb.create_parser().parse_args([
        "--model-location", "ckA.pt", "ckB.pt", "--model_type", "ESM1v", "--dms_mapping", os.path.join(workdir, "map.csv"),
        "--dms-input", workdir, "--dms-output", os.path.join(workdir, "out"), "--backend", "gloo"])
    rb.main(args, make_model=_FakeScorer)
    q.put(rank)


def test_assay_shards_world8_on_217_shaped_table(tmp_path):
    """Eight gloo ranks over a 217-assay table with the benchmark's sequence lengths and 1/400 of its mutant counts:
    todo list broadcast from rank 0, LPT by planned seconds, owners write their CSVs from a background thread, one
    fixed-stride all_gather, rank 0's scores_summary.csv lists every assay."""
    import pandas as pd
    from proteingym_amd import synthetic
    shapes = synthetic.dms_shapes()
    rows, truth = [], {}
    for s in shapes:
        n = max(2, s["n_total"] // 400)
        seq, muts, score = synthetic.random_assay(seed=s["DMS_index"], L=s["seq_len"], n_single=n, n_multi=0)
        pd.DataFrame({"mutant": muts, "DMS_score": score}).to_csv(tmp_path / f"{s['DMS_id']}.csv", index=False)
        rows.append({"DMS_id": s["DMS_id"], "DMS_filename": f"{s['DMS_id']}.csv", "target_seq": seq, "DMS_total_number_mutants": n})
        truth[s["DMS_id"]] = (seq, muts)
    pd.DataFrame(rows).to_csv(tmp_path / "map.csv", index=False)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_world8_worker, args=(r, 8, port, str(tmp_path), q)) for r in range(8)]
    for p in procs:
        p.start()
    assert sorted(q.get(timeout=600) for _ in procs) == list(range(8))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    summary = pd.read_csv(tmp_path / "out" / "scores_summary.csv")
    assert sorted(summary["DMS_id"]) == sorted(truth) and int(summary["mutants"].sum()) == sum(len(m) for _, m in truth.values())
    for name in list(truth)[::23]:
        seq, muts = truth[name]
        got = pd.read_csv(tmp_path / "out" / f"{name}.csv", float_precision="round_trip")
        a, b = _FakeScorer("ckA.pt").score(seq, muts, 1), _FakeScorer("ckB.pt").score(seq, muts, 1)
        assert np.array_equal(got["ckA"].to_numpy(), a) and np.array_equal(got["Ensemble_ESM1v"].to_numpy(), (a + b) / 2)


# ---- MSA Transformer: (seed, position) pairs sharded over ranks ----------------------------------------
def _msa_pos_worker(rank, world, port, golden_dir, outdir, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    from proteingym_amd import msa_transformer as pmsa, run_sharded

    class Fake:                                   # a deterministic row per (sampled grid, masked column): no device, no oracle
        calls = []

        def __init__(self, path):
            pass

        def masked_logprobs(self, tokens, positions, seq_len, window=1024):
            Fake.calls.append(list(positions))
            t = np.asarray(tokens)
            out = []
            for p in positions:
                rng = np.random.default_rng(int(t.sum()) * 1000003 + int(p))
                x = rng.standard_normal(33).astype(np.float32)
                out.append(x - np.log(np.exp(x).sum()))
            return np.stack(out)

        def close(self):
            pass

    pmsa.load_model_and_alphabet = lambda loc, device=0, max_rows=0: (Fake(loc), pmsa.MsaAlphabet())
    run_sharded.main(["msa_transformer", "--shard", "positions", "--backend", "gloo", "--",
                      "--model-location", os.path.join(golden_dir, "msa_toy.pt"), "--model_type", "MSA_transformer",
                      "--dms_mapping", os.path.join(golden_dir, "TOY_MSA_MAPPING.csv"), "--dms-input", golden_dir, "--dms-output", outdir,
                      "--scoring-strategy", "masked-marginals", "--msa-path", golden_dir, "--msa-weights-folder", golden_dir,
                      "--msa-samples", "12", "--seeds", "1", "2"])
    q.put((rank, Fake.calls))


def test_msa_transformer_seed_position_pairs_shard_over_two_ranks(tmp_path):
    """run_sharded msa_transformer --shard positions: both gloo ranks run the assay, each forwards every second masked
    position of every seed, the tables are all_gathered; the CSV rank 0 writes equals the single-rank CSV bit for bit."""
    import pandas as pd
    golden_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    ctx = mp.get_context("spawn")
    res = {}
    for world, tag in ((1, "w1"), (2, "w2")):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_msa_pos_worker, args=(r, world, port, golden_dir, str(tmp_path / tag), q)) for r in range(world)]
        for p in procs:
            p.start()
        calls = dict(q.get(timeout=300) for _ in procs)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        res[tag] = (pd.read_csv(tmp_path / tag / "TOY_MSA_DMS.csv", float_precision="round_trip"), calls)
    (d1, c1), (d2, c2) = res["w1"], res["w2"]
    assert list(d1.columns) == list(d2.columns) and "msa_toy_ensemble" in d1.columns
    for c in ("msa_toy_seed1", "msa_toy_seed2", "msa_toy_ensemble"):
        assert np.array_equal(d1[c].to_numpy(), d2[c].to_numpy())
    for seed_k in range(2):                       # per seed: the two ranks' positions are disjoint and together the single rank's
        assert sorted(c2[0][seed_k] + c2[1][seed_k]) == sorted(c1[0][seed_k]) and not set(c2[0][seed_k]) & set(c2[1][seed_k])


# ---- bench.py --gpus N: the strong-scaling pass over the 217-assay-shaped table (scripts/bench_scale.py) -----------------
class _FakeBenchAssay:
    def __init__(self, model, seq, muts):
        self.positions = np.arange(min(len(seq), len(muts)))
        self.T = min(len(seq) + 2, 1024)
        self.n = len(muts)
        model.append(self.n)

    def run_device_only(self, ptr=0):
        pass

    def close(self):
        pass


def _bench_scale_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import bench_scale
    pdist.init_from_env("gloo")
    seen = []
    st = bench_scale.run(seen, rank, world, steps=5, warmup=2, torch=torch, tdist=dist, max_assays=24, make_assay=_FakeBenchAssay)
    q.put((rank, sum(seen), st))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_strong_scaling_pass_covers_the_table_once_on_two_ranks():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import bench_scale
    from proteingym_amd import synthetic
    for n, k in ((0, 4), (3, 20), (20, 20), (41, 5)):                       # K slices tile a work list of any length
        sl = bench_scale.step_slices(n, k)
        assert len(sl) == k and sl[0][0] == 0 and sl[-1][1] == n and all(a[1] == b[0] for a, b in zip(sl, sl[1:]))
    shapes, assignment = bench_scale.plan(8)
    assert len(shapes) == 217 and sorted(i for part in assignment for i in part) == list(range(217))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_scale_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = sum(s["n_total"] for s in synthetic.dms_shapes()[:24])
    assert sum(r[1] for r in res) == want                                   # every assay generated and uploaded by exactly one rank
    for _, _, st in res:
        assert st["mutants"] == want and st["assays"] == 24 and st["seconds"] >= st["fastest_rank_seconds"] > 0
        assert st["positions_run"] > 0 and st["executed_algorithmic_flops"] > 0


# ---- failure isolation: a bad assay / an fp16 overflow must neither lose the job nor hang the other ranks ----------------
class _FlakyScorer(_FakeScorer):
    """_FakeScorer with real-data trouble: the 55-residue assay's file has a wild-type mismatch (the assertion of
    compute_fitness.py:243), the 41-residue assay leaves the fp16 range on checkpoint ckA unless the model is fp32."""

    def __init__(self, location, precision=None):
        super().__init__(location)
        self.location, self.precision = location, precision

    def score(self, seq, mutants, offset):
        if len(seq) == 55:
            raise AssertionError("The listed wildtype does not match the provided sequence")
        if len(seq) == 41 and self.location == "ckA.pt" and self.precision != "fp32":
            from proteingym_amd import _lib
            raise _lib.PgmiError("libpgmi error -6: non-finite log-probabilities", code=_lib.EOVERFLOW)
        v = super().score(seq, mutants, offset)
        return v + 0.5 if self.precision == "fp32" else v


def _flaky_assay_worker(rank, world, port, workdir, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    from proteingym_amd import run_benchmark as rb
    args = rb.create_parser().parse_args([
        "--model-location", "ckA.pt", "ckB.pt", "--model_type", "ESM1v", "--dms_mapping", os.path.join(workdir, "map.csv"),
        "--dms-input", workdir, "--dms-output", os.path.join(workdir, "out_flaky"), "--backend", "gloo"])
    try:
        rb.main(args, make_model=_FlakyScorer)
    except SystemExit as e:
        q.put((rank, str(e)))
        raise
    q.put((rank, "clean exit"))


def _write_five_assays(tmp_path):
    import pandas as pd
    from proteingym_amd import synthetic
    rows, assays = [], {}
    for k, L in enumerate((30, 90, 55, 41, 120)):
        seq, muts, score = synthetic.random_assay(seed=10 + k, L=L, n_single=20 + k, n_multi=5)
        pd.DataFrame({"mutant": muts, "DMS_score": score}).to_csv(tmp_path / f"B{k}.csv", index=False)
        rows.append({"DMS_id": f"B{k}", "DMS_filename": f"B{k}.csv", "target_seq": seq})
        assays[f"B{k}"] = (seq, muts)
    pd.DataFrame(rows).to_csv(tmp_path / "map.csv", index=False)
    return assays


def test_assay_shards_world2_survive_a_bad_assay_and_retry_an_overflow_in_fp32(tmp_path):
    """One assay raises the reference's wild-type assertion, one leaves the fp16 range on one checkpoint: both gloo ranks
    reach every collective (no hang), the other assays' CSVs are byte-identical to a clean one-process run, the overflowing
    (assay, checkpoint) carries the fp32 model's scores and scores_summary.csv says so, the bad assay has no CSV, and both
    ranks exit non-zero."""
    import pandas as pd
    from proteingym_amd import run_benchmark as rb
    assays = _write_five_assays(tmp_path)
    clean = rb.create_parser().parse_args([
        "--model-location", "ckA.pt", "ckB.pt", "--model_type", "ESM1v", "--dms_mapping", str(tmp_path / "map.csv"),
        "--dms-input", str(tmp_path), "--dms-output", str(tmp_path / "out_clean")])
    rb.main(clean, make_model=_FakeScorer)
    assert not (tmp_path / "out_clean" / "scores_summary.csv").exists()          # one process, nothing to report
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_flaky_assay_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    said = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode not in (0, None)
    assert all("1 assay(s) failed" in v for v in said.values()), said
    out = tmp_path / "out_flaky"
    for name in ("B0", "B1", "B4"):
        assert (out / f"{name}.csv").read_bytes() == (tmp_path / "out_clean" / f"{name}.csv").read_bytes()
    assert not (out / "B2.csv").exists()
    seq, muts = assays["B3"]
    got = pd.read_csv(out / "B3.csv", float_precision="round_trip")
    a, b = _FakeScorer("ckA.pt").score(seq, muts, 1) + 0.5, _FakeScorer("ckB.pt").score(seq, muts, 1)
    assert np.array_equal(got["ckA"].to_numpy(), a) and np.array_equal(got["ckB"].to_numpy(), b)
    assert np.array_equal(got["Ensemble_ESM1v"].to_numpy(), (a + b) / 2)
    summary = pd.read_csv(out / "scores_summary.csv", keep_default_na=False).set_index("DMS_id")
    assert list(summary["status"]) == ["ok", "ok", "failed", "ok", "ok"]
    assert "wildtype does not match" in summary.loc["B2", "error"]
    assert summary.loc["B3", "precision_ckA"] == "fp32" and summary.loc["B3", "precision_ckB"] == ""
    assert set(summary["precision_ckA"]) == {"", "fp32"}
    # a re-run takes up exactly the assay that has no CSV
    os.environ.pop("WORLD_SIZE", None)
    again = rb.create_parser().parse_args([
        "--model-location", "ckA.pt", "ckB.pt", "--model_type", "ESM1v", "--dms_mapping", str(tmp_path / "map.csv"),
        "--dms-input", str(tmp_path), "--dms-output", str(out)])
    stats = rb.main(again, make_model=_FakeScorer)
    assert stats["assays"] == 1 and (out / "B2.csv").read_bytes() == (tmp_path / "out_clean" / "B2.csv").read_bytes()


class _FlakyTables(_FakeTables):
    """Position chunks: the 61-residue assay leaves the fp16 range in ONE of its chunks on ckA (position 40); a table must not
    mix precisions, so every rank redoes its chunks of that assay on an fp32 model (rows + 1 here)."""

    def __init__(self, location, precision=None):
        super().__init__(location)
        self.location, self.precision = location, precision

    def table_rows(self, seq, positions, offset):
        if len(seq) == 61 and self.location == "ckA.pt" and self.precision != "fp32" and 40 in [int(p) for p in positions]:
            from proteingym_amd import _lib
            raise _lib.PgmiError("libpgmi error -6: non-finite log-probabilities", code=_lib.EOVERFLOW)
        rows = super().table_rows(seq, positions, offset)
        return rows + np.float32(1.0) * (np.arange(33, dtype=np.float32) % 2) if self.precision == "fp32" else rows


def _flaky_position_worker(rank, world, port, workdir, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    from proteingym_amd import run_benchmark as rb
    args = rb.create_parser().parse_args([
        "--model-location", "ckA.pt", "ckB.pt", "--model_type", "ESM1v", "--dms_mapping", os.path.join(workdir, "map.csv"),
        "--dms-input", workdir, "--dms-output", os.path.join(workdir, "out"), "--backend", "gloo", "--shard", "positions",
        "--chunk-forwards", "5"])
    try:
        rb.main(args, make_model=_FlakyTables)
    except SystemExit as e:
        q.put((rank, str(e)))
        raise
    q.put((rank, "clean exit"))


def test_position_shards_world2_bad_file_and_whole_assay_fp32_retry(tmp_path):
    """--shard positions with trouble: one assay's file lists a wrong wild type (dropped on every rank before the plan), one
    overflows in a single chunk (the whole assay, on both ranks, goes through fp32).  No hang, the clean assay's CSV equals
    the scores from the complete f16x3 tables, the retried one equals the scores from complete fp32 tables, exit code != 0."""
    import pandas as pd
    from proteingym_amd import esm as pesm, synthetic
    rows, assays = [], {}
    for k, L in enumerate((23, 61, 40)):
        seq, muts, score = synthetic.random_assay(seed=k, L=L, n_single=30, n_multi=12)
        if k == 2:
            muts[3] = ("A" if seq[4] != "A" else "C") + "5" + "W"          # wild-type letter that is not in the sequence
        pd.DataFrame({"mutant": muts, "DMS_score": score}).to_csv(tmp_path / f"A{k}.csv", index=False)
        rows.append({"DMS_id": f"A{k}", "DMS_filename": f"A{k}.csv", "target_seq": seq})
        assays[f"A{k}"] = (seq, muts)
    pd.DataFrame(rows).to_csv(tmp_path / "map.csv", index=False)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_flaky_position_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    said = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode not in (0, None)
    assert all("1 assay(s) failed" in v for v in said.values()), said
    assert not (tmp_path / "out" / "A2.csv").exists()
    for name, fp32_on in (("A0", ()), ("A1", ("ckA",))):
        seq, muts = assays[name]
        got = pd.read_csv(tmp_path / "out" / f"{name}.csv", float_precision="round_trip")
        for ck in ("ckA", "ckB"):
            fake = _FlakyTables(ck + ".pt", precision="fp32" if ck in fp32_on else None)
            table = np.full((len(seq) + 2, 33), np.nan, dtype=np.float32)
            for p in pesm.positions_read(muts, seq, 1):
                table[p] = fake.table_rows(seq, [int(p)], 1)[0]
            assert np.array_equal(got[ck].to_numpy(), pesm.score_from_table(table, muts, seq, 1))
    summary = pd.read_csv(tmp_path / "out" / "scores_summary.csv", keep_default_na=False).set_index("DMS_id")
    assert list(summary["status"]) == ["ok", "ok", "failed"] and summary.loc["A1", "precision_ckA"] == "fp32"
    assert summary.loc["A0", "precision_ckA"] == "" and "does not match" in summary.loc["A2", "error"]


class _FlakyPppl(_FakePppl):
    """Pooled indel libraries: sequences of the 60-residue assay hold a character the tokenizer rejects; the 220-residue
    assay leaves the fp16 range on esm2_a unless the model is fp32 (scores - 1 there)."""

    def __init__(self, location, precision=None):
        super().__init__(location)
        self.location, self.precision = location, precision

    def score(self, sequences):
        if any("?" in s for s in sequences):
            raise KeyError("?")
        if self.location == "esm2_a.pt" and self.precision != "fp32" and any(200 < len(s) < 240 for s in sequences):
            from proteingym_amd import _lib
            raise _lib.PgmiError("libpgmi error -6: non-finite log-probabilities", code=_lib.EOVERFLOW)
        v = super().score(sequences)
        return v - 1.0 if self.precision == "fp32" else v


def _flaky_indel_worker(rank, world, port, workdir, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    from proteingym_amd import run_indels as ri
    args = ri.create_parser().parse_args([
        "--model-location", "esm2_a.pt", "esm2_b.pt", "--model_type", "ESM2", "--dms_mapping", os.path.join(workdir, "map.csv"),
        "--dms-input", workdir, "--dms-output", os.path.join(workdir, "out"), "--backend", "gloo"])
    try:
        ri.main(args, make_model=_FlakyPppl)
    except SystemExit as e:
        q.put((rank, str(e)))
        raise
    q.put((rank, "clean exit"))


def test_run_indels_world2_isolates_a_bad_library_and_retries_an_overflow(tmp_path):
    """The pooled pseudo-ppl runner with trouble in two of four assays (one of them unreadable): the pool is built without the
    unreadable one on both ranks, the rank shares are re-scored assay by assay, the overflowing assay goes through fp32 on
    both ranks, the clean assay's CSV holds the clean scores, exit code != 0, no hang."""
    import pandas as pd
    from proteingym_amd import synthetic
    rng = np.random.default_rng(5)
    rows, truth = [], {}
    for k, (L, n) in enumerate(((60, 25), (400, 60), (220, 40))):
        wt = synthetic.random_sequence(rng, L)
        seqs = []
        for _ in range(n):
            p, d = int(rng.integers(1, L - 4)), int(rng.integers(-3, 4))
            seqs.append(wt[:p] + (synthetic.random_sequence(rng, d) if d > 0 else "") + wt[p - min(d, 0):])
        if k == 0:
            seqs[7] = seqs[7][:5] + "?" + seqs[7][6:]
        pd.DataFrame({"mutant": seqs, "mutated_sequence": seqs, "DMS_score": rng.standard_normal(n)}).to_csv(tmp_path / f"I{k}.csv", index=False)
        rows.append({"DMS_id": f"I{k}", "DMS_filename": f"I{k}.csv", "target_seq": wt})
        truth[f"I{k}"] = seqs
    rows.append({"DMS_id": "I3", "DMS_filename": "I3_is_missing.csv", "target_seq": "MKV"})
    pd.DataFrame(rows).to_csv(tmp_path / "map.csv", index=False)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_flaky_indel_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    said = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode not in (0, None)
    assert all("2 assay(s) failed" in v for v in said.values()), said
    assert not (tmp_path / "out" / "I0.csv").exists() and not (tmp_path / "out" / "I3.csv").exists()
    got = pd.read_csv(tmp_path / "out" / "I1.csv", float_precision="round_trip")
    for ck in ("esm2_a", "esm2_b"):
        assert np.array_equal(got[ck].to_numpy(), _FakePppl(ck + ".pt").score(truth["I1"]))
    got = pd.read_csv(tmp_path / "out" / "I2.csv", float_precision="round_trip")
    assert np.array_equal(got["esm2_a"].to_numpy(), _FakePppl("esm2_a.pt").score(truth["I2"]) - 1.0)
    assert np.array_equal(got["esm2_b"].to_numpy(), _FakePppl("esm2_b.pt").score(truth["I2"]))
    summary = pd.read_csv(tmp_path / "out" / "scores_summary.csv", keep_default_na=False).set_index("DMS_id")
    assert list(summary["status"]) == ["failed", "ok", "ok", "failed"]
    assert summary.loc["I2", "precision_esm2_a"] == "fp32" and summary.loc["I2", "precision_esm2_b"] == ""


# ---- bench.py --gpus N: the JSON line is the ONLY thing on stdout (backend chatter goes to stderr) ------------------------------
class _FakeBenchModel(list):
    """Stands in for EsmModel in bench.main's N > 1 branch: the per-class profile with plausible numbers, nothing else."""

    def profile_reset(self):
        pass

    def profile_enable(self, on=True):
        pass

    def profile(self):
        from proteingym_amd import _lib
        return {k: dict(ms=1.0, launches=66, flops=1.0e12, bytes=0.0) for k in _lib.K_NAMES}

    def close(self):
        pass


def _bench_main_entry():
    """Runs in a child process (see the test below): bench.main's N > 1 branch over gloo with the seams."""
    import bench

    def model(cfg, blob, dev, prec):
        print("[Gloo] Rank chatter that a backend might print on stdout")        # inside bench.main: must end up on stderr
        os.system("echo chatter of a child process that inherits file descriptor 1")
        return _FakeBenchModel()
    bench.main(["--gpus", "2", "--steps", "2", "--warmup", "1"], make_model=model, make_assay=_FakeBenchAssay)


def test_bench_n_gt_1_prints_exactly_one_json_line_on_stdout():
    """The driver launches `bench.py --gpus N` under torch.distributed.run and reads stdout: two gloo ranks through bench.main
    (model and assay seams, 24 assays of the table) -- rank 0's stdout is ONE line and it is the JSON object with the contract's
    keys, rank 1's stdout is empty; everything any library printed is on stderr."""
    import json
    import subprocess
    import sys
    port = _free_port()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = []
    for r in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2",
                   PGMI_BENCH_217_ASSAYS="24", OMP_NUM_THREADS="1")
        code = f"import sys; sys.path[:0] = [{root!r}, {os.path.join(root, 'tests')!r}]; import test_dist_cpu as t; t._bench_main_entry()"
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs
    lines = outs[0][0].splitlines()
    assert len(lines) == 1, outs[0][0]
    line = json.loads(lines[0])
    assert outs[1][0] == ""
    assert "Rank chatter" in outs[0][1] and "Rank chatter" in outs[1][1] and "chatter of a child process" in outs[0][1]
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "strong_scaling_217", "rccl"):
        assert key in line, key
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 2 and line["rccl"]["world_size"] == 2
    assert line["strong_scaling_217"]["assays"] == 24 and line["value"] > 0


# ---- run_indels: a job that dies is taken up where its saved slices ended -------------------------------------------------------
class _CountingPppl(_FakePppl):
    """_FakePppl that counts the sequences it is asked for and can be told to die (the way a node does: no exception handler
    gets to run the failure-isolation path) after a number of score() calls."""
    scored, calls, die_after = [], 0, None

    def score(self, sequences):
        cls = type(self)
        if cls.die_after is not None and cls.calls >= cls.die_after:
            raise KeyboardInterrupt("the node went away")
        cls.calls += 1
        cls.scored += list(sequences)
        return super().score(sequences)


def test_run_indels_takes_up_a_dead_job_where_its_slices_ended(tmp_path):
    """Config 5 is days of GPU time: a rank scores its share in slices and saves the scores it has after each; the same command run
    again scores only what is missing (per checkpoint), writes the files of a run that never died -- byte for byte --, and removes the
    slice files; another pool (a different digest) does not take them up."""
    import pandas as pd
    from proteingym_amd import run_indels as ri, synthetic
    rng = np.random.default_rng(9)
    rows = []
    for k, (L, n) in enumerate(((50, 30), (90, 40))):
        _, seqs = synthetic.random_indel_library(seed=30 + k, L=L, n=n)
        pd.DataFrame({"mutant": seqs, "mutated_sequence": seqs, "DMS_score": rng.standard_normal(n)}).to_csv(tmp_path / f"I{k}.csv", index=False)
        rows.append({"DMS_id": f"I{k}", "DMS_filename": f"I{k}.csv", "target_seq": "M"})
    pd.DataFrame(rows).to_csv(tmp_path / "map.csv", index=False)
    for name in ("esm2_a.pt", "esm2_b.pt"):
        (tmp_path / name).write_bytes(b"not read by the seam; its size is part of the digest")

    def args(out, every):
        return ri.create_parser().parse_args(["--model-location", str(tmp_path / "esm2_a.pt"), str(tmp_path / "esm2_b.pt"), "--model_type", "ESM2",
                                              "--dms_mapping", str(tmp_path / "map.csv"), "--dms-input", str(tmp_path), "--dms-output", str(tmp_path / out),
                                              "--save-every-forwards", str(every)])
    _CountingPppl.scored, _CountingPppl.calls, _CountingPppl.die_after = [], 0, None
    ri.main(args("clean", 0), make_model=_CountingPppl)                                  # one piece, nothing saved
    assert _CountingPppl.calls == 2 and not (tmp_path / "clean" / ".partial").exists()
    total = len(_CountingPppl.scored)                                                    # 70 sequences x 2 checkpoints
    _CountingPppl.scored, _CountingPppl.calls, _CountingPppl.die_after = [], 0, 3
    with pytest.raises(KeyboardInterrupt):
        ri.main(args("out", 900), make_model=_CountingPppl)                              # ~12 sequences of 90 residues per slice
    first = len(_CountingPppl.scored)
    saved = sorted(os.listdir(tmp_path / "out" / ".partial"))
    assert saved == ["esm2_a_rank0of1.npz"] and 0 < first < total // 2 and not (tmp_path / "out" / "I0.csv").exists()
    _CountingPppl.scored, _CountingPppl.calls, _CountingPppl.die_after = [], 0, None
    ri.main(args("out", 900), make_model=_CountingPppl)
    # the slice in flight when the job died was not saved: it is scored again; nothing that was saved is
    assert len(_CountingPppl.scored) < total - (first - 14) and len(_CountingPppl.scored) >= total - first
    for k in range(2):
        assert open(tmp_path / "out" / f"I{k}.csv").read() == open(tmp_path / "clean" / f"I{k}.csv").read()
    assert not (tmp_path / "out" / ".partial").exists()
    # a slice file of another share is not taken up: same name, other digest
    _CountingPppl.scored, _CountingPppl.calls, _CountingPppl.die_after = [], 0, 2
    with pytest.raises(KeyboardInterrupt):
        ri.main(args("other", 900), make_model=_CountingPppl)
    pd.read_csv(tmp_path / "I1.csv").iloc[::-1].to_csv(tmp_path / "I1.csv", index=False)         # the pool changes
    _CountingPppl.scored, _CountingPppl.calls, _CountingPppl.die_after = [], 0, None
    ri.main(args("other", 900), make_model=_CountingPppl)
    assert len(_CountingPppl.scored) == total


def test_run_sharded_skip_existing_takes_up_the_missing_assays(tmp_path):
    """run_sharded tranception --skip-existing: after a run that lost two of four CSVs (a dead job, failed assays) the same command
    scores exactly those two, leaves the others' files alone and writes what the first run wrote."""
    from proteingym_amd import run_sharded
    workdir = str(tmp_path)
    rows = _make_tranception_assays(workdir, False)
    argv = _tranception_args(workdir, "optimal", False)
    cli_part = argv[argv.index("--") + 1:]
    run_sharded.main(["tranception", "--", *cli_part], make_model=_fake_tranception)
    out = os.path.join(workdir, "out")
    first = {r["DMS_id"]: open(os.path.join(out, r["DMS_id"] + ".csv")).read() for r in rows}
    stamp = {r["DMS_id"]: os.stat(os.path.join(out, r["DMS_id"] + ".csv")).st_mtime_ns for r in rows}
    os.remove(os.path.join(out, "T1.csv"))
    open(os.path.join(out, "T3.csv"), "w").write("mutated_sequence\n")                # a torn file: no score column
    items = run_sharded.main(["tranception", "--skip-existing", "--", *cli_part], make_model=_fake_tranception)
    assert [n for _, _, n in items] == [rows[1]["DMS_total_number_mutants"], rows[3]["DMS_total_number_mutants"]]
    for r in rows:
        path = os.path.join(out, r["DMS_id"] + ".csv")
        assert open(path).read() == first[r["DMS_id"]]
        assert (os.stat(path).st_mtime_ns == stamp[r["DMS_id"]]) == (r["DMS_id"] in ("T0", "T2"))
    assert run_sharded.main(["tranception", "--skip-existing", "--", *cli_part], make_model=_fake_tranception) == []      # nothing left
