#!/usr/bin/env python
"""Benchmark of the hot path: mutants scored per second, ESM-1v 650M masked-marginals.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--precision fp32]

One *step* = one pass of the hot path over one assay-shaped batch resident in HBM:
BASELINE.json configs[1] -- ESM-1v 650M (33 x 1280 x 20 heads x 5120), one assay shaped like
BLAT_ECOLX_Stiffler_2015 (L=286 -> T=288 tokens, 4 996 single mutants), ONE checkpoint:
masked windows -> 33-layer forward over every masked position -> LM head on the masked rows
-> log-softmax table -> per-mutant score (label_row).  Synthetic sequence, synthetic
random-init weights (no network), deterministic seeds.

N > 1 (launched by torch.distributed.run, one rank per GPU): assays shard over ranks with no
data-path collective (each rank scores its own assay of the same shape: weak scaling), followed
by the RCCL all_gather of the per-mutant score vectors that the north-star names.  value =
all ranks' mutants / max-over-ranks time.

Extra objects on the JSON line: `roofline` (dominant kernel = the FFN GEMMs, HIP-event timed
inside the timed region, against the MFMA peak of the dtype) and `cpu_baseline` (the oracle's
CPU restatement of the reference path timed on this box's host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"fp32": 157.3, "bf16": 2500.0, "f16x3": 2500.0}   # MI355X_MICROARCH.md, dense
L_BLAT, N_MUT_BLAT = 286, 4996
PMC_TRAFFIC = os.path.join(ROOT, "profiles", "r2", "pmc_traffic.json")


def ffn_traffic(precision, M, D, F):
    """HBM-side bytes per FFN GEMM launch from the committed rocprofv3 --pmc passes of this same command
    (scripts/pmc_profile.sh: FETCH_SIZE and WRITE_SIZE in separate runs, FETCH_SIZE doubled per MI355X_MICROARCH.md).
    Counters cannot be read from inside the run; the summary is per launch and shape-specific, so it is only used when
    it was taken on this precision's kernels."""
    if precision != "f16x3" or not os.path.exists(PMC_TRAFFIC):
        return None, None
    k = json.load(open(PMC_TRAFFIC))["kernels"]
    fc1 = next((v for n, v in k.items() if "gemm16x_kernel<1, 1," in n), None)          # FC1 + GELU, split output
    fc2 = next((v for n, v in k.items() if "gemm16x_kernel<0, 0," in n), None)          # FC2 and out-projection share a kernel
    if not fc1 or not fc2:
        return None, None
    algo = {"fc1": {"read": 4.0 * (M * D + F * D), "write": 4.0 * M * F}, "fc2_and_out_mean": {"read": 4.0 * (M * (F + D) / 2 + (D * F + D * D) / 2) + 4.0 * M * D, "write": 4.0 * M * D}}
    detail = {"fc1": {"fetch_bytes": fc1["fetch_bytes"], "write_bytes": fc1["write_bytes"], "algorithmic_read": algo["fc1"]["read"],
                      "algorithmic_write": algo["fc1"]["write"], "fetch_over_algorithmic": fc1["fetch_bytes"] / algo["fc1"]["read"],
                      "l2_hit_rate": fc1["l2_hit_rate"]},
              "fc2_and_out_projection_mean": {"fetch_bytes": fc2["fetch_bytes"], "write_bytes": fc2["write_bytes"],
                                              "algorithmic_read": algo["fc2_and_out_mean"]["read"], "algorithmic_write": algo["fc2_and_out_mean"]["write"],
                                              "fetch_over_algorithmic": fc2["fetch_bytes"] / algo["fc2_and_out_mean"]["read"],
                                              "l2_hit_rate": fc2["l2_hit_rate"]},
              "source": "profiles/r2/pmc_traffic.json (rocprofv3 --pmc, separate FETCH_SIZE / WRITE_SIZE passes of `bench.py --layers 4`; "
                        "bytes at the L2<->fabric boundary: requests served by the 256 MB Infinity Cache are counted)"}
    return fc1["fetch_bytes"] + fc1["write_bytes"], detail


def usable_cores() -> int:
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota
    (the GPU box shows 256 logical CPUs but grants a 16-CPU quota; oversubscribing it with 256
    threads is ~1000x slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(cfg, blob, seq, n_mut, budget_s, gpu_table=None):
    """The reference algorithm on the host cores: batch-1 masked forwards (compute_fitness.py:
    489-503) through oracle/esm_oracle.py (torch CPU fp32), bounded to ~budget_s seconds: when a
    full 33-layer forward does not fit the budget, k of the 33 (identical-cost) layers are timed
    and scaled by 33/k."""
    import torch
    from oracle import esm_oracle as eo
    from proteingym_amd import synthetic
    cores = usable_cores()
    torch.set_num_threads(cores)
    ocfg, W = eo.from_arrays(arrays=synthetic.blob_to_arrays(cfg, blob), **cfg)
    toks = eo.tokenize(seq)
    n_tok, L = len(toks), ocfg["layers"]

    rows = {}

    def fwd(i, k):
        t = toks.copy()
        t[i] = eo.MASK
        t0 = time.perf_counter()
        with torch.no_grad():
            row = torch.log_softmax(eo.forward_logits(ocfg, W, t[None], n_layers=k), dim=-1)[0, i]
        dt = time.perf_counter() - t0
        if k == L:
            rows[i] = row.numpy()
        return dt

    fwd(1, 2)                                   # warm-up (threads, first-touch of two layers)
    per_layer = fwd(2, 2) / 2
    reps = 3
    k = int(max(2, min(L, budget_s / reps / max(per_layer, 1e-6))))
    if k < L:
        fwd(3, k)                               # first touch of the k layers' weights: not timed
    ts = [fwd(4 + r, k) for r in range(reps)]
    per_fwd = float(np.mean(ts)) * L / k
    assay_s = per_fwd * n_tok                   # the reference runs all L+2 positions, batch 1
    out = {"value": n_mut / assay_s, "unit": "mutants/s", "cores": cores, "kind": "port",
           "sample": f"{reps} batch-1 masked forwards at T={n_tok} through {k} of {L} layers "
                     f"(oracle/esm_oracle.py, torch CPU fp32, {cores} threads), scaled x{L}/{k}: "
                     f"{per_fwd:.3f} s/forward, x{n_tok} forwards for the assay"}
    # the full-depth oracle rows just computed double as a live parity check of the GPU table (checker only)
    parity = None
    if gpu_table is not None:
        common = [i for i in rows if not np.isnan(gpu_table[i, 0])]
        if common:
            parity = {"max_abs_err_vs_oracle": float(max(np.abs(gpu_table[i] - rows[i]).max() for i in common)),
                      "rows_compared": len(common), "tolerance": 1e-4,
                      "what": "log-prob table rows (33 values each) of the timed assay, HIP path vs CPU fp32 oracle"}
    return out, parity


def secondary(precision, budget_note="bounded: every leg is a few seconds of GPU time"):
    """Other BASELINE.json configurations and the ensemble / end-to-end rates, measured on this GPU after the headline
    (never inside its timed region).  Same synthetic conventions as the headline."""
    import tempfile
    import pandas as pd
    from proteingym_amd import esm as pesm, synthetic, dist as pdist
    out = {"note": budget_note}

    def timed(fn, reps=2):
        fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return (time.perf_counter() - t0) / reps

    seq, muts, _ = synthetic.random_assay(seed=23, L=L_BLAT, n_single=N_MUT_BLAT, n_multi=0)
    # (1) ESM-1v ensemble rate: five checkpoints per assay, plain mean (compute_fitness.py:532-537)
    cfg = dict(synthetic.ESM1V_650M)
    blob = synthetic.random_weights(cfg, seed=2)
    models = [pesm.EsmModel(cfg, blob, device=0, precision=precision) for _ in range(5)]
    assays = [pesm.Assay(m, seq, muts) for m in models]
    dt = timed(lambda: [a.run_device_only() for a in assays])
    out["esm1v_5_checkpoint_ensemble"] = {"mutants_per_s": len(muts) / dt, "ms_per_assay": dt * 1e3,
                                          "what": "BLAT-shaped assay scored with 5 resident ESM-1v-650M-shaped checkpoints"}
    for a in assays:
        a.close()
    for m in models[1:]:
        m.close()
    # (2) end to end through the product runner incl. checkpoint read, assay upload and CSV writes: the first 8 assays of the
    #     217-assay-shaped benchmark (DMS_substitutions.csv rows 0-7)
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import bench_217
    from proteingym_amd import run_benchmark
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "dms"))
        shapes = synthetic.dms_shapes()[:8]
        rows = []
        for sh in shapes:
            rng = np.random.default_rng(sh["DMS_index"])
            sq, df = bench_217.make_assay(rng, sh["seq_len"], sh["n_single"], sh["n_total"] - sh["n_single"])
            df.to_csv(os.path.join(d, "dms", sh["DMS_id"] + ".csv"), index=False)
            rows.append({"DMS_id": sh["DMS_id"], "DMS_filename": sh["DMS_id"] + ".csv", "target_seq": sq, "DMS_total_number_mutants": len(df)})
        pd.DataFrame(rows).to_csv(os.path.join(d, "map.csv"), index=False)
        ck = synthetic.save_fair_esm_checkpoint(os.path.join(d, "esm1v_synth_1.pt"), cfg, blob)
        a8 = run_benchmark.create_parser().parse_args(["--model-location", ck, "--model_type", "ESM1v", "--dms_mapping", os.path.join(d, "map.csv"),
                                                       "--dms-input", os.path.join(d, "dms"), "--dms-output", os.path.join(d, "out"), "--precision", precision])
        t0 = time.perf_counter()
        run_benchmark.main(a8)
        dt = time.perf_counter() - t0
        n = sum(sh["n_total"] for sh in shapes)
        out["run_benchmark_8_assays_end_to_end"] = {"mutants_per_s": n / dt, "seconds": dt, "mutants": n, "assays": len(shapes),
                                                    "what": "checkpoint read + upload, 8 assay uploads, masked-marginals, CSVs written (1 checkpoint)"}
    models[0].close()
    # (3) BASELINE config 3's model: ESM2-3B (36 x 2560 x 40), one BLAT-shaped assay
    cfg3 = dict(synthetic.ESM2_3B)
    n3 = sum(int(np.prod(sh)) for _, sh in synthetic.key_shapes(cfg3))
    block = (np.random.default_rng(3).random(1 << 24, dtype=np.float32) - 0.5) * 0.04      # timing only: a tiled random block
    blob3 = np.tile(block, n3 // block.size + 1)[:n3]
    m3 = pesm.EsmModel(cfg3, blob3, device=0, precision=precision)
    del blob3
    a3 = pesm.Assay(m3, seq, muts)
    dt = timed(a3.run_device_only, reps=1)
    fl = len(a3.positions) * pdist.forward_flops(a3.T, layers=36, D=2560, F=10240)
    out["esm2_3b_one_assay"] = {"mutants_per_s": len(muts) / dt, "ms_per_assay": dt * 1e3, "algorithmic_tflops": fl / dt / 1e12,
                                "what": "config 3 model shape, BLAT-shaped assay (286 positions x 288 tokens), 1 GPU"}
    a3.close()
    m3.close()
    # (4) BASELINE config 5: pseudo-ppl on a CAPSD_AAV2S-shaped slice (variable-length members, ESM2-650M shape)
    cfg5 = dict(synthetic.ESM2_650M)
    m5 = pesm.EsmModel(cfg5, synthetic.random_weights(cfg5, seed=5), device=0, precision=precision)
    lib5 = pesm.SequenceLibrary(m5, synthetic.random_indel_library(7, 735, 2)[1])
    t0 = time.perf_counter()
    lib5.score()
    dt = time.perf_counter() - t0
    st = lib5.stats()
    out["esm2_650m_pseudo_ppl_capsd_shaped"] = {"mutants_per_s": 2 / dt, "masked_forwards_per_s": st["rows"] / dt, "tokens_per_s": st["tokens"] / dt,
                                                "packing_efficiency": st["packing_efficiency"],
                                                "what": "config 5: 2 members of a 735-residue indel library = 1 470 masked forwards of ~737 tokens"}
    lib5.close()
    m5.close()
    # (5) BASELINE config 4's model: Tranception-L shape, both directions, no retrieval
    from proteingym_amd import tranception as ptr
    cfgt = dict(synthetic.TRANCEPTION_L)
    mt = ptr.TranceptionModel(cfgt, synthetic.random_tranception_weights(cfgt, seed=3), device=0)
    sq, mu, _ = synthetic.random_assay(seed=23, L=L_BLAT, n_single=512, n_multi=0)
    df = pd.DataFrame({"mutant": mu})
    df["mutated_sequence"] = df["mutant"].apply(lambda m: ptr.get_mutated_sequence(sq, m))
    df = df.drop_duplicates("mutated_sequence")
    mt.score_mutants(DMS_data=df.iloc[:32], target_seq=sq)
    t0 = time.perf_counter()
    mt.score_mutants(DMS_data=df, target_seq=sq, scoring_mirror=True)
    dt = time.perf_counter() - t0
    out["tranception_l_one_batch"] = {"mutants_per_s": len(df) / dt, "seconds": dt, "mutants": len(df),
                                      "what": "config 4 model shape (36 x 1280 x 20), 286-residue protein, both directions, no retrieval"}
    mt.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--precision", default="f16x3", choices=["fp32", "bf16", "f16x3"],
                    help="f16x3 (default) and fp32 are parity-gated (1e-4 abs vs the reference); bf16 is not")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU-baseline time budget (0 = skip)")
    ap.add_argument("--layers", type=int, default=33, help="debug only; the headline config is 33")
    ap.add_argument("--checkpoints", type=int, default=1,
                    help="score with N checkpoints per step and average (ESM-1v ensemble rate; the headline is 1)")
    ap.add_argument("--variant", type=int, default=None, help="debug: PGMI_GEMM_VARIANT tile configuration")
    ap.add_argument("--no-secondary", action="store_true", help="skip the `secondary` object (other configs; ~1 min after the headline)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")

    if args.variant is not None:
        os.environ["PGMI_GEMM_VARIANT"] = str(args.variant)
    import torch
    from proteingym_amd import build_native, esm as pesm, synthetic
    build_native.build(verbose=False)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    cfg = dict(synthetic.ESM1V_650M, layers=args.layers)
    blob = synthetic.random_weights(cfg, seed=1)                      # same "checkpoint" on every rank
    model = pesm.EsmModel(cfg, blob, device=local_rank, precision=args.precision)
    seq, muts, _ = synthetic.random_assay(seed=23 + rank, L=L_BLAT, n_single=N_MUT_BLAT, n_multi=0)
    assay = pesm.Assay(model, seq, muts, offset_idx=1)                # uploads: inputs resident in HBM
    n_mut = len(muts)
    scores_dev = torch.zeros(n_mut, dtype=torch.float64, device="cuda")
    gathered = torch.zeros(world * n_mut, dtype=torch.float64, device="cuda") if world > 1 else None
    extra = []                                                        # checkpoints 2..N of an ensemble step
    for c in range(1, args.checkpoints):
        m2 = pesm.EsmModel(cfg, synthetic.random_weights(cfg, seed=1 + c), device=local_rank, precision=args.precision)
        extra.append((m2, pesm.Assay(m2, seq, muts, offset_idx=1), torch.zeros(n_mut, dtype=torch.float64, device="cuda")))

    def step():
        assay.run_device_only(scores_dev.data_ptr())                  # whole hot path, synchronised at return
        if extra:                                                     # compute_fitness.py:532-537: plain mean
            for _, a2, buf in extra:
                a2.run_device_only(buf.data_ptr())
            scores_dev.add_(sum(buf for _, _, buf in extra)).div_(args.checkpoints)
        if world > 1:
            dist.all_gather_into_tensor(gathered, scores_dev)         # RCCL over xGMI

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    model.profile_reset()
    model.profile_enable(True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    model.profile_enable(False)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    prof = model.profile()
    if rank == 0:
        ffn_ms = prof["gemm_fc1"]["ms"] + prof["gemm_fc2"]["ms"]
        ffn_fl = prof["gemm_fc1"]["flops"] + prof["gemm_fc2"]["flops"]
        ffn_n = prof["gemm_fc1"]["launches"] + prof["gemm_fc2"]["launches"]
        achieved = ffn_fl / (ffn_ms * 1e-3) / 1e12 if ffn_ms > 0 else 0.0
        peak = PEAK_TFLOPS[args.precision]
        passes = 3 if args.precision == "f16x3" else 1          # MFMA FLOPs executed per algorithmic FLOP
        total_fl = sum(v["flops"] for v in prof.values())
        traffic, traffic_detail = ffn_traffic(args.precision, len(assay.positions) * assay.T, cfg["embed_dim"], cfg["ffn_dim"])
        kern = {k: {"ms_per_step": round(v["ms"] / args.steps, 3), "launches_per_step": v["launches"] // args.steps,
                    "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["ms"] > 0 and v["flops"] > 0 else None,
                    "gbps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] > 0 and v["bytes"] > 0 else None}
                for k, v in prof.items()}
        out = {
            "metric": "mutants scored/sec (ESM-1v 650M masked-marginal)",
            "value": world * n_mut * args.steps / dt,
            "unit": "mutants/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32", "bf16": "bf16",
                      "f16x3": "f16x3 (fp32 operands split into 2 fp16 planes, 3 fp16 MFMAs per product, fp32 accumulate)"}[args.precision],
            "data": "synthetic",
            "config": {"workload": "ESM-1v 650M (33x1280, 20 heads, FFN 5120) masked-marginals, one "
                                   "BLAT_ECOLX_Stiffler_2015-shaped assay per GPU per step (L=286, T=288, "
                                   f"{len(assay.positions)} masked positions run, {n_mut} single mutants), "
                                   f"{args.checkpoints} checkpoint{'s averaged (ensemble rate)' if args.checkpoints > 1 else ''}; "
                                   "assays shard over ranks + RCCL all_gather of score vectors",
                       "precision": args.precision, "layers": args.layers,
                       "last_layer": ("after its attention the last layer runs on the masked row of every sequence only -- the one row "
                                      "masked-marginals reads (class kept_rows); scores bit-identical to the full evaluation"
                                      if os.environ.get("PGMI_KEEP_ROWS", "1") != "0" else "all rows (PGMI_KEEP_ROWS=0)"),
                       "positions_run": int(len(assay.positions)), "tokens_per_step": int(len(assay.positions) * assay.T)},
            "roofline": {"bound": "mfma", "kernel": "gemm (fc1+GELU, fc2+residual)", "achieved": achieved,
                         "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "avg_launch_ms": ffn_ms / max(ffn_n, 1), "traffic": traffic, "traffic_detail": traffic_detail,
                         "mfma_passes": passes, "mfma_util": passes * achieved / peak,
                         "vs_fp32_mfma_peak": achieved / PEAK_TFLOPS["fp32"],
                         "note": "achieved = algorithmic FLOPs (2*M*N*K per GEMM) / HIP-event time of the FFN GEMM "
                                 "launches in the timed region; f16x3 issues 3 fp16 MFMAs per product block, so MFMA "
                                 "pipe utilisation is mfma_util = 3*achieved/peak",
                         "whole_step_tflops": total_fl / dt / 1e12},
            "kernels": kern,
        }
        if world == 1 and args.cpu_seconds > 0:
            _, gpu_table = assay.run(want_table=True)                 # outside the timed region
            for m2, a2, _ in extra:
                a2.close()
                m2.close()
            assay.close()
            model.close()
            out["cpu_baseline"], out["parity"] = cpu_baseline(cfg, blob, seq, n_mut, args.cpu_seconds, gpu_table)
            out["cpu_baseline"]["reference_vs_port"] = ("kind 'port': /root/reference does not exist on the GPU box; in the build container the "
                                                        "unmodified reference model and this port were timed side by side on the same input "
                                                        "(profiles/r2/cpu_reference_vs_port.json): same s/forward within noise")
            if not args.no_secondary and args.layers == 33:
                del blob
                import contextlib
                try:                                                 # the runner and scorer print progress: keep stdout to the ONE JSON line
                    with contextlib.redirect_stdout(sys.stderr):
                        out["secondary"] = secondary(args.precision)
                except Exception as e:                               # a secondary leg must never take the headline line down
                    out["secondary"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
