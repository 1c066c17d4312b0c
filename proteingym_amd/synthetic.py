"""Synthetic inputs of the reference's shapes (no network: no real checkpoints or DMS files).

* random-weight ESM-1v / ESM2 models of a given architecture, as the flat ABI blob
  (include/pgmi.h) or as a fair-esm ``.pt`` file (layouts: SURVEY.md Appendix A,
  /root/reference/proteingym/baselines/esm/esm/pretrained.py:85-99,162-181);
* DMS assays shaped like rows of reference_files/DMS_substitutions.csv
  (``data/dms_substitutions_shapes.csv`` holds seq_len and mutant counts of the 217 assays;
  SURVEY.md section 8d describes the generator: positions uniform over [1,L], target amino acid
  uniform over the 19 non-wild-type letters, multi-mutants of depth 2-5).
"""
from __future__ import annotations

import argparse
import os
from typing import Dict, List, Tuple

import numpy as np

from . import _lib
from .esm import expected_keys

AA = "ACDEFGHIKLMNPQRSTVWY"

ESM1V_650M = dict(arch=_lib.ARCH_ESM1B, layers=33, embed_dim=1280, heads=20, ffn_dim=5120,
                  max_positions=1024, token_dropout=1, emb_layer_norm_before=0)
ESM2_650M = dict(arch=_lib.ARCH_ESM2, layers=33, embed_dim=1280, heads=20, ffn_dim=5120,
                 max_positions=0, token_dropout=1, emb_layer_norm_before=0)
ESM2_3B = dict(arch=_lib.ARCH_ESM2, layers=36, embed_dim=2560, heads=40, ffn_dim=10240,
               max_positions=0, token_dropout=1, emb_layer_norm_before=0)
ESM2_35M = dict(arch=_lib.ARCH_ESM2, layers=12, embed_dim=480, heads=20, ffn_dim=1920,        # head_dim 24, embed_dim not a multiple of 64
                max_positions=0, token_dropout=1, emb_layer_norm_before=0)                       # (pretrained.py:355-360; /root/reference/config.json:18)
ESM2_15B = dict(arch=_lib.ARCH_ESM2, layers=48, embed_dim=5120, heads=40, ffn_dim=20480,      # head_dim 128 (pretrained.py:387-394)
                max_positions=0, token_dropout=1, emb_layer_norm_before=0)


def key_shapes(cfg) -> List[Tuple[str, Tuple[int, ...]]]:
    D, F, V = cfg["embed_dim"], cfg["ffn_dim"], 33
    out = []
    for k in expected_keys(cfg):
        leaf = k.split(".")[-2] + "." + k.split(".")[-1]
        if k == "embed_tokens.weight":
            s = (V, D)
        elif k == "embed_positions.weight":
            s = (cfg["max_positions"] + 2, D)
        elif k == "lm_head.bias":
            s = (V,)
        elif leaf == "fc1.weight":
            s = (F, D)
        elif leaf == "fc1.bias":
            s = (F,)
        elif leaf == "fc2.weight":
            s = (D, F)
        elif k.endswith("_proj.weight") or k == "lm_head.dense.weight":
            s = (D, D)
        else:
            s = (D,)
        out.append((k, s))
    return out


def random_weights(cfg, seed: int, embed_std: float = 0.25) -> np.ndarray:
    """Flat fp32 blob in ABI order.  Linear layers U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (the
    nn.Linear default scale), LayerNorm gamma 1+0.1N / beta 0.05N, embeddings N(0, embed_std^2)."""
    rng = np.random.default_rng(seed)
    n = sum(int(np.prod(s)) for _, s in key_shapes(cfg))
    blob = np.empty(n, dtype=np.float32)
    o = 0
    for k, s in key_shapes(cfg):
        m = int(np.prod(s))
        v = blob[o:o + m]
        if "layer_norm" in k:
            v[:] = rng.standard_normal(m, dtype=np.float32)
            if k.endswith("weight"):
                v *= 0.1
                v += 1.0
            else:
                v *= 0.05
        elif k.startswith("embed_"):
            v[:] = rng.standard_normal(m, dtype=np.float32)
            v *= embed_std
        else:
            fan_in = s[-1] if len(s) == 2 else {"fc1.bias": cfg["embed_dim"], "fc2.bias": cfg["ffn_dim"]}.get(
                k.split(".")[-2] + ".bias", cfg["embed_dim"])
            b = 1.0 / np.sqrt(fan_in)
            v[:] = rng.random(m, dtype=np.float32)
            v *= 2 * b
            v -= b
        o += m
    return blob


def outlier_weights(cfg, seed: int, n_outlier: int = 6, magnitude: float = 3000.0, max_ln_gain: float = 30.0,
                    tail_df: float = 3.0, embed_std: float = 0.15, inject_layer: int = 2) -> np.ndarray:
    """A checkpoint with the features real ESM / LLM checkpoints are known for and ``random_weights`` lacks (flat ABI blob):
    * heavy-tailed Linear weights: Student-t(tail_df) scaled to nn.Linear's default variance (max / median of a matrix ~ 10^2-10^3,
      so the per-tensor power-of-two pre-scale of the f16x3 weight planes leaves most entries far below the largest);
    * ``n_outlier`` "massive" residual channels: the learned positions (ESM-1b/1v) carry +-magnitude there from the first layer on,
      and layer ``inject_layer``'s FC2 bias adds +-magnitude as the feed-forward of real models does -- every LayerNorm after that
      sees a row whose variance is a few channels, every residual add works at fp32's resolution AT that magnitude;
    * LayerNorm gains up to ``max_ln_gain`` on a handful of channels of every LayerNorm (the outlier channels among them).
    The tied embedding / LM-head rows stay N(0, embed_std^2): the log-likelihood ratios keep the size real checkpoints produce."""
    rng = np.random.default_rng(seed)
    blob = random_weights(cfg, seed=seed, embed_std=embed_std)
    arrs = blob_to_arrays(cfg, blob)                       # views into blob
    D = cfg["embed_dim"]
    chans = rng.choice(D, size=n_outlier, replace=False)
    signs = rng.choice([-1.0, 1.0], size=n_outlier)
    for k, v in arrs.items():
        if k == "lm_head.weight" or k.startswith("embed_tokens"):
            continue
        if v.ndim == 2 and not k.startswith("embed_"):
            b = 1.0 / np.sqrt(v.shape[1])                  # U(-b, b) has variance b^2 / 3; t(df) has df / (df - 2)
            v[:] = (rng.standard_t(tail_df, size=v.shape) * (b / np.sqrt(3.0) / np.sqrt(tail_df / (tail_df - 2.0)))).astype(np.float32)
        elif "layer_norm" in k and k.endswith("weight"):
            hot = np.concatenate([chans[:3], rng.choice(D, size=5, replace=False)])
            v[hot] = rng.uniform(0.3 * max_ln_gain, max_ln_gain, size=hot.size).astype(np.float32)
    if "embed_positions.weight" in arrs:
        pe = arrs["embed_positions.weight"]
        pe[:, chans] = (signs * magnitude * (1.0 + 0.1 * rng.standard_normal((pe.shape[0], n_outlier)))).astype(np.float32)
    arrs[f"layers.{inject_layer}.fc2.bias"][chans] += (signs * magnitude).astype(np.float32)
    return blob


def blob_to_arrays(cfg, blob: np.ndarray) -> Dict[str, np.ndarray]:
    out, o = {}, 0
    for k, s in key_shapes(cfg):
        m = int(np.prod(s))
        out[k] = blob[o:o + m].reshape(s)
        o += m
    assert o == blob.size
    out["lm_head.weight"] = out["embed_tokens.weight"]     # tied (esm1.py:101-105)
    return out


def save_fair_esm_checkpoint(path: str, cfg, blob: np.ndarray):
    """Write the blob as a fair-esm v1 (ESM-1b/1v) or v2 (ESM2; file stem must start with
    'esm2') checkpoint that both the reference loader and this package read."""
    import torch
    arrs = blob_to_arrays(cfg, blob)
    model = {}
    for k, v in arrs.items():
        pref = "encoder." if k.startswith("lm_head") else "encoder.sentence_encoder."
        model[pref + k] = torch.from_numpy(np.ascontiguousarray(v)).clone()
    # like real fair-esm files, the tied tensors share storage
    model["encoder.lm_head.weight"] = model["encoder.sentence_encoder.embed_tokens.weight"]
    stem = os.path.basename(path).split(".")[0]
    if cfg["arch"] == _lib.ARCH_ESM2:
        assert stem.startswith("esm2"), "ESM2 checkpoints are dispatched by file stem (pretrained.py:187)"
        dh = cfg["embed_dim"] // cfg["heads"]                                          # rotary_embedding.py:39-41: one frequency per pair of head dims
        inv = (1.0 / (10000 ** (np.arange(0, dh, 2, dtype=np.float32) / dh))).astype(np.float32)
        for i in range(cfg["layers"]):
            model[f"encoder.sentence_encoder.layers.{i}.self_attn.rot_emb.inv_freq"] = torch.from_numpy(inv.copy())
        c = argparse.Namespace(encoder_layers=cfg["layers"], encoder_embed_dim=cfg["embed_dim"],
                               encoder_attention_heads=cfg["heads"], token_dropout=bool(cfg["token_dropout"]))
        torch.save({"cfg": {"model": c}, "model": model}, path)
    else:
        assert not stem.startswith("esm2")
        a = argparse.Namespace(arch="roberta_large", encoder_layers=cfg["layers"],
                               encoder_embed_dim=cfg["embed_dim"], encoder_ffn_embed_dim=cfg["ffn_dim"],
                               encoder_attention_heads=cfg["heads"], max_positions=cfg["max_positions"],
                               token_dropout=bool(cfg["token_dropout"]))
        torch.save({"args": a, "model": model}, path)
    return path


def random_sequence(rng, L: int) -> str:
    return "".join(rng.choice(list(AA), size=L))


def random_assay(seed: int, L: int, n_single: int, n_multi: int, offset: int = 1):
    """(sequence, mutants list, DMS_score) shaped like one DMS_substitutions row."""
    rng = np.random.default_rng(seed)
    seq = random_sequence(rng, L)
    aa = np.array(list(AA))
    seq_arr = np.array(list(seq))

    def one(p):
        choices = aa[aa != seq_arr[p]]
        return f"{seq_arr[p]}{p + offset}{choices[rng.integers(0, 19)]}"

    muts = [one(int(p)) for p in rng.integers(0, L, size=n_single)]
    for _ in range(n_multi):
        k = int(rng.integers(2, 6))
        ps = np.sort(rng.choice(L, size=min(k, L), replace=False))
        muts.append(":".join(one(int(p)) for p in ps))
    score = rng.standard_normal(len(muts))
    return seq, muts, score


def dms_shapes():
    import csv
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "dms_substitutions_shapes.csv")
    with open(path) as f:
        return [dict(DMS_index=i, DMS_id=r["DMS_id"], seq_len=int(r["seq_len"]),
                     n_total=int(r["DMS_total_number_mutants"]), n_single=int(r["DMS_number_single_mutants"]),
                     n_multi=int(r["DMS_number_multiple_mutants"])) for i, r in enumerate(csv.DictReader(f))]


def indel_shapes():
    """seq_len and mutant count of the 66 indel assays (reference_files/DMS_indels.csv; 287 207 mutants)."""
    import csv
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "dms_indels_shapes.csv")
    with open(path) as f:
        return [dict(DMS_index=i, DMS_id=r["DMS_id"], seq_len=int(r["seq_len"]), n_total=int(r["DMS_total_number_mutants"]))
                for i, r in enumerate(csv.DictReader(f))]


def random_indel_library(seed: int, L: int, n: int, max_edit: int = 3):
    """(wild type, n mutated sequences): each member deletes or inserts 1..max_edit residues at a random site, like the
    single-event indel libraries of DMS_indels.csv; lengths spread over L-max_edit .. L+max_edit."""
    rng = np.random.default_rng(seed)
    wt = random_sequence(rng, L)
    out = []
    for _ in range(n):
        k = int(rng.integers(1, max_edit + 1))
        if rng.random() < 0.5 and L - k > 4:
            p = int(rng.integers(0, L - k + 1))
            out.append(wt[:p] + wt[p + k:])
        else:
            p = int(rng.integers(0, L + 1))
            out.append(wt[:p] + random_sequence(rng, k) + wt[p:])
    return wt, out


# ---- Tranception ------------------------------------------------------------------------------------
TRANCEPTION_L = dict(arch=_lib.ARCH_TRANCEPTION, layers=36, embed_dim=1280, heads=20, ffn_dim=5120, vocab=25,
                     max_positions=1024, ln_eps=1e-5)     # paper "Large"; real dims come from the checkpoint's config.json


def tranception_key_shapes(cfg):
    from .tranception import expected_keys
    D, F, V = cfg["embed_dim"], cfg["ffn_dim"], cfg["vocab"]
    out = []
    for k in expected_keys(cfg["layers"]):
        if k in ("transformer.wte.weight", "lm_head.weight"):
            s = (V, D)
        elif k.endswith("c_attn.weight"):
            s = (D, 3 * D)
        elif k.endswith("c_attn.bias"):
            s = (3 * D,)
        elif "depthwiseconv" in k:
            ksz = (3, 5, 7)[int(k.split("depthwiseconv.")[1][0])]
            s = (64, ksz) if k.endswith("weight") else (64,)
        elif k.endswith("attn.c_proj.weight"):
            s = (D, D)
        elif k.endswith("mlp.c_fc.weight"):
            s = (D, F)
        elif k.endswith("mlp.c_fc.bias"):
            s = (F,)
        elif k.endswith("mlp.c_proj.weight"):
            s = (F, D)
        else:
            s = (D,)
        out.append((k, s))
    return out


def random_tranception_weights(cfg, seed: int, embed_std: float = 0.3) -> np.ndarray:
    """Flat fp32 blob in the Tranception ABI order; Conv1D weights N(0, 1/fan_in), tied lm_head."""
    rng = np.random.default_rng(seed)
    parts = []
    wte = None
    for k, s in tranception_key_shapes(cfg):
        m = int(np.prod(s))
        if k == "lm_head.weight":
            v = wte.copy()
        elif k.endswith("wte.weight"):
            v = rng.standard_normal(m, dtype=np.float32) * embed_std
            wte = v
        elif "ln_" in k:
            v = rng.standard_normal(m, dtype=np.float32) * (0.1 if k.endswith("weight") else 0.05)
            if k.endswith("weight"):
                v += 1.0
        elif "depthwiseconv" in k:
            v = rng.standard_normal(m, dtype=np.float32) * (0.4 if k.endswith("weight") else 0.05)
        elif k.endswith(".bias"):
            v = rng.standard_normal(m, dtype=np.float32) * 0.02
        else:
            v = rng.standard_normal(m, dtype=np.float32) / np.float32(np.sqrt(s[0]))
        parts.append(v.astype(np.float32))
    return np.concatenate(parts)


def tranception_blob_to_arrays(cfg, blob):
    out, o = {}, 0
    for k, s in tranception_key_shapes(cfg):
        m = int(np.prod(s))
        a = blob[o:o + m].reshape(s)
        if k.endswith("conv.weight"):
            a = a.reshape(s[0], 1, s[1])
        out[k] = a
        o += m
    assert o == blob.size
    return out


# ---- MSA Transformer (esm_msa1b_t12_100M_UR50S shape: 12 x 768, 12 heads, F = 3072) ------------------
MSA_1B = dict(arch=4, layers=12, embed_dim=768, heads=12, ffn_dim=3072, max_positions=1024, embed_positions_msa=True)


def random_msa_transformer_arrays(cfg, seed: int, embed_std: float = 0.25) -> Dict[str, np.ndarray]:
    """State-dict-keyed random arrays (names after the loader's row/column swap) for
    ``msa_transformer.pack_state_dict`` and the oracle's ``from_arrays``."""
    from . import msa_transformer as pmsa
    rng = np.random.default_rng(seed)
    D, F = cfg["embed_dim"], cfg["ffn_dim"]
    out = {}
    for k in pmsa.expected_keys(cfg):
        if k == "embed_tokens.weight":
            out[k] = (rng.standard_normal((33, D)) * embed_std).astype(np.float32)
        elif k == "embed_positions.weight":
            out[k] = (rng.standard_normal((cfg["max_positions"] + 2, D)) * embed_std).astype(np.float32)
        elif k == "msa_position_embedding":
            out[k] = (rng.standard_normal((1, 1024, 1, D)) * 0.1).astype(np.float32)
        elif "layer_norm" in k:
            out[k] = (1.0 + 0.1 * rng.standard_normal(D)).astype(np.float32) if k.endswith("weight") \
                else (0.05 * rng.standard_normal(D)).astype(np.float32)
        elif k == "lm_head.bias":
            out[k] = (0.1 * rng.standard_normal(33)).astype(np.float32)
        else:
            if "fc1" in k:
                shape, fan = ((F, D) if k.endswith("weight") else (F,)), D
            elif "fc2" in k:
                shape, fan = ((D, F) if k.endswith("weight") else (D,)), F
            else:
                shape, fan = ((D, D) if k.endswith("weight") else (D,)), D
            b = 1.0 / np.sqrt(fan)
            out[k] = ((rng.random(shape) * 2 - 1) * b).astype(np.float32)
    out["lm_head.weight"] = out["embed_tokens.weight"]
    return out
