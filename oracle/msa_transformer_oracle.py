"""TEST INFRASTRUCTURE ONLY -- CPU restatement of ProteinGym's MSA Transformer zero-shot path.

Only ``tests/`` (and benchmark CPU baselines) may import this file, as the checker.  The product
path (``proteingym_amd``) never imports it.

What is restated (file:line relative to /root/reference/proteingym/baselines/esm):
  * checkpoint upgrade (row/column swap)     esm/pretrained.py:107-121
  * alphabet / MSA batch converter           esm/data.py:158-164, 300-334 (<cls> prepended, no <eos>)
  * MSATransformer.forward                   esm/model/msa_transformer.py:146-205
  * AxialTransformerLayer                    esm/modules.py:145-232, NormalizedResidualBlock :374-406,
                                             FeedForwardNetwork :409-432
  * RowSelfAttention (tied)                  esm/axial_attention.py:33-168 (scaling :72-74, weights :108-139,
                                             update :141-156)
  * ColumnSelfAttention                      esm/axial_attention.py:171-297
  * MSA sampling + masked-marginals loop     compute_fitness.py:26-73, 360-399; seeds ensemble :538-542
  * MSA_processing (ESM variant)             proteingym/utils/msa_utils.py:24-258 (same EVE pre-processing as
                                             the Tranception copy restated in tranception_oracle.py)

Arithmetic on torch CPU tensors (array library only), float32 by default / float64 for noise floors.
PINNING: tests/golden/make_golden_msa_transformer.py ran the unmodified reference model and CLI (through
oracle/ref_harness.py) and froze their outputs; tests/test_oracle_pinning.py checks this file against them.
"""
from __future__ import annotations

import argparse
import math
import random

import numpy as np
import torch

from . import esm_oracle as eo
from . import tranception_oracle as to

CLS, PAD, MASK = eo.TOK_TO_IDX["<cls>"], eo.TOK_TO_IDX["<pad>"], eo.TOK_TO_IDX["<mask>"]


def load_checkpoint(path, dtype=torch.float32):
    torch.serialization.add_safe_globals([argparse.Namespace])
    data = torch.load(str(path), map_location="cpu", weights_only=False)
    a = data["args"]
    if a.arch != "msa_transformer":
        raise ValueError("not an msa_transformer checkpoint")
    prs1 = lambda s: "".join(s.split("encoder.")[1:] if "encoder" in s else s)
    prs2 = lambda s: "".join(s.split("sentence_encoder.")[1:] if "sentence_encoder" in s else s)
    prs3 = lambda s: s.replace("row", "column") if "row" in s else s.replace("column", "row")
    sd = {prs1(prs2(prs3(k))): v for k, v in data["model"].items()}                  # pretrained.py:110-116
    if "lm_head.weight" in sd:                                                     # tied (msa_transformer.py:141-145)
        sd["embed_tokens.weight"] = sd["lm_head.weight"]
    cfg = dict(arch="msa_transformer", layers=int(a.encoder_layers), embed_dim=int(a.encoder_embed_dim),
               ffn_dim=int(a.encoder_ffn_embed_dim), heads=int(a.encoder_attention_heads),
               max_positions=int(a.max_positions), embed_positions_msa=bool(getattr(a, "embed_positions_msa", False)))
    W = {k: v.to(dtype) for k, v in sd.items() if not k.startswith("contact_head")}
    return cfg, W


def tokenize_msa(msa):
    """MSABatchConverter for one MSA (data.py:300-334): [R, C+1] with <cls> first, no <eos>."""
    lens = {len(s) for _, s in msa}
    if len(lens) != 1:
        raise RuntimeError("Received unaligned sequences for input to MSA, all sequence lengths must be equal.")
    return np.array([[CLS] + [eo.get_idx(c) for c in s] for _, s in msa], dtype=np.int64)


def forward_logits(cfg, W, tokens: np.ndarray) -> torch.Tensor:
    """tokens [R, C] (one MSA, batch of 1) -> logits [R, C, 33]."""
    tok = torch.as_tensor(np.asarray(tokens), dtype=torch.long)
    R, C = tok.shape
    D, H = cfg["embed_dim"], cfg["heads"]
    dh = D // H
    dtype = W["embed_tokens.weight"].dtype
    pad = tok.eq(PAD)
    if pad.any():
        raise NotImplementedError("oracle restates the unpadded MSA path (masked-marginals never pads)")
    x = W["embed_tokens.weight"][tok]                                               # :158
    positions = torch.cumsum((~pad).long(), dim=1) * (~pad).long() + PAD           # modules.py:262-270
    x = x + W["embed_positions.weight"][positions]                                 # :159
    if cfg["embed_positions_msa"]:
        if R > 1024:
            raise RuntimeError("Using model with MSA position embedding trained on maximum MSA depth of 1024, "
                               f"but received {R} alignments.")
        x = x + W["msa_position_embedding"][0, :R]                                  # [R,1,D or 1] broadcast (:160-166)
    x = eo._layer_norm(x, W["emb_layer_norm_before.weight"], W["emb_layer_norm_before.bias"])   # :168
    for i in range(cfg["layers"]):
        p = f"layers.{i}."
        # -- tied row attention (axial_attention.py:72-74,108-156) --
        q0 = p + "row_self_attention."
        h = eo._layer_norm(x, W[q0 + "layer_norm.weight"], W[q0 + "layer_norm.bias"])
        q = (h @ W[q0 + "layer.q_proj.weight"].T + W[q0 + "layer.q_proj.bias"]).view(R, C, H, dh)
        k = (h @ W[q0 + "layer.k_proj.weight"].T + W[q0 + "layer.k_proj.bias"]).view(R, C, H, dh)
        v = (h @ W[q0 + "layer.v_proj.weight"].T + W[q0 + "layer.v_proj.bias"]).view(R, C, H, dh)
        q = q * ((dh ** -0.5) / math.sqrt(R))
        s = torch.einsum("rihd,rjhd->hij", q, k)
        a = s.softmax(-1)
        ctx = torch.einsum("hij,rjhd->rihd", a, v).reshape(R, C, D)
        x = x + ctx @ W[q0 + "layer.out_proj.weight"].T + W[q0 + "layer.out_proj.bias"]
        # -- column attention (axial_attention.py:232-275) --
        q0 = p + "column_self_attention."
        h = eo._layer_norm(x, W[q0 + "layer_norm.weight"], W[q0 + "layer_norm.bias"])
        if R == 1:
            vv = h @ W[q0 + "layer.v_proj.weight"].T + W[q0 + "layer.v_proj.bias"]
            x = x + vv @ W[q0 + "layer.out_proj.weight"].T + W[q0 + "layer.out_proj.bias"]
        else:
            q = (h @ W[q0 + "layer.q_proj.weight"].T + W[q0 + "layer.q_proj.bias"]).view(R, C, H, dh) * (dh ** -0.5)
            k = (h @ W[q0 + "layer.k_proj.weight"].T + W[q0 + "layer.k_proj.bias"]).view(R, C, H, dh)
            v = (h @ W[q0 + "layer.v_proj.weight"].T + W[q0 + "layer.v_proj.bias"]).view(R, C, H, dh)
            s = torch.einsum("ichd,jchd->hcij", q, k)
            a = s.softmax(-1)
            ctx = torch.einsum("hcij,jchd->ichd", a, v).reshape(R, C, D)
            x = x + ctx @ W[q0 + "layer.out_proj.weight"].T + W[q0 + "layer.out_proj.bias"]
        # -- feed forward --
        q0 = p + "feed_forward_layer."
        h = eo._layer_norm(x, W[q0 + "layer_norm.weight"], W[q0 + "layer_norm.bias"])
        h = eo._gelu(h @ W[q0 + "layer.fc1.weight"].T + W[q0 + "layer.fc1.bias"])
        x = x + h @ W[q0 + "layer.fc2.weight"].T + W[q0 + "layer.fc2.bias"]
    x = eo._layer_norm(x, W["emb_layer_norm_after.weight"], W["emb_layer_norm_after.bias"])
    h = eo._gelu(x @ W["lm_head.dense.weight"].T + W["lm_head.dense.bias"])
    h = eo._layer_norm(h, W["lm_head.layer_norm.weight"], W["lm_head.layer_norm.bias"])
    return h @ W["lm_head.weight"].T + W["lm_head.bias"]


def masked_marginals_table(cfg, W, tokens: np.ndarray, seq_len: int, positions=None) -> np.ndarray:
    """compute_fitness.py:380-394: mask column i of the first row, forward the whole MSA (cropped to the
    optimal 1024 window when wider), keep log_softmax(logits)[0, i - start]."""
    R, T = tokens.shape
    table = np.full((T, 33), np.nan, dtype=np.float64 if W["embed_tokens.weight"].dtype == torch.float64 else np.float32)
    for i in (range(T) if positions is None else positions):
        t = tokens.copy()
        t[0, i] = MASK
        start = 0
        if T > 1024:
            start, end = eo.get_optimal_window(i, seq_len + 2, 1024)
            t = t[:, start:end]
        with torch.no_grad():
            lp = torch.log_softmax(forward_logits(cfg, W, t), dim=-1)
        table[i] = lp[0, i - start].numpy()
    return table


# ---- MSA handling (compute_fitness.py:26-98) --------------------------------------------------------
class ProcessedMSA:
    """The attributes of MSA_processing that sample_msa reads (utils/msa_utils.py:24-258)."""

    def __init__(self, MSA_location, weights=None, theta=0.2):
        AA = "ACDEFGHIKLMNPQRSTVWY"
        raw, order, name = {}, [], ""
        with open(MSA_location) as f:
            for line in f:
                line = line.rstrip()
                if line.startswith(">"):
                    name = line
                    if name not in raw:
                        raw[name] = ""
                        order.append(name)
                else:
                    raw[name] += line
        self.focus_seq_name = order[0]
        seqs = {n: raw[n].replace(".", "-").upper() for n in order}
        keep = [i for i, c in enumerate(seqs[self.focus_seq_name]) if c != "-"]
        seqs = {n: "".join(s[i] for i in keep) for n, s in seqs.items()}
        seqs = {n: s for n, s in seqs.items() if sum(c == "-" for c in s) / len(s) <= 0.5}
        # threshold_focus_cols_frac_gaps = 1.0: every column stays upper-case
        self.raw_seq_name_to_sequence = dict(seqs)
        focus = seqs[self.focus_seq_name]
        cols = [j for j, c in enumerate(focus) if c == c.upper() and c != "-"]
        trimmed = {n: "".join(s[j].upper() for j in cols) for n, s in seqs.items()}
        self.seq_name_to_sequence = {n: s for n, s in trimmed.items() if all((c in AA or c == "-") for c in s)}
        w = to.eve_sequence_weights(MSA_location, theta=theta) if weights is None else None
        names = list(self.seq_name_to_sequence.keys())
        if weights is not None:
            assert len(weights) == len(names)
            self.seq_name_to_weight = {n: weights[i] for i, n in enumerate(names)}
        else:
            self.seq_name_to_weight = {n: w[n] for n in names}


def sample_msa(msa: ProcessedMSA, nseq: int, random_seed: int):
    """'sequence-reweighting' branch (compute_fitness.py:41-69)."""
    random.seed(random_seed)
    out = [(msa.focus_seq_name, msa.raw_seq_name_to_sequence[msa.focus_seq_name])]
    non_wt_w = np.array([w for k, w in msa.seq_name_to_weight.items() if k != msa.focus_seq_name])
    non_wt = [(k, s) for k, s in msa.seq_name_to_sequence.items() if k != msa.focus_seq_name]
    non_wt_w = non_wt_w / non_wt_w.sum()
    if len(non_wt) > 0:
        out.extend(random.choices(non_wt, weights=non_wt_w, k=nseq - 1))
    return [(d, s.upper()) for d, s in out]


def score_dms(ckpt, msa_path, weights, sequence, mutants, offset_idx, seeds, nseq, dtype=torch.float32):
    """Per-seed score columns + their mean (compute_fitness.py:364-399, 538-542)."""
    cfg, W = load_checkpoint(ckpt, dtype)
    pm = ProcessedMSA(msa_path, weights=weights)
    cols = {}
    for seed in seeds:
        tokens = tokenize_msa(sample_msa(pm, nseq, seed))
        table = masked_marginals_table(cfg, W, tokens, len(sequence))
        cols[f"seed{seed}"] = np.array([eo.label_row(m, sequence, table, offset_idx) for m in mutants])
    cols["ensemble"] = sum(cols[f"seed{s}"] for s in seeds) / len(seeds)
    return cols


def from_arrays(layers, embed_dim, heads, ffn_dim, max_positions, arrays, dtype=torch.float32, **_):
    cfg = dict(arch="msa_transformer", layers=layers, embed_dim=embed_dim, ffn_dim=ffn_dim, heads=heads,
               max_positions=max_positions, embed_positions_msa=True)
    W = {k: torch.as_tensor(np.asarray(v)).to(dtype) for k, v in arrays.items()}
    return cfg, W
