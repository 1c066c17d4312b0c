"""TEST INFRASTRUCTURE ONLY -- CPU restatement ("port") of ProteinGym's ESM zero-shot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this file, and only as the checker / the timed CPU baseline.  The product path
(``proteingym_amd``) never imports it and has no CPU fallback.

What is restated (all file:line are relative to /root/reference):
  * vocabulary + tokenisation          proteingym/baselines/esm/esm/data.py:92-174,262-297
                                       esm/constants.py:8
  * checkpoint upgrade                 esm/pretrained.py:67-99,162-218
  * ESM-1b/1v forward                  esm/model/esm1.py:116-177
  * ESM2 forward                       esm/model/esm2.py:76-130
  * transformer layer                  esm/modules.py:120-142  (gelu :17-24)
  * attention                          esm/multihead_attention.py:159-405 (softmax fp32 :18-22)
  * rotary                             esm/rotary_embedding.py:11-69
  * learned positions                  esm/modules.py:254-271
  * LM head                            esm/modules.py:322-328
  * masked-marginals / wt-marginals / pseudo-ppl / label_row / ensemble
                                       proteingym/baselines/esm/compute_fitness.py:240-279,426-543
  * optimal window                     proteingym/utils/scoring_utils.py:43-52

The float arithmetic of the reference lives in PyTorch (third-party; reference pins
torch==1.13.1, environments/proteingym_env.txt:113; this image has 2.10).  The forward below
is written from the formulae, op by op, on torch CPU tensors (torch is only the array library
here, as numpy would be), in float32 (default, the reference's precision) or float64 (truth
for noise-floor measurements).  Integer/host logic is plain Python/numpy.

PINNING: the reference has no tests or golden vectors for this path.  The
oracle is pinned against the reference *itself*: ``tests/test_oracle_vs_reference.py`` runs the
unmodified reference (oracle/ref_harness.py) in this container, and
``tests/golden/make_golden.py`` froze reference outputs into ``tests/golden/*.npz`` which
travel to the GPU box.
"""
from __future__ import annotations

import argparse
import math
import os
import re

import numpy as np
import torch

# --- vocabulary (esm/constants.py:8, esm/data.py:92-174: "roberta_large"/"ESM-1b") -------
_STANDARD = ['L', 'A', 'G', 'V', 'S', 'E', 'R', 'T', 'I', 'D', 'P', 'K', 'Q', 'N', 'F', 'Y',
             'M', 'H', 'W', 'C', 'X', 'B', 'U', 'Z', 'O', '.', '-']
ALL_TOKS = ['<cls>', '<pad>', '<eos>', '<unk>'] + _STANDARD
while len(ALL_TOKS) % 8:                      # data.py:110-111 pads to a multiple of 8
    ALL_TOKS.append(f"<null_{len(ALL_TOKS) - 30}>")
ALL_TOKS.append('<mask>')
TOK_TO_IDX = {t: i for i, t in enumerate(ALL_TOKS)}
CLS, PAD, EOS, UNK, MASK = 0, 1, 2, 3, 32
VOCAB = len(ALL_TOKS)
assert VOCAB == 33 and TOK_TO_IDX['<mask>'] == 32 and ALL_TOKS[31] == '<null_1>'


def get_idx(tok: str) -> int:
    """Alphabet.get_idx (data.py:125-126): unknown -> <unk>."""
    return TOK_TO_IDX.get(tok, UNK)


def tokenize(seq: str) -> np.ndarray:
    """BatchConverter for one sequence (data.py:262-297): cls + residues + eos, int64.  Alphabet.encode (data.py:253-254)
    indexes tok_to_idx directly: a character outside the vocabulary raises KeyError (it is NOT mapped to <unk>; only
    get_idx does that).  Residue letters only: literal special tokens inside the text are not handled here."""
    return np.array([CLS] + [TOK_TO_IDX[c] for c in seq] + [EOS], dtype=np.int64)


# --- checkpoint (esm/pretrained.py) ------------------------------------------------------
def load_checkpoint(path: str, dtype=torch.float32):
    """Returns (cfg, weights).  cfg keys: arch ('esm1b'|'esm2'), layers, embed_dim, ffn_dim,
    heads, token_dropout, emb_layer_norm_before, max_positions."""
    torch.serialization.add_safe_globals([argparse.Namespace])
    data = torch.load(str(path), map_location="cpu", weights_only=False)
    stem = os.path.basename(str(path)).split(".")[0]
    if stem.startswith("esm2"):                                   # pretrained.py:187
        c = data["cfg"]["model"]
        pat = re.compile("^" + "|".join(["encoder.sentence_encoder.", "encoder."]))
        sd = {pat.sub("", k): v for k, v in data["model"].items()}   # pretrained.py:163-168
        cfg = dict(arch="esm2", layers=int(c.encoder_layers), embed_dim=int(c.encoder_embed_dim),
                   ffn_dim=4 * int(c.encoder_embed_dim), heads=int(c.encoder_attention_heads),
                   token_dropout=bool(c.token_dropout), emb_layer_norm_before=False,
                   max_positions=0)
    else:
        a = data["args"]
        if a.arch != "roberta_large":
            raise ValueError("oracle restates only the roberta_large (ESM-1b/1v) v1 arch")
        prs1 = lambda s: "".join(s.split("encoder.")[1:] if "encoder" in s else s)
        prs2 = lambda s: "".join(s.split("sentence_encoder.")[1:] if "sentence_encoder" in s else s)
        sd = {prs1(prs2(k)): v for k, v in data["model"].items()}       # pretrained.py:91-96
        sd["embed_tokens.weight"][MASK].zero_()                      # pretrained.py:97 (in place:
        # reaches lm_head.weight too when the file stores the tied tensors with shared storage)
        cfg = dict(arch="esm1b", layers=int(a.encoder_layers), embed_dim=int(a.encoder_embed_dim),
                   ffn_dim=int(a.encoder_ffn_embed_dim), heads=int(a.encoder_attention_heads),
                   token_dropout=bool(getattr(a, "token_dropout", False)),
                   emb_layer_norm_before=any(k.startswith("emb_layer_norm_before") for k in sd),
                   max_positions=int(a.max_positions))
    # embed_tokens / lm_head.weight are one tied parameter (esm1.py:101-105): load_state_dict
    # (pretrained.py:216) leaves the lm_head.weight entry (copied last) in it.
    if "lm_head.weight" in sd:
        sd["embed_tokens.weight"] = sd["lm_head.weight"]
    W = {k: v.to(dtype) for k, v in sd.items() if not k.startswith("contact_head")}
    return cfg, W


# --- forward -----------------------------------------------------------------------------
def _layer_norm(x, w, b, eps=1e-5):
    """torch.nn.LayerNorm semantics (modules.py:80-81): biased variance, eps inside sqrt."""
    mu = x.mean(-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(-1, keepdim=True)
    return xc / torch.sqrt(var + eps) * w + b


def _gelu(x):
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))           # modules.py:17-24


def _rotary_tables(T, dh, dtype):
    inv_freq = 1.0 / (10000 ** (torch.arange(0, dh, 2).float() / dh))   # rotary_embedding.py:40
    t = torch.arange(T).type_as(inv_freq)
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)                     # tables built in f32


def _rotate_half(x):
    x1, x2 = x.chunk(2, dim=-1)
    return torch.cat((-x2, x1), dim=-1)


def forward_logits(cfg, W, tokens: np.ndarray, n_layers: int = None) -> torch.Tensor:
    """tokens int64 [B,T] -> logits [B,T,33].  Follows esm1.py:116-177 / esm2.py:76-130.
    Key padding (tokens == <pad>) is honoured exactly as the reference does: embeddings of pad
    rows zeroed, pad keys masked with -inf before the softmax.
    ``n_layers`` (default: all) truncates the encoder -- only bench.py's bounded CPU-baseline
    sample uses it, to time k of the 33 identical-cost layers."""
    tok = torch.as_tensor(np.asarray(tokens), dtype=torch.int64)
    B, T = tok.shape
    D, H = cfg["embed_dim"], cfg["heads"]
    dh = D // H
    dtype = W["embed_tokens.weight"].dtype
    pad = tok.eq(PAD)
    x = W["embed_tokens.weight"][tok]                                # embed_scale == 1
    if cfg["token_dropout"]:                                          # esm1.py:125-131
        is_mask = tok.eq(MASK)
        x = x.masked_fill(is_mask.unsqueeze(-1), 0.0)
        src_len = (~pad).sum(-1)
        ratio = is_mask.sum(-1).to(dtype) / src_len
        x = x * (1 - 0.15 * 0.8) / (1 - ratio)[:, None, None]
    if cfg["arch"] == "esm1b":
        if T > cfg["max_positions"]:                                  # modules.py:256-260
            raise ValueError(f"Sequence length {T} above maximum sequence length of "
                             f"{cfg['max_positions']}")
        m = (~pad).long()
        positions = torch.cumsum(m, dim=1) * m + PAD                  # modules.py:261-262
        x = x + W["embed_positions.weight"][positions]
        if cfg["emb_layer_norm_before"]:
            x = _layer_norm(x, W["emb_layer_norm_before.weight"], W["emb_layer_norm_before.bias"])
    x = x * (1 - pad.unsqueeze(-1).to(dtype))                         # esm1.py:138-139
    use_pad = bool(pad.any())
    if cfg["arch"] == "esm2":
        cos, sin = _rotary_tables(T, dh, dtype)
    scaling = dh ** -0.5
    for i in range(cfg["layers"] if n_layers is None else min(n_layers, cfg["layers"])):
        p = f"layers.{i}."
        res = x
        h = _layer_norm(x, W[p + "self_attn_layer_norm.weight"], W[p + "self_attn_layer_norm.bias"])
        q = h @ W[p + "self_attn.q_proj.weight"].T + W[p + "self_attn.q_proj.bias"]
        k = h @ W[p + "self_attn.k_proj.weight"].T + W[p + "self_attn.k_proj.bias"]
        v = h @ W[p + "self_attn.v_proj.weight"].T + W[p + "self_attn.v_proj.bias"]
        q = q * scaling                                               # multihead_attention.py:261
        q = q.view(B, T, H, dh).transpose(1, 2)
        k = k.view(B, T, H, dh).transpose(1, 2)
        v = v.view(B, T, H, dh).transpose(1, 2)
        if cfg["arch"] == "esm2":                                     # :354-355 (q already scaled)
            q = q * cos + _rotate_half(q) * sin
            k = k * cos + _rotate_half(k) * sin
        s = q @ k.transpose(-1, -2)                                   # :357
        if use_pad:                                                   # :370-376
            s = s.masked_fill(pad[:, None, None, :], float("-inf"))
        a = torch.softmax(s.float() if dtype != torch.float64 else s, dim=-1).to(dtype)  # :379
        ctx = (a @ v).transpose(1, 2).reshape(B, T, D)
        x = res + ctx @ W[p + "self_attn.out_proj.weight"].T + W[p + "self_attn.out_proj.bias"]
        res = x
        h = _layer_norm(x, W[p + "final_layer_norm.weight"], W[p + "final_layer_norm.bias"])
        h = _gelu(h @ W[p + "fc1.weight"].T + W[p + "fc1.bias"])
        x = res + h @ W[p + "fc2.weight"].T + W[p + "fc2.bias"]
    x = _layer_norm(x, W["emb_layer_norm_after.weight"], W["emb_layer_norm_after.bias"])
    h = _gelu(x @ W["lm_head.dense.weight"].T + W["lm_head.dense.bias"])        # modules.py:322-328
    h = _layer_norm(h, W["lm_head.layer_norm.weight"], W["lm_head.layer_norm.bias"])
    return h @ W["lm_head.weight"].T + W["lm_head.bias"]


# --- scoring strategies (compute_fitness.py) ----------------------------------------------
def get_optimal_window(mutation_position_relative, seq_len_wo_special, model_window):
    """proteingym/utils/scoring_utils.py:43-52."""
    half = model_window // 2
    if seq_len_wo_special <= model_window:
        return [0, seq_len_wo_special]
    elif mutation_position_relative < half:
        return [0, model_window]
    elif mutation_position_relative >= seq_len_wo_special - half:
        return [seq_len_wo_special - model_window, seq_len_wo_special]
    else:
        return [max(0, mutation_position_relative - half),
                min(seq_len_wo_special, mutation_position_relative + half)]


def masked_marginals_table(cfg, W, seq: str, positions=None, batch: int = 1) -> np.ndarray:
    """compute_fitness.py:486-504: for each token position i mask it, (optionally window),
    forward, log-softmax, keep row i-start.  Returns float [L+2,33] (rows not in
    ``positions`` are NaN).  ``batch`` > 1 stacks same-length windows in one forward -- the
    reference uses batch 1; results agree to float rounding."""
    tokens = tokenize(seq)
    n = len(tokens)
    out = np.full((n, VOCAB), np.nan, dtype=np.float64 if W["lm_head.bias"].dtype == torch.float64
                  else np.float32)
    todo = list(range(n)) if positions is None else list(positions)
    work = []
    for i in todo:
        t = tokens.copy()
        t[i] = MASK
        if n > 1024:
            start, end = get_optimal_window(i, len(seq) + 2, 1024)
            t = t[start:end]
        else:
            start = 0
        work.append((i, start, t))
    with torch.no_grad():
        for c in range(0, len(work), batch):
            chunk = work[c:c + batch]
            T = len(chunk[0][2])
            assert all(len(w[2]) == T for w in chunk)
            lg = forward_logits(cfg, W, np.stack([w[2] for w in chunk]))
            lp = torch.log_softmax(lg, dim=-1)
            for b, (i, start, _) in enumerate(chunk):
                out[i] = lp[b, i - start].numpy()
    return out


def wt_marginals_table(cfg, W, seq: str, scoring_window: str = "optimal") -> np.ndarray:
    """compute_fitness.py:433-475: one unmasked forward; for T>1024 with 'overlapping' the
    1024-wide windows stepped by 511 from both ends with sigmoid-ramped weights."""
    tokens = tokenize(seq)[None, :]
    T = tokens.shape[1]
    with torch.no_grad():
        if T > 1024 and scoring_window == "overlapping":
            dt = W["lm_head.bias"].dtype
            token_probs = torch.zeros((1, T, VOCAB), dtype=dt)
            token_weights = torch.zeros((1, T), dtype=dt)
            weights = torch.ones(1024, dtype=dt)
            for i in range(1, 257):
                weights[i] = 1 / (1 + math.exp(-(i - 128) / 16))
            for i in range(1022 - 256, 1023):
                weights[i] = 1 / (1 + math.exp((i - 1022 + 128) / 16))
            sl, el = 0, 1023
            sr, er = (T - 1) - 1024 + 1, T - 1

            def win(s, e):
                return torch.log_softmax(forward_logits(cfg, W, tokens[:, s:e + 1]), dim=-1)
            while True:
                token_probs[:, sl:el + 1] += win(sl, el) * weights.view(-1, 1)
                token_weights[:, sl:el + 1] += weights
                token_probs[:, sr:er + 1] += win(sr, er) * weights.view(-1, 1)
                token_weights[:, sr:er + 1] += weights
                if el > sr:
                    break
                sl += 511; el += 511; sr -= 511; er -= 511
            if el - sr + 1 < 511:
                sc = int(T / 2) - 512
                ec = sc + 1023
                token_probs[:, sc:ec + 1] += win(sc, ec) * weights.view(-1, 1)
                token_weights[:, sc:ec + 1] += weights
            token_probs = token_probs / token_weights.view(-1, 1)
        else:
            token_probs = torch.log_softmax(forward_logits(cfg, W, tokens), dim=-1)
    return token_probs[0].numpy()


def label_row(mutant: str, sequence: str, table: np.ndarray, offset_idx: int) -> float:
    """compute_fitness.py:240-250: sum over ':'-separated substitutions of
    (lp[1+idx, mt] - lp[1+idx, wt]); the f32 difference is taken first, then summed in double."""
    score = 0
    for mutation in mutant.split(":"):
        wt, idx, mt = mutation[0], int(mutation[1:-1]) - offset_idx, mutation[-1]
        assert sequence[idx] == wt, "The listed wildtype does not match the provided sequence"
        d = table[1 + idx, get_idx(mt)] - table[1 + idx, get_idx(wt)]   # same dtype as table
        score += float(d)
    return score


def compute_pppl(cfg, W, sequence: str) -> float:
    """compute_fitness.py:258-279, quirks preserved: loops i in range(1, len(sequence)-1);
    masks *token* i (= residue i-1) but looks up sequence[i]; no windowing."""
    tokens = tokenize(sequence)[None, :]
    lps = []
    with torch.no_grad():
        for i in range(1, len(sequence) - 1):
            t = tokens.copy()
            t[0, i] = MASK
            lp = torch.log_softmax(forward_logits(cfg, W, t), dim=-1)
            lps.append(lp[0, i, get_idx(sequence[i])].item())
    return sum(lps)


def score_dms(checkpoints, sequence: str, mutants, offset_idx: int = 1,
              strategy: str = "masked-marginals", model_type=("ESM1v",), dtype=torch.float32,
              scoring_window: str = "optimal"):
    """The per-assay scoring loop + ensemble (compute_fitness.py:348-537).  Returns
    {column_name: float64 array}, column names = checkpoint file stems (:350) and
    'Ensemble_ESM1v' (plain mean, :532-537) when 'ESM1v' in model_type."""
    cols = {}
    for path in checkpoints:
        name = str(path).split("/")[-1].split(".")[0]
        cfg, W = load_checkpoint(path, dtype)
        if strategy == "masked-marginals":
            table = masked_marginals_table(cfg, W, sequence)
            cols[name] = np.array([label_row(m, sequence, table, offset_idx) for m in mutants])
        elif strategy == "wt-marginals":
            table = wt_marginals_table(cfg, W, sequence, scoring_window)
            cols[name] = np.array([label_row(m, sequence, table, offset_idx) for m in mutants])
        elif strategy == "pseudo-ppl":
            def mutate(m):
                wt, idx, mt = m[0], int(m[1:-1]) - offset_idx, m[-1]        # :252-257
                assert sequence[idx] == wt
                return sequence[:idx] + mt + sequence[idx + 1:]
            cols[name] = np.array([compute_pppl(cfg, W, mutate(m)) for m in mutants])
        else:
            raise ValueError(strategy)
    if "ESM1v" in model_type:
        ens = np.zeros(len(mutants))
        for path in checkpoints:
            ens += cols[str(path).split("/")[-1].split(".")[0]]
        cols["Ensemble_ESM1v"] = ens / len(checkpoints)
    return cols


def from_arrays(arch: int, layers: int, embed_dim: int, heads: int, ffn_dim: int, max_positions: int,
                token_dropout: int, emb_layer_norm_before: int, arrays, dtype=torch.float32, **_):
    """Build (cfg, W) from in-memory arrays keyed by upgraded state-dict names (used with
    synthetic weights at the real 650M shape, where writing a .pt first would be wasteful).
    ``arch``: 1 = ESM-1b/1v, 2 = ESM2 (the ABI's numbering, include/pgmi.h)."""
    cfg = dict(arch="esm1b" if arch == 1 else "esm2", layers=layers, embed_dim=embed_dim, heads=heads,
               ffn_dim=ffn_dim, max_positions=max_positions, token_dropout=bool(token_dropout),
               emb_layer_norm_before=bool(emb_layer_norm_before))
    W = {k: torch.as_tensor(np.asarray(v)).to(dtype) for k, v in arrays.items()}
    return cfg, W
