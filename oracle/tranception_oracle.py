"""TEST INFRASTRUCTURE ONLY -- CPU restatement ("port") of ProteinGym's Tranception scoring path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
file.  The product (``proteingym_amd``) never does.

Restated (file:line relative to /root/reference/proteingym/baselines/tranception):
  * tokenizer (25 symbols, [CLS] seq [SEP], right-padding)   tranception/utils/tokenizers/Basic_tokenizer
  * ALiBi slopes / grouped bias                               tranception/model_pytorch.py:50-71,373-380
  * causal depth-wise convolution on q,k,v                    :73-88,240-251
  * attention (scale, causal -1e4 fill, +alibi, softmax)      :155-183
  * block (ln_1, attn, ln_2, squared-ReLU MLP)                :265-360 ; tranception/activations.py:79-84
  * model + LM head (tied to wte)                             :438-632,731-783
  * retrieval fusion                                          :783-846
  * MSA prior                                                 tranception/utils/msa_utils.py:63-138
  * scoring: slices, NLL, /len, delta to WT, mirror, average  tranception/utils/scoring_utils.py:77-203 ;
                                                              model_pytorch.py:878-928

PINNING: the reference has no tests for this path.  The oracle is pinned against the reference
itself, run in this container through the import shims of oracle/ref_harness.py
(load_reference_tranception): tests/golden/make_golden_tranception.py froze its outputs into
tests/golden/golden_tranception.npz, tests/test_oracle_pinning.py checks this file against them.
"""
from __future__ import annotations

import json
import math
import os

import numpy as np
import pandas as pd
import torch

VOCAB = {'[UNK]': 0, '[CLS]': 1, '[SEP]': 2, '[PAD]': 3, '[MASK]': 4, 'A': 5, 'C': 6, 'D': 7, 'E': 8, 'F': 9,
         'G': 10, 'H': 11, 'I': 12, 'K': 13, 'L': 14, 'M': 15, 'N': 16, 'P': 17, 'Q': 18, 'R': 19, 'S': 20,
         'T': 21, 'V': 22, 'W': 23, 'Y': 24}
CLS, SEP, PAD, UNK = 1, 2, 3, 0
AA_vocab = "ACDEFGHIKLMNPQRSTVWY"


def encode(seq: str):
    """[CLS] + one token per letter (unknown -> [UNK]) + [SEP] (the tokenizer's TemplateProcessing)."""
    return [CLS] + [VOCAB.get(c, UNK) for c in seq] + [SEP]


def encode_batch(seqs, n_ctx=1024):
    """model_pytorch.py:930-939: X/B/J/Z replaced by random eligible letters (np.random.choice, as
    the reference -- seed numpy for reproducibility), truncation to n_ctx, right-padding with [PAD].
    Returns (input_ids int64 [B,T], attention_mask int64 [B,T])."""
    def rep(s, ch, choices):
        idx = [i for i, c in enumerate(s) if c == ch]
        if not idx:
            return s
        r = np.random.choice(a=list(choices), size=len(idx), replace=True)
        s = list(s)
        for i, p in enumerate(idx):
            s[p] = r[i]
        return "".join(s)
    out = []
    for s in seqs:
        for ch, choices in (("X", AA_vocab), ("B", "DN"), ("J", "IL"), ("Z", "EQ")):
            s = rep(s, ch, choices)
        out.append(encode(s)[:n_ctx])
    T = max(len(e) for e in out)
    ids = np.full((len(out), T), PAD, dtype=np.int64)
    mask = np.zeros((len(out), T), dtype=np.int64)
    for i, e in enumerate(out):
        ids[i, :len(e)] = e
        mask[i, :len(e)] = 1
    return ids, mask


def get_slopes(n, mode="standard_alibi"):
    """model_pytorch.py:50-71."""
    def pow2(n):
        start = (2 ** (-2 ** -(math.log2(n) - 3)))
        return [start * start ** i for i in range(n)]
    if mode == "grouped_alibi":
        n = n // 4
    if math.log2(n).is_integer():
        result = pow2(n)
    else:
        c = 2 ** math.floor(math.log2(n))
        result = pow2(c) + get_slopes(2 * c)[0::2][:n - c]
    if mode == "grouped_alibi":
        result = result * 4
    return result


def load_checkpoint(dirpath, dtype=torch.float32):
    """config.json + pytorch_model.bin (score_tranception_proteingym.py:79,100)."""
    c = json.load(open(os.path.join(dirpath, "config.json")))
    sd = torch.load(os.path.join(dirpath, "pytorch_model.bin"), map_location="cpu")
    cfg = dict(n_layer=int(c["n_layer"]), n_embd=int(c["n_embd"]), n_head=int(c["n_head"]),
               n_ctx=int(c.get("n_ctx", c.get("n_positions", 1024))),
               n_inner=int(c["n_inner"]) if c.get("n_inner") else 4 * int(c["n_embd"]),
               eps=float(c.get("layer_norm_epsilon", 1e-5)), vocab=int(c.get("vocab_size", 25)))
    W = {k: v.to(dtype) for k, v in sd.items()}
    if "lm_head.weight" not in W:                       # tied to wte (GPT2 tie_word_embeddings)
        W["lm_head.weight"] = W["transformer.wte.weight"]
    return cfg, W


def _ln(x, w, b, eps):
    mu = x.mean(-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(-1, keepdim=True)
    return xc / torch.sqrt(var + eps) * w + b


def _dwconv(x, w, b):
    """SpatialDepthWiseConvolution (model_pytorch.py:73-88): x [B,h,T,dh]; per-channel causal conv,
    y[t] = sum_j w[c,j] x[t-(k-1)+j] + b[c]  (Conv1d padding k-1, last k-1 outputs dropped)."""
    k = w.shape[-1]
    B, h, T, dh = x.shape
    xp = torch.nn.functional.pad(x, (0, 0, k - 1, 0))                 # pad time on the left
    y = torch.zeros_like(x)
    for j in range(k):
        y = y + xp[:, :, j:j + T, :] * w[:, 0, j]
    return y + b


def forward_logits(cfg, W, input_ids, attention_mask=None):
    """input_ids int64 [B,T] -> logits [B,T,vocab]  (model_pytorch.py:438-632, 731-783)."""
    ids = torch.as_tensor(np.asarray(input_ids), dtype=torch.int64)
    B, T = ids.shape
    D, H = cfg["n_embd"], cfg["n_head"]
    dh = D // H
    dtype = W["transformer.wte.weight"].dtype
    x = W["transformer.wte.weight"][ids]                                 # no positional embedding
    slopes = torch.tensor(get_slopes(H, mode="grouped_alibi"), dtype=torch.float32)
    alibi = (slopes[:, None] * torch.arange(T, dtype=torch.float32)[None, :]).to(dtype)   # [H,T] bias on the key index
    causal = torch.tril(torch.ones(T, T, dtype=torch.bool))
    if attention_mask is not None:
        am = (1.0 - torch.as_tensor(np.asarray(attention_mask)).to(dtype)) * -10000.0      # :505-506
    hk = H // 4
    for i in range(cfg["n_layer"]):
        p = f"transformer.h.{i}."
        h = _ln(x, W[p + "ln_1.weight"], W[p + "ln_1.bias"], cfg["eps"])
        qkv = h @ W[p + "attn.c_attn.weight"] + W[p + "attn.c_attn.bias"]             # Conv1D: x W + b, W [in,out]
        q, k, v = [t.view(B, T, H, dh).permute(0, 2, 1, 3) for t in qkv.split(D, dim=2)]
        ql, kl, vl = [q[:, :hk]], [k[:, :hk]], [v[:, :hk]]
        for ki in range(3):                                                            # kernels 3,5,7 (:240-251)
            sl = slice((ki + 1) * hk, (ki + 2) * hk)
            ql.append(_dwconv(q[:, sl], W[p + f"attn.query_depthwiseconv.{ki}.conv.weight"], W[p + f"attn.query_depthwiseconv.{ki}.conv.bias"]))
            kl.append(_dwconv(k[:, sl], W[p + f"attn.key_depthwiseconv.{ki}.conv.weight"], W[p + f"attn.key_depthwiseconv.{ki}.conv.bias"]))
            vl.append(_dwconv(v[:, sl], W[p + f"attn.value_depthwiseconv.{ki}.conv.weight"], W[p + f"attn.value_depthwiseconv.{ki}.conv.bias"]))
        q, k, v = torch.cat(ql, 1), torch.cat(kl, 1), torch.cat(vl, 1)
        s = (q @ k.transpose(-1, -2)) / (float(dh) ** 0.5)                             # :158-159
        s = torch.where(causal, s, torch.tensor(-1e4, dtype=dtype))                    # :161-165
        s = s + alibi[None, :, None, :]                                                # :167-168
        if attention_mask is not None:
            s = s + am[:, None, None, :]
        a = torch.softmax(s, dim=-1)
        ctx = (a @ v).permute(0, 2, 1, 3).reshape(B, T, D)
        x = x + (ctx @ W[p + "attn.c_proj.weight"] + W[p + "attn.c_proj.bias"])
        h = _ln(x, W[p + "ln_2.weight"], W[p + "ln_2.bias"], cfg["eps"])
        h = torch.relu(h @ W[p + "mlp.c_fc.weight"] + W[p + "mlp.c_fc.bias"])
        h = h * h                                                                       # squared ReLU
        x = x + (h @ W[p + "mlp.c_proj.weight"] + W[p + "mlp.c_proj.bias"])
    x = _ln(x, W["transformer.ln_f.weight"], W["transformer.ln_f.bias"], cfg["eps"])
    return x @ W["lm_head.weight"].T


# ---- retrieval prior (tranception/utils/msa_utils.py:63-138, uniform weights branch + weights) ----
def process_msa_data(path):
    from collections import defaultdict
    msa = defaultdict(str)
    name = ""
    with open(path) as f:
        for line in f:
            line = line.rstrip()
            if line.startswith(">"):
                name = line
            else:
                msa[name] += line.upper()
    return msa


def get_msa_prior(MSA_data_file, MSA_start, MSA_end, len_target_seq, weights=None, filter_MSA=True):
    """``weights``: optional {sequence name -> weight}; names missing from it are dropped like the
    reference drops sequences without an EVE weight (msa_utils.py:104-115)."""
    msa = process_msa_data(MSA_data_file)
    V = len(VOCAB)

    def one_hot(s):
        o = np.zeros((len(s), V))
        for j, c in enumerate(s):
            if c in VOCAB:
                o[j, VOCAB[c]] = 1.0
        return o.flatten()
    if filter_MSA:
        names = list(msa.keys())
        ref = one_hot(msa[names[0]])
        for n in names:
            if np.dot(ref, one_hot(msa[n])) / np.dot(ref, ref) < 0.2:
                del msa[n]
    if weights is not None:
        for n in list(msa.keys()):
            if n not in weights:
                del msa[n]
        w = np.array([weights[n] for n in msa.keys()])
    else:
        w = np.ones(len(msa))
    one_hots = np.zeros((len(msa), MSA_end - MSA_start, V))
    for i, n in enumerate(msa.keys()):
        for j, c in enumerate(msa[n]):
            if c in VOCAB:
                one_hots[i, j, VOCAB[c]] = 1.0
    weighted = (one_hots + 1e-5) * w[:, None, None]
    norm = weighted.sum(-1).sum(0)
    avg = weighted.sum(0) / np.tile(norm.reshape(-1, 1), (1, V))
    prior = np.zeros((len_target_seq, V))
    prior[MSA_start:MSA_end, :] = avg
    return prior


# ---- scoring ---------------------------------------------------------------------------------
def eve_sequence_weights(MSA_location, theta=0.2, seq_gap_thr=0.5, col_gap_thr=1.0):
    """{sequence name -> weight} as ``MSA_processing`` derives them (msa_utils.py:230-368), written
    as plain loops: drop WT-gap columns; drop sequences with > 50 % gaps; lower-case columns with
    > 100 % gaps (none); keep focus columns (upper-case, non-gap in the WT); drop sequences holding a
    non-AA, non-gap letter there; weight_i = 1 / #{j: matches(i,j) / nongap(i) > 1 - theta}."""
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
    focus_name = order[0]
    seqs = {n: raw[n].replace(".", "-").upper() for n in order}
    keep = [i for i, c in enumerate(seqs[focus_name]) if c != "-"]
    seqs = {n: "".join(s[i] for i in keep) for n, s in seqs.items()}
    seqs = {n: s for n, s in seqs.items() if sum(c == "-" for c in s) / len(s) <= seq_gap_thr}
    ncol = len(keep)
    col_ok = [sum(s[j] == "-" for s in seqs.values()) / len(seqs) <= col_gap_thr for j in range(ncol)]
    seqs = {n: "".join(c.upper() if ok else c.lower() for c, ok in zip(s, col_ok)) for n, s in seqs.items()}
    focus = seqs[focus_name]
    focus_cols = [j for j, c in enumerate(focus) if c == c.upper() and c != "-"]
    trimmed = {n: [s[j].upper() for j in focus_cols] for n, s in seqs.items()}
    trimmed = {n: s for n, s in trimmed.items() if all((c in AA or c == "-") for c in s)}
    names = list(trimmed.keys())
    weights = {}
    for a in names:
        sa = trimmed[a]
        nongap = sum(c != "-" for c in sa)
        if nongap == 0:
            weights[a] = 0.0
            continue
        cnt = 0
        for b in names:
            m = sum(1 for x, y in zip(sa, trimmed[b]) if x == y and x != "-")
            if m / nongap > 1 - theta:
                cnt += 1
        weights[a] = 1.0 / cnt
    return weights


def get_mutated_sequence(focus_seq, mutant, start_idx=1):
    s = list(focus_seq)
    for m in mutant.split(":"):
        a, pos, b = m[0], int(m[1:-1]), m[-1]
        rp = pos - start_idx
        assert a == focus_seq[rp], "Invalid from_AA or mutant position: " + str(m)
        assert b in AA_vocab, "Mutant to_AA is invalid: " + str(m)
        s[rp] = b
    return "".join(s)


def get_optimal_window(p, n, w):
    h = w // 2
    if n <= w:
        return [0, n]
    elif p < h:
        return [0, w]
    elif p >= n - h:
        return [n - w, n]
    return [max(0, p - h), min(n, p + h)]


def get_sequence_slices(df, target_seq, model_context_len, start_idx=1, scoring_window="optimal", indel_mode=False):
    """scoring_utils.py:152-203: 'optimal' (:158-182; indels: window = whole sequence, :161-164,175-176) and
    'sliding' (:183-201: 1 + len // ctx consecutive windows, mutated then wild type per window)."""
    df = df.reset_index(drop=True).copy()
    n = len(df)
    L = len(target_seq)
    if scoring_window == "sliding":
        parts = []
        start = 0
        for _ in range(1 + int(L / model_context_len)):
            m = df.copy()
            m["sliced_mutated_sequence"] = m["mutated_sequence"].map(lambda x: x[start:start + model_context_len])
            m["window_start"] = [start] * n
            m["window_end"] = m["mutated_sequence"].map(lambda x: min(len(x), start + model_context_len))
            w = m.copy()
            w["mutated_sequence"] = [target_seq] * n
            w["sliced_mutated_sequence"] = w["mutated_sequence"].map(lambda x: x[start:start + model_context_len])
            w["window_end"] = w["mutated_sequence"].map(lambda x: min(len(x), start + model_context_len))
            parts += [m, w]
            start += model_context_len
        out = pd.concat(parts, axis=0)
        if "mutant" in out:
            del out["mutant"]
        return out.drop_duplicates()
    if indel_mode:
        win = df["mutated_sequence"].apply(lambda x: (0, len(x)))
    else:
        bary = df["mutant"].apply(lambda x: int(np.array([int(m[1:-1]) - start_idx for m in x.split(":")]).mean()))
        win = bary.apply(lambda x: get_optimal_window(x, L, model_context_len))
    df["sliced_mutated_sequence"] = [df["mutated_sequence"][i][win[i][0]:win[i][1]] for i in range(n)]
    df["window_start"] = win.map(lambda x: x[0])
    df["window_end"] = win.map(lambda x: x[1])
    if "mutant" in df:
        del df["mutant"]
    wt = df.copy()
    wt["mutated_sequence"] = [target_seq] * n
    if indel_mode:
        wt["window_end"] = wt["mutated_sequence"].map(len)
    wt["sliced_mutated_sequence"] = [target_seq[wt["window_start"][i]:wt["window_end"][i]] for i in range(n)]
    return pd.concat([df, wt], axis=0).drop_duplicates()


def clustal_aligner(MSA_filename, clustal_omega_location, work_dir, tag="oracle", num_sequences_kept=100000):
    """The file protocol around the aligner (msa_utils.py:148-173): once per alignment, its (at most 100 000: the first + a
    ``random.sample`` of the rest) sequences are written upper case with '.' as '-', 80 characters per line, the first one renamed
    >REFERENCE_SEQUENCE; per scored sequence a one-record file >SEQ_TO_SCORE; the executable is called as Biopython's
    ClustalOmegaCommandline calls it (``--profile1 A --profile2 B -o OUT --force``); the rows >SEQ_TO_SCORE and >REFERENCE_SEQUENCE of
    its output (read upper case) come back.  Returns ``aligner(sequence) -> (aligned sequence, aligned reference)``."""
    import random
    import subprocess
    os.makedirs(work_dir, exist_ok=True)
    base = os.path.basename(MSA_filename)
    sampled = os.path.join(work_dir, f"Sampled_{tag}_{base}")
    if not os.path.exists(sampled):
        msa = process_msa_data(MSA_filename)
        names = list(msa)
        if len(names) > num_sequences_kept:
            names = [names[0]] + random.sample(names[1:], k=num_sequences_kept - 1)
        with open(sampled, "w") as f:
            for i, n in enumerate(names):
                sq = msa[n].upper().replace(".", "-")
                f.write((">REFERENCE_SEQUENCE" if i == 0 else n) + "\n" + "\n".join(sq[k:k + 80] for k in range(0, len(sq), 80)) + "\n")
    to_align = os.path.join(work_dir, f"Seq_to_align_{tag}_{base}")
    expanded = os.path.join(work_dir, f"Expanded_{tag}_{base}")

    def aligner(sequence):
        with open(to_align, "w") as f:
            f.write("\n".join([">SEQ_TO_SCORE"] + [sequence[k:k + 80] for k in range(0, len(sequence), 80)]) + "\n")
        subprocess.run([clustal_omega_location, "--profile1", sampled, "--profile2", to_align, "-o", expanded, "--force"], check=True,
                       capture_output=True)
        rows = process_msa_data(expanded)
        return rows[">SEQ_TO_SCORE"], rows[">REFERENCE_SEQUENCE"]
    return aligner


def update_prior_indel(log_prior, MSA_start, MSA_end, aligned_seq, aligned_ref):
    """msa_utils.py:174-191: the walk over the two aligned rows (the sequence to score, the alignment's reference sequence) that edits the
    log-prior [rows, V] of the family alignment for ONE sequence.  Column by column: both gaps -> nothing; a gap in the sequence -> that
    prior row is dropped; a gap in the reference (an inserted residue) -> a ZERO row is inserted, at the index of the ALIGNMENT COLUMN (the
    reference indexes the prior with the column counter, so columns skipped as 'both gaps' shift every later insertion); then
    MSA_end = MSA_start + rows.  When the mask and the prior disagree in length the reference's bare ``except`` prints and returns the prior
    as edited so far with the OLD MSA_end; so does this."""
    prior = log_prior
    try:
        keep = []
        for col in range(len(aligned_seq)):
            a, b = aligned_seq[col], aligned_ref[col]
            if a == "-" and b == "-":
                continue
            if a == "-":
                keep.append(False)
            elif b == "-":
                prior = torch.cat((prior[:col], torch.zeros(1, prior.shape[1], dtype=prior.dtype), prior[col:]), dim=0)
                keep.append(True)
            else:
                keep.append(True)
        prior = prior[torch.tensor(keep, dtype=torch.bool)]
        MSA_end = MSA_start + len(prior)
    except Exception:
        pass
    return prior, MSA_start, MSA_end


def sequence_scores(cfg, W, sliced, window_start, window_end, reverse=False, retrieval=None, batch=20, mutated=None):
    """Sum over positions of log p(token t+1 | tokens <= t) for each sliced sequence
    (scoring_utils.py:97-128), with the MSA-prior fusion of model_pytorch.py:806-830 when
    ``retrieval`` = dict(log_prior [L,25], MSA_start, MSA_end, weight) is given.  With ``retrieval["aligner"]`` (indel scoring with
    retrieval, model_pytorch.py:794-799, 832-839; one sequence per forward): ``aligner(full mutated sequence)`` returns the two aligned rows
    (sequence, reference), the prior is edited for that sequence (``update_prior_indel``), and positions whose prior row sums to exactly 0
    -- the inserted residues -- and the last position keep the network's own log-probabilities; a prior slice that does not span the
    scored window is an IndexError, as in the reference."""
    out = []
    indel = retrieval is not None and retrieval.get("aligner") is not None
    if indel:
        batch = 1
    with torch.no_grad():
        for b0 in range(0, len(sliced), batch):
            seqs = list(sliced[b0:b0 + batch])
            ids, mask = encode_batch(seqs, cfg["n_ctx"])
            lg = forward_logits(cfg, W, ids, mask)
            lp = torch.log_softmax(lg[:, :-1, :], dim=-1)
            if retrieval is not None:
                fused = lp.clone()
                a = retrieval["weight"]
                log_prior, m_start, m_end = retrieval["log_prior"], retrieval["MSA_start"], retrieval["MSA_end"]
                if indel:
                    row_seq, row_ref = retrieval["aligner"](mutated[b0])
                    log_prior, m_start, m_end = update_prior_indel(torch.as_tensor(log_prior, dtype=lp.dtype), m_start, m_end, row_seq, row_ref)
                pr = None
                for s in range(len(seqs)):
                    st, en = int(window_start[b0 + s]), int(window_end[b0 + s])
                    lo, hi = max(st, m_start), min(en, m_end)
                    if hi <= lo:
                        continue
                    pr = torch.as_tensor(log_prior[lo:hi], dtype=lp.dtype)
                    if reverse:
                        pr = torch.flip(pr, dims=(0,))
                        a0 = max(0, en - m_end)
                    else:
                        a0 = max(0, m_start - st)
                    fused[s, a0:a0 + (hi - lo)] = (1 - a) * lp[s, a0:a0 + (hi - lo)] + a * pr
                if indel:
                    inserted = torch.tensor([bool(pr[i].sum() == 0) for i in range(len(pr))] + [True])
                    fused[:, inserted, :] = lp[:, inserted, :]          # IndexError when the slice (+1) and the positions differ in length
                lp = fused
            tgt = torch.as_tensor(ids[:, 1:])
            ll = torch.gather(lp, 2, tgt.unsqueeze(-1)).squeeze(-1)
            m = torch.as_tensor(mask[:, 1:]).to(ll.dtype)
            out.extend((ll * m).sum(1).tolist())
    return np.array(out)


def score_mutants(cfg, W, df, target_seq, scoring_mirror=True, retrieval=None, scoring_window="optimal", indel_mode=False):
    """model_pytorch.py:878-928.  Returns a DataFrame with mutated_sequence, avg_score_L_to_R, (avg_score_R_to_L),
    avg_score; the zero-score WT row is appended when the WT is among the inputs -- in indel mode with the sequence
    in column 'mutant' (:915-924), as the reference does.  'sliding' windows: per-window scores are summed per
    sequence before the length normalisation, one WT reference (scoring_utils.py:136-147)."""
    d = df.copy()
    if "mutated_sequence" not in d:
        d["mutated_sequence"] = d["mutant"].apply(lambda x: get_mutated_sequence(target_seq, x))
    if "mutant" not in d:
        d["mutant"] = d["mutated_sequence"]
    d = d[["mutated_sequence", "mutant"]]
    sl = get_sequence_slices(d, target_seq, cfg["n_ctx"] - 2, scoring_window=scoring_window,
                             indel_mode=indel_mode).reset_index(drop=True)

    def direction(name, rev):
        s = sl.copy()
        seqs = s["sliced_mutated_sequence"].apply(lambda x: x[::-1]) if rev else s["sliced_mutated_sequence"]
        s["score"] = sequence_scores(cfg, W, list(seqs), list(s["window_start"]), list(s["window_end"]), reverse=rev,
                                     retrieval=retrieval, mutated=list(s["mutated_sequence"]))
        if scoring_window == "sliding":
            s = s[["mutated_sequence", "score"]].groupby("mutated_sequence").sum().reset_index()
        s["score"] = s["score"] / s["mutated_sequence"].map(len)
        mut = s[s.mutated_sequence != target_seq]
        wt = s[s.mutated_sequence == target_seq]
        if scoring_window == "sliding":
            dl = mut.copy()
            dl[name] = dl["score"] - list(wt["score"])[0]
        else:
            dl = pd.merge(mut, wt, how="left", on=["window_start"], suffixes=("", "_wt"))
            dl[name] = dl["score"] - dl["score_wt"]
        return dl[["mutated_sequence", name]]
    out = direction("avg_score_L_to_R", False)
    if scoring_mirror:
        r = direction("avg_score_R_to_L", True)
        out = pd.merge(out, r, on="mutated_sequence", how="left", suffixes=("", "_R_to_L"))
        out["avg_score"] = (out["avg_score_L_to_R"] + out["avg_score_R_to_L"]) / 2.0
    else:
        out["avg_score"] = out["avg_score_L_to_R"]
    key = "mutant" if indel_mode else "mutated_sequence"
    if target_seq in df[key].values if key in df else False:
        cols = [key, "avg_score_L_to_R"] + (["avg_score_R_to_L"] if scoring_mirror else []) + ["avg_score"]
        out = pd.concat([out, pd.DataFrame([[target_seq] + [0] * (len(cols) - 1)], columns=cols)], ignore_index=True)
    return out


def from_arrays(layers, embed_dim, heads, ffn_dim, vocab, max_positions, arrays, ln_eps=1e-5, dtype=torch.float32, **_):
    """(cfg, W) from in-memory arrays keyed by HF state-dict names (synthetic real-shape weights)."""
    cfg = dict(n_layer=layers, n_embd=embed_dim, n_head=heads, n_ctx=max_positions, n_inner=ffn_dim, eps=ln_eps, vocab=vocab)
    W = {k: torch.as_tensor(np.asarray(v)).to(dtype) for k, v in arrays.items()}
    return cfg, W
