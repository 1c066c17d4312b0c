"""Host-side mirror of the reference's ESM operator seam, backed by libpgmi.so (HIP, gfx950).

Reference interface being mirrored (all under /root/reference/proteingym/baselines/esm):
  * ``pretrained.load_model_and_alphabet(path) -> (model, alphabet)``  esm/pretrained.py:24-28,67-218
  * ``alphabet.get_idx / get_batch_converter``                         esm/data.py:92-174,262-297
  * ``model(tokens)["logits"]``                                         esm/model/esm1.py:116-177, esm2.py:76-130
  * the masked-marginals loop and ``label_row``                         compute_fitness.py:240-250,486-514

Same names, argument meaning and error behaviour; the arithmetic runs in hand-written HIP
kernels through the C ABI (include/pgmi.h).  No CPU fallback exists: without the built library
and a GPU every compute call raises ``PgmiError``.
"""
from __future__ import annotations

import argparse
import ctypes as C
import os
import re
from pathlib import Path
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import PgmiError, Config

# The 33 symbols of the ESM-1b / ESM2 / MSA Transformer vocabulary in index order -- the checkpoint's embedding rows ARE this order
# (esm/constants.py:8 between the four leading specials, one filler up to a multiple of 8, and <mask>; esm/data.py:92-174).
VOCABULARY = ("<cls>", "<pad>", "<eos>", "<unk>",
              *"LAGVSERTIDPKQNFYMHWCXBUZO.-",
              "<null_1>", "<mask>")
assert len(VOCABULARY) == 33
_INDEX = {tok: i for i, tok in enumerate(VOCABULARY)}


class Alphabet:
    """The ESM-1b / ESM2 alphabet as one constant table; attribute names are the ones the reference's callers read
    (``mask_idx``, ``padding_idx``, ``all_toks``, ``get_idx`` ...; arch "roberta_large" / "ESM-1b")."""

    all_toks = list(VOCABULARY)
    tok_to_idx = _INDEX
    cls_idx, padding_idx, eos_idx, unk_idx = _INDEX["<cls>"], _INDEX["<pad>"], _INDEX["<eos>"], _INDEX["<unk>"]
    mask_idx = _INDEX["<mask>"]
    prepend_bos = append_eos = True                      # <cls> before and <eos> after every sequence
    _single = frozenset(filter(lambda t: len(t) == 1, VOCABULARY))
    _multi = sorted(filter(lambda t: len(t) > 1, VOCABULARY), key=len, reverse=True)

    @classmethod
    def from_architecture(cls, name: str) -> "Alphabet":
        if name in ("ESM-1b", "roberta_large"):
            return cls()
        raise ValueError("Unknown architecture selected")

    def __len__(self):
        return len(VOCABULARY)

    def get_idx(self, tok):
        """Index of a token; anything outside the vocabulary is <unk> (mutant letters go through here)."""
        return self.tok_to_idx.get(tok, self.unk_idx)

    def get_tok(self, ind):
        return VOCABULARY[ind]

    def to_dict(self):
        return dict(self.tok_to_idx)

    def tokenize(self, text: str) -> List[str]:
        """Behaviour of the reference tokenizer (esm/data.py:178-251; every vocabulary entry is a no-split token): the text
        is cut at every vocabulary token -- the single residue letters and literal special tokens such as "<mask>" --
        whitespace between tokens is dropped, and whatever is left (a run of characters outside the vocabulary: lower
        case, 'J', '*', ...) stays ONE token, which ``encode`` then fails on exactly like the reference."""
        if all(ch in self._single for ch in text):
            return list(text)
        out, run, i, n = [], [], 0, len(text)

        def flush():
            if run:
                out.extend("".join(run).split())
                run.clear()
        while i < n:
            ch = text[i]
            if ch in self._single:
                flush()
                out.append(ch)
                i += 1
                continue
            if ch == "<":
                hit = next((t for t in self._multi if text.startswith(t, i)), None)
                if hit:
                    flush()
                    out.append(hit)
                    i += len(hit)
                    continue
            run.append(ch)
            i += 1
        flush()
        return out

    def encode(self, text: str) -> List[int]:
        """esm/data.py:253-254: ``[self.tok_to_idx[tok] for tok in self.tokenize(text)]`` -- a character outside the
        vocabulary is a KeyError (the reference does not map it to <unk>; only ``get_idx`` does, for mutant letters)."""
        return [self.tok_to_idx[tok] for tok in self.tokenize(text)]

    def get_batch_converter(self, truncation_seq_length: int = None):
        return BatchConverter(self, truncation_seq_length)


class BatchConverter:
    """esm/data.py:262-297: (label, seq) pairs -> (labels, strs, int64 tokens [B, maxlen+2])."""

    def __init__(self, alphabet, truncation_seq_length: int = None):
        self.alphabet = alphabet
        self.truncation_seq_length = truncation_seq_length

    def __call__(self, raw_batch: Sequence[Tuple[str, str]]):
        labels, strs = zip(*raw_batch)
        enc = [self.alphabet.encode(s) for s in strs]
        if self.truncation_seq_length:
            enc = [e[: self.truncation_seq_length] for e in enc]
        max_len = max(len(e) for e in enc)
        tokens = np.full((len(enc), max_len + 2), self.alphabet.padding_idx, dtype=np.int64)
        for i, e in enumerate(enc):
            tokens[i, 0] = self.alphabet.cls_idx
            tokens[i, 1:len(e) + 1] = e
            tokens[i, len(e) + 1] = self.alphabet.eos_idx
        return list(labels), list(strs), tokens


# ---- checkpoint -> flat weight blob ------------------------------------------------------------
def strip_fairseq_prefixes(key: str) -> str:
    """fair-esm v1 files keep fairseq's module path in the key: ``encoder.sentence_encoder.<name>`` for the body and
    ``encoder.<name>`` for the LM head.  Drop everything up to the innermost of those prefixes (the renaming that
    pretrained.py:91-96 applies before ``load_state_dict``)."""
    for marker in ("sentence_encoder.", "encoder."):
        if marker.rstrip(".") in key:
            key = "".join(key.split(marker)[1:])
    return key


def load_checkpoint_file(path):
    """``torch.load`` restricted to tensors and plain containers plus ``argparse.Namespace`` (what fair-esm checkpoints hold,
    esm/pretrained.py:70): a ``.pt`` is a pickle, and an unrestricted load runs whatever it names.  Files that need more are
    refused unless the user opts in with PGMI_UNSAFE_TORCH_LOAD=1."""
    import torch
    torch.serialization.add_safe_globals([argparse.Namespace])
    try:
        return torch.load(str(path), map_location="cpu", weights_only=True)
    except Exception as e:
        if os.environ.get("PGMI_UNSAFE_TORCH_LOAD") == "1":
            return torch.load(str(path), map_location="cpu", weights_only=False)
        raise RuntimeError(f"{path}: not loadable with weights_only=True ({type(e).__name__}: {e}); if the file is trusted, set "
                           "PGMI_UNSAFE_TORCH_LOAD=1 to unpickle it without restrictions") from e


def split_fused_in_proj(sd):
    """Legacy fairseq attention blocks store q, k and v as one ``in_proj_weight`` / ``in_proj_bias``; cut them into the
    three projections (the arithmetic of esm/multihead_attention.py:481-508).  The reference carries that hook but its
    loader never calls it (esm/pretrained.py goes straight to ``load_state_dict``), so it rejects such a file with missing
    keys; accepting it here is a superset.  Released ESM-1v/1b/ESM2 checkpoints do not use the fused form."""
    out = {}
    for k, v in sd.items():
        for fused, leaf in (("in_proj_weight", "weight"), ("in_proj_bias", "bias")):
            if k.endswith(fused):
                stem, dim = k[: -len(fused)], v.shape[0] // 3
                for j, name in enumerate(("q_proj", "k_proj", "v_proj")):
                    out[f"{stem}{name}.{leaf}"] = v[j * dim:(j + 1) * dim]
                break
        else:
            out[k] = v
    return out


def _upgrade_state_dict(path: str):
    """esm/pretrained.py:67-99 (v1) and :162-181 (v2).  torch is used only to unpickle the .pt."""
    data = load_checkpoint_file(path)
    model_name = Path(path).stem
    if model_name.startswith("esm2"):                                   # pretrained.py:187
        c = data["cfg"]["model"]
        pat = re.compile("^" + "|".join(["encoder.sentence_encoder.", "encoder."]))
        sd = {pat.sub("", k): v for k, v in data["model"].items()}
        cfg = dict(arch=_lib.ARCH_ESM2, layers=int(c.encoder_layers),
                   embed_dim=int(c.encoder_embed_dim), heads=int(c.encoder_attention_heads),
                   ffn_dim=4 * int(c.encoder_embed_dim), max_positions=0,
                   token_dropout=int(bool(c.token_dropout)), emb_layer_norm_before=0)
    else:
        a = data["args"]
        if a.arch != "roberta_large":
            raise ValueError("Unknown architecture selected")           # only ESM-1b/1v v1 + ESM2
        sd = {strip_fairseq_prefixes(k): v for k, v in data["model"].items()}
        sd["embed_tokens.weight"][32].zero_()      # in place, "For token drop" (pretrained.py:97)
        cfg = dict(arch=_lib.ARCH_ESM1B, layers=int(a.encoder_layers),
                   embed_dim=int(a.encoder_embed_dim), heads=int(a.encoder_attention_heads),
                   ffn_dim=int(a.encoder_ffn_embed_dim), max_positions=int(a.max_positions),
                   token_dropout=int(bool(getattr(a, "token_dropout", False))),
                   emb_layer_norm_before=int(any(k.startswith("emb_layer_norm_before") for k in sd)))
    sd = split_fused_in_proj({k: v for k, v in sd.items() if not k.startswith("contact_head")})
    # embed_tokens.weight and lm_head.weight are ONE tied parameter (esm1.py:101-105); load_state_dict
    # (pretrained.py:216) copies the entries in module order, so the value the model ends up with
    # is the lm_head.weight entry.  In real fair-esm files both entries share storage, so the
    # in-place zeroing above reaches both; in files where they do not, the reference keeps the
    # un-zeroed lm_head.weight row -- reproduce either case by using that entry for both roles.
    if "lm_head.weight" in sd:
        sd["embed_tokens.weight"] = sd["lm_head.weight"]
    return cfg, sd


def expected_keys(cfg) -> List[str]:
    """State-dict keys in ABI blob order (include/pgmi.h, pgmi_weight_count)."""
    keys = ["embed_tokens.weight"]
    if cfg["arch"] == _lib.ARCH_ESM1B:
        keys.append("embed_positions.weight")
    if cfg["emb_layer_norm_before"]:
        keys += ["emb_layer_norm_before.weight", "emb_layer_norm_before.bias"]
    for i in range(cfg["layers"]):
        p = f"layers.{i}."
        keys += [p + "self_attn_layer_norm.weight", p + "self_attn_layer_norm.bias"]
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            keys += [p + f"self_attn.{n}.weight", p + f"self_attn.{n}.bias"]
        keys += [p + "final_layer_norm.weight", p + "final_layer_norm.bias",
                 p + "fc1.weight", p + "fc1.bias", p + "fc2.weight", p + "fc2.bias"]
    keys += ["emb_layer_norm_after.weight", "emb_layer_norm_after.bias",
             "lm_head.dense.weight", "lm_head.dense.bias",
             "lm_head.layer_norm.weight", "lm_head.layer_norm.bias", "lm_head.bias"]
    return keys


def pack_state_dict(cfg, sd) -> np.ndarray:
    """Strict key check as esm/pretrained.py:192-210, then flatten to one fp32 blob."""
    keys = expected_keys(cfg)
    allowed_extra = {"lm_head.weight"}                     # tied to embed_tokens (esm1.py:101-105)
    if cfg["arch"] == _lib.ARCH_ESM2:
        allowed_extra |= {f"layers.{i}.self_attn.rot_emb.inv_freq" for i in range(cfg["layers"])}
    missing = [k for k in keys if k not in sd]
    unexpected = [k for k in sd if k not in keys and k not in allowed_extra]
    msgs = []
    if missing:
        msgs.append(f"Missing key(s) in state_dict: {set(missing)}.")
    if unexpected:
        msgs.append(f"Unexpected key(s) in state_dict: {set(unexpected)}.")
    if msgs:
        raise RuntimeError("Error(s) in loading state_dict:\n\t" + "\n\t".join(msgs))
    parts = []
    for k in keys:
        t = sd[k]
        a = t.detach().to("cpu").float().numpy() if hasattr(t, "detach") else np.asarray(t, np.float32)
        parts.append(np.ascontiguousarray(a, dtype=np.float32).ravel())
    return np.concatenate(parts)


class EsmModel:
    """Device-resident ESM-1b/1v/ESM2 masked LM.  ``model(tokens)["logits"]`` mirrors the
    reference call (compute_fitness.py:502) but returns log-probabilities, i.e. already
    ``log_softmax``-ed logits -- log_softmax is idempotent, so reference-style callers that
    apply ``torch.log_softmax(..., dim=-1)`` on top get the same numbers."""

    def __init__(self, cfg: dict, weights: np.ndarray, device: int = 0, precision: str = "f16x3",
                 max_rows: int = 0):
        lib = _lib.load()
        self.cfg = dict(cfg)
        if precision == "bf16" and (cfg["embed_dim"] % 64 or cfg["ffn_dim"] % 64):
            # the bf16 GEMM's K tile is 64 deep (ESM2-35M has embed_dim 480): that (un-gated, throughput) mode hands such a
            # model to fp32 -- said out loud, never a silent switch.  f16x3 takes any multiple of 32.
            import sys
            print(f"[proteingym_amd] embed_dim={cfg['embed_dim']} / ffn_dim={cfg['ffn_dim']} is not a multiple of 64: "
                  f"using precision fp32 instead of {precision}", file=sys.stderr)
            precision = "fp32"
        self.precision = precision
        c = Config(abi_version=_lib.ABI_VERSION, arch=cfg["arch"], layers=cfg["layers"],
                   embed_dim=cfg["embed_dim"], heads=cfg["heads"], ffn_dim=cfg["ffn_dim"], vocab=33,
                   max_positions=cfg["max_positions"], token_dropout=cfg["token_dropout"],
                   emb_layer_norm_before=cfg["emb_layer_norm_before"],
                   precision=_lib.PRECISIONS[precision], max_rows=max_rows, ln_eps=0.0)
        self._c = c
        n = lib.pgmi_weight_count(C.byref(c))
        w = _lib.as_f32(weights)
        if w.size != n:
            raise PgmiError(f"weight blob has {w.size} elements, config needs {n}")
        h = C.c_void_p()
        _lib.check(lib.pgmi_model_create(C.byref(c), _lib.ptr(w, _lib._f32p), w.size, device, C.byref(h)))
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            _lib.load().pgmi_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def eval(self):
        return self

    def cuda(self, *a, **k):
        return self

    # -- forward --------------------------------------------------------------------------------
    def token_logprobs(self, tokens) -> np.ndarray:
        t = _lib.as_i32(np.asarray(tokens))
        assert t.ndim == 2
        B, T = t.shape
        out = np.empty((B, T, 33), dtype=np.float32)
        _lib.check(_lib.load().pgmi_token_logprobs(self._h, _lib.ptr(t, _lib._i32p), B, T, _lib.ptr(out, _lib._f32p)))
        return out

    def __call__(self, tokens, **_):
        return {"logits": self.token_logprobs(tokens)}

    def masked_logprobs(self, tokens, mask_pos) -> np.ndarray:
        t = _lib.as_i32(np.asarray(tokens))
        mp = _lib.as_i32(np.asarray(mask_pos))
        B, T = t.shape
        out = np.empty((B, 33), dtype=np.float32)
        _lib.check(_lib.load().pgmi_masked_logprobs(self._h, _lib.ptr(t, _lib._i32p), _lib.ptr(mp, _lib._i32p),
                                                    B, T, _lib.ptr(out, _lib._f32p)))
        return out

    # -- profiling ------------------------------------------------------------------------------
    def profile_enable(self, on=True):
        _lib.check(_lib.load().pgmi_profile_enable(self._h, int(on)))

    def profile_reset(self):
        _lib.check(_lib.load().pgmi_profile_reset(self._h))

    def profile(self) -> dict:
        lib = _lib.load()
        out = {}
        for k, name in enumerate(_lib.K_NAMES):
            ms, n, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
            _lib.check(lib.pgmi_profile_get(self._h, k, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)))
            out[name] = dict(ms=ms.value, launches=n.value, flops=fl.value, bytes=by.value)
        return out


class Assay:
    """One DMS assay resident on the device (pgmi_assay_*): wild-type tokens, the positions to
    mask, and the flattened substitutions of every mutant."""

    def __init__(self, model: EsmModel, sequence: str, mutants: Sequence[str], offset_idx: int = 1,
                 alphabet: Optional[Alphabet] = None, window: int = 1024, all_positions: bool = False,
                 positions: Optional[Sequence[int]] = None):
        lib = _lib.load()
        self.model = model
        alphabet = alphabet or Alphabet()
        _, _, toks = alphabet.get_batch_converter()([("protein1", sequence)])
        self.wt_tokens = _lib.as_i32(toks[0])
        self.n_tok = int(self.wt_tokens.size)
        sub_pos, sub_wt, sub_mt, mut_off = parse_mutants(mutants, sequence, offset_idx)
        self.n_mut = len(mutants)
        if positions is not None:                               # an explicit shard of the token positions (dist.py)
            positions = np.unique(np.asarray(positions, dtype=np.int32))
            if positions.size and (positions[0] < 0 or positions[-1] >= self.n_tok):
                raise ValueError("positions outside the token range")
        elif all_positions:
            positions = np.arange(self.n_tok, dtype=np.int32)   # what the reference runs (:489)
        else:
            positions = np.unique(sub_pos).astype(np.int32)     # rows some mutant reads
        self.positions = positions
        h = C.c_void_p()
        _lib.check(lib.pgmi_assay_create(
            model._h, _lib.ptr(self.wt_tokens, _lib._i32p), self.n_tok,
            _lib.ptr(positions, _lib._i32p), int(positions.size), int(window),
            _lib.ptr(sub_pos, _lib._i32p), _lib.ptr(sub_wt, _lib._i32p), _lib.ptr(sub_mt, _lib._i32p),
            _lib.ptr(mut_off, _lib._i64p), self.n_mut, C.byref(h)))
        self._h = h
        self.T = min(self.n_tok, window)

    def run(self, want_table: bool = False, scores_dev_ptr: int = 0):
        lib = _lib.load()
        scores = np.empty(self.n_mut, dtype=np.float64)
        table = np.empty((self.n_tok, 33), dtype=np.float32) if want_table else None
        if not self.model._h:
            raise PgmiError("bad model/assay handle (assay belongs to a destroyed model)")
        _lib.check(lib.pgmi_assay_run(self.model._h, self._h, _lib.ptr(scores, _lib._f64p),
                                      _lib.ptr(table, _lib._f32p) if want_table else None,
                                      C.c_void_p(scores_dev_ptr) if scores_dev_ptr else None))
        return (scores, table) if want_table else scores

    def run_device_only(self, scores_dev_ptr: int = 0):
        """Run with no host copies (bench: results stay in HBM / go to a device buffer)."""
        _lib.check(_lib.load().pgmi_assay_run(self.model._h, self._h, None, None,
                                              C.c_void_p(scores_dev_ptr) if scores_dev_ptr else None))

    def close(self):
        if getattr(self, "_h", None):
            _lib.load().pgmi_assay_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pack_sequences(sequences: Sequence[str], alphabet: Optional[Alphabet] = None):
    """BatchConverter (esm/data.py:262-297) for a whole column without padding: every sequence as <cls> + one token
    per residue letter + <eos>, concatenated, one byte per token (anything but residue letters takes the slow path through
    Alphabet.encode: same tokens and same KeyError as the reference).
    Returns (tokens uint8 [sum(len)+2N], seq_off int64 [N+1])."""
    alphabet = alphabet or Alphabet()
    n = len(sequences)
    lut = np.full(256, 255, dtype=np.uint8)                            # vocabulary index per residue letter, 255 = not a letter token
    for tok, i in alphabet.tok_to_idx.items():
        if len(tok) == 1:
            lut[ord(tok)] = i
    flat = lut[np.frombuffer("".join(sequences).encode("latin-1", "replace"), dtype=np.uint8)]
    if flat.size and flat.max() == 255:
        # some member holds something else than residue letters (whitespace, a literal "<mask>", or a character the
        # reference tokenizer raises KeyError on): those go through Alphabet.encode, one by one
        enc = [np.asarray(alphabet.encode(s), dtype=np.uint8) for s in sequences]
        lens = np.fromiter((e.size for e in enc), dtype=np.int64, count=n)
        flat = np.concatenate(enc) if n else np.zeros(0, np.uint8)
    else:
        lens = np.fromiter((len(s) for s in sequences), dtype=np.int64, count=n)
    seq_off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens + 2, out=seq_off[1:])
    toks = np.empty(int(seq_off[-1]), dtype=np.uint8)
    body = np.ones(toks.size, dtype=bool)
    body[seq_off[:-1]] = False
    body[seq_off[1:] - 1] = False
    toks[seq_off[:-1]] = alphabet.cls_idx
    toks[seq_off[1:] - 1] = alphabet.eos_idx
    toks[body] = flat
    return toks, seq_off


class SequenceLibrary:
    """A library of variable-length sequences resident on the device for pseudo-perplexity scoring
    (pgmi_pppl_*; compute_fitness.py:258-279,515-529).  Tokens are uploaded once, one byte each; every
    (sequence, masked position) row is built on the device, mixed lengths share a batch."""

    MAX_ROWS_PER_CALL = 1 << 24          # bounds the per-call device term buffer (64 MB) and the time of one C call

    def __init__(self, model: "EsmModel", sequences: Sequence[str], alphabet: Optional[Alphabet] = None):
        lib = _lib.load()
        alphabet = alphabet or Alphabet()
        self.model = model
        self.n = len(sequences)
        toks, self.seq_off = pack_sequences(sequences, alphabet)
        lens = np.diff(self.seq_off) - 2
        self.rows = np.maximum(lens - 2, 0)                            # masked forwards per sequence: range(1, L-1)
        h = C.c_void_p()
        _lib.check(lib.pgmi_pppl_create(model._h, toks.ctypes.data_as(C.POINTER(C.c_uint8)), _lib.ptr(self.seq_off, _lib._i64p),
                                        self.n, C.byref(h)))
        self._h = h

    def score(self, first: int = 0, count: Optional[int] = None, want_terms: bool = False, scores_dev_ptr: int = 0):
        """Scores of sequences [first, first+count) (float64), optionally with the per-position terms
        (list of float32 arrays).  Long ranges are cut into calls of at most MAX_ROWS_PER_CALL rows."""
        lib = _lib.load()
        count = self.n - first if count is None else count
        out = np.empty(count, dtype=np.float64)
        terms = [] if want_terms else None
        a = first
        csum = np.concatenate([[0], np.cumsum(self.rows[first:first + count])])
        while a < first + count:
            b = int(np.searchsorted(csum, csum[a - first] + self.MAX_ROWS_PER_CALL, side="right")) - 1 + first
            b = min(max(b, a + 1), first + count)
            nrow = int(csum[b - first] - csum[a - first])
            tbuf = np.empty(max(nrow, 1), dtype=np.float32) if want_terms else None
            dev = C.c_void_p(scores_dev_ptr + 8 * (a - first)) if scores_dev_ptr else None
            _lib.check(lib.pgmi_pppl_run(self.model._h, self._h, a, b - a, _lib.ptr(out[a - first:b - first], _lib._f64p),
                                         _lib.ptr(tbuf, _lib._f32p) if want_terms else None, dev))
            if want_terms:
                o = 0
                for n in range(a, b):
                    terms.append(tbuf[o:o + int(self.rows[n])].copy())
                    o += int(self.rows[n])
            a = b
        return (out, terms) if want_terms else out

    def stats(self) -> dict:
        v = [C.c_int64() for _ in range(4)]
        _lib.check(_lib.load().pgmi_pppl_stats(self._h, *[C.byref(x) for x in v]))
        r, c, t, p = (x.value for x in v)
        return dict(rows=r, batches=c, tokens=t, padded_tokens=p, packing_efficiency=(t / p if p else 1.0))

    def close(self):
        if getattr(self, "_h", None):
            _lib.load().pgmi_pppl_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def parse_mutants(mutants: Sequence[str], sequence: str, offset_idx: int):
    """label_row's string handling (compute_fitness.py:240-250) for a whole column at once, in
    C++ (pgmi_parse_mutants).  Raises AssertionError on a wild-type mismatch like the reference."""
    lib = _lib.load()
    enc = [str(m).encode() for m in mutants]
    off = np.zeros(len(enc) + 1, dtype=np.int64)
    np.cumsum([len(e) for e in enc], out=off[1:])
    text = b"".join(enc)
    seq = sequence.encode()
    n_sub = C.c_int64()
    rc = lib.pgmi_parse_mutants(text, _lib.ptr(off, _lib._i64p), len(enc), seq, len(seq), int(offset_idx),
                                None, None, None, None, C.byref(n_sub))
    _raise_parse(rc)
    n = n_sub.value
    sub_pos = np.empty(n, np.int32); sub_wt = np.empty(n, np.int32); sub_mt = np.empty(n, np.int32)
    mut_off = np.empty(len(enc) + 1, np.int64)
    rc = lib.pgmi_parse_mutants(text, _lib.ptr(off, _lib._i64p), len(enc), seq, len(seq), int(offset_idx),
                                _lib.ptr(sub_pos, _lib._i32p), _lib.ptr(sub_wt, _lib._i32p),
                                _lib.ptr(sub_mt, _lib._i32p), _lib.ptr(mut_off, _lib._i64p), C.byref(n_sub))
    _raise_parse(rc)
    return sub_pos, sub_wt, sub_mt, mut_off


def positions_read(mutants: Sequence[str], sequence: str, offset_idx: int = 1) -> np.ndarray:
    """Token positions (1 + residue index: <cls> is token 0) whose table rows some mutant reads."""
    sub_pos, _, _, _ = parse_mutants(mutants, sequence, offset_idx)
    return np.unique(sub_pos).astype(np.int32)


def score_from_table(table: np.ndarray, mutants: Sequence[str], sequence: str, offset_idx: int = 1) -> np.ndarray:
    """``label_row`` (compute_fitness.py:240-250) for a whole column on the host from a log-prob table: per
    substitution an f32 difference table[pos, mt] - table[pos, wt], accumulated in double in the order of the
    mutation string -- the arithmetic of the reference's ``.item()`` sum and of ``score_mutants_kernel``, so the
    result is bit-identical to ``Assay.run()``.  Used when the table was assembled from position shards."""
    return score_parsed(table, *parse_mutants(mutants, sequence, offset_idx))


def score_parsed(table: np.ndarray, sub_pos, sub_wt, sub_mt, mut_off) -> np.ndarray:
    """score_from_table on the arrays parse_mutants returns (callers that need the parse for something else as well):
    ``pgmi_score_mutants``, host-side C with the arithmetic of ``score_mutants_kernel``."""
    t = np.ascontiguousarray(table, dtype=np.float32)
    if t.ndim != 2:
        raise ValueError("table must be [positions, vocabulary]")
    sub_pos, sub_wt, sub_mt = _lib.as_i32(sub_pos), _lib.as_i32(sub_wt), _lib.as_i32(sub_mt)
    mut_off = np.ascontiguousarray(mut_off, dtype=np.int64)
    out = np.empty(len(mut_off) - 1, dtype=np.float64)
    _lib.check(_lib.load().pgmi_score_mutants(_lib.ptr(t, _lib._f32p), t.shape[0], t.shape[1], _lib.ptr(sub_pos, _lib._i32p),
                                              _lib.ptr(sub_wt, _lib._i32p), _lib.ptr(sub_mt, _lib._i32p), _lib.ptr(mut_off, _lib._i64p),
                                              len(mut_off) - 1, _lib.ptr(out, _lib._f64p)))
    return out


def _raise_parse(rc):
    if rc == 0:
        return
    msg = _lib.load().pgmi_last_error().decode(errors="replace")
    if "does not match" in msg:
        raise AssertionError(msg)                          # compute_fitness.py:244
    raise ValueError(msg)


def get_optimal_window(mutation_position_relative, seq_len_wo_special, model_window):
    """proteingym/utils/scoring_utils.py:43-52 (C implementation: pgmi_optimal_window)."""
    s, e = C.c_int32(), C.c_int32()
    _lib.load().pgmi_optimal_window(int(mutation_position_relative), int(seq_len_wo_special),
                                    int(model_window), C.byref(s), C.byref(e))
    return [s.value, e.value]


def load_model_and_alphabet(model_location: str, device: int = 0, precision: str = "f16x3",
                            max_rows: int = 0):
    """Mirror of esm/pretrained.py:24-28 for local ``.pt`` files."""
    if not str(model_location).endswith(".pt"):
        raise ValueError("only local .pt checkpoints are supported (no network): " + str(model_location))
    cfg, sd = _upgrade_state_dict(model_location)
    blob = pack_state_dict(cfg, sd)
    return EsmModel(cfg, blob, device=device, precision=precision, max_rows=max_rows), Alphabet()
