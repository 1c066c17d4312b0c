"""TEST INFRASTRUCTURE ONLY -- drives the *unmodified* reference Python on CPU.

This module imports ProteinGym's own ``proteingym/baselines/esm/compute_fitness.py``
(and the vendored fair-esm package next to it) from ``/root/reference`` through four
out-of-tree shims, so that the reference itself can (a) pin ``oracle/esm_oracle.py`` and
(b) generate the golden fixtures committed under ``tests/golden/``.

It only works where ``/root/reference`` exists (the build container).  Nothing on the GPU
box may import it: GPU tests, ``smoke()`` and ``bench.py`` use ``oracle/esm_oracle.py`` and
the committed fixtures instead.

Shims:
  1. sys.path: ``<ref>/proteingym/baselines/esm`` first (how ``python compute_fitness.py``
     resolves ``from esm import ...``, compute_fitness.py:16) + ``<ref>`` and
     ``<ref>/proteingym`` (``from proteingym.baselines.esm import esm``, esm/pretrained.py:86;
     ``from utils.scoring_utils import ...``, compute_fitness.py:18).
  2. stub ``Bio`` / ``numba`` modules (compute_fitness.py:9, utils/msa_utils.py:8-11) -- MSA only.
  3. ``torch.serialization.add_safe_globals([argparse.Namespace])`` (esm/pretrained.py:70
     pickles a Namespace; torch>=2.6 defaults to weights_only=True).
  4. ``torch.Tensor.cuda -> identity`` (compute_fitness.py:502 calls .cuda() even with --nogpu).
"""
from __future__ import annotations

import argparse
import importlib.util
import os
import sys
import types

REF_ROOT = os.environ.get("PGMI_REFERENCE_ROOT", "/root/reference")
REF_ESM = os.path.join(REF_ROOT, "proteingym", "baselines", "esm")

_cf = None


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REF_ESM, "compute_fitness.py"))


def load_reference():
    """Return the reference ``compute_fitness`` module (imported once)."""
    global _cf
    if _cf is not None:
        return _cf
    if not reference_available():
        raise RuntimeError(f"reference tree not found under {REF_ROOT}")
    import torch

    for n in ["Bio", "Bio.SeqIO", "Bio.SeqRecord", "Bio.Seq"]:
        if n not in sys.modules:
            sys.modules[n] = types.ModuleType(n)
    sys.modules["Bio"].SeqIO = sys.modules["Bio.SeqIO"]
    sys.modules["Bio.SeqRecord"].SeqRecord = object
    sys.modules["Bio.Seq"].Seq = object
    if "numba" not in sys.modules:
        nb = types.ModuleType("numba")
        _dec = lambda *a, **k: a[0] if (len(a) == 1 and callable(a[0]) and not k) else (lambda f: f)
        nb.jit = nb.njit = _dec
        nb.prange = range
        sys.modules["numba"] = nb
    torch.serialization.add_safe_globals([argparse.Namespace])
    torch.Tensor.cuda = lambda self, *a, **k: self
    if REF_ESM not in sys.path:
        sys.path.insert(0, REF_ESM)
    for p in (REF_ROOT, os.path.join(REF_ROOT, "proteingym")):
        if p not in sys.path:
            sys.path.append(p)
    spec = importlib.util.spec_from_file_location(
        "ref_compute_fitness", os.path.join(REF_ESM, "compute_fitness.py"))
    cf = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cf)
    _cf = cf
    return cf


def make_esm1v_checkpoint(path, layers, embed_dim, ffn_dim, heads, seed,
                          embed_std=None, max_positions=1024, emb_layer_norm_before=False,
                          token_dropout=True, randomize_affine=True):
    """Random-weight fair-esm *v1* checkpoint built with the reference constructor
    (esm/model/esm1.py:49-105), saved in the layout pretrained.py:85-99 expects."""
    import torch
    cf = load_reference()
    esm = sys.modules["esm"]
    torch.manual_seed(seed)
    ns = argparse.Namespace(arch="roberta_large", layers=layers, embed_dim=embed_dim,
                            ffn_embed_dim=ffn_dim, attention_heads=heads,
                            max_positions=max_positions, token_dropout=token_dropout,
                            emb_layer_norm_before=emb_layer_norm_before)
    alphabet = esm.Alphabet.from_architecture("roberta_large")
    model = esm.ProteinBertModel(ns, alphabet)
    sd = model.state_dict()
    _randomize(sd, embed_std, randomize_affine, seed)
    out = {}
    for k, v in sd.items():
        if k.startswith("contact_head"):
            continue
        pref = "encoder." if k.startswith("lm_head") else "encoder.sentence_encoder."
        out[pref + k] = v.clone()
    args = argparse.Namespace(arch="roberta_large", encoder_layers=layers,
                              encoder_embed_dim=embed_dim, encoder_ffn_embed_dim=ffn_dim,
                              encoder_attention_heads=heads, max_positions=max_positions,
                              token_dropout=token_dropout)
    torch.save({"args": args, "model": out}, path)
    return path


def make_esm2_checkpoint(path, layers, embed_dim, heads, seed, embed_std=None,
                         token_dropout=True, randomize_affine=True):
    """Random-weight fair-esm *v2* (ESM2) checkpoint (esm/model/esm2.py:40-74;
    pretrained.py:162-181).  The file stem must start with ``esm2`` (pretrained.py:187)."""
    import torch
    cf = load_reference()
    esm = sys.modules["esm"]
    assert os.path.basename(str(path)).startswith("esm2")
    torch.manual_seed(seed)
    alphabet = esm.data.Alphabet.from_architecture("ESM-1b")
    model = esm.model.esm2.ESM2(num_layers=layers, embed_dim=embed_dim, attention_heads=heads,
                                alphabet=alphabet, token_dropout=token_dropout)
    sd = model.state_dict()
    _randomize(sd, embed_std, randomize_affine, seed)
    out = {}
    for k, v in sd.items():
        if k.startswith("contact_head"):
            continue
        pref = "encoder." if k.startswith("lm_head") else "encoder.sentence_encoder."
        out[pref + k] = v.clone()
    cfg = argparse.Namespace(encoder_layers=layers, encoder_embed_dim=embed_dim,
                             encoder_attention_heads=heads, token_dropout=token_dropout)
    torch.save({"cfg": {"model": cfg}, "model": out}, path)
    return path


def _randomize(sd, embed_std, randomize_affine, seed):
    """Make the synthetic checkpoint a harder parity test than default init: non-trivial
    LayerNorm affines and biases (default init has weight=1 / bias=0, which would hide
    a dropped affine or bias), and an embedding scale giving realistic |LLR|."""
    import torch
    g = torch.Generator().manual_seed(1000 + seed)
    with torch.no_grad():
        for k, v in sd.items():
            if k == "embed_tokens.weight" and embed_std is not None:
                v.copy_(torch.randn(v.shape, generator=g) * embed_std)
            elif randomize_affine and ("layer_norm" in k or "emb_layer_norm" in k):
                if k.endswith("weight"):
                    v.copy_(1.0 + 0.1 * torch.randn(v.shape, generator=g))
                else:
                    v.copy_(0.05 * torch.randn(v.shape, generator=g))
            elif randomize_affine and k.endswith(".bias") and v.ndim == 1:
                v.copy_(0.02 * torch.randn(v.shape, generator=g))


def run_reference_cli(argv):
    """Run the reference ``main(args)`` with the reference parser (compute_fitness.py:100-238,282)."""
    cf = load_reference()
    args = cf.create_parser().parse_args(argv)
    cf.main(args)


def reference_model(path):
    """(model, alphabet) via the reference loader (esm/pretrained.py:24-28)."""
    cf = load_reference()
    model, alphabet = cf.pretrained.load_model_and_alphabet(str(path))
    model.eval()
    return model, alphabet
