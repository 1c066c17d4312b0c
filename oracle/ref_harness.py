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
        nb.prange = lambda n: range(int(n))
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


# ---- Tranception ----------------------------------------------------------------------------
# The reference module targets transformers==4.32.1 (environments/proteingym_env.txt:119); this
# image has 5.x.  Out-of-tree shims, none of which touches the arithmetic:
#   * transformers.modeling_utils.{Conv1D, SequenceSummary, find_pruneable_heads_and_indices,
#     prune_conv1d_layer} and transformers.file_utils docstring decorators were moved/removed
#     (tranception/model_pytorch.py:13-27) -> re-exported / no-op decorators;
#   * transformers.utils.model_parallel_utils is gone (:33) -> stub (model-parallel code is dead
#     on the scoring path);
#   * PreTrainedModel.init_weights() is called before post_init() by the old-style constructors
#     (:386,:642) -> provide the attribute it expects; get_head_mask() was removed (:524) ->
#     returns [None]*n_layer exactly like the old implementation does for head_mask=None;
#   * Bio.Align.Applications (tranception/utils/msa_utils.py:7) -> stub (indel re-alignment only).
REF_TRANCEPTION = os.path.join(REF_ROOT, "proteingym", "baselines", "tranception")
_tr = None


def torch_cuda_available() -> bool:
    import torch
    return torch.cuda.is_available()


class _ClustalOmegaCommandline:
    """Biopython is not installed here.  The reference only builds ``ClustalOmegaCommandline(cmd=<executable>, profile1=..., profile2=...,
    outfile=..., force=True)`` and calls it (msa_utils.py:167-172); Biopython turns that into
    ``<executable> --profile1 A --profile2 B -o OUT --force`` and returns (stdout, stderr).  Same here."""

    def __init__(self, cmd="clustalo", **options):
        self.cmd, self.options = cmd, options

    def __call__(self):
        import subprocess
        o = self.options
        argv = [self.cmd, "--profile1", o["profile1"], "--profile2", o["profile2"], "-o", o["outfile"]] + (["--force"] if o.get("force") else [])
        done = subprocess.run(argv, capture_output=True, text=True, check=True)
        return done.stdout, done.stderr


def load_reference_tranception():
    """Returns (tranception package, tokenizer) from the unmodified reference."""
    global _tr
    if _tr is not None:
        return _tr
    import transformers
    import transformers.modeling_utils as mu
    import transformers.file_utils as fu
    from transformers.pytorch_utils import Conv1D
    if not hasattr(mu, "Conv1D"):
        mu.Conv1D = Conv1D
    for name in ("SequenceSummary", "find_pruneable_heads_and_indices", "prune_conv1d_layer"):
        if not hasattr(mu, name):
            setattr(mu, name, getattr(transformers.pytorch_utils, name, None) or (lambda *a, **k: None))
    for name in ("ModelOutput", "add_code_sample_docstrings", "add_start_docstrings",
                 "add_start_docstrings_to_model_forward", "replace_return_docstrings"):
        if not hasattr(fu, name):
            src = getattr(transformers.utils, name, None)
            setattr(fu, name, src if src is not None else (lambda *a, **k: (lambda f: f)))
    if "transformers.utils.model_parallel_utils" not in sys.modules:
        mp = types.ModuleType("transformers.utils.model_parallel_utils")
        mp.assert_device_map = lambda *a, **k: None
        mp.get_device_map = lambda *a, **k: None
        sys.modules["transformers.utils.model_parallel_utils"] = mp
    for n in ["Bio", "Bio.Align", "Bio.Align.Applications"]:
        sys.modules.setdefault(n, types.ModuleType(n))
    sys.modules["Bio.Align.Applications"].ClustalOmegaCommandline = _ClustalOmegaCommandline
    if not torch_cuda_available():
        # msa_utils.update_retrieved_MSA_log_prior_indel builds its zero row with .cuda() (msa_utils.py:183): on a machine without a GPU
        # that call is the identity here, so the reference's own walk runs instead of falling into its bare except
        import torch
        torch.Tensor.cuda = lambda self, *a, **k: self
    if not getattr(mu.PreTrainedModel, "_pgmi_patched", False):
        _orig_iw = mu.PreTrainedModel.init_weights

        def _iw(self):
            if not hasattr(self, "all_tied_weights_keys"):
                self.all_tied_weights_keys = {}
            return _orig_iw(self)
        mu.PreTrainedModel.init_weights = _iw
        if not hasattr(mu.PreTrainedModel, "get_head_mask"):
            mu.PreTrainedModel.get_head_mask = lambda self, head_mask, n, *a, **k: [None] * n
        mu.PreTrainedModel._pgmi_patched = True
    if REF_TRANCEPTION not in sys.path:
        sys.path.insert(0, REF_TRANCEPTION)
    import tranception
    from tranception import config as tconfig, model_pytorch  # noqa: F401
    from transformers import PreTrainedTokenizerFast
    tok = PreTrainedTokenizerFast(
        tokenizer_file=os.path.join(REF_TRANCEPTION, "tranception", "utils", "tokenizers", "Basic_tokenizer"),
        unk_token="[UNK]", sep_token="[SEP]", pad_token="[PAD]", cls_token="[CLS]", mask_token="[MASK]")
    _tr = (tranception, tok)
    return _tr


def make_tranception_checkpoint(dirpath, n_layer, n_embd, n_head, seed, n_ctx=1024, embed_std=0.3):
    """Random-weight Tranception checkpoint directory (config.json + pytorch_model.bin, the two
    files score_tranception_proteingym.py:79,100 read) built with the reference constructor."""
    import json
    import torch
    tranception, tok = load_reference_tranception()
    cfg = tranception.config.TranceptionConfig(
        vocab_size=25, n_embd=n_embd, n_head=n_head, n_layer=n_layer, n_positions=n_ctx, n_ctx=n_ctx,
        attention_mode="tranception", position_embedding="grouped_alibi", activation_function="squared_relu",
        attn_pdrop=0.0, resid_pdrop=0.0, embd_pdrop=0.0, bos_token_id=1, eos_token_id=2)
    torch.manual_seed(seed)
    model = tranception.model_pytorch.TranceptionLMHeadModel(cfg)
    g = torch.Generator().manual_seed(500 + seed)
    sd = model.state_dict()
    with torch.no_grad():
        for k, v in sd.items():
            if k.endswith(".attn.bias") or k.endswith("masked_bias") or k.endswith("alibi"):
                continue
            if k.endswith("wte.weight"):
                v.copy_(torch.randn(v.shape, generator=g) * embed_std)
            elif "ln_" in k:
                v.copy_((1.0 if k.endswith("weight") else 0.0) + (0.1 if k.endswith("weight") else 0.05) * torch.randn(v.shape, generator=g))
            elif "depthwiseconv" in k:
                v.copy_(torch.randn(v.shape, generator=g) * (0.4 if k.endswith("weight") else 0.05))
            elif k.endswith(".bias"):
                v.copy_(0.02 * torch.randn(v.shape, generator=g))
            elif k.endswith("c_attn.weight") or k.endswith("c_fc.weight") or k.endswith("c_proj.weight"):
                v.copy_(torch.randn(v.shape, generator=g) / (v.shape[0] ** 0.5))      # Conv1D: [in, out]
        sd["lm_head.weight"] = sd["transformer.wte.weight"]                            # tied (GPT2 default)
    os.makedirs(dirpath, exist_ok=True)
    keep = {k: v.clone() for k, v in sd.items()
            if not (k.endswith(".attn.bias") or k.endswith("masked_bias"))}      # causal-mask buffers
    torch.save(keep, os.path.join(dirpath, "pytorch_model.bin"))
    jc = dict(vocab_size=25, n_embd=n_embd, n_head=n_head, n_layer=n_layer, n_positions=n_ctx, n_ctx=n_ctx,
              n_inner=None, activation_function="squared_relu", layer_norm_epsilon=cfg.layer_norm_epsilon,
              scale_attn_weights=True, attention_mode="tranception", position_embedding="grouped_alibi",
              model_type="tranception", architectures=["TranceptionLMHeadModel"])
    json.dump(jc, open(os.path.join(dirpath, "config.json"), "w"), indent=1)
    return dirpath


def reference_tranception_model(dirpath, scoring_window="optimal", retrieval=None):
    """Instantiate the reference model from a checkpoint directory the way
    score_tranception_proteingym.py:79-104 does (config.json -> TranceptionConfig, forced
    attention_mode / position_embedding, tokenizer, optional retrieval fields) and load the
    weights with load_state_dict (from_pretrained's file handling differs across transformers
    versions; the tensors and module code are the reference's)."""
    import json
    import torch
    tranception, tok = load_reference_tranception()
    c = json.load(open(os.path.join(dirpath, "config.json")))
    c.pop("model_type", None)
    c.pop("architectures", None)
    cfg = tranception.config.TranceptionConfig(**c)
    cfg.attention_mode = "tranception"
    cfg.position_embedding = "grouped_alibi"
    cfg.tokenizer = tok
    cfg.scoring_window = scoring_window
    if retrieval:
        for k, v in retrieval.items():
            setattr(cfg, k, v)
    else:
        cfg.retrieval_aggregation_mode = None
    model = tranception.model_pytorch.TranceptionLMHeadModel(cfg)
    sd = torch.load(os.path.join(dirpath, "pytorch_model.bin"), map_location="cpu")
    missing, unexpected = model.load_state_dict(sd, strict=False)
    bad = [k for k in missing if not (k.endswith(".attn.bias") or k.endswith("masked_bias") or k.endswith("alibi"))]
    assert not bad and not unexpected, (bad, unexpected)
    model.eval()
    return model, tok


# ---- proteingym/utils/weights.py (numba) --------------------------------------------------------
def load_reference_weights():
    """Import the reference's proteingym/utils/weights.py unmodified.  numba is not installed in this
    image; a stub module whose ``jit`` is the identity decorator and whose ``prange`` is ``range`` runs
    the same function bodies as plain python (slow: small alignments only)."""
    import importlib.util
    import types
    if "numba" not in sys.modules:
        nb = types.ModuleType("numba")
        nb.jit = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda f: f))
        nb.prange = lambda n: range(int(n))
        nb.set_num_threads = lambda n: None
        nb.get_num_threads = lambda: 1
        nb.config = types.SimpleNamespace(NUMBA_NUM_THREADS=1)
        sys.modules["numba"] = nb
    spec = importlib.util.spec_from_file_location("pg_ref_weights", os.path.join(REF_ROOT, "proteingym", "utils", "weights.py"))
    mod = importlib.util.module_from_spec(spec)
    # the jitted bodies do ``L = 1.0 * L; for k in range(L)`` (numba accepts the float bound): give the
    # module a ``range`` that truncates like numba does, instead of touching the source
    mod.range = lambda *a: range(*[int(v) for v in a])
    spec.loader.exec_module(mod)
    return mod


# ---- MSA Transformer (esm/model/msa_transformer.py) -----------------------------------------------
def make_msa_transformer_checkpoint(path, layers, embed_dim, ffn_dim, heads, seed, max_positions=1024,
                                    embed_std=0.25, msa_pos_std=0.1):
    """Random-weight checkpoint in the fair-esm *v1* file layout of esm_msa1b_t12_100M_UR50S
    (pretrained.py:107-121: ``args.arch == "msa_transformer"``, keys with ``encoder.`` /
    ``sentence_encoder.`` prefixes and the released file's swapped row/column naming, which the loader
    swaps back).  The model is built by the reference constructor (msa_transformer.py:84-145)."""
    import torch
    load_reference()
    esm = sys.modules["esm"]
    torch.manual_seed(seed)
    alphabet = esm.data.Alphabet.from_architecture("msa_transformer")
    margs = argparse.Namespace(arch="msa_transformer", layers=layers, embed_dim=embed_dim, ffn_embed_dim=ffn_dim,
                               attention_heads=heads, dropout=0.1, attention_dropout=0.1, activation_dropout=0.1,
                               max_positions=max_positions, embed_positions_msa=True, max_tokens_per_msa=2 ** 14,
                               max_tokens=2 ** 14)
    model = esm.model.msa_transformer.MSATransformer(margs, alphabet)
    sd = model.state_dict()
    _randomize(sd, embed_std, True, seed)
    g = torch.Generator().manual_seed(seed + 99)
    sd["msa_position_embedding"] = torch.randn(sd["msa_position_embedding"].shape, generator=g) * msa_pos_std
    swap = lambda s: s.replace("row", "column") if "row" in s else s.replace("column", "row")
    out = {}
    for k, v in sd.items():
        if k.startswith("contact_head"):
            continue
        pref = "encoder." if k.startswith("lm_head") else "encoder.sentence_encoder."
        out[pref + swap(k)] = v.clone()
    fargs = argparse.Namespace(arch="msa_transformer", encoder_layers=layers, encoder_embed_dim=embed_dim,
                               encoder_ffn_embed_dim=ffn_dim, encoder_attention_heads=heads, dropout=0.1,
                               attention_dropout=0.1, activation_dropout=0.1, max_positions=max_positions,
                               embed_positions_msa=True, max_tokens_per_msa=2 ** 14, max_tokens=2 ** 14)
    torch.save({"args": fargs, "model": out}, path)
    return path
