"""Host-side mirror of the reference's MSA Transformer scoring path, backed by libpgmi.so (HIP, gfx950).

Reference (under /root/reference/proteingym/baselines/esm unless noted):
  * checkpoint loading, row/column key swap        esm/pretrained.py:107-121,184-218
  * "msa_transformer" alphabet, MSABatchConverter  esm/data.py:158-164,300-334
  * MSATransformer.forward                         esm/model/msa_transformer.py:146-205  (device: run_msa, csrc/api_msa.hip)
  * sample_msa / process_msa                       compute_fitness.py:26-98
  * masked-marginals over the first row            compute_fitness.py:380-399
  * MSA_processing (EVE pre-processing + weights)  proteingym/utils/msa_utils.py:24-258
The network (embeddings, tied row attention, column attention, feed forward, LM head) runs in HIP kernels
through the C ABI (``pgmi_msa_token_logprobs`` / ``pgmi_msa_masked_logprobs``); no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import itertools
import os
import random
from typing import List, Tuple

import numpy as np

from . import _lib, alignment
from ._lib import Config, PgmiError
from . import esm as pesm

ARCH_MSA = 4
GAP = "-"
ALPHABET_PROTEIN_NOGAP = "ACDEFGHIKLMNPQRSTVWY"
ALPHABET_PROTEIN_GAP = GAP + ALPHABET_PROTEIN_NOGAP


class MsaAlphabet(pesm.Alphabet):
    """Same 33 symbols as ESM-1b; <cls> is prepended, no <eos> (esm/data.py:158-164)."""

    append_eos = False
    use_msa = True

    def get_batch_converter(self, truncation_seq_length: int = None):
        return MSABatchConverter(self)


class MSABatchConverter:
    """esm/data.py:300-334: one MSA (list of (label, seq)) or a list of MSAs -> int64 [B, R, C+1]."""

    def __init__(self, alphabet):
        self.alphabet = alphabet

    def __call__(self, inputs):
        raw_batch = [inputs] if isinstance(inputs[0][0], str) else inputs
        batch_size = len(raw_batch)
        max_alignments = max(len(msa) for msa in raw_batch)
        max_seqlen = max(len(msa[0][1]) for msa in raw_batch)
        tokens = np.full((batch_size, max_alignments, max_seqlen + 1), self.alphabet.padding_idx, dtype=np.int64)
        labels, strs = [], []
        for i, msa in enumerate(raw_batch):
            if len(set(len(seq) for _, seq in msa)) != 1:
                raise RuntimeError("Received unaligned sequences for input to MSA, all sequence lengths must be equal.")
            labels.append([l for l, _ in msa])
            strs.append([s for _, s in msa])
            for r, (_, seq) in enumerate(msa):
                tokens[i, r, 0] = self.alphabet.cls_idx
                tokens[i, r, 1:len(seq) + 1] = self.alphabet.encode(seq)
        return labels, strs, tokens


# ---- checkpoint -------------------------------------------------------------------------------------
def _upgrade_state_dict(path: str):
    data = pesm.load_checkpoint_file(path)              # weights_only=True + argparse.Namespace (see esm.load_checkpoint_file)
    a = data["args"]
    if a.arch != "msa_transformer":
        raise ValueError("Unknown architecture selected")
    def swap_axes_names(key):          # the released file names the two attention blocks the other way round (pretrained.py:114)
        return key.replace("row", "column") if "row" in key else key.replace("column", "row")
    sd = {pesm.strip_fairseq_prefixes(swap_axes_names(k)): v for k, v in data["model"].items()}
    sd = {k: v for k, v in sd.items() if not k.startswith("contact_head")}
    if "lm_head.weight" in sd:                                      # tied parameter, entry copied last wins
        sd["embed_tokens.weight"] = sd["lm_head.weight"]
    cfg = dict(arch=ARCH_MSA, layers=int(a.encoder_layers), embed_dim=int(a.encoder_embed_dim),
               heads=int(a.encoder_attention_heads), ffn_dim=int(a.encoder_ffn_embed_dim),
               max_positions=int(a.max_positions), token_dropout=0, emb_layer_norm_before=1,
               embed_positions_msa=bool(getattr(a, "embed_positions_msa", False)))
    return cfg, sd


def expected_keys(cfg) -> List[str]:
    keys = ["embed_tokens.weight", "embed_positions.weight", "msa_position_embedding",
            "emb_layer_norm_before.weight", "emb_layer_norm_before.bias"]
    for i in range(cfg["layers"]):
        for blk in ("row_self_attention", "column_self_attention"):
            p = f"layers.{i}.{blk}."
            keys += [p + "layer_norm.weight", p + "layer_norm.bias"]
            for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
                keys += [p + f"layer.{n}.weight", p + f"layer.{n}.bias"]
        p = f"layers.{i}.feed_forward_layer."
        keys += [p + "layer_norm.weight", p + "layer_norm.bias", p + "layer.fc1.weight", p + "layer.fc1.bias",
                 p + "layer.fc2.weight", p + "layer.fc2.bias"]
    keys += ["emb_layer_norm_after.weight", "emb_layer_norm_after.bias", "lm_head.dense.weight", "lm_head.dense.bias",
             "lm_head.layer_norm.weight", "lm_head.layer_norm.bias", "lm_head.bias"]
    return keys


def pack_state_dict(cfg, sd) -> np.ndarray:
    keys = expected_keys(cfg)
    sd = dict(sd)
    D = cfg["embed_dim"]
    if not cfg.get("embed_positions_msa", True) or "msa_position_embedding" not in sd:
        sd["msa_position_embedding"] = np.zeros((1024, D), np.float32)        # register_parameter(None): nothing is added
    missing = [k for k in keys if k not in sd]
    unexpected = [k for k in sd if k not in keys and k != "lm_head.weight"]
    msgs = []
    if missing:
        msgs.append(f"Missing key(s) in state_dict: {set(missing)}.")
    if unexpected:
        msgs.append(f"Unexpected key(s) in state_dict: {set(unexpected)}.")
    if msgs:
        raise RuntimeError("Error(s) in loading state_dict for MSATransformer:\n\t" + "\n\t".join(msgs))
    parts = []
    for k in keys:
        t = sd[k]
        a = t.detach().to("cpu").float().numpy() if hasattr(t, "detach") else np.asarray(t, np.float32)
        if k == "msa_position_embedding":
            a = np.broadcast_to(a.reshape(1024, -1), (1024, D))       # [1,1024,1,D] or the first release's [1,1024,1,1]
        parts.append(np.ascontiguousarray(a, dtype=np.float32).ravel())
    return np.concatenate(parts)


class MsaTransformerModel:
    """Device-resident MSA Transformer.  ``model(tokens)["logits"]`` mirrors the reference call
    (compute_fitness.py:390) and returns log-probabilities [B, R, C, 33] (log_softmax is idempotent)."""

    def __init__(self, cfg: dict, weights: np.ndarray, device: int = 0, max_rows: int = 0):
        lib = _lib.load()
        self.cfg = dict(cfg)
        self.precision = "f16x3"
        if max_rows <= 0:
            max_rows = 416 * 1024                                   # 400 sampled rows x 1024 columns, padded to 32
        c = Config(abi_version=_lib.ABI_VERSION, arch=ARCH_MSA, layers=cfg["layers"], embed_dim=cfg["embed_dim"],
                   heads=cfg["heads"], ffn_dim=cfg["ffn_dim"], vocab=33, max_positions=cfg["max_positions"],
                   token_dropout=0, emb_layer_norm_before=1, precision=_lib.PRECISIONS["f16x3"], max_rows=max_rows,
                   ln_eps=0.0)
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

    def token_logprobs(self, tokens) -> np.ndarray:
        t = _lib.as_i32(np.asarray(tokens))
        if t.ndim != 2:
            raise ValueError("tokens must be [R, C]")
        out = np.empty(t.shape + (33,), np.float32)
        _lib.check(_lib.load().pgmi_msa_token_logprobs(self._h, _lib.ptr(t, _lib._i32p), t.shape[0], t.shape[1],
                                                       _lib.ptr(out, _lib._f32p)))
        return out

    def __call__(self, tokens, **_):
        t = np.asarray(tokens)
        assert t.ndim == 3
        return {"logits": np.stack([self.token_logprobs(x) for x in t])}

    def masked_logprobs(self, tokens, positions, seq_len: int, window: int = 1024) -> np.ndarray:
        """Rows of the masked-marginals table (compute_fitness.py:380-394): for every column in ``positions``
        mask it in the first row, forward the alignment (cropped to the optimal ``window`` columns when it has
        more), return log-probabilities of that cell: [len(positions), 33]."""
        t = _lib.as_i32(np.asarray(tokens))
        R, T = t.shape
        pos = _lib.as_i32(np.asarray(positions))
        if T > window:
            starts = _lib.as_i32([pesm.get_optimal_window(int(p), seq_len + 2, window)[0] for p in pos])
            w = window
        else:
            starts = np.zeros(len(pos), np.int32)
            w = T
        out = np.empty((len(pos), 33), np.float32)
        _lib.check(_lib.load().pgmi_msa_masked_logprobs(self._h, _lib.ptr(t, _lib._i32p), R, T, w, _lib.ptr(pos, _lib._i32p),
                                                        _lib.ptr(starts, _lib._i32p), len(pos), _lib.ptr(out, _lib._f32p)))
        return out


def load_model_and_alphabet(model_location: str, device: int = 0, max_rows: int = 0):
    cfg, sd = _upgrade_state_dict(model_location)
    return MsaTransformerModel(cfg, pack_state_dict(cfg, sd), device=device, max_rows=max_rows), MsaAlphabet()


# ---- alignment handling -----------------------------------------------------------------------------
class MSA_processing:
    """proteingym/utils/msa_utils.py:24-258 -- the attributes the scorer reads (focus_seq_name,
    raw_seq_name_to_sequence, seq_name_to_sequence, seq_name_to_weight, weights, Neff, num_sequences).
    Weights are loaded from ``weights_location`` when the file exists, else computed on the HIP device
    (``weights.calc_weights_fast``, the numba kernel's replacement) and saved there, like the reference."""

    def __init__(self, MSA_location="", theta=0.2, use_weights=True, weights_location="./data/weights",
                 preprocess_MSA=True, threshold_sequence_frac_gaps=0.5, threshold_focus_cols_frac_gaps=1.0,
                 remove_sequences_with_indeterminate_AA_in_focus_cols=True, weights_calc_method="eve", num_cpus=1,
                 skip_one_hot_encodings=False, device=0):
        np.random.seed(2021)
        self.MSA_location = MSA_location
        self.weights_location = weights_location
        self.theta = theta
        self.alphabet = ALPHABET_PROTEIN_NOGAP
        self.use_weights = use_weights
        self.device = device
        al = alignment.FocusAlignment(MSA_location, preprocess_MSA, threshold_sequence_frac_gaps, threshold_focus_cols_frac_gaps,
                                      remove_sequences_with_indeterminate_AA_in_focus_cols)
        self.focus_seq_name, self.focus_seq = al.focus_name, al.focus_seq
        self.focus_cols = al.focus_cols.tolist()
        self.focus_seq_trimmed = "".join(self.focus_seq[c] for c in self.focus_cols)
        self.seq_len = len(self.focus_cols)
        self.alphabet_size = len(self.alphabet)
        self.raw_seq_name_to_sequence = al.raw
        self.seq_name_to_sequence = dict(zip(al.names, alignment.to_strings(al.trimmed)))
        self.num_sequences = len(al.names)
        if use_weights:
            if os.path.isfile(str(weights_location)):
                self.weights = np.load(file=weights_location)
            else:
                from . import weights as _w
                mat = _w.symbol_table(ALPHABET_PROTEIN_GAP, default=GAP)[al.trimmed].astype(np.int8)
                self.weights = _w.calc_weights_fast(mat, identity_threshold=1 - theta, empty_value=0, num_cpus=num_cpus,
                                                    device=device)
                np.save(file=weights_location, arr=self.weights)
        else:
            self.weights = np.ones(self.num_sequences)
        self.Neff = np.sum(self.weights)
        assert self.weights.shape[0] == self.num_sequences, \
            f"Expected {self.num_sequences} sequences, loaded weights have {self.weights.shape[0]}"
        self.seq_name_to_weight = {n: self.weights[i] for i, n in enumerate(al.names)}


def read_fasta_records(filename):
    """(description, sequence) pairs of a FASTA/a2m file -- what the reference gets from Bio.SeqIO.parse
    (record.description is the header line without '>')."""
    desc, chunks = None, []
    with open(filename) as f:
        for line in f:
            line = line.rstrip("\n")
            if line.startswith(">"):
                if desc is not None:
                    yield desc, "".join(chunks)
                desc, chunks = line[1:].strip(), []
            elif desc is not None:
                chunks.append(line.strip())
    if desc is not None:
        yield desc, "".join(chunks)


def sample_msa(filename, nseq: int, sampling_strategy: str, random_seed: int, weight_filename=None, processed_msa=None,
               device=0) -> List[Tuple[str, str]]:
    """compute_fitness.py:26-73 (python's ``random`` with the same seed -> the same rows as the reference)."""
    random.seed(random_seed)
    if sampling_strategy == "first_x_rows":
        msa = list(itertools.islice(read_fasta_records(filename), nseq))
    elif sampling_strategy == "random":
        msa = list(read_fasta_records(filename))
        nseq = min(len(msa), nseq)
        msa = random.sample(msa, nseq)
    elif sampling_strategy == "sequence-reweighting":
        MSA = processed_msa if processed_msa is not None else MSA_processing(MSA_location=filename, use_weights=True,
                                                                             weights_location=weight_filename, device=device)
        msa = [(MSA.focus_seq_name, MSA.raw_seq_name_to_sequence[MSA.focus_seq_name])]
        non_wt_weights = np.array([w for k, w in MSA.seq_name_to_weight.items() if k != MSA.focus_seq_name])
        non_wt_sequences = [(k, s) for k, s in MSA.seq_name_to_sequence.items() if k != MSA.focus_seq_name]
        non_wt_weights = non_wt_weights / non_wt_weights.sum()
        if len(non_wt_sequences) > 0:
            msa.extend(random.choices(non_wt_sequences, weights=non_wt_weights, k=nseq - 1))
    else:
        raise ValueError("unknown --msa-sampling-strategy " + str(sampling_strategy))
    msa = [(desc, "".join(seq) if isinstance(seq, list) else seq) for desc, seq in msa]
    return [(desc, seq.upper()) for desc, seq in msa]


def hhfilter_alignment(filename, path_to_hhfilter, min_cov=75, max_seq_id=100, min_seq_id=0) -> str:
    """compute_fitness.py:77-89: the alignment with every '.' turned into '-' and every byte upper case (headers included, as the
    reference's ``tr`` and ``dd conv=ucase`` do) is written to ``<folder>/preprocessed/<name>_UC.a2m`` and handed to the user's
    ``<path_to_hhfilter>/bin/hhfilter -cov C -id I -qid Q -i ... -o <folder>/hhfiltered/<name>_hhfiltered_cov_C_maxid_I_minid_Q.a2m``;
    returns that output path.  (The reference APPENDS to its intermediate file, so a second run filters every sequence twice; here
    the file is rewritten.)"""
    import subprocess
    folder, name = os.path.dirname(filename), os.path.basename(filename).split(".")[0]
    for sub in ("preprocessed", "hhfiltered"):
        os.makedirs(os.path.join(folder, sub), exist_ok=True)
    upper = os.path.join(folder, "preprocessed", name + "_UC.a2m")
    with open(filename) as f, open(upper, "w") as g:
        g.write(f.read().replace(".", "-").upper())
    out = os.path.join(folder, "hhfiltered", f"{name}_hhfiltered_cov_{min_cov}_maxid_{max_seq_id}_minid_{min_seq_id}.a2m")
    exe = os.path.join(str(path_to_hhfilter), "bin", "hhfilter")
    try:
        done = subprocess.run([exe, "-cov", str(min_cov), "-id", str(max_seq_id), "-qid", str(min_seq_id), "-i", upper, "-o", out],
                              capture_output=True, text=True)
    except OSError as e:
        raise RuntimeError(f"--filter-msa: cannot run {exe} ({e}); --path-to-hhfilter must point at an hh-suite installation") from e
    if done.returncode != 0 or not os.path.exists(out):
        raise RuntimeError(f"--filter-msa: {exe} failed ({done.returncode}): {done.stderr.strip()[-500:]}")
    return out


def process_msa(filename, weight_filename, filter_msa=False, path_to_hhfilter=None, hhfilter_min_cov=75, hhfilter_max_seq_id=100,
                hhfilter_min_seq_id=0, device=0):
    """compute_fitness.py:76-97: optional hhfilter pre-filtering (the user's executable), then the EVE pre-processing and weights."""
    if filter_msa:
        filename = hhfilter_alignment(filename, path_to_hhfilter, hhfilter_min_cov, hhfilter_max_seq_id, hhfilter_min_seq_id)
    return MSA_processing(MSA_location=filename, use_weights=True, weights_location=weight_filename, device=device)
