"""Host-side mirror of the reference's Tranception scorer, backed by libpgmi.so (HIP, gfx950).

Reference (all under /root/reference/proteingym/baselines/tranception):
  * ``TranceptionLMHeadModel.score_mutants``                        tranception/model_pytorch.py:878-928
  * ``get_sequence_slices`` / ``get_tranception_scores_mutated_sequences``
                                                                    tranception/utils/scoring_utils.py:77-203
  * ``encode_batch`` + Basic_tokenizer (25 symbols)                 model_pytorch.py:930-939
  * retrieval prior ``get_msa_prior``                               tranception/utils/msa_utils.py:63-138
  * inference-time retrieval fusion                                 model_pytorch.py:806-830 (on the device)
Same method names, arguments and output columns; the network itself (36 x [LN, c_attn, depth-wise
conv on q/k/v, causal grouped-ALiBi attention, c_proj, LN, squared-ReLU MLP], LM head, per-token
log-likelihood, prior fusion, per-sequence sum) runs in HIP kernels through the C ABI.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from typing import Optional, Sequence

import numpy as np
import pandas as pd

from . import _lib, alignment
from ._lib import PgmiError, Config

VOCAB = {'[UNK]': 0, '[CLS]': 1, '[SEP]': 2, '[PAD]': 3, '[MASK]': 4, 'A': 5, 'C': 6, 'D': 7, 'E': 8, 'F': 9,
         'G': 10, 'H': 11, 'I': 12, 'K': 13, 'L': 14, 'M': 15, 'N': 16, 'P': 17, 'Q': 18, 'R': 19, 'S': 20,
         'T': 21, 'V': 22, 'W': 23, 'Y': 24}
CLS, SEP, PAD, UNK = 1, 2, 3, 0
AA_vocab = "ACDEFGHIKLMNPQRSTVWY"


# ---- scoring_utils mirrors ----------------------------------------------------------------------
def _parse_substitutions(mutants, start_idx):
    """The 'A25G:K30R' strings of a whole column as flat arrays: (row of each substitution, 0-based position, from byte, to byte)."""
    rows, pos, frm, to = [], [], bytearray(), bytearray()
    for i, m in enumerate(mutants):
        for one in m.split(":"):
            rows.append(i)
            pos.append(int(one[1:-1]) - start_idx)
            frm += one[0].encode()
            to += one[-1].encode()
    return (np.asarray(rows, dtype=np.int64), np.asarray(pos, dtype=np.int64),
            np.frombuffer(bytes(frm), dtype=np.uint8), np.frombuffer(bytes(to), dtype=np.uint8))


def mutated_sequences(focus_seq, mutants, start_idx=1, AA_vocab=AA_vocab):
    """``get_mutated_sequence`` (scoring_utils.py:16-31) for a whole 'mutant' column: one [rows, L] byte matrix of wild-type copies,
    every substitution written with one fancy assignment (a later substitution at the same position wins, as in the reference's
    loop).  Same checks in the same order -- per substitution: the from letter against the UNMUTATED sequence, then the to letter
    against the amino-acid alphabet -- with the reference's messages; negative positions index from the end like python strings."""
    mutants = list(mutants)
    wt = np.frombuffer(focus_seq.encode("ascii"), dtype=np.uint8)
    rows, pos, frm, to = _parse_substitutions(mutants, start_idx)
    if ((pos >= len(wt)) | (pos < -len(wt))).any():
        raise IndexError("string index out of range")
    known = np.zeros(256, dtype=bool)
    known[np.frombuffer(AA_vocab.encode("ascii"), dtype=np.uint8)] = True
    bad = np.flatnonzero((wt[pos] != frm) | ~known[to])
    if bad.size:
        k = int(bad[0])
        label = mutants[rows[k]].split(":")[int(k - np.searchsorted(rows, rows[k]))]
        if wt[pos[k]] != frm[k]:
            raise AssertionError("Invalid from_AA or mutant position: " + label + " from_AA: " + chr(frm[k]) + " relative pos: " +
                                 str(int(pos[k])) + " focus_seq: " + str(focus_seq))
        raise AssertionError("Mutant to_AA is invalid: " + label)
    out = np.tile(wt, (len(mutants), 1))
    out[rows, pos] = to
    return alignment.to_strings(out)


def get_mutated_sequence(focus_seq, mutant, start_idx=1, AA_vocab=AA_vocab):
    """scoring_utils.py:16-31 (one mutant; columns go through ``mutated_sequences``)."""
    return mutated_sequences(focus_seq, [mutant], start_idx, AA_vocab)[0]


def get_optimal_window(mutation_position_relative, seq_len_wo_special, model_window):
    """scoring_utils.py:47-60 -- the same rule as proteingym/utils/scoring_utils.py:43-52: the library's ``pgmi_optimal_window``."""
    from .esm import get_optimal_window as c_window
    return c_window(mutation_position_relative, seq_len_wo_special, model_window)


def optimal_windows(centres, seq_len_wo_special, model_window):
    """``get_optimal_window`` for an array of positions: int64 [n, 2] of (start, end)."""
    c = np.asarray(centres, dtype=np.int64)
    n, w, half = int(seq_len_wo_special), int(model_window), int(model_window) // 2
    if n <= w:
        return np.tile(np.array([0, n], dtype=np.int64), (len(c), 1))
    lo = np.where(c < half, 0, np.where(c >= n - half, n - w, np.maximum(0, c - half)))
    hi = np.where(c < half, w, np.where(c >= n - half, n, np.minimum(n, c + half)))
    return np.stack([lo, hi], axis=1)


def sequence_replace_single(sequence, char_to_replace, char_replacements):
    """scoring_utils.py:62-69: every ``char_to_replace`` becomes a letter drawn by ONE ``np.random.choice`` call over all of its
    positions (the reference's draw, so a seeded run replaces the same letters)."""
    codes = np.frombuffer(sequence.encode("ascii"), dtype=np.uint8)
    where = np.flatnonzero(codes == ord(char_to_replace))
    if not where.size:
        return sequence
    drawn = np.random.choice(a=list(char_replacements), size=where.size, replace=True)
    codes = codes.copy()
    codes[where] = np.frombuffer("".join(drawn).encode("ascii"), dtype=np.uint8)
    return codes.tobytes().decode("ascii")


def wild_type_rows(slices, target_seq):
    """For every row of ``get_sequence_slices``' frame the position of the wild-type row cut to the same window (the row the
    reference subtracts, scoring_utils.py:142-148), its own position where there is none: what ``sequence_loglik`` shares prefixes with."""
    n = len(slices)
    own = np.arange(n, dtype=np.int64)
    if target_seq is None or n == 0:
        return own
    wt = (slices['mutated_sequence'] == target_seq).to_numpy()
    starts, ends = slices['window_start'].to_numpy(), slices['window_end'].to_numpy()
    first = {}
    for i in np.flatnonzero(wt):
        first.setdefault((int(starts[i]), int(ends[i])), int(i))
    return np.array([first.get((int(a), int(b)), int(i)) for i, (a, b) in enumerate(zip(starts, ends))], dtype=np.int64)


def get_sequence_slices(df, target_seq, model_context_len, start_idx=1, scoring_window="optimal", indel_mode=False):
    """Behaviour of scoring_utils.py:152-203, written on plain lists: every input row yields a (mutated, window)
    row and a wild-type row cropped to the SAME window; rows identical in every column are merged; the 'mutant'
    column is dropped.  Columns added: sliced_mutated_sequence, window_start, window_end."""
    L = len(target_seq)
    base = df.reset_index(drop=True)
    seqs = base["mutated_sequence"].tolist()
    keep = base.drop(columns=[c for c in ("mutant",) if c in base.columns])

    def block(sequences, windows):
        out = keep.copy()
        out["mutated_sequence"] = sequences
        out["sliced_mutated_sequence"] = [s[w0:w1] for s, (w0, w1) in zip(sequences, windows)]
        out["window_start"] = [w0 for w0, _ in windows]
        out["window_end"] = [w1 for _, w1 in windows]
        return out

    if scoring_window == "optimal":
        if indel_mode:                                      # whole sequence; the wild type keeps its own length
            win_mut = [(0, len(s)) for s in seqs]
            win_wt = [(0, L)] * len(seqs)
        else:                                               # window centred on the mean mutated position
            rows, pos, _, _ = _parse_substitutions(base["mutant"], start_idx)
            depth = np.bincount(rows, minlength=len(seqs))
            centres = (np.bincount(rows, weights=pos, minlength=len(seqs)) / depth).astype(np.int64)   # int(np.mean(...)): exact sums of small integers
            win_mut = [(int(a), int(b)) for a, b in optimal_windows(centres, L, model_context_len)]
            win_wt = win_mut
        blocks = [block(seqs, win_mut), block([target_seq] * len(seqs), win_wt)]
    elif scoring_window == "sliding":                       # consecutive context-sized chunks, mutated then wild type
        blocks = []
        for k in range(1 + int(L / model_context_len)):
            lo = k * model_context_len
            blocks.append(block(seqs, [(lo, min(len(s), lo + model_context_len)) for s in seqs]))
            blocks.append(block([target_seq] * len(seqs), [(lo, min(L, lo + model_context_len))] * len(seqs)))
    else:
        raise ValueError("scoring_window must be 'optimal' or 'sliding'")
    return pd.concat(blocks, axis=0).drop_duplicates().reset_index(drop=True)


# ---- retrieval prior (tranception/utils/msa_utils.py:28-138) -------------------------------------
def process_msa_data(MSA_data_file):
    """{header line -> upper-case sequence} (msa_utils.py:28-43)."""
    return alignment.read_records(MSA_data_file, upper=True)[1]


class MSA_processing:
    """EVE-style alignment pre-processing and sequence weights (tranception/utils/msa_utils.py:194-368): the reference's constructor
    arguments, and the attributes its callers read.  The pre-processing itself is ``alignment.FocusAlignment`` (numpy over the byte
    matrix).  Weights are loaded from ``weights_location`` when the file exists, otherwise computed (1 / number of sequences within
    Hamming distance theta over the focus columns, by the HIP pair-count kernel on ``device``) and saved there, as the reference does."""

    def __init__(self, MSA_location="", theta=0.2, use_weights=True, weights_location="./data/weights",
                 preprocess_MSA=True, threshold_sequence_frac_gaps=0.5, threshold_focus_cols_frac_gaps=1.0,
                 remove_sequences_with_indeterminate_AA_in_focus_cols=True, device=0):
        np.random.seed(2021)
        self.device = device                 # additive: HIP device of the O(N^2 L) weight computation
        self.MSA_location = MSA_location
        self.weights_location = weights_location
        self.theta = theta
        self.alphabet = alignment.AMINO_ACIDS
        self.use_weights = use_weights
        self.preprocess_MSA = preprocess_MSA
        self.threshold_sequence_frac_gaps = threshold_sequence_frac_gaps
        self.threshold_focus_cols_frac_gaps = threshold_focus_cols_frac_gaps
        self.remove_sequences_with_indeterminate_AA_in_focus_cols = remove_sequences_with_indeterminate_AA_in_focus_cols
        self.gen_alignment()

    def gen_alignment(self):
        al = alignment.FocusAlignment(self.MSA_location, self.preprocess_MSA, self.threshold_sequence_frac_gaps,
                                      self.threshold_focus_cols_frac_gaps, self.remove_sequences_with_indeterminate_AA_in_focus_cols)
        self.aa_dict = {aa: i for i, aa in enumerate(self.alphabet)}
        self.alphabet_size = len(self.alphabet)
        self.focus_seq_name, self.focus_seq = al.focus_name, al.focus_seq
        self.focus_cols = al.focus_cols.tolist()
        self.focus_seq_trimmed = [self.focus_seq[c] for c in self.focus_cols]
        self.seq_len = len(self.focus_cols)
        self.focus_start_loc, self.focus_stop_loc = al.focus_range()
        self.uniprot_focus_col_to_wt_aa_dict = {c + self.focus_start_loc: self.focus_seq[c] for c in self.focus_cols}
        self.uniprot_focus_col_to_focus_idx = {c + self.focus_start_loc: c for c in self.focus_cols}
        self.raw_seq_name_to_sequence = al.raw
        self.seq_name_to_sequence = {n: list(s) for n, s in zip(al.names, alignment.to_strings(al.trimmed))}
        self.encoded = al.residue_codes(gap=-1)          # 0..19, -1 for gaps (the reference's all-zero one-hot rows)
        self.num_sequences = len(al.names)
        if self.use_weights:
            try:
                self.weights = np.load(file=self.weights_location)
            except Exception:
                self.weights = compute_sequence_weights(self.encoded, self.theta, device=self.device)
                np.save(file=self.weights_location, arr=self.weights)
        else:
            self.weights = np.ones(self.num_sequences)
        self.Neff = np.sum(self.weights)
        self.seq_name_to_weight = {n: self.weights[i] for i, n in enumerate(al.names)}


def compute_sequence_weights(enc: np.ndarray, theta: float, device: int = 0) -> np.ndarray:
    """msa_utils.py:341-352: weight_i = 1 / #{j : <x_j, x_i> / <x_i, x_i> > 1 - theta} on one-hot encodings, i.e. matches over the
    non-gap positions of i (0 for an all-gap sequence).  The O(N^2 L) pair count is the HIP kernel ``pgmi_msa_cluster_counts``
    (csrc/msa_weights.hip); there is no host version in this package (``PgmiError`` without the library or a GPU)."""
    from . import weights as _w
    counts = _w.num_cluster_members(enc, 1 - theta, -1, device=device)
    return np.where(counts > 0, 1.0 / np.maximum(counts, 1), 0.0)


def get_msa_prior(MSA_data_file, MSA_weight_file_name, MSA_start, MSA_end, len_target_seq, vocab=VOCAB,
                  retrieval_aggregation_mode="aggregate_substitution", filter_MSA=True, seq_name_to_weight=None,
                  block_bytes=64 << 20):
    """Per-position amino-acid distribution of the retrieved alignment with 1e-5 pseudo-counts, [len_target_seq, V],
    zero outside [MSA_start, MSA_end) -- the quantity msa_utils.py:63-138 builds.  Sequences sharing less than 20 %
    of the query's symbols are dropped (:83-91); with ``MSA_weight_file_name`` the EVE weights come from
    ``MSA_processing`` and sequences without a weight are dropped (:100-115); ``seq_name_to_weight`` injects weights
    directly (additive, like ``block_bytes``: the size of the float64 working set).  The float expression of the reference
    is kept term by term so the result is bit-identical."""
    V = len(vocab)
    records = process_msa_data(MSA_data_file)
    names = list(records)
    width = MSA_end - MSA_start
    table = np.full(256, -1, dtype=np.int8)               # byte -> vocabulary index (25 symbols), -1 = not in the vocabulary
    for symbol, index in vocab.items():
        if len(symbol) == 1:
            table[ord(symbol)] = index
    rows = [records[n] for n in names]
    lengths = np.array([len(r) for r in rows], dtype=np.int64)
    if filter_MSA and len(rows) and (lengths != lengths[0]).any():
        # the reference's similarity filter takes np.dot of two flattened one-hot rows (:83-91): rows of another length end there
        k = int(np.flatnonzero(lengths != lengths[0])[0])
        raise ValueError(f"shapes ({lengths[0] * V},) and ({lengths[k] * V},) not aligned: alignment row {k} differs in length from the query")
    # one byte per alignment cell; short rows (tolerated by the reference without the filter: their tail stays all-zero, :44-52) are
    # padded with a byte outside the vocabulary
    full = int(lengths.max()) if len(rows) else 0
    codes = np.full((len(rows), full), -1, dtype=np.int8)
    for i, r in enumerate(rows):
        codes[i, :len(r)] = table[np.frombuffer(r.encode("ascii"), dtype=np.uint8)]
    keep = np.ones(len(names), dtype=bool)
    if filter_MSA:
        query = codes[0]
        n_query = float((query >= 0).sum())
        keep &= ~(((codes == query) & (query >= 0)).sum(axis=1) / n_query < 0.2)
    if MSA_weight_file_name is not None and seq_name_to_weight is None:
        assert os.path.exists(MSA_weight_file_name), "Weights file not located on disk."
        seq_name_to_weight = MSA_processing(MSA_location=MSA_data_file, use_weights=True,
                                            weights_location=MSA_weight_file_name).seq_name_to_weight
    if seq_name_to_weight is not None:
        keep &= np.array([n in seq_name_to_weight for n in names], dtype=bool)
        weights = np.array([seq_name_to_weight[n] for n, k in zip(names, keep) if k])
    else:
        weights = np.array([1] * int(keep.sum()))
    if retrieval_aggregation_mode not in ("aggregate_substitution", "aggregate_indel"):
        return np.ones((len_target_seq, V)) / V
    codes = codes[keep]
    if codes.shape[1] > width:
        if (codes[:, width:] >= 0).any():
            raise IndexError(f"the alignment has residues beyond column {width} = MSA_end - MSA_start")
        codes = codes[:, :width]
    elif codes.shape[1] < width:                          # an alignment narrower than the declared range: the missing columns hold no counts
        codes = np.concatenate([codes, np.full((codes.shape[0], width - codes.shape[1]), -1, dtype=np.int8)], axis=1)
    # The reference materialises one_hots [sequences, width, V] in float64 (and two temporaries of that size) and reduces over the
    # sequences.  Same expressions here on blocks of sequences: numpy reduces a leading axis by adding the slices one after the other,
    # so carrying the running sum in as slice 0 of the next block gives the same bits with memory bounded by the block (the byte
    # matrix is widened to an index type one block at a time).
    acc = np.zeros((width, V))
    total = np.zeros(width)
    rows_per_block = max(1, block_bytes // max(1, width * V * 8))
    for lo in range(0, codes.shape[0], rows_per_block):
        c = codes[lo:lo + rows_per_block].astype(np.intp)
        one_hots = np.zeros((c.shape[0], width, V))
        i, j = np.nonzero(c >= 0)
        one_hots[i, j, c[i, j]] = 1.0
        counts = (one_hots + np.ones_like(one_hots) * 1e-5) * np.expand_dims(weights[lo:lo + rows_per_block], axis=(1, 2))
        first = lo == 0
        total = np.add.reduce(counts.sum(axis=-1) if first else np.concatenate([total[None], counts.sum(axis=-1)]), axis=0)
        acc = np.add.reduce(counts if first else np.concatenate([acc[None], counts]), axis=0)
    prior = np.zeros((len_target_seq, V))
    prior[MSA_start:MSA_end, :] = acc / np.tile(total.reshape(-1, 1), (1, V))
    return prior


# ---- indel scoring with retrieval: one re-alignment per scored sequence (msa_utils.py:141-192) ------
class SequenceAligner:
    """Aligns one sequence to the family alignment with the user's Clustal Omega executable, the way the reference drives it: the
    alignment's sequences (the first one renamed >REFERENCE_SEQUENCE; at most 100 000 of them, the rest sampled once with python's
    ``random``) upper case with '.' as '-' in ``<alignment folder>/Sampled/Sampled_<id>_<name>``, the sequence as >SEQ_TO_SCORE in a file
    next to it, ``<executable> --profile1 <sampled> --profile2 <sequence> -o <expanded> --force`` (Biopython's ClustalOmegaCommandline
    line), and the rows >SEQ_TO_SCORE / >REFERENCE_SEQUENCE of the output read back upper case.  One subprocess per sequence, as in the
    reference: this path is bound by the aligner, not by the GPU."""

    def __init__(self, MSA_filename: str, clustal_omega_location: str, max_sequences: int = 100000):
        import uuid
        if not clustal_omega_location:
            raise ValueError("indel scoring with retrieval re-aligns every sequence: --clustal_omega_location <executable> is required")
        self.executable = clustal_omega_location
        folder = os.path.join(os.path.dirname(MSA_filename) or ".", "Sampled")
        self._own_folder = None                           # a temporary folder made here is removed with the files (close())
        try:                                              # the reference's place; a read-only alignment folder gets a temporary one
            os.makedirs(folder, exist_ok=True)
            if not os.access(folder, os.W_OK):
                raise PermissionError(folder)
        except OSError:
            import tempfile
            folder = self._own_folder = tempfile.mkdtemp(prefix="pgmi_sampled_")
        name, tag = os.path.basename(MSA_filename), str(uuid.uuid4())
        self.sampled, self.query, self.expanded = (os.path.join(folder, f"{kind}_{tag}_{name}") for kind in ("Sampled", "Seq_to_align", "Expanded"))
        records = process_msa_data(MSA_filename)
        names = list(records)
        if len(names) > max_sequences:
            import random
            names = names[:1] + random.sample(names[1:], k=max_sequences - 1)
        with open(self.sampled, "w") as f:
            for k, n in enumerate(names):
                f.write((">REFERENCE_SEQUENCE" if k == 0 else n) + "\n" + _wrap(records[n].replace(".", "-"), 80) + "\n")

    def close(self):
        """Removes this aligner's three files (the reference leaves one set per model behind; a sharded job builds one aligner per
        rank and assay, so they are taken away with the retrieval state they belong to)."""
        for path in (self.sampled, self.query, self.expanded):
            try:
                os.remove(path)
            except OSError:
                pass
        if getattr(self, "_own_folder", None):
            try:
                os.rmdir(self._own_folder)
            except OSError:
                pass
            self._own_folder = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __call__(self, sequence: str):
        import subprocess
        with open(self.query, "w") as f:
            f.write(">SEQ_TO_SCORE\n" + _wrap(sequence, 80) + "\n")
        done = subprocess.run([self.executable, "--profile1", self.sampled, "--profile2", self.query, "-o", self.expanded, "--force"],
                              capture_output=True, text=True)
        if done.returncode != 0:
            raise RuntimeError(f"{self.executable} failed ({done.returncode}): {done.stderr.strip()[-500:]}")
        rows = process_msa_data(self.expanded)
        return rows[">SEQ_TO_SCORE"], rows[">REFERENCE_SEQUENCE"]


def _wrap(text: str, width: int) -> str:
    return "\n".join(text[k:k + width] for k in range(0, len(text), width))


def realigned_prior_rows(n_rows: int, aligned_seq: str, aligned_ref: str):
    """Which row of the family log-prior stands at each position of ONE re-aligned sequence (msa_utils.py:174-191): an int array of prior
    row indices with -1 for an inserted residue (the reference puts an all-zero row there), or None where the reference's own bookkeeping
    breaks.  The reference walks the alignment columns: both rows gaps -> skipped; a gap in the sequence -> that prior row goes; a gap in
    the reference row -> a zero row is put in AT THE COLUMN'S INDEX in the prior as edited so far (so every 'both gaps' column before it
    shifts it: kept as is); at the end the keep / drop marks must be exactly as many as the rows, else the reference logs an error, keeps
    the half-edited prior and fails in the forward -- None here, and the caller raises."""
    a = np.frombuffer(aligned_seq.encode("ascii"), dtype=np.uint8) == ord("-")
    b = np.frombuffer(aligned_ref.encode("ascii"), dtype=np.uint8) == ord("-")
    if a.shape != b.shape:
        return None
    rows = list(range(n_rows))
    for col in np.flatnonzero(b & ~a):                   # inserted residues, left to right (list.insert clamps like the reference's slices)
        rows.insert(int(col), -1)
    marks = ~a[~(a & b)]                                  # one mark per column that is not 'both gaps': keep unless the sequence has a gap
    if marks.size != len(rows):
        return None
    return np.asarray(rows, dtype=np.int64)[marks]


# ---- checkpoint ------------------------------------------------------------------------------------
def expected_keys(n_layer):
    keys = ["transformer.wte.weight"]
    for i in range(n_layer):
        p = f"transformer.h.{i}."
        keys += [p + "ln_1.weight", p + "ln_1.bias", p + "attn.c_attn.weight", p + "attn.c_attn.bias"]
        for which in ("query", "key", "value"):
            for ki in range(3):
                keys += [p + f"attn.{which}_depthwiseconv.{ki}.conv.weight", p + f"attn.{which}_depthwiseconv.{ki}.conv.bias"]
        keys += [p + "attn.c_proj.weight", p + "attn.c_proj.bias", p + "ln_2.weight", p + "ln_2.bias",
                 p + "mlp.c_fc.weight", p + "mlp.c_fc.bias", p + "mlp.c_proj.weight", p + "mlp.c_proj.bias"]
    keys += ["transformer.ln_f.weight", "transformer.ln_f.bias", "lm_head.weight"]
    return keys


def load_checkpoint(checkpoint_dir: str):
    """config.json + pytorch_model.bin / model.safetensors (score_tranception_proteingym.py:79,100).
    Returns (cfg dict, flat fp32 blob in the ABI order of include/pgmi.h)."""
    c = json.load(open(os.path.join(checkpoint_dir, "config.json")))
    bin_path = os.path.join(checkpoint_dir, "pytorch_model.bin")
    if os.path.exists(bin_path):
        import torch
        sd = {k: v.float().numpy() for k, v in torch.load(bin_path, map_location="cpu").items()}
    else:
        from safetensors.numpy import load_file
        sd = load_file(os.path.join(checkpoint_dir, "model.safetensors"))
    if "lm_head.weight" not in sd:                        # tied to wte (_keys_to_ignore_on_load_missing, :635)
        sd["lm_head.weight"] = sd["transformer.wte.weight"]
    n_embd, n_head, n_layer = int(c["n_embd"]), int(c["n_head"]), int(c["n_layer"])
    cfg = dict(arch=_lib.ARCH_TRANCEPTION, layers=n_layer, embed_dim=n_embd, heads=n_head,
               ffn_dim=int(c["n_inner"]) if c.get("n_inner") else 4 * n_embd, vocab=int(c.get("vocab_size", 25)),
               max_positions=int(c.get("n_ctx", c.get("n_positions", 1024))), ln_eps=float(c.get("layer_norm_epsilon", 1e-5)))
    if c.get("activation_function", "squared_relu") != "squared_relu":
        raise ValueError("only the squared_relu activation of the released Tranception checkpoints is supported")
    keys = expected_keys(n_layer)
    missing = [k for k in keys if k not in sd]
    if missing:
        raise RuntimeError(f"Missing key(s) in Tranception state_dict: {missing[:8]}...")
    parts = []
    for k in keys:
        a = np.asarray(sd[k], dtype=np.float32)
        if k.endswith("conv.weight"):
            a = a.reshape(a.shape[0], a.shape[-1])        # [dh, 1, k] -> [dh, k]
        parts.append(np.ascontiguousarray(a).ravel())
    return cfg, np.concatenate(parts)


class TranceptionModel:
    """Device-resident Tranception.  ``score_mutants`` mirrors the reference method of the same name."""

    share_prefix = True          # forward a mutated sequence from its first mutated token on (sequence_loglik)
    share_intermediate = True    # ... and multi-mutants from their SECOND mutated token on where an intermediate root pays (intermediate_roots)
    rows_forwarded = 0           # token rows that went through the network ...
    rows_full = 0                # ... and the rows the reference's loop forwards for the same calls

    def __init__(self, cfg: dict, weights: np.ndarray, device: int = 0, scoring_window: str = "optimal",
                 retrieval: Optional[dict] = None, max_rows: int = 0):
        lib = _lib.load()
        self.cfg = dict(cfg)
        c = Config(abi_version=_lib.ABI_VERSION, arch=_lib.ARCH_TRANCEPTION, layers=cfg["layers"], embed_dim=cfg["embed_dim"],
                   heads=cfg["heads"], ffn_dim=cfg["ffn_dim"], vocab=cfg["vocab"], max_positions=cfg["max_positions"],
                   token_dropout=0, emb_layer_norm_before=0, precision=_lib.PREC_F16X3, max_rows=max_rows,
                   ln_eps=cfg.get("ln_eps", 1e-5))
        w = _lib.as_f32(weights)
        n = lib.pgmi_weight_count(C.byref(c))
        if w.size != n:
            raise PgmiError(f"weight blob has {w.size} elements, config needs {n}")
        h = C.c_void_p()
        _lib.check(lib.pgmi_model_create(C.byref(c), _lib.ptr(w, _lib._f32p), w.size, device, C.byref(h)))
        self._h = h
        self.n_ctx = cfg["max_positions"]
        self.scoring_window = scoring_window
        # retrieval: dict(log_prior [L,25] float32, MSA_start (0-based), MSA_end, weight)
        self.retrieval = retrieval
        self.share_prefix = os.environ.get("PGMI_TR_SHARE_PREFIX", "1") != "0"

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

    # -- tokenisation (model_pytorch.py:930-939) -------------------------------------------------------
    def encode_batch(self, sequences: Sequence[str]):
        out = []
        for s in sequences:
            for ch, choices in (("X", AA_vocab), ("B", "DN"), ("J", "IL"), ("Z", "EQ")):
                s = sequence_replace_single(s, ch, choices)
            out.append(([CLS] + [VOCAB.get(ch, UNK) for ch in s] + [SEP])[: self.n_ctx])
        T = max(len(e) for e in out)
        ids = np.full((len(out), T), PAD, dtype=np.int32)
        lens = np.zeros(len(out), dtype=np.int32)
        for i, e in enumerate(out):
            ids[i, :len(e)] = e
            lens[i] = len(e)
        return ids, lens

    def token_logprobs(self, input_ids) -> np.ndarray:
        t = _lib.as_i32(np.asarray(input_ids))
        B, T = t.shape
        out = np.empty((B, T, self.cfg["vocab"]), dtype=np.float32)
        _lib.check(_lib.load().pgmi_tr_token_logprobs(self._h, _lib.ptr(t, _lib._i32p), B, T, _lib.ptr(out, _lib._f32p)))
        return out

    def sequence_loglik(self, sliced_sequences, window_start=None, window_end=None, reverse=False, mutated_sequences=None,
                        reference=None) -> np.ndarray:
        """sum_t log p(token_{t+1} | tokens_{<=t}) per sliced sequence (scoring_utils.py:97-128), fused
        with the retrieval prior (model_pytorch.py:806-830) when the model was built with one.  ``mutated_sequences`` (the full
        sequences the slices were cut from) are what indel scoring with retrieval re-aligns (model_pytorch.py:794-799).

        ``reference`` (additive; int array, one entry per sequence): the position in ``sliced_sequences`` of the sequence whose
        PREFIX this one shares -- the wild type cut to the same window -- or its own position.  The model is causal, so the rows of a
        mutated sequence before its first mutated token are the wild type's: with ``reference`` only the rows from there on are
        forwarded (``pgmi_tr_sequence_loglik_shared``; the reference forwards every sequence in full, scoring_utils.py:77-150).  The
        values are bit-identical to the unshared call; ``self.share_prefix = False`` (or PGMI_TR_SHARE_PREFIX=0) ignores it."""
        lib = _lib.load()
        seqs = list(sliced_sequences)
        n = len(seqs)
        if self.retrieval is not None and self.retrieval.get("aligner") is not None:
            return np.array([self._realigned_loglik(seqs[i], int(window_start[i]), int(window_end[i]), reverse, mutated_sequences[i])
                             for i in range(n)], dtype=np.float32)
        out = np.empty(n, dtype=np.float32)
        lengths = np.array([len(s) for s in seqs], dtype=np.int64)
        order = np.argsort(lengths, kind="stable")
        r = self.retrieval
        share = reference is not None and getattr(self, "share_prefix", True)
        if share:
            reference = np.asarray(reference, dtype=np.int64)
        # groups of equal length -> no padding waste; every group goes through one ABI call
        start = 0
        while start < n:
            L = len(seqs[order[start]])
            end = start
            while end < n and len(seqs[order[end]]) == L:
                end += 1
            idx = order[start:end]
            ids, lens = self.encode_batch([seqs[i] for i in idx])
            B, T = ids.shape
            res = np.empty(B, dtype=np.float32)
            prior = (None, 0, None, None, None, None, 0.0)
            if r is not None:
                a0 = np.zeros(B, np.int32); row0 = np.zeros(B, np.int32); nn = np.zeros(B, np.int32)
                flip = np.full(B, 1 if reverse else 0, np.int32)
                for j, i in enumerate(idx):
                    st, en = int(window_start[i]), int(window_end[i])
                    lo, hi = max(st, r["MSA_start"]), min(en, r["MSA_end"])
                    if hi <= lo:
                        print("Non overlapping region detected: min_prior_slice {} and max_prior_slice {}".format(lo, hi))
                        continue
                    a0[j] = max(0, en - r["MSA_end"]) if reverse else max(0, r["MSA_start"] - st)
                    row0[j] = lo
                    nn[j] = hi - lo
                lp = _lib.as_f32(r["log_prior"])
                prior = (_lib.ptr(lp, _lib._f32p), lp.shape[0], _lib.ptr(a0, _lib._i32p), _lib.ptr(row0, _lib._i32p),
                         _lib.ptr(nn, _lib._i32p), _lib.ptr(flip, _lib._i32p), float(r["weight"]))
            ref_local = self._local_references(idx, reference, lengths, ids) if share else None
            if ref_local is not None:
                # multi-mutants: a sequence shares MORE with "the wild type + its first mutation" than with the wild type; where that pays, such
                # intermediate sequences are added to the call as roots of their own (forwarded in full, their scores dropped)
                ids_c, ref_c, extra = self.intermediate_roots(ids, ref_local) if getattr(self, "share_intermediate", True) else (ids, ref_local, 0)
                Bc = B + extra
                res_c = np.empty(Bc, dtype=np.float32)
                prior_c = prior
                if r is not None and extra:                       # an added root is scored like the wild-type row it was made from (same window)
                    src = np.concatenate([np.arange(B), ref_c[B:]])
                    a0c, r0c, nnc, flc = (np.ascontiguousarray(v[src]) for v in (a0, row0, nn, flip))
                    prior_c = (prior[0], prior[1], _lib.ptr(a0c, _lib._i32p), _lib.ptr(r0c, _lib._i32p), _lib.ptr(nnc, _lib._i32p),
                               _lib.ptr(flc, _lib._i32p), prior[6])
                if extra:
                    ref_c = ref_c.copy()
                    ref_c[B:] = np.arange(B, Bc)                  # (ref_c[B:] held the wild-type row each new root was made from)
                rows = np.zeros(1, dtype=np.int64)
                _lib.check(lib.pgmi_tr_sequence_loglik_shared(self._h, _lib.ptr(ids_c, _lib._i32p), _lib.ptr(ref_c, _lib._i32p), Bc, T,
                                                              *prior_c, _lib.ptr(res_c, _lib._f32p), None, _lib.ptr(rows, _lib._i64p)))
                res = res_c[:B]
                self.rows_forwarded += int(rows[0])
            else:
                _lib.check(lib.pgmi_tr_sequence_loglik(self._h, _lib.ptr(ids, _lib._i32p), _lib.ptr(lens, _lib._i32p), B, T,
                                                       *prior, _lib.ptr(res, _lib._f32p)))
                self.rows_forwarded += B * T
            self.rows_full += B * T
            out[idx] = res
            start = end
        return out

    @staticmethod
    def _local_references(idx, reference, lengths, ids):
        """``reference`` (positions in the caller's list) as positions inside one equal-length group, int32 [B]; a sequence whose
        reference is not in the group (another length), that is its own reference's reference, or that shares no token with it
        stands for itself.  None when nothing is shared (every sequence a root: the plain call does the same work)."""
        where = {int(i): j for j, i in enumerate(idx)}
        local = np.arange(len(idx), dtype=np.int32)
        for j, i in enumerate(idx):
            k = where.get(int(reference[i]), j)
            if k != j and int(reference[idx[k]]) == int(idx[k]) and ids[j, 0] == ids[k, 0]:
                local[j] = k
        return local if (local != np.arange(len(idx))).any() else None

    @staticmethod
    def intermediate_roots(ids, ref_local):
        """Roots between the wild type and the multi-mutants of one equal-length group (additive; the reference has nothing like it).
        A sequence that differs from its wild-type row at tokens f < g < ... is forwarded from f on when it shares the wild type's prefix,
        but only from g on when it shares the prefix of "wild type + the substitution at f".  For every such (wild-type row, f, token) the
        members are counted: if the rows they save, sum (g - f), exceed what forwarding the intermediate sequence in full costs -- T rows for
        a new sequence, f rows when it is itself a row of the group (it then stops sharing with the wild type) -- it becomes a root.
        Returns (ids with the new root rows appended, reference positions, number of rows appended); for an appended root the reference entry
        holds the wild-type row it was made from (the caller copies that row's retrieval arguments, then makes the root its own reference).
        A sequence's bits do not depend on which root serves its prefix (any root holds the same tokens there)."""
        B, T = ids.shape
        ref = np.asarray(ref_local, dtype=np.int64)
        member = ref != np.arange(B)
        if not member.any():
            return ids, ref_local, 0
        diff = ids != ids[ref]
        count = diff.sum(axis=1)
        multi = np.flatnonzero(member & (count >= 2))
        if not multi.size:
            return ids, ref_local, 0
        first = diff[multi].argmax(axis=1)
        rest = diff[multi].copy()
        rest[np.arange(multi.size), first] = False
        second = rest.argmax(axis=1)
        token = ids[multi, first]
        key = (ref[multi] * T + first) * 64 + token                    # (wild-type row, position, token): vocabulary of 25
        uniq, inverse = np.unique(key, return_inverse=True)
        saved = np.bincount(inverse, weights=(second - first).astype(np.float64), minlength=uniq.size)
        # is the intermediate sequence already a row of the group?  (a single mutant listed in the assay)
        single = np.flatnonzero(member & (count == 1))
        where = {}
        if single.size:
            f1 = diff[single].argmax(axis=1)
            for row, k in zip(single, (ref[single] * T + f1) * 64 + ids[single, f1]):
                where.setdefault(int(k), int(row))
        out_ref = ref.copy()
        new_rows, new_src = [], []
        root_of_key = np.full(uniq.size, -1, dtype=np.int64)           # per paying key its root row; the members are re-pointed in ONE pass below
        for u, k in enumerate(uniq):                                    # (thousands of keys at most: no per-key scan of the multi-mutants)
            k = int(k)
            f = (k // 64) % T
            cost = f if k in where else T
            if saved[u] <= cost:
                continue
            if k in where:
                root = where[k]
                out_ref[root] = root                                    # forwarded in full from now on
            else:
                root = B + len(new_rows)
                wt_row = k // 64 // T
                seq = ids[wt_row].copy()
                seq[f] = k % 64
                new_rows.append(seq)
                new_src.append(wt_row)
            root_of_key[u] = root
        pays = root_of_key[inverse] >= 0
        out_ref[multi[pays]] = root_of_key[inverse[pays]]
        if not new_rows and (out_ref == ref).all():
            return ids, ref_local, 0
        ids_c = np.ascontiguousarray(np.concatenate([ids, np.stack(new_rows)]) if new_rows else ids, dtype=np.int32)
        ref_c = np.ascontiguousarray(np.concatenate([out_ref, np.asarray(new_src, dtype=np.int64)]), dtype=np.int32)
        return ids_c, ref_c, len(new_rows)

    def _realigned_loglik(self, sliced, start, end, reverse, mutated_sequence) -> float:
        """Indel scoring with retrieval, one sequence (model_pytorch.py:794-839): the family log-prior is re-indexed through the
        sequence's own alignment (a row dropped per deleted residue, none for an inserted one), fused over the scored window, and the
        positions of inserted residues keep the network's log-probabilities.  The device reduction fuses ONE contiguous run of
        positions per sequence, and the log-likelihood is a sum over positions, so the sequence goes through it once per run of
        positions that have a prior row plus once without any prior:  sum_runs S(run) - (runs - 1) S(no prior)."""
        r = self.retrieval
        rows = realigned_prior_rows(r["log_prior"].shape[0], *r["aligner"](mutated_sequence))
        if rows is None:
            raise IndexError("indel scoring with retrieval: the alignment's reference sequence does not span the protein the log-prior "
                             "was built for (the reference fails on the same input, model_pytorch.py:836)")
        m_start, m_end = r["MSA_start"], r["MSA_start"] + len(rows)
        lo, hi = max(start, m_start), min(end, m_end)
        if hi <= lo:
            raise IndexError(f"indel scoring with retrieval: no overlap between the scored window [{start}, {end}) and the alignment")
        window = rows[lo:hi][::-1] if reverse else rows[lo:hi]           # prior row per fused position, in scoring order
        if len(window) != len(sliced):                                   # the reference's mask (one entry per prior row + the end token) against the scored positions
            raise IndexError(f"indel scoring with retrieval: the prior covers {len(window)} of the window's {len(sliced)} residues "
                             "(the reference raises here: model_pytorch.py:836)")
        first = max(0, end - m_end) if reverse else max(0, m_start - start)
        has_prior = window >= 0
        edges = np.flatnonzero(np.diff(np.concatenate([[False], has_prior, [False]]).astype(np.int8)))
        runs = list(zip(edges[0::2], edges[1::2]))                       # [i0, i1) runs of positions with a prior row
        prior = np.zeros((len(window) + 1, r["log_prior"].shape[1]), dtype=np.float32)
        prior[:len(window)][has_prior] = r["log_prior"][window[has_prior]]   # this sequence's prior, already in scoring order: never flipped below
        ids, lens = self.encode_batch([sliced])                          # encoded ONCE: X / B / J / Z are replaced at random
        ids, lens = np.repeat(ids, len(runs) + 1, axis=0), np.repeat(lens, len(runs) + 1)
        B, T = ids.shape
        a0 = np.array([first + i0 for i0, _ in runs] + [0], dtype=np.int32)
        row0 = np.array([i0 for i0, _ in runs] + [0], dtype=np.int32)
        count = np.array([i1 - i0 for i0, i1 in runs] + [0], dtype=np.int32)
        flip = np.zeros(B, dtype=np.int32)
        res = np.empty(B, dtype=np.float32)
        lp = _lib.as_f32(prior)
        _lib.check(_lib.load().pgmi_tr_sequence_loglik(self._h, _lib.ptr(ids, _lib._i32p), _lib.ptr(lens, _lib._i32p), B, T,
                                                       _lib.ptr(lp, _lib._f32p), lp.shape[0], _lib.ptr(a0, _lib._i32p), _lib.ptr(row0, _lib._i32p),
                                                       _lib.ptr(count, _lib._i32p), _lib.ptr(flip, _lib._i32p), float(r["weight"]),
                                                       _lib.ptr(res, _lib._f32p)))
        plain = float(res[-1])
        return float(np.sum(res[:-1], dtype=np.float64) - (len(runs) - 1) * plain) if runs else plain

    # -- scoring (scoring_utils.py:77-150) ----------------------------------------------------------------
    def _scores(self, slices, column, target_seq, reverse=False):
        """Per-sequence score of one reading direction (scoring_utils.py:129-150): window log-likelihoods from the
        device, summed per sequence in 'sliding' mode, divided by the FULL sequence length, then (with a target)
        the wild type's value for the same window start (optimal) / the single wild-type total (sliding) subtracted."""
        realign = dict(mutated_sequences=list(slices['mutated_sequence'])) if (self.retrieval or {}).get("aligner") else {}
        loglik = self.sequence_loglik(slices['sliced_mutated_sequence'], slices['window_start'].to_numpy(),
                                      slices['window_end'].to_numpy(), reverse=reverse,
                                      reference=wild_type_rows(slices, target_seq), **realign)
        per = pd.DataFrame({'mutated_sequence': list(slices['mutated_sequence']),
                            'sliced_mutated_sequence': list(slices['sliced_mutated_sequence']),
                            'window_start': list(slices['window_start']), 'window_end': list(slices['window_end']),
                            'score': loglik.astype(np.float32)})
        sliding = self.scoring_window == "sliding"
        if sliding:
            per = per[['mutated_sequence', 'score']].groupby('mutated_sequence').sum().reset_index()
        per['score'] = per['score'] / per['mutated_sequence'].map(len)
        if target_seq is None:
            per[column] = per['score']
            return per[['mutated_sequence', column]]
        is_wt = per.mutated_sequence == target_seq
        variants, wild = per[~is_wt], per[is_wt]
        if sliding:
            out = variants.copy()
            out[column] = out['score'] - list(wild['score'])[0]
        else:
            out = pd.merge(variants, wild, how='left', on=['window_start'], suffixes=('', '_wt'))
            out[column] = out['score'] - out['score_wt']
        return out[['mutated_sequence', column]]

    def score_mutants(self, DMS_data, target_seq=None, scoring_mirror=True, batch_size_inference=10, num_workers=10,
                      indel_mode=False, append_wildtype_row=True):
        """Same contract as the reference method (model_pytorch.py:878-928): returns mutated_sequence,
        avg_score_L_to_R, [avg_score_R_to_L,] avg_score; a zero row for the wild type when it is among the inputs
        (under column 'mutant' in indel mode, as the reference writes it).  batch_size_inference / num_workers are
        accepted and ignored: batching happens on the device side.  ``append_wildtype_row=False`` (additive; used by
        run_sharded when an assay's rows are scored in chunks) leaves that row to the caller."""
        frame = DMS_data.copy()
        if 'mutated_sequence' not in frame and not indel_mode:
            frame['mutated_sequence'] = mutated_sequences(target_seq, frame['mutant'])
        assert ('mutated_sequence' in frame), "DMS file to score does not have mutated_sequence column"
        if 'mutant' not in frame:
            frame['mutant'] = frame['mutated_sequence']
        frame = frame[['mutated_sequence', 'mutant']]
        context = self.n_ctx - 2                                         # [CLS] and [SEP]
        if target_seq is not None:
            slices = get_sequence_slices(frame, target_seq=target_seq, model_context_len=context, indel_mode=indel_mode,
                                         scoring_window=self.scoring_window)
        else:                                                            # no reference: raw log-likelihoods, sliding windows
            slices = get_sequence_slices(frame, target_seq=list(frame['mutated_sequence'])[0], model_context_len=context,
                                         indel_mode=indel_mode, scoring_window='sliding')
        print("Scoring sequences from left to right")
        result = self._scores(slices, 'avg_score_L_to_R', target_seq)
        if scoring_mirror:
            print("Scoring sequences from right to left")
            mirrored = slices.copy()
            mirrored['sliced_mutated_sequence'] = [x[::-1] for x in mirrored['sliced_mutated_sequence']]
            backward = self._scores(mirrored, 'avg_score_R_to_L', target_seq, reverse=True)
            result = pd.merge(result, backward, on='mutated_sequence', how='left', suffixes=('', '_R_to_L'))
            result['avg_score'] = (result['avg_score_L_to_R'] + result['avg_score_R_to_L']) / 2.0
        else:
            result['avg_score'] = result['avg_score_L_to_R']
        key = "mutant" if indel_mode else "mutated_sequence"
        if append_wildtype_row and target_seq in DMS_data[key].values:   # the scorer drops the wild type: add it back, score 0
            names = [key, 'avg_score_L_to_R'] + (['avg_score_R_to_L'] if scoring_mirror else []) + ['avg_score']
            result = pd.concat([result, pd.DataFrame([[target_seq] + [0] * (len(names) - 1)], columns=names)], ignore_index=True)
        return result


def from_pretrained(checkpoint_dir: str, device: int = 0, scoring_window: str = "optimal", retrieval: Optional[dict] = None,
                    max_rows: int = 0) -> TranceptionModel:
    """Mirror of ``TranceptionLMHeadModel.from_pretrained(checkpoint, config=config)``
    (score_tranception_proteingym.py:100).  ``retrieval`` = dict(MSA_filename, MSA_start (0-based),
    MSA_end, full_protein_length, retrieval_inference_weight, MSA_weight_file_name=None,
    seq_name_to_weight=None) builds the log-prior exactly as model_pytorch.py:662-672 does."""
    cfg, blob = load_checkpoint(checkpoint_dir)
    return TranceptionModel(cfg, blob, device=device, scoring_window=scoring_window, retrieval=build_retrieval(retrieval),
                            max_rows=max_rows)


def build_retrieval(retrieval: Optional[dict]) -> Optional[dict]:
    """The per-assay retrieval state of a model (``TranceptionModel.retrieval``): the log-prior exactly as
    model_pytorch.py:662-672 builds it.  A resident model scores many assays by swapping this dict
    (run_sharded): the network weights do not depend on the assay."""
    if not retrieval:
        return None
    prior = get_msa_prior(retrieval["MSA_filename"], retrieval.get("MSA_weight_file_name"), retrieval["MSA_start"],
                          retrieval["MSA_end"], retrieval["full_protein_length"],
                          retrieval_aggregation_mode=retrieval.get("retrieval_aggregation_mode", "aggregate_substitution"),
                          seq_name_to_weight=retrieval.get("seq_name_to_weight"))
    import torch
    log_prior = torch.log(torch.tensor(prior).float()).numpy()              # same rounding as the reference (:662-672)
    state = dict(log_prior=log_prior, MSA_start=int(retrieval["MSA_start"]), MSA_end=int(retrieval["MSA_end"]),
                 weight=float(retrieval.get("retrieval_inference_weight", 0.6)))
    if retrieval.get("retrieval_aggregation_mode") == "aggregate_indel":    # every scored sequence is re-aligned (model_pytorch.py:794-799)
        state["aligner"] = SequenceAligner(retrieval["MSA_filename"], retrieval.get("clustal_omega_location"))
    return state
