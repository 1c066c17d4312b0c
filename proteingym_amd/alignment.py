"""a2m / FASTA alignments as byte matrices: the EVE-style pre-processing both retrieval paths start from
(Tranception's ``tranception/utils/msa_utils.py:230-340`` and the MSA Transformer's ``proteingym/utils/msa_utils.py:60-170``
do the same steps on python strings and pandas frames, one character at a time; here every step is one numpy expression over
the [sequences, columns] uint8 matrix).  Host logic only -- the O(N^2 L) weights run on the GPU (``weights.py``).

Steps, in the reference's order:
  1. read: a header line (kept verbatim, '>' included) names a record, the following lines are its sequence; the first line of the
     file names the focus sequence;
  2. (``preprocess``) '.' -> '-', upper case; keep the columns where the focus sequence has a residue; drop sequences with more than
     ``max_seq_gaps`` gaps; columns with more than ``max_col_gaps`` gaps among the survivors are written lower case;
  3. focus columns = columns where the focus sequence is neither lower case nor '-';
  4. every sequence cut to the focus columns, upper case, '.' -> '-';
  5. (``drop_indeterminate``) sequences with a symbol outside the 20 amino acids and '-' in a focus column are dropped.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np

AMINO_ACIDS = "ACDEFGHIKLMNPQRSTVWY"
_GAP, _DOT = ord("-"), ord(".")


def read_records(path: str, upper: bool = False) -> Tuple[str, Dict[str, str]]:
    """(name of the first record, {header line -> sequence}) of an a2m / FASTA file.  Lines under a repeated header extend that
    record, lines before the first header go to the record '' -- as in the reference's readers (msa_utils.py:28-43, :252-262)."""
    chunks: Dict[str, List[str]] = {}
    first, name = None, ""
    with open(path, "r") as f:
        for lineno, line in enumerate(f):
            line = line.rstrip()
            if line.startswith(">"):
                name = line
                if lineno == 0:
                    first = name
            else:
                chunks.setdefault(name, []).append(line.upper() if upper else line)
    return first, {n: "".join(c) for n, c in chunks.items()}


def to_matrix(sequences: List[str], columns: np.ndarray = None) -> np.ndarray:
    """uint8 [n, width] of equal-length ASCII sequences; with ``columns`` (bool [width]) only those columns, picked row by row so that
    the full-width matrix of a large alignment (10^5 - 10^6 sequences x thousands of raw columns) never exists."""
    if not sequences:
        return np.zeros((0, 0), dtype=np.uint8)
    width = len(sequences[0])
    if any(len(s) != width for s in sequences):
        raise ValueError("alignment rows differ in length")
    out = np.empty((len(sequences), width if columns is None else int(columns.sum())), dtype=np.uint8)
    for i, s in enumerate(sequences):
        row = np.frombuffer(s.encode("ascii"), dtype=np.uint8)
        out[i] = row if columns is None else row[columns]
    return out


def to_strings(matrix: np.ndarray) -> List[str]:
    width = matrix.shape[1]
    flat = np.ascontiguousarray(matrix).tobytes().decode("ascii")
    return [flat[i * width:(i + 1) * width] for i in range(matrix.shape[0])]


def _is_lower(m):
    return (m >= ord("a")) & (m <= ord("z"))


def _upper(m):
    """Upper case, in place for an array this module owns."""
    m[_is_lower(m)] -= 32
    return m


def _lower_where(m, columns):
    """Lower case in the columns flagged (bool [width]), in place."""
    sub = m[:, columns]
    sub[(sub >= ord("A")) & (sub <= ord("Z"))] += 32
    m[:, columns] = sub
    return m


class FocusAlignment:
    """The result of steps 1-5 above.

    names        header lines of the sequences that survive, file order
    raw          {name -> sequence after step 2}  (what the MSA Transformer is fed)
    focus_name, focus_seq, focus_cols (int array)
    trimmed      uint8 [len(names), len(focus_cols)]: upper-case residues and '-' over the focus columns (step 4, after step 5)
    """

    def __init__(self, path: str, preprocess: bool = True, max_seq_gaps: float = 0.5, max_col_gaps: float = 1.0,
                 drop_indeterminate: bool = True):
        self.focus_name, records = read_records(path)
        if self.focus_name is None:
            raise ValueError(f"{path}: the first line is not a '>' header")
        names = list(records)
        rows = [records[n] for n in names]
        if preprocess:
            focus = _upper(np.frombuffer(records[self.focus_name].encode("ascii"), dtype=np.uint8).copy())
            m = _upper(to_matrix(rows, columns=(focus != _GAP) & (focus != _DOT)))
            m[m == _DOT] = _GAP
            gaps = m == _GAP
            seq_ok = gaps.mean(axis=1) <= max_seq_gaps
            col_ok = gaps[seq_ok].mean(axis=0) <= max_col_gaps
            m = _lower_where(m[seq_ok], ~col_ok)
            names = [n for n, ok in zip(names, seq_ok) if ok]
        else:
            m = to_matrix(rows)
        self.raw = dict(zip(names, to_strings(m)))
        self.focus_seq = self.raw[self.focus_name]
        f = m[names.index(self.focus_name)]
        self.focus_cols = np.flatnonzero(~_is_lower(f) & (f != _GAP))
        t = _upper(m[:, self.focus_cols])                     # fancy indexing: a copy
        t[t == _DOT] = _GAP
        if drop_indeterminate:
            known = np.zeros(256, dtype=bool)
            known[np.frombuffer((AMINO_ACIDS + "-").encode("ascii"), dtype=np.uint8)] = True
            ok = known[t].all(axis=1)
            t = t[ok]
            names = [n for n, k in zip(names, ok) if k]
        self.names = names
        self.trimmed = t

    def residue_codes(self, gap: int = -1) -> np.ndarray:
        """int8 [n, focus columns]: index in ``AMINO_ACIDS``; ``gap`` for '-' and for anything else."""
        table = np.full(256, gap, dtype=np.int8)
        table[np.frombuffer(AMINO_ACIDS.encode("ascii"), dtype=np.uint8)] = np.arange(len(AMINO_ACIDS), dtype=np.int8)
        return table[self.trimmed]

    def focus_range(self) -> Tuple[int, int]:
        """(start, stop) residue numbers from a 'name/start-stop' header; (1, length) when the header has none."""
        try:
            start, stop = self.focus_name.split("/")[-1].split("-")
            return int(start), int(stop)
        except ValueError:
            return 1, len(self.focus_seq)
