#!/usr/bin/env python3
"""A stand-in for the Clustal Omega executable, for tests only: the command line the reference builds for scoring indels with retrieval
(``clustalo --profile1 <alignment> --profile2 <one sequence> -o <out> --force``, tranception/utils/msa_utils.py:167-172) is accepted and
answered with a deterministic profile-to-sequence alignment.  It is NOT Clustal Omega (no guide tree, no HMM): it exists so that the code
AROUND the aligner -- the files written for it, the walk over the two aligned rows, the edited log-prior, the fusion rule for inserted
positions -- can be run through the unmodified reference and through this repository with the same aligner answers.

Alignment: Needleman-Wunsch (match +2, mismatch -1, gap -2; ties: diagonal, then a gap in the new sequence, then a gap in the profile) of the
new sequence against the residues of the profile's FIRST row; profile columns where that row has a gap stay gap columns for the new sequence;
residues of the new sequence aligned to nothing open new all-gap columns in the profile.  Output: FASTA, the profile's rows then the new
sequence, 60 characters per line.
"""
import sys


def read_fasta(path):
    names, seqs = [], []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            if line.startswith(">"):
                names.append(line)
                seqs.append([])
            else:
                seqs[-1].append(line)
    return names, ["".join(s) for s in seqs]


def needleman_wunsch(a, b, match=2, mismatch=-1, gap=-2):
    """Global alignment of strings a (profile residues) and b (new sequence): list of (i or None, j or None) pairs."""
    n, m = len(a), len(b)
    score = [[0] * (m + 1) for _ in range(n + 1)]
    for i in range(1, n + 1):
        score[i][0] = i * gap
    for j in range(1, m + 1):
        score[0][j] = j * gap
    for i in range(1, n + 1):
        ai, row, prev = a[i - 1], score[i], score[i - 1]
        for j in range(1, m + 1):
            d = prev[j - 1] + (match if ai == b[j - 1] else mismatch)
            u = prev[j] + gap                      # profile residue against a gap in the new sequence
            l = row[j - 1] + gap                   # new residue against a gap in the profile
            row[j] = d if d >= u and d >= l else (u if u >= l else l)
    pairs, i, j = [], n, m
    while i > 0 or j > 0:
        if i > 0 and j > 0 and score[i][j] == score[i - 1][j - 1] + (match if a[i - 1] == b[j - 1] else mismatch):
            pairs.append((i - 1, j - 1))
            i, j = i - 1, j - 1
        elif i > 0 and score[i][j] == score[i - 1][j] + gap:
            pairs.append((i - 1, None))
            i -= 1
        else:
            pairs.append((None, j - 1))
            j -= 1
    return pairs[::-1]


def main(argv):
    args = {}
    k = 0
    while k < len(argv):
        if argv[k] in ("--profile1", "--profile2", "-o"):
            args[argv[k]] = argv[k + 1]
            k += 2
        elif argv[k] == "--force":
            k += 1
        else:
            sys.exit(f"stand_in_clustalo: unknown argument {argv[k]}")
    names, rows = read_fasta(args["--profile1"])
    new_names, new_rows = read_fasta(args["--profile2"])
    width = len(rows[0])
    if any(len(r) != width for r in rows) or len(new_rows) != 1:
        sys.exit("stand_in_clustalo: profile1 must be an alignment, profile2 one sequence")
    first = rows[0]
    columns = [c for c in range(width) if first[c] != "-"]                 # profile columns that hold a residue of the first row
    new = new_rows[0].replace("-", "")
    pairs = needleman_wunsch("".join(first[c] for c in columns), new)
    out_rows = [[] for _ in rows]
    out_new = []
    col = 0                                                                 # next profile column to emit
    for i, j in pairs:
        target = columns[i] if i is not None else None
        if target is not None:
            while col < target:                                             # gap columns of the first row: kept, gap in the new sequence
                for r, o in zip(rows, out_rows):
                    o.append(r[col])
                out_new.append("-")
                col += 1
            for r, o in zip(rows, out_rows):
                o.append(r[col])
            out_new.append(new[j] if j is not None else "-")
            col += 1
        else:                                                               # a residue of the new sequence aligned to nothing
            for o in out_rows:
                o.append("-")
            out_new.append(new[j])
    while col < width:
        for r, o in zip(rows, out_rows):
            o.append(r[col])
        out_new.append("-")
        col += 1
    with open(args["-o"], "w") as f:
        for name, o in list(zip(names, out_rows)) + [(new_names[0], out_new)]:
            s = "".join(o)
            f.write(name + "\n" + "\n".join(s[p:p + 60] for p in range(0, len(s), 60)) + "\n")


if __name__ == "__main__":
    main(sys.argv[1:])
