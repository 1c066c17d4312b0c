#!/usr/bin/env python3
"""A stand-in for hh-suite's ``hhfilter`` executable, for tests only: takes the command line the reference builds
(``hhfilter -cov C -id I -qid Q -i <a2m> -o <a2m>``, baselines/esm/compute_fitness.py:88) and filters deterministically -- the first
sequence is the query and always stays; a sequence stays if it covers at least C % of the query's residues, is at least Q % identical to
the query over the query's residues, and is at most I % identical to every sequence kept before it.  It is NOT hhfilter: it exists so that
the code AROUND the filter can be run through the unmodified reference and through this repository with the same filter answers."""
import sys


def main(argv):
    opt = dict(zip(argv[0::2], argv[1::2]))
    cov, max_id, min_qid = float(opt.get("-cov", 0)), float(opt.get("-id", 100)), float(opt.get("-qid", 0))
    names, seqs = [], []
    with open(opt["-i"]) as f:
        for line in f:
            line = line.rstrip("\n")
            if line.startswith(">"):
                names.append(line)
                seqs.append("")
            elif names:
                seqs[-1] += line.strip()
    query = seqs[0]
    cols = [k for k, c in enumerate(query) if c != "-"]

    def identity(a, b):
        both = [(a[k], b[k]) for k in cols if k < len(a) and k < len(b) and a[k] != "-" and b[k] != "-"]
        return 100.0 * sum(x == y for x, y in both) / max(len(both), 1)
    kept = [0]
    for i in range(1, len(seqs)):
        s = seqs[i]
        covered = 100.0 * sum(1 for k in cols if k < len(s) and s[k] != "-") / max(len(cols), 1)
        if covered < cov or identity(s, query) < min_qid:
            continue
        if any(identity(s, seqs[j]) > max_id for j in kept):
            continue
        kept.append(i)
    with open(opt["-o"], "w") as f:
        for i in kept:
            f.write(names[i] + "\n" + seqs[i] + "\n")


if __name__ == "__main__":
    main(sys.argv[1:])
