"""CPU side of BASELINE config 5 (pseudo-perplexity over variable-length indel mutants): the oracle against the
reference-generated goldens (tests/golden/make_golden_pppl_indels.py ran the unmodified reference CLI and the reference
model), the product's token packing, and the rank partition."""
import os

import numpy as np
import pandas as pd
import pytest
import torch

from oracle import esm_oracle as eo


@pytest.fixture(scope="module")
def gp(golden_dir):
    return np.load(os.path.join(golden_dir, "golden_pppl_indels.npz")), pd.read_csv(os.path.join(golden_dir, "TOY_INDELS.csv"))


def test_golden_covers_indels_and_degenerate_lengths(gp, golden):
    g, df = gp
    seq = str(golden["seq"])
    lens = [len(s) for s in df["mutated_sequence"]]
    assert min(lens) == 2 and 3 in lens and 4 in lens                  # zero, one and two scored terms
    assert any(l > len(seq) for l in lens) and any(l < len(seq) for l in lens) and len(seq) in lens
    assert list(g["cli/esm2_toy/columns"]) == list(df.columns) + ["esm2_toy"]      # a single non-ESM1v checkpoint: no ensemble column
    assert list(g["cli/esm1v_toy_1/columns"]) == list(df.columns) + ["esm1v_toy_1", "Ensemble_ESM1v"]
    assert g["cli/esm2_toy"][list(df["mutant"]).index("len2")] == 0.0   # range(1, 1) is empty: sum([]) == 0


@pytest.mark.parametrize("name", ["esm2_toy", "esm1v_toy_1"])
def test_oracle_pppl_reproduces_reference(gp, golden_dir, name):
    """oracle.compute_pppl (compute_fitness.py:258-279 restated) vs the reference CLI column and, term by term, the
    reference model run with the reference's loop."""
    g, df = gp
    cfg, W = eo.load_checkpoint(os.path.join(golden_dir, name + ".pt"))
    worst_term = 0.0
    for r, s in enumerate(df["mutated_sequence"]):
        toks = eo.tokenize(s)[None, :]
        terms = []
        with torch.no_grad():
            for i in range(1, len(s) - 1):
                t = toks.copy()
                t[0, i] = eo.MASK
                terms.append(torch.log_softmax(eo.forward_logits(cfg, W, t), -1)[0, i, eo.get_idx(s[i])].item())
        ref = g[f"terms/{name}/{r}"]
        assert len(terms) == len(ref) == max(0, len(s) - 2)
        if len(ref):
            worst_term = max(worst_term, float(np.abs(np.array(terms) - ref).max()))
        assert abs(eo.compute_pppl(cfg, W, s) - g[f"cli/{name}"][r]) < 2e-4
    assert worst_term < 2e-5


def test_pack_sequences_equals_batch_converter(gp):
    from proteingym_amd import esm as pesm
    _, df = gp
    seqs = list(df["mutated_sequence"]) + ["", "XBZ-."]               # empty, rare symbols
    toks, off = pesm.pack_sequences(seqs)
    conv = pesm.Alphabet().get_batch_converter()
    for n, s in enumerate(seqs):
        _, _, t = conv([("x", s)])
        assert np.array_equal(toks[off[n]:off[n + 1]], t[0].astype(np.uint8))
        assert np.array_equal(t[0], eo.tokenize(s))
    assert off[-1] == toks.size and toks.dtype == np.uint8
    # anything but residue letters takes the slow path = Alphabet.encode = the reference tokenizer's behaviour
    # (esm/data.py:178-254; pinned against the live reference in tests/test_oracle_pinning.py): whitespace is dropped, a literal
    # special token is ONE token, a character outside the vocabulary is a KeyError -- not <unk>
    toks2, off2 = pesm.pack_sequences(["AC D", "AC<mask>D", "ACD"])
    assert toks2.tolist() == [0, 5, 23, 13, 2, 0, 5, 23, 32, 13, 2, 0, 5, 23, 13, 2] and off2.tolist() == [0, 5, 11, 16]
    for bad in ("acd", "AJC", "A*C"):
        with pytest.raises(KeyError):
            pesm.pack_sequences(["ACD", bad])
        with pytest.raises(KeyError):
            eo.tokenize(bad)
