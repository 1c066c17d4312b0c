"""proteingym_amd -- MI355X (gfx950) native zero-shot mutation-effect scoring for ProteinGym.

Host-side mirrors of the reference's scoring entry points over a C-ABI library of hand-written HIP kernels
(``libpgmi.so``, ``include/pgmi.h``):

    compute_fitness               ESM-1v / ESM-1b / ESM2 / MSA Transformer CLI (masked-, wt-marginals, pseudo-ppl)
    score_tranception_proteingym  Tranception CLI (autoregressive, mirrored, inference-time retrieval)
    run_benchmark, run_sharded    many assays on the GPUs of a node (one process per GPU)
    esm, tranception, msa_transformer, weights     in-process APIs with the reference's names
    build_native                  hipcc --offload-arch=gfx950 build of the library

There is no CPU fallback: every compute entry point raises ``PgmiError`` without the library or a GPU.
"""
__version__ = "0.1.0"
ABI_VERSION = 4
