"""GPU numerics of each HIP kernel against the torch fp32 op it replaces (through the C ABI)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from proteingym_amd import _lib

pytestmark = pytest.mark.gpu


def _p(a, ty=_lib._f32p):
    return a.ctypes.data_as(ty) if a is not None else None


@pytest.mark.parametrize("rows,D", [(7, 128), (1000, 1280), (33, 2560), (5, 64), (3, 320)])
def test_layernorm(lib, rows, D):
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((rows, D)) * 3 + 0.5).astype(np.float32)
    w = (1 + 0.1 * rng.standard_normal(D)).astype(np.float32)
    b = (0.1 * rng.standard_normal(D)).astype(np.float32)
    y = np.empty_like(x)
    _lib.check(lib.pgmi_op_layernorm(0, _p(x), _p(w), _p(b), rows, D, 1e-5, _p(y)))
    ref = torch.nn.functional.layer_norm(torch.from_numpy(x), (D,), torch.from_numpy(w), torch.from_numpy(b), 1e-5).numpy()
    assert np.abs(y - ref).max() < 2e-5


@pytest.mark.parametrize("M,N,K,epi,res", [
    (128, 128, 32, 0, False), (300, 384, 128, 0, False), (257, 1280, 1280, 1, False),
    (513, 1280, 5120, 0, True), (64, 3840, 1280, 0, False), (1, 128, 256, 1, True), (1000, 96, 64, 0, True)])
def test_gemm_fp32(lib, M, N, K, epi, res):
    rng = np.random.default_rng(1)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)   # asymmetric, non-square
    bias = rng.standard_normal(N).astype(np.float32)
    R = rng.standard_normal((M, N)).astype(np.float32) if res else None
    Cc = np.empty((M, N), np.float32)
    _lib.check(lib.pgmi_op_gemm(0, _lib.PREC_FP32, _p(A), _p(W), _p(bias), _p(R), M, N, K, epi, _p(Cc)))
    ref = torch.from_numpy(A).double() @ torch.from_numpy(W).double().T + torch.from_numpy(bias).double()
    if epi:
        ref = ref * 0.5 * (1.0 + torch.erf(ref / np.sqrt(2.0)))
    if res:
        ref = ref + torch.from_numpy(R).double()
    err = np.abs(Cc - ref.numpy()).max()
    assert err < 2e-5 * np.sqrt(K / 128), err


@pytest.mark.parametrize("prec", [_lib.PREC_FP32, _lib.PREC_F16X3])
@pytest.mark.parametrize("B,T,H,kv", [(2, 32, 2, None), (3, 70, 2, None), (2, 288, 4, None), (1, 1024, 2, None),
                                      (3, 72, 2, [72, 43, 1]), (2, 129, 1, [100, 129]), (2, 5, 1, None)])
def test_attention(lib, prec, B, T, H, kv):
    rng = np.random.default_rng(2)
    D = H * 64
    qkv = rng.standard_normal((B, T, 3 * D)).astype(np.float32)
    qkv[..., :D] *= 0.4
    kvl = np.asarray(kv, np.int32) if kv is not None else None
    ctx = np.empty((B, T, D), np.float32)
    qkv[0, 0, :D] *= 6.0            # a spiky query row: the running max jumps (online-softmax rescale path)
    qkv[..., 2 * D:] += rng.choice([0.0, 1e-3, 5.0], size=(B, T, 1)).astype(np.float32)   # tiny and large V rows
    _lib.check(lib.pgmi_op_attention(0, prec, _p(qkv), _p(kvl, _lib._i32p) if kvl is not None else None,
                                     B, T, H, 0, _p(ctx)))
    t = torch.from_numpy(qkv).double()
    q, k, v = [t[..., i * D:(i + 1) * D].reshape(B, T, H, 64).transpose(1, 2) for i in range(3)]
    s = q @ k.transpose(-1, -2)
    if kv is not None:
        mask = torch.arange(T)[None, :] >= torch.tensor(kv)[:, None]
        s = s.masked_fill(mask[:, None, None, :], float("-inf"))
    ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, T, D).numpy()
    if kv is not None:   # rows of padded queries are don't-care
        for b in range(B):
            ctx[b, kv[b]:] = 0
            ref[b, kv[b]:] = 0
    assert np.abs(ctx - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("prec", [_lib.PREC_FP32, _lib.PREC_F16X3])
def test_attention_rotary(lib, prec):
    rng = np.random.default_rng(3)
    B, T, H = 2, 50, 2
    D = H * 64
    qkv = rng.standard_normal((B, T, 3 * D)).astype(np.float32)
    ctx = np.empty((B, T, D), np.float32)
    _lib.check(lib.pgmi_op_attention(0, prec, _p(qkv), None, B, T, H, 1, _p(ctx)))
    t = torch.from_numpy(qkv)
    q, k, v = [t[..., i * D:(i + 1) * D].reshape(B, T, H, 64).transpose(1, 2) for i in range(3)]
    inv_freq = 1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64))
    fr = torch.einsum("i,j->ij", torch.arange(T).float(), inv_freq)
    emb = torch.cat((fr, fr), -1)
    cos, sin = emb.cos().double(), emb.sin().double()
    rot = lambda x: torch.cat((-x[..., 32:], x[..., :32]), -1)
    q, k, v = q.double(), k.double(), v.double()
    q = q * cos + rot(q) * sin
    k = k * cos + rot(k) * sin
    ref = (torch.softmax(q @ k.transpose(-1, -2), -1) @ v).transpose(1, 2).reshape(B, T, D).numpy()
    assert np.abs(ctx - ref).max() < 3e-5


GEMM_VARIANTS = [0,                  # the product's launch parameters
                 1002, 1008]         # 2 / 8 row panels per group of the tile order (the product: 4, or 8 for wide outputs)


@pytest.mark.parametrize("variant", GEMM_VARIANTS)
@pytest.mark.parametrize("M,N,K,epi,res", [(300, 384, 128, 0, False), (257, 1280, 1280, 1, False),
                                            (513, 1280, 5120, 0, True), (1, 128, 256, 1, True),
                                            (2300, 1280, 1280, 0, True),      # 45 tiles: every one as two half-height items
                                            (4200, 5120, 128, 1, False),      # 340 tiles on 256 CUs: a second item per workgroup
                                            (15100, 1280, 256, 0, True),      # 300 tiles: 256 full + 44 x 2 halves
                                            (15100, 1284, 256, 0, True),      # the same with a ragged last column tile (N % 256 = 4)
                                            (700, 1440, 480, 0, False),       # K = 480: 15 K tiles, an ODD count (ESM2-35M's embed_dim)
                                            (2300, 1920, 480, 1, False),      # its FC1 (GELU), half-height tail items with an odd tile count
                                            (300, 480, 96, 0, True)])         # three K tiles, residual
def test_gemm_f16x3(lib, monkeypatch, variant, M, N, K, epi, res):
    """Split-fp16 3-pass GEMM: fp32-class accuracy (same bound as the fp32 kernel)."""
    monkeypatch.setenv("PGMI_GEMM_VARIANT", str(variant))
    rng = np.random.default_rng(4)
    A = (rng.standard_normal((M, K)) * rng.choice([0.01, 1.0, 30.0], size=(M, 1))).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    R = rng.standard_normal((M, N)).astype(np.float32) if res else None
    Cc = np.empty((M, N), np.float32)
    _lib.check(lib.pgmi_op_gemm(0, _lib.PREC_F16X3, _p(A), _p(W), _p(bias), _p(R), M, N, K, epi, _p(Cc)))
    pre = torch.from_numpy(A).double() @ torch.from_numpy(W).double().T + torch.from_numpy(bias).double()
    ref = pre * 0.5 * (1.0 + torch.erf(pre / np.sqrt(2.0))) if epi else pre
    if res:
        ref = ref + torch.from_numpy(R).double()
    if not epi and not res:       # pure linear, no bias: error relative to each row's own magnitude
        C0 = np.empty((M, N), np.float32)
        _lib.check(lib.pgmi_op_gemm(0, _lib.PREC_F16X3, _p(A), _p(W), None, None, M, N, K, 0, _p(C0)))
        lin = (torch.from_numpy(A).double() @ torch.from_numpy(W).double().T).numpy()
        err = (np.abs(C0 - lin) / np.abs(A).max(1, keepdims=True)).max()
        assert err < 2e-6 * np.sqrt(K / 128), err     # small-magnitude rows keep fp32-class relative accuracy
    scale = np.abs(A).max(1, keepdims=True) * 1.0        # row-wise magnitude of the dot products
    err = (np.abs(Cc - ref.numpy()) / np.maximum(scale, 1.0)).max()
    assert err < 2e-5 * np.sqrt(K / 128), err


@pytest.mark.parametrize("M,N,K,epi,res", [(2300, 1280, 1280, 0, True),      # 45 tiles: every one of them as two half-height items
                                            (15100, 1280, 256, 1, False),     # 300 tiles on 256 CUs: 256 full + 44 x 2 halves
                                            (4300, 5120, 128, 0, True),       # 340 tiles: 256 + 84 x 2
                                            (200, 1280, 384, 1, True)])       # one row of tiles, rows past M in the lower halves
def test_gemm_f16x3_half_tail_bit_identical(lib, monkeypatch, M, N, K, epi, res):
    """The half-height tail items (gemm_f16.hip: TilePlan.half) compute every element exactly as a full tile does."""
    monkeypatch.setenv("PGMI_GEMM_VARIANT", "0")
    rng = np.random.default_rng(6)
    A = (rng.standard_normal((M, K)) * rng.choice([0.01, 1.0, 30.0], size=(M, 1))).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    R = rng.standard_normal((M, N)).astype(np.float32) if res else None
    out = {}
    for half in ("1", "0"):
        monkeypatch.setenv("PGMI_GEMM_HALF_TAIL", half)
        C = np.full((M, N), np.nan, np.float32)
        _lib.check(lib.pgmi_op_gemm(0, _lib.PREC_F16X3, _p(A), _p(W), _p(bias), _p(R), M, N, K, epi, _p(C)))
        out[half] = C
    assert np.isfinite(out["1"]).all()
    assert np.array_equal(out["1"], out["0"])


@pytest.mark.parametrize("planes", [0, 256])
def test_gemm_f16x3_gelu_epilogue_keeps_non_finite_values(lib, planes):
    """The fp16 range guard (PGMI_EOVERFLOW) looks for non-finite log-probabilities at the END of the network, and the LM head has a
    GELU in front of them: a NaN or an infinity that enters the GELU epilogue has to come out non-finite (a form built from max / min,
    which return their other operand for a NaN, would hand on a finite number), for the fp32 and the split-plane output alike; the
    finite rows next to it are untouched."""
    M, N, K = 300, 1280, 128
    rng = np.random.default_rng(3)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    clean = np.zeros((M, N), np.float32)
    _lib.check(lib.pgmi_op_gemm(0, _lib.PREC_F16X3, _p(A), _p(W), _p(bias), None, M, N, K, 1 + planes, _p(clean)))
    assert np.isfinite(clean).all()
    bad_bias = bias.copy()
    bad_bias[[5, 600, 1279]] = [np.nan, np.inf, -np.inf]
    out = np.zeros((M, N), np.float32)
    _lib.check(lib.pgmi_op_gemm(0, _lib.PREC_F16X3, _p(A), _p(W), _p(bad_bias), None, M, N, K, 1 + planes, _p(out)))
    assert not np.isfinite(out[:, [5, 600, 1279]]).any()
    keep = np.ones(N, bool)
    keep[[5, 600, 1279]] = False
    assert np.array_equal(out[:, keep], clean[:, keep])


@pytest.mark.parametrize("M,N,K,epi", [(2300, 1280, 128, 1),        # 45 tiles: every one of them as two half-height items
                                        (15100, 1280, 256, 2),       # 300 tiles on 256 CUs: 256 full + 44 x 2 halves; squared ReLU
                                        (4300, 5120, 128, 1),        # 340 tiles: 256 + 84 x 2; the FC1 kind (GELU)
                                        (200, 1344, 384, 1),         # rows past M in the lower halves; the last column tile holds one wave's worth of columns
                                        (600, 1312, 128, 2)])        # N % 64 != 0: the 8-byte store path
def test_gemm_f16x3_split_plane_epilogue_bit_identical_across_item_kinds(lib, monkeypatch, M, N, K, epi):
    """The split-plane (next GEMM's operand) epilogue through the full-height and the half-height items: same bits, and the
    values of the fp32 epilogue's to the operand's 22 bits."""
    monkeypatch.setenv("PGMI_GEMM_VARIANT", "0")
    rng = np.random.default_rng(7)
    A = (rng.standard_normal((M, K)) * rng.choice([0.01, 1.0, 30.0], size=(M, 1))).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    out = {}
    for half in ("1", "0"):
        monkeypatch.setenv("PGMI_GEMM_HALF_TAIL", half)
        C = np.full((M, N), np.nan, np.float32)
        _lib.check(lib.pgmi_op_gemm(0, _lib.PREC_F16X3, _p(A), _p(W), _p(bias), None, M, N, K, epi + 256, _p(C)))
        out[half] = C
    assert np.isfinite(out["1"]).all()
    assert np.array_equal(out["1"], out["0"])
    F = np.empty((M, N), np.float32)
    _lib.check(lib.pgmi_op_gemm(0, _lib.PREC_F16X3, _p(A), _p(W), _p(bias), None, M, N, K, epi, _p(F)))
    assert np.abs(out["1"] - F).max() <= np.abs(F).max() * 2.0 ** -21


@pytest.mark.parametrize("M,N,K,epi,res", [(15100, 1280, 256, 0, True), (15100, 1284, 256, 1, True), (70000, 1280, 128, 0, True),
                                            (300, 384, 128, 0, False)])
def test_gemm_f16x3_launch_parameters_bit_identical(lib, monkeypatch, M, N, K, epi, res):
    """The tile order does not touch a row's arithmetic."""
    rng = np.random.default_rng(8)
    A = (rng.standard_normal((M, K)) * rng.choice([0.01, 1.0, 30.0], size=(M, 1))).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    R = rng.standard_normal((M, N)).astype(np.float32) if res else None
    out = []
    for variant in GEMM_VARIANTS:
        monkeypatch.setenv("PGMI_GEMM_VARIANT", str(variant))
        C = np.full((M, N), np.nan, np.float32)
        _lib.check(lib.pgmi_op_gemm(0, _lib.PREC_F16X3, _p(A), _p(W), _p(bias), _p(R), M, N, K, epi, _p(C)))
        out.append(C)
    assert np.isfinite(out[0]).all()
    for C in out[1:]:
        assert np.array_equal(out[0], C)


@pytest.mark.parametrize("variant", [0, 2000])                   # the persistent ping-pong kernel in its one-plane form / the round-1 kernel
@pytest.mark.parametrize("M,N,K,epi,res", [(300, 384, 256, 0, False), (3600, 384, 128, 0, False),
                                            (3600, 256, 128, 1, False), (3600, 128, 256, 0, True), (70, 128, 128, 1, True),
                                            (2300, 1280, 1280, 0, True),        # half-height tail items, 20 K tiles of 64
                                            (4200, 5120, 1280, 1, False),       # FC1's shape class: a second item per workgroup, GELU
                                            (700, 1280, 5120, 0, True),         # FC2's: 80 K tiles
                                            (513, 1284, 192, 0, False)])        # ragged last column tile, an odd number of K tiles
def test_gemm_bf16(lib, monkeypatch, variant, M, N, K, epi, res):
    monkeypatch.setenv("PGMI_GEMM_VARIANT", str(variant))
    rng = np.random.default_rng(5)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    R = rng.standard_normal((M, N)).astype(np.float32) if res else None
    Cc = np.empty((M, N), np.float32)
    _lib.check(lib.pgmi_op_gemm(0, _lib.PREC_BF16, _p(A), _p(W), _p(bias), _p(R), M, N, K, epi, _p(Cc)))
    ref = (torch.from_numpy(A).bfloat16().double() @ torch.from_numpy(W).bfloat16().double().T
           + torch.from_numpy(bias).double())
    if epi:
        ref = ref * 0.5 * (1.0 + torch.erf(ref / np.sqrt(2.0)))
    if res:
        ref = ref + torch.from_numpy(R).double()
    assert np.abs(Cc - ref.numpy()).max() < 1e-4 * max(1.0, np.sqrt(K / 256))       # exact bf16 inputs, fp32 accumulate
    if not res and N % 4 == 0 and variant == 0:        # the bf16-plane epilogue (the next GEMM's operand): the same values rounded to bf16
        Cp = np.empty((M, N), np.float32)
        _lib.check(lib.pgmi_op_gemm(0, _lib.PREC_BF16, _p(A), _p(W), _p(bias), None, M, N, K, epi | 256, _p(Cp)))
        want = torch.from_numpy(Cc).bfloat16().float().numpy()
        assert np.array_equal(Cp, want)


def test_gemm16x_row_chunks_are_bit_identical(lib, monkeypatch):
    """The f16x3 GEMM addresses an operand with 32-bit byte offsets: launches above rows * K * 4 = 4 GiB are cut into row
    chunks (gemm_f16.hip launch_gemm16x).  Forced here at a small shape (PGMI_GEMM_MAX_ROWS): same bits as one launch,
    fp32 output with residual and the split-plane path behind GELU alike."""
    rng = np.random.default_rng(3)
    M, N, K = 2100, 640, 256
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    R = rng.standard_normal((M, N)).astype(np.float32)

    def run(epi, res):
        Cc = np.empty((M, N), np.float32)
        _lib.check(lib.pgmi_op_gemm(0, _lib.PREC_F16X3, _p(A), _p(W), _p(bias), _p(R) if res else None, M, N, K, epi, _p(Cc)))
        return Cc
    whole = [run(0, True), run(1, False)]
    monkeypatch.setenv("PGMI_GEMM_MAX_ROWS", "700")                      # -> chunks of 512 rows (whole 256-row tiles) + a tail
    parts = [run(0, True), run(1, False)]
    for a, b in zip(whole, parts):
        assert np.array_equal(a, b)


def test_gemm16x_operand_beyond_4_gib(lib):
    """ESM2-15B's FC2 shape class: K = 20480 allows 52 428 rows per launch; 52 480 rows = a 4.3 GB split operand.  Rows on
    both sides of the chunk boundary against float64 (advisor finding r2: this used to be refused with EINVAL)."""
    M, N, K = 52480, 64, 20480
    rng = np.random.default_rng(5)
    block = rng.standard_normal((256, K)).astype(np.float32)
    A = np.empty((M, K), np.float32)
    for r0 in range(0, M, 256):                                          # distinct rows without 1e9 random draws
        A[r0:r0 + 256] = block * np.float32(1.0 + (r0 // 256) * 1e-3)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    Cc = np.empty((M, N), np.float32)
    _lib.check(lib.pgmi_op_gemm(0, _lib.PREC_F16X3, _p(A), _p(W), _p(bias), None, M, N, K, 0, _p(Cc)))
    rows = [0, 255, 30000, 52223, 52224, 52428, 52429, M - 1]
    ref = A[rows].astype(np.float64) @ W.astype(np.float64).T + bias.astype(np.float64)
    err = np.abs(Cc[rows] - ref).max()
    assert err < 2e-5, err


@pytest.mark.parametrize("B,T,H,kv", [(3, 288, 4, None), (2, 1000, 2, None), (5, 130, 2, [130, 97, 128, 1, 66]), (70, 160, 20, None)])
def test_attention_launch_options_keep_the_bits(lib, B, T, H, kv):
    """The launch option of the dense attention (pgmi_set_option "att_xcd_local"): the XCD-local block order walks the same (sequence, head,
    query block) items through the same tiles in the same order -- the context rows must be those of the (query block, head, sequence) grid, bit
    for bit (fp32 output through the op entry; the split-plane output through a model: tests/test_gpu_esm.py)."""
    rng = np.random.default_rng(T)
    D = H * 64
    qkv = rng.standard_normal((B, T, 3 * D)).astype(np.float32)
    qkv[..., :D] *= 0.4
    qkv[0, 0, :D] *= 6.0
    kvl = np.asarray(kv, np.int32) if kv is not None else None

    def run():
        ctx = np.full((B, T, D), np.nan, np.float32)
        _lib.check(lib.pgmi_op_attention(0, _lib.PREC_F16X3, _p(qkv), _p(kvl, _lib._i32p) if kvl is not None else None, B, T, H, 0, _p(ctx)))
        if kv is not None:
            for b in range(B):
                ctx[b, kv[b]:] = 0
        return ctx
    try:
        _lib.check(lib.pgmi_set_option(b"att_xcd_local", 0))
        base = run()
        assert np.isfinite(base).all()
        for name, value in ((b"att_xcd_local", 1), (b"att_xcd_local", -1)):
            _lib.check(lib.pgmi_set_option(name, value))
            assert np.array_equal(run(), base), (name, value)
    finally:
        lib.pgmi_set_option(b"att_xcd_local", -1)


@pytest.mark.parametrize("B,T,H,kv", [(3, 288, 4, None), (2, 1000, 2, None), (1, 1024, 3, None), (5, 230, 2, [230, 197, 228, 1, 66]),
                                      (7, 224, 1, None), (33, 290, 20, None), (2, 737, 20, [737, 700]), (4, 30, 2, None), (3, 64, 2, [64, 33, 5]),
                                      (2, 5, 1, None), (3, 70, 2, None), (2, 129, 1, [100, 129])])
def test_attention_v3_bits_equal_v2(lib, B, T, H, kv):
    """The software-pipelined attention kernel (attention_f16x3_v3_kernel: every step runs P V of key tile kt - 1 and the scores of tile
    kt + 1 beside the softmax of tile kt; bundles {K tile m, V^T tile m - 2} in the ring; Q fragments re-read from LDS) walks every query
    row through the v2 kernel's arithmetic in the same order: context rows equal bit for bit, with and without key padding, one / two / four
    waves per workgroup, T on and off multiples of 32, a single key tile."""
    rng = np.random.default_rng(T + H)
    D = H * 64
    qkv = rng.standard_normal((B, T, 3 * D)).astype(np.float32)
    qkv[..., :D] *= 0.4
    qkv[0, 0, :D] *= 6.0
    qkv[..., 2 * D:] += rng.choice([0.0, 1e-3, 5.0], size=(B, T, 1)).astype(np.float32)
    kvl = np.asarray(kv, np.int32) if kv is not None else None

    def run():
        ctx = np.full((B, T, D), np.nan, np.float32)
        _lib.check(lib.pgmi_op_attention(0, _lib.PREC_F16X3, _p(qkv), _p(kvl, _lib._i32p) if kvl is not None else None, B, T, H, 0, _p(ctx)))
        if kv is not None:
            for b in range(B):
                ctx[b, kv[b]:] = 0
        return ctx
    try:
        _lib.check(lib.pgmi_set_option(b"att_v3", 0))
        base = run()
        assert np.isfinite(base).all()
        _lib.check(lib.pgmi_set_option(b"att_v3", 1))
        got = run()
        assert np.array_equal(got, base), float(np.nanmax(np.abs(got - base)))
    finally:
        lib.pgmi_set_option(b"att_v3", -1)
