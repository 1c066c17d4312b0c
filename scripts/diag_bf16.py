import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from proteingym_amd import esm as pesm
from oracle import esm_oracle as eo
g = np.load("tests/golden/golden_esm.npz")
seq = str(g["seq"])
name = "esm1v_toy_1"
cfg, W = eo.load_checkpoint(f"tests/golden/{name}.pt")
toks = eo.tokenize(seq)[None]
omm = torch.Tensor.__matmul__
def mm(a, b):
    if a.dtype != torch.float32: return omm(a, b)
    return omm(a.bfloat16().float(), b.bfloat16().float())
torch.Tensor.__matmul__ = mm
with torch.no_grad():
    emu = torch.log_softmax(eo.forward_logits(cfg, W, toks), -1)[0].numpy()
torch.Tensor.__matmul__ = omm
ref = g[f"{name}/wt_logprobs"]
print("emulated bf16 vs ref", np.abs(emu - ref).max())
for prec in ("fp32", "f16x3", "bf16"):
    m = pesm.load_model_and_alphabet(f"tests/golden/{name}.pt", precision=prec)[0]
    lp = m.token_logprobs(toks)[0]
    print(prec, "vs ref", np.abs(lp - ref).max(), "vs emu", np.abs(lp - emu).max())
    m.close()
