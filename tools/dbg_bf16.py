import sys, os, math
sys.path.insert(0, "/root/repo/mb-vae-dti_amd"); sys.path.insert(0, "/root/repo")
import torch
from mbvae_amd import ops, _lib
from tests.test_gpu_bf16_storage import _layer_run
ref, _ = _layer_run(ops, False)
got, _ = _layer_run(ops, True)
for k in ref:
    d = (got[k] - ref[k]).abs()
    print(f"{k:28s} scale {float(ref[k].abs().max()):9.3e} rms {float(ref[k].pow(2).mean().sqrt()):9.3e} relmax {float(d.max() / ref[k].abs().max()):9.3e} relrms {float(d.pow(2).mean().sqrt() / ref[k].pow(2).mean().sqrt()):9.3e}")
d = (got["dE"] - ref["dE"]).abs()
idx = torch.nonzero(d > 0.5 * d.max())
print(idx[:10].tolist(), len(idx))
b, i, j, c = idx[0].tolist()
print("ref", ref["dE"][b, i, j, :8].tolist()); print("got", got["dE"][b, i, j, :8].tolist())
print("row err", float(d[b, i, j].max()), "row ref max", float(ref["dE"][b, i, j].abs().max()))
