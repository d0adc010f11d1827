"""tests/golden/layer_bs256.npz: ONE transformer layer of the reference at the benchmark batch (256 ragged graphs x 64 atoms,
real widths), forward + backward on the CPU, down-sampled: 1024 / 4096 / 32 rows of the three outputs, up to 2048 entries
and the norm of every parameter and input gradient.

Two runs of the reference's layer are stored side by side:
  *  float64, in chunks of 32 graphs (the layer treats graphs independently; parameter gradients add up over chunks): the
     values a test compares with, free of fp32 rounding;
  *  float32 on the whole batch, as the reference itself would run it: stored as `ref32_err/<name>`, the largest distance of
     the reference's OWN fp32 result from the float64 one over the sampled entries, relative to the tensor's largest
     magnitude.  It says how much of a tolerance fp32 arithmetic at this size consumes by itself (sums over 4096 pair
     rows and 256 graphs; a ReLU whose pre-activation rounds to the other side of zero).
The CPU oracle is checked against the reference on the same case before the fixture is written.
Build container only (needs /root/reference; about 20 s on 8 cores, ~30 GB of memory at the peak).

    python tools/gen_golden_bs256.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE); sys.path.insert(0, ROOT)
import ref_loader  # noqa: E402
from tests._weights import bs256_layer_case, sample_entries, synthetic_state_dict  # noqa: E402

ref = ref_loader.load()
from oracle import denoiser as O_den  # noqa: E402

torch.set_float32_matmul_precision("highest")        # (the reference switches CPU matmuls to bf16 at import)
c = bs256_layer_case()
pick = c["pick"]


def new_layer(dtype):
    layer = ref.gt.XEyTransformerLayer(dx=256, de=64, dy=1024, n_head=8, dim_ffX=256, dim_ffE=128, dim_ffy=1024).eval()
    layer.load_state_dict(synthetic_state_dict({k: tuple(v.shape) for k, v in layer.state_dict().items()}))
    return layer.to(dtype)


def run(dtype, chunk):
    layer = new_layer(dtype)
    outs, gin = [], []
    for s in range(0, 256, chunk):
        sl = slice(s, s + chunk)
        X, E, y = (c[k][sl].to(dtype).requires_grad_(True) for k in ("X", "E", "y"))
        oX, oE, oy = layer(X, E, y, c["mask"][sl])
        loss = (oX * c["cX"][sl].to(dtype)).sum() + (oE * c["cE"][sl].to(dtype)).sum() + (oy * c["cy"][sl].to(dtype)).sum()
        loss.backward()
        outs.append((oX.detach(), oE.detach(), oy.detach()))
        gin.append((X.grad, E.grad, y.grad))
        print(f"  {dtype}: graphs {s}..{s + chunk - 1} done", flush=True)
    oX, oE, oy = (torch.cat([o[i] for o in outs]) for i in range(3))
    grads = {"dX": torch.cat([g[0] for g in gin]), "dE": torch.cat([g[1] for g in gin]), "dy": torch.cat([g[2] for g in gin])}
    grads.update({k: p.grad for k, p in layer.named_parameters()})
    return layer, (oX, oE, oy), grads


layer32, (oX, oE, oy), g32 = run(torch.float32, 256)
with torch.no_grad():
    qX, qE, qy = O_den.xey_layer(layer32.state_dict(), c["X"], c["E"], c["y"], c["mask"], 8)
for a, b, nm in ((qX, oX, "X"), (qE, oE, "E"), (qy, oy, "y")):
    err = float((a - b).abs().max() / b.abs().max())
    assert err <= 1e-5, f"oracle != reference for out {nm}: {err}"
print("oracle forward == reference forward (fp32)", flush=True)
s32 = {k: v.detach().reshape(-1)[sample_entries(k, v.numel())].double() for k, v in g32.items()}
n32 = {k: float(v.detach().double().norm()) for k, v in g32.items()}
o32 = (oX.reshape(-1, 256)[pick["rows_X"]].double(), oE.reshape(-1, 64)[pick["rows_E"]].double(), oy[pick["rows_y"]].double())
del layer32, g32, oX, oE, oy, qX, qE, qy

_, (oX, oE, oy), g64 = run(torch.float64, 32)
out = {"out_X": oX.reshape(-1, 256)[pick["rows_X"]], "out_E": oE.reshape(-1, 64)[pick["rows_E"]], "out_y": oy[pick["rows_y"]]}
for nm, a, b in (("out_X", o32[0], out["out_X"]), ("out_E", o32[1], out["out_E"]), ("out_y", o32[2], out["out_y"])):
    out["ref32_err/" + nm] = ((a - b).abs().max() / b.abs().max()).float()
    out[nm] = b.float()
    print(f"reference fp32 vs float64  {nm:28s} {float(out['ref32_err/' + nm]):.2e}")
for k, gr in g64.items():
    flat = gr.detach().reshape(-1)
    smp = flat[sample_entries(k, flat.numel())]
    out["grad/" + k] = smp.float()                         # (float64 values rounded once: 6e-8 relative)
    out["norm/" + k] = flat.norm().float()
    out["absmax/" + k] = flat.abs().max().float()
    out["ref32_err/" + k] = ((s32[k] - smp).abs().max() / flat.abs().max()).float()
    out["ref32_norm_err/" + k] = torch.tensor(abs(n32[k] - float(flat.norm())) / float(flat.norm()), dtype=torch.float32)
    print(f"reference fp32 vs float64  {k:28s} {float(out['ref32_err/' + k]):.2e}   norm {float(out['ref32_norm_err/' + k]):.2e}")
path = os.path.join(ROOT, "tests", "golden", "layer_bs256.npz")
np.savez_compressed(path, **{k: v.numpy() for k, v in out.items()})
print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")
