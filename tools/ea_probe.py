"""Times the edge-attention entry points alone at the cfg3 shape (GPU box only).
    python tools/ea_probe.py [iters]        or under  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mb-vae-dti_amd")); sys.path.insert(0, ROOT)
import torch
from mbvae_amd import ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = "cuda"
bs, n, dx = 256, 64, 256
sizes = torch.randint(20, n + 1, (bs,), device=dev) if os.environ.get("EA_RAGGED") else torch.full((bs,), n, device=dev)
mask = torch.arange(n, device=dev)[None, :] < sizes[:, None]
QKV = torch.randn(bs, n, 3 * dx, device=dev, requires_grad=True)
E12 = (torch.randn(bs, n, n, 2 * dx, device=dev) * 0.5).requires_grad_(True)
ya = torch.randn(bs, dx, device=dev, requires_grad=True); ym = torch.randn(bs, dx, device=dev, requires_grad=True)
dP, dW = torch.randn(bs, n, n, dx, device=dev), torch.randn(bs, n, dx, device=dev)
tf = tb = 0.0
for it in range(iters + 2):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    P, wV = ops.edge_attention(QKV, E12, ya, ym, mask, 8)
    e[1].record()
    torch.autograd.backward((P, wV), (dP, dW))
    e[2].record()
    torch.cuda.synchronize()
    if it >= 2:
        tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
    QKV.grad = E12.grad = ya.grad = ym.grad = None
    del P, wV
print(f"edge_attn fwd {1e3 * tf / iters:8.1f} us   bwd (incl. memsets) {1e3 * tb / iters:8.1f} us")
