"""Do the edge path's weight-gradient kernels overlap with the kernels that follow them in the backward pass when
they are issued on a second stream?  Times pairs (weight gradient, successor) back to back on one stream and
concurrently on two.  GPU box only.    python tools/overlap_probe.py
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mb-vae-dti_amd")); sys.path.insert(0, ROOT)
import torch
from mbvae_amd import ops

dev = torch.device("cuda")
bs, n, dx, de = 256, 64, 256, 64
M = bs * n * n
mask = torch.ones(bs, n, dtype=torch.bool, device=dev)
side = torch.cuda.Stream()


def wgrad_fn(K, N, act=0):
    x = torch.randn(M, K, device=dev)
    w = (torch.randn(N, K, device=dev) / K ** 0.5).requires_grad_(True)
    b = torch.zeros(N, device=dev, requires_grad=True)
    dy = torch.randn(M, N, device=dev)
    y = ops.linear(x, w, b, act=act, mask=mask, mask_mode=ops.MASK_EDGE, n_nodes=n)
    return lambda: torch.autograd.grad(y, (w, b), dy, retain_graph=True)


def dgrad_fn(K, N):
    x = torch.randn(M, K, device=dev, requires_grad=True)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    dy = torch.randn(M, N, device=dev)
    y = ops.linear(x, w, None, mask=mask, mask_mode=ops.MASK_EDGE, n_nodes=n)
    return lambda: torch.autograd.grad(y, x, dy, retain_graph=True)


def ln_bwd_fn():
    x = torch.randn(M, de, device=dev, requires_grad=True); r = torch.randn(M, de, device=dev, requires_grad=True)
    g = torch.ones(de, device=dev, requires_grad=True); b = torch.zeros(de, device=dev, requires_grad=True)
    y = ops.ln_residual(x, r, g, b, 1e-5, 0.1, 3)
    dy = torch.randn_like(y)
    return lambda: torch.autograd.grad(y, (x, r, g, b), dy, retain_graph=True)


def attn_bwd_fn():
    QKV = torch.randn(bs, n, 3 * dx, device=dev, requires_grad=True)
    E12 = (torch.randn(bs, n, n, 2 * dx, device=dev) * 0.5).requires_grad_(True)
    ya = torch.randn(bs, dx, device=dev, requires_grad=True); ym = torch.randn(bs, dx, device=dev, requires_grad=True)
    P, wV = ops.edge_attention(QKV, E12, ya, ym, mask, 8)
    dP, dW = torch.randn_like(P), torch.randn_like(wV)
    return lambda: torch.autograd.grad((P, wV), (QKV, E12, ya, ym), (dP, dW), retain_graph=True)


def timeit(f, iters=10):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def pair(name, fw, fm):
    def seq():
        fw(); fm()

    def par():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            fw()
        fm()
        cur.wait_stream(side)

    tw, tm, ts, tp = timeit(fw), timeit(fm), timeit(seq), timeit(par)
    print(f"{name:44s} wgrad {tw:6.3f}  successor {tm:6.3f}  one stream {ts:6.3f}  two streams {tp:6.3f} ms", flush=True)


pair("wgrad 128->64 (linE2) || dgrad 64->128", wgrad_fn(128, 64), dgrad_fn(64, 128))
pair("wgrad 64->128 (linE1) || LN bwd", wgrad_fn(64, 128, act=1), ln_bwd_fn())
pair("wgrad 256->64 (e_out) || attention bwd", wgrad_fn(256, 64), attn_bwd_fn())
pair("wgrad 64->512 (e12) || LN bwd", wgrad_fn(64, 512), ln_bwd_fn())
pair("wgrad 64->512 (e12) || dgrad 128->64", wgrad_fn(64, 512), dgrad_fn(128, 64))
