"""Micro-benchmark of the linear kernels on the E-path shapes (GPU box only)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mb-vae-dti_amd")); sys.path.insert(0, ROOT)
import torch
from mbvae_amd import ops

def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

bs, n = 256, 64
M = bs * n * n
mask = torch.ones(bs, n, dtype=torch.bool, device="cuda")
shapes = ((64, 256), (256, 64), (64, 128), (128, 64))
if len(sys.argv) > 2:
    shapes = ((int(sys.argv[1]), int(sys.argv[2])),)
if len(sys.argv) > 3:          # K N M n: rows and nodes per graph (M must equal bs*n*n for the edge mask)
    M = int(sys.argv[3]); n = int(sys.argv[4]); bs = M // (n * n)
    mask = torch.ones(bs, n, dtype=torch.bool, device="cuda")
for K, N in shapes:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.zeros(N, device="cuda")
    dy = torch.randn(M, N, device="cuda")
    for mode, mm in (("none", None), ("edge", mask)):
        with torch.no_grad():
            t = timeit(lambda: ops.linear(x, w, b, mask=mm, mask_mode=ops.MASK_EDGE, n_nodes=n))
        print(f"fwd  K={K:4d} N={N:4d} mask={mode:5s} {t*1000:8.1f} us  {2*M*K*N/t/1e9:7.1f} TFLOP/s  {(M*K+M*N)*4/t/1e6:7.1f} GB/s", flush=True)
    from mbvae_amd import _lib
    ws = ops.workspace(x.device); st = torch.cuda.current_stream().cuda_stream
    m8 = mask.view(torch.uint8)
    dx = torch.empty_like(x); dw = torch.empty_like(w); db = torch.empty_like(b)
    t = timeit(lambda: _lib.call("mbvae_linear_bwd_input", dy.data_ptr(), N, w.data_ptr(), dx.data_ptr(), K, M, K, N, 0, 1.0,
                                 m8.data_ptr(), ops.MASK_EDGE, n, 0, ws.data_ptr(), ws.numel(), st))
    print(f"dgrad K={K:4d} N={N:4d} mask=edge  {t*1000:8.1f} us  {2*M*K*N/t/1e9:7.1f} TFLOP/s  {(M*K+M*N)*4/t/1e6:7.1f} GB/s", flush=True)
    t = timeit(lambda: _lib.call("mbvae_linear_bwd_weight", dy.data_ptr(), N, x.data_ptr(), K, dw.data_ptr(), db.data_ptr(), M, K, N,
                                 0, 1.0, m8.data_ptr(), ops.MASK_EDGE, n, 0, ws.data_ptr(), ws.numel(), st))
    print(f"wgrad K={K:4d} N={N:4d} mask=edge  {t*1000:8.1f} us  {2*M*K*N/t/1e9:7.1f} TFLOP/s  {(M*K+M*N)*4/t/1e6:7.1f} GB/s", flush=True)
