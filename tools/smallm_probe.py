"""Times the Linear entry points at the small-row shapes of the global-feature path and the encoders (GPU box only):
    python tools/smallm_probe.py            (MBVAE_NO_SMALLM=1 for the generic split-K route)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mb-vae-dti_amd")); sys.path.insert(0, ROOT)
import torch
from mbvae_amd import _lib, ops


def timeit(fn, iters=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / iters


ws = ops.workspace(torch.device("cuda", 0)); st = torch.cuda.current_stream().cuda_stream
tot = [0.0, 0.0, 0.0]
for M, K, N in ((256, 1024, 1024), (256, 1024, 2048), (256, 2048, 1024), (256, 256, 1024), (256, 2048, 512), (256, 512, 2048),
                (256, 512, 128), (256, 128, 512), (256, 4170, 512), (256, 1037, 2048), (32, 2048, 512), (32, 512, 2048)):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.zeros(N, device="cuda")
    dy = torch.randn(M, N, device="cuda"); y = torch.empty(M, N, device="cuda")
    dx = torch.empty_like(x); dw = torch.empty_like(w); db = torch.empty_like(b)
    tf = timeit(lambda: _lib.call("mbvae_linear_fwd", x.data_ptr(), K, w.data_ptr(), b.data_ptr(), y.data_ptr(), N, M, K, N, 1, 0, 0, 0,
                                  0.0, 0, 0, ws.data_ptr(), ws.numel(), st))
    td = timeit(lambda: _lib.call("mbvae_linear_bwd_input", dy.data_ptr(), N, w.data_ptr(), dx.data_ptr(), K, M, K, N, y.data_ptr(), 1.0,
                                  0, 0, 0, 0, ws.data_ptr(), ws.numel(), st))
    tw = timeit(lambda: _lib.call("mbvae_linear_bwd_weight", dy.data_ptr(), N, x.data_ptr(), K, dw.data_ptr(), db.data_ptr(), M, K, N,
                                  y.data_ptr(), 1.0, 0, 0, 0, 0, ws.data_ptr(), ws.numel(), st))
    fl = 2.0 * M * K * N
    print(f"M={M:4d} K={K:5d} N={N:5d}  fwd {tf:7.1f} us ({fl / tf / 1e6:5.1f} TF/s)  dgrad {td:7.1f} us ({fl / td / 1e6:5.1f})  "
          f"wgrad {tw:7.1f} us ({fl / tw / 1e6:5.1f})", flush=True)
    tot = [tot[0] + tf, tot[1] + td, tot[2] + tw]
print(f"sum: fwd {tot[0]:.1f} us  dgrad {tot[1]:.1f} us  wgrad {tot[2]:.1f} us")
