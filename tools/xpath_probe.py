"""Node-path Linear shapes (M = bs*n = 16384 rows) under the generic kernel's tile choices (GPU box only)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mb-vae-dti_amd")); sys.path.insert(0, ROOT)
import torch
from mbvae_amd import ops, _lib

def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1000

M = 16384
mask = torch.ones(256, 64, dtype=torch.bool, device="cuda"); m8 = mask.view(torch.uint8)
ws = ops.workspace(mask.device); st = torch.cuda.current_stream().cuda_stream
for K, N in ((256, 256), (256, 768), (768, 256)):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.zeros(N, device="cuda")
    dy = torch.randn(M, N, device="cuda"); dx = torch.empty_like(x); dw = torch.empty_like(w); db = torch.empty_like(b); y = torch.empty(M, N, device="cuda")
    for tile in (-1, -2):                         # -2: column groups off (generic LDS-tiled kernel)
        with _lib.option("NO_COLGROUPS", 1 if tile == -2 else 0):
            f = timeit(lambda: _lib.call("mbvae_linear_fwd", x.data_ptr(), K, w.data_ptr(), b.data_ptr(), y.data_ptr(), N, M, K, N, 0,
                                         m8.data_ptr(), ops.MASK_NODE, 64, 0.0, 0, 0, ws.data_ptr(), ws.numel(), st))
            d = timeit(lambda: _lib.call("mbvae_linear_bwd_input", dy.data_ptr(), N, w.data_ptr(), dx.data_ptr(), K, M, K, N, 0, 1.0,
                                         m8.data_ptr(), ops.MASK_NODE, 64, 0, ws.data_ptr(), ws.numel(), st))
            g = timeit(lambda: _lib.call("mbvae_linear_bwd_weight", dy.data_ptr(), N, x.data_ptr(), K, dw.data_ptr(), db.data_ptr(), M, K, N,
                                         0, 1.0, m8.data_ptr(), ops.MASK_NODE, 64, 0, ws.data_ptr(), ws.numel(), st))
        print(f"K={K:4d} N={N:4d} tile={tile:2d}  fwd {f:6.1f} us  dgrad {d:6.1f} us  wgrad {g:6.1f} us   ({2*M*K*N/f/1e6:5.1f} / {2*M*K*N/d/1e6:5.1f} / {2*M*K*N/g/1e6:5.1f} TFLOP/s)", flush=True)
