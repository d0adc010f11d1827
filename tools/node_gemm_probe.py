"""Linear layers of the node path (M = 16384 rows) and the global-feature path (M = 256) at the benchmark batch: time
of the forward, input-gradient and weight-gradient entry points under the generic kernel's tile choices (GEMM_TILE).
GPU box only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mb-vae-dti_amd")); sys.path.insert(0, ROOT)
import torch
from mbvae_amd import _lib, ops as O

g = torch.Generator().manual_seed(0)
st = lambda: torch.cuda.current_stream().cuda_stream


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


tiles = [int(t) for t in os.environ.get("PROBE_TILES", "-1,0,3,4").split(",")]
for M, K, N in ((16384, 256, 256), (16384, 256, 768), (256, 1024, 1024), (256, 1024, 2048), (256, 256, 1024)):
    x = torch.randn(M, K, generator=g).cuda(); w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda(); dy = torch.randn(M, N, generator=g).cuda()
    y = torch.empty(M, N, device="cuda"); dx = torch.empty(M, K, device="cuda"); dw = torch.empty(N, K, device="cuda"); db = torch.empty(N, device="cuda")
    ws = O.workspace(x.device)
    fl = 2.0 * M * K * N
    for tile in tiles:
        with _lib.option("GEMM_TILE", tile):
            f = timed(lambda: _lib.call("mbvae_linear_fwd", x.data_ptr(), K, w.data_ptr(), b.data_ptr(), y.data_ptr(), N, M, K, N, 0, None, 0, 0, 0.0, 0, 0, ws.data_ptr(), ws.numel(), st()))
            bi = timed(lambda: _lib.call("mbvae_linear_bwd_input", dy.data_ptr(), N, w.data_ptr(), dx.data_ptr(), K, M, K, N, None, 1.0, None, 0, 0, 0, ws.data_ptr(), ws.numel(), st()))
            bw = timed(lambda: _lib.call("mbvae_linear_bwd_weight", dy.data_ptr(), N, x.data_ptr(), K, dw.data_ptr(), db.data_ptr(), M, K, N, None, 1.0, None, 0, 0, 0, ws.data_ptr(), ws.numel(), st()))
        print(f"M={M:6d} K={K:5d} N={N:5d} tile {tile:2d}: fwd {f:6.1f} us ({fl / f / 1e6:5.1f} TF/s)  bwd_input {bi:6.1f} us ({fl / bi / 1e6:5.1f})  bwd_weight {bw:6.1f} us ({fl / bw / 1e6:5.1f})", flush=True)
