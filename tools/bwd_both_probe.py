"""Times mbvae_linear_bwd_both at the cfg3 shape (GPU box only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mb-vae-dti_amd")); sys.path.insert(0, ROOT)
import torch
from mbvae_amd import ops, _lib
bs, n, K, N = 256, 64, 64, 512
M = bs * n * n
dev = "cuda"
mask = torch.ones(bs, n, dtype=torch.bool, device=dev); m8 = mask.view(torch.uint8)
dy = torch.randn(M, N, device=dev); x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / 8
add = torch.randn(M, K, device=dev)
dx = torch.empty(M, K, device=dev); dw = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev)
ws = ops.workspace(dy.device); st = torch.cuda.current_stream().cuda_stream
def run():
    _lib.call("mbvae_linear_bwd_both", dy.data_ptr(), N, w.data_ptr(), x.data_ptr(), K, add.data_ptr(), dx.data_ptr(), K,
              dw.data_ptr(), db.data_ptr(), M, K, N, m8.data_ptr(), 2, n, ws.data_ptr(), ws.numel(), st)
for _ in range(3): run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): run()
b.record(); torch.cuda.synchronize()
print(os.environ.get("MBVAE_PB_DBG", "0"), os.environ.get("MBVAE_PB_MIRROR", "-"), f"{a.elapsed_time(b) / 20:.3f} ms/call")
