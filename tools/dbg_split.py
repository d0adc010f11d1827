import sys, os
sys.path.insert(0, "/root/repo/mb-vae-dti_amd"); sys.path.insert(0, "/root/repo")
import torch
from mbvae_amd import ops, _lib
torch.manual_seed(0)
M, K, N, n = 65536, 128, 128, 64
bs = M // (n * n)
mask = torch.ones(bs, n, dtype=torch.bool, device="cuda")
dy = torch.randn(M, N, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5
y = torch.randn(M, N, device="cuda")
ws = ops.workspace(dy.device); st = torch.cuda.current_stream().cuda_stream
m8 = mask.view(torch.uint8)
def run(gate):
    dx = torch.empty(M, K, device="cuda")
    rc = _lib.call("mbvae_linear_bwd_input", dy.data_ptr(), N, w.data_ptr(), dx.data_ptr(), K, M, K, N, y.data_ptr() if gate else 0, 1.0,
              m8.data_ptr(), ops.MASK_EDGE, n, 0, ws.data_ptr(), ws.numel(), st)
    torch.cuda.synchronize()
    return dx
for gate in (False, True):
    a = run(gate)
    with _lib.option("NO_SPLIT_MFMA", 1):
        b = run(gate)
    d = (a - b).abs()
    print("gate", gate, "max", float(d.max()), "scale", float(b.abs().max()))
    bad = (d > 1e-3)
    print(" bad rows", int(bad.any(1).sum()), "bad cols", bad.any(0).nonzero().flatten().tolist()[:70])
    rows = bad.any(1).nonzero().flatten()
    print(" first bad rows", rows[:40].tolist())
