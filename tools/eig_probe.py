"""Time of the batched Jacobi eigensolver (mbvae_sym_eig) on graph Laplacians of the benchmark shape (256 graphs x 64 atoms,
bond density of the synthetic molecules) and on a ragged sampling batch; eigenvalues against torch.linalg.eigh in float64.
GPU box only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mb-vae-dti_amd")); sys.path.insert(0, ROOT)
import torch
from mbvae_amd import ops

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
for bs, n, dens, ragged in ((256, 64, 0.10, False), (1250, 57, 0.10, True), (256, 64, 0.5, False)):
    U = (torch.rand(bs, n, n, generator=g) < dens).float().triu(1)
    if ragged:
        sizes = torch.randint(8, n + 1, (bs,), generator=g)
        m = (torch.arange(n)[None] < sizes[:, None]).float()
        U = U * m[:, :, None] * m[:, None, :]
    A = U + U.transpose(1, 2)
    L = (torch.diag_embed(A.sum(-1)) - A).to(dev)
    w, V = ops.sym_eig(L)
    ref = torch.linalg.eigvalsh(L.double().cpu())
    err = float((w.cpu().double() - ref).abs().max())
    res = float((L @ V - V * w.unsqueeze(1)).abs().max())
    orth = float((V.transpose(1, 2) @ V - torch.eye(n, device=dev)).abs().max())
    w2, V2 = ops.sym_eig(L)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(20):
        ops.sym_eig(L)
    b.record(); torch.cuda.synchronize()
    print(f"bs {bs} n {n} density {dens} ragged {ragged}: {a.elapsed_time(b) / 20 * 1e3:.1f} us per call; eigenvalue error {err:.2e}, "
          f"residual {res:.2e}, orthogonality {orth:.2e}, rerun bitwise equal {torch.equal(w, w2) and torch.equal(V, V2)}", flush=True)
