"""GPU parity: every HIP kernel, through the C ABI, against the CPU oracle and the golden fixtures.

Tolerance: fp32 results within 1e-5 of the reference tensor's largest magnitude (no floor on that scale; measured
errors are <= 3.2e-6 everywhere, see tests/_util.py for how to list them); class ids, masks and one-hot samples
bit-exact.  The few checks with another bound say why next to it.
"""
import math

import pytest
import torch
import torch.nn.functional as F

from mbvae_amd import _lib
from tests._util import assert_close, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from mbvae_amd import ops as _ops
    return _ops


@pytest.fixture
def default_routes():
    """Tests that assert WHICH entry points ran pin the library's route switches to their defaults (the suite may be
    run with MBVAE_NO_SPLIT_MFMA / MBVAE_NO_E12_RECOMPUTE in the environment to exercise the other routes)."""
    from mbvae_amd import _lib
    with _lib.option("NO_SPLIT_MFMA", 0), _lib.option("NO_E12_RECOMPUTE", 0), _lib.option("NO_BWD_BOTH", 0):
        yield


def dev(t):
    return t.cuda() if torch.is_tensor(t) else t


def ragged_mask(bs, n, gen):
    sizes = torch.randint(1, n + 1, (bs,), generator=gen)
    sizes[0] = n
    return torch.arange(n)[None, :] < sizes[:, None]


# ------------------------------------------------------------------------------------------------ linear
@pytest.mark.parametrize("M,K,N,act,mode", [
    (300, 64, 256, 0, "none"), (1000, 5, 128, 1, "none"), (4 * 9, 13, 24, 1, "node"), (4 * 81, 16, 32, 0, "edge"),
    (2 * 64 * 64, 256, 64, 0, "edge"), (256, 1037, 2048, 1, "none"), (256, 1024, 1024, 0, "none"), (70, 128, 5, 0, "none"),
    (5000, 64, 128, 1, "none"), (33, 4170, 512, 0, "none"),
    # edge-path shapes that take the persistent resident-weight kernel (M >= 32768, K*N <= 16384)
    (16 * 64 * 64, 64, 256, 0, "edge"), (16 * 64 * 64, 256, 64, 0, "edge"), (16 * 64 * 64, 64, 128, 1, "none"),
    (16 * 64 * 64, 128, 64, 0, "edge"), (9 * 64 * 64, 64, 256, 1, "edge"), (16 * 64 * 64, 128, 128, 1, "edge"),
    # fused e_mul|e_add projection and its gradients (weights up to 128 KB resident in LDS)
    (16 * 64 * 64, 64, 512, 0, "edge"),
    # node-path shapes: column groups of the tall-skinny kernel (weights sliced over workgroup columns)
    (4 * 64 * 64, 256, 768, 1, "edge"), (4 * 64 * 64, 256, 768, 0, "none"), (2 * 64 * 64, 128, 512, 1, "none"),
    (4 * 64 * 64, 384, 640, 0, "edge"), (4 * 64 * 64, 256, 256, 1, "edge"),
    # few output tiles, long reduction: one 32x32 tile per workgroup, the reduction split over its four waves
    (256, 4170, 512, 0, "none"), (256, 1024, 2048, 1, "none"), (256, 256, 1024, 0, "none"), (32, 2048, 512, 1, "none"),
    (256, 2048, 100, 1, "none"), (200, 1000, 72, 0, "none"), (256, 512, 128, 1, "none"), (48, 136, 1024, 0, "none"),
    # thin layers over the pair rows (the 5 bond classes at either end of the edge MLPs): streaming kernels, no tiles;
    # odd row counts exercise the clamped second row of a trip
    (16 * 64 * 64, 5, 128, 1, "none"), (16 * 64 * 64, 128, 5, 0, "none"), (9 * 64 * 64 + 3, 5, 128, 0, "none"),
    (9 * 64 * 64 + 3, 128, 5, 1, "none"), (40000, 8, 64, 1, "none"), (40000, 64, 8, 0, "none"), (33000, 3, 256, 0, "none"),
    (33000, 256, 1, 0, "none"),
])
def test_linear_fwd_bwd(ops, M, K, N, act, mode):
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g) * 0.1
    cot = torch.randn(M, N, generator=g)
    mask, n, mm = None, 0, ops.MASK_NONE
    keep = torch.ones(M, 1)
    if mode == "node":
        n = 9; mask = ragged_mask(M // n, n, g); mm = ops.MASK_NODE; keep = mask.reshape(M, 1).float()
    elif mode == "edge":
        n = 9 if M == 4 * 81 else 64; bs = M // (n * n)
        mask = ragged_mask(bs, n, g); mm = ops.MASK_EDGE
        keep = (mask.unsqueeze(1) & mask.unsqueeze(2)).reshape(M, 1).float()
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    y = F.linear(xr, wr, br)
    if act:
        # a pre-activation within rounding of zero may land on either side of the ReLU in two correct fp32
        # implementations (expected about once per 10^7 outputs); its cotangent is dropped so that the gradient
        # comparison does not depend on that coin flip
        cot = cot * (y.detach().abs() > 1e-5)
    y = (F.relu(y) if act else y) * keep
    (y * cot).sum().backward()
    xg, wg, bg = (t.clone().cuda().requires_grad_(True) for t in (x, w, b))
    yg = ops.linear(xg, wg, bg, act=act, mask=dev(mask), mask_mode=mm, n_nodes=n)
    (yg * cot.cuda()).sum().backward()
    assert_close(yg, y, what="y")
    assert_close(xg.grad, xr.grad, what="dx")
    assert_close(wg.grad, wr.grad, what="dw")
    assert_close(bg.grad, br.grad, what="db")


@pytest.mark.parametrize("M,K,N,act,mode", [
    (16384, 256, 256, 1, "node"), (16384, 256, 256, 0, "none"), (8192, 256, 768, 0, "node"), (4096, 128, 192, 1, "none"),
    (512, 1024, 1024, 1, "none"), (1024, 2048, 128, 0, "none"), (640, 96, 64, 0, "node"),
    # a sampling chunk's global-feature product: 320 tiles of 64x64 (ragged last row tile), reduction split in three
    (1250, 1024, 1024, 1, "none"),
])
def test_generic_gemm_on_the_bf16_pipeline_matches_the_fp32_route(ops, M, K, N, act, mode):
    """Whole aligned tiles of the generic LDS-tiled kernel run on the bf16 matrix pipeline (operand octets split once,
    three planes in LDS); NO_GEMM_SPLIT keeps them on the fp32 matrix instruction.  Forward, input, weight and bias
    gradients (the weight gradient's reduction runs over the rows and is split over workgroups; the bias gradient comes
    from the staged octets' column sums) of both routes against float64, 1e-5 of each tensor's scale; run twice: same bits."""
    from mbvae_amd import _lib
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) / math.sqrt(K); b = torch.randn(N, generator=g) * 0.1
    cot = torch.randn(M, N, generator=g)
    mask, n, mm, keep = None, 0, ops.MASK_NONE, torch.ones(M, 1)
    if mode == "node":
        n = 64; mask = ragged_mask(M // n, n, g); mm = ops.MASK_NODE; keep = mask.reshape(M, 1).float()
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    y = F.linear(xr, wr, br)
    if act:
        cot = cot * (y.detach().abs() > 1e-5)            # (a pre-activation within rounding of zero: either side is right)
    y = (F.relu(y) if act else y) * keep.double()
    (y * cot.double()).sum().backward()

    def run(no_split):
        with _lib.option("NO_GEMM_SPLIT", no_split), _lib.option("NO_TSGEMM", 1), _lib.option("NO_SMALLM", 1):
            xg, wg, bg = (t.clone().cuda().requires_grad_(True) for t in (x, w, b))
            yg = ops.linear(xg, wg, bg, act=act, mask=dev(mask), mask_mode=mm, n_nodes=n)
            (yg * cot.cuda()).sum().backward()
            torch.cuda.synchronize()
        return yg.detach(), xg.grad, wg.grad, bg.grad

    for no_split in (0, 1):
        got = run(no_split)
        for a, r, name in zip(got, (y, xr.grad, wr.grad, br.grad), ("y", "dx", "dw", "db")):
            assert_close(a, r, what=f"{name} (NO_GEMM_SPLIT={no_split})")
        again = run(no_split)
        assert all(torch.equal(a, b_) for a, b_ in zip(got, again))


def test_linear_relu_dropout_statistics(ops):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4096, 64, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(128, 64, generator=g) / 8).cuda().requires_grad_(True)
    b = torch.zeros(128).cuda().requires_grad_(True)
    y0 = ops.linear(x, w, b, act=1)
    y1 = ops.linear(x, w, b, act=1, drop_p=0.25, seed=1234)
    y2 = ops.linear(x, w, b, act=1, drop_p=0.25, seed=1234)
    assert torch.equal(y1, y2)                                   # same seed -> same mask
    pos = y0 > 0
    kept = (y1 > 0) & pos
    frac = kept.sum().item() / pos.sum().item()
    assert abs(frac - 0.75) < 0.01
    assert_close(y1[kept], y0[kept] / 0.75, what="kept values scaled by 1/(1-p)")
    # backward uses the saved post-dropout activation as the gate
    y1.sum().backward()
    gx = x.grad.clone(); x.grad = None
    (y0 * kept.float() / 0.75).sum().backward()
    assert_close(gx, x.grad, what="dropout backward")


# ------------------------------------------------------------------------------------------------ edge attention
@pytest.mark.parametrize("bs,n,dx,H", [(4, 9, 32, 4), (2, 64, 256, 8), (3, 17, 64, 4), (2, 70, 32, 2)])
def test_edge_attention(ops, bs, n, dx, H):
    from oracle import denoiser as O
    g = torch.Generator().manual_seed(bs * n + dx)
    mask = ragged_mask(bs, n, g)
    xm = mask.unsqueeze(-1).float(); em = xm.unsqueeze(2) * xm.unsqueeze(1)
    Q, K, V = (torch.randn(bs, n, dx, generator=g) * xm for _ in range(3))
    E1, E2 = (torch.randn(bs, n, n, dx, generator=g) * 0.5 * em for _ in range(2))
    ya, ym = torch.randn(bs, dx, generator=g), torch.randn(bs, dx, generator=g) * 0.3
    cP, cW = torch.randn(bs, n, n, dx, generator=g), torch.randn(bs, n, dx, generator=g)
    cpu = [t.clone().requires_grad_(True) for t in (Q, K, V, E1, E2, ya, ym)]
    P, wV = O.edge_attention(*cpu, mask, H)
    ((P * cP).sum() + (wV * cW).sum()).backward()
    # the kernel reads q|k|v and e_mul|e_add as column blocks of two fused projection outputs
    QKV = torch.cat((Q, K, V), -1).cuda().requires_grad_(True)
    E12 = torch.cat((E1, E2), -1).cuda().requires_grad_(True)
    yag, ymg = (t.clone().cuda().requires_grad_(True) for t in (ya, ym))
    Pg, wVg = ops.edge_attention(QKV, E12, yag, ymg, mask.cuda(), H)
    ((Pg * cP.cuda()).sum() + (wVg * cW.cuda()).sum()).backward()
    assert_close(Pg, P, what="P"); assert_close(wVg, wV, what="wV")
    got = list(QKV.grad.split(dx, -1)) + list(E12.grad.split(dx, -1)) + [yag.grad, ymg.grad]
    for name, a, b in zip(("dQ", "dK", "dV", "dE1", "dE2", "dye_add", "dye_mul"), got, cpu):
        assert_close(a, b.grad, what=name)


# ------------------------------------------------------------------------------------------------ residual + LayerNorm
@pytest.mark.parametrize("rows,D", [(4 * 81, 16), (1000, 64), (300, 256), (256, 1024), (77, 32), (50, 2048), (64, 24), (10, 48)])
def test_ln_residual(ops, rows, D):
    g = torch.Generator().manual_seed(rows + D)
    x, r = torch.randn(rows, D, generator=g) * 2 + 0.5, torch.randn(rows, D, generator=g)
    gamma, beta = 1 + 0.2 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
    cot = torch.randn(rows, D, generator=g)
    c = [t.clone().requires_grad_(True) for t in (x, r, gamma, beta)]
    y = F.layer_norm(c[0] + c[1], (D,), c[2], c[3], 1e-5)
    (y * cot).sum().backward()
    q = [t.clone().cuda().requires_grad_(True) for t in (x, r, gamma, beta)]
    yg = ops.ln_residual(q[0], q[1], q[2], q[3], 1e-5)
    (yg * cot.cuda()).sum().backward()
    assert_close(yg, y, what="y")
    for name, a, b in zip(("dx", "dr", "dgamma", "dbeta"), q, c):
        assert_close(a.grad, b.grad, what=name)
    # r = None path
    y2 = ops.ln_residual(q[0].detach(), None, q[2].detach(), q[3].detach(), 1e-5)
    assert_close(y2, F.layer_norm(x, (D,), gamma, beta, 1e-5), what="plain LN")


def test_ln_residual_dropout_consistent(ops):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(512, 64, generator=g).cuda().requires_grad_(True)
    r = torch.randn(512, 64, generator=g).cuda().requires_grad_(True)
    gamma, beta = torch.ones(64).cuda().requires_grad_(True), torch.zeros(64).cuda().requires_grad_(True)
    y = ops.ln_residual(x, r, gamma, beta, 1e-5, 0.3, 77)
    y.pow(2).sum().backward()
    # LayerNorm output is invariant to the scale, so a dropped element shows up as zero gradient on r
    dropped = (r.grad == 0)
    frac = dropped.float().mean().item()
    assert abs(frac - 0.3) < 0.02
    kept = ~dropped
    assert_close(r.grad[kept], x.grad[kept] / 0.7, rtol=1e-5, what="dr = dz/(1-p) on kept elements")


# ------------------------------------------------------------------------------------------------ pooling
@pytest.mark.parametrize("shape", [(4, 9, 32), (4, 9, 9, 16), (3, 64, 256), (2, 64, 64, 64), (5, 7, 7, 8)])
def test_pool_stats(ops, shape):
    from oracle import denoiser as O
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g)
    x[0, 0] = 0                               # padded-looking rows stay in the statistics
    dims = tuple(range(1, len(shape) - 1))
    cot = torch.randn(shape[0], 4 * shape[-1], generator=g)
    xr = x.clone().requires_grad_(True)
    s = O.pool_stats(xr, dims)
    (s * cot).sum().backward()
    xg = x.clone().cuda().requires_grad_(True)
    sg = ops.pool_stats(xg)
    (sg * cot.cuda()).sum().backward()
    assert_close(sg, s, what="stats")
    assert_close(xg.grad, xr.grad, what="dx")


# ------------------------------------------------------------------------------------------------ symmetrise + mask
@pytest.mark.parametrize("D,zero_diag,skip_w", [(16, False, 0), (5, True, 5), (64, False, 0), (5, True, 13)])
def test_sym_mask(ops, D, zero_diag, skip_w):
    g = torch.Generator().manual_seed(D)
    bs, n = 3, 11
    mask = ragged_mask(bs, n, g)
    x = torch.randn(bs, n, n, D, generator=g)
    skip = torch.randn(bs, n, n, skip_w, generator=g) if skip_w else None
    em = (mask.unsqueeze(1) & mask.unsqueeze(2)).unsqueeze(-1).float()
    od = (~torch.eye(n, dtype=torch.bool))[None, :, :, None].float() if zero_diag else 1.0
    xr = x.clone().requires_grad_(True)
    t = xr + skip[..., :D] if skip is not None else xr
    t = t * od
    ref = 0.5 * (t + t.transpose(1, 2)) * em
    cot = torch.randn(bs, n, n, D, generator=g)
    (ref * cot).sum().backward()
    xg = x.clone().cuda().requires_grad_(True)
    out = ops.sym_mask(xg, mask.cuda(), skip=dev(skip), zero_diag=zero_diag)
    (out * cot.cuda()).sum().backward()
    assert_close(out, ref, what="sym"); assert_close(xg.grad, xr.grad, what="d sym")


# ------------------------------------------------------------------------------------------------ losses
def test_masked_ce_golden(ops):
    g = load_golden("heads")
    pX, pE = g["rc_pred_X"].cuda().requires_grad_(True), g["rc_pred_E"].cuda().requires_grad_(True)
    lX, sX = ops.masked_ce(pX, g["rc_true_X"].cuda())
    lE, sE = ops.masked_ce(pE, g["rc_true_E"].cuda())
    loss = lX * 1 + lE * 5
    loss.backward()
    assert_close(loss, g["rc_loss"], what="loss")
    assert_close(pX.grad, g["rc_dX"], what="dX"); assert_close(pE.grad, g["rc_dE"], what="dE")
    assert_close(sE[2:7] / sE[1], g["rc_bce_E"], what="per-class BCE E")
    assert_close(sX[2:10] / sX[1], g["rc_bce_X"], what="per-class BCE X")


def test_masked_ce_large(ops):
    from oracle import heads as OH
    g = torch.Generator().manual_seed(9)
    R, C = 200_000, 5
    logits = torch.randn(R, C, generator=g) * 3
    cls = torch.randint(0, C, (R,), generator=g)
    tgt = F.one_hot(cls, C).float() * (torch.rand(R, 1, generator=g) < 0.8)
    lr = logits.clone().requires_grad_(True)
    ref = OH.reconstruction_loss(lr[:1], lr, tgt[:1], tgt, [0, 1]); ref.backward()
    lg = logits.clone().cuda().requires_grad_(True)
    out, _ = ops.masked_ce(lg, tgt.cuda()); out.backward()
    assert_close(out, ref, what="ce"); assert_close(lg.grad, lr.grad, atol=1e-9, what="dlogits")


def test_kl_reparam_golden(ops):
    g = load_golden("heads")
    mu, lv = g["kl_mu"].cuda().requires_grad_(True), g["kl_logvar"].cuda().requires_grad_(True)
    z, kl = ops.kl_reparam(mu, lv, g["kl_eps"].cuda())
    assert_close(z, g["kl_z"], what="z"); assert_close(kl, g["kl_value"], what="kl")
    cot = torch.randn(z.shape, generator=torch.Generator().manual_seed(1))
    ((z * cot.cuda()).sum() + 3 * kl).backward()
    mr, lr = g["kl_mu"].clone().requires_grad_(True), g["kl_logvar"].clone().requires_grad_(True)
    from oracle import heads as OH
    zr = mr + g["kl_eps"] * torch.exp(0.5 * lr)
    ((zr * cot).sum() + 3 * OH.kl_divergence(mr, lr)).backward()
    assert_close(mu.grad, mr.grad, what="dmu"); assert_close(lv.grad, lr.grad, what="dlogvar")


# ------------------------------------------------------------------------------------------------ diffusion samplers
def test_apply_noise_bit_exact_golden(ops):
    g = load_golden("noise")
    mask = g["mask"].bool().cuda()
    for mode in ("train", "eval"):
        Xc, Ec, Xt, Et = ops.apply_noise(g["X"].cuda(), g["E"].cuda(), mask, g[f"{mode}_alpha_t_bar"].cuda(), g["mx"].cuda(),
                                         g["me"].cuda(), g[f"{mode}_expX"].cuda(), g[f"{mode}_expE"].cuda())
        assert torch.equal(Xt.cpu(), g[f"{mode}_X_t"]) and torch.equal(Et.cpu(), g[f"{mode}_E_t"])
        assert torch.equal(Xc.cpu(), g[f"{mode}_Xc"]) and torch.equal(Ec.cpu(), g[f"{mode}_Ec"])


def test_apply_noise_bit_exact_large(ops):
    from oracle import diffusion as OD
    g = torch.Generator().manual_seed(5)
    bs, n, cx, ce = 16, 64, 5, 5
    mask = ragged_mask(bs, n, g)
    mx = torch.tensor([.72, .12, .10, .04, .02]); me = torch.tensor([.90, .05, .03, .01, .01])
    Xc0 = torch.multinomial(mx, bs * n, True, generator=g).reshape(bs, n)
    Ec0 = torch.multinomial(me, bs * n * n, True, generator=g).reshape(bs, n, n)
    Ec0 = torch.triu(Ec0, 1); Ec0 = Ec0 + Ec0.transpose(1, 2)
    X0 = F.one_hot(Xc0, cx).float() * mask.unsqueeze(-1)
    E0 = F.one_hot(Ec0, ce).float(); E0[:, torch.arange(n), torch.arange(n)] = 0
    sch = OD.Schedule(500)
    t_int = torch.randint(0, 501, (bs, 1), generator=g).float()
    eX = torch.empty(bs * n, cx).exponential_(1, generator=g); eE = torch.empty(bs * n * n, ce).exponential_(1, generator=g)
    Xt, Et, npar, Xc, Ec = OD.apply_noise(sch, mx, me, X0, E0, mask, t_int, eX, eE)
    gXc, gEc, gXt, gEt = ops.apply_noise(X0.cuda(), E0.cuda(), mask.cuda(), npar["alpha_t_bar"].cuda(), mx.cuda(), me.cuda(),
                                         eX.cuda(), eE.cuda())
    assert torch.equal(gXt.cpu(), Xt) and torch.equal(gEt.cpu(), Et)
    assert torch.equal(gXc.cpu(), Xc) and torch.equal(gEc.cpu(), Ec)
    assert torch.equal(gEt, gEt.transpose(1, 2))


def _flip_ok(prob, cls_gpu, cls_ref, expo):
    """Samples may differ from the reference only where the two best ratios are within fp32 noise."""
    bad = cls_gpu != cls_ref
    if not bad.any():
        return True
    ratio = (prob / expo)[bad]
    top2 = ratio.topk(2, dim=-1).values
    return bool(((top2[:, 0] - top2[:, 1]) / top2[:, 0] < 1e-4).all())


def test_masks_select_not_multiply_nan_poisoned_memory(ops):
    """Rows of masked nodes / pairs may hold anything, NaN and inf included (kernels that skip padding leave them
    unwritten): every mask in the product path must be a select.  (1) gradient mode with frozen weights takes the
    no-gradient attention forward, which does not write P on masked pairs; (2) graphs with more than 64 nodes run
    the generic kernels over buffers the padding-skipping kernels left partly unwritten."""
    from mbvae_amd.graph_transformer import GraphTransformer
    from mbvae_amd.trainstep import CFG3, synthetic_graphs
    cfg = dict(CFG3); cfg["n_layers"] = 2
    torch.manual_seed(4)
    net = GraphTransformer(**cfg).cuda().eval()
    g = torch.Generator().manual_seed(12)

    def poison():
        junk = [torch.full((64 << 20,), v, device="cuda") for v in (float("nan"), float("inf"), float("-inf"))]   # 3 x 256 MB
        del junk                                     # back to the caching allocator: the next torch.empty reuses them

    for bs, n, sizes in ((6, 40, torch.tensor([40, 7, 33, 1, 25, 12])), (2, 70, torch.tensor([70, 31]))):
        X0, E0, mask = synthetic_graphs(bs, n, "cuda", g, sizes=sizes)
        Xin = torch.cat((X0, torch.randn(bs, n, 8, generator=g).cuda() * mask.unsqueeze(-1)), 2)
        E0 = E0 * (mask.unsqueeze(1) & mask.unsqueeze(2)).unsqueeze(-1)
        yin = torch.randn(bs, 1037, generator=g).cuda()
        ref = net(Xin, E0, yin, mask)                                   # training-form kernels: every row written
        net.requires_grad_(False)
        poison()
        frozen = net(Xin, E0, yin, mask)                                # grad mode on, nothing requires grad
        net.requires_grad_(True)
        for a, b, nm in ((frozen.X, ref.X, "X"), (frozen.E, ref.E, "E"), (frozen.y, ref.y, "y")):
            assert bool(torch.isfinite(a).all()), f"n={n}: non-finite {nm} with frozen weights"
            assert_close(a, b.detach(), what=f"n={n} frozen-weight forward {nm}")
        with torch.no_grad():
            net(Xin, E0, yin, mask)                                     # first sight of the mask
            poison()
            again = net(Xin, E0, yin, mask)                             # second sight: row order (sparse only if n <= 64)
        assert net._pad_cache["order"] is not None and net._pad_cache["order"].sparse == (n <= 64)
        for a, b, nm in ((again.X, ref.X, "X"), (again.E, ref.E, "E"), (again.y, ref.y, "y")):
            assert bool(torch.isfinite(a).all()), f"n={n}: non-finite {nm} on the inference path"
            assert_close(a, b.detach(), what=f"n={n} inference forward {nm}")


def test_dropout_masks_agree_between_forward_and_backward_at_edge_widths(ops):
    """Dropout masks are never stored: the backward kernels regenerate them from (seed, element index).  At the edge
    path's shapes (2^16 pair rows, the resident-weight kernels) the elements dropped in the forward pass are exactly
    the ones that receive no gradient."""
    g = torch.Generator().manual_seed(21)
    M, p = 16 * 64 * 64, 0.1
    # feed-forward hidden layer: ReLU + dropout in the product's epilogue, gate = the saved output
    x = torch.randn(M, 64, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(128, 64, generator=g) / 8).cuda().requires_grad_(True)
    b = (torch.randn(128, generator=g) * 0.1).cuda().requires_grad_(True)
    cot = torch.randn(M, 128, generator=g).cuda()
    y0 = ops.linear(x.detach(), w.detach(), b.detach(), act=1)
    y = ops.linear(x, w, b, act=1, drop_p=p, seed=991)
    keep = ((y != 0) | (y0 == 0)).float()                               # dropped = positive before, zero after
    frac = 1 - float(((y != 0) & (y0 > 0)).sum()) / float((y0 > 0).sum())
    assert abs(frac - p) < 5e-3
    (y * cot).sum().backward()
    xr, wr, br = (t.detach().clone().requires_grad_(True) for t in (x, w, b))
    (ops.linear(xr, wr, br, act=1) * keep / (1 - p) * cot).sum().backward()
    assert_close(x.grad, xr.grad, what="dx under dropout"); assert_close(w.grad, wr.grad, what="dw under dropout")
    assert_close(b.grad, br.grad, what="db under dropout")
    # residual dropout in front of the LayerNorm
    xe, r = torch.randn(M, 64, generator=g).cuda().requires_grad_(True), torch.randn(M, 64, generator=g).cuda().requires_grad_(True)
    gamma, beta = (1 + 0.1 * torch.randn(64, generator=g)).cuda(), torch.zeros(64).cuda()
    out = ops.ln_residual(xe, r, gamma, beta, 1e-5, p, 4242)
    (out * cot[:, :64]).sum().backward()
    dropped = r.grad == 0
    assert abs(float(dropped.float().mean()) - p) < 5e-3
    assert_close(r.grad[~dropped], xe.grad[~dropped] / (1 - p), what="dr = dz / (1 - p) on kept elements")
    # the forward dropped exactly those elements: rebuild the output from the mask the BACKWARD revealed
    ref = F.layer_norm(xe.detach() + r.detach() * (~dropped).float() / (1 - p), (64,), gamma, beta, 1e-5)
    assert_close(out, ref, what="forward mask == backward mask")


def test_reverse_sample_golden(ops):
    g = load_golden("reverse_step")
    from oracle import diffusion as OD
    sch = OD.Schedule(500)
    mask = g["mask"].bool()
    t, s = g["t"], g["s"]
    beta, a_s, a_t = sch.beta(t_normalized=t), sch.alpha_bar(t_normalized=s), sch.alpha_bar(t_normalized=t)
    Xc, Ec, Xs, Es, pX, pE = ops.reverse_sample(g["logits_X"].cuda(), g["logits_E"].cuda(), g["X_t"].cuda(), g["E_t"].cuda(),
                                                mask.cuda(), beta.cuda(), a_s.cuda(), a_t.cuda(), g["mx"].cuda(), g["me"].cuda(),
                                                g["expX"].cuda(), g["expE"].cuda(), want_probs=True)
    xm = mask.unsqueeze(-1).float(); em = xm.unsqueeze(2) * xm.unsqueeze(1)
    assert_close(pX.cpu() * xm, g["prob_X"] * xm, rtol=1e-5, what="prob X")
    assert_close(pE.cpu() * em, g["prob_E"] * em, rtol=1e-5, what="prob E")
    assert torch.equal(Xs.cpu(), g["X_s"]) and torch.equal(Es.cpu(), g["E_s"])


# ------------------------------------------------------------------------------------------------ whole denoiser
def _build_denoiser(dims, sd):
    from mbvae_amd.graph_transformer import GraphTransformer
    net = GraphTransformer(**dims)
    net.load_state_dict(sd)
    return net.cuda()


def test_denoiser_tiny_forward_backward_golden():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    g = load_golden("denoiser_tiny")
    dims = dict(n_layers=2, input_dims={"X": 13, "E": 5, "y": 45}, hidden_mlp_dims={"X": 24, "E": 16, "y": 40},
                hidden_dims={"dx": 32, "de": 16, "dy": 32, "n_head": 4, "dim_ffX": 24, "dim_ffE": 16, "dim_ffy": 40},
                output_dims={"X": 5, "E": 5, "y": 32})
    net = _build_denoiser(dims, g["sd"]).eval()
    X, E, y = (g[k].cuda().requires_grad_(True) for k in ("X", "E", "y"))
    out = net(X, E, y, g["mask"].bool().cuda())
    assert_close(out.X, g["out_X"], what="X"); assert_close(out.E, g["out_E"], what="E"); assert_close(out.y, g["out_y"], what="y")
    ((out.X * g["cX"].cuda()).sum() + (out.E * g["cE"].cuda()).sum() + (out.y * g["cy"].cuda()).sum()).backward()
    assert_close(X.grad, g["dX"], what="dX"); assert_close(E.grad, g["dE"], what="dE")
    assert_close(y.grad, g["dy"], what="dy")
    for k, p in net.named_parameters():
        assert_close(p.grad, g["grad"][k], what="grad " + k)


def test_layer_tiny_golden():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from mbvae_amd.graph_transformer import XEyTransformerLayer
    g = load_golden("layer_tiny")
    layer = XEyTransformerLayer(dx=32, de=16, dy=32, n_head=4, dim_ffX=24, dim_ffE=16, dim_ffy=40)
    layer.load_state_dict(g["sd"]); layer = layer.cuda().eval()
    mask = g["mask"].bool().cuda()
    nX, nE, ny = layer.self_attn(g["X"].cuda(), g["E"].cuda(), g["y"].cuda(), mask)
    assert_close(nX, g["blk_X"], what="block X"); assert_close(nE, g["blk_E"], what="block E"); assert_close(ny, g["blk_y"], what="block y")
    X, E, y = layer(g["X"].cuda(), g["E"].cuda(), g["y"].cuda(), mask)
    assert_close(X, g["out_X"], what="layer X"); assert_close(E, g["out_E"], what="layer E"); assert_close(y, g["out_y"], what="layer y")


def test_denoiser_realwidth_golden():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from tests._weights import synthetic_state_dict
    g = load_golden("denoiser_realwidth")
    shapes = {str(k): tuple(int(d) for d in dims if d > 0) for k, dims in zip(g["shape_names"], g["shape_dims"])}
    dims = dict(n_layers=1, input_dims={"X": 13, "E": 5, "y": 1037}, hidden_mlp_dims={"X": 256, "E": 128, "y": 2048},
                hidden_dims={"dx": 256, "de": 64, "dy": 1024, "n_head": 8, "dim_ffX": 256, "dim_ffE": 128, "dim_ffy": 1024},
                output_dims={"X": 5, "E": 5, "y": 1024})
    net = _build_denoiser(dims, synthetic_state_dict(shapes)).eval()
    with torch.no_grad():
        out = net(g["X"].cuda(), g["E"].cuda(), g["y"].cuda(), g["mask"].bool().cuda())
    assert_close(out.X, g["out_X"], what="X"); assert_close(out.E, g["out_E"], what="E")
    assert_close(out.y, g["out_y"], what="y")


# ------------------------------------------------------------------------------------------------ eigensolver + features
@pytest.mark.parametrize("bs,n", [(8, 64), (5, 9), (3, 33), (4, 1), (2, 2)])
def test_sym_eig(ops, bs, n):
    g = torch.Generator().manual_seed(n)
    A = (torch.rand(bs, n, n, generator=g) < 0.15).float()
    A = torch.triu(A, 1); A = A + A.transpose(1, 2)
    if bs > 2:
        A[1, : n // 2, n // 2:] = 0; A[1, n // 2:, : n // 2] = 0           # two components
        A[2] = 0                                                            # no edges at all
    L = torch.diag_embed(A.sum(-1)) - A
    W, V = ops.sym_eig(L.cuda())
    W, V = W.cpu().double(), V.cpu().double()
    ref = torch.linalg.eigvalsh(L.double())
    assert_close(W, ref, rtol=1e-5, what="eigenvalues")
    assert bool((W[:, 1:] >= W[:, :-1]).all()), "ascending"
    Ld = L.double()
    resid = Ld @ V - V * W.unsqueeze(1)
    assert float(resid.abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
    eye = torch.eye(n, dtype=torch.float64)
    # fp32 Jacobi: ~600 accumulated plane rotations per column leave V orthonormal to a few 1e-5
    assert float((V.transpose(1, 2) @ V - eye).abs().max()) < 5e-5


@pytest.mark.parametrize("bs,n,p", [(64, 64, 0.06), (7, 33, 0.15), (5, 9, 0.4), (3, 2, 1.0), (2, 1, 0.0), (300, 40, 0.08)])
def test_cycle_counts_kernel_matches_adjacency_power_formulas(ops, bs, n, p):
    """One-kernel cycle counts against the oracle's restatement of the reference's adjacency-power formulas on random
    molecule-like graphs (symmetric 0/1 adjacency, no self loops): integer counts, so the match is exact."""
    from oracle.features import cycle_features
    g = torch.Generator().manual_seed(100 * n + bs)
    U = (torch.rand(bs, n, n, generator=g) < p).float().triu(1)
    A = U + U.transpose(1, 2)
    xr, yr = cycle_features(A.double())
    xg, yg = ops.cycle_counts(A.cuda())
    assert torch.equal(xg.cpu(), xr.float()), "per-node 3/4/5-cycle counts"
    assert torch.equal(yg.cpu(), yr.float()), "per-graph 3/4/5/6-cycle counts"


def test_extra_features_golden():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from mbvae_amd.features import ExtraFeatures, ExtraMolecularFeatures
    g = load_golden("features")
    mask = g["mask"].bool().cuda()
    X_t, E_t = g["X_t"].cuda(), g["E_t"].cuda()
    torch.manual_seed(0)
    Xg, yg = ExtraFeatures(int(g["max_n_nodes"]))(E_t, mask)
    Xm, ym = ExtraMolecularFeatures(g["valencies"].tolist(), float(g["max_weight"]), g["atom_weights"].tolist())(X_t, E_t)
    Xe = torch.cat((Xg, Xm), 2).cpu()
    ye = torch.cat((yg, ym, g["t"].cuda()), 1).cpu()
    ref = g["X_extra"]
    assert_close(Xe[..., :3], ref[..., :3], what="cycle features")
    assert_close(Xe[..., 6:], ref[..., 6:], what="charge / valency")
    assert_close(ye, g["y_extra"], rtol=1e-5, what="y_extra (n, cycles, components, eigenvalues, weight, t)")
    # eigenvector features are gauge dependent: check them against the Laplacian instead of LAPACK's choice
    L = g["laplacian"].double()
    m = g["mask"].bool()
    for b in range(L.shape[0]):
        nv = int(m[b].sum())
        for col in (4, 5):
            v = Xe[b, :, col].double()
            if float(v.abs().max()) == 0:
                continue
            lam = float(v @ (L[b] @ v) / (v @ v))
            assert float(((L[b] @ v) - lam * v).abs().max()) < 1e-4, "not an eigenvector"
            assert abs(float(v @ v) - 1.0) < 1e-4 and float(v[nv:].abs().max() if nv < len(v) else 0) == 0


# ------------------------------------------------------------------------------------------------ full model (boundary)
def _tiny_full_model(phase, g, seed_sd):
    from mbvae_amd.full import FullDTIModel
    cx, emb, hid = 8, 32, 16
    infos = {"general": {"atom_types": ["C", "O", "P", "N", "S", "Cl", "F", "H"], "max_n_nodes": 64,
                         "atom_valencies": [4, 2, 3, 3, 2, 1, 1, 1],
                         "atom_weights": [12.011, 15.999, 30.974, 14.007, 32.06, 35.45, 18.998, 1.008], "max_weight": 1000},
             "dataset": {"node_count_distribution": {"3": 1, "5": 2, "7": 3, "9": 2}, "node_marginals": g["mx"].tolist(),
                         "edge_marginals": g["me"].tolist()}}
    gt = dict(n_layers=2, input_dims={"X": cx + 8, "E": 5, "y": emb + 13}, output_dims={"X": cx, "E": 5, "y": emb},
              hidden_mlp_dims={"X": 24, "E": 16, "y": 40},
              hidden_dims={"dx": 32, "de": 16, "dy": 32, "n_head": 4, "dim_ffX": 24, "dim_ffE": 16, "dim_ffy": 40})
    model = FullDTIModel(
        phase=phase, finetune_score=None, learning_rate=5e-4, weight_decay=1e-4, scheduler="const", weights=[1.0, 0.001, 0.5, 3.0],
        dti_weights=[1.0, 0.903964, 0.282992, 0.755172] if phase == "train" else None, diff_weights=[1, 5], contrastive_temp=0.1,
        drug_features={"FP-Morgan": 64, "EMB-BiomedGraph": 24},
        target_features={"FP-ESP": 48, "EMB-ESM": 40, "EMB-NT": 20} if phase == "train" else {},
        embedding_dim=emb, hidden_dim=hid, factor=4, n_layers=2, activation="relu", dropout=0.1, bias=True,
        encoder_type="transformer", aggregator_type="attentive", diffusion_steps=500, sample_every_val=5,
        val_samples_per_embedding=2, test_samples_per_embedding=2, graph_transformer_kwargs=gt, dataset_infos=infos)
    if seed_sd is not None:
        missing, unexpected = model.load_state_dict(seed_sd, strict=False)
        assert not missing, f"state_dict keys missing from the reference checkpoint: {missing}"
    return model.cuda()


def test_full_model_common_step_golden():
    """FullDTIModel._common_step('train') on the reference's own random draws: every intermediate and loss component."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from mbvae_amd.containers import GraphData
    g = load_golden("common_step")
    model = _tiny_full_model("train", g, g["sd"])
    model.eval()                       # dropout off (the fixture was captured that way) ...
    model.training = True              # ... but t ~ U{0..T} as in training
    model.replay = {k: g[k] for k in ("t_int", "expX", "expE", "lcc_noise", "eps")}
    mask = g["mask"].bool().cuda()
    gd = GraphData(X=g["X"].cuda(), E=g["E"].cuda(), node_mask=mask)
    y = {k: (v.bool() if k.endswith("_mask") else v).cuda() for k, v in g["y"].items()}
    batch = {"id": torch.arange(4), "y": y,
             "drug": {"features": {k: v.cuda() for k, v in g["feat_d"].items()}, "representations": {"smiles": ["C"] * 4}, "G": gd},
             "target": {"features": {k: v.cuda() for k, v in g["feat_t"].items()}}}
    bd, emb, pred, loss = model._common_step(batch, "train")
    total = loss.compute_loss(model.weights)
    assert torch.equal(bd.graph_data.X_t.cpu(), g["X_t"]) and torch.equal(bd.graph_data.E_t.cpu(), g["E_t"])
    assert_close(bd.graph_data.y_extra, g["y_extra"], what="y_extra")
    assert_close(bd.graph_data.X_extra[..., :3], g["X_extra"][..., :3], what="X_extra cycles")
    assert_close(bd.graph_data.X_extra[..., 6:], g["X_extra"][..., 6:], what="X_extra charge/valency")
    assert_close(emb.drug_mu, g["drug_mu"], what="mu"); assert_close(emb.drug_logvar, g["drug_logvar"], what="logvar")
    assert_close(emb.drug_embedding, g["drug_embedding"], what="z")
    assert_close(emb.target_embedding, g["target_embedding"], what="target embedding")
    assert_close(emb.fused_embedding, g["fused"], what="fused")
    assert_close(loss.accuracy, g["loss_accuracy"], what="accuracy"); assert_close(loss.complexity, g["loss_complexity"], what="kl")
    assert_close(loss.contrastive, g["loss_contrastive"], what="contrastive")
    # eigenvector input features are gauge dependent (sign): inject the reference's and compare the decoder exactly
    bd.graph_data.X_extra = g["X_extra"].cuda()
    out = model.denoise_drug_graph(emb, bd)
    assert_close(out.X, g["pred_X"], what="pred X"); assert_close(out.E, g["pred_E"], what="pred E")
    rec = model.reconstruction_head(out.X, out.E, bd.graph_data.X, bd.graph_data.E)
    assert_close(rec, g["loss_reconstruction"], what="reconstruction")
    loss.reconstruction = rec
    total = loss.compute_loss(model.weights)
    assert_close(total, g["loss_total"], what="total")
    total.backward()
    for k, v in g["grad"].items():
        assert_close(dict(model.named_parameters())[k].grad, v, what="grad " + k)
    for k, v in g["gradnorm"].items():
        got = dict(model.named_parameters())[k].grad
        assert got is not None, k
        # (the attention-logit bias gradients are zero in exact arithmetic -- softmax shift invariance -- and hold rounding noise)
        bound = 1e-6 if k.endswith("to_attn_logits.bias") else 1e-5 * float(v)
        assert abs(float(got.norm()) - float(v)) <= bound, f"grad norm {k}: {float(got.norm())} vs {float(v)}"


def test_full_model_sampling_and_val_step_run():
    """sample_batch (5 reverse steps) yields symmetric, masked class-id graphs; validation_step returns a finite NLL."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from mbvae_amd.containers import GraphData
    g = load_golden("common_step")
    sd = {k: v for k, v in g["sd"].items() if k.startswith("drug_") or k.startswith("noise_schedule")}
    model = _tiny_full_model("pretrain_drug", g, sd)
    model.T = 5
    model.noise_schedule = type(model.noise_schedule)(5).cuda()
    model.eval()
    torch.manual_seed(0)
    emb = torch.randn(3, 32, device="cuda")
    groups = model.sample_batch(emb, num_nodes=torch.tensor([9, 4, 6]), num_samples_per_embedding=2, return_graphs=True)
    assert len(groups) == 3 and all(len(x) == 2 for x in groups)
    for grp, n in zip(groups, (9, 4, 6)):
        for atoms, bonds in grp:
            assert atoms.shape == (n,) and bonds.shape == (n, n)
            assert int(atoms.min()) >= 0 and int(atoms.max()) < 8 and torch.equal(bonds, bonds.T)
            assert int(bonds.diagonal().abs().sum()) == 0 and int(bonds.max()) < 5
    # size-bucketed sampling: same contract, molecules come back in the caller's order
    groups = model.sample_batch(emb, num_nodes=torch.tensor([9, 4, 6]), num_samples_per_embedding=2, return_graphs=True,
                                size_buckets=2)
    for grp, n in zip(groups, (9, 4, 6)):
        for atoms, bonds in grp:
            assert atoms.shape == (n,) and bonds.shape == (n, n) and torch.equal(bonds, bonds.T)
    groups = model.sample_batch(emb, num_nodes=torch.tensor([9, 4, 6]), num_samples_per_embedding=2, return_graphs=True,
                                balance=True)
    for grp, n in zip(groups, (9, 4, 6)):
        for atoms, bonds in grp:
            assert atoms.shape == (n,) and bonds.shape == (n, n) and torch.equal(bonds, bonds.T)
    # one reverse step captured as a HIP graph and replayed: same contract
    groups = model.sample_batch(emb, num_nodes=torch.tensor([9, 4, 6]), num_samples_per_embedding=2, return_graphs=True,
                                use_graph=True)
    for grp, n in zip(groups, (9, 4, 6)):
        for atoms, bonds in grp:
            assert atoms.shape == (n,) and bonds.shape == (n, n) and torch.equal(bonds, bonds.T)
            assert int(atoms.min()) >= 0 and int(atoms.max()) < 8 and int(bonds.max()) < 5
    # rank-sharded driver without a process group = the chunked single-device loop: same draws as sample_batch
    from mbvae_amd.sharded_sampling import sample_sharded
    sizes = torch.tensor([9, 4, 6])
    sharded = sample_sharded(model, emb.cpu(), num_nodes=sizes, num_samples_per_embedding=2, chunk=6, seed=11)
    torch.manual_seed(11)
    direct = model.sample_batch(emb, num_nodes=sizes, num_samples_per_embedding=2, return_graphs=True)
    for s_grp, d_grp in zip(sharded, direct):
        for (sa, sb_), (da, db) in zip(s_grp, d_grp):
            assert torch.equal(sa, da) and torch.equal(sb_, db)
    mask = g["mask"].bool().cuda()
    batch = {"features": {k: v.cuda() for k, v in g["feat_d"].items()}, "representations": {"smiles": ["C"] * 4},
             "G": GraphData(X=g["X"].cuda(), E=g["E"].cuda(), node_mask=mask)}
    with torch.no_grad():
        nll = model.validation_step(batch, 0)
    assert torch.isfinite(nll)


def test_elbo_terms_golden():
    """Validation-path ELBO terms of the product model (kl_prior, T * KL of the posteriors) against the reference's
    values for the same clean / noised graphs and denoiser logits (full.py:569-677 of the reference)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from mbvae_amd.containers import BatchData, GraphData, PlaceHolder, PredictionData
    g = load_golden("posterior")
    model = _tiny_full_model("pretrain_drug", g, None)
    model.eval()
    noise = {k: g["n_" + k].cuda() for k in ("t", "beta_t", "alpha_s_bar", "alpha_t_bar")}
    gd = GraphData(X=g["X"].cuda(), E=g["E"].cuda(), node_mask=g["mask"].bool().cuda(), X_t=g["X_t"].cuda(),
                   E_t=g["E_t"].cuda(), noise_params=noise)
    bd = BatchData(graph_data=gd)
    pred = PredictionData(graph_reconstruction=PlaceHolder(X=g["logits_X"].cuda(), E=g["logits_E"].cuda()))
    with torch.no_grad():
        # ~alpha_bar_T^2 ~ 1e-7 per graph: both sides hold the rounding of ~400 differences of O(1) logs (absolute bound)
        assert_close(model.kl_prior(bd), g["kl_prior"], atol=2e-6, what="kl_prior")
        assert_close(model.compute_Lt(bd, pred), g["Lt"], what="Lt")


def test_conditional_generator_batched_optimisation_matches_per_latent_loop():
    """The batched latent optimisation follows, row by row, the reference's one-latent-at-a-time Adam loop
    (validating/conditioning.py:333-391), and the generation step returns one record per molecule."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from mbvae_amd.conditioning import ConditionalDrugGenerator
    g = load_golden("common_step")
    model = _tiny_full_model("train", g, g["sd"])
    model.T = 3
    model.noise_schedule = type(model.noise_schedule)(3).cuda()
    gen = ConditionalDrugGenerator(model, max_batch_size=8, size_buckets=2)
    torch.manual_seed(1)
    targets = gen.process_targets({"FP-ESP": torch.rand(2, 48).round(), "EMB-ESM": torch.randn(2, 40), "EMB-NT": torch.randn(2, 20)})
    assert targets.shape == (2, 32)
    lat0, flags = gen.sample_initial_latents(2, 3, reference_fingerprints={"FP-Morgan": torch.rand(5, 64).round(),
                                                                            "EMB-BiomedGraph": torch.randn(5, 24)})
    assert lat0.shape == (6, 32) and sorted(flags.view(2, 3).sum(1).tolist()) == [2.0, 2.0]
    best, meta = gen.optimize_latents_for_interaction(lat0, flags, targets, n_samples_per_target=3, n_optimization_steps=12,
                                                      learning_rate=0.01)
    # reference schedule: every latent alone
    for row in range(6):
        lat = lat0[row:row + 1].clone().requires_grad_(True)
        opt = torch.optim.Adam([lat], lr=0.01)
        best_l, best_s = lat.clone(), float("-inf")
        for _ in range(12):
            opt.zero_grad()
            fused, _ = model.fusion(lat, targets[row // 3:row // 3 + 1])
            prob = torch.sigmoid(model.dti_head(fused)["Y"])
            (-prob.mean()).backward()
            opt.step()
            if prob.item() > best_s:
                best_s, best_l = prob.item(), lat.clone().detach()
        assert abs(best_s - float(meta[row, 3])) < 1e-5
        assert_close(best[row:row + 1], best_l, rtol=1e-4, what=f"optimised latent {row}")
        assert int(meta[row, 0]) == row // 3 and int(meta[row, 1]) == row % 3 and float(meta[row, 2]) == float(flags[row])
    with torch.no_grad():
        start = torch.sigmoid(model.dti_head(model.fusion(lat0, targets.repeat_interleave(3, 0))[0])["Y"]).mean()
    assert float(meta[:, 3].mean()) > float(start) - 1e-6
    res = gen.generate_molecules_batch(best, meta, n_molecules_per_latent=2, return_graphs=True)
    assert len(res) == 12 and {r["target_idx"] for r in res} == {0, 1} and all(r["molecule"] is not None for r in res)


def test_dense_from_ids_matches_to_dense_convention(ops):
    """The int8 class-id wire format expands to exactly what to_dense + encode_no_edge build from the sparse batch
    (padded pairs are class 0, diagonal zero, X rows of padding zero, nothing masked)."""
    from mbvae_amd.full import to_dense
    g = torch.Generator().manual_seed(5)
    sizes, cx, nb = [7, 3, 9, 1], 8, 4
    xs, ei, ea, bt, off = [], [], [], [], 0
    n = max(sizes)
    atom_ids = torch.full((len(sizes), n), -1, dtype=torch.int8)
    bond_ids = torch.zeros((len(sizes), n, n), dtype=torch.int8)
    for b, k in enumerate(sizes):
        a = torch.randint(0, cx, (k,), generator=g)
        atom_ids[b, :k] = a.to(torch.int8)
        xs.append(F.one_hot(a, cx).float()); bt.append(torch.full((k,), b))
        for i in range(k):
            for j in range(i + 1, k):
                if torch.rand((), generator=g) < 0.4:
                    t = int(torch.randint(1, nb + 1, (), generator=g))
                    bond_ids[b, i, j] = bond_ids[b, j, i] = t
                    ei += [[off + i, off + j], [off + j, off + i]]
                    ea += [F.one_hot(torch.tensor(t - 1), nb).float()] * 2
        off += k
    dense, mask = to_dense(torch.cat(xs), torch.tensor(ei).T.contiguous(), torch.stack(ea), torch.cat(bt))
    X, E, m = ops.dense_from_ids(atom_ids.cuda(), bond_ids.cuda(), cx, nb + 1)
    assert torch.equal(m.cpu(), mask) and torch.equal(X.cpu(), dense.X) and torch.equal(E.cpu(), dense.E)


def test_denoiser_full_size_properties():
    """Size-independent properties of the 8-layer denoiser at the benchmark widths (bs*n*n = 65 536 edge rows, so the
    persistent tall-skinny kernels run): node-permutation equivariance, symmetric / zero-diagonal / masked edge
    logits, masked node logits, and batch independence (a graph's outputs do not depend on its batch mates)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from mbvae_amd.graph_transformer import GraphTransformer
    from mbvae_amd.trainstep import CFG3, synthetic_graphs
    torch.manual_seed(3)
    net = GraphTransformer(**CFG3).cuda().eval()
    g = torch.Generator().manual_seed(11)
    bs, n = 16, 64
    sizes = torch.randint(20, n + 1, (bs,), generator=g); sizes[0] = n
    X0, E0, mask = synthetic_graphs(bs, n, "cuda", g, sizes=sizes)
    X = torch.cat((X0, torch.randn(bs, n, 8, generator=g).cuda() * mask.unsqueeze(-1)), -1)
    y = torch.randn(bs, 1037, generator=g).cuda()
    with torch.no_grad():
        out = net(X, E0, y, mask)
        assert float((out.E - out.E.transpose(1, 2)).abs().max()) == 0.0
        assert float(torch.diagonal(out.E, dim1=1, dim2=2).abs().max()) == 0.0
        pair = mask.unsqueeze(1) & mask.unsqueeze(2)
        assert float(out.E[~pair].abs().max()) == 0.0 and float(out.X[~mask].abs().max()) == 0.0
        assert bool(torch.isfinite(out.X).all() and torch.isfinite(out.E).all() and torch.isfinite(out.y).all())
        # permute the nodes of every graph (valid nodes among themselves, padding stays behind them)
        perm = torch.stack([torch.cat((torch.randperm(int(k), generator=g), torch.arange(int(k), n))) for k in sizes]).cuda()
        bi = torch.arange(bs, device="cuda").unsqueeze(1)
        Xp = X[bi, perm]
        Ep = E0[bi.unsqueeze(2), perm.unsqueeze(2), perm.unsqueeze(1)]
        outp = net(Xp, Ep, y, mask)
        assert_close(outp.X, out.X[bi, perm], what="permuted X")
        assert_close(outp.E, out.E[bi.unsqueeze(2), perm.unsqueeze(2), perm.unsqueeze(1)], what="permuted E")
        assert_close(outp.y, out.y, what="permuted y")
        # batch independence: the first 9 graphs alone (36 864 edge rows, still on the persistent kernels)
        sub = net(X[:9], E0[:9], y[:9], mask[:9])
        assert_close(sub.X, out.X[:9], what="sub-batch X"); assert_close(sub.E, out.E[:9], what="sub-batch E")
        assert_close(sub.y, out.y[:9], what="sub-batch y")


@pytest.mark.parametrize("bs,n,de,N", [(16, 64, 64, 512), (3, 9, 16, 64)])
def test_edge_input_fanout(ops, bs, n, de, N):
    """One autograd node for the three consumers of E (fused projection, pooling statistics, residual alias): values
    and the summed gradient equal the three separate paths (large case: residual gradient folded into the
    input-gradient epilogue; small case: the separate-add route)."""
    g = torch.Generator().manual_seed(bs + n)
    mask = ragged_mask(bs, n, g)
    E = torch.randn(bs, n, n, de, generator=g)
    w = torch.randn(N, de, generator=g) / math.sqrt(de); b = torch.randn(N, generator=g) * 0.1
    c1, c2, c3 = torch.randn(bs, n, n, N, generator=g), torch.randn(bs, 4 * de, generator=g), torch.randn(bs, n, n, de, generator=g)
    Er, wr, br = (t.clone().requires_grad_(True) for t in (E, w, b))
    em = (mask.unsqueeze(1) & mask.unsqueeze(2)).unsqueeze(-1).float()
    y = F.linear(Er, wr, br) * em
    flat = Er.reshape(bs, -1, de)
    stats = torch.cat((flat.mean(1), flat.min(1).values, flat.max(1).values, flat.std(1)), 1)
    ((y * c1).sum() + (stats * c2).sum() + (Er * c3).sum()).backward()
    Eg, wg, bg = (t.clone().cuda().requires_grad_(True) for t in (E, w, b))
    yg, sg, alias = ops.edge_input(Eg, wg, bg, mask.cuda(), n)
    ((yg * c1.cuda()).sum() + (sg * c2.cuda()).sum() + (alias * c3.cuda()).sum()).backward()
    assert_close(yg, y, what="projection"); assert_close(sg, stats, what="pool stats"); assert torch.equal(alias, Eg)
    assert_close(Eg.grad, Er.grad, what="dE"); assert_close(wg.grad, wr.grad, what="dw")
    assert_close(bg.grad, br.grad, what="db")


@pytest.mark.parametrize("bs,n,D", [(256, 64, 256), (3, 9, 16), (5, 33, 520)])
def test_film(ops, bs, n, D):
    """add + (mul + 1) * x over the nodes of a graph (graph_transformer.py:254): values and the three gradients against
    the torch expression; bitwise equal from run to run."""
    g = torch.Generator().manual_seed(bs + n + D)
    add, mul, x, c = (torch.randn(*sh, generator=g) for sh in ((bs, D), (bs, D), (bs, n, D), (bs, n, D)))
    ref = [t.clone().requires_grad_(True) for t in (add, mul, x)]
    yr = ref[0].unsqueeze(1) + (ref[1].unsqueeze(1) + 1) * ref[2]
    (yr * c).sum().backward()

    def run():
        ins = [t.clone().cuda().requires_grad_(True) for t in (add, mul, x)]
        y = ops.film(*ins)
        (y * c.cuda()).sum().backward()
        return y.detach(), [t.grad for t in ins]

    y, grads = run()
    assert_close(y, yr, what="film")
    for got, t, name in zip(grads, ref, ("dadd", "dmul", "dx")):
        assert_close(got, t.grad, what=name)
    y2, grads2 = run()
    assert torch.equal(y, y2) and all(torch.equal(a, b_) for a, b_ in zip(grads, grads2))


@pytest.mark.parametrize("shape,own,drop", [((256, 512), True, True), ((256, 1024), False, True), ((37, 24), True, False),
                                            ((5, 3, 7), False, False)])
def test_gate_update(ops, shape, own, drop):
    """x + u * dropout(sigmoid(z)) as one kernel each way (models/blocks.py:125, :318-322): values and every gradient against
    the torch expression in float64 with the same keep mask; u = x (self gate) and a separate u (cross gate); bitwise equal
    from run to run; in training mode the op draws a mask that keeps about 1 - p of the entries and scales them."""
    from mbvae_amd.ops import _GateUpdate
    g = torch.Generator().manual_seed(sum(shape))
    x, u, z, c = (torch.randn(*shape, generator=g) for _ in range(4))
    keep = (torch.rand(*shape, generator=g) < 0.9).to(torch.uint8) if drop else None
    scale = 1 / 0.9 if drop else 1.0
    ref = [t.double().requires_grad_(True) for t in (x, u, z)]
    gate = torch.sigmoid(ref[2]) * (keep.double() * scale if drop else 1.0)
    yr = ref[0] + (ref[0] if own else ref[1]) * gate
    (yr * c.double()).sum().backward()

    def run():
        ins = [t.clone().cuda().requires_grad_(True) for t in (x, u, z)]
        y = _GateUpdate.apply(ins[0], None if own else ins[1], ins[2], keep.cuda() if drop else None, scale)
        (y * c.cuda()).sum().backward()
        return y.detach(), [t.grad for t in ins]

    y, grads = run()
    assert_close(y, yr.float(), what="gate update")
    for got, t, name in zip(grads, ref, ("dx", "du", "dz")):
        if name == "du" and own:
            assert got is None
        else:
            assert_close(got, t.grad.float(), what=name)
    y2, grads2 = run()
    assert torch.equal(y, y2) and all(a is b_ or torch.equal(a, b_) for a, b_ in zip(grads, grads2))
    if drop:
        xs = torch.ones(*shape).cuda(); zs = torch.full(shape, 20.0).cuda()          # sigmoid(20) = 1 in fp32
        torch.manual_seed(3)
        kept = ops.gate_update(xs, None, zs, 0.25, training=True) - 1
        frac = float((kept > 0).float().mean())
        assert abs(frac - 0.75) < 0.01 and torch.all((kept == 0) | ((kept - 1 / 0.75).abs() < 1e-6))
        assert torch.equal(ops.gate_update(xs, None, zs, 0.25, training=False), xs + xs * torch.sigmoid(zs))


@pytest.mark.parametrize("M,d,dff", [(16 * 64 * 64, 64, 128), (4 * 64 * 64, 256, 512), (256, 1024, 1024), (300, 24, 40)])
def test_ffn_gate_handover(ops, M, d, dff, default_routes):
    """ops.ffn: the ReLU gate of the first Linear is applied by the second one's input-gradient call (in the product's
    epilogue at the edge widths, by an element-wise pass elsewhere); values and all gradients against plain torch, and
    against the two-Linear composition that gates inside the first layer's gradient kernels."""
    g = torch.Generator().manual_seed(M + d)
    # x, w1, b1 are small multiples of 1/8: every pre-activation is then exact in fp32 on any summation order, so the
    # GPU and the CPU reference take the same side of every ReLU gate (a flipped gate is an O(1) gradient difference)
    x = torch.randint(-8, 9, (M, d), generator=g).float() / 8
    w1 = torch.randint(-8, 9, (dff, d), generator=g).float() / 8; b1 = torch.randint(-8, 9, (dff,), generator=g).float() / 8
    w2 = torch.randn(d, dff, generator=g) / math.sqrt(dff); b2 = torch.randn(d, generator=g) * 0.1
    cy, cx = torch.randn(M, d, generator=g), torch.randn(M, d, generator=g)
    ref = [t.clone().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    hr = F.relu(F.linear(ref[0], ref[1], ref[2]))
    ((F.linear(hr, ref[3], ref[4]) * cy).sum() + (ref[0] * cx).sum()).backward()

    def run(handover):
        ins = [t.clone().cuda().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
        _lib.PROFILE = {}
        try:
            if handover:
                y, xa = ops.ffn(*ins, handover=True)
            else:
                h, xa = ops.linear(ins[0], ins[1], ins[2], act=1, with_alias=True)
                y = ops.linear(h, ins[3], ins[4])
            ((y * cy.cuda()).sum() + (xa * cx.cuda()).sum()).backward()
            torch.cuda.synchronize()
            called = {name for name, _ in _lib.PROFILE}
        finally:
            _lib.PROFILE = None
        return y.detach(), [t.grad for t in ins], called

    y, grads, called = run(True)
    assert "mbvae_linear_bwd_input_outgate" in called
    assert_close(y, F.linear(F.relu(F.linear(x, w1, b1)), w2, b2), what="ffn output")
    for got, t, name in zip(grads, ref, ("dx", "dw1", "db1", "dw2", "db2")):
        assert_close(got, t.grad, what=name)
    y0, grads0, called0 = run(False)
    assert "mbvae_linear_bwd_input_outgate" not in called0 and torch.equal(y, y0)
    for a, b_, name in zip(grads, grads0, ("dx", "dw1", "db1", "dw2", "db2")):
        assert_close(a, b_, what=name + " hand-over vs in-layer gate")


@pytest.mark.parametrize("M,p1,p2", [(8 * 64 * 64, 0.0, 0.0), (16 * 64 * 64, 0.1, 0.1), (32768 + 32 * 37, 0.25, 0.0), (40 * 32 * 32, 0.0, 0.1)])
def test_edge_ffn_training_form_one_kernel(ops, M, p1, p2, default_routes):
    """The edge tensor's feed-forward block in training (graph_transformer.py:139-142: linE1, ReLU, dropoutE2, linE2,
    dropoutE3, residual, normE2) as ONE forward kernel that stores the hidden activation and the pre-LayerNorm sum for
    the backward, and ONE backward kernel for both activation gradients (dh gated by the stored hidden activation, dx with
    the residual branch's gradient): output and all seven gradients against plain torch without dropout; with dropout against the
    two-kernel route on the same seeds (the keep-masks are hashed from the element position, so they are identical);
    bitwise equal from run to run."""
    g = torch.Generator().manual_seed(M)
    # x, w1, b1 are small multiples of 1/8: every pre-activation is exact in fp32 in any summation order, so all routes
    # and the CPU reference take the same side of every ReLU gate
    x = torch.randint(-8, 9, (M, 64), generator=g).float() / 8
    w1 = torch.randint(-8, 9, (128, 64), generator=g).float() / 8; b1 = torch.randint(-8, 9, (128,), generator=g).float() / 8
    w2 = torch.randn(64, 128, generator=g) / math.sqrt(128); b2 = torch.randn(64, generator=g) * 0.1
    gamma = 1 + 0.2 * torch.randn(64, generator=g); beta = 0.1 * torch.randn(64, generator=g)
    cot = torch.randn(M, 64, generator=g)
    names = ("dx", "dw1", "db1", "dw2", "db2", "dgamma", "dbeta")

    def run(one_kernel):
        ins = [t.clone().cuda().requires_grad_(True) for t in (x, w1, b1, w2, b2, gamma, beta)]
        _lib.PROFILE = {}
        try:
            with _lib.option("NO_EDGE_FFN_TRAIN", 0 if one_kernel else 1):
                y, xa = ops.ffn(*ins[:5], drop_p=p1, seed=777, ln=(ins[5], ins[6], 1e-5, p2, 4242))
                assert xa is None
                (y * cot.cuda()).sum().backward()
                torch.cuda.synchronize()
            called = {name for name, _ in _lib.PROFILE}
        finally:
            _lib.PROFILE = None
        return y.detach(), [t.grad for t in ins], called

    y, grads, called = run(True)
    assert "mbvae_edge_ffn_fwd_train" in called and "mbvae_linear_ln_fwd" not in called and "mbvae_linear_fwd" not in called
    assert {"mbvae_ln_sum_bwd", "mbvae_edge_ffn_bwd_data", "mbvae_linear_bwd_weight"} <= called
    assert not {"mbvae_linear_bwd_input_outgate", "mbvae_linear_bwd_input_add", "mbvae_linear_bwd_input"} & called
    if p1 == 0.0 and p2 == 0.0:
        ref = [t.clone().requires_grad_(True) for t in (x, w1, b1, w2, b2, gamma, beta)]
        yr = F.layer_norm(ref[0] + F.linear(F.relu(F.linear(ref[0], ref[1], ref[2])), ref[3], ref[4]), (64,), ref[5], ref[6], 1e-5)
        (yr * cot).sum().backward()
        assert_close(y, yr, what="ffn + LayerNorm output")
        for got, t, name in zip(grads, ref, names):
            assert_close(got, t.grad, what=name)
    y0, grads0, called0 = run(False)
    assert not {"mbvae_edge_ffn_fwd_train", "mbvae_edge_ffn_bwd_data"} & called0
    assert {"mbvae_linear_fwd", "mbvae_linear_ln_fwd", "mbvae_linear_bwd_input_outgate", "mbvae_linear_bwd_input_add"} <= called0
    assert_close(y, y0, what="one kernel vs two")
    if p1 > 0:                                          # the same hidden units were dropped: compare through db1's support
        assert torch.equal(grads[2] == 0, grads0[2] == 0)
    for a, b_, name in zip(grads, grads0, names):
        assert_close(a, b_, what=name + ": one kernel vs two")
    y2, grads2, _ = run(True)
    assert torch.equal(y, y2) and all(torch.equal(a, b_) for a, b_ in zip(grads, grads2))


@pytest.mark.parametrize("bs,n,K,masked,drop_p", [(16, 64, 256, True, 0.0), (16, 64, 128, False, 0.0), (2, 64, 256, True, 0.0),
                                                   (9, 64, 256, True, 0.1), (16, 32, 128, False, 0.1), (3, 64, 64, True, 0.25)])
def test_linear_layernorm_epilogue(ops, bs, n, K, masked, drop_p, default_routes):
    """ops.linear_ln (a 64-wide Linear with residual + dropout + LayerNorm in the product's epilogue, backward from the
    stored pre-LayerNorm sum): values and all six gradients against plain torch without dropout, and against the
    separate Linear + residual-LayerNorm kernels on the same dropout seed (identical masks) with it."""
    g = torch.Generator().manual_seed(bs * 1000 + n + K)
    M = bs * n * n
    x = torch.randn(M, K, generator=g)
    w = torch.randn(64, K, generator=g) / math.sqrt(K); b = torch.randn(64, generator=g) * 0.1
    resid = torch.randn(bs, n, n, 64, generator=g)
    gamma = 1 + 0.2 * torch.randn(64, generator=g); beta = 0.1 * torch.randn(64, generator=g)
    cot = torch.randn(bs, n, n, 64, generator=g)
    mask = ragged_mask(bs, n, g) if masked else None
    kw = dict(mask=dev(mask), mask_mode=ops.MASK_EDGE, n_nodes=n) if masked else {}
    seed = 12345

    def run(fused):
        ins = [t.clone().cuda().requires_grad_(True) for t in (x, w, b, resid, gamma, beta)]
        _lib.PROFILE = {}
        try:
            if fused:
                assert ops.linear_ln_supported(ins[0], ins[1], ins[3])
                y = ops.linear_ln(ins[0], ins[1], ins[2], ins[3], ins[4], ins[5], 1e-5, drop_p, seed, **kw)
            else:
                r = ops.linear(ins[0], ins[1], ins[2], **kw).reshape(bs, n, n, 64)
                y = ops.ln_residual(ins[3], r, ins[4], ins[5], 1e-5, drop_p, seed)
            (y * cot.cuda()).sum().backward()
            torch.cuda.synchronize()
            called = {name for name, _ in _lib.PROFILE}
        finally:
            _lib.PROFILE = None
        return y.detach(), [t.grad for t in ins], called

    names = ("dx", "dw", "db", "dresid", "dgamma", "dbeta")
    y, grads, called = run(True)
    assert {"mbvae_linear_ln_fwd", "mbvae_ln_sum_bwd"} <= called and "mbvae_ln_residual_fwd" not in called
    y0, grads0, called0 = run(False)
    assert "mbvae_linear_ln_fwd" not in called0
    assert_close(y, y0, what="fused vs separate kernels")
    for a, b_, name in zip(grads, grads0, names):
        assert_close(a, b_, what=name + " fused vs separate kernels")
    if drop_p == 0.0:
        ref = [t.clone().requires_grad_(True) for t in (x, w, b, resid, gamma, beta)]
        r = F.linear(ref[0], ref[1], ref[2]).reshape(bs, n, n, 64)
        if masked:
            r = r * (mask[:, :, None] & mask[:, None, :]).unsqueeze(-1)
        yr = F.layer_norm(ref[3] + r, (64,), ref[4], ref[5], 1e-5)
        (yr * cot).sum().backward()
        assert_close(y, yr, what="output")
        for got, t, name in zip(grads, ref, names):
            assert_close(got, t.grad, what=name)


@pytest.mark.parametrize("bs,n,with_add,ragged", [(16, 64, True, True), (9, 64, False, True), (32, 32, True, False)])
def test_projection_backward_single_pass(ops, bs, n, with_add, ragged, default_routes):
    """mbvae_linear_bwd_both (input gradient, weight gradient and bias gradient of the 64 -> 512 projection from one
    pass over dy) against an fp64 restatement and against the two separate entry points; bitwise equal run to run."""
    K, N = 64, 512
    M = bs * n * n
    g = torch.Generator().manual_seed(bs * 7 + n)
    mask = (ragged_mask(bs, n, g) if ragged else torch.ones(bs, n, dtype=torch.bool)).cuda()
    dy = torch.randn(M, N, generator=g).cuda(); x = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) / 8).cuda()
    add = torch.randn(M, K, generator=g).cuda() if with_add else None
    m8 = mask.view(torch.uint8)
    ws = ops.workspace(dy.device)
    st = torch.cuda.current_stream().cuda_stream

    def both():
        dx = torch.empty(M, K, device="cuda"); dw = torch.empty(N, K, device="cuda"); db = torch.empty(N, device="cuda")
        _lib.call("mbvae_linear_bwd_both", dy.data_ptr(), N, w.data_ptr(), x.data_ptr(), K, add.data_ptr() if with_add else None,
                  dx.data_ptr(), K, dw.data_ptr(), db.data_ptr(), M, K, N, m8.data_ptr(), ops.MASK_EDGE, n, ws.data_ptr(), ws.numel(), st)
        return dx, dw, db

    dx, dw, db = both()
    keep = (mask.unsqueeze(1) & mask.unsqueeze(2)).reshape(M, 1).double()
    dyk = dy.double() * keep
    dx_ref = dyk @ w.double() + (add.double() if with_add else 0.0)
    assert_close(dx, dx_ref.float(), what="dx"); assert_close(dw, (dyk.t() @ x.double()).float(), what="dw")
    assert_close(db, dyk.sum(0).float(), what="db")
    dx2, dw2, db2 = both()
    assert torch.equal(dx, dx2) and torch.equal(dw, dw2) and torch.equal(db, db2), "not reproducible"
    # the separate kernels
    dxs = torch.empty(M, K, device="cuda"); dws = torch.empty(N, K, device="cuda"); dbs = torch.empty(N, device="cuda")
    if with_add:
        _lib.call("mbvae_linear_bwd_input_add", dy.data_ptr(), N, w.data_ptr(), add.data_ptr(), dxs.data_ptr(), K, M, K, N, 0, 1.0,
                  m8.data_ptr(), ops.MASK_EDGE, n, st)
    else:
        _lib.call("mbvae_linear_bwd_input", dy.data_ptr(), N, w.data_ptr(), dxs.data_ptr(), K, M, K, N, 0, 1.0,
                  m8.data_ptr(), ops.MASK_EDGE, n, 0, ws.data_ptr(), ws.numel(), st)
    _lib.call("mbvae_linear_bwd_weight", dy.data_ptr(), N, x.data_ptr(), K, dws.data_ptr(), dbs.data_ptr(), M, K, N, 0, 1.0,
              m8.data_ptr(), ops.MASK_EDGE, n, 0, ws.data_ptr(), ws.numel(), st)
    assert_close(dx, dxs, what="dx vs separate"); assert_close(dw, dws, what="dw vs separate"); assert_close(db, dbs, what="db vs separate")
    with _lib.option("NO_BWD_BOTH", 1):
        assert not _lib.call("mbvae_linear_bwd_both", dy.data_ptr(), N, w.data_ptr(), x.data_ptr(), K, None, dx.data_ptr(), K,
                             dw.data_ptr(), db.data_ptr(), M, K, N, m8.data_ptr(), ops.MASK_EDGE, n, ws.data_ptr(), ws.numel(), st,
                             allow_unsupported=True)


def _edge_block_reference(E, w12, b12, QKV, ya, ym, mask, n_head):
    """Plain fp32 torch restatement of graph_transformer.py:196-249 for the fused edge block (CPU)."""
    bs, n, dx3 = QKV.shape
    dx = dx3 // 3
    em = (mask.unsqueeze(1) & mask.unsqueeze(2)).unsqueeze(-1).float()
    E12 = F.linear(E, w12, b12) * em
    E1, E2 = E12[..., :dx], E12[..., dx:]
    Q, K, V = QKV[..., :dx], QKV[..., dx:2 * dx], QKV[..., 2 * dx:]
    Y = Q.unsqueeze(2) * K.unsqueeze(1) / math.sqrt(dx // n_head)
    Y = Y * (E1 + 1) + E2
    P = ya[:, None, None, :] + (ym[:, None, None, :] + 1) * Y
    key = mask[:, None, :, None].expand(-1, n, -1, dx)
    A = torch.softmax(Y.masked_fill(~key, float("-inf")), dim=2)
    wV = (A * V.unsqueeze(1)).sum(2)
    flat = E.reshape(bs, -1, E.shape[-1])
    stats = torch.cat((flat.mean(1), flat.min(1).values, flat.max(1).values, flat.std(1)), 1)
    return P, wV, stats


@pytest.mark.parametrize("bs,n,sizes", [(3, 64, None), (4, 37, (37, 20, 5, 33)), (2, 20, (20, 11)), (2, 64, (64, 31)),
                                        (1, 2, None), (5, 33, (33, 1, 32, 2, 17)), (70, 8, None)])
def test_edge_block_fused_forward(ops, bs, n, sizes, monkeypatch):
    """Projection-inside-attention forward kernel (de = 64, dx = 256): values and all gradients against a plain torch
    restatement, and identical behaviour to the two separate forward kernels, on full, ragged and single-tile graphs."""
    de, dx, H = 64, 256, 8
    g = torch.Generator().manual_seed(7 * bs + n)
    sz = torch.full((bs,), n) if sizes is None else torch.tensor(sizes)
    mask = torch.arange(n)[None, :] < sz[:, None]
    xm = mask.unsqueeze(-1).float()
    E = torch.randn(bs, n, n, de, generator=g)
    w = torch.randn(2 * dx, de, generator=g) / math.sqrt(de); b = torch.randn(2 * dx, generator=g) * 0.1
    QKV = torch.randn(bs, n, 3 * dx, generator=g) * xm
    ya, ym = torch.randn(bs, dx, generator=g) * 0.3, torch.randn(bs, dx, generator=g) * 0.3
    cP, cW = torch.randn(bs, n, n, dx, generator=g), torch.randn(bs, n, dx, generator=g) * xm
    cS, cA = torch.randn(bs, 4 * de, generator=g), torch.randn(bs, n, n, de, generator=g)
    ref_in = [t.clone().requires_grad_(True) for t in (E, w, b, QKV, ya, ym)]
    Pr, Wr, Sr = _edge_block_reference(*ref_in, mask, H)
    ((Pr * cP).sum() + (Wr * cW).sum() + (Sr * cS).sum() + (ref_in[0] * cA).sum()).backward()

    def run(grad):
        ins = [t.clone().cuda().requires_grad_(grad) for t in (E, w, b, QKV, ya, ym)]
        P, wV, stats, alias = ops.edge_block(*ins, mask.cuda(), H)
        if grad:
            ((P * cP.cuda()).sum() + (wV * cW.cuda()).sum() + (stats * cS.cuda()).sum() + (alias * cA.cuda()).sum()).backward()
        return P, wV, stats, [t.grad for t in ins]

    P, wV, stats, grads = run(True)
    vm = xm.bool().expand(-1, -1, dx)
    assert_close(P, Pr, what="P"); assert_close(wV * xm.cuda(), Wr * xm, what="wV"); assert_close(stats, Sr, what="stats")
    for got, t, name in zip(grads, ref_in, ("dE", "dw12", "db12", "dQKV", "dye_add", "dye_mul")):
        assert_close(got, t.grad, what=name)
    P0, wV0, _, _ = run(False)        # inference: no projection output is written at all, P only on valid pairs
    pm = (mask.unsqueeze(1) & mask.unsqueeze(2)).cuda()
    assert torch.equal(P0[pm], P[pm]) and torch.equal(wV0 * xm.cuda(), wV * xm.cuda())
    with _lib.option("NO_EDGE_FUSED", 1):            # the two separate forward kernels
        P1, wV1, stats1, grads1 = run(True)
    assert_close(P1, P, rtol=2e-6, what="P fused vs separate"); assert_close(wV1 * xm.cuda(), wV * xm.cuda(), rtol=2e-6, what="wV fused vs separate")
    for a, b_ in zip(grads1, grads):
        assert_close(a, b_, rtol=1e-5, what="grads fused vs separate")


@pytest.mark.parametrize("bs,n,ragged", [(128, 64, True), (160, 16, False), (130, 33, True), (128, 40, True)])
def test_edge_block_recompute_backward(ops, bs, n, ragged, default_routes):
    """Batches of at least one graph per two compute units train through the pair that never stores E1|E2: the forward
    writes P / wV / lse only and the backward forms the projection again from E.  Values and all gradients against the
    plain torch restatement; bitwise equal from run to run; and the storing pair gives the same numbers."""
    de, dx, H = 64, 256, 8
    g = torch.Generator().manual_seed(11 * bs + n)
    sz = torch.randint(1, n + 1, (bs,), generator=g) if ragged else torch.full((bs,), n)
    sz[0] = n
    mask = torch.arange(n)[None, :] < sz[:, None]
    xm = mask.unsqueeze(-1).float()
    E = torch.randn(bs, n, n, de, generator=g)
    w = torch.randn(2 * dx, de, generator=g) / math.sqrt(de); b = torch.randn(2 * dx, generator=g) * 0.1
    QKV = torch.randn(bs, n, 3 * dx, generator=g) * xm
    ya, ym = torch.randn(bs, dx, generator=g) * 0.3, torch.randn(bs, dx, generator=g) * 0.3
    pm = (mask.unsqueeze(1) & mask.unsqueeze(2)).unsqueeze(-1).float()
    cP, cW = torch.randn(bs, n, n, dx, generator=g) * pm, torch.randn(bs, n, dx, generator=g) * xm   # (e_out masks dP likewise)
    cS, cA = torch.randn(bs, 4 * de, generator=g), torch.randn(bs, n, n, de, generator=g)
    ref_in = [t.clone().requires_grad_(True) for t in (E, w, b, QKV, ya, ym)]
    Pr, Wr, Sr = _edge_block_reference(*ref_in, mask, H)
    ((Pr * cP).sum() + (Wr * cW).sum() + (Sr * cS).sum() + (ref_in[0] * cA).sum()).backward()

    def run():
        ins = [t.clone().cuda().requires_grad_(True) for t in (E, w, b, QKV, ya, ym)]
        _lib.PROFILE = {}
        try:
            P, wV, stats, alias = ops.edge_block(*ins, mask.cuda(), H)
            ((P * cP.cuda()).sum() + (wV * cW.cuda()).sum() + (stats * cS.cuda()).sum() + (alias * cA.cuda()).sum()).backward()
            torch.cuda.synchronize()
            called = {name for name, _ in _lib.PROFILE}
        finally:
            _lib.PROFILE = None
        return P.detach(), wV.detach(), [t.grad for t in ins], called

    P, wV, grads, called = run()
    assert {"mbvae_edge_fused_fwd_recompute", "mbvae_edge_fused_bwd"} <= called and "mbvae_edge_attn_bwd" not in called
    assert_close(P, Pr, what="P"); assert_close(wV * xm.cuda(), Wr * xm, what="wV")
    names = ("dE", "dw12", "db12", "dQKV", "dye_add", "dye_mul")
    for got, t, name in zip(grads, ref_in, names):
        assert_close(got, t.grad, what=name)
    P2, wV2, grads2, _ = run()
    assert torch.equal(P, P2) and torch.equal(wV, wV2) and all(torch.equal(a, b_) for a, b_ in zip(grads, grads2)), "not reproducible"
    with _lib.option("NO_E12_RECOMPUTE", 1):
        P1, wV1, grads1, called1 = run()
    assert "mbvae_edge_attn_bwd" in called1 and "mbvae_edge_fused_bwd" not in called1
    assert torch.equal(P1, P) and torch.equal(wV1, wV)          # the same forward kernel, minus the E1|E2 stores
    for a, b_, name in zip(grads1, grads, names):
        assert_close(a, b_, what=name + " storing vs recomputing")


def test_train_step_gradients_are_bitwise_reproducible():
    """Two runs of the same denoiser step (same weights, data, dropout seeds) at the benchmark widths give bit-identical
    losses and gradients: no kernel on the path accumulates with floating-point atomics (the attention backward sums
    its per-key-group partials in a fixed order)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from mbvae_amd import ops as O
    from mbvae_amd.graph_transformer import GraphTransformer
    from mbvae_amd.trainstep import CFG3, synthetic_graphs
    cfg = dict(CFG3); cfg["n_layers"] = 2
    torch.manual_seed(3)
    net = GraphTransformer(**cfg).cuda().train()
    g = torch.Generator().manual_seed(5)
    bs, n = 16, 64
    sizes = torch.randint(20, n + 1, (bs,), generator=g); sizes[0] = n
    X0, E0, mask = synthetic_graphs(bs, n, "cuda", g, sizes=sizes)
    Xin = torch.cat((X0, torch.randn(bs, n, 8, generator=g).cuda() * mask.unsqueeze(-1)), 2)
    yin = torch.randn(bs, 1037, generator=g).cuda()
    runs = []
    for _ in range(2):
        net.zero_grad(set_to_none=True)
        torch.manual_seed(11)                            # same dropout seeds
        out = net(Xin, E0, yin, mask)
        lX, _ = O.masked_ce(out.X, X0); lE, _ = O.masked_ce(out.E, E0)
        loss = lX + 5 * lE + out.y.pow(2).mean()
        loss.backward()
        runs.append((loss.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters()}))
    assert torch.equal(runs[0][0], runs[1][0])
    for k in runs[0][1]:
        assert torch.equal(runs[0][1][k], runs[1][1][k]), f"gradient of {k} differs between two identical runs"

@pytest.mark.parametrize("rows", [32, 4 * 37 * 37 // 32 * 32, 3 * 64 * 64, 5, 3 * 37 * 37, 32 * 2048 * 3 + 7])
def test_edge_ffn_inference_fused(ops, rows, monkeypatch):
    """Inference-only fused edge feed-forward + LayerNorm (64 -> 128 -> 64): against plain torch and against the
    three separate kernels."""
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, 64, generator=g) * 1.5
    w1, b1 = torch.randn(128, 64, generator=g) / 8, torch.randn(128, generator=g) * 0.1
    w2, b2 = torch.randn(64, 128, generator=g) / 11, torch.randn(64, generator=g) * 0.1
    gamma, beta = 1 + 0.2 * torch.randn(64, generator=g), 0.1 * torch.randn(64, generator=g)
    ref = F.layer_norm(x + F.linear(F.relu(F.linear(x, w1, b1)), w2, b2), (64,), gamma, beta, 1e-5)
    dv = [t.cuda() for t in (x, w1, b1, w2, b2, gamma, beta)]
    got = ops.ffn_ln_inference(*dv, 1e-5)
    assert_close(got, ref, what="fused ffn + LN")
    # padding tiles: every row of a tile marked 0 holds the same values; the result must not change by a bit
    if rows >= 64:
        tiles = (rows + 31) // 32
        tm = (torch.rand(tiles, generator=g) < 0.5).to(torch.uint8)
        tm[0], tm[-1] = 1, 0
        xp = torch.cat((x, x[:tiles * 32 - rows])).view(tiles, 32, 64).clone()
        xp[tm == 0] = x[5]
        xp = xp.view(-1, 64)[:rows].contiguous().cuda()
        full = ops.ffn_ln_inference(xp, *dv[1:], 1e-5)
        skip = ops.ffn_ln_inference(xp, *dv[1:], 1e-5, tile_map=tm.cuda())
        assert torch.equal(full, skip), "padding-tile shortcut changed the result"
        # packed row order: the valid tiles' rows first, then the (uniform) padding rows
        class _Ord:
            pass
        rowflag = tm.repeat_interleave(32)[:rows].bool()
        valid_rows, pad_rows = rowflag.nonzero().flatten(), (~rowflag).nonzero().flatten()
        nv = valid_rows.numel()
        head = torch.cat((valid_rows, valid_rows[-1:].repeat(-nv % 32)))
        tail = torch.cat((pad_rows, pad_rows[-1:].repeat(-pad_rows.numel() % 32 + 64)))
        o = _Ord(); o.rows = rows
        o.idx = torch.cat((head, tail)).to(torch.int32).cuda(); o.nvalid = torch.tensor([nv, 0], dtype=torch.int32).cuda()
        packed = ops.ffn_ln_inference(xp, *dv[1:], 1e-5, order=o)
        assert torch.equal(full, packed), "packed row order changed the result"
        # sparse form: the valid rows and the first padding tile are written, nothing else is
        o.nvalid = torch.tensor([nv, 1], dtype=torch.int32).cuda()
        canvas = torch.full_like(full, 7.0)
        import mbvae_amd._lib as L
        assert L.call("mbvae_edge_ffn_fwd", xp.data_ptr(), *(t.data_ptr() for t in dv[1:5]), dv[5].data_ptr(), dv[6].data_ptr(),
                      canvas.data_ptr(), 0, o.idx.data_ptr(), o.nvalid.data_ptr(), rows, 64, 128, 1e-5,
                      torch.cuda.current_stream().cuda_stream)
        written = torch.zeros(rows, dtype=torch.bool)
        written[valid_rows] = True
        written[tail[:32].long()] = True
        assert torch.equal(canvas[written.cuda()], full[written.cuda()])
        assert bool((canvas[~written.cuda()] == 7.0).all()), "the sparse order wrote padding rows beyond its first padding tile"
        skip0 = ops.ffn_ln_inference(xp[:64].contiguous().fill_(0.25), *dv[1:], 1e-5, tile_map=torch.zeros(2, dtype=torch.uint8).cuda())
        assert torch.equal(skip0, ops.ffn_ln_inference(torch.full((64, 64), 0.25).cuda(), *dv[1:], 1e-5))
    with _lib.option("NO_EDGE_FFN", 1):
        sep = ops.ffn_ln_inference(*dv, 1e-5)
    assert_close(sep, ref, what="separate kernels")
    assert_close(got, sep, rtol=3e-6, what="fused vs separate")


@pytest.mark.parametrize("bs,n,sizes", [(2, 64, None), (2, 32, (32, 9)), (8, 20, (20, 3, 11, 20, 1, 7, 19, 2)),
                                        (3, 9, (9, 4, 7)), (5, 57, (57, 20, 8, 33, 41)), (1, 3, None)])
def test_edge_out_ln_inference_fused(ops, bs, n, sizes, monkeypatch):
    """Inference-only e_out + residual + LayerNorm (256 -> 64, edge-masked projection): against plain torch and the
    separate kernels."""
    g = torch.Generator().manual_seed(bs * n)
    sz = torch.full((bs,), n) if sizes is None else torch.tensor(sizes)
    mask = torch.arange(n)[None, :] < sz[:, None]
    em = (mask.unsqueeze(1) & mask.unsqueeze(2)).unsqueeze(-1).float()
    P = torch.randn(bs, n, n, 256, generator=g)
    E = torch.randn(bs, n, n, 64, generator=g) * 1.3
    w, b = torch.randn(64, 256, generator=g) / 16, torch.randn(64, generator=g) * 0.1
    gamma, beta = 1 + 0.2 * torch.randn(64, generator=g), 0.1 * torch.randn(64, generator=g)
    ref = F.layer_norm(E + F.linear(P, w, b) * em, (64,), gamma, beta, 1e-5)
    args = (P.cuda(), w.cuda(), b.cuda(), E.cuda(), mask.cuda(), n, gamma.cuda(), beta.cuda(), 1e-5)
    got = ops.edge_out_ln_inference(*args)
    assert_close(got, ref, what="fused e_out + LN")
    # P of masked pairs is never read: poison it, and let whole padding tiles go unfetched
    Pn = P.clone()
    Pn[~em.bool().expand_as(P)] = float("nan")
    tm = ops.edge_tile_map(mask.cuda())
    flat = em.reshape(-1).bool()
    flat = torch.cat((flat, flat.new_zeros(-flat.numel() % 32)))
    assert int(tm.sum()) == int(flat.view(-1, 32).any(1).sum())
    got2 = ops.edge_out_ln_inference(Pn.cuda(), *args[1:], tile_map=tm)
    assert torch.equal(got2, got), "masked pairs of P leaked into the result"
    if bs * n * n >= 32:
        order = ops.edge_row_order(mask.cuda())
        assert int(order.nvalid[0]) == int(em.sum()) and order.idx.numel() >= bs * n * n + 64
        got3 = ops.edge_out_ln_inference(Pn.cuda(), *args[1:], order=order)
        assert torch.equal(got3, got), "packed row order changed the result"
        # uniform padding rows of E: one is normalised per wave, the others receive a copy
        Eu = torch.where(em.bool(), E, E[0, 0, 0].expand_as(E)).contiguous()
        au = (args[0], args[1], args[2], Eu.cuda()) + args[4:]
        base = ops.edge_out_ln_inference(*au)
        for kw in (dict(tile_map=tm), dict(order=order)):
            gu = ops.edge_out_ln_inference(Pn.cuda(), *au[1:], uniform_padding=True, **kw)
            assert torch.equal(gu, base), f"uniform-padding shortcut changed the result ({list(kw)})"
        # sparse order: valid rows + the first padding tile only; the pair pooling fills in the rest analytically
        sp = ops.edge_row_order(mask.cuda(), sparse=True)
        gs = ops.edge_out_ln_inference(Pn.cuda(), *au[1:], uniform_padding=True, order=sp)
        nv = int(sp.nvalid[0])
        written = em.reshape(-1).bool().clone()
        if nv < written.numel():
            written[sp.idx[(nv + 31) // 32 * 32:(nv + 31) // 32 * 32 + 32].long().cpu()] = True
        assert torch.equal(gs.reshape(-1, 64)[written.cuda()], base.reshape(-1, 64)[written.cuda()])
        if bs * n * n >= 64 and n <= 64:
            dense = ops.pool_stats(base)
            ragged = ops.pool_stats_pairs(gs, sp)
            assert_close(ragged, dense, rtol=1e-5, what="pair pooling with analytic padding")
    with _lib.option("NO_EDGE_FFN", 1):
        sep = ops.edge_out_ln_inference(*args)
        sep2 = ops.edge_out_ln_inference(Pn.cuda(), *args[1:])
    assert torch.equal(sep2, sep), "separate kernels: masked pairs of P leaked into the result"
    assert_close(sep, ref, what="separate kernels")
    assert_close(got, sep, rtol=3e-6, what="fused vs separate")


def test_denoiser_inference_padding_shortcuts_are_exact():
    """Ragged batch without gradients: the denoiser's outputs with the padding-tile shortcuts of the fused inference
    kernels equal, bit for bit, the outputs with the map switched off, and match the gradient-mode forward (separate
    training kernels, everything evaluated) to rounding."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from mbvae_amd import ops as O
    from mbvae_amd.graph_transformer import GraphTransformer
    from mbvae_amd.trainstep import CFG3, synthetic_graphs
    cfg = dict(CFG3); cfg["n_layers"] = 3
    torch.manual_seed(3)
    net = GraphTransformer(**cfg).cuda().eval()
    g = torch.Generator().manual_seed(5)
    bs, n = 12, 40
    sizes = torch.tensor([40, 7, 33, 12, 1, 25, 40, 18, 9, 31, 2, 16])
    X0, E0, mask = synthetic_graphs(bs, n, "cuda", g, sizes=sizes)
    Xin = torch.cat((X0, torch.randn(bs, n, 8, generator=g).cuda() * mask.unsqueeze(-1)), 2)
    yin = torch.randn(bs, 1037, generator=g).cuda()
    E0 = E0 * (mask.unsqueeze(1) & mask.unsqueeze(2)).unsqueeze(-1)
    with torch.no_grad():
        a = net(Xin, E0, yin, mask)
        real = O.edge_tile_map
        try:
            O.edge_tile_map = lambda m: None
            net.__dict__.pop("_pad_cache")           # first sight of the mask again, and no tile map this time
            b = net(Xin, E0, yin, mask)
        finally:
            O.edge_tile_map = real
            net.__dict__.pop("_pad_cache")
    assert real(mask) is not None and int(real(mask).sum()) < real(mask).numel()
    assert torch.equal(a.X, b.X) and torch.equal(a.E, b.E) and torch.equal(a.y, b.y)
    with torch.no_grad():                            # second sight of the same mask tensor: packed row order
        a1 = net(Xin, E0, yin, mask)
        assert net._pad_cache["order"] is None and net._pad_cache["tile_map"] is not None
        a2 = net(Xin, E0, yin, mask)
    assert torch.equal(a.E, a1.E)
    assert net._pad_cache["order"] is not None and net._pad_cache["order"].sparse and net._pad_cache["pack"] is not None
    # the sparse order pools the padding pairs analytically: same mathematics, different rounding
    assert_close(a2.X, a.X, what="X (sparse order)"); assert_close(a2.E, a.E, what="E (sparse order)")
    assert_close(a2.y, a.y, what="y (sparse order)")
    c = net(Xin, E0, yin, mask)                      # gradients enabled: training-form kernels
    assert_close(a.X, c.X.detach(), what="X"); assert_close(a.E, c.E.detach(), what="E")
    assert_close(a.y, c.y.detach(), what="y")


def test_packed_node_rows(ops):
    """NodePack: packed node pooling equals the dense pooling (padding nodes = copies of one row); masks that do not
    keep a prefix of the nodes get no pack and the denoiser still matches its gradient-mode forward."""
    g = torch.Generator().manual_seed(11)
    bs, n, D = 6, 23, 256
    sizes = torch.tensor([23, 1, 12, 23, 7, 2])
    mask = (torch.arange(n)[None, :] < sizes[:, None]).cuda()
    pack = ops.NodePack.build(mask)
    assert pack is not None and pack.n_valid == int(sizes.sum()) and pack.rows.numel() == pack.n_valid + 1
    assert pack.off.tolist() == [0] + sizes.cumsum(0).tolist() and int(pack.keep.sum()) == pack.n_valid
    X = torch.randn(bs, n, D, generator=g).cuda()
    X[~mask] = X[1, 5].clone()                           # all padding nodes hold one row
    dense = ops.pool_stats(X)
    packed = ops.pool_stats_nodes(X.reshape(-1, D)[pack.rows].contiguous(), pack)
    assert_close(packed, dense, rtol=1e-5, what="packed node pooling")
    full = torch.ones(3, 9, dtype=torch.bool).cuda()      # no padding at all: no padding row either
    pf = ops.NodePack.build(full)
    assert pf.rows.numel() == 27 and pf.n_valid == 27
    Xf = torch.randn(3, 9, 64, generator=g).cuda()
    assert_close(ops.pool_stats_nodes(Xf.reshape(-1, 64), pf), ops.pool_stats(Xf), rtol=1e-5, what="packed pooling, full graphs")
    holes = mask.clone(); holes[0, 3] = False
    assert ops.NodePack.build(holes) is None
    from mbvae_amd.graph_transformer import GraphTransformer
    from mbvae_amd.trainstep import CFG3, synthetic_graphs
    cfg = dict(CFG3); cfg["n_layers"] = 2
    torch.manual_seed(4)
    net = GraphTransformer(**cfg).cuda().eval()
    X0, E0, _ = synthetic_graphs(bs, n, "cuda", g)
    hm = holes
    Xin = torch.cat((X0, torch.randn(bs, n, 8, generator=g).cuda()), 2) * hm.unsqueeze(-1)
    E0 = E0 * (hm.unsqueeze(1) & hm.unsqueeze(2)).unsqueeze(-1)
    yin = torch.randn(bs, 1037, generator=g).cuda()
    with torch.no_grad():
        net(Xin, E0, yin, hm)
        b = net(Xin, E0, yin, hm)                        # second sight: sparse row order, but no node pack
    assert net._pad_cache["order"] is not None and net._pad_cache["pack"] is None
    c = net(Xin, E0, yin, hm)
    # more than 64 nodes per graph: the fused attention kernel does not take the shape, so no node pack either
    big = (torch.arange(70)[None, :] < torch.tensor([[70], [31]])).cuda()
    Xb, Eb, _ = synthetic_graphs(2, 70, "cuda", g)
    Xb = torch.cat((Xb, torch.randn(2, 70, 8, generator=g).cuda()), 2) * big.unsqueeze(-1)
    Eb = Eb * (big.unsqueeze(1) & big.unsqueeze(2)).unsqueeze(-1)
    with torch.no_grad():
        net(Xb, Eb, yin[:2], big)
        b2 = net(Xb, Eb, yin[:2], big)
    assert net._pad_cache["order"] is not None and net._pad_cache["pack"] is None
    c2 = net(Xb, Eb, yin[:2], big)
    assert_close(b2.X, c2.X.detach(), what="X, n = 70"); assert_close(b2.E, c2.E.detach(), what="E, n = 70")
    assert_close(b.X, c.X.detach(), what="X"); assert_close(b.E, c.E.detach(), what="E")
    assert_close(b.y, c.y.detach(), what="y")


def test_sample_batch_full_width_packed_path(monkeypatch):
    """sample_batch at the reference widths (the packed inference path: sparse row order, packed node rows, optional
    HIP graph): well-formed molecules, and the same molecules as the dense evaluation of every padded row for the
    same random draws (a draw could only differ where two classes are within rounding of a tie)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from mbvae_amd.full import FullDTIModel
    from mbvae_amd.trainstep import ATOM_WEIGHTS, CFG3, ME, MX, VALENCIES
    cfg = dict(CFG3); cfg["n_layers"] = 2
    infos = {"general": {"atom_types": ["C", "O", "P", "N", "S"], "max_n_nodes": 64, "atom_valencies": list(VALENCIES),
                         "atom_weights": list(ATOM_WEIGHTS), "max_weight": 1000},
             "dataset": {"node_count_distribution": {"9": 1, "20": 1}, "node_marginals": list(MX), "edge_marginals": list(ME)}}
    torch.manual_seed(0)
    model = FullDTIModel(phase="pretrain_drug", finetune_score=None, learning_rate=5e-4, weight_decay=1e-4, scheduler="const",
                         weights=[1.0, 0.001, 0.0, 3.0], dti_weights=None, diff_weights=[1, 5], contrastive_temp=0.1,
                         drug_features={"FP-Morgan": 64, "EMB-BiomedGraph": 32}, target_features={}, embedding_dim=1024,
                         hidden_dim=32, factor=2, n_layers=1, activation="relu", dropout=0.1, bias=True,
                         encoder_type="transformer", aggregator_type="attentive", diffusion_steps=4, sample_every_val=5,
                         val_samples_per_embedding=2, test_samples_per_embedding=2, graph_transformer_kwargs=cfg,
                         dataset_infos=infos).cuda().eval()
    emb = torch.randn(7, 1024, device="cuda")
    sizes = torch.tensor([30, 9, 17, 33, 12, 5, 21])

    def run(**kw):
        torch.manual_seed(123)
        return model.sample_batch(emb, num_nodes=sizes, num_samples_per_embedding=2, return_graphs=True, **kw)

    packed = run()
    assert model.drug_decoder._pad_cache["pack"] is not None and model.drug_decoder._pad_cache["order"].sparse
    for grp, n in zip(packed, sizes.tolist()):
        for atoms, bonds in grp:
            assert atoms.shape == (n,) and bonds.shape == (n, n) and torch.equal(bonds, bonds.T)
            assert int(atoms.min()) >= 0 and int(atoms.max()) < 5 and int(bonds.max()) < 5 and int(bonds.diagonal().abs().sum()) == 0
    # One reverse step from the same state on the same draws, packed path vs dense evaluation of every padded row: the
    # class probabilities agree to rounding, and a sampled class may differ only where the two best ratios p / Exp(1) are
    # within rounding of a tie (a whole trajectory cannot be compared that way: after one such flip it legitimately differs).
    from mbvae_amd.diffusion import exp_draws, sample_discrete_feature_noise
    bs, n = 14, int(sizes.max())
    nn_ = sizes.repeat_interleave(2).cuda()
    mask = torch.arange(n, device="cuda").unsqueeze(0) < nn_.unsqueeze(1)
    torch.manual_seed(7)
    zT = sample_discrete_feature_noise(model.limit_dist, mask)
    eX, eE, lcc = exp_draws(bs * n, 5, "cuda"), exp_draws(bs * n * n, 5, "cuda"), torch.randn(bs, n, device="cuda")
    t = torch.full((bs, 1), 3 / 4, device="cuda"); s = torch.full((bs, 1), 2 / 4, device="cuda")
    emb2 = emb.repeat_interleave(2, 0)

    def one_step():
        seen = {}
        real = O.reverse_sample

        def spy(*a, **k):
            out = real(*a, want_probs=True, **k)
            seen["cls"], seen["prob"] = (out[0], out[1]), (out[4], out[5])
            return out[:4]

        O.reverse_sample = spy
        try:
            model.replay = {"expX": eX, "expE": eE, "lcc_noise": lcc}
            with torch.no_grad():
                model.sample_p_zs_given_zt(s, t, zT.X, zT.E, emb2, mask)
        finally:
            O.reverse_sample = real
            model.replay = None
        return seen

    from mbvae_amd import ops as O
    model.drug_decoder.prepare_padding_shortcuts(mask)
    a = one_step()
    assert model.drug_decoder._pad_cache["pack"] is not None
    monkeypatch.setenv("MBVAE_NO_ROW_ORDER", "1")
    model.drug_decoder.__dict__.pop("_pad_cache")
    b = one_step()
    monkeypatch.delenv("MBVAE_NO_ROW_ORDER")
    pair = (mask.unsqueeze(1) & mask.unsqueeze(2))
    for k, valid, expo in ((0, mask, eX.reshape(bs, n, 5)), (1, pair, eE.reshape(bs, n, n, 5))):
        # probabilities of valid rows: the sparse order pools padding pairs analytically (same mathematics, other rounding)
        assert_close(a["prob"][k][valid], b["prob"][k][valid], what="step probabilities, packed vs dense")
        assert _flip_ok(b["prob"][k][valid], a["cls"][k][valid], b["cls"][k][valid], expo[valid]), "a sample differs away from a tie"
    for kw in (dict(use_graph=True), dict(balance=True, use_graph=True), dict(size_buckets=2)):
        for grp, n in zip(run(**kw), sizes.tolist()):
            for atoms, bonds in grp:
                assert atoms.shape == (n,) and bonds.shape == (n, n) and torch.equal(bonds, bonds.T)
                assert int(atoms.min()) >= 0 and int(atoms.max()) < 5 and int(bonds.max()) < 5
