"""Opt-in bf16 STORAGE of the edge block's large activations (E1|E2, P and their gradients): fp32 weights, fp32
arithmetic, bf16 in memory.  The fp32 HIP path (itself held to 1e-5 of the oracle / the reference's goldens elsewhere
in this suite) is the comparison; the bound is the 2e-2 the reduced-precision mode is specified to, relative to each
tensor's own scale, and is stated per check."""
import math
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mb-vae-dti_amd"))
sys.path.insert(0, ROOT)
from tests._util import assert_close  # noqa: E402

pytestmark = pytest.mark.gpu
BF16_RTOL = 2e-2


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
    with _lib.option("NO_SPLIT_MFMA", 0), _lib.option("NO_E12_RECOMPUTE", 0):
        yield


def _ragged(bs, n, g):
    sizes = torch.randint(n // 2, n + 1, (bs,), generator=g)
    sizes[0] = n
    return (torch.arange(n)[None, :] < sizes[:, None]).cuda()


def test_bf16_linear_entry_points_match_fp32_on_bf16_representable_inputs(ops):
    """With inputs that ARE bf16 values the bf16-reading kernels must reproduce the fp32 kernels to fp32 accuracy
    (1e-5); a bf16 output differs from the fp32 one by its own rounding only (2^-9 relative per element)."""
    from mbvae_amd import _lib
    g = torch.Generator().manual_seed(3)
    bs, n = 16, 64
    M = bs * n * n
    mask = _ragged(bs, n, g)
    m8 = mask.view(torch.uint8)
    st = torch.cuda.current_stream().cuda_stream
    ws = ops.workspace(mask.device)
    keep = (mask.unsqueeze(1) & mask.unsqueeze(2)).reshape(M, 1)

    def rnd(*shape):
        return torch.randn(*shape, generator=g).cuda()

    # e_out forward: x [M,256] bf16 -> y [M,64]
    x = rnd(M, 256).bfloat16(); w = rnd(64, 256) / 16; b = rnd(64) * 0.1
    y = torch.empty(M, 64, device="cuda")
    _lib.call("mbvae_linear_fwd_bf16", x.data_ptr(), 256, w.data_ptr(), b.data_ptr(), y.data_ptr(), 64, M, 256, 64,
              m8.data_ptr(), ops.MASK_EDGE, n, st)
    ref = ops.linear(x.float(), w, b, mask=mask, mask_mode=ops.MASK_EDGE, n_nodes=n)
    assert_close(y, ref, what="e_out fwd from bf16 x")
    # e_out weight gradient: dy fp32 [M,64], x bf16 [M,256]
    dy = rnd(M, 64)
    dw = torch.empty(64, 256, device="cuda"); db = torch.empty(64, device="cuda")
    _lib.call("mbvae_linear_bwd_weight_bf16", dy.data_ptr(), 0, 64, x.data_ptr(), 1, 256, dw.data_ptr(), db.data_ptr(), M, 256, 64,
              m8.data_ptr(), ops.MASK_EDGE, n, ws.data_ptr(), ws.numel(), st)
    dyk = dy * keep
    assert_close(dw, dyk.double().t() @ x.double(), what="e_out dw from bf16 x")
    assert_close(db, dyk.double().sum(0), what="e_out db")
    # e_out input gradient: dy fp32 -> dx bf16 [M,256]
    dx = torch.empty(M, 256, dtype=torch.bfloat16, device="cuda")
    _lib.call("mbvae_linear_bwd_input_bf16", dy.data_ptr(), 0, 64, w.data_ptr(), 0, dx.data_ptr(), 1, 256, M, 256, 64,
              m8.data_ptr(), ops.MASK_EDGE, n, st)
    ref = (dyk.double() @ w.double()).float()
    assert torch.equal(dx, ref.bfloat16()) or float((dx.float() - ref).abs().max()) <= 2.0 ** -8 * float(ref.abs().max()), \
        "dx is the fp32 result rounded to bf16"
    # e_mul|e_add input gradient: dy bf16 [M,512] -> dE fp32 [M,64] (+ fp32 addend), and its weight gradient
    dy12 = rnd(M, 512).bfloat16(); w12 = rnd(512, 64) / 8; add = rnd(M, 64); E = rnd(M, 64)
    dE = torch.empty(M, 64, device="cuda")
    _lib.call("mbvae_linear_bwd_input_bf16", dy12.data_ptr(), 1, 512, w12.data_ptr(), add.data_ptr(), dE.data_ptr(), 0, 64, M, 64, 512,
              m8.data_ptr(), ops.MASK_EDGE, n, st)
    dk = dy12.double() * keep
    assert_close(dE, dk @ w12.double() + add.double(), what="e12 dgrad from bf16 dy")
    dw12 = torch.empty(512, 64, device="cuda"); db12 = torch.empty(512, device="cuda")
    _lib.call("mbvae_linear_bwd_weight_bf16", dy12.data_ptr(), 1, 512, E.data_ptr(), 0, 64, dw12.data_ptr(), db12.data_ptr(), M, 64, 512,
              m8.data_ptr(), ops.MASK_EDGE, n, ws.data_ptr(), ws.numel(), st)
    assert_close(dw12, dk.t() @ E.double(), what="e12 dw from bf16 dy")
    assert_close(db12, dk.sum(0), what="e12 db")


def _layer_run(ops, bf16, seed=0):
    """One transformer layer at the real widths, forward + backward, in the given storage mode."""
    from mbvae_amd import _lib
    from mbvae_amd.graph_transformer import XEyTransformerLayer
    torch.manual_seed(seed)
    layer = XEyTransformerLayer(dx=256, de=64, dy=1024, n_head=8, dim_ffX=256, dim_ffE=128, dim_ffy=1024).cuda().eval()
    g = torch.Generator().manual_seed(seed + 1)
    bs, n = 16, 64
    mask = _ragged(bs, n, g)
    X = torch.randn(bs, n, 256, generator=g).cuda().requires_grad_(True)
    E = torch.randn(bs, n, n, 64, generator=g).cuda()
    E = ((E + E.transpose(1, 2)) / 2).requires_grad_(True)
    y = torch.randn(bs, 1024, generator=g).cuda().requires_grad_(True)
    cX, cE, cy = (torch.randn(t.shape, generator=g).cuda() for t in (X, E, y))
    ops.set_bf16_storage(bf16)
    _lib.PROFILE = {}
    try:
        oX, oE, oy = layer(X, E, y, mask)
        ((oX * cX).sum() + (oE * cE).sum() + (oy * cy).sum()).backward()
        torch.cuda.synchronize()
        called = {name for name, _ in _lib.PROFILE}
    finally:
        _lib.PROFILE = None
        ops.set_bf16_storage(False)
    grads = {k: p.grad.clone() for k, p in layer.named_parameters()}
    return dict(oX=oX.detach(), oE=oE.detach(), oy=oy.detach(), dX=X.grad, dE=E.grad, dy=y.grad, **grads), called


def test_bf16_storage_layer_matches_fp32_within_2e_2(ops, default_routes):
    ref, called_ref = _layer_run(ops, False)
    got, called = _layer_run(ops, True)
    # the mode really ran on the bf16 entry points (and the default really did not)
    need = {"mbvae_edge_fused_fwd_bf16", "mbvae_edge_attn_bwd_bf16", "mbvae_linear_bwd_input_bf16", "mbvae_linear_bwd_weight_bf16"}
    assert need <= called, f"bf16 entry points not used: {need - called}"
    # e_out reads the bf16 P either on its own or with the LayerNorm in its epilogue (shapes of at least 8192 pair rows)
    e_out = {"mbvae_linear_fwd_bf16", "mbvae_linear_ln_fwd_bf16"}
    assert e_out & called, "e_out did not read P as bf16"
    assert not ((need | e_out) & called_ref)
    # forward: every element within 2e-2 of the tensor's scale
    for k in ("oX", "oE", "oy"):
        assert_close(got[k], ref[k], rtol=BF16_RTOL, what=k)
    # gradients: relative L2 error <= 2e-2 per tensor.  (An element-wise bound cannot hold for them in ANY reduced
    # precision: the 1e-3 perturbation of the edge activations flips the ReLU gate of about one FFN unit in a thousand,
    # and each flip changes a few gradient entries by O(1) of their size; measured 1.6e-2 L2 on linE1.weight,
    # <= 6e-3 everywhere else.)
    worst = 0.0
    for k in ref:
        if k in ("oX", "oE", "oy"):
            continue
        assert bool(torch.isfinite(got[k]).all()), k
        den = float(ref[k].double().norm())
        err = float((got[k].double() - ref[k].double()).norm()) / max(den, 1e-30)
        worst = max(worst, err)
        assert err <= BF16_RTOL, f"{k}: relative L2 error {err:.3e} > {BF16_RTOL}"
    assert worst > 1e-6, "bf16 storage left no trace at all: the mode was not exercised"


def test_bf16_storage_train_steps_track_fp32(ops):
    """Three AdamW steps of the full denoiser train step in both modes from the same seeds: the losses stay within
    2e-2 of each other (dropout masks are position-hashed, hence identical in both modes)."""
    from mbvae_amd.trainstep import CFG3, DenoiserTrainStep, synthetic_graphs
    losses = {}
    for mode in (False, True):
        ops.set_bf16_storage(mode)
        try:
            torch.manual_seed(11)
            tr = DenoiserTrainStep(dict(CFG3, n_layers=2), device=torch.device("cuda", 0), distributed=False, seed=5)
            g = torch.Generator().manual_seed(21)
            X0, E0, mask = synthetic_graphs(16, 64, torch.device("cuda", 0), g)
            cond = torch.randn(16, 1024, generator=g).cuda()
            torch.manual_seed(13)
            losses[mode] = [float(tr.step(X0, E0, mask, cond)) for _ in range(3)]
        finally:
            ops.set_bf16_storage(False)
    for a, b in zip(losses[False], losses[True]):
        assert math.isfinite(b) and abs(a - b) <= BF16_RTOL * abs(a), (losses[False], losses[True])


def test_bf16_recomputing_pair_matches_fp32_edge_block(ops, default_routes):
    """Batches of at least one graph per two compute units take the recomputing pair in bf16 mode as well (P, dP and
    dE1|dE2 in bf16, E1|E2 never stored): forward element-wise and gradients in relative L2 within 2e-2 of fp32."""
    from mbvae_amd import _lib
    bs, n, dx, de = 128, 40, 256, 64
    g = torch.Generator().manual_seed(5)
    mask = _ragged(bs, n, g)
    E = torch.randn(bs, n, n, de, generator=g).cuda()
    w = (torch.randn(2 * dx, de, generator=g) / 8).cuda(); b = (torch.randn(2 * dx, generator=g) * 0.1).cuda()
    QKV = torch.randn(bs, n, 3 * dx, generator=g).cuda() * 0.6
    ya = torch.randn(bs, dx, generator=g).cuda() * 0.3; ym = torch.randn(bs, dx, generator=g).cuda() * 0.3
    pm = (mask.unsqueeze(1) & mask.unsqueeze(2)).unsqueeze(-1).float()
    cP = torch.randn(bs, n, n, dx, generator=g).cuda() * pm; cW = torch.randn(bs, n, dx, generator=g).cuda()
    res, called = {}, {}
    for mode in (False, True):
        ops.set_bf16_storage(mode)
        _lib.PROFILE = {}
        try:
            ins = [t.clone().requires_grad_(True) for t in (E, w, b, QKV, ya, ym)]
            P, wV, stats, _ = ops.edge_block(*ins, mask, 8)
            ((P.float() * cP).sum() + (wV * cW).sum()).backward()
            torch.cuda.synchronize()
            called[mode] = {name for name, _ in _lib.PROFILE}
        finally:
            _lib.PROFILE = None
            ops.set_bf16_storage(False)
        res[mode] = [P.detach().float() * pm, wV.detach()] + [t.grad for t in ins]
    assert {"mbvae_edge_fused_fwd_recompute_bf16", "mbvae_edge_fused_bwd_bf16"} <= called[True]
    assert {"mbvae_edge_fused_fwd_recompute", "mbvae_edge_fused_bwd"} <= called[False]
    assert P.dtype == torch.bfloat16
    assert_close(res[True][0], res[False][0], rtol=BF16_RTOL, what="P"); assert_close(res[True][1], res[False][1], rtol=BF16_RTOL, what="wV")
    for a, r, name in zip(res[True][2:], res[False][2:], ("dE", "dw12", "db12", "dQKV", "dye_add", "dye_mul")):
        err = float((a.double() - r.double()).norm() / r.double().norm())
        assert err <= BF16_RTOL, f"{name}: relative L2 error {err:.3e}"


# What "within 2e-2" means for a GRADIENT in the bf16 storage mode: relative L2 per tensor, plus at most one element in a
# thousand further than 2e-2 of the tensor's largest magnitude from the oracle.  Element-wise closeness of every entry is
# not on offer for any bf16 implementation of this layer: the ~1e-3 perturbation that bf16 storage puts on the edge
# activations flips the ReLU gate of about one hidden unit in a thousand of the edge feed-forward block, and a flipped gate
# changes the gradient of its pair row by that whole unit, not by a rounding-sized amount.
BF16_OUTLIER_FRACTION = 1e-3


def test_bf16_storage_layer_matches_the_oracle_within_2e_2(ops, default_routes):
    """The bf16 storage mode next to the CPU ORACLE (not only next to the fp32 HIP path): one real-width layer on 16 ragged
    graphs (65 536 pair rows, so every *_bf16 entry point runs), forward and all gradients.  Forward tensors: element-wise
    within 2e-2 of the tensor's largest magnitude; gradients: 2e-2 in relative L2 and at most BF16_OUTLIER_FRACTION of the
    entries further than 2e-2 of the largest magnitude."""
    from oracle import denoiser as O_den
    got, called = _layer_run(ops, True)
    assert {"mbvae_edge_fused_fwd_bf16", "mbvae_edge_attn_bwd_bf16", "mbvae_linear_bwd_input_bf16", "mbvae_linear_bwd_weight_bf16"} <= called
    # the same case on the CPU: _layer_run draws weights and inputs from fixed seeds
    from mbvae_amd.graph_transformer import XEyTransformerLayer
    torch.manual_seed(0)
    layer = XEyTransformerLayer(dx=256, de=64, dy=1024, n_head=8, dim_ffX=256, dim_ffE=128, dim_ffy=1024)
    p = {k: v.detach().clone().requires_grad_(True) for k, v in layer.state_dict().items()}
    g = torch.Generator().manual_seed(1)
    bs, n = 16, 64
    mask = _ragged(bs, n, g).cpu()
    X = torch.randn(bs, n, 256, generator=g).requires_grad_(True)
    E = torch.randn(bs, n, n, 64, generator=g)
    E = ((E + E.transpose(1, 2)) / 2).requires_grad_(True)
    y = torch.randn(bs, 1024, generator=g).requires_grad_(True)
    cX, cE, cy = (torch.randn(t.shape, generator=g) for t in (X, E, y))
    oX, oE, oy = O_den.xey_layer(p, X, E, y, mask, 8)
    ((oX * cX).sum() + (oE * cE).sum() + (oy * cy).sum()).backward()
    want = dict(oX=oX.detach(), oE=oE.detach(), oy=oy.detach(), dX=X.grad, dE=E.grad, dy=y.grad, **{k: v.grad for k, v in p.items()})
    report, failures = [], []
    for k, ref in want.items():
        a, b = got[k].double().cpu(), ref.double()
        assert bool(torch.isfinite(a).all()), k
        scale = max(float(b.abs().max()), 1e-30)
        d = (a - b).abs()
        elem = float(d.max()) / scale
        l2 = float((a - b).norm()) / max(float(b.norm()), 1e-30)
        far = float((d > BF16_RTOL * scale).double().mean())
        report.append(f"{k:28s} element-wise {elem:.2e}  L2 {l2:.2e}  entries beyond 2e-2: {far:.1e}")
        if k in ("oX", "oE", "oy"):
            if elem > BF16_RTOL:
                failures.append(f"{k}: element-wise error {elem:.3e} > {BF16_RTOL} of scale")
        elif l2 > BF16_RTOL or far > BF16_OUTLIER_FRACTION:
            failures.append(f"{k}: relative L2 {l2:.3e}, {far:.2e} of the entries beyond {BF16_RTOL} of scale")
    if os.environ.get("MBVAE_TEST_REPORT"):
        with open(os.environ["MBVAE_TEST_REPORT"], "a") as f:
            f.write("\n".join(report) + "\n")
    assert not failures, "\n".join(failures + ["--- all tensors:"] + report)
