"""Does a short kernel give other bits when a long kernel of another stream shares the GPU with it?  GPU box only.

Victims (second stream): the node-path LayerNorm backward / forward, FiLM, pooling, a node-path Linear.  Partners (main
stream, issued in a loop so that they are running while the victims execute): the edge block forward + backward, the
LayerNorm-epilogue Linear, the edge feed-forward block, a plain device copy.  Every victim result is compared bit for
bit with the result of the same call on an idle GPU.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mb-vae-dti_amd")); sys.path.insert(0, ROOT)
import torch

from mbvae_amd import _lib, ops as O

torch.manual_seed(0)
dev = torch.device("cuda")
bs, n, dx, de = int(os.environ.get("CHECK_BS", "256")), 64, 256, 64
REPS = int(os.environ.get("CHECK_REPS", "40"))
side = torch.cuda.Stream()
main = torch.cuda.current_stream()

# ---- victims: closures that run one library call into fresh outputs and return them
x = torch.randn(bs, n, dx, device=dev); r = torch.randn(bs, n, dx, device=dev); dy = torch.randn(bs, n, dx, device=dev) * torch.rand(bs, n, 1, device=dev)
gamma = torch.randn(dx, device=dev); beta = torch.randn(dx, device=dev)
mean = torch.empty(bs * n, device=dev); rstd = torch.empty(bs * n, device=dev); yln = torch.empty_like(x)
_lib.call("mbvae_ln_residual_fwd", x.data_ptr(), r.data_ptr(), gamma.data_ptr(), beta.data_ptr(), yln.data_ptr(), mean.data_ptr(),
          rstd.data_ptr(), bs * n, dx, 1e-5, 0.1, 1234, main.cuda_stream)
wn = torch.randn(dx, dx, device=dev) * 0.05; bn = torch.randn(dx, device=dev)
mul = torch.randn(bs, dx, device=dev); add = torch.randn(bs, dx, device=dev)


VLIB = None
if os.environ.get("VICTIM_LIB"):            # the victim from another build of the library (the partners stay on the default one)
    import ctypes
    VLIB = ctypes.CDLL(os.environ["VICTIM_LIB"])
    VLIB.mbvae_ln_residual_bwd.argtypes = _lib.SIGNATURES["mbvae_ln_residual_bwd"]


def _vcall(name, *a):
    if VLIB is None:
        return _lib.call(name, *a)
    st = getattr(VLIB, name)(*a)
    assert st == 0, st


def v_ln_bwd():
    o = [torch.empty_like(x), torch.empty_like(x), torch.empty_like(gamma), torch.empty_like(gamma)]
    ws = O.workspace(dev)
    _vcall("mbvae_ln_residual_bwd", x.data_ptr(), r.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dy.data_ptr(),
              o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), bs * n, dx, 0.1, 1234, ws.data_ptr(), ws.numel(),
              torch.cuda.current_stream().cuda_stream)
    return o


def v_ln_fwd():
    o = [torch.empty_like(x), torch.empty_like(mean), torch.empty_like(mean)]
    _lib.call("mbvae_ln_residual_fwd", x.data_ptr(), r.data_ptr(), gamma.data_ptr(), beta.data_ptr(), o[0].data_ptr(), o[1].data_ptr(),
              o[2].data_ptr(), bs * n, dx, 1e-5, 0.1, 1234, torch.cuda.current_stream().cuda_stream)
    return o


def v_film_bwd():
    o = [torch.empty_like(x), torch.empty_like(mul), torch.empty_like(mul)]
    _lib.call("mbvae_film_bwd", dy.data_ptr(), mul.data_ptr(), x.data_ptr(), o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), bs, n, dx,
              torch.cuda.current_stream().cuda_stream)
    return o


def v_linear():
    with torch.no_grad():
        return [O.linear(x, wn, bn)]


def v_linear_bwd():
    xx = x.clone().requires_grad_(True); w = wn.clone().requires_grad_(True)
    O.linear(xx, w, bn).backward(dy)
    return [xx.grad, w.grad]


def v_torch_sum():
    return [x.sum(-1), (x * r).sum(0)]


VICTIMS = {"ln_residual_bwd[256]": v_ln_bwd}
if os.environ.get("CHECK_ALL_VICTIMS") == "1":
    VICTIMS.update({"ln_residual_fwd[256]": v_ln_fwd, "film_bwd": v_film_bwd, "linear 256->256 fwd": v_linear,
                    "linear 256->256 fwd+bwd": v_linear_bwd, "torch row / column sums": v_torch_sum})

# ---- partners: closures that keep the main stream busy with one kind of long kernel
E = torch.randn(bs, n, n, de, device=dev); mask = torch.ones(bs, n, dtype=torch.bool, device=dev)
w12 = torch.randn(2 * dx, de, device=dev) * 0.1; b12 = torch.randn(2 * dx, device=dev) * 0.1
QKV = torch.randn(bs, n, 3 * dx, device=dev); ya = torch.randn(bs, dx, device=dev); ym = torch.randn(bs, dx, device=dev) * 0.1
wo = torch.randn(de, dx, device=dev) * 0.05; bo = torch.randn(de, device=dev); g64 = torch.randn(de, device=dev); b64 = torch.randn(de, device=dev)
w1 = torch.randn(128, de, device=dev) * 0.1; b1 = torch.randn(128, device=dev); w2 = torch.randn(de, 128, device=dev) * 0.1
big = torch.empty(1 << 28, device=dev); big2 = torch.empty(1 << 28, device=dev)
dP = torch.randn(bs, n, n, dx, device=dev)


def p_edge_block():
    Eg = E.clone().requires_grad_(True); wg = w12.clone().requires_grad_(True); q = QKV.clone().requires_grad_(True)
    P, wV, st, Ea = O.edge_block(Eg, wg, b12, q, ya, ym, mask, 8)
    torch.autograd.backward([P, wV], [dP, x])


def p_linear_ln():
    Pg = dP.clone().requires_grad_(True); wg = wo.clone().requires_grad_(True)
    out = O.linear_ln(Pg, wg, bo, E, g64, b64, 1e-5, 0.1, 99, mask=mask, mask_mode=O.MASK_EDGE, n_nodes=n)
    out.backward(E)


def p_ffn():
    Eg = E.clone().requires_grad_(True); a = w1.clone().requires_grad_(True); b = w2.clone().requires_grad_(True)
    out, _ = O.ffn(Eg, a, b1, b, bo, drop_p=0.1, seed=5, ln=(g64, b64, 1e-5, 0.1, 6))
    out.backward(E)


def p_copy():
    big2.copy_(big)


def p_none():
    pass


zE = torch.randn(bs, n, n, de, device=dev); dyE = torch.randn(bs, n, n, de, device=dev)
oE1, oE2 = torch.empty_like(zE), torch.empty_like(zE); dg64, db64 = torch.empty(de, device=dev), torch.empty(de, device=dev)
h128 = torch.empty(bs, n, n, 128, device=dev); dw_ = torch.empty(128, de, device=dev); db_ = torch.empty(128, device=dev)
M_ = bs * n * n


def p_ln_sum_bwd():
    ws = O.workspace(dev)
    _lib.call("mbvae_ln_sum_bwd", zE.data_ptr(), g64.data_ptr(), dyE.data_ptr(), oE1.data_ptr(), oE2.data_ptr(), dg64.data_ptr(), db64.data_ptr(),
              M_, de, 1e-5, 0.1, 77, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)


def p_tsgemm2_fwd():
    ws = O.workspace(dev)
    _lib.call("mbvae_linear_fwd", zE.data_ptr(), de, w1.data_ptr(), b1.data_ptr(), h128.data_ptr(), 128, M_, de, 128, 1, 0, 0, 0, 0.1, 5, 0,
              ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)


def p_tswgrad3():
    ws = O.workspace(dev)
    _lib.call("mbvae_linear_bwd_weight", h128.data_ptr(), 128, zE.data_ptr(), de, dw_.data_ptr(), db_.data_ptr(), M_, de, 128, 0, 1.0, 0, 0, 0, 0,
              ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)


def p_linear_ln_fwd():
    _lib.call("mbvae_linear_ln_fwd", h128.data_ptr(), 128, w2.data_ptr(), bo.data_ptr(), zE.data_ptr(), g64.data_ptr(), b64.data_ptr(), oE1.data_ptr(),
              oE2.data_ptr(), M_, 128, de, 0, 0, 0, 1e-5, 0.1, 6, torch.cuda.current_stream().cuda_stream)


def p_ln_fwd64():
    mm, rr = torch.empty(M_, device=dev), torch.empty(M_, device=dev)
    _lib.call("mbvae_ln_residual_fwd", zE.data_ptr(), dyE.data_ptr(), g64.data_ptr(), b64.data_ptr(), oE1.data_ptr(), mm.data_ptr(), rr.data_ptr(),
              M_, de, 1e-5, 0.1, 1234, torch.cuda.current_stream().cuda_stream)


PARTNERS = {"(idle)": p_none, "ln_sum_bwd alone": p_ln_sum_bwd, "linear fwd 64->128 alone (tsgemm2)": p_tsgemm2_fwd,
            "weight gradient alone (tswgrad3)": p_tswgrad3, "Linear+LayerNorm fwd alone": p_linear_ln_fwd, "ln_residual_fwd[64] alone": p_ln_fwd64, "device copy 1 GB": p_copy, "edge block fwd+bwd": p_edge_block, "e_out Linear+LayerNorm fwd+bwd": p_linear_ln,
            "edge feed-forward fwd+bwd": p_ffn}

if os.environ.get("CHECK_FEW_PARTNERS", "1") == "1":
    PARTNERS = {k: PARTNERS[k] for k in ("(idle)", "linear fwd 64->128 alone (tsgemm2)", "weight gradient alone (tswgrad3)")}
for vname, victim in VICTIMS.items():
    ref = victim()
    torch.cuda.synchronize()
    for pname, partner in PARTNERS.items():
        bad = 0
        results = []
        side.wait_stream(main)
        for it in range(REPS):
            partner()
            with torch.cuda.stream(side):
                results.append(victim())
        torch.cuda.synchronize()
        shown = False
        for res in results:
            isbad = any(not torch.equal(a, b) for a, b in zip(ref, res))
            bad += isbad
            if isbad and not shown and vname.startswith("ln_residual_bwd"):
                shown = True
                which = [i for i, (a, b) in enumerate(zip(ref, res)) if not torch.equal(a, b)]
                d = (ref[0] != res[0]).reshape(-1, dx)
                rws = d.any(1).nonzero().flatten()
                print(f"    outputs {which} differ; dx: {rws.numel()} bad rows {rws[:10].tolist()}, bad elements per row {d[rws].sum(1)[:10].tolist()}", flush=True)
                if rws.numel():
                    # fit (dz_bad - dz_good) / rstd = -(dm1) - xhat * (dm2): were the two row sums off, and by how much?
                    row = int(rws[0])
                    xs = x.reshape(-1, dx)[row].double(); rs = r.reshape(-1, dx)[row].double()
                    off = row * dx + torch.arange(dx, device=dev)
                    drs = (res[1].reshape(-1, dx)[row] != 0).double() / 0.9       # dropout scale as dr shows it (0 where dz happens to be 0)
                    good, badv = ref[0].reshape(-1, dx)[row].double(), res[0].reshape(-1, dx)[row].double()
                    delta = (badv - good) / float(rstd[row])
                    print(f"    row {row}: delta/rstd min {float(delta.min()):.4e} max {float(delta.max()):.4e} mean {float(delta.mean()):.4e}; "
                          f"g = dy*gamma of that row: max |g| {float((dy.reshape(-1, dx)[row] * gamma).abs().max()):.3e}, "
                          f"sum g / 256 = {float((dy.reshape(-1, dx)[row] * gamma).double().sum() / 256):.4e}", flush=True)
                    dd_ = (badv - good).abs()
                    ev = dd_[192:256:2]; od = dd_[193:256:2]; rest = dd_[:192]
                    print(f"    row {row}: |dx_bad - dx_good| max over columns 192..254 even {float(ev.max()):.3e}, odd {float(od.max()):.3e}, "
                          f"columns 0..191 {float(rest.max()):.3e}; row max |dx| {float(good.abs().max()):.3e}", flush=True)
                    # which z = x + r * scale did the kernel use in columns 192..255?  m1, m2 from the columns that are right
                    gd = (dy.reshape(-1, dx)[row] * gamma).double()
                    rs_, mu_ = float(rstd[row]), float(mean[row])
                    gx, gdx = ref[0].reshape(-1, dx)[row].double(), ref[1].reshape(-1, dx)[row].double()
                    scale = torch.where(gx != 0, gdx / gx, torch.full_like(gx, float("nan")))
                    z_true = xs + rs * torch.nan_to_num(scale, nan=1.0 / 0.9)
                    xh_true = (z_true - mu_) * rs_
                    A = torch.stack((torch.ones(192, device=dev, dtype=torch.double), xh_true[:192]), 1)
                    sol = torch.linalg.lstsq(A, (gd - badv / rs_)[:192].unsqueeze(1)).solution.flatten()
                    m1, m2 = float(sol[0]), float(sol[1])
                    xh_impl = (gd - m1 - badv / rs_) / m2
                    z_impl = xh_impl / rs_ + mu_
                    print(f"    fitted row sums m1 {m1:.5e} m2 {m2:.5e}; columns 192..203:", flush=True)
                    for c in range(192, 204):
                        print(f"      col {c}: x {float(xs[c]):+.4f} r {float(rs[c]):+.4f} scale {float(scale[c]):.4f} z_true {float(z_true[c]):+.4f} "
                              f"z_used {float(z_impl[c]):+.4f}  z_used - r*scale = {float(z_impl[c] - rs[c] * torch.nan_to_num(scale, nan=1.0 / 0.9)[c]):+.4f}  "
                              f"(z_used - x)/r = {float((z_impl[c] - xs[c]) / rs[c]):+.4f}", flush=True)
                    # the offset of z in the bad columns against differences of row means
                    for row2 in [int(q) for q in rws[:40].tolist()][::3]:
                        g2 = (dy.reshape(-1, dx)[row2] * gamma).double(); rs2, mu2 = float(rstd[row2]), float(mean[row2])
                        gx2, gdx2 = ref[0].reshape(-1, dx)[row2].double(), ref[1].reshape(-1, dx)[row2].double()
                        sc2 = torch.nan_to_num(torch.where(gx2 != 0, gdx2 / gx2, torch.full_like(gx2, float("nan"))), nan=1.0 / 0.9)
                        zt = x.reshape(-1, dx)[row2].double() + r.reshape(-1, dx)[row2].double() * sc2
                        xt = (zt - mu2) * rs2
                        bv = res[0].reshape(-1, dx)[row2].double()
                        A2 = torch.stack((torch.ones(192, device=dev, dtype=torch.double), xt[:192]), 1)
                        so2 = torch.linalg.lstsq(A2, (g2 - bv / rs2)[:192].unsqueeze(1)).solution.flatten()
                        zi = ((g2 - float(so2[0]) - bv / rs2) / float(so2[1])) / rs2 + mu2
                        off_ = (zi - zt)[192:]
                        big_ = off_[off_.abs() > 1e-3]
                        cand = {"mean[R]-mean[R-2048]": mu2 - float(mean[row2 - 2048]) if row2 >= 2048 else None,
                                "mean[R]-mean[R-4096]": mu2 - float(mean[row2 - 4096]) if row2 >= 4096 else None,
                                "mean[R]-mean[R+2048]": mu2 - float(mean[row2 + 2048]) if row2 + 2048 < bs * n else None}
                        if big_.numel():
                            wrong = mu2 - float(big_.mean())
                            near = (mean.double() - wrong).abs()
                            hits = (near < 3e-6).nonzero().flatten().tolist()
                            print(f"    row {row2}: true mean {mu2:+.6f}, value used instead {wrong:+.6f}; rows whose mean equals it: {hits[:6]} "
                                  f"(their distance from this row: {[h - row2 for h in hits[:6]]}); bad comp: {'x' if bool((off_.abs() > 1e-3)[0::4].any()) else 'z'}", flush=True)
                        print(f"    row {row2}: z offset in {big_.numel()} columns = {float(big_.mean()) if big_.numel() else 0:+.5f} (spread {float(big_.std()) if big_.numel() > 1 else 0:.1e}); "
                              + "  ".join(f"{k} = {v:+.5f}" for k, v in cand.items() if v is not None), flush=True)
                    dgb = (ref[2] != res[2]).nonzero().flatten()
                    print(f"    dgamma: {dgb.numel()} columns differ: {dgb[:40].tolist()}", flush=True)
        print(f"{vname:28s} next to {pname:34s}: {bad} of {REPS} results differ from the idle-GPU result", flush=True)
        del results
