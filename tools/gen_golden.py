"""Generate the golden fixtures under tests/golden/ from the reference implementation.

Runs ONLY in the build container (needs /root/reference, CPU).  For every fixture
it also cross-checks this repo's CPU oracle (oracle/*.py) against the reference
output before writing, so a fixture is never written from a disagreeing pair.

    python tools/gen_golden.py

Fixtures are plain .npz files (numpy arrays only; loaded with allow_pickle=False).
State dicts are stored with keys "sd/<state_dict key>".
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_loader  # noqa: E402
from tests._weights import synthetic_state_dict  # noqa: E402

ref = ref_loader.load()
from oracle import denoiser as O_den, diffusion as O_dif, features as O_feat, heads as O_heads  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def npz(name, **arrays):
    flat = {}
    for k, v in arrays.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                flat[f"{k}/{kk}"] = vv.detach().cpu().numpy() if torch.is_tensor(vv) else np.asarray(vv)
        else:
            flat[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **flat)
    print(f"wrote {path}  ({os.path.getsize(path)/1024:.1f} KiB)")


def close(a, b, tol=1e-5, what=""):
    a, b = a.detach().double(), b.detach().double()
    err = (a - b).abs().max().item()
    ref_mag = max(b.abs().max().item(), 1e-12)
    assert err <= tol * max(ref_mag, 1.0), f"oracle != reference for {what}: {err} (mag {ref_mag})"


def ragged_graph(bs, n, sizes, cx, ce, gen):
    """Clean dense graphs in the reference's to_dense convention: padded nodes all-zero X,
    every off-diagonal pair (padded ones too) has a one-hot E (class 0 = no bond), diagonal zero."""
    mask = torch.arange(n)[None, :] < torch.tensor(sizes)[:, None]
    Xc = torch.randint(0, cx, (bs, n), generator=gen)
    X = F.one_hot(Xc, cx).float() * mask.unsqueeze(-1)
    Ec = torch.randint(0, ce, (bs, n, n), generator=gen)
    Ec = Ec * (torch.rand(bs, n, n, generator=gen) < 0.35)
    Ec = torch.triu(Ec, 1)
    Ec = Ec + Ec.transpose(1, 2)
    pair = mask.unsqueeze(1) & mask.unsqueeze(2)
    Ec = Ec * pair                                     # padded pairs: class 0
    E = F.one_hot(Ec, ce).float()
    E[:, torch.arange(n), torch.arange(n)] = 0
    return X, E, mask


# =============================================================================== G1/G2 schedule + transitions
def gen_schedule():
    T = 500
    ns = ref.dnoise.PredefinedNoiseScheduleDiscrete(timesteps=T)
    sch = O_dif.Schedule(T)
    assert torch.equal(ns.betas, sch.betas) and torch.equal(ns.alphas_bar, sch.alphas_bar)
    mx = torch.tensor([0.72, 0.12, 0.10, 0.04, 0.02])
    me = torch.tensor([0.90, 0.05, 0.03, 0.01, 0.01])
    tm = ref.dnoise.MarginalUniformTransition(mx, me, y_classes=4)
    t_int = torch.tensor([[0.], [1.], [250.], [499.], [500.]])
    beta = ns(t_normalized=t_int / T)
    abar = ns.get_alpha_bar(t_normalized=t_int / T)
    abar_s = ns.get_alpha_bar(t_normalized=(t_int - 1) / T)
    qt, qtb = tm.get_Qt(beta, "cpu"), tm.get_Qt_bar(abar, "cpu")
    assert torch.equal(qt.X, O_dif.Qt(beta, mx)) and torch.equal(qtb.E, O_dif.Qt_bar(abar, me))
    npz("schedule", betas=ns.betas, alphas_bar=ns.alphas_bar, t_int=t_int, beta_t=beta, alpha_t_bar=abar,
        alpha_s_bar=abar_s, mx=mx, me=me, Qt_X=qt.X, Qt_E=qt.E, Qtb_X=qtb.X, Qtb_E=qtb.E)


# =============================================================================== G3/G4 noising
class _NoiseHost(nn.Module):
    """Just enough of FullDTIModel's attribute surface for apply_noise / get_extra_features."""


def make_full_model(phase="pretrain_drug", cx=8, tiny=True, T=500, seed=0, n_dec_layers=2):
    full = ref_loader.load_full()
    torch.manual_seed(seed)
    atom_types = ["C", "O", "P", "N", "S", "Cl", "F", "H"][:cx]
    emb, hid = (32, 16) if tiny else (1024, 512)
    gt_kwargs = dict(
        n_layers=n_dec_layers,
        input_dims={"X": cx + 8, "E": 5, "y": emb + 13},
        output_dims={"X": cx, "E": 5, "y": emb},
        hidden_mlp_dims={"X": 24, "E": 16, "y": 40} if tiny else {"X": 256, "E": 128, "y": 2048},
        hidden_dims=({"dx": 32, "de": 16, "dy": 32, "n_head": 4, "dim_ffX": 24, "dim_ffE": 16, "dim_ffy": 40} if tiny else
                     {"dx": 256, "de": 64, "dy": 1024, "n_head": 8, "dim_ffX": 256, "dim_ffE": 128, "dim_ffy": 1024}),
    )
    nm = torch.tensor([0.55, 0.15, 0.02, 0.12, 0.05, 0.04, 0.05, 0.02][:cx]); nm = nm / nm.sum()
    infos = {
        "general": {"atom_types": atom_types, "bond_types": ["SINGLE", "DOUBLE", "TRIPLE", "AROMATIC"],
                    "num_atom_types": cx, "max_n_nodes": 64,
                    "atom_valencies": [4, 2, 3, 3, 2, 1, 1, 1][:cx],
                    "atom_weights": [12.011, 15.999, 30.974, 14.007, 32.06, 35.45, 18.998, 1.008][:cx],
                    "max_weight": 1000},
        "dataset": {"dataset_name": "synthetic", "max_nodes": 64,
                    "node_count_distribution": {"3": 1, "5": 2, "7": 3, "9": 2},
                    "node_marginals": nm.tolist(),
                    "edge_marginals": [0.90, 0.05, 0.03, 0.01, 0.01], "valency_distribution": []},
    }
    kw = dict(
        phase=phase, finetune_score=None, learning_rate=5e-4, weight_decay=1e-4, scheduler="const",
        weights=[1.0, 0.001, 0.5, 3.0], dti_weights=[1.0, 0.903964, 0.282992, 0.755172] if phase == "train" else None,
        diff_weights=[1, 5], contrastive_temp=0.1,
        drug_features={"FP-Morgan": 64, "EMB-BiomedGraph": 24} if tiny else {"FP-Morgan": 2048},
        target_features={"FP-ESP": 48, "EMB-ESM": 40, "EMB-NT": 20} if phase == "train" else {},
        embedding_dim=emb, hidden_dim=hid, factor=4, n_layers=2, activation="relu", dropout=0.1, bias=True,
        encoder_type="transformer", aggregator_type="attentive", diffusion_steps=T, sample_every_val=5,
        val_samples_per_embedding=2, test_samples_per_embedding=2, graph_transformer_kwargs=gt_kwargs, dataset_infos=infos)
    model = full.FullDTIModel(**kw)
    return model, kw, infos


def gen_noise():
    model, kw, infos = make_full_model()
    cx, ce, bs, n = 8, 5, 4, 9
    g = torch.Generator().manual_seed(11)
    X, E, mask = ragged_graph(bs, n, [9, 7, 5, 3], cx, ce, g)
    mx = torch.tensor(infos["dataset"]["node_marginals"]); me = torch.tensor(infos["dataset"]["edge_marginals"])
    sch = O_dif.Schedule(500)

    # --- G3: sample_discrete_features on arbitrary row-stochastic inputs
    probX = torch.softmax(torch.randn(bs, n, cx, generator=g), -1)
    probE = torch.softmax(torch.randn(bs, n, n, ce, generator=g), -1)
    probE[0, 0, 1] = torch.tensor([0., 0.5, 0., 0.5, 0.])          # zero-probability classes
    torch.manual_seed(123)
    out = ref.dutils.sample_discrete_features(probX.clone(), probE.clone(), mask)
    torch.manual_seed(123)
    expX = torch.empty(bs * n, cx).exponential_(1)
    expE = torch.empty(bs * n * n, ce).exponential_(1)
    oX, oE = O_dif.sample_graph(probX, probE, mask, expX, expE)
    assert torch.equal(out.X, oX) and torch.equal(out.E, oE), "sampler mismatch"
    rec = dict(X=X, E=E, mask=mask, mx=mx, me=me, s_probX=probX, s_probE=probE, s_expX=expX, s_expE=expE,
               s_X=out.X, s_E=out.E)

    # --- G4: apply_noise, train (t may be 0) and eval (t >= 1)
    for mode, seed in (("train", 5), ("eval", 6)):
        model.train(mode == "train")
        torch.manual_seed(seed)
        X_t, E_t, npar = model.apply_noise(X.clone(), E.clone(), mask)
        torch.manual_seed(seed)
        low = 0 if mode == "train" else 1
        t_int = torch.randint(low, 501, size=(bs, 1)).float()
        eX = torch.empty(bs * n, cx).exponential_(1)
        eE = torch.empty(bs * n * n, ce).exponential_(1)
        assert torch.equal(t_int, npar["t_int"])
        oXt, oEt, onp, Xc, Ec = O_dif.apply_noise(sch, mx, me, X, E, mask, t_int, eX, eE)
        assert torch.equal(oXt, X_t.float()) and torch.equal(oEt, E_t.float())
        for k in ("t", "beta_t", "alpha_s_bar", "alpha_t_bar"):
            assert torch.equal(onp[k], npar[k]), k
        rec.update({f"{mode}_t_int": t_int, f"{mode}_expX": eX, f"{mode}_expE": eE, f"{mode}_X_t": X_t.float(),
                    f"{mode}_E_t": E_t.float(), f"{mode}_Xc": Xc, f"{mode}_Ec": Ec,
                    **{f"{mode}_{k}": npar[k] for k in ("t", "beta_t", "alpha_s_bar", "alpha_t_bar")}})
    # force t = 0 in one extra train-mode case (s = -1 wraps to alphas_bar[T])
    t0 = torch.tensor([[0.], [500.], [1.], [2.]])
    _, _, onp, _, _ = O_dif.apply_noise(sch, mx, me, X, E, mask, t0, eX, eE)
    ns = model.noise_schedule
    assert torch.equal(ns.get_alpha_bar(t_normalized=(t0 - 1) / 500), onp["alpha_s_bar"])
    rec.update(t0_t_int=t0, t0_alpha_s_bar=onp["alpha_s_bar"], t0_beta_t=onp["beta_t"])

    # --- limit-distribution sampling (sample_discrete_feature_noise)
    lim = ref.dutils.PlaceHolder(X=mx, E=me, y=None)
    torch.manual_seed(77)
    zT = ref.dutils.sample_discrete_feature_noise(lim, mask)
    torch.manual_seed(77)
    eX = torch.empty(bs * n, cx).exponential_(1)
    eE = torch.empty(bs * n * n, ce).exponential_(1)
    lX, lE = O_dif.sample_limit(mx, me, mask, eX, eE)
    assert torch.equal(lX, zT.X) and torch.equal(lE, zT.E)
    rec.update(lim_expX=eX, lim_expE=eE, lim_X=zT.X, lim_E=zT.E)
    npz("noise", **rec)
    return model, infos, X, E, mask, rec


# =============================================================================== G5 extra features
def gen_features(model, infos, X, E, mask, rec):
    X_t, E_t = rec["eval_X_t"], rec["eval_E_t"]
    t = rec["eval_t"]
    bs, n = mask.shape
    torch.manual_seed(31)
    Xe, ye = model.get_extra_features(X_t, E_t, {"t": t}, mask)
    torch.manual_seed(31)
    noise = torch.randn(bs, n)
    gen_ = infos["general"]
    oXe, oye = O_feat.extra_features(X_t, E_t, t, mask, noise, gen_["max_n_nodes"], gen_["atom_valencies"],
                                     gen_["atom_weights"], gen_["max_weight"])
    # eigenvector columns (4,5) are sign-gauge dependent; everything else must match
    close(oXe[..., :4], Xe[..., :4], what="X_extra cycles/lcc")
    close(oXe[..., 6:], Xe[..., 6:], what="X_extra charge/valency")
    close(oXe[..., 4:6].abs(), Xe[..., 4:6].abs(), tol=1e-4, what="X_extra |eigvec|")
    close(oye, ye, tol=1e-5, what="y_extra")
    # a denser graph exercises the cycle formulas harder
    g = torch.Generator().manual_seed(3)
    Ad = (torch.rand(3, 12, 12, generator=g) < 0.4).float()
    Ad = torch.triu(Ad, 1); Ad = Ad + Ad.transpose(1, 2)
    kx, ky = ref.daug.KNodeCycles().k_cycles(Ad)
    okx, oky = O_feat.cycle_features(Ad)
    close(okx, kx, what="cycles x"); close(oky, ky, what="cycles y")
    L = O_feat.laplacian_masked(E_t, mask)
    w = torch.linalg.eigvalsh(L)
    npz("features", X_t=X_t, E_t=E_t, t=t, mask=mask, lcc_noise=noise, X_extra=Xe, y_extra=ye,
        dense_A=Ad, dense_kx=kx, dense_ky=ky, laplacian=L, eigvals=w,
        valencies=torch.tensor(gen_["atom_valencies"]), atom_weights=torch.tensor(gen_["atom_weights"]),
        max_weight=torch.tensor(gen_["max_weight"]), max_n_nodes=torch.tensor(gen_["max_n_nodes"]))


# =============================================================================== G6/G7 denoiser
def gen_denoiser():
    torch.manual_seed(2)
    dims = dict(n_layers=2, input_dims={"X": 13, "E": 5, "y": 45}, hidden_mlp_dims={"X": 24, "E": 16, "y": 40},
                hidden_dims={"dx": 32, "de": 16, "dy": 32, "n_head": 4, "dim_ffX": 24, "dim_ffE": 16, "dim_ffy": 40},
                output_dims={"X": 5, "E": 5, "y": 32})
    net = ref.gt.GraphTransformer(act_fn_in=nn.ReLU(), act_fn_out=nn.ReLU(), **dims).eval()
    bs, n = 4, 9
    g = torch.Generator().manual_seed(8)
    mask = torch.arange(n)[None, :] < torch.tensor([9, 7, 5, 3])[:, None]
    X = torch.randn(bs, n, 13, generator=g)
    E = torch.randn(bs, n, n, 5, generator=g); E = (E + E.transpose(1, 2)) / 2
    y = torch.randn(bs, 45, generator=g)
    cX, cE, cy = (torch.randn(bs, n, 5, generator=g), torch.randn(bs, n, n, 5, generator=g), torch.randn(bs, 32, generator=g))
    Xr, Er, yr = (t.clone().requires_grad_(True) for t in (X, E, y))
    out = net(Xr, Er, yr, mask)
    loss = (out.X * cX).sum() + (out.E * cE).sum() + (out.y * cy).sum()
    loss.backward()
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    grads = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
    # oracle check (fwd + autograd grads)
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    Xo, Eo, yo = (t.clone().requires_grad_(True) for t in (X, E, y))
    oX, oE, oy = O_den.graph_transformer(p, Xo, Eo, yo, mask, 2, 4, dims["output_dims"])
    close(oX, out.X, what="denoiser X"); close(oE, out.E, what="denoiser E"); close(oy, out.y, what="denoiser y")
    ((oX * cX).sum() + (oE * cE).sum() + (oy * cy).sum()).backward()
    for k in grads:
        close(p[k].grad, grads[k], tol=2e-5, what="grad " + k)
    close(Xo.grad, Xr.grad, what="dX"); close(Eo.grad, Er.grad, what="dE"); close(yo.grad, yr.grad, what="dy")
    npz("denoiser_tiny", sd=sd, grad=grads, X=X, E=E, y=y, mask=mask, cX=cX, cE=cE, cy=cy,
        out_X=out.X, out_E=out.E, out_y=out.y, dX=Xr.grad, dE=Er.grad, dy=yr.grad, loss=loss.detach())

    # ---- one layer + NodeEdgeBlock intermediates (hidden-width inputs, properly masked)
    layer = net.tf_layers[0]
    hX = torch.randn(bs, n, 32, generator=g) * mask.unsqueeze(-1)
    hE = torch.randn(bs, n, n, 16, generator=g); hE = (hE + hE.transpose(1, 2)) / 2
    hE = hE * (mask.unsqueeze(1) & mask.unsqueeze(2)).unsqueeze(-1)
    hy = torch.randn(bs, 32, generator=g)
    cap = {}
    blk = layer.self_attn
    hooks = [getattr(blk, nm).register_forward_hook(lambda m, i, o, nm=nm: cap.__setitem__(nm, o.detach().clone()))
             for nm in ("q", "k", "v", "e_mul", "e_add", "y_e_add", "y_e_mul", "y_x_add", "y_x_mul")]
    hooks += [blk.e_out.register_forward_hook(lambda m, i, o: cap.__setitem__("e_out_in", i[0].detach().clone())),
              blk.x_out.register_forward_hook(lambda m, i, o: cap.__setitem__("x_out_in", i[0].detach().clone())),
              blk.x_y.register_forward_hook(lambda m, i, o: cap.__setitem__("x_y", o.detach().clone())),
              blk.e_y.register_forward_hook(lambda m, i, o: cap.__setitem__("e_y", o.detach().clone()))]
    with torch.no_grad():
        nX, nE, ny = blk(hX, hE, hy, mask)
        lX, lE, ly = layer(hX, hE, hy, mask)
    for h in hooks:
        h.remove()
    lsd = {k: v.detach().clone() for k, v in layer.state_dict().items()}
    with torch.no_grad():
        olX, olE, oly = O_den.xey_layer(lsd, hX, hE, hy, mask, 4)
        xm = mask.unsqueeze(-1).float(); em = xm.unsqueeze(2) * xm.unsqueeze(1)
        pre, wV = O_den.edge_attention(cap["q"] * xm, cap["k"] * xm, cap["v"] * xm, cap["e_mul"] * em, cap["e_add"] * em,
                                       cap["y_e_add"], cap["y_e_mul"], mask, 4)
    close(olX, lX, what="layer X"); close(olE, lE, what="layer E"); close(oly, ly, what="layer y")
    close(pre, cap["e_out_in"], what="newE_pre")
    close(cap["y_x_add"].unsqueeze(1) + (cap["y_x_mul"].unsqueeze(1) + 1) * wV, cap["x_out_in"], what="wV film")
    # Xtoy / Etoy
    close(O_den.pool_stats(hX, (1,)) @ lsd["self_attn.x_y.lin.weight"].T + lsd["self_attn.x_y.lin.bias"], cap["x_y"], what="Xtoy")
    close(O_den.pool_stats(hE, (1, 2)) @ lsd["self_attn.e_y.lin.weight"].T + lsd["self_attn.e_y.lin.bias"], cap["e_y"], what="Etoy")
    npz("layer_tiny", sd=lsd, X=hX, E=hE, y=hy, mask=mask, cap=cap, blk_X=nX, blk_E=nE, blk_y=ny,
        out_X=lX, out_E=lE, out_y=ly, wV=wV)

    # ---- one real-width single graph (dx=256,de=64,dy=1024,H=8), forward only, 1 layer
    torch.manual_seed(4)
    rdims = dict(n_layers=1, input_dims={"X": 13, "E": 5, "y": 1037}, hidden_mlp_dims={"X": 256, "E": 128, "y": 2048},
                 hidden_dims={"dx": 256, "de": 64, "dy": 1024, "n_head": 8, "dim_ffX": 256, "dim_ffE": 128, "dim_ffy": 1024},
                 output_dims={"X": 5, "E": 5, "y": 1024})
    rnet = ref.gt.GraphTransformer(act_fn_in=nn.ReLU(), act_fn_out=nn.ReLU(), **rdims).eval()
    # weights too large to store: both sides rebuild them from tests/_weights.py (name-seeded)
    rnet.load_state_dict(synthetic_state_dict({k: tuple(v.shape) for k, v in rnet.state_dict().items()}))
    n2 = 6
    m2 = torch.tensor([[True] * 5 + [False]])
    X2 = torch.randn(1, n2, 13, generator=g); E2 = torch.randn(1, n2, n2, 5, generator=g); E2 = (E2 + E2.transpose(1, 2)) / 2
    y2 = torch.randn(1, 1037, generator=g)
    with torch.no_grad():
        o2 = rnet(X2, E2, y2, m2)
        q2 = O_den.graph_transformer(rnet.state_dict(), X2, E2, y2, m2, 1, 8, rdims["output_dims"])
    close(q2[0], o2.X, what="real X"); close(q2[1], o2.E, what="real E"); close(q2[2], o2.y, tol=2e-5, what="real y")
    shapes = {k: tuple(v.shape) for k, v in rnet.state_dict().items()}
    npz("denoiser_realwidth", X=X2, E=E2, y=y2, mask=m2, out_X=o2.X, out_E=o2.E, out_y=o2.y,
        shape_names=np.array(list(shapes.keys())), shape_dims=np.array([list(s) + [0] * (2 - len(s)) for s in shapes.values()]))


# =============================================================================== G8-G10 heads / encoders
def gen_heads():
    g = torch.Generator().manual_seed(21)
    rec = {}
    bs, n, cx, ce = 4, 9, 8, 5
    X, E, mask = ragged_graph(bs, n, [9, 7, 5, 3], cx, ce, g)
    xm = mask.unsqueeze(-1).float(); em = xm.unsqueeze(2) * xm.unsqueeze(1)
    pX = (torch.randn(bs, n, cx, generator=g) * xm).requires_grad_(True)
    pE = torch.randn(bs, n, n, ce, generator=g); pE = ((pE + pE.transpose(1, 2)) / 2 * em).requires_grad_(True)
    head = ref.heads.ReconstructionHead([1, 5])
    loss = head(pX, pE, X, E)
    loss.backward()
    oX, oE = pX.detach().clone().requires_grad_(True), pE.detach().clone().requires_grad_(True)
    ol = O_heads.reconstruction_loss(oX, oE, X, E, [1, 5]); ol.backward()
    close(ol, loss, what="recon loss"); close(oX.grad, pX.grad, what="recon dX"); close(oE.grad, pE.grad, what="recon dE")
    # per-class BCE trackers
    full = ref_loader.load_full()
    from mb_vae_dti.training.metrics.molecular_metrics import CEPerClass
    bce_X = torch.stack([CEPerClass(c)(pX.detach(), X) for c in range(cx)])
    bce_E = torch.stack([CEPerClass(c)(pE.detach(), E) for c in range(ce)])
    tX, nX = O_heads.per_class_bce(pX.detach(), X); tE, nE = O_heads.per_class_bce(pE.detach(), E)
    close(tX / nX, bce_X, what="bce X"); close(tE / nE, bce_E, what="bce E")
    rec.update(rc_true_X=X, rc_true_E=E, rc_mask=mask, rc_pred_X=pX.detach(), rc_pred_E=pE.detach(), rc_loss=loss.detach(),
               rc_dX=pX.grad, rc_dE=pE.grad, rc_bce_X=bce_X, rc_bce_E=bce_E)

    # KL head
    torch.manual_seed(3)
    kl = ref.heads.KLVariationalHead(32, 32)
    x = torch.randn(6, 32, generator=g)
    torch.manual_seed(9); z, mu, lv = kl(x)
    torch.manual_seed(9); eps = torch.randn_like(mu)
    oz, omu, olv = O_heads.kl_head(kl.state_dict(), x, eps)
    close(oz, z, what="kl z"); close(O_heads.kl_divergence(omu, olv), kl.kl_divergence(mu, lv), what="kl")
    rec.update(kl_sd_w_mu=kl.fc_mu.weight, kl_sd_b_mu=kl.fc_mu.bias, kl_sd_w_lv=kl.fc_logvar.weight, kl_sd_b_lv=kl.fc_logvar.bias,
               kl_x=x, kl_eps=eps, kl_z=z, kl_mu=mu, kl_logvar=lv, kl_value=kl.kl_divergence(mu, lv))
    npz("heads", **rec)

    # encoders, aggregators, fusion, DTI head, InfoNCE: state dicts + in/out (+ input grads)
    torch.manual_seed(5)
    mods = {
        "resenc": ref.blocks.ResidualEncoder(40, 16, 24, n_layers=2, factor=4, dropout=0.1).eval(),
        "trfenc": ref.blocks.TransformerEncoder(40, 16, 24, n_layers=2, factor=4, dropout=0.1).eval(),
        "concat": ref.blocks.ConcatAggregator(16, 24, 3).eval(),
        "attagg": ref.blocks.AttentiveAggregator(16, 24, 3).eval(),
        "fusion": ref.blocks.CrossAttentionFusion(24, 16, n_layers=2, factor=4, dropout=0.1).eval(),
        "dtihead": ref.heads.DTIHead(16, 8, 0.1, True, nn.ReLU()).eval(),
        "infonce": ref.heads.InfoNCEHead(24, 8, 0.1, True, nn.ReLU()).eval(),
    }
    B = 6
    enc = {}
    xin = torch.randn(B, 40, generator=g)
    feats = [torch.randn(B, 16, generator=g) for _ in range(3)]
    a, b = torch.randn(B, 24, generator=g), torch.randn(B, 24, generator=g)
    fp = (torch.rand(B, 64, generator=g) < 0.2).float()
    h16 = torch.randn(B, 16, generator=g)
    with torch.no_grad():
        enc["resenc_out"] = mods["resenc"](xin); close(O_heads.residual_encoder(mods["resenc"].state_dict(), xin), enc["resenc_out"], what="resenc")
        enc["trfenc_out"] = mods["trfenc"](xin); close(O_heads.transformer_encoder(mods["trfenc"].state_dict(), xin), enc["trfenc_out"], what="trfenc")
        enc["concat_out"] = mods["concat"](feats); close(O_heads.concat_aggregator(mods["concat"].state_dict(), feats), enc["concat_out"], what="concat")
        o, w = mods["attagg"](feats); enc["attagg_out"], enc["attagg_w"] = o, w
        oo, ow = O_heads.attentive_aggregator(mods["attagg"].state_dict(), feats); close(oo, o, what="attagg"); close(ow, w, what="attagg w")
        o, w = mods["fusion"](a, b); enc["fusion_out"], enc["fusion_w"] = o, w
        oo, ow = O_heads.cross_attention_fusion(mods["fusion"].state_dict(), a, b); close(oo, o, what="fusion"); close(ow, w, what="fusion w")
        d = mods["dtihead"](h16); od = O_heads.dti_head(mods["dtihead"].state_dict(), h16)
        for k in d:
            enc["dtihead_" + k] = d[k]; close(od[k], d[k], what="dti " + k)
        enc["infonce_loss"] = mods["infonce"](a, fp, 0.1); close(O_heads.info_nce(mods["infonce"].state_dict(), a, fp, 0.1), enc["infonce_loss"], what="infonce")
    # gradients (values above were taken without autograd): scalar = sum(output * cotangent), cotangents stored
    def grads_of(mod, out_fn, ins, tag):
        mod.zero_grad()
        leaves = [t.clone().requires_grad_(True) for t in ins]
        out = out_fn(leaves)
        outs = out if isinstance(out, (tuple, list)) else [out]
        cots = [torch.randn(o.shape, generator=g) for o in outs]
        sum((o * c).sum() for o, c in zip(outs, cots)).backward()
        for i, c in enumerate(cots):
            enc[f"{tag}_cot{i}"] = c
        for i, t in enumerate(leaves):
            enc[f"{tag}_din{i}"] = t.grad.clone()
        return {k: p.grad.clone() for k, p in mod.named_parameters() if p.grad is not None}

    pg = {}
    pg["resenc"] = grads_of(mods["resenc"], lambda l: mods["resenc"](l[0]), [xin], "resenc")
    pg["trfenc"] = grads_of(mods["trfenc"], lambda l: mods["trfenc"](l[0]), [xin], "trfenc")
    pg["concat"] = grads_of(mods["concat"], lambda l: mods["concat"](l), feats, "concat")
    pg["attagg"] = grads_of(mods["attagg"], lambda l: mods["attagg"](l)[0], feats, "attagg")
    pg["fusion"] = grads_of(mods["fusion"], lambda l: mods["fusion"](l[0], l[1])[0], [a, b], "fusion")
    pg["dtihead"] = grads_of(mods["dtihead"], lambda l: list(mods["dtihead"](l[0]).values()), [h16], "dtihead")
    pg["infonce"] = grads_of(mods["infonce"], lambda l: mods["infonce"](l[0], fp, 0.1), [a], "infonce")
    # the oracle's autograd must give the same parameter gradients
    for nm, fn, ins in (("resenc", lambda p_, l: O_heads.residual_encoder(p_, l[0]), [xin]),
                        ("concat", lambda p_, l: O_heads.concat_aggregator(p_, l), feats),
                        ("fusion", lambda p_, l: O_heads.cross_attention_fusion(p_, l[0], l[1])[0], [a, b])):
        p_ = {k: v.detach().clone().requires_grad_(True) for k, v in mods[nm].state_dict().items()}
        out = fn(p_, ins)
        (out * enc[f"{nm}_cot0"]).sum().backward()
        for k, gr in pg[nm].items():
            close(p_[k].grad, gr, tol=2e-5, what=f"oracle grad {nm}.{k}")
    enc.update({f"pg_{nm}": d for nm, d in pg.items()})
    targets = {"Y": (torch.rand(B, generator=g) < 0.3).float(), "Y_pKd": torch.randn(B, generator=g) + 6,
               "Y_pKi": torch.randn(B, generator=g) + 6, "Y_KIBA": torch.randn(B, generator=g) + 11}
    masks = {"Y_pKd": torch.tensor([1, 0, 1, 1, 0, 0]).bool(), "Y_pKi": torch.zeros(B).bool(), "Y_KIBA": torch.tensor([0, 1, 1, 0, 1, 1]).bool()}
    acc, comps = O_heads.dti_accuracy_loss(d, targets, masks, [1.0, 0.903964, 0.282992, 0.755172])
    enc.update(xin=xin, f0=feats[0], f1=feats[1], f2=feats[2], a=a, b=b, fp=fp, h16=h16)
    npz("encoders", **enc, **{f"sd_{nm}": {k: v for k, v in m.state_dict().items()} for nm, m in mods.items()},
        tgt=targets, msk=masks, dti_acc=acc)


# =============================================================================== G11 posteriors / ELBO pieces
def gen_posterior(model, infos, X, E, mask, rec):
    mx = torch.tensor(infos["dataset"]["node_marginals"]); me = torch.tensor(infos["dataset"]["edge_marginals"])
    sch = O_dif.Schedule(500)
    bs, n = mask.shape
    X_t, E_t = rec["eval_X_t"], rec["eval_E_t"]
    noise = {k: rec["eval_" + k] for k in ("t", "beta_t", "alpha_s_bar", "alpha_t_bar")}
    tm = model.transition_model
    Qtb = tm.get_Qt_bar(noise["alpha_t_bar"], "cpu"); Qsb = tm.get_Qt_bar(noise["alpha_s_bar"], "cpu"); Qt_ = tm.get_Qt(noise["beta_t"], "cpu")
    g = torch.Generator().manual_seed(14)
    lX = torch.randn(bs, n, 8, generator=g); lE = torch.randn(bs, n, n, 5, generator=g); lE = (lE + lE.transpose(1, 2)) / 2
    out = {}
    pt = ref.dutils.compute_posterior_distribution(E, E_t, Qt_.E, Qsb.E, Qtb.E)
    close(O_dif.posterior(E, E_t, O_dif.Qt(noise["beta_t"], me), O_dif.Qt_bar(noise["alpha_s_bar"], me), O_dif.Qt_bar(noise["alpha_t_bar"], me)), pt, what="posterior E")
    out["post_true_E"] = pt
    pb = ref.dutils.compute_batched_over0_posterior_distribution(X_t, Qt_.X, Qsb.X, Qtb.X)
    close(O_dif.posterior_over_x0(X_t, O_dif.Qt(noise["beta_t"], mx), O_dif.Qt_bar(noise["alpha_s_bar"], mx), O_dif.Qt_bar(noise["alpha_t_bar"], mx)), pb, what="batched posterior X")
    out["post_over0_X"] = pb
    # kl_prior
    dc = ref.dc
    bd = dc.BatchData(graph_data=dc.GraphData(X=X, E=E, node_mask=mask, X_t=X_t, E_t=E_t, noise_params=noise))
    klp = model.kl_prior(bd)
    close(O_dif.kl_prior(sch, mx, me, X, E, mask), klp, what="kl_prior")
    out["kl_prior"] = klp
    # compute_Lt
    pd = dc.PredictionData(graph_reconstruction=ref.dutils.PlaceHolder(X=lX, E=lE, y=None))
    model.val_X_kl.reset(); model.val_E_kl.reset()
    Lt = model.compute_Lt(bd, pd, test=False)
    *_, oLt = O_dif.compute_Lt_terms(500, mx, me, X, E, X_t, E_t, lX, lE, mask, noise)
    close(oLt, Lt, tol=1e-5, what="Lt")
    out["Lt"] = Lt
    # reverse-step probabilities + sampling
    pX, pE = O_dif.reverse_step_probs(lX, lE, X_t, E_t, noise["beta_t"], noise["alpha_s_bar"], noise["alpha_t_bar"], mx, me)
    out.update(rev_prob_X=pX, rev_prob_E=pE)
    npz("posterior", X=X, E=E, mask=mask, X_t=X_t, E_t=E_t, logits_X=lX, logits_E=lE, mx=mx, me=me,
        **{"n_" + k: v for k, v in noise.items()}, **out)


# =============================================================================== G12 reverse step with the network, G13 common step
def gen_full_steps():
    model, kw, infos = make_full_model(phase="pretrain_drug", seed=1)
    model.eval()
    mx = torch.tensor(infos["dataset"]["node_marginals"]); me = torch.tensor(infos["dataset"]["edge_marginals"])
    g = torch.Generator().manual_seed(40)
    bs, n, cx, ce = 4, 9, 8, 5
    X, E, mask = ragged_graph(bs, n, [9, 7, 5, 3], cx, ce, g)
    sch = O_dif.Schedule(500)
    gen_ = infos["general"]
    dec_sd = {k[len("drug_decoder."):]: v.detach().clone() for k, v in model.state_dict().items() if k.startswith("drug_decoder.")}
    gtk = kw["graph_transformer_kwargs"]

    # ---- one reverse step from z_t (seeded)
    torch.manual_seed(50)
    eX = torch.empty(bs * n, cx).exponential_(1); eE = torch.empty(bs * n * n, ce).exponential_(1)
    X_t, E_t = O_dif.sample_limit(mx, me, mask, eX, eE)
    emb = torch.randn(bs, 32, generator=g)
    s = torch.full((bs, 1), 498.0) / 500; t = torch.full((bs, 1), 499.0) / 500
    torch.manual_seed(51)
    with torch.no_grad():
        zs = model.sample_p_zs_given_zt(s, t, X_t, E_t, emb, mask)
    torch.manual_seed(51)
    lcc = torch.randn(bs, n)
    rX = torch.empty(bs * n, cx).exponential_(1); rE = torch.empty(bs * n * n, ce).exponential_(1)
    with torch.no_grad():
        Xe, ye = O_feat.extra_features(X_t, E_t, t, mask, lcc, gen_["max_n_nodes"], gen_["atom_valencies"], gen_["atom_weights"], gen_["max_weight"])
        oX, oE, oy = O_den.graph_transformer(dec_sd, torch.cat([X_t, Xe], 2), E_t, torch.cat([emb, ye], 1), mask,
                                             gtk["n_layers"], gtk["hidden_dims"]["n_head"], gtk["output_dims"])
        pX, pE = O_dif.reverse_step_probs(oX, oE, X_t, E_t, sch.beta(t_normalized=t), sch.alpha_bar(t_normalized=s),
                                          sch.alpha_bar(t_normalized=t), mx, me)
        Xc, Ec = O_dif.sample_graph(pX, pE, mask, rX, rE)
        Xs, Es = O_dif.one_hot_masked(Xc, Ec, mask, cx, ce)
    assert torch.equal(Xs, zs.X) and torch.equal(Es, zs.E), "reverse step mismatch"
    rec = dict(X_t=X_t, E_t=E_t, emb=emb, mask=mask, s=s, t=t, lcc_noise=lcc, expX=rX, expE=rE, X_s=zs.X, E_s=zs.E,
               X_extra=Xe, y_extra=ye, logits_X=oX, logits_E=oE, prob_X=pX, prob_E=pE, mx=mx, me=me)

    # ---- _common_step("train") in eval-dropout mode on a synthetic batch bypassing to_dense
    model2, kw2, infos2 = make_full_model(phase="train", seed=2)
    B = bs
    feats_d = {"FP-Morgan": (torch.rand(B, 64, generator=g) < 0.15).float(), "EMB-BiomedGraph": torch.randn(B, 24, generator=g)}
    feats_t = {"FP-ESP": (torch.rand(B, 48, generator=g) < 0.15).float(), "EMB-ESM": torch.randn(B, 40, generator=g), "EMB-NT": torch.randn(B, 20, generator=g)}
    ydict = {"Y": (torch.rand(B, generator=g) < 0.4).float(), "Y_pKd": torch.randn(B, generator=g) + 6, "Y_pKi": torch.randn(B, generator=g) + 6,
             "Y_KIBA": torch.randn(B, generator=g) + 11, "Y_pKd_mask": torch.tensor([True, False, True, True]),
             "Y_pKi_mask": torch.tensor([False, True, False, True]), "Y_KIBA_mask": torch.tensor([True, True, False, False])}
    batch = {"id": torch.arange(B), "y": ydict, "drug": {"features": feats_d, "representations": {"smiles": ["C"] * B}, "G": None},
             "target": {"features": feats_t}}
    dc = ref.dc
    model2._extract_graph_data = lambda b: dc.GraphData(X=X.clone(), E=E.clone(), node_mask=mask)
    # eval() freezes dropout; force the train-time lower bound on t by toggling only `training` of the top module
    model2.eval()
    object.__setattr__(model2, "training", True)
    torch.manual_seed(60)
    bd, ed, pdta, ld = model2._common_step(batch, "train")
    total = ld.compute_loss(model2.weights)
    total.backward()
    gn = {k: p.grad.norm() for k, p in model2.named_parameters() if p.grad is not None}
    gsel = {k: p.grad.detach().clone() for k, p in model2.named_parameters()
            if p.grad is not None and (k.endswith("self_attn.e_mul.weight") or k.endswith("self_attn.q.weight") or
                                       k.endswith("mlp_in_E.0.weight") or k.endswith("mlp_out_E.2.weight") or k.startswith("drug_kl_head"))}
    # replay RNG stream: randint(t) -> exp X -> exp E -> randn lcc -> randn_like eps
    torch.manual_seed(60)
    t_int = torch.randint(0, 501, size=(B, 1)).float()
    aX = torch.empty(B * n, cx).exponential_(1); aE = torch.empty(B * n * n, ce).exponential_(1)
    lcc2 = torch.randn(B, n)
    eps = torch.randn(B, 32)
    assert torch.equal(t_int, bd.graph_data.noise_params["t_int"])
    object.__setattr__(model2, "training", False)
    rec2 = dict(X=X, E=E, mask=mask, t_int=t_int, expX=aX, expE=aE, lcc_noise=lcc2, eps=eps,
                X_t=bd.graph_data.X_t.float(), E_t=bd.graph_data.E_t.float(), X_extra=bd.graph_data.X_extra, y_extra=bd.graph_data.y_extra,
                drug_embedding=ed.drug_embedding, drug_mu=ed.drug_mu, drug_logvar=ed.drug_logvar, target_embedding=ed.target_embedding,
                fused=ed.fused_embedding, pred_X=pdta.graph_reconstruction.X, pred_E=pdta.graph_reconstruction.E,
                loss_accuracy=ld.accuracy, loss_complexity=ld.complexity, loss_contrastive=ld.contrastive,
                loss_reconstruction=ld.reconstruction, loss_total=total.detach(),
                feat_d=feats_d, feat_t=feats_t, y=ydict, sd={k: v for k, v in model2.state_dict().items()},
                gradnorm=gn, grad=gsel, weights=torch.tensor(model2.weights), mx=torch.tensor(infos2["dataset"]["node_marginals"]),
                me=torch.tensor(infos2["dataset"]["edge_marginals"]))
    close(ed.drug_mu + eps * torch.exp(0.5 * ed.drug_logvar), ed.drug_embedding, what="eps replay")
    npz("reverse_step", **rec, sd=dec_sd)
    npz("common_step", **rec2)


# =============================================================================== validation ELBO step, sampling trajectory
def _record_extra_features(model):
    """Wraps model.get_extra_features so that every call's outputs are kept (eigenvector columns are gauge dependent:
    a replay must inject them)."""
    calls = []
    orig = model.get_extra_features

    def wrapped(*a, **k):
        out = orig(*a, **k)
        calls.append((out[0].detach().clone(), out[1].detach().clone()))
        return out

    model.get_extra_features = wrapped
    return calls


def gen_val_step():
    model, kw, infos = make_full_model(phase="pretrain_drug", seed=3)
    model.eval()
    mx = torch.tensor(infos["dataset"]["node_marginals"]); me = torch.tensor(infos["dataset"]["edge_marginals"])
    g = torch.Generator().manual_seed(71)
    bs, n, cx, ce = 4, 9, 8, 5
    X, E, mask = ragged_graph(bs, n, [9, 7, 5, 3], cx, ce, g)
    feats = {"FP-Morgan": (torch.rand(bs, 64, generator=g) < 0.15).float(), "EMB-BiomedGraph": torch.randn(bs, 24, generator=g)}
    batch = {"id": torch.arange(bs), "features": feats, "representations": {"smiles": ["C"] * bs}, "G": None}
    dc = ref.dc
    model._extract_graph_data = lambda b: dc.GraphData(X=X.clone(), E=E.clone(), node_mask=mask)
    calls = _record_extra_features(model)
    probs = {}
    orig_logp = model.reconstruction_logp

    def logp(*a, **k):
        out = orig_logp(*a, **k)
        probs["X"], probs["E"] = out[0].detach().clone(), out[1].detach().clone()
        return out

    model.reconstruction_logp = logp
    torch.manual_seed(70)
    with torch.no_grad():
        bd, ed, pdta, ld = model._common_step(batch, "val")
    total = ld.compute_loss(model.weights)
    # the same RNG stream, drawn by hand: t, Exp(1) X / E, lcc noise, eps, then the t = 0 sample's Exp(1) X / E and lcc noise
    torch.manual_seed(70)
    t_int = torch.randint(1, 501, size=(bs, 1)).float()
    eX = torch.empty(bs * n, cx).exponential_(1); eE = torch.empty(bs * n * n, ce).exponential_(1)
    lcc = torch.randn(bs, n)
    eps = torch.randn(bs, 32)
    eX0 = torch.empty(bs * n, cx).exponential_(1); eE0 = torch.empty(bs * n * n, ce).exponential_(1)
    lcc0 = torch.randn(bs, n)
    gd = bd.graph_data
    assert torch.equal(t_int, gd.noise_params["t_int"]) and len(calls) == 2
    close(ed.drug_mu + eps * torch.exp(0.5 * ed.drug_logvar), ed.drug_embedding, what="eps replay (val)")
    # oracle restatement of the whole estimator on the same draws (eigenvector features injected from the reference)
    sch = O_dif.Schedule(500)
    gen_ = infos["general"]
    dec_sd = {k[len("drug_decoder."):]: v.detach().clone() for k, v in model.state_dict().items() if k.startswith("drug_decoder.")}
    gtk = kw["graph_transformer_kwargs"]
    oXt, oEt, onp, _, _ = O_dif.apply_noise(sch, mx, me, X, E, mask, t_int, eX, eE)
    assert torch.equal(oXt, gd.X_t.float()) and torch.equal(oEt, gd.E_t.float())

    def o_denoise(Xn, En, tt, noise, ref_extra, trunc=False):
        Xe, ye = O_feat.extra_features(Xn, En, tt, mask, noise, gen_["max_n_nodes"], gen_["atom_valencies"], gen_["atom_weights"],
                                       gen_["max_weight"], truncate_weight=trunc)
        close(Xe[..., :4], ref_extra[0][..., :4], what="val X_extra"); close(ye, ref_extra[1], what="val y_extra")
        Xe = torch.cat((Xe[..., :4], ref_extra[0][..., 4:6], Xe[..., 6:]), -1)
        oX, oE, _ = O_den.graph_transformer(dec_sd, torch.cat([Xn, Xe], 2), En, torch.cat([ed.drug_embedding, ye], 1), mask,
                                            gtk["n_layers"], gtk["hidden_dims"]["n_head"], gtk["output_dims"])
        return oX, oE

    with torch.no_grad():
        lX, lE = o_denoise(oXt, oEt, onp["t"], lcc, calls[0], trunc=True)       # (apply_noise hands over int64 one-hots)
        close(lX, pdta.graph_reconstruction.X, tol=2e-5, what="val logits X"); close(lE, pdta.graph_reconstruction.E, tol=2e-5, what="val logits E")
        pX0, pE0, X0s, E0s = O_dif.reconstruction_probs(sch, mx, me, X, E, mask, eX0, eE0,
                                                        lambda a, b, t0: o_denoise(a, b, t0, lcc0, calls[1]))
        close(pX0, probs["X"], tol=2e-5, what="probX0"); close(pE0, probs["E"], tol=2e-5, what="probE0")
        nll, parts = O_dif.val_nll(sch, mx, me, model.nodes_dist.prob, X, E, oXt, oEt, lX, lE, mask, onp, pX0, pE0)
    close(nll, ld.reconstruction, tol=2e-5, what="val NLL")
    close(parts["loss_term_0"], ld.components["loss_term_0"], tol=2e-5, what="loss_term_0")
    close(parts["kl_prior"].mean(), ld.components["kl prior"], what="kl prior"); close(parts["log_pN"].mean(), ld.components["log_pn"], what="log pN")
    sd = {k: v for k, v in model.state_dict().items()}
    npz("val_step", sd=sd, X=X, E=E, mask=mask, feat=feats, t_int=t_int, expX=eX, expE=eE, lcc_noise=lcc, eps=eps,
        expX0=eX0, expE0=eE0, lcc_noise0=lcc0, X_extra=calls[0][0], y_extra=calls[0][1], X_extra0=calls[1][0], y_extra0=calls[1][1],
        X_t=gd.X_t.float(), E_t=gd.E_t.float(), pred_X=pdta.graph_reconstruction.X, pred_E=pdta.graph_reconstruction.E,
        probX0=probs["X"], probE0=probs["E"], nll=ld.reconstruction, kl_prior=ld.components["kl prior"],
        Lt=ld.components["Estimator loss terms"], log_pn=ld.components["log_pn"], loss_term_0=ld.components["loss_term_0"],
        loss_total=total, loss_complexity=ld.complexity, loss_contrastive=ld.contrastive,
        epoch_NLL=model.val_nll.compute(), epoch_X_kl=model.val_X_kl.compute() * model.T, epoch_E_kl=model.val_E_kl.compute() * model.T,
        epoch_X_logp=model.val_X_logp.compute(), epoch_E_logp=model.val_E_logp.compute(),
        mx=mx, me=me, weights=torch.tensor(model.weights))


def gen_trajectory():
    """Five reverse steps of sample_batch (T = 5), every step's collapsed class ids; the loop of the reference's
    sample_batch (full.py:997-1027) is restated here around its own sample_discrete_feature_noise / sample_p_zs_given_zt so
    that the per-step random draws and eigenvector features can be recorded."""
    T = 5
    model, kw, infos = make_full_model(phase="pretrain_drug", seed=4, T=T)
    model.eval()
    mx = torch.tensor(infos["dataset"]["node_marginals"]); me = torch.tensor(infos["dataset"]["edge_marginals"])
    g = torch.Generator().manual_seed(81)
    n_nodes = torch.tensor([9, 4, 6, 7, 2])
    bs, n, cx, ce = 5, 9, 8, 5
    emb = torch.randn(bs, 32, generator=g)
    mask = torch.arange(n).unsqueeze(0) < n_nodes.unsqueeze(1)
    calls = _record_extra_features(model)
    torch.manual_seed(80)
    with torch.no_grad():
        zT = ref.dutils.sample_discrete_feature_noise(limit_dist=model.limit_dist, node_mask=mask)
        Xr, Er = zT.X, zT.E
        traj = []
        for s_int in reversed(range(T)):
            s_arr = s_int * torch.ones((bs, 1))
            z = model.sample_p_zs_given_zt(s_arr / T, (s_arr + 1) / T, Xr, Er, emb, mask)
            Xr, Er = z.X, z.E
            c = ref.dutils.PlaceHolder(X=Xr.clone(), E=Er.clone(), y=None).mask(mask, collapse=True)
            traj.append((c.X.clone(), c.E.clone()))
    # replay of the RNG stream + oracle restatement
    torch.manual_seed(80)
    eXs, eEs, lccs = [], [], []
    eXs.append(torch.empty(bs * n, cx).exponential_(1)); eEs.append(torch.empty(bs * n * n, ce).exponential_(1))
    sch = O_dif.Schedule(T)
    gen_ = infos["general"]
    dec_sd = {k[len("drug_decoder."):]: v.detach().clone() for k, v in model.state_dict().items() if k.startswith("drug_decoder.")}
    gtk = kw["graph_transformer_kwargs"]
    Xo, Eo = O_dif.sample_limit(mx, me, mask, eXs[0], eEs[0])
    assert torch.equal(Xo, zT.X) and torch.equal(Eo, zT.E)
    with torch.no_grad():
        for k, s_int in enumerate(reversed(range(T))):
            lccs.append(torch.randn(bs, n))
            eXs.append(torch.empty(bs * n, cx).exponential_(1)); eEs.append(torch.empty(bs * n * n, ce).exponential_(1))
            s = torch.full((bs, 1), float(s_int)) / T; t = torch.full((bs, 1), float(s_int + 1)) / T
            Xe, ye = O_feat.extra_features(Xo, Eo, t, mask, lccs[-1], gen_["max_n_nodes"], gen_["atom_valencies"], gen_["atom_weights"],
                                           gen_["max_weight"])
            close(ye, calls[k][1], what=f"traj y_extra {k}")
            Xe = torch.cat((Xe[..., :4], calls[k][0][..., 4:6], Xe[..., 6:]), -1)
            oX, oE, _ = O_den.graph_transformer(dec_sd, torch.cat([Xo, Xe], 2), Eo, torch.cat([emb, ye], 1), mask,
                                                gtk["n_layers"], gtk["hidden_dims"]["n_head"], gtk["output_dims"])
            pX, pE = O_dif.reverse_step_probs(oX, oE, Xo, Eo, sch.beta(t_normalized=t), sch.alpha_bar(t_normalized=s),
                                              sch.alpha_bar(t_normalized=t), mx, me)
            Xc, Ec = O_dif.sample_graph(pX, pE, mask, eXs[-1], eEs[-1])
            Xo, Eo = O_dif.one_hot_masked(Xc, Ec, mask, cx, ce)
            cX = torch.where(mask, Xc, torch.full_like(Xc, -1))
            cE = torch.where(mask.unsqueeze(1) & mask.unsqueeze(2), Ec, torch.full_like(Ec, -1))
            assert torch.equal(cX, traj[k][0]) and torch.equal(cE, traj[k][1]), f"trajectory step {k}: oracle != reference"
    sd = {k: v for k, v in model.state_dict().items()}
    rec = dict(sd=sd, emb=emb, n_nodes=n_nodes, mask=mask, mx=mx, me=me, T=torch.tensor(T), z_T_X=zT.X, z_T_E=zT.E)
    for k in range(T + 1):
        rec[f"expX_{k}"], rec[f"expE_{k}"] = eXs[k], eEs[k]
    for k in range(T):
        rec[f"lcc_{k}"], rec[f"eigvec_{k}"] = lccs[k], calls[k][0][..., 4:6]
        rec[f"traj_X_{k}"], rec[f"traj_E_{k}"] = traj[k]
    npz("trajectory", **rec)


# =============================================================================== non-diffusion model variants
def gen_variants():
    import mb_vae_dti.training.modules.baseline as m_base
    import mb_vae_dti.training.modules.multi_modal as m_modal
    import mb_vae_dti.training.modules.multi_output as m_out
    import mb_vae_dti.training.modules.multi_hybrid as m_hyb
    torch.set_float32_matmul_precision("highest")
    g = torch.Generator().manual_seed(90)
    B = 6
    feats_d = {"FP-Morgan": (torch.rand(B, 64, generator=g) < 0.15).float(), "EMB-BiomedGraph": torch.randn(B, 24, generator=g),
               "EMB-BiomedText": torch.randn(B, 20, generator=g)}
    feats_t = {"FP-ESP": (torch.rand(B, 48, generator=g) < 0.15).float(), "EMB-ESM": torch.randn(B, 40, generator=g)}
    ydict = {"Y": (torch.rand(B, generator=g) < 0.4).float(), "Y_pKd": torch.randn(B, generator=g) + 6, "Y_pKi": torch.randn(B, generator=g) + 6,
             "Y_KIBA": torch.randn(B, generator=g) + 11, "Y_pKd_mask": torch.tensor([True, False, True, True, False, True]),
             "Y_pKi_mask": torch.tensor([False, True, False, True, False, False]), "Y_KIBA_mask": torch.tensor([True, True, False, False, True, True])}
    batch = {"id": torch.arange(B), "y": {k: v.clone() for k, v in ydict.items()},
             "drug": {"features": feats_d, "representations": {"smiles": ["C"] * B}}, "target": {"features": feats_t}}
    common = dict(learning_rate=1e-3, weight_decay=1e-4, scheduler="const", embedding_dim=32, hidden_dim=16, factor=4, n_layers=2,
                  activation="relu", dropout=0.1, bias=True)
    cases = {
        "baseline": (m_base.BaselineDTIModel, dict(phase="finetune", finetune_score="Y_pKd", drug_features={"FP-Morgan": 64},
                                                   target_features={"FP-ESP": 48}, encoder_type="transformer", **common)),
        "baseline_resnet": (m_base.BaselineDTIModel, dict(phase="finetune", finetune_score="Y_KIBA", drug_features={"FP-Morgan": 64},
                                                          target_features={"FP-ESP": 48}, encoder_type="resnet", **common)),
        "multi_modal": (m_modal.MultiModalDTIModel, dict(phase="finetune", finetune_score="Y_pKd",
                                                         drug_features={"FP-Morgan": 64, "EMB-BiomedGraph": 24, "EMB-BiomedText": 20},
                                                         target_features={"FP-ESP": 48, "EMB-ESM": 40}, encoder_type="resnet",
                                                         aggregator_type="concat", **common)),
        "multi_output": (m_out.MultiOutputDTIModel, dict(phase="train", finetune_score=None, drug_features={"FP-Morgan": 64},
                                                         target_features={"FP-ESP": 48}, encoder_type="transformer",
                                                         dti_weights=[1.0, 0.903964, 0.282992, 0.755172], **common)),
        "multi_hybrid": (m_hyb.MultiHybridDTIModel, dict(phase="train", finetune_score=None, weights=[1.0, 0.0, 0.5, 0.0],
                                                         dti_weights=[1.0, 0.903964, 0.282992, 0.755172], contrastive_temp=0.1,
                                                         drug_features={"FP-Morgan": 64, "EMB-BiomedGraph": 24, "EMB-BiomedText": 20},
                                                         target_features={"FP-ESP": 48, "EMB-ESM": 40}, encoder_type="resnet",
                                                         aggregator_type="concat", **common)),
    }
    rec = dict(feat_d=feats_d, feat_t=feats_t, y=ydict)
    for seed, (name, (cls, kw)) in enumerate(cases.items()):
        torch.manual_seed(100 + seed)
        model = cls(**kw).eval()
        b = {"id": batch["id"], "y": {k: v.clone() for k, v in ydict.items()}, "drug": batch["drug"], "target": batch["target"]}
        bd, ed, pdta, ld = model._common_step(b)
        total = ld.compute_loss(model.weights) if hasattr(model, "weights") else ld.accuracy
        total.backward()
        rec[f"{name}_sd"] = {k: v.detach().clone() for k, v in model.state_dict().items()}
        rec[f"{name}_grad"] = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
        rec[f"{name}_loss"] = total.detach()
        rec[f"{name}_drug_embedding"] = ed.drug_embedding.detach()
        rec[f"{name}_target_embedding"] = ed.target_embedding.detach()
        if ed.fused_embedding is not None:
            rec[f"{name}_fused"] = ed.fused_embedding.detach()
        sc = pdta.dti_scores
        if isinstance(sc, dict):
            for k, v in sc.items():
                rec[f"{name}_score_{k}"] = v.detach()
        else:
            rec[f"{name}_score"] = sc.detach()
        if ld.contrastive is not None:
            rec[f"{name}_contrastive"] = ld.contrastive.detach()
        # oracle: encoders / aggregators / fusion / heads composed by hand
        sd = model.state_dict()

        def sub(prefix):
            return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}

        enc_fn = O_heads.residual_encoder if kw["encoder_type"] == "resnet" else O_heads.transformer_encoder
        if name.startswith("baseline") or name == "multi_output":
            od = enc_fn(sub("drug_encoder."), feats_d["FP-Morgan"]); ot = enc_fn(sub("target_encoder."), feats_t["FP-ESP"])
        else:
            od = O_heads.concat_aggregator(sub("drug_aggregator."), [enc_fn(sub(f"drug_encoders.{k}."), feats_d[k]) for k in kw["drug_features"]])
            ot = O_heads.concat_aggregator(sub("target_aggregator."), [enc_fn(sub(f"target_encoders.{k}."), feats_t[k]) for k in kw["target_features"]])
        close(od, ed.drug_embedding, what=name + " drug embedding"); close(ot, ed.target_embedding, what=name + " target embedding")
        if name.startswith("baseline") or name == "multi_modal":
            key = kw["finetune_score"]; m = ydict[key + "_mask"]
            close(F.mse_loss((od * ot).sum(-1)[m], ydict[key][m]), total, what=name + " loss")
        else:
            fused, _ = O_heads.cross_attention_fusion(sub("fusion."), od, ot)
            scores = O_heads.dti_head(sub("dti_head."), fused)
            acc, _ = O_heads.dti_accuracy_loss(scores, {k: ydict[k] for k in ("Y", "Y_pKd", "Y_pKi", "Y_KIBA")},
                                               {k: ydict[k + "_mask"] for k in ("Y_pKd", "Y_pKi", "Y_KIBA")}, kw["dti_weights"])
            if name == "multi_hybrid":
                con = (O_heads.info_nce(sub("drug_contrastive_head."), od, feats_d["FP-Morgan"], 0.1)
                       + O_heads.info_nce(sub("target_contrastive_head."), ot, feats_t["FP-ESP"], 0.1))
                close(acc + 0.5 * con, total, what=name + " loss")
            else:
                close(acc, total, what=name + " loss")
    npz("variants", **rec)


# =============================================================================== target-conditioned latent optimisation
def gen_conditioning():
    """The reference's per-latent optimisation (validating/conditioning.py:333-391: for every target and every sample an
    Adam(lr) loop of n steps on -sigmoid(dti_head(fusion(latent, target))['Y']).mean(), keeping the best latent seen
    BEFORE the step that follows it) run on the reference's own CrossAttentionFusion / DTIHead modules.  The module
    itself cannot be imported (it loads CSV files and checkpoints at import, conditioning.py:697-790), so the loop is
    restated here around the reference's model components."""
    model, kw, infos = make_full_model(phase="train", seed=2)
    model.eval()
    g = torch.Generator().manual_seed(95)
    n_targets, per_target, steps, lr = 2, 3, 12, 0.01
    targets = torch.randn(n_targets, 32, generator=g)
    latents = torch.randn(n_targets * per_target, 32, generator=g)
    best_all, scores, first_scores = [], [], []
    for ti in range(n_targets):
        tgt = targets[ti:ti + 1]
        for si in range(per_target):
            latent = latents[ti * per_target + si:ti * per_target + si + 1].clone().requires_grad_(True)
            opt = torch.optim.Adam([latent], lr=lr)
            best, best_score = latent.clone(), float("-inf")
            for step in range(steps):
                opt.zero_grad()
                fused, _ = model.fusion(latent, tgt.expand(latent.shape[0], -1))
                prob = torch.sigmoid(model.dti_head(fused)["Y"])
                (-prob.mean()).backward()
                opt.step()
                if step == 0:
                    first_scores.append(prob.item())
                if prob.item() > best_score:
                    best_score, best = prob.item(), latent.clone().detach()
            best_all.append(best); scores.append(best_score)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items() if k.startswith("fusion.") or k.startswith("dti_head.")}
    # oracle: same loop on the functional restatement
    o_best = []
    fsd = {k[len("fusion."):]: v for k, v in sd.items() if k.startswith("fusion.")}
    hsd = {k[len("dti_head."):]: v for k, v in sd.items() if k.startswith("dti_head.")}
    for ti in range(n_targets):
        for si in range(per_target):
            latent = latents[ti * per_target + si:ti * per_target + si + 1].clone().requires_grad_(True)
            opt = torch.optim.Adam([latent], lr=lr)
            best, best_score = latent.clone(), float("-inf")
            for step in range(steps):
                opt.zero_grad()
                fused, _ = O_heads.cross_attention_fusion(fsd, latent, targets[ti:ti + 1])
                prob = torch.sigmoid(O_heads.dti_head(hsd, fused)["Y"])
                (-prob.mean()).backward()
                opt.step()
                if prob.item() > best_score:
                    best_score, best = prob.item(), latent.clone().detach()
            o_best.append(best)
    close(torch.cat(o_best), torch.cat(best_all), tol=1e-5, what="optimised latents")
    npz("conditioning", sd=sd, targets=targets, latents=latents, best=torch.cat(best_all), scores=torch.tensor(scores),
        first_scores=torch.tensor(first_scores), steps=torch.tensor(steps), lr=torch.tensor(lr), per_target=torch.tensor(per_target))


if __name__ == "__main__":
    torch.set_num_threads(4)
    if "--only-new" not in sys.argv:
        gen_schedule()
        model, infos, X, E, mask, rec = gen_noise()
        gen_features(model, infos, X, E, mask, rec)
        gen_denoiser()
        gen_heads()
        gen_posterior(model, infos, X, E, mask, rec)
        gen_full_steps()
    gen_val_step()
    gen_trajectory()
    gen_variants()
    gen_conditioning()
    print("all fixtures written and oracle == reference on every one")
