"""CPU: the oracle (oracle/*.py) reproduces every golden fixture captured from the reference."""
import torch

from tests._util import assert_close, load_golden
from oracle import denoiser as O_den, diffusion as O_dif, features as O_feat, heads as O_heads


def test_schedule_known_answers():
    g = load_golden("schedule")
    sch = O_dif.Schedule(500)
    assert torch.equal(sch.betas, g["betas"]) and torch.equal(sch.alphas_bar, g["alphas_bar"])
    assert abs(float(sch.betas[0]) - 8.723023e-05) < 1e-10 and float(sch.betas[500]) == 1.0
    assert abs(float(sch.alphas_bar[250]) - 0.492289) < 1e-6
    assert torch.equal(O_dif.Qt(g["beta_t"], g["mx"]), g["Qt_X"])
    assert torch.equal(O_dif.Qt_bar(g["alpha_t_bar"], g["me"]), g["Qtb_E"])
    # t = 0 -> s = -1 wraps to alphas_bar[T]
    assert torch.equal(sch.alpha_bar(t_normalized=(g["t_int"] - 1) / 500), g["alpha_s_bar"])


def test_sampler_and_apply_noise_bit_exact():
    g = load_golden("noise")
    mask = g["mask"].bool()
    X, E = O_dif.sample_graph(g["s_probX"], g["s_probE"], mask, g["s_expX"], g["s_expE"])
    assert torch.equal(X, g["s_X"]) and torch.equal(E, g["s_E"])
    sch = O_dif.Schedule(500)
    for mode in ("train", "eval"):
        Xt, Et, npar, Xc, Ec = O_dif.apply_noise(sch, g["mx"], g["me"], g["X"], g["E"], mask, g[f"{mode}_t_int"],
                                                 g[f"{mode}_expX"], g[f"{mode}_expE"])
        assert torch.equal(Xt, g[f"{mode}_X_t"]) and torch.equal(Et, g[f"{mode}_E_t"])
        assert torch.equal(Xc, g[f"{mode}_Xc"]) and torch.equal(Ec, g[f"{mode}_Ec"])
        for k in ("t", "beta_t", "alpha_s_bar", "alpha_t_bar"):
            assert torch.equal(npar[k], g[f"{mode}_{k}"])
    lX, lE = O_dif.sample_limit(g["mx"], g["me"], mask, g["lim_expX"], g["lim_expE"])
    assert torch.equal(lX, g["lim_X"]) and torch.equal(lE, g["lim_E"])


def test_extra_features():
    g = load_golden("features")
    mask = g["mask"].bool()
    Xe, ye = O_feat.extra_features(g["X_t"], g["E_t"], g["t"], mask, g["lcc_noise"], int(g["max_n_nodes"]),
                                   g["valencies"].tolist(), g["atom_weights"].tolist(), float(g["max_weight"]))
    assert_close(Xe[..., :4], g["X_extra"][..., :4], what="cycles/lcc")
    assert_close(Xe[..., 6:], g["X_extra"][..., 6:], what="charge/valency")
    assert_close(Xe[..., 4:6].abs(), g["X_extra"][..., 4:6].abs(), rtol=1e-4, what="|eigvec|")
    assert_close(ye, g["y_extra"], what="y_extra")
    kx, ky = O_feat.cycle_features(g["dense_A"])
    assert_close(kx, g["dense_kx"]); assert_close(ky, g["dense_ky"])


def test_denoiser_forward_and_grads():
    g = load_golden("denoiser_tiny")
    mask = g["mask"].bool()
    p = {k: v.clone().requires_grad_(True) for k, v in g["sd"].items()}
    X, E, y = (g[k].clone().requires_grad_(True) for k in ("X", "E", "y"))
    oX, oE, oy = O_den.graph_transformer(p, X, E, y, mask, 2, 4, {"X": 5, "E": 5, "y": 32})
    assert_close(oX, g["out_X"]); assert_close(oE, g["out_E"]); assert_close(oy, g["out_y"])
    ((oX * g["cX"]).sum() + (oE * g["cE"]).sum() + (oy * g["cy"]).sum()).backward()
    for k, v in g["grad"].items():
        assert_close(p[k].grad, v, rtol=2e-5, what="grad " + k)
    assert_close(X.grad, g["dX"]); assert_close(E.grad, g["dE"]); assert_close(y.grad, g["dy"])


def test_layer_and_block_intermediates():
    g = load_golden("layer_tiny")
    mask = g["mask"].bool()
    X, E, y = O_den.xey_layer(g["sd"], g["X"], g["E"], g["y"], mask, 4)
    assert_close(X, g["out_X"]); assert_close(E, g["out_E"]); assert_close(y, g["out_y"])
    sub = {k[len("self_attn."):]: v for k, v in g["sd"].items() if k.startswith("self_attn.")}
    nX, nE, ny = O_den.node_edge_block(sub, g["X"], g["E"], g["y"], mask, 4)
    assert_close(nX, g["blk_X"]); assert_close(nE, g["blk_E"]); assert_close(ny, g["blk_y"])


def test_heads():
    g = load_golden("heads")
    pX, pE = g["rc_pred_X"].clone().requires_grad_(True), g["rc_pred_E"].clone().requires_grad_(True)
    loss = O_heads.reconstruction_loss(pX, pE, g["rc_true_X"], g["rc_true_E"], [1, 5])
    loss.backward()
    assert_close(loss, g["rc_loss"]); assert_close(pX.grad, g["rc_dX"]); assert_close(pE.grad, g["rc_dE"])
    tE, nE = O_heads.per_class_bce(g["rc_pred_E"], g["rc_true_E"])
    assert_close(tE / nE, g["rc_bce_E"])
    sd = {"fc_mu.weight": g["kl_sd_w_mu"], "fc_mu.bias": g["kl_sd_b_mu"], "fc_logvar.weight": g["kl_sd_w_lv"],
          "fc_logvar.bias": g["kl_sd_b_lv"]}
    z, mu, lv = O_heads.kl_head(sd, g["kl_x"], g["kl_eps"])
    assert_close(z, g["kl_z"]); assert_close(O_heads.kl_divergence(mu, lv), g["kl_value"])


def test_encoders_fusion_heads():
    g = load_golden("encoders")
    feats = [g["f0"], g["f1"], g["f2"]]
    assert_close(O_heads.residual_encoder(g["sd_resenc"], g["xin"]), g["resenc_out"])
    assert_close(O_heads.transformer_encoder(g["sd_trfenc"], g["xin"]), g["trfenc_out"])
    assert_close(O_heads.concat_aggregator(g["sd_concat"], feats), g["concat_out"])
    o, w = O_heads.attentive_aggregator(g["sd_attagg"], feats)
    assert_close(o, g["attagg_out"]); assert_close(w, g["attagg_w"])
    o, w = O_heads.cross_attention_fusion(g["sd_fusion"], g["a"], g["b"])
    assert_close(o, g["fusion_out"]); assert_close(w, g["fusion_w"])
    d = O_heads.dti_head(g["sd_dtihead"], g["h16"])
    for k in ("Y", "Y_pKd", "Y_pKi", "Y_KIBA"):
        assert_close(d[k], g["dtihead_" + k])
    assert_close(O_heads.info_nce(g["sd_infonce"], g["a"], g["fp"], 0.1), g["infonce_loss"])
    acc, _ = O_heads.dti_accuracy_loss(d, g["tgt"], {k: v.bool() for k, v in g["msk"].items()},
                                       [1.0, 0.903964, 0.282992, 0.755172])
    assert_close(acc, g["dti_acc"])


def test_posteriors_and_elbo_terms():
    g = load_golden("posterior")
    mask = g["mask"].bool()
    noise = {k: g["n_" + k] for k in ("t", "beta_t", "alpha_s_bar", "alpha_t_bar")}
    mx, me = g["mx"], g["me"]
    pt = O_dif.posterior(g["E"], g["E_t"], O_dif.Qt(noise["beta_t"], me), O_dif.Qt_bar(noise["alpha_s_bar"], me),
                         O_dif.Qt_bar(noise["alpha_t_bar"], me))
    assert_close(pt, g["post_true_E"])
    pb = O_dif.posterior_over_x0(g["X_t"], O_dif.Qt(noise["beta_t"], mx), O_dif.Qt_bar(noise["alpha_s_bar"], mx),
                                 O_dif.Qt_bar(noise["alpha_t_bar"], mx))
    assert_close(pb, g["post_over0_X"])
    assert_close(O_dif.kl_prior(O_dif.Schedule(500), mx, me, g["X"], g["E"], mask), g["kl_prior"])
    *_, Lt = O_dif.compute_Lt_terms(500, mx, me, g["X"], g["E"], g["X_t"], g["E_t"], g["logits_X"], g["logits_E"], mask, noise)
    assert_close(Lt, g["Lt"])
    pX, pE = O_dif.reverse_step_probs(g["logits_X"], g["logits_E"], g["X_t"], g["E_t"], noise["beta_t"],
                                      noise["alpha_s_bar"], noise["alpha_t_bar"], mx, me)
    assert_close(pX, g["rev_prob_X"]); assert_close(pE, g["rev_prob_E"])


def test_reverse_step_with_network():
    g = load_golden("reverse_step")
    mask = g["mask"].bool()
    sch = O_dif.Schedule(500)
    X_t, E_t, t, s = g["X_t"], g["E_t"], g["t"], g["s"]
    oX, oE, _ = O_den.graph_transformer(g["sd"], torch.cat([X_t, g["X_extra"]], 2), E_t, torch.cat([g["emb"], g["y_extra"]], 1),
                                        mask, 2, 4, {"X": 8, "E": 5, "y": 32})
    assert_close(oX, g["logits_X"]); assert_close(oE, g["logits_E"])
    pX, pE = O_dif.reverse_step_probs(oX, oE, X_t, E_t, sch.beta(t_normalized=t), sch.alpha_bar(t_normalized=s),
                                      sch.alpha_bar(t_normalized=t), g["mx"], g["me"])
    Xc, Ec = O_dif.sample_graph(pX, pE, mask, g["expX"], g["expE"])
    Xs, Es = O_dif.one_hot_masked(Xc, Ec, mask, 8, 5)
    assert torch.equal(Xs, g["X_s"]) and torch.equal(Es, g["E_s"])
