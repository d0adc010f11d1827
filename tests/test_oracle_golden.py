"""CPU: the oracle (oracle/*.py) reproduces every golden fixture captured from the reference."""
import torch

from tests._util import assert_close, load_golden
from oracle import denoiser as O_den, diffusion as O_dif, features as O_feat, heads as O_heads


def _ulp_close(a, b):
    """alphas_bar = exp(cumsum(log(1 - beta))) in fp32: torch's vectorised exp / log round differently from one CPU's
    vector ISA to the next (a few ulp at the end of a 501-term sum of magnitude 20), so the host that wrote the fixture
    and the host that runs this test agree element-wise to 5e-6, not bit for bit."""
    assert a.shape == b.shape and bool(((a.double() - b.double()).abs() <= 5e-6 * b.double().abs()).all())


def test_schedule_known_answers():
    g = load_golden("schedule")
    sch = O_dif.Schedule(500)
    assert torch.equal(sch.betas, g["betas"])
    _ulp_close(sch.alphas_bar, g["alphas_bar"])
    assert abs(float(sch.betas[0]) - 8.723023e-05) < 1e-10 and float(sch.betas[500]) == 1.0
    assert abs(float(sch.alphas_bar[250]) - 0.492289) < 1e-6
    assert torch.equal(O_dif.Qt(g["beta_t"], g["mx"]), g["Qt_X"])
    assert torch.equal(O_dif.Qt_bar(g["alpha_t_bar"], g["me"]), g["Qtb_E"])
    # t = 0 -> s = -1 wraps to alphas_bar[T]
    _ulp_close(sch.alpha_bar(t_normalized=(g["t_int"] - 1) / 500), g["alpha_s_bar"])


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
        for k in ("t", "beta_t"):
            assert torch.equal(npar[k], g[f"{mode}_{k}"])
        for k in ("alpha_s_bar", "alpha_t_bar"):
            _ulp_close(npar[k], g[f"{mode}_{k}"])
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
    assert_close(O_dif.kl_prior(O_dif.Schedule(500), mx, me, g["X"], g["E"], mask), g["kl_prior"], atol=2e-6)   # zero in exact arithmetic at T = 500: absolute bound
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


def test_encoder_fusion_head_gradients():
    """Parameter and input gradients of the encoders / aggregators / fusion / heads (autograd through the oracle)."""
    import torch.nn.functional as F  # noqa: F401
    g = load_golden("encoders")
    feats = [g["f0"], g["f1"], g["f2"]]
    cases = {
        "resenc": (lambda p, l: [O_heads.residual_encoder(p, l[0])], [g["xin"]]),
        "trfenc": (lambda p, l: [O_heads.transformer_encoder(p, l[0])], [g["xin"]]),
        "concat": (lambda p, l: [O_heads.concat_aggregator(p, l)], feats),
        "attagg": (lambda p, l: [O_heads.attentive_aggregator(p, l)[0]], feats),
        "fusion": (lambda p, l: [O_heads.cross_attention_fusion(p, l[0], l[1])[0]], [g["a"], g["b"]]),
        "dtihead": (lambda p, l: list(O_heads.dti_head(p, l[0]).values()), [g["h16"]]),
        "infonce": (lambda p, l: [O_heads.info_nce(p, l[0], g["fp"], 0.1)], [g["a"]]),
    }
    for nm, (fn, ins) in cases.items():
        p = {k: v.clone().requires_grad_(True) for k, v in g["sd_" + nm].items()}
        leaves = [t.clone().requires_grad_(True) for t in ins]
        outs = fn(p, leaves)
        sum((o * g[f"{nm}_cot{i}"]).sum() for i, o in enumerate(outs)).backward()
        for k, v in g["pg_" + nm].items():
            # softmax is shift invariant: the gradient of the attention-logit bias is exactly zero in exact arithmetic,
            # and what both sides hold is rounding noise (1e-8) -- compared absolutely
            assert_close(p[k].grad, v, rtol=2e-5, atol=1e-6 if k.endswith("to_attn_logits.bias") else None, what=f"{nm} grad {k}")
        for i, t in enumerate(leaves):
            assert_close(t.grad, g[f"{nm}_din{i}"], rtol=2e-5, what=f"{nm} input grad {i}")


def _oracle_denoise(g, sd_prefix, emb, mask, n_layers=2, n_head=4, cx=8):
    dec = {k[len(sd_prefix):]: v for k, v in g["sd"].items() if k.startswith(sd_prefix)}

    def run(Xn, En, X_extra, y_extra):
        oX, oE, _ = O_den.graph_transformer(dec, torch.cat([Xn, X_extra], 2), En, torch.cat([emb, y_extra], 1), mask, n_layers, n_head,
                                            {"X": cx, "E": 5, "y": 32})
        return oX, oE
    return run


def test_validation_elbo_step():
    """The whole NLL estimator of a validation step (noise -> denoiser -> KL prior + T*KL + t=0 reconstruction) on the
    reference's own random draws."""
    g = load_golden("val_step")
    mask = g["mask"].bool()
    sch = O_dif.Schedule(500)
    mx, me = g["mx"], g["me"]
    X_t, E_t, npar, _, _ = O_dif.apply_noise(sch, mx, me, g["X"], g["E"], mask, g["t_int"], g["expX"], g["expE"])
    assert torch.equal(X_t, g["X_t"]) and torch.equal(E_t, g["E_t"])
    enc = {k[len("drug_encoders."):]: v for k, v in g["sd"].items() if k.startswith("drug_encoders.")}
    views = [O_heads.transformer_encoder({k[len(nm) + 1:]: v for k, v in enc.items() if k.startswith(nm + ".")}, g["feat"][nm])
             for nm in ("FP-Morgan", "EMB-BiomedGraph")]
    agg, _ = O_heads.attentive_aggregator({k[len("drug_aggregator."):]: v for k, v in g["sd"].items() if k.startswith("drug_aggregator.")}, views)
    z, mu, lv = O_heads.kl_head({k[len("drug_kl_head."):]: v for k, v in g["sd"].items() if k.startswith("drug_kl_head.")}, agg, g["eps"])
    assert_close(O_heads.kl_divergence(mu, lv), g["loss_complexity"], what="KL")
    run = _oracle_denoise(g, "drug_decoder.", z, mask)
    with torch.no_grad():
        lX, lE = run(X_t, E_t, g["X_extra"], g["y_extra"])
        assert_close(lX, g["pred_X"], rtol=2e-5, what="logits X"); assert_close(lE, g["pred_E"], rtol=2e-5, what="logits E")
        pX0, pE0, _, _ = O_dif.reconstruction_probs(sch, mx, me, g["X"], g["E"], mask, g["expX0"], g["expE0"],
                                                    lambda a, b, t0: run(a, b, g["X_extra0"], g["y_extra0"]))
        assert_close(pX0, g["probX0"], rtol=2e-5, what="probX0"); assert_close(pE0, g["probE0"], rtol=2e-5, what="probE0")
        hist = torch.zeros(10); hist[[3, 5, 7, 9]] = torch.tensor([1., 2., 3., 2.]); hist = hist / hist.sum()
        nll, parts = O_dif.val_nll(sch, mx, me, hist, g["X"], g["E"], X_t, E_t, lX, lE, mask, npar, pX0, pE0)
    assert_close(nll, g["nll"], rtol=2e-5, what="NLL"); assert_close(parts["Lt"], g["Lt"], rtol=2e-5, what="Lt")
    assert_close(parts["loss_term_0"], g["loss_term_0"], rtol=2e-5, what="t=0 term")
    assert_close(parts["kl_prior"].mean(), g["kl_prior"], atol=2e-6, what="KL prior (zero in exact arithmetic)"); assert_close(parts["log_pN"].mean(), g["log_pn"], what="log pN")
    assert_close(nll, g["epoch_NLL"], rtol=2e-5, what="epoch NLL (single batch)")


def test_sampling_trajectory_five_steps():
    """sample_batch's reverse loop (T = 5): every step's class ids, bit for bit, on the reference's draws."""
    g = load_golden("trajectory")
    mask = g["mask"].bool()
    T = int(g["T"])
    sch = O_dif.Schedule(T)
    mx, me = g["mx"], g["me"]
    X, E = O_dif.sample_limit(mx, me, mask, g["expX_0"], g["expE_0"])
    assert torch.equal(X, g["z_T_X"]) and torch.equal(E, g["z_T_E"])
    run = _oracle_denoise(g, "drug_decoder.", g["emb"], mask)
    bs = mask.shape[0]
    with torch.no_grad():
        for k, s_int in enumerate(reversed(range(T))):
            s = torch.full((bs, 1), float(s_int)) / T; t = torch.full((bs, 1), float(s_int + 1)) / T
            Xe, ye = O_feat.extra_features(X, E, t, mask, g[f"lcc_{k}"], 64, [4, 2, 3, 3, 2, 1, 1, 1],
                                           [12.011, 15.999, 30.974, 14.007, 32.06, 35.45, 18.998, 1.008], 1000)
            Xe = torch.cat((Xe[..., :4], g[f"eigvec_{k}"], Xe[..., 6:]), -1)
            oX, oE = run(X, E, Xe, ye)
            pX, pE = O_dif.reverse_step_probs(oX, oE, X, E, sch.beta(t_normalized=t), sch.alpha_bar(t_normalized=s),
                                              sch.alpha_bar(t_normalized=t), mx, me)
            Xc, Ec = O_dif.sample_graph(pX, pE, mask, g[f"expX_{k + 1}"], g[f"expE_{k + 1}"])
            X, E = O_dif.one_hot_masked(Xc, Ec, mask, 8, 5)
            cX = torch.where(mask, Xc, torch.full_like(Xc, -1))
            cE = torch.where(mask.unsqueeze(1) & mask.unsqueeze(2), Ec, torch.full_like(Ec, -1))
            assert torch.equal(cX, g[f"traj_X_{k}"]) and torch.equal(cE, g[f"traj_E_{k}"]), f"step {k}"


def test_model_variants_losses():
    """Baseline / multi-modal / multi-output / multi-hybrid _common_step: embeddings, scores and losses composed from the
    oracle's encoders, aggregators, fusion and heads."""
    g = load_golden("variants")
    fd, ft, y = g["feat_d"], g["feat_t"], g["y"]
    w = [1.0, 0.903964, 0.282992, 0.755172]

    def sub(sd, prefix):
        return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}

    for name, enc_fn, multi in (("baseline", O_heads.transformer_encoder, False), ("baseline_resnet", O_heads.residual_encoder, False),
                                ("multi_modal", O_heads.residual_encoder, True), ("multi_output", O_heads.transformer_encoder, False),
                                ("multi_hybrid", O_heads.residual_encoder, True)):
        sd = g[name + "_sd"]
        if multi:
            od = O_heads.concat_aggregator(sub(sd, "drug_aggregator."), [enc_fn(sub(sd, f"drug_encoders.{k}."), fd[k])
                                                                         for k in ("FP-Morgan", "EMB-BiomedGraph", "EMB-BiomedText")])
            ot = O_heads.concat_aggregator(sub(sd, "target_aggregator."), [enc_fn(sub(sd, f"target_encoders.{k}."), ft[k])
                                                                           for k in ("FP-ESP", "EMB-ESM")])
        else:
            od, ot = enc_fn(sub(sd, "drug_encoder."), fd["FP-Morgan"]), enc_fn(sub(sd, "target_encoder."), ft["FP-ESP"])
        assert_close(od, g[name + "_drug_embedding"], what=name + " drug"); assert_close(ot, g[name + "_target_embedding"], what=name + " target")
        if name in ("baseline", "baseline_resnet", "multi_modal"):
            key = "Y_KIBA" if name == "baseline_resnet" else "Y_pKd"
            m = y[key + "_mask"].bool()
            assert_close((od * ot).sum(-1)[m], g[name + "_score"], what=name + " scores")
            assert_close(torch.nn.functional.mse_loss((od * ot).sum(-1)[m], y[key][m]), g[name + "_loss"], what=name + " loss")
        else:
            fused, _ = O_heads.cross_attention_fusion(sub(sd, "fusion."), od, ot)
            assert_close(fused, g[name + "_fused"], what=name + " fused")
            acc, _ = O_heads.dti_accuracy_loss(O_heads.dti_head(sub(sd, "dti_head."), fused), {k: y[k] for k in ("Y", "Y_pKd", "Y_pKi", "Y_KIBA")},
                                               {k: y[k + "_mask"].bool() for k in ("Y_pKd", "Y_pKi", "Y_KIBA")}, w)
            if name == "multi_hybrid":
                con = (O_heads.info_nce(sub(sd, "drug_contrastive_head."), od, fd["FP-Morgan"], 0.1)
                       + O_heads.info_nce(sub(sd, "target_contrastive_head."), ot, ft["FP-ESP"], 0.1))
                assert_close(con, g[name + "_contrastive"], what="contrastive")
                acc = acc + 0.5 * con
            assert_close(acc, g[name + "_loss"], what=name + " loss")
