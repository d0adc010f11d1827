"""Throughput of the two encoder-bearing benchmark shapes on one MI355X (or one rank per GPU under torch.distributed.run):

  encoders : multi-branch VAE encoder train step -- drug views Morgan 2048 + MMELON graph 512 / image 512 / text 768,
             target views ESP 4170 + ESM-C 1152 + NT 1024, transformer encoders (hidden 512, 3 layers, factor 4),
             attentive aggregators, KL head, cross-attention fusion, DTI head; loss = DTI (BCE + 3 masked MSE) +
             0.001 * KL; backward; AdamW(amsgrad).  256 pairs per step.
  joint    : FullDTIModel(phase="train").training_step on 256 pairs per GPU: the encoders above + forward noising +
             extra features + the 8-layer graph-transformer decoder conditioned on the drug embedding + masked CE,
             weights [1, 0.001, 0, 3]; backward; (gradient all-reduce); AdamW(amsgrad).

    python tools/bench_configs.py [--which encoders|joint|both] [--steps 5] [--warmup 2]

Mirrors mb_vae_dti/training/modules/full.py:381-448 (forward), :816-952 (_common_step), :1174-1211 (training_step) of
the reference.  Synthetic inputs, random-init weights, fp32.  Prints one JSON line per shape.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mb-vae-dti_amd"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn.functional as F  # noqa: E402

DRUG = {"FP-Morgan": 2048, "EMB-BiomedGraph": 512, "EMB-BiomedImg": 512, "EMB-BiomedText": 768}
TARGET = {"FP-ESP": 4170, "EMB-ESM": 1152, "EMB-NT": 1024}
DTI_W = [1.0, 0.903964, 0.282992, 0.755172]


def build_model(dev):
    from mbvae_amd.full import FullDTIModel
    from mbvae_amd.trainstep import ATOM_WEIGHTS, CFG3, ME, MX, VALENCIES
    infos = {"general": {"atom_types": ["C", "O", "P", "N", "S"], "max_n_nodes": 64, "atom_valencies": list(VALENCIES),
                         "atom_weights": list(ATOM_WEIGHTS), "max_weight": 1000},
             "dataset": {"node_count_distribution": {"64": 1}, "node_marginals": list(MX), "edge_marginals": list(ME)}}
    torch.manual_seed(0)
    model = FullDTIModel(
        phase="train", finetune_score=None, learning_rate=5e-4, weight_decay=1e-4, scheduler="const",
        weights=[1.0, 0.001, 0.0, 3.0], dti_weights=DTI_W, diff_weights=[1, 5], contrastive_temp=0.1,
        drug_features=DRUG, target_features=TARGET, embedding_dim=1024, hidden_dim=512, factor=4, n_layers=3,
        activation="relu", dropout=0.1, bias=True, encoder_type="transformer", aggregator_type="attentive",
        diffusion_steps=500, sample_every_val=5, val_samples_per_embedding=2, test_samples_per_embedding=2,
        graph_transformer_kwargs=CFG3, dataset_infos=infos)
    return model.to(dev).train()


def synthetic_batch(bs, n, dev, g):
    from mbvae_amd.containers import GraphData
    from mbvae_amd.trainstep import synthetic_graphs
    X0, E0, mask = synthetic_graphs(bs, n, dev, g)
    feat_d = {k: ((torch.rand(bs, d, generator=g) < 0.03).float() if k.startswith("FP") else torch.randn(bs, d, generator=g)).to(dev)
              for k, d in DRUG.items()}
    feat_t = {k: ((torch.rand(bs, d, generator=g) < 0.02).float() if k.startswith("FP") else torch.randn(bs, d, generator=g)).to(dev)
              for k, d in TARGET.items()}
    y = {"Y": (torch.rand(bs, generator=g) < 0.25).float().to(dev)}
    for k in ("Y_pKd", "Y_pKi", "Y_KIBA"):
        y[k] = (5.4 + 0.9 * torch.randn(bs, generator=g)).clamp(5, 10.8).to(dev)
        y[k + "_mask"] = (torch.rand(bs, generator=g) < 0.3).to(dev)
    return {"id": torch.arange(bs), "y": y,
            "drug": {"features": feat_d, "representations": {"smiles": ["C"] * bs}, "G": GraphData(X=X0, E=E0, node_mask=mask)},
            "target": {"features": feat_t}}


def encoder_step(model, batch, opt, static_shapes=False):
    """forward() of the encoder branches + KL + DTI losses (no decoder).  static_shapes: the masked MSE as a masked
    mean instead of boolean indexing (same value; no data-dependent shape, so the step can be captured in a HIP graph)."""
    bd = model._create_batch_data(batch)
    emb, pred = model(bd.drug_features, bd.target_features)
    kl = model.drug_kl_head.kl_divergence(emb.drug_mu, emb.drug_logvar)
    comps = [F.binary_cross_entropy_with_logits(pred.dti_scores["Y"], bd.dti_targets["Y"])]
    for name in ("Y_pKd", "Y_pKi", "Y_KIBA"):
        m = bd.dti_masks[name]
        if static_shapes:
            mf = m.float()
            comps.append((((pred.dti_scores[name] - bd.dti_targets[name]) ** 2) * mf).sum() / mf.sum().clamp(min=1.0))
        else:
            comps.append(F.mse_loss(pred.dti_scores[name][m], bd.dti_targets[name][m]))
    loss = sum(w * c for w, c in zip(DTI_W, comps)) + 0.001 * kl
    loss.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)
    return loss.detach()


def timed(fn, steps, warmup, world):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = fn()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    return el, float(loss)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--which", default="both", choices=("encoders", "joint", "both"))
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--kernel-table", action="store_true")
    ap.add_argument("--one-stream", action="store_true", help="joint: keep the decoder's node / global kernels on the main stream")
    ap.add_argument("--graph", action="store_true",
                    help="encoders only: capture the whole step (forward, backward, AdamW) in one HIP graph and time replays")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    from mbvae_amd import _lib, ops
    from mbvae_amd.ddp import BucketedGradSync
    ops.set_side_stream(not args.one_stream)     # (never active while a HIP graph is being captured)

    model = build_model(dev)
    g = torch.Generator().manual_seed(1000 + rank)
    batch = synthetic_batch(args.batch, 64, dev, g)
    torch.manual_seed(7 + rank)
    n_par = {"encoders": sum(p.numel() for k, p in model.named_parameters() if not k.startswith("drug_decoder")),
             "joint": sum(p.numel() for p in model.parameters())}

    for which in (("encoders", "joint") if args.which == "both" else (args.which,)):
        if which == "encoders":
            params = [p for k, p in model.named_parameters()
                      if not k.startswith("drug_decoder") and "contrastive" not in k]
            opt = torch.optim.AdamW(params, lr=5e-4, weight_decay=1e-4, amsgrad=True, fused=True)
            sync = BucketedGradSync(params) if world > 1 else None

            def fn():
                if sync is None:
                    return encoder_step(model, batch, opt)
                raise SystemExit("the encoder-only shape is a single-GPU case")

            if args.graph:
                # The encoder step is launch bound (~900 kernels of a few microseconds on 256 rows): one captured HIP graph
                # replays it without per-launch host work.  Its dropout is torch's Philox dropout, whose offsets a
                # captured graph advances on every replay; the library calls only launch kernels on the capture stream.
                opt = torch.optim.AdamW(params, lr=5e-4, weight_decay=1e-4, amsgrad=True, fused=True, capturable=True)
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(3):
                        encoder_step(model, batch, opt, static_shapes=True)
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                opt.zero_grad(set_to_none=True)
                with torch.cuda.graph(graph):
                    static_loss = encoder_step(model, batch, opt, static_shapes=True)

                def fn():  # noqa: F811
                    graph.replay()
                    return static_loss
        else:
            params = [p for p in model.parameters() if p.requires_grad]
            opt = torch.optim.AdamW(params, lr=5e-4, weight_decay=1e-4, amsgrad=True, fused=True)
            sync = BucketedGradSync(params) if world > 1 else None

            def fn():
                total = model.training_step(batch, 0)
                total.backward()
                if sync is not None:
                    sync.finish()
                opt.step()
                if sync is not None:
                    sync.zero_grad()
                else:
                    opt.zero_grad(set_to_none=True)
                return total.detach()

        if args.kernel_table and rank == 0:
            for _ in range(args.warmup):
                fn()
            torch.cuda.synchronize()
            _lib.PROFILE = {}
        el, loss = timed(fn, args.steps, args.warmup, world)
        if args.kernel_table and rank == 0:
            prof, _lib.PROFILE = _lib.PROFILE, None
            rows = {}
            for (name, tag), evs in prof.items():
                r = rows.setdefault(name, [0.0, 0, 0.0])
                ms = sum(a.elapsed_time(b) for a, b in evs)
                r[0] += ms
                r[1] += len(evs)
                if tag is not None and name.startswith("mbvae_linear"):
                    r[2] += 2.0 * tag[0] * tag[1] * tag[2] * len(evs)
            denom = args.steps + args.warmup
            for name, (ms, calls, fl) in sorted(rows.items(), key=lambda kv: -kv[1][0]):
                print(f"  {which:9s} {name:30s} {ms / denom:8.3f} ms/step {calls // denom:5d} calls"
                      + (f"  {fl / ms / 1e9:6.1f} TFLOP/s" if fl else ""), file=sys.stderr)
        if rank == 0:
            unit = "pairs/s"
            print(json.dumps({"workload": which + (" (one HIP graph per step)" if (args.graph and which == "encoders") else ""), "value": round(world * args.batch * args.steps / el, 1), "unit": unit,
                              "ms_per_step": round(1000 * el / args.steps, 3), "n_gpus": world, "pairs_per_gpu": args.batch,
                              "parameters": n_par[which], "dtype": "f32", "loss": round(loss, 5),
                              "data": "synthetic, random-init weights"}), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
