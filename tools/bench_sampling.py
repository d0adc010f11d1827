"""Reverse-diffusion sampling throughput (conditional generation, one chunk): molecule-steps/s on one MI355X.

    python tools/bench_sampling.py [--chunk 512] [--steps 20] [--nmax 64]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/bench_sampling.py ...   (one chunk per rank)

One step = FullDTIModel.sample_p_zs_given_zt: extra features (cycle counts, LDS Jacobi eigensolver) -> 8-layer denoiser
forward (eval) -> reverse-step posterior + categorical draw (HIP).  Molecule sizes ~ N(28, 7^2) clipped to [8, nmax],
random-init weights, synthetic conditioning embeddings; the full job is 500 steps x 10 000 molecules.
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mb-vae-dti_amd")); sys.path.insert(0, ROOT)
import torch
from mbvae_amd.full import FullDTIModel
from mbvae_amd.diffusion import sample_discrete_feature_noise

ap = argparse.ArgumentParser()
ap.add_argument("--chunk", type=int, default=512)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--nmax", type=int, default=64)
ap.add_argument("--fixed-n", type=int, default=0, help="all molecules of this size (0: N(28,7) clipped)")
ap.add_argument("--buckets", type=int, default=1, help="size-sorted sub-batches (sample_batch(size_buckets=...))")
ap.add_argument("--sort", action="store_true", help="with one bucket: still lay the molecules out by size")
ap.add_argument("--graph", action="store_true", help="capture one reverse step in a HIP graph and replay it")
ap.add_argument("--kernel-table", action="store_true", help="after the timed steps: 5 more with every library call bracketed by HIP events, per (entry point, shape) to stderr")
a = ap.parse_args()
# molecules are independent: every rank samples its own chunk (generator seeded seed + rank), no collective in the loop
rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
torch.cuda.set_device(dev)
if world > 1:
    import torch.distributed as dist
    dist.init_process_group("nccl", device_id=dev)
cx, emb = 8, 1024
mx = [0.72, 0.12, 0.06, 0.04, 0.02, 0.02, 0.01, 0.01]
me = [0.90, 0.05, 0.03, 0.01, 0.01]
infos = {"general": {"atom_types": ["C", "O", "P", "N", "S", "Cl", "F", "H"], "max_n_nodes": a.nmax,
                     "atom_valencies": [4, 2, 3, 3, 2, 1, 1, 1],
                     "atom_weights": [12.011, 15.999, 30.974, 14.007, 32.06, 35.45, 18.998, 1.008], "max_weight": 1000},
         "dataset": {"node_count_distribution": {str(k): 1 for k in range(8, a.nmax + 1)}, "node_marginals": mx, "edge_marginals": me}}
gt = dict(n_layers=8, input_dims={"X": cx + 8, "E": 5, "y": emb + 13}, output_dims={"X": cx, "E": 5, "y": emb},
          hidden_mlp_dims={"X": 256, "E": 128, "y": 2048},
          hidden_dims={"dx": 256, "de": 64, "dy": 1024, "n_head": 8, "dim_ffX": 256, "dim_ffE": 128, "dim_ffy": 1024})
torch.manual_seed(0)
model = FullDTIModel(phase="pretrain_drug", finetune_score=None, learning_rate=5e-4, weight_decay=1e-4, scheduler="const",
                     weights=[1.0, 0.001, 0.0, 3.0], dti_weights=None, diff_weights=[1, 5], contrastive_temp=0.1,
                     drug_features={"FP-Morgan": 2048, "EMB-BiomedGraph": 512}, target_features={}, embedding_dim=emb, hidden_dim=512,
                     factor=4, n_layers=3, activation="relu", dropout=0.1, bias=True, encoder_type="transformer",
                     aggregator_type="attentive", diffusion_steps=500, sample_every_val=5, val_samples_per_embedding=2,
                     test_samples_per_embedding=2, graph_transformer_kwargs=gt, dataset_infos=infos).to(dev).eval()
g = torch.Generator().manual_seed(rank)
torch.manual_seed(1000 + rank)
bs = a.chunk
if a.fixed_n:
    n_nodes = torch.full((bs,), a.fixed_n)
else:
    n_nodes = (torch.randn(bs, generator=g) * 7 + 28).round().clamp(8, a.nmax).long()
n_nodes = n_nodes.to(dev)
z_all = torch.randn(bs, emb, generator=g).to(dev)
T = model.T
if a.buckets > 1 or a.sort:                 # the layout sample_batch(size_buckets=...) uses
    order = torch.argsort(n_nodes, descending=True, stable=True)
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    parts = []
    for p in torch.tensor_split(order, max(a.buckets, 1)):
        if p.numel():
            parts.append(torch.cat([blk.flip(0) if k % 2 else blk for k, blk in enumerate(torch.split(p, cus))]))
else:
    parts = [torch.arange(bs, device=dev)]
total_ms = 0.0
with torch.no_grad():
    for part in parts:
        nn_, z = n_nodes[part], z_all[part]
        n_max = int(nn_.max())
        node_mask = torch.arange(n_max, device=dev).unsqueeze(0) < nn_.unsqueeze(1)
        ones = torch.ones((part.numel(), 1), device=dev)
        zT = sample_discrete_feature_noise(model.limit_dist, node_mask)
        X, E = zT.X, zT.E

        def run(k, s0):
            global X, E
            for s_int in range(s0, s0 - k, -1):
                out = model.sample_p_zs_given_zt(s_int * ones / T, (s_int + 1) * ones / T, X, E, z, node_mask)
                X, E = out.X, out.E

        run(a.warmup, T - 1)
        torch.cuda.synchronize()
        if a.graph:
            Xb, Eb = X.contiguous().clone(), E.contiguous().clone()
            sb, tb = torch.empty_like(ones), torch.empty_like(ones)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = model.sample_p_zs_given_zt(sb, tb, Xb, Eb, z, node_mask)
                Xb.copy_(out.X); Eb.copy_(out.E)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for s_int in range(T - 1 - a.warmup, T - 1 - a.warmup - a.steps, -1):
                sb.copy_(s_int * ones / T); tb.copy_((s_int + 1) * ones / T)
                graph.replay()
        else:
            t0 = time.perf_counter()
            run(a.steps, T - 1 - a.warmup)
        torch.cuda.synchronize()
        total_ms += 1000 * (time.perf_counter() - t0) / a.steps
        if a.kernel_table and not a.graph and rank == 0:
            from mbvae_amd import _lib
            _lib.PROFILE = {}
            run(5, T - 1 - a.warmup - a.steps)
            torch.cuda.synchronize()
            prof, _lib.PROFILE = _lib.PROFILE, None
            rows = sorted(((sum(x.elapsed_time(y) for x, y in evs), len(evs), name, tag) for (name, tag), evs in prof.items()), reverse=True)
            print(f"library calls: {sum(r[0] for r in rows) / 5:.2f} ms per step", file=sys.stderr)
            for ms_, calls, name, tag in rows[:30]:
                print(f"  {ms_ / 5:7.3f} ms/step {calls / 5:6.1f} calls  avg {1e3 * ms_ / calls:7.1f} us  {name} {tag}", file=sys.stderr)
ms = total_ms                     # one reverse step of the whole chunk (all buckets)
if world > 1:
    worst = torch.tensor([ms], device=dev)
    dist.all_reduce(worst, op=dist.ReduceOp.MAX)          # the job ends with the slowest rank
    ms = float(worst)
    dist.destroy_process_group()
if rank == 0:
    print(json.dumps({"metric": "molecule-steps/sec (reverse diffusion, 8-layer denoiser, fp32)", "value": round(world * bs / (ms * 1e-3), 1),
                  "n_gpus": world, "ms_per_step": round(ms, 3), "chunk": bs, "size_buckets": len(parts), "hip_graph": bool(a.graph), "n_max": int(n_nodes.max()),
                  "mean_atoms": round(float(n_nodes.float().mean()), 1), "steps_timed": a.steps,
                  "projected_500x10000_s": round(ms * 1e-3 * 500 * 10000 / (bs * world), 1), "data": "synthetic, random-init weights"}))
