"""Benchmark: graphs/sec of the denoiser train step (256 graphs x 64 atoms, 8 layers) on N MI355X of one node.

    python bench.py --gpus N --steps 5 --warmup 2          (N > 1: starts one worker process per GPU itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = forward noising (HIP) -> extra features -> 8-layer graph-transformer denoiser forward -> masked
node/edge cross-entropy -> backward -> gradient all-reduce (N>1) -> AdamW(amsgrad), fp32, dropout active,
on 256 synthetic graphs of 64 atoms per GPU (weak scaling).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "mb-vae-dti_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak
MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32-input MFMA peak


def cpu_baseline(cfg, n, seconds=20.0):
    """The CPU oracle (this repo's restatement of the reference path) timed on the host cores over the SAME step the
    GPU line measures: forward noising, extra features (cycle counts + eigh), denoiser forward, masked CE, backward,
    AdamW(amsgrad); fp32, batch 8 (the unfused path keeps ~12 [bs,n,n,256] tensors per layer), as many steps as fit in
    `seconds`.  The oracle has no dropout (the reference's 9 dropouts per layer are element-wise passes it skips)."""
    from oracle import denoiser as O_den, diffusion as O_dif, features as O_feat, heads as O_heads
    from mbvae_amd.graph_transformer import GraphTransformer
    from mbvae_amd.trainstep import ATOM_WEIGHTS, ME, MX, VALENCIES, synthetic_graphs
    torch.manual_seed(0)
    bs = 8
    net = GraphTransformer(**cfg)
    params = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    opt = torch.optim.AdamW(list(params.values()), lr=5e-4, weight_decay=1e-4, amsgrad=True)
    g = torch.Generator().manual_seed(0)
    X0, E0, mask = synthetic_graphs(bs, n, "cpu", g)
    cond = torch.randn(bs, cfg["output_dims"]["y"], generator=g)
    sch = O_dif.Schedule(500)
    mx, me = torch.tensor(MX), torch.tensor(ME)
    heads = cfg["hidden_dims"]["n_head"]

    def one():
        t_int = torch.randint(0, 501, (bs, 1), generator=g).float()
        eX = torch.empty(bs * n, len(MX)).exponential_(1, generator=g)
        eE = torch.empty(bs * n * n, len(ME)).exponential_(1, generator=g)
        Xt, Et, npar, _, _ = O_dif.apply_noise(sch, mx, me, X0, E0, mask, t_int, eX, eE)
        Xe, ye = O_feat.extra_features(Xt, Et, npar["t"], mask, torch.randn(bs, n, generator=g), 64, list(VALENCIES),
                                       list(ATOM_WEIGHTS), 1000.0)
        oX, oE, _ = O_den.graph_transformer(params, torch.cat((Xt, Xe), 2), Et, torch.cat((cond, ye), 1), mask,
                                            cfg["n_layers"], heads, cfg["output_dims"])
        O_heads.reconstruction_loss(oX, oE, X0, E0, [1, 5]).backward()
        opt.step()
        opt.zero_grad(set_to_none=True)

    one()  # warm-up
    steps, t0 = 0, time.perf_counter()
    while True:
        one()
        steps += 1
        el = time.perf_counter() - t0
        if el > seconds or steps >= 8:
            break
    return {"value": round(bs * steps / el, 3), "unit": "graphs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"CPU oracle (oracle/*.py): noise + extra features + denoiser fwd + masked CE + bwd + AdamW(amsgrad), "
                      f"batch {bs} x {steps} steps, n={n}, {cfg['n_layers']} layers, fp32, no dropout"}


def kernel_table(profile):
    rows = {}
    for (name, tag), evs in profile.items():
        ms = sum(s.elapsed_time(e) for s, e in evs)
        r = rows.setdefault(name, {"ms": 0.0, "calls": 0, "flops": 0.0})
        r["ms"] += ms
        r["calls"] += len(evs)
        if tag is not None and name.startswith("mbvae_linear"):
            r["flops"] += linear_flops(name, tag) * len(evs)
    return rows


def linear_flops(name, tag):
    """fp32-equivalent flops of one launch: 2*M*K*N per product (the single-pass backward holds two)."""
    M, K, N = tag
    return (4.0 if name == "mbvae_linear_bwd_both" else 2.0) * M * K * N


def extra_objects():
    """cfg5 / cfg2 / cfg4 of the benchmark list on this GPU: reverse-diffusion sampling (one rank's share of 10 000 molecules:
    1 250, sizes ~ N(28, 7^2), 20 reverse steps), the multi-branch encoder train step (as one captured HIP graph) and the
    joint FullDTIModel training step, 256 pairs each.  Child processes (tools/bench_sampling.py, tools/bench_configs.py);
    a failing child leaves an error string in its place."""
    def child(script, *argv):
        cmd = [sys.executable, os.path.join(ROOT, "tools", script), *argv]
        try:
            res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
        except subprocess.TimeoutExpired:
            return [{"error": "timed out"}]
        lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
        if res.returncode != 0 or not lines:
            return [{"error": f"exit status {res.returncode}: " + res.stderr.strip()[-300:]}]
        return [json.loads(l) for l in lines]

    out = {"sampling": child("bench_sampling.py", "--chunk", "1250", "--steps", "20")[0]}
    enc = child("bench_configs.py", "--which", "encoders", "--graph", "--steps", "10")
    out["cfg2_encoder_step"] = enc[0]
    out["cfg4_joint_step"] = child("bench_configs.py", "--which", "joint", "--steps", "5")[0]
    return out


def launch_workers(n, script=None, argv=None, have=None):
    """`python bench.py --gpus N` without a launcher: start N fresh worker processes (this same script, one per GPU,
    rendezvous on 127.0.0.1) BEFORE anything in this process touches the GPU, wait for them, and pass on the worst
    exit status.  Rank 0 prints the JSON line on the inherited stdout.
    script / argv / have: the worker program, its arguments and the number of visible GPUs (defaults: this file,
    this process' arguments, what torch counts) - parameters so that the launcher can be exercised without a GPU."""
    if have is None:
        have = torch.cuda.device_count()        # (the only GPU-related call of this process; the children start fresh)
    if have < n:
        print(f"bench.py: --gpus {n} requested but {have} GPU(s) visible", file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv),
                                      env=env))
    codes = [p.wait() for p in procs]
    return max((abs(c) for c in codes), default=0)      # (a child killed by signal S reports -S)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="graphs per GPU")
    ap.add_argument("--nodes", type=int, default=64, help="atoms per graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-bf16-line", action="store_true", help="skip the second, labelled bf16-storage measurement")
    ap.add_argument("--kernel-table", action="store_true", help="print per-kernel times to stderr")
    ap.add_argument("--one-stream", action="store_true", help="A/B aid: the whole step on one HIP stream (not the headline mode)")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the additional labelled objects (sampling, encoder step, joint step; N = 1 only)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_workers(args.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    from mbvae_amd import _lib
    from mbvae_amd.trainstep import CFG3, DenoiserTrainStep, synthetic_graphs

    # second stream for the node / global-feature kernels (bit-identical to one stream, tests/test_gpu_two_streams.py);
    # on one GPU the weight update additionally runs on it under the next step's noising and graph features
    trainer = DenoiserTrainStep(CFG3, device=dev, distributed=world > 1, seed=0, side_stream=not args.one_stream,
                                overlap_optimizer=world == 1 and not args.one_stream)
    g = torch.Generator().manual_seed(1000 + rank)          # a different data shard per rank
    bs, n = args.batch, args.nodes
    X0, E0, mask = synthetic_graphs(bs, n, dev, g)
    cond = torch.randn(bs, 1024, generator=g).to(dev)
    torch.manual_seed(7 + rank)

    from mbvae_amd import ops
    ops.set_bf16_storage(False)          # the headline figure is the fp32 path, whatever the environment says
    for _ in range(args.warmup):
        trainer.step(X0, E0, mask, cond)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()                     # the timed region carries no instrumentation
    for _ in range(args.steps):
        loss = trainer.step(X0, E0, mask, cond)
    trainer.wait_optimizer()                     # the last weight update is part of the last step
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # Per-kernel durations for the roofline object come from a SEPARATE, untimed pass over the same steps: every C-ABI
    # call bracketed by HIP events on the stream it is launched on (two event records per call would otherwise sit
    # inside the timed region).
    # That pass runs on ONE stream, so that a kernel's duration is its own and not that of whatever shared the GPU with it.
    profile = {}
    trainer.wait_optimizer()
    ops.set_side_stream(False)
    try:
        if rank == 0:
            _lib.PROFILE = profile
        for _ in range(args.steps):              # (every rank runs the same steps: the gradient all-reduce is collective)
            trainer.step(X0, E0, mask, cond)
        torch.cuda.synchronize()
    finally:
        _lib.PROFILE = None
        ops.set_side_stream(not args.one_stream)
    # second, separately labelled measurement (N = 1 only): the opt-in bf16 STORAGE mode of the edge block's large
    # activations (fp32 weights and arithmetic; parity bound 2e-2 instead of 1e-5).  Never the headline `value`.
    bf16_line = None
    if world == 1 and not args.no_bf16_line:
        ops.set_bf16_storage(True)
        try:
            trainer.step(X0, E0, mask, cond)
            torch.cuda.synchronize()
            tb = time.perf_counter()
            for _ in range(args.steps):
                trainer.step(X0, E0, mask, cond)
            trainer.wait_optimizer()
            torch.cuda.synchronize()
            eb = time.perf_counter() - tb
        finally:
            ops.set_bf16_storage(False)
        bf16_line = {"value": round(bs * args.steps / eb, 2), "unit": "graphs/s", "ms_per_step": round(1000 * eb / args.steps, 3),
                     "dtype": "f32 arithmetic and weights; E1|E2, P and their gradients stored as bf16",
                     "note": "opt-in (ops.set_bf16_storage / MBVAE_BF16_STORAGE=1), parity 2e-2 against the fp32 path "
                             "(tests/test_gpu_bf16_storage.py); NOT the headline value"}

    if rank == 0:
        rows = kernel_table(profile)
        if args.kernel_table:
            tot = sum(r["ms"] for r in rows.values())
            for name, r in sorted(rows.items(), key=lambda kv: -kv[1]["ms"]):
                print(f"{name:32s} {r['ms'] / args.steps:9.3f} ms/step  {r['calls'] // args.steps:5d} calls/step"
                      f"  {100 * r['ms'] / max(tot, 1e-9):5.1f}%", file=sys.stderr)
            shapes = {}
            for (name, tag), evs in profile.items():
                if tag is not None:
                    ms = sum(a.elapsed_time(b) for a, b in evs)
                    shapes[(name, tag)] = (ms, len(evs))
            for (name, tag), (ms, calls) in sorted(shapes.items(), key=lambda kv: -kv[1][0])[:40]:
                M, K, N = tag
                print(f"  {name[6:]:18s} M={M:<8d} K={K:<5d} N={N:<5d} {ms / args.steps:8.3f} ms/step {calls // args.steps:4d} calls  "
                      f"{linear_flops(name, tag) * calls / (ms * 1e-3) / 1e12:6.1f} TFLOP/s", file=sys.stderr)
            print(f"{'sum of HIP kernels':32s} {tot / args.steps:9.3f} ms/step;  wall {1000 * elapsed / args.steps:.3f} ms/step",
                  file=sys.stderr)
        # Dominant kernel = the (entry point, shape) with the largest share of the timed region.  Projections are
        # priced against the dense fp32 MFMA peak with 2*M*K*N flops per launch, the edge-attention streams against
        # HBM with their algorithmic bytes per launch; both over the HIP-event average launch duration.
        s = 4
        dx = CFG3["hidden_dims"]["dx"]
        de = CFG3["hidden_dims"]["de"]
        ea_bytes = {"mbvae_edge_attn_bwd": s * (5 * bs * n * n * dx + 8 * bs * n * dx),      # E1,E2,dP in; dE1,dE2 out
                    "mbvae_edge_attn_fwd": s * (3 * bs * n * n * dx + 4 * bs * n * dx + 2 * bs * dx) + bs * n,
                    # projection inside the attention forward, training form: E in; E1|E2 and P out
                    "mbvae_edge_fused_fwd": s * (bs * n * n * (de + 3 * dx) + 4 * bs * n * dx),
                    # the default training pair: forward E in, P out; backward E, dP in, dE1|dE2 out (E1|E2 recomputed)
                    "mbvae_edge_fused_fwd_recompute": s * (bs * n * n * (de + dx) + 4 * bs * n * dx),
                    "mbvae_edge_fused_bwd": s * (bs * n * n * (de + 3 * dx) + 10 * bs * n * dx),
                    # both gradients of the e_mul|e_add projection from one pass: dE1|dE2, E, addend in; dE out (+ weights)
                    "mbvae_linear_bwd_both": s * (bs * n * n * (2 * dx + 3 * de) + 2 * (2 * dx * de)),
                    # edge feed-forward block, training pair: x in; hidden (2 de wide), pre-LayerNorm sum and output out /
                    # dr, dres and the hidden activation in; dh and dx out
                    "mbvae_edge_ffn_fwd_train": s * bs * n * n * (de + 2 * de + de + de),
                    "mbvae_edge_ffn_bwd_data": s * bs * n * n * (de + de + 2 * de + 2 * de + de)}
        cands = []
        for (name, tag), evs in profile.items():
            ms = sum(a.elapsed_time(b) for a, b in evs)
            if name in ea_bytes:                 # streaming kernels: priced against HBM
                cands.append((ms, name, None, len(evs)))
            elif name.startswith("mbvae_linear") and tag is not None:
                cands.append((ms, name, tag, len(evs)))
        ms, name, tag, calls = max(cands)
        avg_ms = ms / calls
        traffic_tbl = {}
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic_by_entry.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                traffic_tbl = json.load(f)
        if tag is not None:
            M, K, N = tag
            achieved = 2.0 * M * K * N / (avg_ms * 1e-3) / 1e12
            key = f"{name}:{M}x{K}x{N}"
            roofline = {"kernel": f"{name} M={M} K={K} N={N}", "bound": "mfma", "achieved": round(achieved, 3),
                        "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / MFMA_F32_PEAK_TFLOPS, 4)}
        else:
            achieved = ea_bytes[name] / (avg_ms * 1e-3) / 1e9
            key = name
            roofline = {"kernel": name, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4)}
            lin_tags = [t for (nm, t) in profile if nm == name and t is not None]
            if name.startswith("mbvae_linear") and lin_tags:      # a streaming GEMM: also its fp32-equivalent rate
                fl = linear_flops(name, lin_tags[0]) / (avg_ms * 1e-3) / 1e12
                roofline["fp32_equiv_TFLOPs"] = round(fl, 2)
                roofline["frac_of_fp32_mfma_peak"] = round(fl / MFMA_F32_PEAK_TFLOPS, 4)
        # HBM bytes per launch of that kernel from the committed rocprofv3 PMC passes (2 x FETCH_SIZE + WRITE_SIZE)
        roofline["traffic"] = traffic_tbl.get(key, {}).get("bytes")
        roofline["avg_launch_ms"] = round(avg_ms, 4)
        roofline["launches_per_step"] = calls // args.steps
        lin = [r for k, r in rows.items() if k.startswith("mbvae_linear")]
        lin_ms, lin_flops = sum(r["ms"] for r in lin), sum(r["flops"] for r in lin)
        roofline["all_projections_TFLOPs"] = round(lin_flops / max(lin_ms, 1e-9) / 1e9, 2)
        # matrix-core busy fraction of the edge projections from the committed counter pass
        # (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)), weighted by each kernel's duration there
        mpath = os.path.join(ROOT, "profiles", "mfma_busy_by_entry.json")
        if os.path.exists(mpath):
            with open(mpath) as f:
                busy = json.load(f)
            num = den = 0.0
            for k, v in busy.items():
                if k.startswith("mbvae_linear") and isinstance(v, dict) and v.get("mfma_busy_frac", 0) > 0:
                    wgt = v.get("gui_active", 1.0)
                    num += wgt * v["mfma_busy_frac"]; den += wgt
            if den > 0:
                roofline["mfma_busy_frac"] = round(num / den, 4)
        # the same products on the native fp32 MFMA kernels (NO_SPLIT_MFMA route, earlier counter pass): busy fractions there
        fpath = os.path.join(ROOT, "profiles", "mfma_busy_fp32_mfma_kernels.json")
        if os.path.exists(fpath):
            with open(fpath) as f:
                vals = [v["mfma_busy_frac"] for k, v in json.load(f).items()
                        if k.startswith("mbvae_linear") and isinstance(v, dict) and v.get("mfma_busy_frac", 0) > 0]
            if vals:
                roofline["mfma_busy_frac_native_fp32_kernels"] = round(sum(vals) / len(vals), 4)
        # edge-path projections alone ([bs*n*n, d] rows): fp32-equivalent rate, also as a fraction of the fp32 MFMA peak
        e_ms = e_fl = 0.0
        for (nm, tg), evs in profile.items():
            if nm.startswith("mbvae_linear") and tg is not None and tg[0] == bs * n * n and min(tg[1], tg[2]) >= 64:
                e_ms += sum(a.elapsed_time(b) for a, b in evs)
                e_fl += linear_flops(nm, tg) * len(evs)
        if e_ms > 0:
            roofline["edge_projections_TFLOPs"] = round(e_fl / e_ms / 1e9, 2)
            roofline["edge_projections_frac_of_fp32_mfma_peak"] = round(e_fl / e_ms / 1e9 / MFMA_F32_PEAK_TFLOPS, 4)
        for nm in ea_bytes:
            r = rows.get(nm)
            if r and r["ms"] > 0:
                roofline[nm[6:] + "_GBps"] = round(ea_bytes[nm] / (r["ms"] / r["calls"] * 1e-3) / 1e9, 1)
        # the loss kernels (node + edge logits of one step): forward reads logits and one-hot targets, backward also writes
        # the logit gradient; 5 classes, fp32
        ce_rows = bs * n + bs * n * n
        for nm, per_row in (("mbvae_masked_ce_fwd", 2 * 5 * 4), ("mbvae_masked_ce_bwd", 3 * 5 * 4)):
            r = rows.get(nm)
            if r and r["ms"] > 0:
                roofline[nm[6:] + "_GBps"] = round(ce_rows * per_row * args.steps / (r["ms"] * 1e-3) / 1e9, 1)
        # whole-step figures: ~15.0 GFLOP and, with only layer inputs/outputs touching memory, ~61.5 MB per graph
        per_gpu_rate = bs * args.steps / elapsed
        roofline["whole_step_TFLOPs"] = round(per_gpu_rate * 15.0e9 / 1e12, 2)
        roofline["whole_step_mfma_frac"] = round(per_gpu_rate * 15.0e9 / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)
        roofline["whole_step_hbm_floor_frac"] = round(per_gpu_rate * 61.5e6 / 1e9 / HBM_PEAK_GBS, 4)
        out = {
            "metric": "graphs/sec (train step, n=64, whole node)", "value": round(world * bs * args.steps / elapsed, 2),
            "unit": "graphs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "cfg3: DiGress graph-transformer denoiser train step (noise + extra features + fwd + masked CE "
                                   "+ bwd + AdamW), 8 layers, dx=256 de=64 dy=1024, 5 atom / 5 bond classes, dropout 0.1",
                       "graphs_per_gpu": bs, "global_batch": world * bs, "n_atoms": n, "parallelism": f"dp{world}"},
            "roofline": roofline, "loss": round(float(loss), 5),
        }
        if bf16_line is not None:
            out["bf16_storage_mode"] = bf16_line
        if world == 1 and not args.no_extra:
            # further labelled measurements of the same library (never `value`): each in a fresh child process after this
            # one has released its GPU memory
            del trainer, X0, E0, mask, cond
            torch.cuda.empty_cache()
            out.update(extra_objects())
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(CFG3, n)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
