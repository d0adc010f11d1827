"""CPU-only checks of the host logic: C-ABI exports, dense-batch convention, schedule mirror, DDP gradient sync."""
import ctypes
import datetime
import os
import sys
import re
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests._util import ROOT, load_golden


def test_c_abi_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "mbvae_hip.h")).read()
    declared = set(re.findall(r"\b(mbvae_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 20
    import torch  # noqa: F401  (HIP runtime first, see mbvae_amd/_lib.py)
    lib = ctypes.CDLL(os.path.join(ROOT, "mb-vae-dti_amd", "libmbvae_hip.so"))
    missing = [name for name in declared if not hasattr(lib, name)]
    assert not missing, f"declared in include/mbvae_hip.h but not exported: {missing}"
    from mbvae_amd import _lib
    assert set(_lib.SIGNATURES) == declared, set(_lib.SIGNATURES) ^ declared
    # size queries are pure host functions: callable without a GPU
    lib.mbvae_masked_ce_workspace.restype = ctypes.c_int64
    assert lib.mbvae_masked_ce_workspace() > 0


def test_header_declarations_match_the_implementations(tmp_path):
    """Every .hip source compiled (syntax only) behind include/mbvae_hip.h: a declaration whose parameters differ from
    the definition is a conflicting-types error."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "mb-vae-dti_amd", "csrc")
    for name in sorted(f for f in os.listdir(csrc) if f.endswith(".hip")):
        tu = tmp_path / ("chk_" + name)
        tu.write_text('#include <hip/hip_runtime.h>\n#include "%s"\n#include "%s"\n'
                      % (os.path.join(root, "include", "mbvae_hip.h"), os.path.join(csrc, name)))
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-std=c++17", "-fsyntax-only", "-I", csrc, str(tu)],
                           capture_output=True, text=True)
        assert r.returncode == 0 and "error" not in r.stderr, f"{name}:\n{r.stderr[-2000:]}"


def test_product_path_refuses_cpu_tensors():
    from mbvae_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.linear(torch.randn(4, 8), torch.randn(3, 8), torch.zeros(3))


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "mb-vae-dti_amd", "mbvae_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src, f"{fn} mentions the oracle"


def test_schedule_mirror_matches_golden():
    from mbvae_amd.diffusion import MarginalUniformTransition, PredefinedNoiseScheduleDiscrete
    g = load_golden("schedule")
    ns = PredefinedNoiseScheduleDiscrete(500)
    # exp(cumsum(log .)) in fp32 rounds differently from one CPU's vector ISA to the next: 5e-6 element-wise, not bit-equal
    def ulp_close(a, b):
        return bool(((a.double() - b.double()).abs() <= 5e-6 * b.double().abs()).all())
    assert torch.equal(ns.betas, g["betas"]) and ulp_close(ns.alphas_bar, g["alphas_bar"])
    t = g["t_int"] / 500
    assert torch.equal(ns(t_normalized=t), g["beta_t"])
    assert ulp_close(ns.get_alpha_bar(t_normalized=(g["t_int"] - 1) / 500), g["alpha_s_bar"])   # t=0 -> index -1 wraps
    tm = MarginalUniformTransition(g["mx"], g["me"], 4)
    assert torch.equal(tm.get_Qt(g["beta_t"], "cpu").X, g["Qt_X"]) and torch.equal(tm.get_Qt_bar(g["alpha_t_bar"], "cpu").E, g["Qtb_E"])
    assert "betas" in ns.state_dict()


def test_to_dense_convention():
    """Dense-batch convention the losses rely on: padded nodes all-zero in X, EVERY off-diagonal pair (padded ones too)
    one-hot in E with class 0 = no bond, all-zero diagonal, result not masked; self loops dropped."""
    from mbvae_amd.full import to_dense
    x = torch.eye(3)[[0, 1, 2, 1, 0]]                      # graph 0: 3 nodes, graph 1: 2 nodes
    batch = torch.tensor([0, 0, 0, 1, 1])
    edge_index = torch.tensor([[0, 1, 1, 2, 3, 4, 2], [1, 0, 2, 1, 4, 3, 2]])     # last one is a self loop
    bond = torch.eye(4)
    edge_attr = bond[[0, 0, 3, 3, 1, 1, 2]]
    dense, mask = to_dense(x, edge_index, edge_attr, batch)
    assert mask.tolist() == [[True, True, True], [True, True, False]]
    assert torch.equal(dense.X[1, 2], torch.zeros(3)) and torch.equal(dense.X[0], torch.eye(3))
    E = dense.E
    assert E.shape == (2, 3, 3, 5)
    assert float(E[:, torch.arange(3), torch.arange(3)].abs().sum()) == 0                    # diagonal cleared
    assert E[0, 0, 1].tolist() == [0, 1, 0, 0, 0] and E[0, 1, 2].tolist() == [0, 0, 0, 0, 1]
    assert E[0, 0, 2].tolist() == [1, 0, 0, 0, 0]                                            # no bond
    assert E[1, 0, 1].tolist() == [0, 0, 1, 0, 0]
    assert E[1, 0, 2].tolist() == [1, 0, 0, 0, 0] and E[1, 2, 1].tolist() == [1, 0, 0, 0, 0]  # padded pairs: class 0
    assert torch.equal(E, E.transpose(1, 2))


def _leave_group():
    """End of a spawned gloo worker: both ranks meet and tear the process group down in order.  That alone does not end
    the 'terminate called without an active exception' aborts: about one worker in thirty (always a rank > 0) still died
    in interpreter finalisation AFTER destroy_process_group() had returned and all checks had passed (faulthandler: no
    Python frame on any thread, i.e. inside a static destructor of the gloo / c10d libraries).  The worker therefore
    leaves without running static destructors once its work is flushed."""
    dist.barrier()
    dist.destroy_process_group()
    sys.stdout.flush(); sys.stderr.flush()
    os._exit(0)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _toy_net():
    """Denoiser-shaped toy: a trunk, a head that enters the loss and a head that does not (the denoiser's global
    output MLP when only node / edge logits are scored)."""
    torch.manual_seed(0)
    return torch.nn.ModuleDict({"trunk": torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 8), torch.nn.ReLU()),
                                "head": torch.nn.Linear(8, 1), "unused_head": torch.nn.Linear(8, 4)})


def _toy_loss(net, x):
    return net["head"](net["trunk"](x)).pow(2).mean()


def _ddp_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=90))
    from mbvae_amd.ddp import BucketedGradSync
    g = torch.Generator().manual_seed(100)
    data = torch.randn(world, 6, 16, generator=g)
    result = {}
    for static in (False, True):
        net = _toy_net()
        sync = BucketedGradSync(net.parameters(), bucket_bytes=1024, static_graph=static)       # several buckets
        assert len(sync.buckets) > 1
        opt = torch.optim.AdamW(net.parameters(), lr=1e-2, weight_decay=0.1, amsgrad=True)
        launched_in_hooks = []
        for step in range(3):                                             # later steps check zero_grad / hook re-arming
            _toy_loss(net, data[rank] * (step + 1)).backward()
            launched_in_hooks.append(sync._next)
            sync.finish()
            # parameters outside the loss end the step WITHOUT a gradient, as in a single process
            assert all(p.grad is None for p in net["unused_head"].parameters())
            grads = torch.cat([p.grad.reshape(-1).clone() for p in net.parameters() if p.grad is not None])
            opt.step()
            sync.zero_grad()
        if static:       # the used set is known after step 0: the bucket holding the unused head no longer blocks the hooks
            assert launched_in_hooks[0] == 0 and launched_in_hooks[2] == len(sync.buckets), launched_in_hooks
        else:            # bucket 0 holds the unused head (reverse registration order): everything waits for finish()
            assert launched_in_hooks == [0, 0, 0], launched_in_hooks
        result[f"grads_{int(static)}"] = grads
        result[f"weights_{int(static)}"] = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    # a parameter used on ONE rank only: the other ranks contribute zeros, nobody drops it, the order of collectives
    # stays 0, 1, 2, ... on both ranks
    net = _toy_net()
    sync = BucketedGradSync(net.parameters(), bucket_bytes=1024)
    loss = _toy_loss(net, data[rank])
    if rank == 1:
        loss = loss + net["unused_head"](net["trunk"](data[rank])).pow(2).mean()
    loss.backward()
    sync.finish()
    assert all(p.grad is not None for p in net["unused_head"].parameters())
    result["one_rank_only"] = torch.cat([p.grad.reshape(-1).clone() for p in net["unused_head"].parameters()])
    if rank == 0:
        torch.save(result, out)
    _leave_group()


def test_bucketed_grad_sync_gloo_world2(tmp_path):
    """Two ranks with BucketedGradSync == one process that averages the two shards' gradients: same gradients, same
    weights after three AdamW(amsgrad, weight decay) steps, unused parameters untouched."""
    out = str(tmp_path / "g.pt")
    port = _free_port()
    mp.spawn(_ddp_worker, args=(2, port, out), nprocs=2, join=True)
    saved = torch.load(out, weights_only=True)
    g = torch.Generator().manual_seed(100)
    data = torch.randn(2, 6, 16, generator=g)
    net = _toy_net()
    w0_unused = torch.cat([p.detach().reshape(-1).clone() for p in net["unused_head"].parameters()])
    opt = torch.optim.AdamW(net.parameters(), lr=1e-2, weight_decay=0.1, amsgrad=True)
    for step in range(3):
        tot = None
        for r in range(2):
            net.zero_grad(set_to_none=True)
            _toy_loss(net, data[r] * (step + 1)).backward()
            v = [p.grad.clone() if p.grad is not None else None for p in net.parameters()]
            tot = v if tot is None else [a if b is None else a + b for a, b in zip(tot, v)]
        for p, gsum in zip(net.parameters(), tot):
            p.grad = None if gsum is None else gsum / 2
        want_grads = torch.cat([p.grad.reshape(-1) for p in net.parameters() if p.grad is not None])
        opt.step()
    want_w = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    assert torch.equal(torch.cat([p.detach().reshape(-1) for p in net["unused_head"].parameters()]), w0_unused)   # no decay
    for k in ("0", "1"):
        assert torch.allclose(saved["grads_" + k], want_grads, rtol=1e-6, atol=1e-7)
        assert torch.allclose(saved["weights_" + k], want_w, rtol=1e-6, atol=1e-7), "post-step weights differ from dp1"
    net = _toy_net()
    net["unused_head"](net["trunk"](data[1])).pow(2).mean().backward()
    want = torch.cat([p.grad.reshape(-1) for p in net["unused_head"].parameters()]) / 2
    assert torch.allclose(saved["one_rank_only"], want, rtol=1e-6, atol=1e-7)


class _FakeSampler(torch.nn.Module):
    """Stands in for FullDTIModel.sample_batch on the CPU: molecules are a deterministic function of the embedding
    row, so the gathered result can be compared with a single-process run."""

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(1))
        self.hparams = type("H", (), {"dataset_infos": {"general": {"max_n_nodes": 9}}})()
        self.calls = []

    def sample_batch(self, drug_embeddings, num_nodes=None, num_samples_per_embedding=1, return_graphs=False, **kw):
        assert return_graphs
        self.calls.append(drug_embeddings.size(0))
        out = []
        for row in drug_embeddings:
            tag = int(row[0].item())
            grp = []
            for s in range(num_samples_per_embedding):
                k = 3 + (tag + s) % 7
                atoms = (torch.arange(k) + tag + s) % 5
                bonds = ((torch.arange(k)[:, None] + torch.arange(k)[None, :] + tag) % 5) * (1 - torch.eye(k, dtype=torch.long))
                grp.append((atoms, bonds))
            out.append(grp)
        return out


def _sampling_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=90))
    from mbvae_amd.sharded_sampling import sample_sharded
    model = _FakeSampler()
    emb = torch.arange(11, dtype=torch.float32).unsqueeze(1).repeat(1, 4)      # 11 embeddings: ranks hold 6 and 5
    res = sample_sharded(model, emb, num_samples_per_embedding=3, chunk=6, seed=5, device="cpu")
    assert model.calls == ([2, 2, 2] if rank == 0 else [2, 2, 1])               # chunk 6 = 2 embeddings x 3 samples
    assert (res is None) == (rank != 0)
    if rank == 0:
        torch.save([[list(pair) for pair in grp] for grp in res], out)
    _leave_group()


def test_sharded_sampling_gloo_world2(tmp_path):
    from mbvae_amd.sharded_sampling import sample_sharded, shard_bounds
    assert [shard_bounds(11, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 11)]
    out = str(tmp_path / "s.pt")
    mp.spawn(_sampling_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out, weights_only=True)
    emb = torch.arange(11, dtype=torch.float32).unsqueeze(1).repeat(1, 4)
    want = sample_sharded(_FakeSampler(), emb, num_samples_per_embedding=3, chunk=6, device="cpu")   # no process group
    assert len(got) == len(want) == 11
    for g_grp, w_grp in zip(got, want):
        assert len(g_grp) == len(w_grp) == 3
        for (ga, gb), (wa, wb) in zip(g_grp, w_grp):
            assert torch.equal(ga, wa) and torch.equal(gb, wb)


def test_header_lists_every_tuning_switch():
    """include/mbvae_hip.h names every option of the library's table (csrc/common.h), and no others."""
    common = open(os.path.join(ROOT, "mb-vae-dti_amd", "csrc", "common.h")).read()
    table = common[common.index("static const char* const names[MBVAE_OPT_COUNT]"):]
    names = re.findall(r'"([A-Z0-9_]+)"', table[:table.index("};")])
    enum = common[common.index("enum MbvaeOption"):common.index("MBVAE_OPT_COUNT")]
    assert len(names) == len(re.findall(r"MBVAE_OPT_[A-Z0-9_]+", enum)) >= 24
    header = open(os.path.join(ROOT, "include", "mbvae_hip.h")).read()
    doc = header[header.index("---- tuning switches"):header.index("int mbvae_set_option")]
    for n in names:
        assert re.search(r"\b" + n + r"\b", doc), f"option {n} is not documented in include/mbvae_hip.h"
    mentioned = set(re.findall(r"\b((?:NO_|EA_|EF_|EFB_|TSGEMM_|GEMM_)[A-Z0-9_]+)\b", doc))
    assert mentioned <= set(names), mentioned - set(names)
    assert f"The {len(names)} names" in doc


def _run_launcher(tmp_path, n, have, extra=()):
    """bench.launch_workers in a child process (so its inherited stdout can be captured) with the stub worker."""
    import json
    import subprocess
    stub = os.path.join(ROOT, "tests", "helpers", "launch_stub_worker.py")
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); import bench; "
            f"sys.exit(bench.launch_workers({n}, script={stub!r}, argv={[str(tmp_path)] + list(extra)!r}, have={have}))")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)
    seen = {}
    for name in os.listdir(tmp_path):
        with open(os.path.join(tmp_path, name)) as f:
            seen[name] = json.load(f)
    return r, seen


def test_bench_launcher_gives_every_worker_its_rank_and_one_rendezvous(tmp_path):
    """`python bench.py --gpus N` without torchrun: N workers, ranks 0..N-1 once each, one shared 127.0.0.1 rendezvous port,
    the dmabuf IPC switch set, the caller's arguments passed through, exactly one JSON line on stdout (rank 0's)."""
    import json
    r, seen = _run_launcher(tmp_path, 4, have=8, extra=("--steps", "3"))
    assert r.returncode == 0, r.stderr
    assert sorted(seen) == [f"rank{i}.json" for i in range(4)]
    ports = set()
    for i in range(4):
        env = seen[f"rank{i}.json"]["env"]
        assert env["RANK"] == env["LOCAL_RANK"] == str(i)
        assert env["WORLD_SIZE"] == env["LOCAL_WORLD_SIZE"] == "4"
        assert env["MASTER_ADDR"] == "127.0.0.1"
        assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        ports.add(env["MASTER_PORT"])
        assert seen[f"rank{i}.json"]["argv"] == [str(tmp_path), "--steps", "3"]
    assert len(ports) == 1 and 1024 <= int(next(iter(ports))) < 65536
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and json.loads(lines[0]) == {"metric": "stub", "n_gpus": 4}


def test_bench_launcher_passes_on_the_worst_exit_status_and_refuses_missing_gpus(tmp_path):
    """One failing rank fails the job with that rank's code (the others are still waited for); asking for more GPUs
    than are visible starts nothing and returns 2."""
    r, seen = _run_launcher(tmp_path, 3, have=3, extra=("--exit", "1:7"))
    assert r.returncode == 7 and len(seen) == 3
    sub = tmp_path / "none"
    sub.mkdir()
    r, seen = _run_launcher(sub, 4, have=2)
    assert r.returncode == 2 and not seen and "4 requested but 2" in r.stderr.replace("--gpus ", "")


def test_stacked_projection_weights_follow_their_parameters():
    """The stacked q|k|v weights are kept as a copy: it must follow every way a parameter can change (optimiser-style
    in-place update, load_state_dict, a swapped .data), refresh_stacked_weights must renew it in one go, and gradients must
    reach the layers' own parameters through it."""
    sys.path.insert(0, os.path.join(ROOT, "mb-vae-dti_amd"))
    from mbvae_amd.graph_transformer import NodeEdgeBlock, refresh_stacked_weights
    torch.manual_seed(0)
    blk = NodeEdgeBlock(8, 4, 8, 2)
    layers = (blk.q, blk.k, blk.v)
    want = lambda: (torch.cat([l.weight for l in layers], 0).detach(), torch.cat([l.bias for l in layers], 0).detach())
    w, b = blk._stacked("qkv", layers)
    assert torch.equal(w, want()[0]) and torch.equal(b, want()[1]) and w.requires_grad
    (w.sum() + 2 * b.sum()).backward()
    for l in layers:
        assert torch.equal(l.weight.grad, torch.ones_like(l.weight)) and torch.equal(l.bias.grad, torch.full_like(l.bias, 2.0))
    buf = blk._stack_cache["qkv"][1]
    with torch.no_grad():
        blk.k.weight.add_(1.0)                                   # in-place update: version counter
    w, b = blk._stacked("qkv", layers)
    assert torch.equal(w, want()[0]) and blk._stack_cache["qkv"][1] is buf      # renewed in place
    blk.load_state_dict({k: v + 0.5 for k, v in blk.state_dict().items()})
    assert torch.equal(blk._stacked("qkv", layers)[0], want()[0])
    blk.v.bias.data = torch.full_like(blk.v.bias, 3.0)           # swapped storage: data_ptr
    assert torch.equal(blk._stacked("qkv", layers)[1], want()[1])
    with torch.no_grad():
        for p in blk.parameters():
            p.mul_(0.9)
    refresh_stacked_weights(blk)                                 # all copies of the model at once
    hit = blk._stack_cache["qkv"]
    assert torch.equal(hit[1], want()[0]) and torch.equal(hit[2], want()[1])
    assert hit[0] == tuple((p.data_ptr(), p._version) for l in layers for p in (l.weight, l.bias))
    with torch.no_grad():
        assert not blk._stacked("qkv", layers)[0].requires_grad
