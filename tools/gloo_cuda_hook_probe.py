"""Does a gloo all-reduce of a GPU tensor complete when it is launched (async) from a post-accumulate-grad hook, i.e. from
the autograd engine's device thread?  Two ranks on one GPU; variants: launched from the hook / from the main thread after
backward.  GPU box only:  for r in 0 1; do RANK=$r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 python tools/gloo_cuda_hook_probe.py & done; wait"""
import datetime, faulthandler, os, sys
faulthandler.dump_traceback_later(40, exit=True)
import torch, torch.distributed as dist
rank = int(os.environ["RANK"])
dist.init_process_group("gloo", rank=rank, world_size=2, timeout=datetime.timedelta(seconds=60))
dev = torch.device("cuda")
for where in ("main thread after backward", "hook on the autograd thread"):
    for step in range(3):
        lin = torch.nn.Linear(256, 256).to(dev)
        flat = torch.zeros(256 * 256, device=dev)
        lin.weight.grad = flat.view(256, 256)
        pending = []
        if where.startswith("hook"):
            lin.weight.register_post_accumulate_grad_hook(lambda p: pending.append(dist.all_reduce(flat, async_op=True)))
        lin(torch.randn(64, 256, device=dev) + rank).pow(2).mean().backward()
        if not where.startswith("hook"):
            pending.append(dist.all_reduce(flat, async_op=True))
        for w in pending:
            w.wait()
        torch.cuda.synchronize()
        print(f"rank {rank}: all-reduce launched from the {where}: step {step} done, sum {float(flat.sum()):.4f}", flush=True)
dist.barrier()
dist.destroy_process_group()
sys.stdout.flush()
os._exit(0)
