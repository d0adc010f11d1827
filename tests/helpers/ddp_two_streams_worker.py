"""One data-parallel rank of tests/test_gpu_two_streams.py::test_grad_sync_world2_with_second_stream (gloo; every rank uses
GPU 0).  Three denoiser train steps on this rank's own shard; rank 0 saves losses and final weights."""
import faulthandler
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mb-vae-dti_amd")); sys.path.insert(0, ROOT)
faulthandler.dump_traceback_later(int(os.environ.get('WORKER_HANG_SECONDS', '80')), exit=True)        # a hang shows where every thread stood

import torch
import torch.distributed as dist


def main():
    rank = int(os.environ["RANK"])
    dist.init_process_group("gloo", rank=rank, world_size=int(os.environ["WORLD_SIZE"]))
    try:
        from mbvae_amd.trainstep import CFG3, DenoiserTrainStep, synthetic_graphs
        cfg = dict(CFG3); cfg["n_layers"] = 2
        side = os.environ["WORKER_SIDE_STREAM"] == "1"
        tr = DenoiserTrainStep(cfg, device="cuda", distributed=True, seed=3, side_stream=side)
        g = torch.Generator().manual_seed(50 + rank)
        sizes = torch.randint(20, 65, (32,), generator=g); sizes[0] = 64
        X0, E0, mask = synthetic_graphs(32, 64, "cuda", g, sizes=sizes)
        cond = torch.randn(32, 1024, generator=g).cuda()
        losses = []
        for k in range(3):
            losses.append(tr.step(X0, E0, mask, cond))
            print(f"rank {rank}: step {k} issued", flush=True)
        losses = torch.stack(losses).cpu()
        torch.cuda.synchronize()
        weights = torch.cat([p.detach().reshape(-1) for p in tr.model.parameters()]).cpu()
        both = [torch.empty_like(weights) for _ in range(2)]
        dist.all_gather(both, weights)
        assert torch.equal(both[0], both[1]), "the two ranks ended with different weights"
        if rank == 0:
            torch.save({"losses": losses, "weights": weights}, os.environ["WORKER_OUT"])
        dist.barrier()
    finally:
        dist.destroy_process_group()
    sys.stdout.flush(); sys.stderr.flush()
    os._exit(0)          # (gloo workers now and then abort inside static destructors at interpreter exit)


if __name__ == "__main__":
    main()
