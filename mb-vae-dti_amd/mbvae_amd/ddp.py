"""Data-parallel gradient synchronisation over torch.distributed (RCCL on MI355X, gloo in CPU tests).

Gradients live as views into a few large flat buckets (filled in reverse registration order, i.e. roughly the
order backward produces them).  A post-accumulate hook counts ready parameters per bucket; the moment a bucket is
complete its all-reduce is launched asynchronously, so the collective overlaps the rest of backward.  xGMI is a
point-to-point mesh, so few large messages beat many small ones: the default bucket is 64 MiB.

Semantics are those of a single process that averages the ranks' gradients:
* Collectives are always issued in bucket order 0, 1, 2, ... on every rank (a bucket that fills early waits for
  its predecessors), so ranks can never pair up all-reduces of different buckets.
* A parameter that received no gradient on ANY rank in a step ends the step with `.grad = None`, exactly as it would
  without this class: the optimiser skips it (no weight decay, no moment updates).  A parameter used on some ranks
  only gets the mean over all ranks, zeros counting for the others.  The ranks agree on the used set with one tiny
  MIN all-reduce per step (a parameter is unused everywhere iff the minimum of the "unused here" flags is still 1); with static_graph=True that happens in the first step only and the set is reused, which
  also lets a bucket holding never-used parameters launch from the hooks instead of waiting for finish().
* Gradients may be produced on more than one GPU stream (ops.set_side_stream: node / global-feature kernels on a
  second stream).  A collective only orders itself behind the stream it is launched from, so every hook notes the
  stream its gradient was accumulated on and a bucket's launch first makes the launching stream wait for all of them.
"""
import os

import torch
import torch.distributed as dist

_TRACE = os.environ.get("MBVAE_DDP_TRACE") is not None      # prints every bucket launch and every wait


class BucketedGradSync:
    def __init__(self, params, bucket_bytes=64 << 20, group=None, static_graph=False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.static_graph = static_graph
        self.params = [p for p in params if p.requires_grad]
        self.buckets = []            # (flat tensor, [params])
        self._bucket_of = {}
        self._index = {p: i for i, p in enumerate(self.params)}
        self._pending = []
        cur, cur_numel = [], 0
        cap = bucket_bytes // 4
        for p in reversed(self.params):
            if cur and cur_numel + p.numel() > cap:
                self._close(cur)
                cur, cur_numel = [], 0
            cur.append(p)
            cur_numel += p.numel()
        if cur:
            self._close(cur)
        self._expected = [len(plist) for _, plist in self.buckets]     # gradients a bucket waits for before launching
        self._unused = None          # static_graph: indices (into self.params) without gradient on any rank
        self._reset_step()
        self._handles = [p.register_post_accumulate_grad_hook(self._hook) for p in self.params]

    def _close(self, plist):
        dev, dt = plist[0].device, plist[0].dtype
        flat = torch.zeros(sum(p.numel() for p in plist), device=dev, dtype=dt)
        self.buckets.append((flat, plist))
        for p in plist:
            self._bucket_of[p] = len(self.buckets) - 1
        self._attach(len(self.buckets) - 1)

    def _attach(self, b):
        """(Re)binds the gradients of bucket b's parameters to their slices of the flat buffer."""
        flat, plist = self.buckets[b]
        off = 0
        for p in plist:
            if p.grad is None or p.grad.data_ptr() != flat.data_ptr() + off * flat.element_size():
                p.grad = flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def _reset_step(self):
        self._ready = [0] * len(self.buckets)
        self._next = 0               # buckets [0, _next) have been launched
        self._seen = [False] * len(self.params)
        self._streams = [{} for _ in self.buckets]      # per bucket: GPU streams its gradients were accumulated on

    def _hook(self, p):
        self._seen[self._index[p]] = True
        b = self._bucket_of[p]
        self._ready[b] += 1
        if p.is_cuda:                # (the hook runs on the stream of the parameter's gradient accumulation)
            st = torch.cuda.current_stream(p.device)
            self._streams[b][st.cuda_stream] = st
        self._launch_ready()

    def _launch_ready(self):
        """Launches, in index order, every bucket that is complete and whose predecessors have been launched."""
        while self._next < len(self.buckets) and self._ready[self._next] >= self._expected[self._next]:
            self._launch(self._next)
            self._next += 1

    def _launch(self, b):
        flat = self.buckets[b][0]
        if flat.is_cuda and self._streams[b]:
            # every kernel that wrote into this bucket has been issued by now (the last hook fired after it): order the
            # launching stream behind the other streams that took part, then the collective behind the launching stream
            cur = torch.cuda.current_stream(flat.device)
            for handle, st in self._streams[b].items():
                if handle != cur.cuda_stream:
                    cur.wait_stream(st)
        if _TRACE:
            import threading
            print(f"[ddp rank {dist.get_rank() if dist.is_initialized() else 0}] launch bucket {b} ({flat.numel()} values) from "
                  f"{threading.current_thread().name}, ready {self._ready} expected {self._expected}", flush=True)
        work = self._collective(flat)
        if work is not None:
            self._pending.append((work, flat))

    def _collective(self, flat):
        """Sum of `flat` over the ranks, asynchronous, ordered behind the current stream; None when there is nothing to wait
        for (one rank)."""
        if self.world > 1:
            return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return None

    def finish(self):
        """Call after backward(): launches what the hooks could not (buckets holding a parameter without gradient),
        waits, turns sums into means and drops the gradient of every parameter no rank used."""
        while self._next < len(self.buckets):
            self._launch(self._next)
            self._next += 1
        if self.static_graph and self._unused is not None:
            unused = self._unused
        else:
            flags = torch.tensor([0 if s else 1 for s in self._seen], dtype=torch.int32, device=self.buckets[0][0].device)
            if self.world > 1:       # unused everywhere <=> the minimum over ranks is still 1
                dist.all_reduce(flags, op=dist.ReduceOp.MIN, group=self.group)
            unused = flags.nonzero().flatten().tolist()
            if self.static_graph:
                self._unused = unused
                drop = set(unused)
                self._expected = [sum(1 for p in plist if self._index[p] not in drop) for _, plist in self.buckets]
        for k, (work, flat) in enumerate(self._pending):
            if _TRACE:
                print(f"[ddp rank {dist.get_rank()}] waiting for collective {k} of {len(self._pending)}", flush=True)
            work.wait()
            flat.div_(self.world)
        self._pending.clear()
        for i in unused:
            self.params[i].grad = None
        self._reset_step()

    def zero_grad(self):
        """Zero the flat buckets in place and re-attach every gradient view (finish() detached the unused ones, and
        an optimiser may have replaced .grad)."""
        for b, (flat, _) in enumerate(self.buckets):
            flat.zero_()
            self._attach(b)

    def remove(self):
        for h in self._handles:
            h.remove()
