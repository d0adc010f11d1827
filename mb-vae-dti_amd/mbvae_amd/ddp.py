"""Data-parallel gradient synchronisation over torch.distributed (RCCL on MI355X, gloo in CPU tests).

Gradients live as views into a few large flat buckets (filled in reverse registration order, i.e. roughly the
order backward produces them).  A post-accumulate hook counts ready parameters per bucket; the moment a bucket is
complete its all-reduce is launched asynchronously, so the collective overlaps the rest of backward.  xGMI is a
point-to-point mesh, so few large messages beat many small ones: the default bucket is 64 MiB.
"""
import torch
import torch.distributed as dist


class BucketedGradSync:
    def __init__(self, params, bucket_bytes=64 << 20, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.params = [p for p in params if p.requires_grad]
        self.buckets = []            # (flat tensor, [params])
        self._bucket_of = {}
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
        self._ready = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._handles = []
        for p in self.params:
            self._handles.append(p.register_post_accumulate_grad_hook(self._hook))

    def _close(self, plist):
        dev, dt = plist[0].device, plist[0].dtype
        flat = torch.zeros(sum(p.numel() for p in plist), device=dev, dtype=dt)
        off = 0
        for p in plist:
            p.grad = flat[off:off + p.numel()].view_as(p)
            off += p.numel()
            self._bucket_of[p] = len(self.buckets)
        self.buckets.append((flat, plist))

    def _hook(self, p):
        b = self._bucket_of[p]
        self._ready[b] += 1
        if self._ready[b] == len(self.buckets[b][1]):
            self._launch(b)

    def _launch(self, b):
        self._launched[b] = True
        if self.world > 1:
            flat = self.buckets[b][0]
            self._pending.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True), flat))

    def finish(self):
        """Wait for the outstanding all-reduces and turn sums into means.  Call after backward().  A bucket holding a
        parameter that received no gradient in this step (e.g. the global-feature output head when the loss only uses
        the node and edge logits) never fills up during backward: it is reduced here, with zeros for that parameter,
        so every rank still ends the step with identical gradients."""
        for b in range(len(self.buckets)):
            if not self._launched[b]:
                self._launch(b)
        for work, flat in self._pending:
            work.wait()
            flat.div_(self.world)
        self._pending.clear()
        self._ready = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)

    def zero_grad(self):
        """Zero the flat buckets in place (gradients stay views: never set them to None)."""
        for flat, plist in self.buckets:
            flat.zero_()
            off = 0
            for p in plist:                      # re-attach in case an optimizer replaced .grad
                if p.grad is None or p.grad.data_ptr() != flat.data_ptr() + off * flat.element_size():
                    p.grad = flat[off:off + p.numel()].view_as(p)
                off += p.numel()

    def remove(self):
        for h in self._handles:
            h.remove()
