"""Gradient all-reduce of one network, bucketed in reverse layer order and launched while the rest
of the backward pass is still running — what apex's DistributedDataParallel does for the reference
(models/MMHandModel.py:109-116), minus its flatten/unflatten copies: a network's gradients already
live in ONE flat fp32 buffer (networks._Net.flatten_parameters), so a bucket is a contiguous slice
of it and the collective runs in place.

Parameters sit in the flat buffer in forward order; the backward pass finishes them back to front.
The buffer is cut into contiguous buckets of about `bucket_bytes` from its END; a
post-accumulate-grad hook on every parameter counts arrivals per bucket, and the hook that completes
a bucket enqueues its all-reduce(SUM) on the side stream (which first waits for the compute stream,
i.e. for that accumulation).  RCCL over xGMI is point-to-point and per-link bound: buckets of a few
tens of MB keep the ring at bandwidth while leaving most of the 285 MB Generator gradient to travel
beneath the remaining backward and the discriminator passes.  The 1/world factor is folded into
the Adam kernel, so the collective is a plain SUM.

Every rank runs the same autograd graph, so buckets complete — and collectives are issued — in the
same order on every rank."""
import os

import torch
import torch.distributed as dist

DEFAULT_BUCKET_MB = float(os.environ.get("MMH_BUCKET_MB", "32"))


class GradBuckets:
    def __init__(self, params, flat_grad, bucket_bytes=None, group=None, comm_stream=None, log=None, name=None):
        """params: the network's parameters in flat-buffer order, each .grad a view of flat_grad.
        log: optional list that receives ("bucket", index) / ("param", index) events - prefixed with
        `name` when given - in the order they happen (tests assert the interleaving from it)."""
        self.flat = flat_grad
        self._tag = (name,) if name is not None else ()
        self.group = group
        self.comm_stream = comm_stream
        self.log = log
        bucket_elems = max(1, int((bucket_bytes if bucket_bytes is not None else DEFAULT_BUCKET_MB * 2 ** 20) // 4))
        params = list(params)
        offs, off = [], 0
        for p in params:
            assert p.grad is not None and p.grad.data_ptr() == flat_grad.data_ptr() + 4 * off, \
                "parameters must be views of the flat gradient buffer, in order"
            offs.append(off)
            off += p.numel()
        assert off == flat_grad.numel()
        # cut from the end: bucket 0 = the last parameters (first to be ready in the backward pass)
        self.buckets = []           # (start, end, [param indices])
        end, members = off, []
        for i in range(len(params) - 1, -1, -1):
            members.append(i)
            if end - offs[i] >= bucket_elems or i == 0:
                self.buckets.append((offs[i], end, members))
                end, members = offs[i], []
        self.bucket_of = {}
        for b, (_, _, mem) in enumerate(self.buckets):
            for i in mem:
                self.bucket_of[i] = b
        self._need = [len(m) for _, _, m in self.buckets]
        self._left = list(self._need)
        self._works = [None] * len(self.buckets)
        self._armed = False
        self._handles = [p.register_post_accumulate_grad_hook(self._make_hook(i)) for i, p in enumerate(params)]

    def _make_hook(self, i):
        def hook(_p):
            if not self._armed:
                return
            if self.log is not None:
                self.log.append(self._tag + ("param", i))
            b = self.bucket_of[i]
            self._left[b] -= 1
            if self._left[b] == 0:
                self._launch(b)
        return hook

    def _launch(self, b):
        if self._works[b] is not None:
            return
        s, e, _ = self.buckets[b]
        if self.log is not None:
            self.log.append(self._tag + ("bucket", b))
        if self.comm_stream is not None:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                self._works[b] = dist.all_reduce(self.flat[s:e], group=self.group, async_op=True)
        else:
            self._works[b] = dist.all_reduce(self.flat[s:e], group=self.group, async_op=True)

    def begin(self):
        """Arm the hooks for the next backward pass (call after zero_grad)."""
        self._left = list(self._need)
        self._works = [None] * len(self.buckets)
        self._armed = True

    def finish(self):
        """Launch whatever the backward pass did not complete (parameters without a gradient this
        pass), in bucket order, then make the current stream wait for every collective."""
        self._armed = False
        for b in range(len(self.buckets)):
            self._launch(b)
        for w in self._works:
            w.wait()
        self._works = [None] * len(self.buckets)

    def launch_remaining(self):
        """Enqueue the collectives of incomplete buckets without waiting (end of a backward pass)."""
        self._armed = False
        for b in range(len(self.buckets)):
            self._launch(b)

    def wait(self):
        for w in self._works:
            if w is not None:
                w.wait()
        self._works = [None] * len(self.buckets)

    def remove(self):
        for h in self._handles:
            h.remove()
