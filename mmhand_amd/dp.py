"""Gradient all-reduce of one network, bucketed in reverse layer order and launched while the rest
of the backward pass is still running — what apex's DistributedDataParallel does for the reference
(models/MMHandModel.py:109-116), minus its flatten/unflatten copies: a network's gradients already
live in ONE flat fp32 buffer (networks._Net.flatten_parameters), so a bucket is a contiguous slice
of it and the collective runs in place.

Parameters sit in the flat buffer in forward order; the backward pass finishes them back to front.
The buffer is cut into contiguous buckets of about `bucket_bytes` from its END; the bucket whose last
parameter gradient has landed enqueues its all-reduce(SUM) on the side stream (which first waits for
the compute stream, i.e. for that accumulation).  RCCL over xGMI is point-to-point and per-link
bound: buckets of a few tens of MB keep the ring at bandwidth while leaving most of the 285 MB
Generator gradient to travel beneath the remaining backward and the discriminator passes.  The
1/world factor is folded into the Adam kernel, so the collective is a plain SUM.

When is a parameter's gradient complete?  Two sources, both counted here:
  * autograd's AccumulateGrad ran for it (a post-accumulate-grad hook): norm scales / shifts, and every
    parameter while ops.ACCUM_PARAM_GRADS is off;
  * the conv shims added it into the flat buffer themselves (the wgrad kernels' accumulate path, the
    same in-place accumulation single-process training uses: no AccumulateGrad add kernels, 239 launches
    per step).  A shim reports every USE of a parameter in its forward (`use`) and every finished
    contribution in its backward (`done`); a network that runs twice in one pass (a discriminator on its
    real and its fake batch) completes a parameter after the second contribution.

The collective itself: on RCCL (backend "nccl") one ncclAllReduce per bucket through the C-ABI
(mmh_allreduce_bucket) on torch.distributed's OWN communicator and the side stream - no Work objects,
no second communicator; MMH_DP_NATIVE=0, or any other backend (the gloo tests), goes through
dist.all_reduce(async_op=True).

Every rank runs the same autograd graph, so buckets complete — and collectives are issued — in the
same order on every rank."""
import ctypes as C
import os
import weakref

import torch
import torch.distributed as dist

DEFAULT_BUCKET_MB = float(os.environ.get("MMH_BUCKET_MB", "32"))
# Timing aid (bench.py `comm_exposed_ms`): with NO_COMM set no collective is issued - gradients stay rank-local
# (the replicas drift apart: never train with it).  Also read by the SyncBN collectives in ops.py.
NO_COMM = os.environ.get("MMH_DP_NO_COMM") == "1"


def set_no_comm(flag):
    global NO_COMM
    NO_COMM = bool(flag)
    from . import ops
    ops.DP_NO_COMM = NO_COMM


def native_comm(group=None):
    """(ncclComm_t as int, None) of torch.distributed's RCCL communicator for `group` with mmh_allreduce_bucket bound to
    the RCCL image it lives in, or (None, reason) - then the buckets go through dist.all_reduce."""
    mode = os.environ.get("MMH_DP_NATIVE", "auto")
    if mode == "0":
        return None, "MMH_DP_NATIVE=0"
    if not dist.is_initialized() or dist.get_backend(group) != "nccl":
        return None, "backend is not nccl(RCCL)"
    # the native path shares torch.distributed's communicator behind ProcessGroupNCCL's back and has only
    # ever run with ONE rank.  Until a world > 1 hardware run (with SyncBN's collectives interleaved) has passed it is
    # opt-in there: MMH_DP_NATIVE=1 forces it, the default ("auto") uses it at world size 1 only.
    if mode != "1" and dist.get_world_size(group) > 1:
        return None, "world > 1: dist.all_reduce is the shipped path (MMH_DP_NATIVE=1 opts in)"
    try:
        pg = group if group is not None else dist.group.WORLD
        backend = pg._get_backend(torch.device("cuda", torch.cuda.current_device()))
        if not hasattr(backend, "_comm_ptr"):
            return None, "this torch build does not expose the communicator"
        comm = int(backend._comm_ptr())
        if not comm:
            return None, "communicator not initialised yet"
        from . import lib as L
        path = loaded_rccl_path()
        if path is None:
            return None, "no librccl image is mapped into this process"
        L.call("mmh_rccl_bind", path.encode())
        n = L.load().mmh_rccl_comm_ranks(C.c_void_p(comm))
        if n != dist.get_world_size(group):
            return None, f"communicator reports {n} ranks, the group has {dist.get_world_size(group)}"
        return comm, None
    except Exception as e:      # noqa: BLE001 - any surprise in torch's private API: the torch path still works
        return None, f"{type(e).__name__}: {e}"


def loaded_rccl_path():
    """Path of the RCCL image this process has ALREADY mapped (the one torch.distributed's communicator lives in), read
    from /proc/self/maps - never a guess at torch/lib: binding a second RCCL image to a communicator created by another
    is undefined behaviour."""
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.rstrip("\n").split(None, 5)[-1] if line.count("/") else ""
                if "/" in path and os.path.basename(path).startswith("librccl.so"):
                    return path
    except OSError:
        pass
    return None


_trackers = []      # weak references to live GradBuckets objects: the conv shims report parameter uses / contributions


def _tracker_of(p, armed_only=True):
    a = p.data_ptr()
    dead = False
    for r in _trackers:
        t = r()
        if t is None:
            dead = True
        elif (t._armed or not armed_only) and t._lo <= a < t._hi:
            return t
    if dead:
        _trackers[:] = [r for r in _trackers if r() is not None]
    return None


def tracking():
    """any GradBuckets object registered (cheap test for the shims)"""
    return bool(_trackers)


def param_use(p):
    """a shim's forward will contribute to p's gradient in the coming backward pass"""
    t = _tracker_of(p, armed_only=False)      # an unarmed tracker notes the use as a stray one (GradBuckets.use)
    if t is not None:
        t.use(p)


def param_done(p):
    """a shim's backward has added its contribution to p.grad in place (or had none to add)"""
    t = _tracker_of(p)
    if t is not None:
        t.done(p)


class GradBuckets:
    def __init__(self, params, flat_grad, bucket_bytes=None, group=None, comm_stream=None, log=None, name=None,
                 flat_param=None, native=None):
        """params: the network's parameters in flat-buffer order, each .grad a view of flat_grad.
        log: optional list that receives ("bucket", index) / ("param", index) events - prefixed with
        `name` when given - in the order they happen (tests assert the interleaving from it).
        flat_param: the network's flat parameter buffer (enables the shims' use / done reports).
        native: an ncclComm_t (int) - the buckets' collectives go through mmh_allreduce_bucket."""
        self.flat = flat_grad
        self._tag = (name,) if name is not None else ()
        self.group = group
        self.comm_stream = comm_stream
        self.log = log
        self.native = native if (native and comm_stream is not None) else None
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
        self._n = len(params)
        self._uses = [0] * self._n
        self._dones = [0] * self._n
        self._untracked = [False] * self._n
        self._stray = [False] * self._n       # a shim forward used the parameter while the buckets were not armed
        self._complete = [False] * self._n
        self._handles = [p.register_post_accumulate_grad_hook(self._make_hook(i)) for i, p in enumerate(params)]
        # the shims find a parameter by the address of its slice of the flat PARAMETER buffer
        self._index, self._lo, self._hi = {}, 0, 0
        if flat_param is not None:
            self._lo = flat_param.data_ptr()
            self._hi = self._lo + flat_param.numel() * flat_param.element_size()
            self._index = {p.data_ptr(): i for i, p in enumerate(params) if p.numel()}
            _trackers.append(weakref.ref(self))

    def _make_hook(self, i):
        def hook(_p):
            if self._armed:
                self._param_complete(i)     # AccumulateGrad sums every use before it runs: one call = all of them
        return hook

    def use(self, p):
        i = self._index.get(p.data_ptr())
        if i is None:
            return
        if self._armed:
            self._uses[i] += 1
        else:
            # the backward of this forward will report a done that no counted use matches; together with counted uses of
            # the same parameter the exact-match rule of done() would then fire one contribution early: the
            # parameter stays untracked for the coming pass and its bucket goes out with launch_remaining()
            self._stray[i] = True

    def done(self, p):
        """Complete on an EXACT match only: every counted use has reported done.  A done without a counted use (the
        forward ran before begin(), or a shim reported twice) leaves the parameter untracked - its bucket then goes
        out with launch_remaining() at the end of the pass, never early under a later in-place add.  So does
        any use reported while unarmed since the last pass ended (`_stray`, set by use(), taken over by begin()): one
        counted and one uncounted use would otherwise match on the first of their two dones."""
        i = self._index.get(p.data_ptr())
        if i is None or not self._armed:
            return
        if self._uses[i] <= 0 or self._untracked[i]:
            self._untracked[i] = True
            return
        self._dones[i] += 1
        if self._dones[i] == self._uses[i]:
            self._param_complete(i)

    def _param_complete(self, i):
        if self._complete[i]:
            return
        self._complete[i] = True
        if self.log is not None:
            self.log.append(self._tag + ("param", i))
        b = self.bucket_of[i]
        self._left[b] -= 1
        if self._left[b] == 0:
            self._launch(b)

    def _launch(self, b):
        if self._works[b] is not None:
            return
        s, e, _ = self.buckets[b]
        if self.log is not None:
            self.log.append(self._tag + ("bucket", b))
        if NO_COMM:
            self._works[b] = False
            return
        if self.native is not None:
            from . import lib as L
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            L.call("mmh_allreduce_bucket", C.c_void_p(self.native), C.c_void_p(self.flat.data_ptr() + 4 * s), e - s, L.F32,
                   C.c_void_p(self.comm_stream.cuda_stream))
            self._works[b] = True
        elif self.comm_stream is not None:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                self._works[b] = dist.all_reduce(self.flat[s:e], group=self.group, async_op=True)
        else:
            self._works[b] = dist.all_reduce(self.flat[s:e], group=self.group, async_op=True)

    def begin(self):
        """Arm the hooks for the next backward pass (call after zero_grad and BEFORE the forward pass whose
        parameter uses the shims report)."""
        self._left = list(self._need)
        self._works = [None] * len(self.buckets)
        self._uses = [0] * self._n
        self._dones = [0] * self._n
        self._untracked = list(self._stray)
        self._stray = [False] * self._n
        self._complete = [False] * self._n
        self._armed = True

    def finish(self):
        """Launch whatever the backward pass did not complete (parameters without a gradient this
        pass), in bucket order, then make the current stream wait for every collective."""
        self.launch_remaining()
        self.wait()

    def launch_remaining(self):
        """Enqueue the collectives of incomplete buckets without waiting (end of a backward pass)."""
        self._armed = False
        for b in range(len(self.buckets)):
            self._launch(b)

    def wait(self):
        """the CURRENT stream waits for this network's collectives"""
        native = False
        for w in self._works:
            if w is True:
                native = True
            elif w is not None and w is not False:
                w.wait()
        if native:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        self._works = [None] * len(self.buckets)

    def remove(self):
        for h in self._handles:
            h.remove()
        _trackers[:] = [r for r in _trackers if r() is not None and r() is not self]
