"""Data-parallel logic on CPU: world_size-2 and world_size-4 gloo process groups (no GPU).

What can run without the HIP kernels: the SyncBN statistics merge used by NormActFn
(mmhand_amd.ops._sync_stats), the all-reduce(SUM) x 1/world gradient rule of MMHandModel, and the
per-rank batch split of the options parser.  The kernels themselves are covered by -m gpu tests;
the N-GPU run is the driver's."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _w_sync_stats(rank, world, port):
    _init(rank, world, port)
    from mmhand_amd import ops
    g = torch.Generator().manual_seed(7)
    full = torch.randn((4, 6, 5, 8), generator=g) * 3 + 2          # global batch, NHWC
    per = 4 // world
    x = full[rank * per:(rank + 1) * per]
    rows = x.shape[0] * x.shape[1] * x.shape[2]
    flat = x.reshape(-1, 8)
    mean = flat.mean(0, keepdim=True)
    m2 = ((flat - mean) ** 2).sum(0, keepdim=True)
    gmean, gm2, count = ops._sync_stats(mean.contiguous(), m2.contiguous(), rows, dist.group.WORLD)
    ff = full.reshape(-1, 8)
    assert count == ff.shape[0]
    assert torch.allclose(gmean[0], ff.mean(0), atol=1e-5)
    assert torch.allclose(gm2[0] / count, ff.var(0, unbiased=False), rtol=1e-4)
    dist.destroy_process_group()


def _w_sync_stats_packed(rank, world, port):
    """several norm sites, ONE collective each way (ops._sync_stats_multi / _sync_bwd_sums_multi): the same global
    statistics and sums as one collective per site"""
    _init(rank, world, port)
    from mmhand_amd import ops
    g = torch.Generator().manual_seed(11)
    items, fulls = [], []
    for Cc in (8, 16, 4):
        full = torch.randn((4, 3, 5, Cc), generator=g) * 2 + 1
        x = full[rank * (4 // world):(rank + 1) * (4 // world)].reshape(-1, Cc)
        mean = x.mean(0, keepdim=True)
        items.append((mean.contiguous(), ((x - mean) ** 2).sum(0, keepdim=True).contiguous(), x.shape[0]))
        fulls.append(full.reshape(-1, Cc))
    ops.collective_counter.clear()
    packed = ops._sync_stats_multi(items, dist.group.WORLD)
    assert ops.collective_counter == {"all_gather": 1, "packed_sites": 3}, ops.collective_counter
    for (gmean, gm2, count), it, ff in zip(packed, items, fulls):
        smean, sm2, scount = ops._sync_stats(*it, dist.group.WORLD)
        assert count == scount == ff.shape[0]
        assert torch.allclose(gmean, smean, atol=1e-6) and torch.allclose(gm2, sm2, rtol=1e-5)
        assert torch.allclose(gmean[0], ff.mean(0), atol=1e-5)
        assert torch.allclose(gm2[0] / count, ff.var(0, unbiased=False), rtol=1e-4)
    pairs = [(torch.full((1, c), float(rank + 1)), torch.arange(c, dtype=torch.float32).view(1, c) * (rank + 1)) for c in (8, 4)]
    ops.collective_counter.clear()
    red = ops._sync_bwd_sums_multi(pairs, dist.group.WORLD)
    assert ops.collective_counter["all_reduce"] == 1
    for (s1, s2), c in zip(red, (8, 4)):
        tot = float(world * (world + 1) // 2)        # sum over ranks of (rank + 1)
        assert torch.equal(s1, torch.full((1, c), tot)) and torch.equal(s2, torch.arange(c, dtype=torch.float32).view(1, c) * tot)
    dist.destroy_process_group()


def _w_grad_average(rank, world, port):
    """sum over ranks of shard gradients x 1/world == gradient of the loss on the global batch
    (InstanceNorm: samples independent; losses are means)."""
    _init(rank, world, port)
    from oracle import mmhand_ref as O
    from tests.golden import recipe as RC
    from mmhand_amd.networks import Discriminator
    net = Discriminator(6, 8, "instance", False, 1)
    sd = RC.recipe_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()})
    x_full = RC.rand("dp.x", (4, 6, 16, 16))

    def grads(x):
        n = O._Net(sd, "instance", False)
        O.gan_loss(O.discriminator_forward(n, x, 1), True).backward()
        return torch.cat([p.grad.reshape(-1) for p in n.parameters()])

    per = 4 // world
    flat = grads(x_full[rank * per:(rank + 1) * per])   # this rank's flat gradient buffer
    dist.all_reduce(flat)                                # what MMHandModel._allreduce_async does
    flat *= 1.0 / world                                  # folded into mmh_adam_step(grad_scale)
    ref = grads(x_full)
    assert torch.allclose(flat, ref, rtol=1e-4, atol=1e-6), (flat - ref).abs().max()
    dist.destroy_process_group()


def _w_grad_buckets(rank, world, port):
    """dp.GradBuckets on a 6-layer toy network with flat parameter / gradient buffers: buckets are cut
    from the END of the flat buffer, each bucket's all-reduce is issued from the autograd hook that
    completes it - i.e. BEFORE the backward pass has reached the earlier layers - and the result is
    the sum over ranks of the full flat gradient."""
    _init(rank, world, port)
    from mmhand_amd.dp import GradBuckets
    torch.manual_seed(3)
    layers = [torch.nn.Linear(16, 16) for _ in range(6)]
    net = torch.nn.Sequential(*[m for l in layers for m in (l, torch.nn.Tanh())])
    ps = list(net.parameters())
    n = sum(p.numel() for p in ps)
    flat, gflat = torch.zeros(n), torch.zeros(n)
    off = 0
    for p in ps:
        k = p.numel()
        flat[off:off + k].copy_(p.detach().reshape(-1))
        p.data = flat[off:off + k].view(p.shape)
        p.grad = gflat[off:off + k].view(p.shape)
        off += k
    log = []
    bk = GradBuckets(ps, gflat, bucket_bytes=2 * 272 * 4, log=log)       # two layers (w + b) per bucket
    assert [(s, e) for s, e, _ in bk.buckets] == [(4 * 272, 6 * 272), (2 * 272, 4 * 272), (0, 2 * 272)]
    x = torch.randn(8, 16, generator=torch.Generator().manual_seed(10 + rank))
    for it in range(2):                                                   # re-armed per backward pass
        gflat.zero_()
        log.clear()
        bk.begin()
        net(x).square().mean().backward()
        bk.launch_remaining()
        bk.wait()
        # reference: plain per-rank gradients, summed over ranks
        ref = torch.autograd.grad(net(x).square().mean(), ps)
        ref = torch.cat([g.reshape(-1) for g in ref])
        dist.all_reduce(ref)
        assert torch.allclose(gflat, ref, rtol=1e-5, atol=1e-7), (gflat - ref).abs().max()
        order = [e for e in log if e[0] == "bucket"]
        assert order == [("bucket", 0), ("bucket", 1), ("bucket", 2)], order
        # bucket 0 (layers 5, 6) went out before any parameter of layers 1..4 had its gradient
        first_early = min(i for i, e in enumerate(log) if e[0] == "param" and e[1] < 8)
        assert log.index(("bucket", 0)) < first_early, log
        assert log.index(("bucket", 1)) < min(i for i, e in enumerate(log) if e[0] == "param" and e[1] < 4)
    # a parameter that gets no gradient this pass: finish() still reduces every bucket
    gflat.zero_()
    bk.begin()
    layers[5](x).sum().backward()
    bk.finish()
    assert float(gflat[: 5 * 272].abs().sum()) == 0.0 and float(gflat[5 * 272:].abs().sum()) > 0.0
    bk.remove()
    dist.destroy_process_group()


def _w_grad_buckets_inplace(rank, world, port):
    """The second way a parameter gradient completes (dp.param_use / dp.param_done): a shim that adds its weight gradient
    into the flat buffer ITSELF and returns None to autograd - what the conv shims do under ops.ACCUM_PARAM_GRADS - beside
    parameters that still go through AccumulateGrad (the biases here).  Each layer is used TWICE in one pass (as a
    discriminator on its real and its fake batch): a weight is complete after its second contribution only, and the
    buckets still go out in reverse layer order, before the early layers are reached, with the sum over ranks in them."""
    _init(rank, world, port)
    from mmhand_amd import dp
    from mmhand_amd.dp import GradBuckets

    class InplaceLinear(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w, b):
            dp.param_use(w)
            ctx.save_for_backward(x, w)
            return x @ w.t() + b

        @staticmethod
        def backward(ctx, g):
            x, w = ctx.saved_tensors
            w.grad.add_(g.t() @ x)          # in place, into the view of the flat gradient buffer
            dp.param_done(w)
            return g @ w, None, g.sum(0)    # the bias gradient still travels through AccumulateGrad

    torch.manual_seed(5)
    n_layers, width = 4, 8
    per = width * width + width
    flat, gflat = torch.zeros(n_layers * per), torch.zeros(n_layers * per)
    ws, bs, ps, off = [], [], [], 0
    for _ in range(n_layers):
        for shape, lst in (((width, width), ws), ((width,), bs)):
            k = 1
            for d in shape:
                k *= d
            flat[off:off + k] = torch.randn(k) * 0.3
            p = torch.nn.Parameter(torch.zeros(shape))
            p.data = flat[off:off + k].view(shape)
            p.grad = gflat[off:off + k].view(shape)
            lst.append(p); ps.append(p)
            off += k
    log = []
    bk = GradBuckets(ps, gflat, bucket_bytes=per * 4, log=log, flat_param=flat)     # one layer per bucket
    assert len(bk.buckets) == n_layers

    def net(x):
        for w, b in zip(ws, bs):
            x = torch.tanh(InplaceLinear.apply(x, w, b))
        return x

    xa = torch.randn(6, width, generator=torch.Generator().manual_seed(20 + rank))
    xb = torch.randn(6, width, generator=torch.Generator().manual_seed(40 + rank))
    gflat.zero_()
    bk.begin()
    (net(xa).square().mean() + net(xb).square().mean()).backward()      # every parameter used twice
    bk.launch_remaining()
    bk.wait()
    # reference: ordinary autograd on plain tensors, summed over ranks
    rw = [w.detach().clone().requires_grad_() for w in ws]
    rb = [b.detach().clone().requires_grad_() for b in bs]

    def ref_net(x):
        for w, b in zip(rw, rb):
            x = torch.tanh(x @ w.t() + b)
        return x
    (ref_net(xa).square().mean() + ref_net(xb).square().mean()).backward()
    ref = torch.cat([t.grad.reshape(-1) for pair in zip(rw, rb) for t in pair])
    dist.all_reduce(ref)
    assert torch.allclose(gflat, ref, rtol=1e-5, atol=1e-7), (gflat - ref).abs().max()
    assert [e for e in log if e[0] == "bucket"] == [("bucket", i) for i in range(n_layers)], log
    # the last layer's bucket went out before the first layer's weight (parameter 0) was complete, and no weight was
    # reported complete twice
    assert log.index(("bucket", 0)) < log.index(("param", 0))
    assert len([e for e in log if e[0] == "param"]) == 2 * n_layers
    # a `done` without a counted `use` - the forward ran BEFORE begin() - must not complete the parameter early
    # (its bucket would go out under a later in-place add): the shim path leaves such a parameter alone and it completes
    # where autograd's AccumulateGrad node of the weight runs - after BOTH contributions (the post-accumulate hook fires
    # for a weight whose shim returned no gradient, too) - or with launch_remaining().  A bucket that left between the
    # two in-place adds would hold the sum over ranks plus one rank-local term: the comparison below would see it.
    gflat.zero_()
    log.clear()
    ya, yb = net(xa), net(xb)               # the forward passes precede begin(): their uses are not counted
    bk.begin()
    (ya.square().mean() + yb.square().mean()).backward()
    assert all(bk._untracked[2 * i] and bk._dones[2 * i] == 0 for i in range(n_layers))     # the shim path stood aside
    bk.launch_remaining()
    bk.wait()
    assert torch.allclose(gflat, ref, rtol=1e-5, atol=1e-7), (gflat - ref).abs().max()
    assert [e for e in log if e[0] == "bucket"] == [("bucket", i) for i in range(n_layers)], log
    # the MIXED case - one forward before begin() (uncounted) and one after it (counted).  With an exact-match
    # rule alone the first of the two dones would match the single counted use and the bucket would leave before the second
    # in-place add; a use reported while unarmed therefore keeps the parameter untracked for the coming pass.
    gflat.zero_()
    log.clear()
    ya = net(xa)                            # before begin(): not counted, remembered as a stray use
    assert all(bk._stray[2 * i] for i in range(n_layers))
    bk.begin()
    yb = net(xb)                            # counted
    assert all(bk._uses[2 * i] == 1 and bk._untracked[2 * i] for i in range(n_layers))
    (ya.square().mean() + yb.square().mean()).backward()
    assert all(bk._dones[2 * i] == 0 for i in range(n_layers))          # no early completion through the shim path
    bk.launch_remaining()
    bk.wait()
    assert torch.allclose(gflat, ref, rtol=1e-5, atol=1e-7), (gflat - ref).abs().max()
    # ... and the pass after it is tracked again (the stray mark does not outlive one pass)
    gflat.zero_()
    log.clear()
    bk.begin()
    assert not any(bk._untracked)
    (net(xa).square().mean() + net(xb).square().mean()).backward()
    bk.launch_remaining()
    bk.wait()
    assert torch.allclose(gflat, ref, rtol=1e-5, atol=1e-7)
    assert log.index(("bucket", 0)) < log.index(("param", 0))
    bk.remove()
    assert not dp.tracking()
    dist.destroy_process_group()


def _w_options(rank, world, port):
    _init(rank, world, port)
    from mmhand_amd.options import TrainOptions
    opt = TrainOptions().parse(["--distributed", "--batchSize", "6", "--name", "dp",
                                "--checkpoints_dir", "/tmp/mmh_dp"], save=False)
    assert opt.world_size == 2 and opt.batchSize == 3 and opt.local_rank == rank
    assert opt.gpu_ids == [rank]
    dist.destroy_process_group()


@pytest.mark.parametrize("worker,port", [(_w_sync_stats, 29611), (_w_grad_average, 29612),
                                          (_w_options, 29613), (_w_grad_buckets, 29614), (_w_sync_stats_packed, 29615),
                                          (_w_grad_buckets_inplace, 29616)])
def test_world2_gloo(worker, port):
    mp.spawn(worker, args=(2, port), nprocs=2, join=True)


@pytest.mark.parametrize("worker,port", [(_w_sync_stats, 29661), (_w_grad_average, 29662), (_w_grad_buckets, 29664),
                                          (_w_sync_stats_packed, 29665), (_w_grad_buckets_inplace, 29666)])
def test_world4_gloo(worker, port):
    """the same workers at world size 4 (no world size above 2 had run, even functionally): statistics merged
    over four ranks' (count, mean, M2), gradients averaged over four shards, buckets cut and issued alike on four ranks"""
    mp.spawn(worker, args=(4, port), nprocs=4, join=True)
