"""Data-parallel logic on CPU: world_size-2 gloo process groups (no GPU).

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
    x = full[rank * 2:(rank + 1) * 2]
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

    flat = grads(x_full[rank * 2:(rank + 1) * 2])       # this rank's flat gradient buffer
    dist.all_reduce(flat)                                # what MMHandModel._allreduce_async does
    flat *= 1.0 / world                                  # folded into mmh_adam_step(grad_scale)
    ref = grads(x_full)
    assert torch.allclose(flat, ref, rtol=1e-4, atol=1e-6), (flat - ref).abs().max()
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
                                          (_w_options, 29613)])
def test_world2_gloo(worker, port):
    mp.spawn(worker, args=(2, port), nprocs=2, join=True)
