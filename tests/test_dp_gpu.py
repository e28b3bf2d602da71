"""Data-parallel MMHandModel on real kernels: 2 processes sharing cuda:0 over the gloo backend
(RCCL refuses two ranks on one device; the N-GPU RCCL run is the driver's).  Checks that a
2-rank step equals the 1-rank step on the concatenated batch: exactly the apex-DDP semantics the
reference relies on (gradient = mean over ranks; SyncBN statistics over the global batch)."""
import os
import random
import sys
from collections import OrderedDict

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _opt(norm, B, distributed, rank=0, wide=False, dev=0):
    from mmhand_amd.options import default_train_opt
    # wide: ngf = ndf = 32 at 64x64 - the 3x3 stack runs on Winograd F(6x6,3x3) with the norm between its
    # convs applied inside their transforms (ops.USE_NORM_FUSION), as at full size
    return default_train_opt(batchSize=B, ngf=32 if wide else 8, ndf=32 if wide else 8, n_layers_D=2,
                             G_n_blocks=1 if wide else 2, norm=norm, fineSize=64 if wide else 32,
                             no_dropout=True, no_dropout_D=True, pool_size=0, name="dp",
                             checkpoints_dir="/tmp/mmh_dp_gpu", local_rank=dev, distributed=distributed)


def _run(norm, batch, distributed, wide=False, dev=0, native=None):
    from mmhand_amd.mmhand_model import MMHandModel
    random.seed(0)
    model = MMHandModel(_opt(norm, batch["H1"].shape[0], distributed, wide=wide, dev=dev))
    if native is not None:
        native.append((bool(getattr(model, "dp_native", None)), bool(getattr(model, "dp_accum", False))))
    out = []
    for _ in range(2):
        model.set_input(batch)
        model.optimize_parameters()
        out.append([float(v) for v in model.get_current_errors().values()])
    torch.cuda.synchronize()
    sd = OrderedDict((k, v.detach().cpu()) for k, v in model.netG.state_dict().items())
    return out, sd


def _worker(rank, world, port, norm, tmp, wide=False, pack=True, backend="gloo"):
    """backend "gloo": both ranks on cuda:0 (RCCL refuses two ranks on one device); "nccl": one GPU per rank on RCCL"""
    dev = rank if backend == "nccl" else 0
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(dev), MMH_DP_LOG="1", MMH_BUCKET_MB="0.02",
                      MMH_PACK_SYNCBN="1" if pack else "0", NCCL_SOCKET_IFNAME="lo")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.path.insert(0, ROOT)
    from oracle import mmhand_ref as O
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    S = 64 if wide else 32
    full = O.synthetic_batch(4, S, S, seed=7)
    shard = {k: v[rank * 2:(rank + 1) * 2] for k, v in full.items()}
    from mmhand_amd import lib, mmhand_model, ops
    fused = []
    real = lib.call

    def spy(name, *a):
        if name in ("mmh_wino_input_normact", "mmh_wino_input_dy_normbwd"):
            fused.append(name)
        return real(name, *a)
    lib.call = spy
    native = []
    losses, sd = _run(norm, shard, True, wide, dev=dev, native=native)
    lib.call = real
    torch.save({"losses": losses, "sd": sd, "log": list(mmhand_model._LAST_BUCKET_LOG or []),
                "syncbn": dict(ops.collective_counter), "fused": len(fused), "native": native[0]},
               os.path.join(tmp, f"rank{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("norm,port,wide,pack", [("instance", 29621, False, True), ("batch", 29622, False, True),
                                                 ("batch", 29623, True, True), ("instance", 29624, True, True),
                                                 ("batch", 29625, False, False)])
def test_two_ranks_equal_one_rank_on_concatenated_batch(norm, port, wide, pack, dev, tmp_path):
    from oracle import mmhand_ref as O
    from tests.golden.recipe import is_null_grad_bias
    mp.spawn(_worker, args=(2, port, norm, str(tmp_path), wide, pack), nprocs=2, join=True)
    r0 = torch.load(os.path.join(str(tmp_path), "rank0.pt"))
    r1 = torch.load(os.path.join(str(tmp_path), "rank1.pt"))
    S = 64 if wide else 32
    full = O.synthetic_batch(4, S, S, seed=7)
    ref_losses, ref_sd = _run(norm, full, False, wide)
    # wide: the fused norm kernels ran under SyncBN (statistics and backward sums all-reduced around them)
    from mmhand_amd import ops
    assert (r0["fused"] > 0) == (wide and ops.USE_NORM_FUSION), r0["fused"]
    # replicas stay identical
    for k in r0["sd"]:
        assert torch.equal(r0["sd"][k], r1["sd"][k]), k
    # SyncBN collectives per iteration (--norm batch).  One per norm SITE (MMH_PACK_SYNCBN=0, what apex does): G has
    # 3 stems x 3 + n_blocks x 4 + 2 up = 19 sites here (47 at full size); D 1 + 2 + 2 x n_layers_D = 7 per pass (9 at
    # full size), 6 passes per iteration (2 in the G step, 2 per discriminator step) -> 61 all-gathers + 61 all-reduces
    # here, 101 + 101 at full size.  PACKED (default; networks.normact_multi): the three generator streams share one
    # collective per depth - 3 (stem, two downs) + 2 per PATBlock + 2 up = 9 here (23 at full size) - and so do the
    # discriminator passes that run side by side - D_PB and D_PP in the generator step, the real and the fake batch in
    # each discriminator step: 7 per pair, three pairs -> 30 + 30 here, 50 + 50 at full size.  --norm instance: none.
    sb = dict(r0["syncbn"])
    packed = sb.pop("packed_sites", 0)
    if norm == "batch" and wide:
        assert sb["all_gather"] == sb["all_reduce"] > 0 and packed > 0, r0["syncbn"]
    elif norm == "batch" and pack:
        assert sb == {"all_gather": 2 * (9 + 3 * 7), "all_reduce": 2 * (9 + 3 * 7)} and packed > 0, r0["syncbn"]
    elif norm == "batch":
        assert sb == {"all_gather": 2 * (19 + 6 * 7), "all_reduce": 2 * (19 + 6 * 7)} and packed == 0, r0["syncbn"]
    else:
        assert sb == {}, r0["syncbn"]
    assert r0["native"] == (False, True), r0["native"]      # gloo: dist.all_reduce; wgrads accumulated in place under DP
    # gradient buckets (20 KB here) went out in reverse layer order DURING the backward pass: the
    # Generator's first bucket (its last layers) was issued before the gradients of its first
    # layers existed, and both ranks issued their collectives in the same order
    log = r0["log"]
    assert [e for e in log if e[1] == "bucket"] == [e for e in r1["log"] if e[1] == "bucket"]
    g_buckets = [e[2] for e in log if e[0] == "G" and e[1] == "bucket"]
    per_iter = max(g_buckets) + 1
    assert per_iter >= 3 and g_buckets[0] == 0 and sorted(g_buckets[:per_iter]) == list(range(per_iter)), g_buckets
    first_layer = next(i for i, e in enumerate(log) if e[:2] == ("G", "param") and e[2] == 0)
    early = [e for e in log[:first_layer] if e[:2] == ("G", "bucket")]
    assert ("G", "bucket", 0) in early and len(early) >= per_iter // 2, (early, per_iter)
    # mean of the rank losses == loss on the global batch (losses are means over the batch)
    mean_losses = (np.array(r0["losses"]) + np.array(r1["losses"])) / 2
    if norm == "instance":
        assert np.allclose(mean_losses, np.array(ref_losses), rtol=2e-4), (mean_losses, ref_losses)
    # parameters after two steps == single-process training on the concatenated batch
    # (wide: two Adam SIGN steps of lr = 2e-4 each - an element whose gradient is within rounding of zero
    # flips with the summation order, tests/test_winograd_step_gpu.py - so up to 4 lr apart, few of them)
    for k, v in ref_sd.items():
        if v.is_floating_point() and not is_null_grad_bias("G", k, norm):
            if wide:
                if "running" in k:
                    continue
                assert torch.allclose(r0["sd"][k], v, atol=4.2 * 2e-4 + 1e-3 * v.abs().max().item()), k
                assert float(((r0["sd"][k] - v).abs() > 1e-4).float().mean()) < 0.25, k
            else:
                assert torch.allclose(r0["sd"][k], v, atol=2e-4 + 1e-3 * v.abs().max().item()), k


def _worker_overflow(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK="0")
    sys.path.insert(0, ROOT)
    from oracle import mmhand_ref as O
    from mmhand_amd.mmhand_model import MMHandModel
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    random.seed(0)
    full = O.synthetic_batch(4, 32, 32, seed=7)
    shard = {k: v[rank * 2:(rank + 1) * 2] for k, v in full.items()}
    model = MMHandModel(_opt("instance", 2, True))
    model.set_input(shard)
    model.optimize_parameters()                                  # clean iteration
    snap = {n: getattr(model, n).flat_param.clone() for n in ("netG", "netD_PP", "netD_PB")}
    if rank == 0:                                                # ONLY rank 0 overflows: its generator
        orig = model.loss_backward                               # loss blows up, as an fp16 backward would

        def poisoned(loss, loss_id=0):
            orig(loss * float("inf") if loss_id == 0 else loss, loss_id)
        model.loss_backward = poisoned
    model.optimize_parameters()
    model._settle_overflow(drain=True)
    torch.cuda.synchronize()
    unchanged = {n: bool(torch.equal(getattr(model, n).flat_param, snap[n])) for n in snap}
    torch.save({"unchanged": unchanged, "skipped": model.skipped_steps,
                "steps": [o.step_count for o in model.optimizers],
                "G": model.netG.flat_param.detach().cpu()}, os.path.join(tmp, f"ovf{rank}.pt"))
    dist.destroy_process_group()


def test_overflow_on_one_rank_skips_the_step_on_every_rank(dev, tmp_path):
    """The overflow flag needs no collective of its own: it is computed on the all-reduced gradients,
    and a non-finite value on ONE rank is non-finite in the sum on every rank (reduce_tensor of
    models/MMHandModel.py:381-384).  Rank 0 alone poisons its generator gradient; both ranks must
    skip all three optimizer steps of that iteration and stay identical replicas."""
    mp.spawn(_worker_overflow, args=(2, 29623, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(str(tmp_path), "ovf0.pt"))
    r1 = torch.load(os.path.join(str(tmp_path), "ovf1.pt"))
    for r in (r0, r1):
        assert r["unchanged"] == {"netG": True, "netD_PP": True, "netD_PB": True}, r["unchanged"]
        assert r["skipped"] == 3 and r["steps"] == [1, 1, 1], (r["skipped"], r["steps"])
    assert torch.equal(r0["G"], r1["G"])


def _worker_rccl(rank, world, port, norm, tmp):
    """ONE rank on RCCL (backend "nccl"): the data-parallel path of MMHandModel on the real communicator"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      MMH_FORCE_DP="1", MMH_DP_LOG="1", MMH_BUCKET_MB="0.02", NCCL_SOCKET_IFNAME="lo")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.path.insert(0, ROOT)
    from oracle import mmhand_ref as O
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    t = torch.ones(1, device="cuda")
    dist.all_reduce(t)
    from mmhand_amd import mmhand_model
    from mmhand_amd.mmhand_model import MMHandModel
    random.seed(0)
    batch = O.synthetic_batch(2, 32, 32, seed=7)
    model = MMHandModel(_opt(norm, 2, True))
    assert model.dp and model.comm_stream is not None
    losses = []
    for _ in range(2):
        model.set_input(batch)
        model.optimize_parameters()
        losses.append([float(v) for v in model.get_current_errors().values()])
    torch.cuda.synchronize()
    sd = {n: getattr(model, n).flat_param.detach().cpu() for n in ("netG", "netD_PB", "netD_PP")}
    torch.save({"losses": losses, "sd": sd, "ranks": int(t.item()), "backend": dist.get_backend(),
                "log": list(mmhand_model._LAST_BUCKET_LOG or [])}, os.path.join(tmp, "rccl.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("norm,port", [("instance", 29631), ("batch", 29632)])
def test_rccl_one_rank_dp_step_equals_plain_step(norm, port, dev, tmp_path):
    """RCCL under test: two optimize_parameters() through the data-parallel path - parameter
    broadcast, bucketed all-reduce from the autograd hooks on the side stream (ProcessGroupNCCL stream semantics:
    async_op + Work.wait() order the CURRENT stream behind the collective), deferred optimizer steps, SyncBN
    collectives under --norm batch - on backend "nccl" with one rank must leave exactly the parameters of the
    single-process step: a SUM over one rank is the identity and 1/world = 1."""
    mp.spawn(_worker_rccl, args=(1, port, norm, str(tmp_path)), nprocs=1, join=True)
    r = torch.load(os.path.join(str(tmp_path), "rccl.pt"))
    assert r["ranks"] == 1 and r["backend"] == "nccl"
    assert any(e[1] == "bucket" for e in r["log"])
    from oracle import mmhand_ref as O
    from mmhand_amd.mmhand_model import MMHandModel
    random.seed(0)
    batch = O.synthetic_batch(2, 32, 32, seed=7)
    model = MMHandModel(_opt(norm, 2, False))
    losses = []
    for _ in range(2):
        model.set_input(batch)
        model.optimize_parameters()
        losses.append([float(v) for v in model.get_current_errors().values()])
    torch.cuda.synchronize()
    assert np.array_equal(np.array(losses), np.array(r["losses"])), (losses, r["losses"])
    for n in ("netG", "netD_PB", "netD_PP"):
        assert torch.equal(getattr(model, n).flat_param.detach().cpu(), r["sd"][n]), n


@pytest.mark.parametrize("norm,port", [("instance", 29641), ("batch", 29642)])
def test_two_rccl_ranks_equal_one_rank_on_concatenated_batch(norm, port, dev, tmp_path):
    """The test above on the REAL transport: backend "nccl" (RCCL), one GPU per rank - bucket all-reduces through
    mmh_allreduce_bucket on torch.distributed's communicator and the side stream, SyncBN collectives on the main stream,
    deferred optimizer steps.  Needs two GPUs: skipped on the one-GPU boxes this repo is developed on, there for the day
    a node exists."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL refuses two ranks on one device)")
    from oracle import mmhand_ref as O
    from tests.golden.recipe import is_null_grad_bias
    mp.spawn(_worker, args=(2, port, norm, str(tmp_path), False, True, "nccl"), nprocs=2, join=True)
    r0 = torch.load(os.path.join(str(tmp_path), "rank0.pt"))
    r1 = torch.load(os.path.join(str(tmp_path), "rank1.pt"))
    assert r0["native"] == (True, True), r0["native"]
    for k in r0["sd"]:
        assert torch.equal(r0["sd"][k], r1["sd"][k]), k
    assert [e for e in r0["log"] if e[1] == "bucket"] == [e for e in r1["log"] if e[1] == "bucket"]
    full = O.synthetic_batch(4, 32, 32, seed=7)
    ref_losses, ref_sd = _run(norm, full, False)
    mean_losses = (np.array(r0["losses"]) + np.array(r1["losses"])) / 2
    if norm == "instance":
        assert np.allclose(mean_losses, np.array(ref_losses), rtol=2e-4), (mean_losses, ref_losses)
    for k, v in ref_sd.items():
        if v.is_floating_point() and not is_null_grad_bias("G", k, norm):
            assert torch.allclose(r0["sd"][k], v, atol=2e-4 + 1e-3 * v.abs().max().item()), k


def test_bench_two_ranks_on_one_gpu_carry_every_multi_gpu_key(dev):
    """`python bench.py --gpus 2` end to end on a one-GPU box: the launcher starts two ranks (both on GPU 0, over gloo:
    --share-gpu-gloo), and the line carries what an N-GPU launch must measure: the headline with its exposed
    communication, configs[2] (bf16), SyncBN with its packed collectives counted, configs[4]'s shape, per-rank times."""
    import json
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu-gloo", "--batch", "2",
                          "--size", "64", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--dp-512"],
                         capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-1000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["rccl_ranks"] == 2 and j["config"]["global_batch"] == 4 and j["value"] > 0
    assert j["inplace_param_grads"] is True and j["bucket_allreduce"].startswith("dist.all_reduce")
    assert "value" in j["comm_exposed_ms"] and j["ms_per_step_fastest_rank"] <= j["ms_per_step"]
    # the shipped combination under DP (SyncBN + O1), the persistent-kernel A/B against RCCL's CU use, and
    # configs[4]'s shape under SyncBN with the host's enqueue time per step on the line
    for key in ("dp_bf16_path", "dp_norm_batch", "dp_norm_batch_o1", "dp_bf16_path_nopersist", "dp_size512_bf16_b4",
                "dp_size512_bf16_b4_norm_batch"):
        r = j[key]
        assert "error" not in r, (key, r)
        assert r["n_gpus"] == 2 and r["images_per_s"] > 0 and r["losses_finite"] and "value" in r["comm_exposed_ms"], (key, r)
        assert r["host_enqueue_ms"] > 0 and r["c_abi_calls_per_step"] > 100 and r["host_enqueue_over_step"] > 0, (key, r)
    assert j["dp_norm_batch_o1"]["syncbn_collectives_per_step"]["all_gather"] == 50, j["dp_norm_batch_o1"]
    assert j["dp_norm_batch"]["syncbn_collectives_per_step"]["all_gather"] == 50, j["dp_norm_batch"]
    assert j["dp_norm_batch"]["syncbn_collectives_per_step"]["all_reduce"] == 50
    assert j["dp_bf16_path"]["syncbn_collectives_per_step"] is None


# ------------------------------------------------------------------ world 4 and 8
def _worker_n(rank, world, port, norm, tmp, poison_rank=-1):
    """rank `rank` of `world` gloo ranks that all share cuda:0, one sample per rank; poison_rank >= 0: that rank's generator
    loss overflows in the second iteration"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      MMH_DP_LOG="1", MMH_BUCKET_MB="0.02", MMH_PACK_SYNCBN="1")
    sys.path.insert(0, ROOT)
    from oracle import mmhand_ref as O
    from mmhand_amd import mmhand_model, ops
    from mmhand_amd.mmhand_model import MMHandModel
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = O.synthetic_batch(world, 32, 32, seed=7)
    shard = {k: v[rank:rank + 1] for k, v in full.items()}
    random.seed(0)
    model = MMHandModel(_opt(norm, 1, True))
    losses = []
    for it in range(2):
        if it == 1 and rank == poison_rank:
            orig = model.loss_backward

            def poisoned(loss, loss_id=0):
                orig(loss * float("inf") if loss_id == 0 else loss, loss_id)
            model.loss_backward = poisoned
        if it == 1 and poison_rank >= 0:
            snap = model.netG.flat_param.clone()
        model.set_input(shard)
        model.optimize_parameters()
        losses.append([float(v) for v in model.get_current_errors().values()])
    model._settle_overflow(drain=True)
    torch.cuda.synchronize()
    out = {"losses": losses, "sd": OrderedDict((k, v.detach().cpu()) for k, v in model.netG.state_dict().items()),
           "log": list(mmhand_model._LAST_BUCKET_LOG or []), "syncbn": dict(ops.collective_counter),
           "skipped": model.skipped_steps, "steps": [o.step_count for o in model.optimizers]}
    if poison_rank >= 0:
        out["unchanged"] = bool(torch.equal(model.netG.flat_param, snap))
    torch.save(out, os.path.join(tmp, f"w{world}_rank{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,norm,port", [(4, "instance", 29651), (4, "batch", 29652), (8, "instance", 29653),
                                             (8, "batch", 29654)])
def test_world_4_and_8_equal_one_rank_on_concatenated_batch(world, norm, port, dev, tmp_path):
    """World sizes 4 and 8 run FUNCTIONALLY before multi-GPU hardware does (options/base_options.py:171-178: the global batch
    is cut by the world size): `world` gloo ranks share the one GPU, one sample each.  Two iterations must equal the
    single-process step on the concatenated batch - gradient = mean over ranks through the bucketed all-reduce, SyncBN
    statistics over the global batch (mmh_syncbn_merge_finalize over `world` rows of (count, mean, M2)) - replicas stay
    bit-identical, and every rank issues its bucket collectives in the same order, cut at the same places."""
    from oracle import mmhand_ref as O
    from tests.golden.recipe import is_null_grad_bias
    mp.spawn(_worker_n, args=(world, port, norm, str(tmp_path)), nprocs=world, join=True)
    rs = [torch.load(os.path.join(str(tmp_path), f"w{world}_rank{r}.pt")) for r in range(world)]
    for r in rs[1:]:
        for k in rs[0]["sd"]:
            assert torch.equal(rs[0]["sd"][k], r["sd"][k]), k
        assert [e for e in r["log"] if e[1] == "bucket"] == [e for e in rs[0]["log"] if e[1] == "bucket"]
        assert r["syncbn"] == rs[0]["syncbn"]
    sb = dict(rs[0]["syncbn"])
    sb.pop("packed_sites", None)
    assert sb == ({"all_gather": 2 * (9 + 3 * 7), "all_reduce": 2 * (9 + 3 * 7)} if norm == "batch" else {}), rs[0]["syncbn"]
    g_buckets = [e[2] for e in rs[0]["log"] if e[0] == "G" and e[1] == "bucket"]
    assert g_buckets[0] == 0 and max(g_buckets) >= 2, g_buckets
    ref_losses, ref_sd = _run(norm, O.synthetic_batch(world, 32, 32, seed=7), False)
    if norm == "instance":
        mean_losses = np.mean([r["losses"] for r in rs], axis=0)
        assert np.allclose(mean_losses, np.array(ref_losses), rtol=2e-4), (mean_losses, ref_losses)
    for k, v in ref_sd.items():
        if v.is_floating_point() and not is_null_grad_bias("G", k, norm):
            assert torch.allclose(rs[0]["sd"][k], v, atol=2e-4 + 1e-3 * v.abs().max().item()), k


def test_overflow_on_rank_5_of_8_skips_the_step_everywhere(dev, tmp_path):
    """eight ranks, rank 5 alone overflows in iteration 2: all eight skip that iteration's three optimizer steps (the flag is
    computed on the all-reduced gradients: one rank's inf is everybody's) and stay identical replicas"""
    mp.spawn(_worker_n, args=(8, 29655, "instance", str(tmp_path), 5), nprocs=8, join=True)
    rs = [torch.load(os.path.join(str(tmp_path), f"w8_rank{r}.pt")) for r in range(8)]
    for r in rs:
        assert r["unchanged"] and r["skipped"] == 3 and r["steps"] == [1, 1, 1], (r["unchanged"], r["skipped"], r["steps"])
        for k in rs[0]["sd"]:
            assert torch.equal(rs[0]["sd"][k], r["sd"][k]), k


def test_bench_eight_ranks_on_one_gpu_carry_every_multi_gpu_key(dev):
    """`python bench.py --gpus 8 --share-gpu-gloo`: the line the driver's 8-GPU command prints, produced by eight ranks on the
    one GPU - eight ranks counted, global batch 16, the bf16 and SyncBN regions sane (50 + 50 packed collectives), `summary` last"""
    import json
    import subprocess
    # (eight processes time-slice the one GPU - 4 minutes for all six regions - so this line runs the two regions of
    # BASELINE.json configs[2] and of SyncBN; the two-rank test above runs every region)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--share-gpu-gloo", "--batch", "2",
                          "--size", "64", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                          "--dp-regions", "dp_bf16_path,dp_norm_batch"],
                         capture_output=True, text=True, timeout=2400)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-1000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["rccl_ranks"] == 8 and j["config"]["global_batch"] == 16 and j["value"] > 0
    assert j["inplace_param_grads"] is True and "value" in j["comm_exposed_ms"]
    for key in ("dp_bf16_path", "dp_norm_batch"):
        r = j[key]
        assert "error" not in r, (key, r)
        assert r["n_gpus"] == 8 and r["global_batch"] == 16 and r["images_per_s"] > 0 and r["losses_finite"], (key, r)
    assert "dp_norm_batch_o1" not in j
    assert j["dp_norm_batch"]["syncbn_collectives_per_step"] == {"all_gather": 50.0, "all_reduce": 50.0, "packed_sites": j["dp_norm_batch"]["syncbn_collectives_per_step"]["packed_sites"]}
    assert list(j)[-1] == "summary" and j["summary"]["dp_bf16_path"] == j["dp_bf16_path"]["images_per_s"]
