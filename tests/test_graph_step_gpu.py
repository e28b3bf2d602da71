"""--graph_step: one optimize_parameters() (models/MMHandModel.py:310-330: forward, three backward passes, three Adam steps)
captured into a hipGraph and replayed.  The replayed iteration must BE the eager one: the same model run in the replayable
form without capture (MMH_GRAPH_CAPTURE=0) gives the reference sequence, bit for bit."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import mmhand_ref as O

pytestmark = pytest.mark.gpu


def _opt(**kw):
    from mmhand_amd.options import default_train_opt
    args = dict(batchSize=2, ngf=8, ndf=8, n_layers_D=2, G_n_blocks=2, norm="instance", pool_size=3, name="graph",
                checkpoints_dir="/tmp/mmh_graph", local_rank=0, graph_step=True)
    args.update(kw)
    return default_train_opt(**args)


def _run(opt, n_iter, capture, monkeypatch, size=32, lr_drop_at=None):
    from mmhand_amd.mmhand_model import MMHandModel
    monkeypatch.setenv("MMH_GRAPH_CAPTURE", "1" if capture else "0")
    random.seed(17)
    model = MMHandModel(opt)
    losses = []
    for it in range(n_iter):
        model.set_input(O.synthetic_batch(opt.batchSize, size, size, seed=100 + it))     # a NEW batch every iteration
        model.optimize_parameters()
        losses.append([float(v) for v in model.get_current_errors().values()])
        if lr_drop_at is not None and it == lr_drop_at:
            for o in model.optimizers:
                o.param_groups[0]["lr"] *= 0.5
                o.sync_lr()
    model._settle_overflow(drain=True)
    torch.cuda.synchronize()
    state = {n: getattr(model, n).flat_param.detach().clone() for n in ("netG", "netD_PB", "netD_PP")}
    state["fake"] = model.fake_p2.detach().clone()
    return model, losses, state


@pytest.mark.parametrize("kw", [dict(), dict(opt_level="O1"), dict(norm="batch"), dict(norm="batch", opt_level="O1"),
                                dict(DG_ratio=2, opt_level="O1_FP16")],
                         ids=["fp32", "bf16", "batchnorm", "batchnorm_bf16", "fp16_dg2"])
def test_graph_step_replays_the_eager_iteration_bit_for_bit(kw, dev, monkeypatch):
    """eight iterations on changing batches, dropout ON, a pool of three images (so that swaps happen), an lr change on
    the way: captured-and-replayed == the same form run eagerly - all six losses of every iteration, every weight of the
    three networks and the generated image identical to the bit; the graph really replayed, and a replay enqueues in a
    fraction of the eager call's host time.  (--norm batch: the norm scales' and shifts' gradients are the only ones that
    still travel through autograd's AccumulateGrad nodes; a node left over from the eager iterations is bound to THEIR stream
    and forks the capture - hipStreamEndCapture then segfaults - so the capture drops the previous iteration's graph first:
    MMHandModel._capture_step, tools/probes/graph_bisect.py.)"""
    eager, l0, s0 = _run(_opt(**kw), 8, False, monkeypatch, lr_drop_at=5)
    assert eager._graph is None and eager.graph_replays == 0
    graph, l1, s1 = _run(_opt(**kw), 8, True, monkeypatch, lr_drop_at=5)
    assert graph.graph_error is None, graph.graph_error
    assert graph._graph is not None and graph.graph_replays == 8 - graph._graph_warm, (graph.graph_replays, graph._graph_warm)
    assert np.array_equal(np.array(l0), np.array(l1)), (l0, l1)
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k
    assert all(np.isfinite(l0[-1]))
    assert [o.step_count for o in graph.optimizers] == [o.step_count for o in eager.optimizers]
    assert [int(o.dev_state[0]) for o in graph.optimizers] == [o.step_count for o in graph.optimizers]
    # dropout masks differ from iteration to iteration (the salt moves) although the by-value seeds repeat
    assert l1[-1] != l1[-2]


def test_graph_step_matches_the_default_path_without_dropout(dev, monkeypatch):
    """with dropout off and an empty-decision pool (pool_size 0) nothing random is left: the replayed step equals the
    DEFAULT eager path (host Adam step count, ImagePool) to the last Adam coefficient ulp"""
    from mmhand_amd.mmhand_model import MMHandModel
    base = dict(no_dropout=True, no_dropout_D=True, pool_size=0)
    g, lg, sg = _run(_opt(**base), 6, True, monkeypatch)
    assert g.graph_replays == 6 - g._graph_warm and g.graph_error is None
    random.seed(17)
    d = MMHandModel(_opt(graph_step=False, **base))
    ld = []
    for it in range(6):
        d.set_input(O.synthetic_batch(2, 32, 32, seed=100 + it))
        d.optimize_parameters()
        ld.append([float(v) for v in d.get_current_errors().values()])
    torch.cuda.synchronize()
    assert np.allclose(np.array(lg), np.array(ld), rtol=2e-6, atol=0), (lg, ld)
    assert torch.allclose(sg["netG"], d.netG.flat_param, rtol=0, atol=2e-7)


def test_graph_step_overflow_skips_on_the_device(dev, monkeypatch):
    """a replayed iteration whose generator gradient is not finite: all three Adam launches of that replay are no-ops, the
    device step counts do not advance (apex does not count a skipped step), the loss scale halves, and the host's
    bookkeeping agrees after the flags arrive - no host decision sits inside the graph"""
    g, _, _ = _run(_opt(opt_level="O1"), 5, True, monkeypatch)
    assert g.graph_replays >= 1 and g.graph_error is None
    skipped0 = g.skipped_steps
    before = {n: getattr(g, n).flat_param.clone() for n in ("netG", "netD_PB", "netD_PP")}
    steps = [int(o.dev_state[0]) for o in g.optimizers]
    scale = g.loss_scale(0)
    g.set_input(O.synthetic_batch(2, 32, 32, seed=7))
    # poison the generator's image input in the buffers the captured stems read (the L1 target would not do: the gradient
    # of |fake - inf| is a finite sign): a non-finite image -> non-finite G gradient -> the sticky flag skips all three steps
    g._static_inputs["x_H1"][0, 0, 0, 0] = float("inf")
    if g._static_twins["x_H1"] is not None:
        g._static_twins["x_H1"][0, 0, 0, 0] = float("inf")
    g.optimize_parameters()
    g._settle_overflow(drain=True)
    torch.cuda.synchronize()
    for n, t in before.items():
        assert torch.equal(getattr(g, n).flat_param, t), n
    assert [int(o.dev_state[0]) for o in g.optimizers] == steps == [o.step_count for o in g.optimizers]
    assert g.loss_scale(0) == scale * 0.5 and g.skipped_steps == skipped0 + 3
    g.set_input(O.synthetic_batch(2, 32, 32, seed=8))
    g.optimize_parameters()
    g._settle_overflow(drain=True)
    assert [int(o.dev_state[0]) for o in g.optimizers] == [s + 1 for s in steps]
    assert all(np.isfinite(float(v)) for v in g.get_current_errors().values())


def test_device_pool_is_image_pool(dev):
    """DevicePool (mmh_pool_exchange + host-drawn indices) returns, query by query, exactly what ImagePool
    (util/image_pool.py:14-34 restated) returns on the same `random` sequence - through the fill phase, swaps, several
    swaps with one slot inside a query, and images that enter and leave within one query"""
    from mmhand_amd.mmhand_model import DevicePool, ImagePool
    for pool_size, B in ((3, 4), (5, 2), (2, 6), (50, 4)):
        a, b = ImagePool(pool_size), DevicePool(pool_size, 1)
        g = torch.Generator(device=dev).manual_seed(pool_size * 10 + B)
        for it in range(12):
            x = torch.rand((B, 4, 4, 8), generator=g, device=dev)
            random.seed(1000 + it)
            want = a.query(x)
            random.seed(1000 + it)
            b.begin_iteration(B, dev)
            got = b.query(x)
            assert torch.equal(want, got), (pool_size, B, it)
        torch.cuda.synchronize()
        stored = torch.cat(a.images, 0)
        assert b.count == len(a.images) and torch.equal(b.buf[: b.count], stored)


def test_adam_step_dev_is_adam_step(dev):
    """mmh_adam_step_dev (step count and lr on the device) against mmh_adam_step (host scalars): the same update over 30
    steps - the two bias-correction coefficients come out of double arithmetic on either side - and a skipped step
    neither moves anything nor counts"""
    from mmhand_amd import lib as L
    from mmhand_amd import ops
    n = 4096 + 12
    g = torch.Generator(device=dev).manual_seed(3)
    p0 = torch.randn(n, generator=g, device=dev)
    pa, pb = p0.clone(), p0.clone()
    ma, va, mb, vb = (torch.zeros(n, device=dev) for _ in range(4))
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    lr = torch.full((1,), 2e-4, device=dev)
    coef = torch.zeros(2, device=dev)
    skip = torch.zeros(1, dtype=torch.int32, device=dev)
    t = 0
    for it in range(30):
        gr = torch.randn(n, generator=g, device=dev)
        skipped = it in (7, 8, 20)
        skip.fill_(1 if skipped else 0)
        if not skipped:
            t += 1
            ops.adam_step(pa, gr, ma, va, 2e-4, 0.5, 0.999, 1e-8, t, 0.5)
        L.call("mmh_adam_step_dev", ops._ptr(pb), ops._ptr(gr), ops._ptr(mb), ops._ptr(vb), n, ops._ptr(lr), 0.5, 0.999, 1e-8,
               ops._ptr(step), 0.5, ops._ptr(skip), None, ops._ptr(coef), ops._stream())
        assert int(step) == t
    assert torch.equal(ma, mb) and torch.equal(va, vb)
    assert torch.allclose(pa, pb, rtol=0, atol=1e-9) and float((pa - p0).abs().max()) > 1e-3


def test_dropout_salt(dev):
    """mmh_set_dropout_salt: the keep bits of one by-value seed are a function of seed + *salt - salt 0 is the unsalted
    draw, another salt another mask with the same keep rate, and the salt advances on the stream (mmh_u64_add)"""
    from mmhand_amd import lib as L
    from mmhand_amd import ops
    n = 1 << 16

    def draw():
        bits = torch.empty(n // 8, dtype=torch.uint8, device=dev)
        L.call("mmh_dropout_bits", n, 0.5, 12345, None, ops._ptr(bits), ops._stream())
        return bits
    plain = draw()
    salt = torch.zeros(1, dtype=torch.int64, device=dev)
    L.call("mmh_set_dropout_salt", ops._ptr(salt))
    try:
        assert torch.equal(draw(), plain)
        L.call("mmh_u64_add", ops._ptr(salt), 0x9E3779B97F4A7C15, ops._stream())
        s1 = draw()
        L.call("mmh_u64_add", ops._ptr(salt), 0x9E3779B97F4A7C15, ops._stream())
        s2 = draw()
    finally:
        L.call("mmh_set_dropout_salt", None)
    assert torch.equal(draw(), plain)
    assert not torch.equal(s1, plain) and not torch.equal(s1, s2)
    ones = lambda b: float(torch.tensor([bin(int(v)).count("1") for v in b.cpu()[:4096]]).sum()) / (4096 * 8)   # noqa: E731
    assert abs(ones(s1) - 0.5) < 0.02 and abs(ones(s2) - 0.5) < 0.02
    assert int(salt) == (2 * 0x9E3779B97F4A7C15) % (1 << 64)


def test_train_main_with_graph_step_on_files(dev, monkeypatch):
    """the training driver end to end with --graph_step on a prepared directory (data.HandFolderLoader): raw uint8 batches are
    decoded on the device and copied INTO the captured buffers, the lr schedule reaches the device copy at the epoch boundary,
    checkpoints and the loss log are written as without the graph"""
    import shutil
    import tempfile
    from tests._dataset_fixture import write_rhd
    from mmhand_amd import train
    d = tempfile.mkdtemp(prefix="mmh_ds_")
    try:
        root = os.path.join(d, "rhd")
        write_rhd(root, n=8, size=32)
        monkeypatch.chdir(d)
        monkeypatch.setenv("MMH_GRAPH_CAPTURE", "1")
        seen = {}
        from mmhand_amd import mmhand_model as MM
        real_init = MM.MMHandModel.__init__

        def spy_init(self, opt):
            real_init(self, opt)
            seen["model"] = self
        monkeypatch.setattr(MM.MMHandModel, "__init__", spy_init)
        train.main(["--name", "g", "--dataroot", root, "--dataset", "rhd", "--augmentation_ratio", "1.0", "--batchSize", "2",
                    "--ngf", "8", "--ndf", "8", "--G_n_blocks", "2", "--n_layers_D", "2", "--norm", "instance", "--fineSize", "32",
                    "--niter", "1", "--niter_decay", "1", "--print_freq", "2", "--vgg_random_init", "--checkpoints_dir", "ck",
                    "--pool_size", "3", "--graph_step", "--opt_level", "O1"])
        m = seen["model"]
        assert m.graph_error is None and m._graph is not None and m.graph_replays == 8 - m._graph_warm
        assert os.path.isfile(os.path.join("ck", "g", "latest_net_netG.pth"))
        log = open(os.path.join("ck", "g", "loss_log.txt")).read().strip().splitlines()
        assert len(log) == 8 and all("pair_L1loss" in l for l in log)
        # the schedule moved twice (network_utils.py:92-95 with niter 1, niter_decay 1: epoch 2 runs at lr 0, and the rule goes
        # negative behind it, as the reference's does): the replayed Adam launches read whatever it says from the device
        lr = m.optimizer_G.param_groups[0]["lr"]
        assert abs(float(m.optimizer_G.dev_state[1]) - lr) <= 1e-6 * max(abs(lr), 1e-12) and lr < m.opt.lr
        m._settle_overflow(drain=True)
        assert m.optimizer_G.step_count == 8 and int(m.optimizer_G.dev_state[0]) == 8 and m.skipped_steps == 0
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_graph_step_survives_a_short_batch(dev, monkeypatch):
    """the last batch of an epoch may be short (the reference's DataLoader does not drop it): in replay mode such an iteration
    runs in the eager form on buffers of its own, and the next full batch is a replay again - the whole sequence equal, bit
    for bit, to the same sequence without any capture"""
    from mmhand_amd.mmhand_model import MMHandModel

    def seq(capture):
        monkeypatch.setenv("MMH_GRAPH_CAPTURE", "1" if capture else "0")
        random.seed(3)
        m = MMHandModel(_opt(opt_level="O1"))
        out = []
        for it, B in enumerate((2, 2, 2, 2, 2, 1, 2, 1, 2)):
            m.set_input(O.synthetic_batch(B, 32, 32, seed=200 + it))
            m.optimize_parameters()
            out.append([float(v) for v in m.get_current_errors().values()])
        m._settle_overflow(drain=True)
        torch.cuda.synchronize()
        return m, out, m.netG.flat_param.detach().clone(), m.fake_p2.detach().clone()
    e, le, we, fe = seq(False)
    g, lg, wg, fg = seq(True)
    assert g.graph_error is None and g.graph_replays == 4          # iterations 4, 5, 7, 9; 6 and 8 are the short ones
    assert np.array_equal(np.array(le), np.array(lg)), (le, lg)
    assert torch.equal(we, wg) and torch.equal(fe, fg) and tuple(fg.shape) == (2, 3, 32, 32)
