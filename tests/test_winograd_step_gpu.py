"""Step- and gradient-level parity of the path bench.py measures: every 3x3 stride-1 conv of the
Generator stack and of the Discriminators' residual blocks on Winograd F(6x6,3x3) (ragged tiles,
fused backward transforms, InstanceNorm statistics from the output transform).

The reference-generated fixtures (tests/golden, ngf 8 at 32x32) are too small to reach F(6x6,3x3)
(`_wino_tile` needs 12x12 feature maps and Cin*Cout >= 128^2), so these tests use ngf = ndf = 32
(PATBlock channels 128 / 256, Discriminator trunk 128) at 64x64 inputs (16x16 feature maps = 3x3
RAGGED tiles) and compare with the CPU oracle (oracle/mmhand_ref.py, pinned against the reference
modules by tests/golden/make_golden.py) in fp64 and fp32.  That the Winograd path really ran is
asserted from the C-ABI calls themselves."""
import random
import statistics
from collections import Counter, OrderedDict

import numpy as np
import pytest
import torch

from oracle import mmhand_ref as O
from oracle import ops_ref as R
from tests.golden import recipe as RC
from tests.test_model_gpu import logical_grads

pytestmark = pytest.mark.gpu
NGF, SIZE, NB, NLD = 32, 64, 2, 2


class _Spy:
    """Counts C-ABI entry points by (name, winograd tile) while installed on mmhand_amd.lib.call."""

    def __init__(self, monkeypatch):
        from mmhand_amd import lib, ops
        self.calls = Counter()
        real = lib.call
        tile_arg = {"mmh_wino_input": 6, "mmh_wino_output": 8, "mmh_wino_input_dy": 5, "mmh_wino_dy": 5,
                    "mmh_wino_dw": 3, "mmh_wino_weights": 4}

        def call(name, *args):
            t = tile_arg.get(name)
            self.calls[(name, args[t] if t is not None else None)] += 1
            if name in ("mmh_wino_gemm", "mmh_wino_gemm_levels"):      # forward (two-level sum) / dgrad (one level) GEMMs
                self.calls[("gemm_planes", args[6])] += 1
            return real(name, *args)

        assert ops.L is lib
        monkeypatch.setattr(lib, "call", call)

    def n(self, name, tile=None):
        return self.calls[(name, tile)]


def _assert_wino6_shapes():
    from mmhand_amd import ops
    hs = SIZE // 4
    for cin, cout in ((4 * NGF, 4 * NGF), (8 * NGF, 8 * NGF), (8 * NGF, 4 * NGF)):      # G stack, D trunk
        for op in ("fprop", "dgrad", "wgrad"):
            assert ops._wino_tile(2, hs, hs, cin, cout, 3, 1, 1, False, op) == 6, (cin, cout, op)


def _opt(norm, **kw):
    from mmhand_amd.options import default_train_opt
    args = dict(batchSize=2, ngf=NGF, ndf=NGF, n_layers_D=NLD, G_n_blocks=NB, norm=norm, no_dropout=True,
                no_dropout_D=True, pool_size=2, name="wino6", checkpoints_dir="/tmp/mmh_pytest_ckpt",
                local_rank=0, fineSize=SIZE)
    args.update(kw)
    return default_train_opt(**args)


@pytest.mark.parametrize("norm", ["instance", "batch"])
def test_optimize_parameters_winograd_vs_oracle(norm, dev, monkeypatch):
    """3 free-running iterations of optimize_parameters() with F(6x6,3x3) engaged everywhere it
    engages in the benchmark, against the fp64 oracle.

      * the six losses of every iteration: 1e-3 (north_star bar);
      * the generated image of iteration 1 (identical weights on both sides): 2e-5;
      * every post-step weight within 6.3 lr (+ 1e-3 max|w|) of the oracle's - three Adam sign
        steps in opposite directions - and >= 75 % of the elements of every weight tensor within
        lr/2 of it;
      * the generated image of iteration 3 (after two Adam steps): 2e-2, and no further from the fp64
        oracle than 3x the larger of (a) the same step on the DIRECT kernels and (b) the oracle's
        own fp32 run.

    Why the last bound is relative.  Adam's first steps are sign steps: every element moves by
    +-lr = 2e-4 (1 % of the N(0, 0.02) initial weights) whatever its gradient's size, so an element
    whose gradient is within rounding of zero flips with ANY change of summation order.  Measured
    (tools/step_noise.py, this configuration): after two steps the oracle's own fp32 run is 5e-3
    from its fp64 run on the image (instance norm; 4e-4 batch norm), the direct kernels 8e-4 / 3e-3,
    F(6x6,3x3) 8e-3 / 3e-3 - while the losses of all of them stay within 1e-4."""
    from mmhand_amd import ops
    from mmhand_amd.mmhand_model import MMHandModel
    assert ops.USE_WINOGRAD and ops.WINOGRAD_TILE == 6
    _assert_wino6_shapes()
    spy = _Spy(monkeypatch)
    lr = 2e-4
    model = MMHandModel(_opt(norm))
    assert model.opt.lr == lr
    monkeypatch.setattr(ops, "USE_WINOGRAD", False)
    direct = MMHandModel(_opt(norm))
    sds = [OrderedDict((k, v.cpu()) for k, v in n.state_dict().items())
           for n in (model.netG, model.netD_PB, model.netD_PP)]
    for net, sd in zip((direct.netG, direct.netD_PB, direct.netD_PP), sds):
        net.load_state_dict(sd)
    vgg = OrderedDict((k, v.cpu()) for k, v in model.vgg.state_dict().items())
    f64 = lambda sd: OrderedDict((k, v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items())
    o64 = O.StepOracle(f64(sds[0]), f64(sds[1]), f64(sds[2]), f64(vgg), norm, False, False, NB, NLD, pool_size=2,
                       rng=random.Random(49))
    o32 = O.StepOracle(sds[0], sds[1], sds[2], vgg, norm, False, False, NB, NLD, pool_size=2, rng=random.Random(49))
    rng_w, rng_d = random.Random(49), random.Random(49)      # ImagePool draws: one stream per model
    drift = {}
    for it in range(3):
        batch = O.synthetic_batch(2, SIZE, SIZE, seed=200 + it)
        want = list(o64.step({k: v.double() for k, v in batch.items()}).values())
        o32.step(batch)
        for tag, m, rng, wino in (("wino", model, rng_w, True), ("direct", direct, rng_d, False)):
            ops.USE_WINOGRAD = wino
            ops.bump_weights_epoch()
            random.setstate(rng.getstate())
            m.set_input(batch)
            m.optimize_parameters()
            rng.setstate(random.getstate())
            got = [float(v) for v in m.get_current_errors().values()]
            assert np.allclose(got, want, rtol=1e-3), (tag, it, got, want)
            drift[tag] = R.rel_l1(m.fake_p2, o64.fake_p2.detach())
        drift["oracle32"] = R.rel_l1(o32.fake_p2.detach(), o64.fake_p2.detach())
        if it == 0:
            assert drift["wino"] < 2e-5 and drift["direct"] < 2e-5, drift
    print(f"\n[{norm}] image drift from the fp64 oracle after two Adam steps: {drift}")
    assert drift["wino"] < 2e-2 and drift["wino"] <= 3 * max(drift["direct"], drift["oracle32"], 1e-3 / 3), drift
    # the path under test really ran: per iteration G has 6*NB convs on F(6x6,3x3) (+ 2*NLD per
    # Discriminator pass), forward AND fused backward
    # ... with the norm between the two convs of every block applied inside their transforms (2 blocks per
    # PATBlock forward, their backward inside the producing conv's fused backward transform)
    if ops.USE_NORM_FUSION:
        assert spy.n("mmh_wino_input_normact") >= 3 * 2 * NB and spy.n("mmh_wino_input_dy_normbwd") >= 3 * 2 * NB, spy.calls
    assert spy.n("mmh_wino_input_dy", 6) + spy.n("mmh_wino_input_dy_normbwd") >= 3 * 6 * NB, spy.calls
    assert spy.calls[("gemm_planes", 64)] >= 3 * 3 * 6 * NB, spy.calls
    # F(4x4,3x3) only where the benchmark uses it too (the 64->64 VGG conv); never F(2x2,3x3)
    assert spy.calls[("gemm_planes", 36)] <= 3 * 3 and spy.calls[("gemm_planes", 16)] == 0, spy.calls
    unstable = 0.0
    for tag, net, onet in (("G", model.netG, o64.G), ("DPB", model.netD_PB, o64.DPB), ("DPP", model.netD_PP, o64.DPP)):
        osd = onet.state_dict()
        for k, v in net.state_dict().items():
            if v.is_floating_point() and not RC.is_null_grad_bias(tag, k, norm) and "running" not in k:
                ref = osd[k].float().numpy()
                got = v.cpu().numpy()
                # three +-lr sign steps: an element whose gradient sign is within rounding of zero on
                # all three differs by 6 lr at most
                assert np.allclose(got, ref, atol=6.3 * lr + 1e-3 * np.abs(ref).max()), (tag, k)
                frac = float((np.abs(got - ref) > 0.5 * lr).mean())
                unstable = max(unstable, frac)
                assert frac < 0.25, (tag, k, frac)
    print(f"[{norm}] largest fraction of a weight tensor further than lr/2 from the oracle: {unstable:.3f}")


def _generator_grads(norm, wino, sd, g_in, probe, dev):
    from mmhand_amd import ops
    from mmhand_amd.networks import Generator
    ops.USE_WINOGRAD = wino
    ops.bump_weights_epoch()
    net = Generator([3, 42, 6], 3, NGF, norm, False, NB)
    net.load_state_dict(sd)
    net.to(dev).train()
    net.flatten_parameters()
    out = net([t.to(dev) for t in g_in])
    (out * probe.to(dev)).sum().backward()
    return out.detach().cpu(), logical_grads(net)


@pytest.mark.parametrize("norm", ["instance", "batch"])
def test_generator_gradients_winograd_bounded(norm, dev, monkeypatch):
    """Parameter gradients of the wide-channel Generator (B=4, 64x64) on both conv paths against the
    fp64 oracle, with a BOUNDED rule (no escape through a conditioning estimate):
      direct kernels     every tensor <= max(1.5e-3, 3 * cond) and <= 5e-3
                         (the 7x7 stem of the depth stream sits at 0.9-1.1e-3 with cond 3.4e-4: its InstanceNorm
                         sees near-constant planes, and which fp32-exact way the statistics are summed - a pass over
                         y or the conv epilogue's per-tile partials, both 1e-7 from float64 on invstd,
                         tools/stats_accuracy.py - moves it by 10 %)
      Winograd F(6x6)    every tensor <= 5e-3, and the median over tensors <= 2e-3
                         (measured: median 1.4e-3 / max 3.4e-3 with instance norm, 3e-4 / 3.5e-3
                         with batch norm)
    cond = distance between the oracle's own fp32 and fp64 gradients of that tensor, i.e. what the
    rounding of a plain fp32 PyTorch implementation (the reference on its CPU path) does to it on
    this problem; it is printed beside the two paths' errors.  Measured (tools/grad_trace.py): the
    Winograd forward activations are within 2-4e-6 of the direct kernels'; the backward pass
    inflates that - a ReLU mask that flips is an O(1) change of one element, which does not enjoy
    the cancellation the dense gradient signal sees in each following layer - to 3e-4 ... 3.5e-3 per
    tensor (largest on the stream-3 parameters: replicated depth planes), 2-5x what PyTorch's own
    fp32 CPU run shows on the same tensors (cond 1e-4 ... 1e-3, varying with the host's oneDNN
    code path); the direct kernels stay at 1e-6."""
    from mmhand_amd import ops
    monkeypatch.setattr(ops, "USE_WINOGRAD", True)
    _assert_wino6_shapes()
    B = 4
    from mmhand_amd.networks import Generator
    sd = Generator([3, 42, 6], 3, NGF, norm, False, NB).init_weights("normal", 49).state_dict()
    b = O.synthetic_batch(B, SIZE, SIZE, seed=11)
    g_in = [b["H1"], torch.cat((b["P1"], b["P2"]), 1), torch.cat((b["D1"], b["D2"]), 1)]
    probe = torch.randn(B, 3, SIZE, SIZE, generator=torch.Generator().manual_seed(3))
    og = {}
    for dt in (torch.float64, torch.float32):
        onet = O._Net({k: (v.to(dt) if v.is_floating_point() else v) for k, v in sd.items()}, norm, False)
        ref = O.generator_forward(onet, [t.to(dt) for t in g_in], NB)
        (ref * probe.to(dt)).sum().backward()
        og[dt] = dict((k, t.grad) for k, t in onet.named_parameters())
        if dt == torch.float64:
            ref64 = ref.detach()
    out_d, gd = _generator_grads(norm, False, sd, g_in, probe, dev)
    out_w, gw = _generator_grads(norm, True, sd, g_in, probe, dev)
    assert R.rel_l1(out_d, ref64) < 2e-5 and R.rel_l1(out_w, ref64) < 2e-5
    rows = []
    for k in gd:
        if RC.is_null_grad_bias("G", k, norm) or og[torch.float64].get(k) is None:
            continue
        ref = og[torch.float64][k]
        cond = R.rel_l1(og[torch.float32][k], ref)
        ed, ew = R.rel_l1(gd[k], ref), R.rel_l1(gw[k], ref)
        rows.append((k, cond, ed, ew))
    report = "\n".join(f"{k:55s} cond {c:.1e} direct {d:.1e} wino {w:.1e}" for k, c, d, w in rows)
    for k, cond, ed, ew in rows:
        assert ed <= min(5e-3, max(1.5e-3, 3 * cond)), (k, cond, ed, "\n" + report)
        assert ew <= 5e-3, (k, cond, ed, ew, "\n" + report)
    med = statistics.median(ew for _, _, _, ew in rows)
    print("\n" + report + f"\nWinograd median {med:.2e}, max {max(r[3] for r in rows):.2e}")
    assert med <= 2e-3, (med, "\n" + report)


def test_gradient_noise_full_size_generator(dev, monkeypatch):
    """What tools/grad_noise.py measures, as a gate: on the FULL-size Generator (ngf 64, 9 PATBlocks,
    256x256, B=2, --norm instance) the F(6x6,3x3) path and the direct kernels agree on the output to
    5e-5 and on the parameter gradients to a median 1e-2 / 90th percentile 3e-2 relative L1 per
    tensor - the level at which two fp32 implementations of this network differ (ReLU masks of
    pre-activations within rounding of zero flip; the norm layers' backward sums nearly cancel)."""
    from bench import synthetic_batch_gpu
    from mmhand_amd import ops
    from mmhand_amd.networks import Generator
    B = 2
    b = synthetic_batch_gpu(B, 256, 256, 49, dev)
    g_in = [b["H1"], torch.cat((b["P1"], b["P2"]), 1), torch.cat((b["D1"], b["D2"]), 1)]
    probe = torch.randn(B, 3, 256, 256, generator=torch.Generator().manual_seed(3)).to(dev)
    res = {}
    for wino in (False, True):
        monkeypatch.setattr(ops, "USE_WINOGRAD", wino)
        ops.bump_weights_epoch()
        net = Generator([3, 42, 6], 3, 64, "instance", False, 9).init_weights("normal", 49).to(dev).train()
        net.flatten_parameters()
        out = net(g_in)
        (out * probe).sum().backward()
        res[wino] = (out.detach().clone(), {n: p.grad.detach().clone() for n, p in net.named_parameters()})
        del net, out
    rel = lambda a, b_: float((a.double() - b_.double()).abs().sum() / b_.double().abs().sum().clamp_min(1e-30))
    assert rel(res[True][0], res[False][0]) < 5e-5
    errs = sorted(rel(res[True][1][n], g) for n, g in res[False][1].items()
                  if float(g.abs().sum()) > 0 and not RC.is_null_grad_bias("G", n, "instance"))
    med, p90 = statistics.median(errs), errs[(len(errs) * 9) // 10]
    print(f"\nfull-size Winograd-vs-direct gradient noise: median {med:.2e}, p90 {p90:.2e}, max {errs[-1]:.2e}")
    assert med < 1e-2 and p90 < 3e-2, (med, p90, errs[-1])


def test_winograd_backward_kernels_add_no_gradient_noise(dev, monkeypatch):
    """VERDICT r2 #4, split by pass (tools/probes/wino_grad_split.py as a gate): on the full-size Generator the F(6x6,3x3)
    dgrad and wgrad kernels behind the DIRECT fprop - identical activations, so no ReLU mask can flip - reproduce the
    all-direct parameter gradients to 5e-5 per tensor (measured: median 1.4e-5, max 2.7e-5 with the dgrad GEMMs on one summation
    level as the package runs them, 7e-6 / 1e-5 with two).  The 3e-3 of the full Winograd
    path is therefore the forward's 7e-6 exciting the network's conditioning; the same run shows the direct path against
    itself with its input scaled by (1 + 2^-22) moving those gradients by more than 5e-4 (measured median 2.2e-3)."""
    from bench import synthetic_batch_gpu
    from mmhand_amd import ops
    from mmhand_amd.networks import Generator
    B = 2
    b = synthetic_batch_gpu(B, 256, 256, 49, dev)
    g_in = [b["H1"], torch.cat((b["P1"], b["P2"]), 1), torch.cat((b["D1"], b["D2"]), 1)]
    probe = torch.randn(B, 3, 256, 256, generator=torch.Generator().manual_seed(3)).to(dev)
    real, allowed = ops._wino_tile, set()
    monkeypatch.setattr(ops, "USE_WINOGRAD", True)
    monkeypatch.setattr(ops, "_wino_tile", lambda *a_, **k_: real(*a_, **k_) if (k_.get("op") or (a_[9] if len(a_) > 9 else "fprop")) in allowed else 0)
    spy = _Spy(monkeypatch)

    def run(which, scale):
        allowed.clear()
        allowed.update(which)
        ops.bump_weights_epoch()
        net = Generator([3, 42, 6], 3, 64, "instance", False, 9).init_weights("normal", 49).to(dev).train()
        net.flatten_parameters()
        out = net([t * scale for t in g_in])
        (out * probe).sum().backward()
        return out.detach().clone(), {n: p.grad.detach().clone() for n, p in net.named_parameters()}

    rel = lambda a, b_: float((a.double() - b_.double()).abs().sum() / b_.double().abs().sum().clamp_min(1e-30))
    ref = run((), 1.0)
    bwd = run(("dgrad", "wgrad"), 1.0)
    assert torch.equal(bwd[0], ref[0])                      # same forward kernels, same image
    errs = sorted(rel(bwd[1][n], g) for n, g in ref[1].items() if float(g.abs().sum()) > 0)
    print(f"\nWinograd dgrad + wgrad behind the direct fprop: median {statistics.median(errs):.2e}, max {errs[-1]:.2e}")
    assert errs[-1] < 5e-5, errs[-1]
    assert spy.n("mmh_wino_wgrad_gemm") >= 2 * 9 and spy.n("mmh_wino_gemm_levels") >= 2 * 9, spy.calls      # they did run
    assert spy.n("mmh_wino_gemm") == 0, spy.calls                                                        # no forward GEMM did
    assert spy.n("mmh_wino_input_normact") == 0, spy.calls                                                # behind direct fprops
    ulp = run((), 1.0 + 2.0 ** -22)
    errs_u = sorted(rel(ulp[1][n], g) for n, g in ref[1].items() if float(g.abs().sum()) > 0)
    print(f"direct path, input * (1 + 2^-22): output {rel(ulp[0], ref[0]):.2e}, gradients median {statistics.median(errs_u):.2e}, max {errs_u[-1]:.2e}")
    assert statistics.median(errs_u) > 5e-4
    ops.bump_weights_epoch()


def test_batched_filter_transforms_change_nothing(dev, monkeypatch):
    """ops._WinoBatch: after an optimizer step a network's F(6x6,3x3) filter transforms run as ONE launch
    (mmh_wino_weights_multi) at the first request instead of one by one.  Four free-running iterations with and without:
    the six losses of every iteration and the final weights bit-identical; with batching, from the third iteration on
    (the second one learns the set) the single-filter entry point is no longer called for the three trained networks."""
    from mmhand_amd import lib, ops
    from mmhand_amd.mmhand_model import MMHandModel
    outs = {}
    for on in (True, False):
        monkeypatch.setattr(ops, "USE_WINO_BATCH", on)
        ops.bump_weights_epoch()
        calls = []
        real = lib.call
        monkeypatch.setattr(lib, "call", lambda n, *a, _c=calls, _r=real: (_c.append(n), _r(n, *a))[1])
        torch.manual_seed(7)
        random.seed(7)
        model = MMHandModel(_opt("instance"))
        losses, per_it = [], []
        for it in range(4):
            mark = len(calls)
            model.set_input(O.synthetic_batch(2, SIZE, SIZE, seed=500 + it))
            model.optimize_parameters()
            losses.append([float(v) for v in model.get_current_errors().values()])
            per_it.append(Counter(calls[mark:]))
        outs[on] = (losses, [p.detach().clone() for p in model.netG.parameters()], per_it)
        monkeypatch.setattr(lib, "call", real)
    assert outs[True][0] == outs[False][0], (outs[True][0], outs[False][0])
    assert all(torch.equal(a, b) for a, b in zip(outs[True][1], outs[False][1]))
    on_it = outs[True][2]
    assert on_it[3]["mmh_wino_weights_multi"] == 3 and on_it[3]["mmh_wino_weights"] == 0, on_it[3]
    assert outs[False][2][3]["mmh_wino_weights_multi"] == 0 and outs[False][2][3]["mmh_wino_weights"] > 10
    ops.bump_weights_epoch()
