"""GPU parity of the HBM-bound kernels (norm+relu+dropout fwd/bwd, gate, losses, Adam, pack,
reflect fold, pose maps) against the CPU oracle, through the C-ABI."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import mmhand_ref as O
from oracle import ops_ref as R

pytestmark = pytest.mark.gpu
TOL = 2e-5
G = os.path.join(os.path.dirname(__file__), "golden")


def _mk(shape, seed, dev, scale=1.0, shift=0.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale + shift).to(dev)


@pytest.mark.parametrize("mode", ["batch", "instance"])
@pytest.mark.parametrize("shape", [(2, 8, 8, 16), (3, 16, 16, 64), (2, 64, 64, 256), (2, 5, 7, 512),
                                   (2, 6, 6, 24)])
@pytest.mark.parametrize("relu,drop", [(False, False), (True, False), (True, True)])
def test_norm_act_fwd_bwd(mode, shape, relu, drop, dev):
    from mmhand_amd import ops
    B, H, W, C = shape
    x = _mk(shape, 1, dev, 2.0, 3.0).requires_grad_(True)      # non-zero mean: exercises the shifted sums
    gamma = _mk((C,), 2, dev, 0.1, 1.0).requires_grad_(True) if mode == "batch" else None
    beta = _mk((C,), 3, dev, 0.1).requires_grad_(True) if mode == "batch" else None
    rm = torch.zeros(C, device=dev) if mode == "batch" else None
    rv = torch.ones(C, device=dev) if mode == "batch" else None
    mask = (torch.rand(shape, generator=torch.Generator().manual_seed(5)) >= 0.5).to(torch.uint8).to(dev) if drop else None
    out = ops.NormActFn.apply(x, gamma, beta, None, rm, rv, mode, relu, 0.5 if drop else 0.0, 0, mask, None)
    dy = _mk(shape, 4, dev)
    out.backward(dy)
    xc = x.detach().cpu().double().requires_grad_(True)
    gc = gamma.detach().cpu().double().requires_grad_(True) if gamma is not None else None
    bc = beta.detach().cpu().double().requires_grad_(True) if beta is not None else None
    ref = R.norm_act(xc, gc, bc, mode, relu, None if mask is None else mask.cpu(), 0.5)
    ref.backward(dy.cpu().double())
    assert R.rel_l1(out, ref) < TOL
    assert R.rel_l1(x.grad, xc.grad) < 5e-5
    if mode == "batch":
        # with ReLU a handful of elements sit within fp32 rounding of 0 and flip their mask
        # relative to the fp64 oracle; each flip moves a channel sum by O(|dy|)
        gtol = 5e-4 if relu else 5e-5
        assert R.rel_l1(gamma.grad, gc.grad) < gtol
        assert R.rel_l1(beta.grad, bc.grad) < gtol
        xn = R.to_nchw(x.detach().cpu())
        assert torch.allclose(rm.cpu(), 0.1 * xn.mean((0, 2, 3)), atol=1e-5)
        assert torch.allclose(rv.cpu(), 0.9 + 0.1 * xn.transpose(0, 1).reshape(C, -1).var(1, unbiased=True), rtol=1e-4)


def test_norm_residual(dev):
    from mmhand_amd import ops
    shape = (2, 8, 8, 32)
    x = _mk(shape, 1, dev).requires_grad_(True)
    res = _mk(shape, 2, dev).requires_grad_(True)
    out = ops.NormActFn.apply(x, None, None, res, None, None, "instance", False, 0.0, 0, None, None)
    dy = _mk(shape, 3, dev)
    out.backward(dy)
    ref = R.norm_act(x.detach().cpu(), None, None, "instance", False, residual=res.detach().cpu())
    assert R.rel_l1(out, ref) < TOL
    assert R.rel_l1(res.grad, dy) == 0.0


def test_dropout_rng_statistics(dev):
    """On-device counter-hash dropout: keep rate ~ 0.5, kept values scaled by 2, masks differ per seed."""
    from mmhand_amd import ops
    x = torch.ones((4, 32, 32, 64), device=dev)
    one = torch.ones((1, 64), device=dev); zero = torch.zeros((1, 64), device=dev)
    a = ops.raw_scale_shift_act(x, one, zero, None, True, 0.5, 123, None)
    b = ops.raw_scale_shift_act(x, one, zero, None, True, 0.5, 124, None)
    a2 = ops.raw_scale_shift_act(x, one, zero, None, True, 0.5, 123, None)
    assert torch.equal(a, a2)
    keep = (a > 0).float().mean().item()
    assert abs(keep - 0.5) < 0.01
    assert set(a.unique().tolist()) == {0.0, 2.0}
    assert ((a > 0) != (b > 0)).float().mean().item() > 0.4


def test_affine_act(dev):
    from mmhand_amd import ops
    x = _mk((2, 8, 8, 4), 1, dev).requires_grad_(True)
    sc = torch.tensor([2.0, 3.0, -1.0, 0.0], device=dev); sh = torch.tensor([0.1, -0.2, 0.3, 0.0], device=dev)
    y = ops.AffineActFn.apply(x, sc, sh, True)
    dy = _mk((2, 8, 8, 4), 2, dev)
    y.backward(dy)
    xr = x.detach().cpu().requires_grad_(True)
    yr = torch.relu(xr * sc.cpu() + sh.cpu())
    yr.backward(dy.cpu())
    assert R.rel_l1(y, yr) < TOL and R.rel_l1(x.grad, xr.grad) < TOL


@pytest.mark.parametrize("want_cat", [True, False])
def test_gate_fwd_bwd(want_cat, dev):
    from mmhand_amd import ops
    shape = (2, 8, 8, 32)
    ts = [_mk(shape, i, dev).requires_grad_(True) for i in range(4)]
    out, x2n, x3n = ops.GateFn.apply(*ts, want_cat)
    cs = [t.detach().cpu().double().requires_grad_(True) for t in ts]
    ro, r2, r3 = R.gate(*cs)
    assert R.rel_l1(out, ro) < TOL
    g0 = _mk(shape, 10, dev)
    if want_cat:
        assert R.rel_l1(x2n, r2) < TOL and R.rel_l1(x3n, r3) < TOL
        g2 = _mk((2, 8, 8, 64), 11, dev); g3 = _mk((2, 8, 8, 64), 12, dev)
        torch.autograd.backward([out, x2n, x3n], [g0, g2, g3])
        torch.autograd.backward([ro, r2, r3], [g0.cpu().double(), g2.cpu().double(), g3.cpu().double()])
    else:
        out.backward(g0)
        ro.backward(g0.cpu().double())
    for t, c in zip(ts, cs):
        assert R.rel_l1(t.grad, c.grad) < TOL


def test_losses_vs_reference_fixture(dev):
    """GANLoss and L1 against values produced by the reference's own GANLoss / F.l1_loss."""
    from mmhand_amd import ops
    fix = dict(np.load(os.path.join(G, "losses.npz")))
    logits = torch.from_numpy(fix["logits"]).permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
    real = ops.BCEWithLogitsConstFn.apply(logits, 1.0, 1.0)
    fake = ops.BCEWithLogitsConstFn.apply(logits, 0.0, 1.0)
    assert abs(real.item() - float(fix["gan_real"])) < 1e-5
    assert abs(fake.item() - float(fix["gan_fake"])) < 1e-5
    (real * 3.0).backward()
    lc = torch.from_numpy(fix["logits"]).requires_grad_(True)
    (O.gan_loss(lc, True) * 3.0).backward()
    assert R.rel_l1(logits.grad.permute(0, 3, 1, 2), lc.grad) < TOL
    a = _mk((2, 16, 16, 4), 1, dev).requires_grad_(True); b = _mk((2, 16, 16, 4), 2, dev)
    l = ops.L1MeanFn.apply(a, b, 10.0, float(a.numel()))
    l.backward()
    ar = a.detach().cpu().requires_grad_(True)
    lr = 10.0 * F.l1_loss(ar, b.cpu()); lr.backward()
    assert abs(l.item() - lr.item()) < 1e-4 and R.rel_l1(a.grad, ar.grad) < TOL


def test_perceptual_criterion_l1_and_mse_vs_reference_fixture(dev):
    """L1PlusPerceptualLoss on the HIP kernels (VGG head convs, ImageNet pre-normalisation, L1 and
    MSE reductions) against the reference criterion's own outputs: --percep_is_l1 1 (losses.npz) and
    0 (losses_mse.npz, F.mse_loss branch of losses/L1_plus_perceptualLoss.py:68-71)."""
    from mmhand_amd import ops
    from mmhand_amd.mmhand_model import L1PlusPerceptualLoss
    from mmhand_amd.networks import VGGHead
    from tests.golden import recipe as RC
    fix = dict(np.load(os.path.join(G, "losses.npz")))
    vgg = VGGHead()
    vgg.load_state_dict(RC.vgg_recipe())
    vgg.to(dev)
    real = ops.raw_pack([(torch.from_numpy(fix["real"]).to(dev), True, 3)], 2, 32, 32, 4, dev)
    for is_l1, f in ((1, fix), (0, dict(np.load(os.path.join(G, "losses_mse.npz"))))):
        fake_nchw = torch.from_numpy(fix["fake"]).to(dev).requires_grad_(True)
        fake = ops.PackFn.apply(4, fake_nchw, True, 3)
        tot, l1, lp = L1PlusPerceptualLoss(10.0, 10.0, vgg, is_l1)(fake, real)
        assert abs(tot.item() - float(f["total"])) < 1e-3 * abs(float(f["total"])), (is_l1, tot.item())
        assert abs(l1.item() - float(f["l1"])) < 1e-3 * abs(float(f["l1"]))
        assert abs(lp.item() - float(f["perceptual"])) < 1e-3 * abs(float(f["perceptual"]))
        tot.backward()
        assert R.rel_l1(fake_nchw.grad, torch.from_numpy(f["dfake"])) < 1e-3, is_l1
    a = _mk((2, 16, 16, 4), 1, dev).requires_grad_(True); b = _mk((2, 16, 16, 4), 2, dev)
    l = ops.MSEMeanFn.apply(a, b, 10.0, float(a.numel()))
    l.backward()
    ar = a.detach().cpu().double().requires_grad_(True)
    lr = 10.0 * F.mse_loss(ar, b.cpu().double()); lr.backward()
    assert abs(l.item() - lr.item()) < 1e-5 * abs(lr.item()) and R.rel_l1(a.grad, ar.grad) < TOL


@pytest.mark.parametrize("pl", [0, 2, 4, 8, 13])
def test_perceptual_layers_other_than_3_vs_oracle(pl, dev):
    """--perceptual_layers slices vgg19.features at ANY index (losses/L1_plus_perceptualLoss.py:22-27: the loop adds
    layers 0 .. perceptual_layers): a conv without its ReLU (0, 2), through MaxPool2d (4, 13), deeper convs (8, 13).
    Features, the three losses and the gradient w.r.t. the generated image against the fp64 oracle (whose layer table
    is torchvision's published "E" configuration - torchvision itself is absent here)."""
    from mmhand_amd import ops
    from mmhand_amd.mmhand_model import L1PlusPerceptualLoss
    from mmhand_amd.networks import VGGHead
    vgg = VGGHead(pl).init_random(seed=5)
    sd = {k: v.double() for k, v in vgg.state_dict().items()}
    vgg.to(dev)
    g = torch.Generator().manual_seed(pl)
    fake_c = torch.rand(2, 3, 32, 32, generator=g) * 2 - 1
    real_c = torch.rand(2, 3, 32, 32, generator=g) * 2 - 1
    real = ops.raw_pack([(real_c.to(dev), True, 3)], 2, 32, 32, 4, dev)
    fake_nchw = fake_c.to(dev).requires_grad_(True)
    fake = ops.PackFn.apply(4, fake_nchw, True, 3)
    crit = L1PlusPerceptualLoss(10.0, 10.0, vgg, 1)
    feats = ops.nhwc_to_nchw_view(crit.features(fake)).detach()
    tot, l1, lp = crit(fake, real)
    tot.backward()
    fr = fake_c.double().requires_grad_(True)
    mean = torch.tensor(O.IMAGENET_MEAN, dtype=torch.float64).view(1, 3, 1, 1)
    std = torch.tensor(O.IMAGENET_STD, dtype=torch.float64).view(1, 3, 1, 1)
    want_feats = O.vgg_features(sd, ((fr + 1) / 2 - mean) / std, pl)
    assert tuple(feats.shape) == tuple(want_feats.shape)
    assert R.rel_l1(feats, want_feats.detach()) < 1e-5, R.rel_l1(feats, want_feats.detach())
    wt, wl1, wlp = O.l1_plus_perceptual(sd, fr, real_c.double(), 10.0, 10.0, 1, pl)
    wt.backward()
    for got, want in ((tot, wt), (l1, wl1), (lp, wlp)):
        assert abs(got.item() - want.item()) < 1e-4 * abs(want.item()), (pl, got.item(), want.item())
    assert R.rel_l1(fake_nchw.grad, fr.grad) < 1e-3, (pl, R.rel_l1(fake_nchw.grad, fr.grad))


def test_adam_matches_torch_fixture(dev):
    from mmhand_amd import ops
    fix = dict(np.load(os.path.join(G, "adam.npz")))
    p = torch.from_numpy(fix["p0"]).to(dev)
    m = torch.zeros_like(p); v = torch.zeros_like(p)
    for i in range(3):
        g = torch.from_numpy(fix["grads"][i]).to(dev)
        ops.adam_step(p, g, m, v, 2e-4, 0.5, 0.999, 1e-8, i + 1)
        assert np.allclose(p.cpu().numpy(), fix["ps"][i], rtol=1e-5, atol=1e-7), i


def test_pack_unpack_concat(dev):
    from mmhand_amd import ops
    a = _mk((2, 3, 8, 10), 1, dev); b = _mk((2, 21, 8, 10), 2, dev)
    nh = _mk((2, 8, 10, 4), 3, dev)
    out = ops.raw_pack([(a, True, 3), (b, True, 21)], 2, 8, 10, 24, dev)
    ref = torch.cat([a, b], 1).permute(0, 2, 3, 1)
    assert torch.equal(out, ref.contiguous())
    out2 = ops.raw_pack([(nh, False, 3), (a.permute(0, 1, 3, 2).contiguous().permute(0, 1, 3, 2), True, 3)],
                        2, 8, 10, 8, dev)      # NHWC source + non-contiguous NCHW source, zero padded
    assert torch.equal(out2[..., :3], nh[..., :3]) and torch.equal(out2[..., 3:6], a.permute(0, 2, 3, 1))
    assert float(out2[..., 6:].abs().sum()) == 0.0
    x = nh.clone().requires_grad_(True)
    y = ops.PackFn.apply(8, x, False, 3, a, True, 3)
    y.backward(torch.ones_like(y))
    assert torch.equal(x.grad[..., :3], torch.ones_like(x.grad[..., :3])) and float(x.grad[..., 3].abs().sum()) == 0.0


@pytest.mark.parametrize("lp", [True, 2], ids=["bf16", "fp16"])
@pytest.mark.parametrize("B,H,W,cs,Cd", [(2, 8, 10, (3, 21), 24), (3, 17, 19, (3, 3), 8), (1, 32, 32, (21, 21), 44),
                                          (2, 5, 7, (3,), 4), (1, 16, 16, (3, 3), 56)])
def test_pack_writes_the_stems_16bit_input_in_the_same_pass(B, H, W, cs, Cd, lp, dev):
    """mmh_pack_nhwc_lp16 (the gather staged through LDS): the fp32 NHWC tensor equals cat + zero pad, the 16-bit copy
    equals mmh_lp16_pad_cvt of it bit for bit (what the 16-bit stems read, ops.lp16_pad8 - which then takes the parked copy
    instead of converting), alone (only16), into one half of a two-batch buffer, and from NHWC / strided sources."""
    from mmhand_amd import ops, lib
    srcs = [_mk((B, c, H, W), 10 + i, dev) for i, c in enumerate(cs)]
    ref = torch.zeros(B, H, W, Cd, device=dev)
    ref[..., :sum(cs)] = torch.cat(srcs, 1).permute(0, 2, 3, 1)
    plain = ops.raw_pack([(t, True, c) for t, c in zip(srcs, cs)], B, H, W, Cd, dev)
    assert torch.equal(plain, ref)
    want16 = ops.lp16_pad8(ref, lp)                         # mmh_lp16_pad_cvt: nothing is parked for `ref`
    calls = []
    orig = lib.call
    lib.call = lambda name, *a: (calls.append(name), orig(name, *a))[1]
    try:
        out = ops.raw_pack([(t, True, c) for t, c in zip(srcs, cs)], B, H, W, Cd, dev, twin=lp)
        got16 = ops.lp16_pad8(out, lp)
        again = ops.lp16_pad8(out, lp)                      # the parked copy is handed out once
    finally:
        lib.call = orig
    assert calls == ["mmh_pack_nhwc_lp16", "mmh_lp16_pad_cvt"], calls
    assert torch.equal(out, ref) and torch.equal(got16.view(torch.int16), want16.view(torch.int16))
    assert torch.equal(again.view(torch.int16), want16.view(torch.int16)) and again.data_ptr() != got16.data_ptr()
    only = ops.raw_pack([(t, True, c) for t, c in zip(srcs, cs)], B, H, W, Cd, dev, twin=lp, only16=True)
    assert torch.equal(only.view(torch.int16), want16.view(torch.int16))
    two = torch.full((2 * B, H, W, want16.shape[3]), 7.0, dtype=want16.dtype, device=dev)
    ops.raw_pack([(t, True, c) for t, c in zip(srcs, cs)], B, H, W, Cd, dev, twin=lp, twin_out=two[B:], only16=True)
    assert torch.equal(two[B:].view(torch.int16), want16.view(torch.int16)) and float(two[:B].float().min()) == 7.0
    # a kept copy survives its use until the tensor changes
    out = ops.raw_pack([(t, True, c) for t, c in zip(srcs, cs)], B, H, W, Cd, dev, twin=lp, keep_twin=True)
    k1, k2 = ops.lp16_pad8(out, lp), ops.lp16_pad8(out, lp)
    assert k1.data_ptr() == k2.data_ptr()
    out.mul_(2.0)
    k3 = ops.lp16_pad8(out, lp)
    assert k3.data_ptr() != k1.data_ptr() and torch.equal(k3.view(torch.int16), ops.lp16_pad8(ref * 2.0, lp).view(torch.int16))
    # NHWC source (the generated image) + a strided NCHW source
    nh = _mk((B, H, W, 4), 30, dev)
    st = srcs[0].permute(0, 1, 3, 2).contiguous().permute(0, 1, 3, 2)
    mix = ops.raw_pack([(nh, False, 3), (st, True, cs[0])], B, H, W, 56, dev, twin=lp)
    refm = torch.zeros(B, H, W, 56, device=dev)
    refm[..., :3] = nh[..., :3]
    refm[..., 3:3 + cs[0]] = srcs[0].permute(0, 2, 3, 1)
    assert torch.equal(mix, refm)
    assert torch.equal(ops.lp16_pad8(mix, lp).view(torch.int16), ops.lp16_pad8(refm, lp).view(torch.int16))


@pytest.mark.parametrize("p,H,W", [(1, 8, 8), (3, 16, 12), (3, 5, 5), (1, 3, 4)])
def test_reflect_fold_is_pad_transpose(p, H, W, dev):
    from mmhand_amd import ops, lib as L
    import ctypes as C
    B, Cc = 2, 8
    gp = _mk((B, H + 2 * p, W + 2 * p, Cc), 1, dev)
    dx = torch.empty((B, H, W, Cc), device=dev)
    L.call("mmh_reflect_fold", ops._ptr(gp), ops._ptr(dx), B, H, W, Cc, p, ops._stream())
    x = torch.zeros((B, Cc, H, W), dtype=torch.float64, requires_grad=True)
    F.pad(x, (p,) * 4, mode="reflect").backward(gp.cpu().double().permute(0, 3, 1, 2))
    assert R.rel_l1(dx.permute(0, 3, 1, 2), x.grad) < 1e-6


def test_pose_maps_bit_exact_vs_reference_fixture(dev):
    """a15: support mask and arg-max indices bit-exact, values within 1 ulp (fp32)."""
    from mmhand_amd import ops
    fix = dict(np.load(os.path.join(G, "pose.npz")))
    n = 0
    for size, sfx in ((64, ""), (256, "256")):               # seven sets at 64 x 64, one at the training resolution
        for uv, maps, cords in zip(fix["uv" + sfx], fix["maps" + sfx], fix["cords" + sfx]):
            out = ops.pose_heatmaps(torch.from_numpy(uv).to(dev), size, size).cpu().numpy()
            assert np.array_equal(out > 0, maps > 0)
            ulp = np.abs(out.view(np.int32).astype(np.int64) - maps.view(np.int32).astype(np.int64))
            assert ulp.max() <= 1
            c = ops.map_to_cord(torch.from_numpy(maps).to(dev)).cpu().numpy()
            assert np.array_equal(c, cords)
            # and the indices of the DEVICE maps (values may differ by 1 ulp: ties must still resolve alike)
            assert np.array_equal(ops.map_to_cord(torch.from_numpy(out).to(dev)).cpu().numpy(), cords)
            n += 1
    assert n == 8


def test_decode_inputs_bit_exact(dev):
    """image normalise, depth decode, pose maps and the channel concat written
    straight into the stems' NHWC buffers; image/depth values bit-exact vs the float64 numpy maths
    of the reference loader, pose maps as in a15."""
    from mmhand_amd import ops
    rs = np.random.RandomState(3)
    B, H, W = 2, 32, 48
    imgs = [rs.randint(0, 256, size=(B, H, W, 3)).astype(np.uint8) for _ in range(2)]
    deps = [rs.randint(0, 256, size=(B, H, W, 3)).astype(np.uint8) for _ in range(2)]
    for d in deps:
        d[..., 1] = rs.randint(0, 3, size=(B, H, W))          # realistic range: depth < 700
    uvs = [rs.uniform(4, 28, size=(B, 21, 2)) for _ in range(2)]
    t = lambda a: torch.from_numpy(a).to(dev)
    xh1, xh2, xp, xd = ops.decode_inputs(t(imgs[0]), t(imgs[1]), t(deps[0]), t(deps[1]), t(uvs[0]), t(uvs[1]))
    for b in range(B):
        h1, d1 = O.decode_sample(imgs[0][b], deps[0][b])
        h2, d2 = O.decode_sample(imgs[1][b], deps[1][b])
        assert torch.equal(xh1[b, ..., :3].cpu(), h1.permute(1, 2, 0)) and float(xh1[b, ..., 3].abs().sum()) == 0
        assert torch.equal(xh2[b, ..., :3].cpu(), h2.permute(1, 2, 0))
        assert torch.equal(xd[b, ..., 0:3].cpu(), d1.permute(1, 2, 0))
        assert torch.equal(xd[b, ..., 3:6].cpu(), d2.permute(1, 2, 0)) and float(xd[b, ..., 6:].abs().sum()) == 0
        p1 = O.pose_heatmaps(uvs[0][b], H, W); p2 = O.pose_heatmaps(uvs[1][b], H, W)
        got = xp[b].cpu().numpy()
        ref = np.concatenate([p1, p2], 0).transpose(1, 2, 0)
        assert np.array_equal(got[..., :42] > 0, ref > 0)
        assert np.abs(got[..., :42].view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64)).max() <= 1
        assert float(np.abs(got[..., 42:]).sum()) == 0


@pytest.mark.parametrize("lp", [True, 2], ids=["bf16", "fp16"])
@pytest.mark.parametrize("act", [1, 2], ids=["relu", "tanh"])
def test_act_bwd_lp16(act, lp, dev):
    """mmh_act_bwd_lp16 == mmh_act_bwd followed by mmh_cvt_lp16, bit for bit (the activation backward of VGG19's
    conv + ReLU epilogues fused with the conversion their 16-bit dgrad needs)."""
    from mmhand_amd import ops
    g = torch.randn(3, 9, 11, 64, device=dev)
    y = torch.randn(3, 9, 11, 64, device=dev)
    if act == 2:
        y = torch.tanh(y)
    want = ops.lp16_twin(ops.raw_act_bwd(g, y, act), lp)
    got = ops.raw_act_bwd_lp16(g, y, act, lp)
    assert got.dtype == want.dtype and torch.equal(got, want)


@pytest.mark.parametrize("case", [(4, 32, 256), (32, 121, 64), (3, 50, 512)])
def test_norm_stats_two_level_merge(case, dev):
    """mmh_norm_stats_merge2 (one group over B * chunks partials, per-image blocks first) == the one-level merge of
    the same partials == the statistics of the tensor they describe (BatchNorm from conv-epilogue partials)."""
    from mmhand_amd import lib, ops
    B, chunks, C = case
    torch.manual_seed(B * 1000 + chunks)
    n = torch.randint(1, 40, (B, chunks, 1, C), device=dev).float()
    mean = torch.randn(B, chunks, 1, C, device=dev) * 2 + 0.7
    m2 = torch.rand(B, chunks, 1, C, device=dev) * n
    part = torch.cat((n, mean, m2), 2).contiguous()
    m1 = torch.empty(1, C, device=dev); q1 = torch.empty(1, C, device=dev)
    lib.call("mmh_norm_stats_merge", ops._ptr(part), 1, B * chunks, C, ops._ptr(m1), ops._ptr(q1), ops._stream())
    mb = torch.empty(1, C, device=dev); qb = torch.empty(1, C, device=dev)
    ws = torch.empty(B * 3 * C, device=dev)
    lib.call("mmh_norm_stats_merge2", ops._ptr(part), B * chunks, C, B, ops._ptr(ws), ws.numel() * 4, ops._ptr(mb),
             ops._ptr(qb), ops._stream())
    nd, md, qd = n.double().reshape(-1, C), mean.double().reshape(-1, C), m2.double().reshape(-1, C)
    tot = nd.sum(0)
    gm = (nd * md).sum(0) / tot
    gq = (qd + nd * (md - gm) ** 2).sum(0)
    for a, b in ((m1, q1), (mb, qb)):
        assert float((a.double().reshape(-1) - gm).abs().max()) < 1e-6
        assert float(((b.double().reshape(-1) - gq).abs() / gq).max()) < 1e-6


@pytest.mark.parametrize("lp", [1, 2])
@pytest.mark.parametrize("shape", [(9, 256, 512), (1, 64, 64), (9, 128, 64), (4, 192, 320), (9, 24, 64)])
def test_prep_weights_16bit_copies(shape, lp, dev):
    """mmh_prep_weights_bf16 / _fp16: the plain 16-bit copy [taps][Cin][Cout] and the transposed one [taps][Cout][Cin] are
    exactly the rounded fp32 values; either output may be omitted."""
    from mmhand_amd import lib as L, ops
    taps, cin, cout = shape
    wd = torch.bfloat16 if lp == 1 else torch.float16
    fn = "mmh_prep_weights_" + ("bf16" if lp == 1 else "fp16")
    w = torch.randn(taps, cin, cout, device=dev)
    wp = torch.zeros(taps, cin, cout, dtype=wd, device=dev)
    wt = torch.zeros(taps, cout, cin, dtype=wd, device=dev)
    L.call(fn, ops._ptr(w), taps, cin, cout, ops._ptr(wp), ops._ptr(wt), ops._stream())
    assert torch.equal(wp, w.to(wd)) and torch.equal(wt, w.to(wd).transpose(1, 2).contiguous())
    wt2 = torch.zeros_like(wt)
    L.call(fn, ops._ptr(w), taps, cin, cout, None, ops._ptr(wt2), ops._stream())
    assert torch.equal(wt2, wt)
    wp2 = torch.zeros_like(wp)
    L.call(fn, ops._ptr(w), taps, cin, cout, ops._ptr(wp2), None, ops._stream())
    assert torch.equal(wp2, wp)


@pytest.mark.parametrize("lp", [True, 2], ids=["bf16", "fp16"])
@pytest.mark.parametrize("n", [8, 4096, 2 * 37 * 41 * 64])
def test_l1_on_16bit_maps_and_its_relu_masked_gradient(lp, n, dev):
    """mmh_l1_fwd_lp16 / mmh_l1_relu_bwd_lp16 (F.l1_loss on the fp16 VGG features of apex O1, losses/L1_plus_perceptualLoss.py:66,
    and its gradient times the mask of features[3]) against fp64 on the same 16-bit values, through the C-ABI; n % 8 != 0 refused."""
    from mmhand_amd import lib as L
    from mmhand_amd import ops
    wd = torch.bfloat16 if lp is True else torch.float16
    g = torch.Generator().manual_seed(n)
    a = torch.relu(torch.randn(n, generator=g)).to(wd).to(dev)         # post-ReLU features: half of them exactly zero
    b = torch.relu(torch.randn(n, generator=g)).to(wd).to(dev)
    b[: n // 16] = a[: n // 16]                                            # equal pairs: sign 0
    weight, denom = 10.0, float(n)
    ws = torch.empty(max(int(L.load().mmh_reduce_ws_bytes(n)), 16) // 4 + 4, dtype=torch.float32, device=dev)
    out = torch.empty((), dtype=torch.float32, device=dev)
    L.call("mmh_l1_fwd_lp16", a.data_ptr(), b.data_ptr(), n, weight, denom, ops._dt(lp), out.data_ptr(), ws.data_ptr(),
           ws.numel() * 4, ops._stream())
    want = weight * float((a.double() - b.double()).abs().sum()) / denom
    assert abs(float(out) - want) <= 2e-6 * max(want, 1e-6), (float(out), want)
    gs = torch.tensor(0.75, dtype=torch.float32, device=dev)
    got = torch.empty_like(a)
    L.call("mmh_l1_relu_bwd_lp16", a.data_ptr(), b.data_ptr(), n, weight, denom, gs.data_ptr(), ops._dt(lp), got.data_ptr(),
           ops._stream())
    k = torch.tensor(weight / denom, dtype=torch.float32) * 0.75
    ref = (torch.sign(a.float() - b.float()) * (a.float() > 0) * k.to(dev)).to(wd)
    assert torch.equal(got, ref), float((got.float() - ref.float()).abs().max())
    lib = L.load()
    assert lib.mmh_l1_fwd_lp16(a.data_ptr(), b.data_ptr(), 12, weight, 12.0, ops._dt(lp), out.data_ptr(), ws.data_ptr(),
                               ws.numel() * 4, ops._stream()) != 0
    assert lib.mmh_l1_relu_bwd_lp16(a.data_ptr(), b.data_ptr(), 12, weight, 12.0, gs.data_ptr(), 0, got.data_ptr(),
                                    ops._stream()) != 0               # and a dtype that is not 16-bit


@pytest.mark.parametrize("groups,chunks,C", [(2, 8, 64), (3, 32, 256), (2, 127, 64), (4, 128, 256), (2, 300, 96), (1, 1024, 32)])
def test_norm_stats_merge_finalize_vs_fp64(groups, chunks, C, dev):
    """mmh_norm_stats_merge_finalize: Chan's merge of per-tile (count, mean, M2) partials - what the conv epilogues leave for
    the InstanceNorm behind them (models/Generator.py:66-77) - into mean / invstd / scale / shift, against fp64 on the same
    partials, at 8 to 1024 partials per group (the 512x512 shapes have 128), empty partials skipped."""
    from mmhand_amd import lib as L
    from mmhand_amd import ops
    g = torch.Generator().manual_seed(groups * 1000 + chunks)
    n = torch.randint(0, 5, (groups, chunks, 1, C), generator=g).float() * 16          # some partials empty (count 0)
    n[:, 0] = 64
    mean_p = torch.randn(groups, chunks, 1, C, generator=g) * 0.5 + 3.0
    m2_p = torch.rand(groups, chunks, 1, C, generator=g) * n
    part = torch.cat([n, mean_p, m2_p], 2).contiguous().to(dev)                        # [groups][chunks][3][C]
    count = n.sum(1).squeeze(1).double()                                               # [groups][C]
    nd, md, qd = n.double().squeeze(2), mean_p.double().squeeze(2), m2_p.double().squeeze(2)
    mean_w = (nd * md).sum(1) / count
    m2_w = (qd + nd * (md - mean_w[:, None]) ** 2).sum(1)
    rows = float(count.max())       # the entry point takes ONE count (rows per plane): use partials with equal totals
    sel = count == rows
    outs = [torch.empty(groups, C, device=dev) for _ in range(5)]
    L.call("mmh_norm_stats_merge_finalize", part.data_ptr(), groups, chunks, C, rows, ops.EPS, *[o.data_ptr() for o in outs],
           ops._stream())
    mean, m2, scale, shift, invstd = [o.cpu().double() for o in outs]
    assert float((mean - mean_w).abs().max()) <= 2e-6 * 3.5
    assert float(((m2 - m2_w).abs() / m2_w.clamp_min(1e-3)).max()) <= 2e-5
    is_w = 1.0 / torch.sqrt(m2_w / rows + ops.EPS)
    assert float(((invstd - is_w).abs() / is_w)[sel].max() if bool(sel.any()) else 0.0) <= 2e-5
    assert torch.allclose(scale, invstd) and torch.allclose(shift, -(mean * invstd), rtol=1e-5, atol=1e-6)


def test_stream_handle_fast_path_matches_torch(dev):
    """ops._stream() reads the current stream through torch's raw bindings (0.3 us instead of 9 us per C-ABI call): the same
    handle torch.cuda.current_stream() reports - on the default stream, inside a side-stream context and after it."""
    from mmhand_amd import ops
    assert ops._stream().value == torch.cuda.current_stream().cuda_stream or (
        ops._stream().value is None and torch.cuda.current_stream().cuda_stream == 0)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        assert ops._stream().value == side.cuda_stream
        assert ops._stream().value == torch.cuda.current_stream().cuda_stream
    assert (ops._stream().value or 0) == torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize("world,C,affine", [(1, 64, True), (2, 256, True), (8, 512, True), (4, 128, False)])
def test_syncbn_merge_finalize_equals_merge_then_finalize(world, C, affine, dev):
    """mmh_syncbn_merge_finalize (one launch per SyncBN site instead of a block copy, mmh_norm_stats_merge and
    mmh_norm_finalize) on a site's triples at a column offset of the all-gather's rows: mean, M2, scale, shift, invstd and the
    running statistics bit for bit those of the two calls on the copied [ranks][3][C] block."""
    import ctypes as C_
    from mmhand_amd import lib as L, ops
    g = torch.Generator().manual_seed(world * 1000 + C)
    off, other = 3 * 32, 3 * 48                      # two more sites packed around this one in the message
    rows = 4096.0
    buf = torch.randn((world, off + 3 * C + other), generator=g).to(dev)
    blk = buf[:, off:off + 3 * C].reshape(world, 3, C)
    blk[:, 0] = rows
    blk[:, 2] = blk[:, 2].abs() * 50.0
    gamma = (1.0 + 0.1 * torch.randn(C, generator=g)).to(dev) if affine else None
    beta = (0.1 * torch.randn(C, generator=g)).to(dev) if affine else None
    rm0, rv0 = torch.randn(C, generator=g).to(dev), (1.0 + torch.rand(C, generator=g)).to(dev)
    count = rows * world
    # the two-call path
    cb = blk.reshape(world * 3, C).contiguous()
    mean_a = torch.empty((1, C), device=dev); m2_a = torch.empty_like(mean_a)
    L.call("mmh_norm_stats_merge", ops._ptr(cb), 1, world, C, ops._ptr(mean_a), ops._ptr(m2_a), ops._stream())
    rm_a, rv_a = rm0.clone(), rv0.clone()
    sc_a, sh_a, is_a = ops.raw_norm_finalize(mean_a, m2_a, count, gamma, beta, rm_a, rv_a)
    # the fused call, reading the message in place
    rm_b, rv_b = rm0.clone(), rv0.clone()
    mean_b, m2_b, sc_b, sh_b, is_b = ops._GatheredStats(buf, off, world, C, count).finalize(gamma, beta, rm_b, rv_b)
    for a, b in ((mean_a, mean_b), (m2_a, m2_b), (sc_a, sc_b), (sh_a, sh_b), (is_a, is_b), (rm_a, rm_b), (rv_a, rv_b)):
        assert torch.equal(a, b)
    assert not torch.equal(rm_a, rm0) and torch.isfinite(sc_b).all()
