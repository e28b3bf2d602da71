"""conv_s2_lp16.hip - the 16-bit fprop of the 3x3 / stride-2 / zero-pad-1 convs (nn.Conv2d(c, 2c, 3, 2, 1), models/Generator.py
:166-180, models/Discriminator.py:86-92) with register-resident weights and a de-interleaved LDS halo - behind
mmh_conv_lp16 (mode 0).  Against the general 16-bit kernel it replaces (mmh_set_option("lp16_s2f", 0)): bit-identical at 64
input channels (same order of summation), 2e-6 at 128 (chunk-outer instead of tap-outer order); and against the fp64
convolution of the same 16-bit operands."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().sum() / b.abs().sum().clamp_min(1e-30))


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 64, 64, 64, 128), (1, 32, 64, 128, 256), (3, 16, 32, 64, 128),
                                            (2, 16, 96, 128, 256), (5, 48, 32, 64, 256)])
@pytest.mark.parametrize("lp", [True, 2], ids=["bf16", "fp16"])
@pytest.mark.parametrize("out16,bias,act", [(True, False, 0), (False, True, 1), (True, True, 0)])
def test_stride2_fprop_vs_general_kernel_and_fp64(B, H, W, Cin, Cout, lp, out16, bias, act, dev):
    import ctypes as C
    from mmhand_amd import lib as L
    from mmhand_amd import ops
    gen = torch.Generator(device=dev).manual_seed(H * 7 + Cin)
    x = torch.randn((B, H, W, Cin), generator=gen, device=dev)
    w = torch.randn((3, 3, Cin, Cout), generator=gen, device=dev) * 0.05
    b = torch.randn((Cout,), generator=gen, device=dev) if bias else None
    x16 = ops.lp16_twin(x, lp)
    d = ops.conv_desc(B, H, W, Cin, Cout, 3, 2, 1, False)
    assert ops.lp16g_ok(d, 0, lp)
    outs = {}
    for on in (1, 0):
        L.check(L.load().mmh_set_option(b"lp16_s2f", 2 * on), "set_option")     # 2: the new kernel also at 128 input channels
        try:
            outs[on] = ops.raw_conv_lp16g(ops.conv_desc(B, H, W, Cin, Cout, 3, 2, 1, False), 0, x16, w, b, act, lp, out16=out16)
        finally:
            L.check(L.load().mmh_set_option(b"lp16_s2f", 1), "set_option")
    torch.cuda.synchronize()
    new, old = outs[1], outs[0]
    assert new.dtype == (ops._wd(lp) if out16 else torch.float32) and bool(torch.isfinite(new.float()).all())
    if Cin == 64:
        assert torch.equal(new, old)
    else:
        assert _rel(new, old) <= (2e-6 if not out16 else 1e-3)
    # fp64 convolution of the SAME 16-bit operands
    wp, wt = ops.bf16_weights(w, lp)        # wt: [3,3,Cout,Cin] 16-bit
    xr = x16.double().permute(0, 3, 1, 2)
    wr = wt.double().permute(2, 3, 0, 1)    # OIHW
    ref = F.conv2d(xr.cpu(), wr.cpu(), None if b is None else b.double().cpu(), stride=2, padding=1)
    if act == 1:
        ref = ref.relu()
    ref = ref.permute(0, 2, 3, 1)
    tol = 2e-5 if not out16 else (3e-3 if lp is True else 4e-4)
    assert _rel(new.cpu(), ref) <= tol, _rel(new.cpu(), ref)


def test_stride2_fprop_kernel_declines_other_shapes(dev):
    from mmhand_amd import lib as L
    from mmhand_amd import ops
    import ctypes as C
    l = L.load()
    d64 = ops.conv_desc(32, 256, 256, 64, 128, 3, 2, 1, False)
    d64.dtype = L.BF16
    assert l.mmh_conv_lp16_supported(C.byref(d64), 0) == 1
    # 20 x 20 outputs: not a multiple of the 8 x 16 tile -> the general kernel (same entry point, same result contract)
    x = torch.randn((1, 40, 40, 64), device=dev)
    w = torch.randn((3, 3, 64, 128), device=dev) * 0.05
    y = ops.raw_conv_lp16g(ops.conv_desc(1, 40, 40, 64, 128, 3, 2, 1, False), 0, ops.lp16_twin(x, True), w, None, 0, True)
    ref = F.conv2d(x.bfloat16().double().permute(0, 3, 1, 2).cpu(), w.bfloat16().double().permute(3, 2, 0, 1).cpu(), stride=2, padding=1)
    assert _rel(y.cpu(), ref.permute(0, 2, 3, 1)) <= 2e-5


@pytest.mark.parametrize("kind,B,H,W,Cin", [("s2", 2, 64, 64, 64), ("s2", 3, 32, 96, 64), ("stem", 2, 64, 64, 24), ("stem", 1, 32, 48, 44),
                                            ("stem", 2, 32, 32, 8)])
@pytest.mark.parametrize("lp", [True, 2], ids=["bf16", "fp16"])
def test_epilogue_statistics_match_the_stored_output(kind, B, H, W, Cin, lp, dev):
    """The partial statistics the stride-2 kernel (conv_s2_lp16.hip) and the 7x7 stem kernel (conv_stem16.hip) leave for the
    InstanceNorm behind them (common.h: wave_tile_stats; mmh_conv_lp16_fprop_stats / mmh_conv_stem16_stats), merged by
    mmh_norm_stats_merge: mean and M2 per (image, channel) of the 16-bit output AS STORED, against float64 over that output."""
    from mmhand_amd import lib as L
    from mmhand_amd import ops
    gen = torch.Generator(device=dev).manual_seed(H + Cin)
    Cout = 128 if kind == "s2" else 64
    k, stride, pad = (3, 2, 1) if kind == "s2" else (7, 1, 3)
    x = torch.randn((B, H, W, ops.pad4(Cin)), generator=gen, device=dev) * 1.3 + 0.2
    w = torch.randn((k, k, ops.pad4(Cin), Cout), generator=gen, device=dev) * 0.05
    bias = torch.randn((Cout,), generator=gen, device=dev)
    ops._pending_stats.clear()
    ops.FUSE_NORM_STATS_NARROW, keep = True, ops.FUSE_NORM_STATS_NARROW      # (off by default: measured slower at step level)
    d = ops.conv_desc(B, H, W, ops.pad4(Cin), Cout, k, stride, pad, kind == "stem")
    if kind == "s2":
        y = ops.raw_conv_lp16g(d, 0, ops.lp16_twin(x, lp), w, bias, 0, lp, out16=True, want_stats=True)
    else:
        y = ops.raw_conv_lp16_flat(d, x, w, bias, 0, lp, out16=True, want_stats=True)
    ops.FUSE_NORM_STATS_NARROW = keep
    pend = ops._pending_stats.get(y.data_ptr())
    assert pend is not None, "the kernel left no partial statistics"
    stats = pend[0]
    assert stats.shape[0] == B and stats.shape[2] == 3 and stats.shape[3] == Cout
    mean, m2, rows = ops.raw_norm_stats(y, B)           # takes the parked partials (merge), does not read y
    torch.cuda.synchronize()
    assert rows == y.shape[1] * y.shape[2] and float(stats[:, :, 0].sum(1).min()) == float(stats[:, :, 0].sum(1).max()) == rows
    yd = y.double().view(B, -1, Cout)
    rm, rq = yd.mean(1), ((yd - yd.mean(1, keepdim=True)) ** 2).sum(1)
    assert float((mean.double() - rm).abs().max()) <= 2e-6 * float(yd.abs().max()), float((mean.double() - rm).abs().max())
    assert float(((m2.double() - rq).abs() / rq).max()) <= 2e-5, float(((m2.double() - rq).abs() / rq).max())


@pytest.mark.parametrize("B,H,W", [(2, 64, 64), (1, 32, 96), (3, 16, 32)])
@pytest.mark.parametrize("lp", [True, 2], ids=["bf16", "fp16"])
@pytest.mark.parametrize("out16,bias", [(True, False), (False, True)])
def test_stride2_dgrad_vs_general_kernel_and_fp64(B, H, W, lp, out16, bias, dev):
    """conv_s2d_kernel (mmh_conv_lp16 mode 1 at 64 -> 128 channels: the input gradient of the stride-2 conv = the forward of
    ConvTranspose2d(128, 64, 3, 2, 1, 1), models/Generator.py:225-243): against the general per-class kernel (2e-6 fp32 output;
    the k order differs) and against the fp64 transposed convolution of the same 16-bit operands."""
    from mmhand_amd import lib as L
    from mmhand_amd import ops
    Cin, Cout = 64, 128
    gen = torch.Generator(device=dev).manual_seed(H * 3 + W)
    dy = torch.randn((B, H // 2, W // 2, Cout), generator=gen, device=dev)
    w = torch.randn((3, 3, Cin, Cout), generator=gen, device=dev) * 0.05
    b = torch.randn((Cin,), generator=gen, device=dev) if bias else None
    dy16 = ops.lp16_twin(dy, lp)
    outs = {}
    for on in (1, 0):
        L.check(L.load().mmh_set_option(b"lp16_s2f", on), "set_option")
        try:
            outs[on] = ops.raw_conv_lp16g(ops.conv_desc(B, H, W, Cin, Cout, 3, 2, 1, False), 1, dy16, w, b, 0, lp, out16=out16)
        finally:
            L.check(L.load().mmh_set_option(b"lp16_s2f", 1), "set_option")
    torch.cuda.synchronize()
    new, old = outs[1], outs[0]
    assert tuple(new.shape) == (B, H, W, Cin) and bool(torch.isfinite(new.float()).all())
    assert _rel(new, old) <= (2e-6 if not out16 else 1e-3), _rel(new, old)
    wp, _ = ops.bf16_weights(w, lp)         # [3,3,Cin,Cout] 16-bit
    ref = F.conv_transpose2d(dy16.double().permute(0, 3, 1, 2).cpu(), wp.double().permute(3, 2, 0, 1).cpu(),
                             None if b is None else b.double().cpu(), stride=2, padding=1, output_padding=1)
    tol = 2e-5 if not out16 else (3e-3 if lp is True else 4e-4)
    assert _rel(new.cpu(), ref.permute(0, 2, 3, 1)) <= tol, _rel(new.cpu(), ref.permute(0, 2, 3, 1))


@pytest.mark.parametrize("B,H,W", [(2, 64, 64), (1, 8, 16), (3, 24, 48), (1, 40, 40)])
@pytest.mark.parametrize("lp", [True, 2], ids=["bf16", "fp16"])
@pytest.mark.parametrize("mode,out16,bias,act", [(0, True, True, 1), (0, False, False, 0), (1, True, False, 0), (1, False, False, 0)])
def test_stride1_64_to_64_vs_general_kernel_and_fp64(B, H, W, lp, mode, out16, bias, act, dev):
    """The stride-1 form of the register-resident-weights kernel (conv_s2f_kernel<..., S = 1, NOUT = 64>): VGG19's conv1_2 as the
    perceptual loss runs it (losses/L1_plus_perceptualLoss.py:22-27: 64 -> 64, zero padding, bias + ReLU) and its input
    gradient (the same conv of dy with mirrored taps on the plain weight copy) - against the general kernel it replaces behind
    mmh_conv_lp16 and against the fp64 convolution / transposed convolution of the same 16-bit operands; a shape off the
    8 x 16 tile (40 x 40) stays on the general kernel."""
    from mmhand_amd import lib as L
    from mmhand_amd import ops
    gen = torch.Generator(device=dev).manual_seed(H * 5 + W + mode)
    xin = torch.randn((B, H, W, 64), generator=gen, device=dev)           # x (mode 0) or dy (mode 1)
    w = torch.randn((3, 3, 64, 64), generator=gen, device=dev) * 0.05
    b = torch.randn((64,), generator=gen, device=dev) if bias else None
    x16 = ops.lp16_twin(xin, lp)
    outs = {}
    for on in (1, 0):
        L.check(L.load().mmh_set_option(b"lp16_s2f", on), "set_option")
        try:
            outs[on] = ops.raw_conv_lp16g(ops.conv_desc(B, H, W, 64, 64, 3, 1, 1, False), mode, x16, w, b, act, lp, out16=out16)
        finally:
            L.check(L.load().mmh_set_option(b"lp16_s2f", 1), "set_option")
    torch.cuda.synchronize()
    new, old = outs[1], outs[0]
    assert new.dtype == (ops._wd(lp) if out16 else torch.float32) and bool(torch.isfinite(new.float()).all())
    if mode == 0 or H % 8 or W % 16:
        assert torch.equal(new, old)            # fprop: the same order of summation; off-tile shapes: the same kernel
    else:
        assert _rel(new, old) <= (2e-6 if not out16 else 2e-3)      # dgrad: taps in mirrored order
    wp, wt = ops.bf16_weights(w, lp)            # wp [3,3,Cin,Cout], wt [3,3,Cout,Cin], 16-bit
    xr = x16.double().permute(0, 3, 1, 2).cpu()
    if mode == 0:
        ref = F.conv2d(xr, wt.double().permute(2, 3, 0, 1).cpu(), None if b is None else b.double().cpu(), stride=1, padding=1)
        if act == 1:
            ref = ref.relu()
    else:                                       # dx = conv_transpose of dy with the OIHW weight [Cout, Cin, 3, 3]
        ref = F.conv_transpose2d(xr, wp.double().permute(3, 2, 0, 1).cpu(), None, stride=1, padding=1)
    ref = ref.permute(0, 2, 3, 1)
    tol = 2e-5 if not out16 else (3e-3 if lp is True else 4e-4)
    assert _rel(new.cpu(), ref) <= tol, _rel(new.cpu(), ref)
