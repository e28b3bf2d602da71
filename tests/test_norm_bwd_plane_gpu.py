"""mmh_norm_bwd_fused - the InstanceNorm backward of the 16-bit mode in one pass over a plane held on chip
(pointwise.hip: norm_bwd_plane_kernel; models/Generator.py:66-77 norm -> ReLU -> Dropout in the backward direction) -
against the two-pass entry points it replaces (mmh_norm_bwd_reduce + mmh_norm_bwd_apply) and against the formula in fp64.

The per-element arithmetic is the two-pass kernels' regrouped (three coefficients instead of four) and the plane sums are
taken in a different fixed order, so the results agree to rounding, not bit for bit: sums within 2e-6 relative, a 16-bit dx
within one unit in the last place on a few elements (relative L1 <= 1e-3 bf16 / 2e-4 fp16), an fp32 dx within 2e-6."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().sum() / b.abs().sum().clamp_min(1e-30))


# (B, H, W, C): rows = H * W per (sample, channel) plane; 4096 x 256 / 512 are the PATBlock sites of the 256x256 step
SHAPES = [(3, 16, 16, 64), (2, 23, 17, 32), (2, 64, 64, 256), (2, 64, 64, 512), (2, 32, 32, 128), (5, 8, 8, 8)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("lp", [True, 2], ids=["bf16", "fp16"])
@pytest.mark.parametrize("masked,g32,dx32,affine", [(True, False, False, False), (False, True, False, False),
                                                     (True, False, True, True), (False, False, False, True),
                                                     (False, True, True, False)])
def test_fused_norm_backward_vs_two_pass_and_fp64(shape, lp, masked, g32, dx32, affine, dev):
    from mmhand_amd import lib as L
    from mmhand_amd import ops
    B, H, W, Cc = shape
    rows = H * W
    wd = ops._wd(lp)
    gen = torch.Generator(device=dev).manual_seed(B * 1000 + Cc + rows)
    x = (torch.randn((B, H, W, Cc), generator=gen, device=dev) * 1.7 + 0.3).to(wd)
    g = torch.randn((B, H, W, Cc), generator=gen, device=dev)
    g = g if g32 else g.to(wd)
    xf = x.float().view(B, rows, Cc)
    mean = xf.mean(1).contiguous()
    var = xf.var(1, unbiased=False)
    invstd = (var + ops.EPS).rsqrt().contiguous()
    gamma = (1 + 0.1 * torch.randn(Cc, generator=gen, device=dev)).contiguous() if affine else None
    drop_p = 0.5 if masked else 0.0
    kb = None
    if masked:      # the keep bits as the forward writes them (ReLU + dropout)
        _, kb = ops.raw_scale_shift_act(x, invstd * (gamma if affine else 1.0), -mean * invstd * (gamma if affine else 1.0),
                                        None, True, drop_p, 12345, None, keep_bits=True, out_lp=lp)
    mk = 2 if masked else 0
    tdt = ops._tdt
    assert L.load().mmh_norm_bwd_fused_supported(B, rows, Cc, mk, tdt(g), tdt(x)) == 1
    dxd = torch.float32 if dx32 else wd
    # two-pass
    ws = ops._ws(L.load().mmh_norm_bwd_ws_bytes(B, rows, Cc), x)
    s1 = torch.empty((B, Cc), device=dev); s2 = torch.empty((B, Cc), device=dev)
    L.call("mmh_norm_bwd_reduce", ops._ptr(g), ops._ptr(kb), ops._ptr(x), ops._ptr(mean), ops._ptr(invstd), B, rows, Cc, mk,
           drop_p, ops._ptr(s1), ops._ptr(s2), ops._ptr(ws), ws.numel() * 4, tdt(g), tdt(x), ops._stream())
    dx2 = torch.empty((B, H, W, Cc), dtype=dxd, device=dev)
    L.call("mmh_norm_bwd_apply", ops._ptr(g), ops._ptr(kb), ops._ptr(x), ops._ptr(mean), ops._ptr(invstd), ops._ptr(gamma),
           ops._ptr(s1), ops._ptr(s2), float(rows), B, rows, Cc, mk, drop_p, ops._ptr(dx2), tdt(g), tdt(x), tdt(dx2), ops._stream())
    # one pass
    f1 = torch.full((B, Cc), float("nan"), device=dev); f2 = torch.full((B, Cc), float("nan"), device=dev)
    dx1 = torch.full((B, H, W, Cc), float("nan"), dtype=dxd, device=dev)
    L.call("mmh_norm_bwd_fused", ops._ptr(g), ops._ptr(kb), ops._ptr(x), ops._ptr(mean), ops._ptr(invstd), ops._ptr(gamma),
           float(rows), B, rows, Cc, mk, drop_p, ops._ptr(f1), ops._ptr(f2), ops._ptr(dx1), tdt(g), tdt(x), tdt(dx1), ops._stream())
    torch.cuda.synchronize()
    assert bool(torch.isfinite(dx1.float()).all()) and bool(torch.isfinite(f1).all()) and bool(torch.isfinite(f2).all())
    scale = lambda t: float(t.abs().double().mean()) + 1e-30     # noqa: E731
    assert float((f1 - s1).abs().max()) <= 2e-6 * rows ** 0.5 * max(scale(s1), scale(g.float()) * rows ** 0.5), "s1"
    assert float((f2 - s2).abs().max()) <= 2e-6 * rows ** 0.5 * max(scale(s2), scale(g.float()) * rows ** 0.5), "s2"
    tol = 2e-6 if dx32 else (1e-3 if lp is True else 2e-4)
    assert _rel(dx1, dx2) <= tol, (_rel(dx1, dx2), tol)
    # fp64 formula: dx = gamma invstd (dz - mean(dz) - xhat mean(dz xhat))
    gd, xd = g.double().view(B, rows, Cc), x.double().view(B, rows, Cc)
    if masked:
        bits = kb.view(B, rows, Cc // 4).to(torch.int32)
        keep = torch.stack([(bits >> e) & 1 for e in range(4)], -1).reshape(B, rows, Cc).double()
        gd = gd * keep / (1 - drop_p)
    xhat = (xd - mean.double()[:, None]) * invstd.double()[:, None]
    ref = (gd - gd.mean(1, keepdim=True) - xhat * (gd * xhat).mean(1, keepdim=True)) * invstd.double()[:, None]
    if affine:
        ref = ref * gamma.double()
    out_tol = 2e-5 if dx32 else (4e-3 if lp is True else 6e-4)
    assert _rel(dx1.view(B, rows, Cc), ref) <= out_tol, _rel(dx1.view(B, rows, Cc), ref)


def test_fused_norm_backward_declines_what_it_cannot_hold(dev):
    from mmhand_amd import lib as L
    l = L.load()
    assert l.mmh_norm_bwd_fused_supported(32, 65536, 64, 2, L.BF16, L.BF16) == 0      # 256x256 planes: two passes
    assert l.mmh_norm_bwd_fused_supported(32, 4096, 256, 2, L.BF16, L.F32) == 0       # fp32 x: the Winograd path's business
    assert l.mmh_norm_bwd_fused_supported(32, 4096, 256, 1, L.BF16, L.BF16) == 0      # masked by the fp32 output
    assert l.mmh_norm_bwd_fused_supported(32, 4096, 256, 2, L.BF16, L.BF16) == 1
    assert l.mmh_norm_bwd_fused_supported(32, 4096, 256, 0, L.F32, L.BF16) == 1
    assert l.mmh_norm_bwd_fused(None, None, None, None, None, None, C.c_double(1.0), 1, 65536, 64, 0, 0.0, None, None, None,
                                L.BF16, L.BF16, L.BF16, None) != 0
