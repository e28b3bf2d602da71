"""fp32 wgrad of the 7x7 / reflect-pad 3 stems on the LDS-band kernel (conv_stem.hip: the input reflect-padded
once, a filter row's input band and a 2 x 64 pixel tile of dy staged in LDS, both MFMA operands plain LDS reads)
against the generic direct wgrad kernel and the fp64 oracle (models/Generator.py:158-168 stems)."""
import pytest
import torch

from oracle import ops_ref as R

pytestmark = pytest.mark.gpu


def _mk(shape, seed, dev, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev)


@pytest.mark.parametrize("Cin", [8, 44])
@pytest.mark.parametrize("B,H,W", [(2, 8, 64), (1, 6, 128)])
def test_stem_wgrad_vs_generic_and_oracle(Cin, B, H, W, dev, monkeypatch):
    from mmhand_amd import lib as L, ops
    x = _mk((B, H, W, Cin), 1, dev)
    dy = _mk((B, H, W, 64), 2, dev)
    d = ops.conv_desc(B, H, W, Cin, 64, 7, 1, 3, True)
    assert L.load().mmh_conv7_stem_wgrad_supported(ops.C.byref(d)) == 1
    calls = []
    real = L.call
    monkeypatch.setattr(L, "call", lambda n, *a: (calls.append(n), real(n, *a))[1])
    dw = ops.raw_conv_wgrad(x, dy, 7, 1, 3, True)
    assert "mmh_conv7_stem_wgrad" in calls and "mmh_conv2d_wgrad" not in calls
    monkeypatch.setattr(ops, "USE_STEM_WGRAD", False)
    dw_gen = ops.raw_conv_wgrad(x, dy, 7, 1, 3, True)
    assert "mmh_conv2d_wgrad" in calls
    ref = R.conv2d_grads(x.double().cpu(), torch.zeros(7, 7, Cin, 64, dtype=torch.float64), None, dy.double().cpu(),
                         1, 3, True)[2]
    assert R.rel_l1(dw, ref) < 2e-6 and R.rel_l1(dw_gen, ref) < 2e-6
    # deterministic (fixed-order split-K sums), and the accumulate path adds into the target
    monkeypatch.setattr(ops, "USE_STEM_WGRAD", True)
    assert torch.equal(ops.raw_conv_wgrad(x, dy, 7, 1, 3, True), dw)
    tgt = _mk((7, 7, Cin, 64), 3, dev)
    t0 = tgt.clone()
    ops.raw_conv_wgrad(x, dy, 7, 1, 3, True, out=tgt)
    assert torch.allclose(tgt, t0 + dw, rtol=1e-6, atol=1e-5)


def test_stem_wgrad_unsupported_shapes_fall_back(dev):
    from mmhand_amd import lib as L, ops
    for (H, W, Cin, Cout, k, pad, refl) in [(8, 32, 8, 64, 7, 3, True), (8, 64, 48, 64, 7, 3, True), (8, 64, 12, 64, 7, 3, True), (8, 64, 24, 64, 7, 3, True), (8, 64, 4, 64, 7, 3, True), (8, 64, 8, 32, 7, 3, True),
                                            (8, 64, 8, 64, 7, 3, False), (7, 64, 8, 64, 7, 3, True)]:
        d = ops.conv_desc(1, H, W, Cin, Cout, k, 1, pad, refl)
        assert L.load().mmh_conv7_stem_wgrad_supported(ops.C.byref(d)) == 0
    x = _mk((1, 8, 32, 8), 1, dev)
    dy = _mk((1, 8, 32, 64), 2, dev)
    dw = ops.raw_conv_wgrad(x, dy, 7, 1, 3, True)           # generic kernel
    ref = R.conv2d_grads(x.double().cpu(), torch.zeros(7, 7, 8, 64, dtype=torch.float64), None, dy.double().cpu(), 1, 3, True)[2]
    assert R.rel_l1(dw, ref) < 2e-6
