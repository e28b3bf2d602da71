"""Full-size (BASELINE.json configs[1]: B=32, 64x64x512 PATBlock tensors, 256x256 stems) checks of
the HIP kernels through size-independent properties — the oracle cannot run these sizes in
seconds, the algebra can:
  * adjointness  <conv(x), dy> == <x, dgrad(dy)> == <w, wgrad(x, dy)>  (exact identities of the
    bilinear map, evaluated in fp64 on the device),
  * linearity    conv(a*x1 + b*x2) == a*conv(x1) + b*conv(x2),
  * norm         per-(sample,channel) mean 0 / variance 1 after InstanceNorm, per-channel for batch,
  * Adam         a zero gradient leaves parameters untouched; a constant gradient moves every
                 parameter by exactly lr on the first step,
  * dropout      keep rate 0.5 and scale 2 on 134M elements.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dot(a, b):
    return (a.double() * b.double()).sum().item()


def _rand(shape, seed, dev, scale=1.0):
    g = torch.Generator(device=dev).manual_seed(seed)
    return (torch.rand(shape, generator=g, device=dev) * 2 - 1) * scale


FULL = [
    # B, H, W, Cin, Cout, k, stride, pad, reflect
    (32, 64, 64, 512, 512, 3, 1, 1, True),      # the dominant PATBlock conv
    (32, 64, 64, 512, 256, 3, 1, 1, True),
    (32, 128, 128, 128, 256, 3, 2, 1, False),   # stride-2 down
    (32, 256, 256, 44, 64, 7, 1, 3, True),      # pose stem
    (32, 256, 256, 64, 4, 7, 1, 3, True),       # head
]


@pytest.mark.parametrize("case", FULL)
@pytest.mark.parametrize("bf16", [False, True])
def test_conv_adjoint_identities_full_size(case, bf16, dev):
    from mmhand_amd import ops
    B, H, W, Cin, Cout, k, s, p, refl = case
    x = _rand((B, H, W, Cin), 1, dev)
    w = _rand((k, k, Cin, Cout), 2, dev, 0.05)
    ops.bump_weights_epoch()
    y = ops.raw_conv_fprop(x, w, None, s, p, refl, 0, bf16=bf16)
    dy = _rand(tuple(y.shape), 3, dev)
    dx = ops.raw_conv_dgrad(dy, w, x.shape, s, p, refl, bf16=bf16)
    dw = ops.raw_conv_wgrad(x, dy, k, s, p, refl, bf16=bf16)
    a, b, c = _dot(y, dy), _dot(x, dx), _dot(w, dw)
    tol = 2e-3 if bf16 else 2e-5          # bf16 rounds the operands of each pass independently
    if bf16 and k == 7 and Cout == 4 and Cin == 64:
        tol = 5e-3                        # the head's input gradient also rounds its padded-domain terms to 16 bits (stem kernel)
    scale = max(abs(a), (y.double().abs() * dy.double().abs()).sum().item() * 1e-3)
    assert abs(a - b) / scale < tol and abs(a - c) / scale < tol, (a, b, c)


def test_conv_linearity_full_size(dev):
    from mmhand_amd import ops
    B, H, W, C = 32, 64, 64, 256
    x1, x2 = _rand((B, H, W, C), 1, dev), _rand((B, H, W, C), 2, dev)
    w = _rand((3, 3, C, C), 3, dev, 0.05)
    y12 = ops.raw_conv_fprop(0.5 * x1 - 2.0 * x2, w, None, 1, 1, True, 0)
    y1 = ops.raw_conv_fprop(x1, w, None, 1, 1, True, 0)
    y2 = ops.raw_conv_fprop(x2, w, None, 1, 1, True, 0)
    ref = 0.5 * y1 - 2.0 * y2
    assert ((y12 - ref).abs().sum() / ref.abs().sum()).item() < 1e-5


@pytest.mark.parametrize("mode", ["instance", "batch"])
def test_norm_moments_full_size(mode, dev):
    from mmhand_amd import ops
    x = _rand((32, 64, 64, 512), 1, dev, 3.0) + 5.0
    out = ops.NormActFn.apply(x, None, None, None, None, None, mode, False, 0.0, 0, None, None)
    dims = (1, 2) if mode == "instance" else (0, 1, 2)
    m = out.double().mean(dims)
    v = out.double().var(dims, unbiased=False)
    assert m.abs().max().item() < 1e-4 and (v - 1).abs().max().item() < 1e-3


def test_dropout_rate_full_size(dev):
    from mmhand_amd import ops
    x = torch.ones((32, 64, 64, 512), device=dev)
    one = torch.ones((1, 512), device=dev); zero = torch.zeros((1, 512), device=dev)
    y = ops.raw_scale_shift_act(x, one, zero, None, True, 0.5, 0xC0FFEE, None)
    keep = (y > 0).double().mean().item()
    assert abs(keep - 0.5) < 2e-4 and y.max().item() == 2.0


def test_adam_properties_full_size(dev):
    from mmhand_amd import ops
    n = 71_272_836                                  # Generator-sized flat buffer (multiple of 4)
    p = _rand((n,), 1, dev)
    p0 = p.clone()
    m = torch.zeros_like(p); v = torch.zeros_like(p)
    ops.adam_step(p, torch.zeros_like(p), m, v, 2e-4, 0.5, 0.999, 1e-8, 1)
    assert torch.equal(p, p0)
    g = torch.full_like(p, 0.37)
    ops.adam_step(p, g, m, v, 2e-4, 0.5, 0.999, 1e-8, 2)
    # m = 0.5*0.37, v = 0.001*0.37^2; bias corrections 1-0.25, 1-0.999^2
    step = 2e-4 / (1 - 0.5 ** 2) * (0.5 * 0.37) / ((0.001 * 0.37 ** 2 / (1 - 0.999 ** 2)) ** 0.5 + 1e-8)
    assert ((p0 - p) - step).abs().max().item() < 1e-7


def test_sketch_inputs_equal_the_fixture_generator():
    """bench.sketch_inputs (no oracle import in bench.py) IS the batch tests/golden/make_golden.py make_fullsize ran on"""
    import bench
    from oracle import mmhand_ref as O
    a, b = bench.sketch_inputs(2, 256, 256, 49), O.synthetic_batch(2, 256, 256, 49)
    assert all(torch.equal(a[k], b[k]) for k in b)


@pytest.mark.parametrize("mode", ["off1", "off", "bwd", "all"])
def test_fullsize_generator_gradients_vs_fp64_sketch(mode, dev):
    """the full-size Generator (ngf 64, 9 PATBlocks, 256x256, B=2, InstanceNorm, dropout off) against FLOAT64
    TRUTH from the reference's own module (tests/golden/fullsize_grad_sketch.npz: per parameter tensor the float64 gradient
    at 1024 seeded positions; models/Generator.py:269-313 in double precision on identical weights and inputs).

    What the fixture says about fp32 on this network: PyTorch's own fp32 CPU run (the same reference module in float32) is
    7.7e-4 (median) / 1.63e-3 (max) from float64 with an output 1.2e-6 away - 9 of 85 tensors above 1e-3.  No fp32
    implementation holds 1e-3 on every tensor here: a forward that differs in the 6th digit flips ReLU masks of
    pre-activations within rounding of zero and the backward pass amplifies that layer by layer.  The distance
    scales with the forward's own distance: the direct kernels' output is 2.9e-6 from float64 (k-ordered fp32 MFMA chains
    4608 deep, fp32 statistics; PyTorch on the CPU accumulates its norm statistics in double) and their gradients a median
    2.2e-3.  With TWO-LEVEL summation in the direct fprop (a fresh chain per 32-deep k-step; ops.set_winograd_mode's default for
    "off" and "bwd"; the 3x3 / stride-2 convs and the 7x7 stems) the output is 1.1e-6 from float64 and the gradients a
    median 9.5e-4, max 1.4e-3 - PyTorch's neighbourhood (its max is 1.6e-3).
    Bars per tensor = what round 6 measured + a third:
      off1 (direct kernels, one level: `direct_path`)  median <= 2.9e-3, max <= 4.1e-3, output <= 5e-6  (measured 2.17e-3 / 3.06e-3 / 2.9e-6)
      off  (direct kernels, two-level fprop)           median <= 1.3e-3, max <= 1.9e-3, output <= 1.6e-6  (measured 9.5e-4 / 1.43e-3 / 1.13e-6)
      bwd  (`--fp32_exact_grads`, `hybrid_path`)       the same bars (the same two-level forward)
      all  (Winograd F(6x6,3x3), the headline)        median <= 4e-3, max <= 5.6e-3, output <= 1e-5  (measured 3.05e-3 / 4.30e-3 / 6.5e-6)
    """
    import os
    import statistics
    import numpy as np
    import bench
    from mmhand_amd import ops
    from mmhand_amd.networks import Generator, logical_grads
    fix = np.load(os.path.join(os.path.dirname(__file__), "golden", "fullsize_grad_sketch.npz"))
    b = {k: v.to(dev) for k, v in bench.sketch_inputs(2, 256, 256, 49).items()}
    g_in = [b["H1"], torch.cat((b["P1"], b["P2"]), 1), torch.cat((b["D1"], b["D2"]), 1)]
    probe = torch.randn(2, 3, 256, 256, generator=torch.Generator().manual_seed(3)).to(dev)
    old = "off" if not ops.USE_WINOGRAD else "all" if ops.WINOGRAD_FPROP else "bwd"
    try:
        if mode == "off1":
            ops.set_winograd_mode("off", direct_levels=1)
        else:
            ops.set_winograd_mode(mode)
        net = Generator([3, 42, 6], 3, 64, "instance", False, 9).init_weights("normal", 49).to(dev).train()
        net.flatten_parameters()
        out = net(g_in)
        (out * probe).sum().backward()
        errs, oerr = bench.fp64_sketch_distance(fix, logical_grads(net), out.detach().contiguous())
    finally:
        ops.set_winograd_mode(old)
    cond = {k: float(fix["cond_sampled/" + k]) for k in errs}
    v = sorted(errs.values())
    worst = max(errs, key=errs.get)
    print(f"\n[{mode}] output {oerr:.2e}; gradients vs fp64: median {statistics.median(v):.2e} max {v[-1]:.2e} ({worst}); "
          f"PyTorch fp32: median {statistics.median(cond.values()):.2e} max {max(cond.values()):.2e}; "
          f"{sum(e > 1e-3 for e in v)} of {len(v)} above 1e-3")
    assert len(v) == 85
    if mode == "all":
        assert oerr < 1e-5 and v[-1] < 5.6e-3 and statistics.median(v) < 4e-3, (oerr, v[-1], statistics.median(v))
    elif mode == "off1":
        assert oerr < 5e-6 and v[-1] < 4.1e-3 and statistics.median(v) < 2.9e-3, (oerr, v[-1], statistics.median(v))
    else:
        assert oerr < 1.6e-6 and v[-1] < 1.9e-3 and statistics.median(v) < 1.3e-3, (oerr, v[-1], statistics.median(v))
