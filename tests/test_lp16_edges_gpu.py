"""16-bit activation / gradient storage between the conv_lp16 convolutions and their norm / gate
neighbours (--opt_level O1/O2: what apex O1 keeps in fp16 - scripts/mm-train-ratio.sh:7-11).

The HBM-bound kernels take an element type per tensor; arithmetic stays fp32.  So a kernel fed the
16-bit tensor must give exactly what the fp32 kernel gives on the same values widened to fp32, and
a 16-bit output must be the round-to-nearest-even of the fp32 output.  The autograd plumbing (fp32
zero-stride proxies on the edges, 16-bit tensors beside them, gradients through ops.lp_grad_out /
lp_grad_in) is checked against the same blocks run with fp32 edges."""
import pytest
import torch

from oracle import ops_ref as R

pytestmark = pytest.mark.gpu
LPS = [(True, torch.bfloat16), (2, torch.float16)]


class _Tap(torch.autograd.Function):
    """Stands for the node on the far side of a 16-bit edge (a leaf cannot: AccumulateGrad copies the
    proxy): catches the 16-bit gradient its backward is handed."""
    got = []

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        from mmhand_amd import ops
        _Tap.got.append(ops.lp_grad_in(g, "tap"))
        return g


def _mk(shape, seed, dev, scale=1.0, shift=0.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale + shift).to(dev)


@pytest.mark.parametrize("lp,td", LPS)
@pytest.mark.parametrize("mode", ["batch", "instance"])
@pytest.mark.parametrize("relu,drop", [(False, False), (True, True)])
def test_norm_kernels_16bit_io_match_fp32_kernels(lp, td, mode, relu, drop, dev):
    from mmhand_amd import ops
    shape = (3, 12, 10, 64)
    x16 = _mk(shape, 1, dev, 2.0, 1.0).to(td)
    g16 = _mk(shape, 2, dev).to(td)
    mask = (torch.rand(shape, generator=torch.Generator().manual_seed(5)) >= 0.5).to(torch.uint8).to(dev) if drop else None
    C = shape[3]
    gamma = _mk((C,), 3, dev, 0.1, 1.0) if mode == "batch" else None
    beta = _mk((C,), 4, dev, 0.1) if mode == "batch" else None

    def run(x_in, x16_in, out_lp, g_fp32):
        ga = None if gamma is None else gamma.clone().requires_grad_(True)
        be = None if beta is None else beta.clone().requires_grad_(True)
        rm = torch.zeros(C, device=dev) if mode == "batch" else None
        rv = torch.ones(C, device=dev) if mode == "batch" else None
        xin = x_in.requires_grad_(True)
        res = ops.NormActFn.apply(_Tap.apply(xin) if x16_in is not None else xin, ga, be, None, rm, rv, mode, relu,
                                  0.5 if drop else 0.0, 0, mask, None, out_lp, x16_in)
        if out_lp:
            proxy, out = res
            proxy.backward(ops.lp_grad_out(g16))
        else:
            out = res
            out.backward(g_fp32)
        return out, xin.grad, ga, be

    # fp32 kernels on the widened values
    out_r, dx_r, ga_r, be_r = run(x16.float(), None, 0, g16.float())
    # 16-bit x, 16-bit out, 16-bit gradient in, 16-bit gradient out
    proxy_x = ops.lp_proxy(shape, dev)
    _Tap.got.clear()
    out_t, _, ga_t, be_t = run(proxy_x, x16, lp, None)
    dx_t, = _Tap.got
    assert out_t.dtype == td and dx_t.dtype == td
    assert torch.equal(out_t, out_r.to(td))
    assert torch.equal(dx_t, dx_r.to(td))
    if mode == "batch":
        assert torch.equal(ga_t.grad, ga_r.grad) and torch.equal(be_t.grad, be_r.grad)
    assert not ops._lp_grads


@pytest.mark.parametrize("lp,td", LPS)
def test_gate_16bit_io_matches_fp32_kernel(lp, td, dev):
    from mmhand_amd import ops
    shape = (2, 9, 7, 64)
    x1 = _mk(shape, 1, dev).requires_grad_(True)
    s1 = _mk(shape, 2, dev).requires_grad_(True)
    s2_16, s3_16 = _mk(shape, 3, dev).to(td), _mk(shape, 4, dev).to(td)
    cshape = shape[:3] + (2 * shape[3],)
    g_out = _mk(shape, 5, dev)
    g2_16, g3_16 = _mk(cshape, 6, dev).to(td), _mk(cshape, 7, dev).to(td)
    # reference: fp32 tensors holding the same values
    s2 = s2_16.float().requires_grad_(True)
    s3 = s3_16.float().requires_grad_(True)
    out_r, x2_r, x3_r = ops.GateFn.apply(x1, s1, s2, s3, True)
    torch.autograd.backward([out_r, x2_r, x3_r], [g_out, g2_16.float(), g3_16.float()])
    ref = [t.grad.clone() for t in (x1, s1, s2, s3)]
    x1.grad = s1.grad = None
    p2, p3 = ops.lp_proxy(shape, dev).requires_grad_(True), ops.lp_proxy(shape, dev).requires_grad_(True)
    _Tap.got.clear()
    out_t, q2, q3, c2, c3 = ops.GateFn.apply(x1, s1, _Tap.apply(p2), _Tap.apply(p3), True, lp, s2_16, s3_16)
    assert c2.dtype == td and torch.equal(out_t, out_r)
    assert torch.equal(c2, x2_r.to(td)) and torch.equal(c3, x3_r.to(td))
    torch.autograd.backward([out_t, q2, q3], [g_out, ops.lp_grad_out(g2_16), ops.lp_grad_out(g3_16)])
    (gs2, gs3) = _Tap.got if _Tap.got[0].data_ptr() != _Tap.got[1].data_ptr() else (None, None)
    if not torch.equal(gs2, ref[2].to(td)):     # the two taps run in either order
        gs2, gs3 = gs3, gs2
    assert torch.equal(x1.grad, ref[0]) and torch.equal(s1.grad, ref[1])
    assert gs2.dtype == td and torch.equal(gs2, ref[2].to(td)) and torch.equal(gs3, ref[3].to(td))
    assert not ops._lp_grads


@pytest.mark.parametrize("lp,td", LPS)
def test_colsum_16bit(lp, td, dev):
    from mmhand_amd import ops
    x16 = _mk((5000, 256), 1, dev).to(td)
    assert torch.equal(ops.raw_colsum(5000, 256, x16), ops.raw_colsum(5000, 256, x16.float()))
    ref = x16.double().sum(0)
    assert float((ops.raw_colsum(5000, 256, x16).double() - ref).abs().max()) < 1e-3


@pytest.mark.parametrize("lp,td", LPS)
@pytest.mark.parametrize("shape", [(2, 16, 16, 256, 256), (1, 13, 10, 512, 256)])
def test_dgrad_reflect_16bit_gradients(lp, td, shape, dev):
    """dgrad of the reflect-pad 3x3 conv from a 16-bit dy: the eight border GEMMs gather the 16-bit
    tensor, border_add works on a 16-bit dx; vs the fp64 adjoint."""
    from mmhand_amd import ops
    B, H, W, Ci, Co = shape
    w = _mk((3, 3, Ci, Co), 1, dev, 0.05)
    dy16 = _mk((B, H, W, Co), 2, dev).to(td)
    dx32 = ops.raw_conv_dgrad(None, w, (B, H, W, Ci), 1, 1, True, lp, dy16=dy16)
    dx16 = ops.raw_conv_dgrad(None, w, (B, H, W, Ci), 1, 1, True, lp, dy16=dy16, out16=True)
    assert dx32.dtype == torch.float32 and dx16.dtype == td
    # the weights the kernel multiplies with are the 16-bit roundings of w
    w16 = w.to(td).double().cpu()
    ref = R.conv2d_grads(torch.zeros(B, H, W, Ci), w16, None, dy16.double().cpu(), 1, 1, True)[1]
    tol = 1e-2 if td == torch.bfloat16 else 2e-3
    assert R.rel_l1(dx32, ref) < tol
    assert R.rel_l1(dx16.float(), ref) < tol
    # ring-adjacent pixels are where the border terms land: check them separately
    ring = torch.zeros(H, W, dtype=torch.bool)
    ring[1], ring[H - 2], ring[:, 1], ring[:, W - 2] = True, True, True, True
    assert R.rel_l1(dx16.float().cpu()[:, ring], ref[:, ring]) < tol


@pytest.mark.parametrize("lp", [True, 2])
@pytest.mark.parametrize("last_norm", [True, False])
def test_two_conv_block_16bit_edges_vs_fp32_edges(lp, last_norm, dev, monkeypatch):
    """RP-conv-norm-ReLU-dropout-RP-conv(-norm) with 16-bit edges everywhere against the same block
    with fp32 tensors between the 16-bit convolutions."""
    from mmhand_amd import networks, ops
    torch.manual_seed(0)

    class Net(networks._Net):
        def __init__(self):
            super().__init__("instance", True)
            self.blk = networks.Bag()
            self._conv(self.blk, 1, 256, 256, 3)
            self._normp(self.blk, 2, 256)
            self._conv(self.blk, 6, 256, 256, 3)
            self._normp(self.blk, 7, 256)

    net = Net().init_weights("normal", seed=3).to(dev)
    net.flatten_parameters()
    net.bf16 = lp
    net.train()
    x = _mk((2, 20, 16, 256), 1, dev)
    gy = _mk((2, 20, 16, 256), 2, dev)
    mask = (torch.rand(x.shape, generator=torch.Generator().manual_seed(5)) >= 0.5).to(torch.uint8).to(dev)
    res = {}
    for edges in (False, True):
        monkeypatch.setattr(ops, "USE_LP16_EDGES", edges)
        net._mask_src = {"site": mask}
        xin = x.clone().requires_grad_(True)
        net.zero_grad()
        y = net.two_conv_block(net.blk, xin, "site", last_norm)
        if isinstance(y, tuple):
            assert edges and not last_norm
            y[0].backward(ops.lp_grad_out(gy.to(y[1].dtype)))
            y = y[1].float()
        else:
            assert not (edges and not last_norm)
            y.backward(gy)
        res[edges] = (y.detach().clone(), xin.grad.clone(), net.flat_grad.clone())
        assert not ops._lp_grads
    tol = 2e-2 if lp is True else 4e-3
    for a, b in zip(res[True], res[False]):
        assert R.rel_l1(a, b) < tol, R.rel_l1(a, b)


@pytest.mark.parametrize("lp", [True, 2])
@pytest.mark.parametrize("norm,drop", [("instance", True), ("batch", False)])
def test_first_norm_backward_sums_from_the_dgrad_epilogue(lp, norm, drop, dev, monkeypatch):
    """Two-conv block with 16-bit edges: the second conv's input gradient is the first norm's output gradient, and its dgrad
    takes that norm's backward sums in its epilogue (ops.USE_NBR, mmh_conv3x3_lp16_dgrad_nbr) - no mmh_norm_bwd_reduce for
    that norm.  Against the same block with the reduce pass: output identical, the 16-bit gradient handed to the norm
    identical, every gradient equal to summation-order rounding."""
    from mmhand_amd import lib, networks, ops
    torch.manual_seed(0)

    class Net(networks._Net):
        def __init__(self):
            super().__init__(norm, drop)
            self.blk = networks.Bag()
            self._conv(self.blk, 1, 256, 256, 3)
            self._normp(self.blk, 2, 256)
            self._conv(self.blk, 6 if drop else 5, 256, 256, 3)
            self._normp(self.blk, 7 if drop else 6, 256)

    net = Net().init_weights("normal", seed=3).to(dev)
    net.flatten_parameters()
    net.bf16 = lp
    net.train()
    x = _mk((2, 32, 48, 256), 1, dev)
    gy = _mk((2, 32, 48, 256), 2, dev)
    mask = (torch.rand(x.shape, generator=torch.Generator().manual_seed(5)) >= 0.5).to(torch.uint8).to(dev)
    res, calls = {}, {}
    for on in (True, False):
        monkeypatch.setattr(ops, "USE_NBR", on)
        net._mask_src = {"site": mask}
        xin = x.clone().requires_grad_(True)
        net.zero_grad()
        seen = calls.setdefault(on, {})
        orig = lib.call
        def spy(name, *a, _seen=seen):
            _seen[name] = _seen.get(name, 0) + 1
            return orig(name, *a)
        lib.call = spy
        try:
            y = net.two_conv_block(net.blk, xin, "site", True)
            y.backward(gy)
        finally:
            lib.call = orig
        res[on] = (y.detach().clone(), xin.grad.clone(), net.flat_grad.clone())
        assert not ops._lp_grads and not ops._nbr_sites
    assert calls[True].get("mmh_conv3x3_lp16_dgrad_nbr") == 1 and "mmh_conv3x3_lp16_dgrad_nbr" not in calls[False]
    assert calls[False]["mmh_norm_bwd_reduce"] == calls[True].get("mmh_norm_bwd_reduce", 0) + 1
    assert torch.equal(res[True][0], res[False][0])
    for a, b in zip(res[True][1:], res[False][1:]):
        assert R.rel_l1(a, b) < 2e-4, R.rel_l1(a, b)


def test_lost_16bit_gradient_fails_loudly(dev):
    from mmhand_amd import ops
    g = ops.lp_proxy((1, 2, 2, 4), dev)
    with pytest.raises(RuntimeError, match="16-bit"):
        ops.lp_grad_in(g, "test")
