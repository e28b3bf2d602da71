"""Norm-apply fused into the neighbouring convolutions (fp32 Winograd F(6x6,3x3) stack;
models/Generator.py:66-77, models/Discriminator.py:29-34: conv -> norm -> ReLU -> Dropout -> pad -> conv).

The fused transforms evaluate the same expressions as the stand-alone kernels, so with the same dropout
decisions (an injected mask) the forward is compared bit for bit: the input transform with the norm-apply
inside against transform(apply(x)), the decide-again backward kernels against the keep-bit kernels.  The
backward transform with the norm backward inside is a different instantiation of the same arithmetic than
transform(apply_rc(...)): the compiler rounds a few of its intermediate products differently, the results
agree to a few ulps of the largest element (checked at 2e-6 of it); whole two-conv blocks with the fusion
on are compared with the fusion off: forward bit for bit, gradients at 1e-5 relative L1."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _same_but_for_ulps(a, b):
    return float((a - b).abs().max()) <= 2e-6 * float(b.abs().max())


def _mk(shape, seed, dev, scale=1.0, shift=0.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale + shift).to(dev)


def _mask(shape, seed, dev):
    return (torch.rand(shape, generator=torch.Generator().manual_seed(seed)) >= 0.5).to(torch.uint8).to(dev)


def test_dropout_bits_pack_mask_and_rate(dev):
    from mmhand_amd import ops
    shape = (2, 5, 7, 64)
    m = _mask(shape, 1, dev)
    bits = ops.raw_dropout_bits(shape, 0.5, 0, m, dev)
    ref = (m.reshape(-1, 8).to(torch.int32) << torch.arange(8, device=dev, dtype=torch.int32)).sum(1).to(torch.uint8)
    assert torch.equal(bits, ref)
    big = (8, 64, 64, 256)
    for p in (0.5, 0.25):
        b = ops.raw_dropout_bits(big, p, 12345, None, dev)
        kept = sum(int(((b >> e) & 1).sum()) for e in range(8)) / (b.numel() * 8)
        assert abs(kept - (1 - p)) < 2e-3, kept
    assert not torch.equal(ops.raw_dropout_bits(big, 0.5, 1, None, dev), ops.raw_dropout_bits(big, 0.5, 2, None, dev))


@pytest.mark.parametrize("shape", [(2, 5, 7, 64), (1, 3, 70, 16), (2, 2, 33, 8)])
def test_dropout_row_words_hold_the_same_decisions(shape, dev):
    """mmh_dropout_bits_rows: [B*H][ceil(W/32)][C] words, bit k of word j = element (row, 32 j + k, c)"""
    from mmhand_amd import ops
    B, H, W, C = shape
    m = _mask(shape, 3, dev)
    _, rows = ops.raw_dropout_bits(shape, 0.5, 0, m, dev, rows=True)
    assert rows.shape == (B * H, (W + 31) // 32, C) and rows.dtype == torch.int32
    # the two stand-alone calls give the same arrays as the single launch, with the hash as with a mask
    for mk, p_ in ((m, 0.5), (None, 0.5), (None, 0.3)):
        b1, r1 = ops.raw_dropout_bits(shape, p_, 77, mk, dev, rows=True)
        b2 = ops.raw_dropout_bits(shape, p_, 77, mk, dev)
        r2 = torch.empty_like(r1)
        ops.L.call("mmh_dropout_bits_rows", ops._ptr(b2), B * H, W, C, ops._ptr(r2), ops._stream())
        assert torch.equal(b1, b2) and torch.equal(r1, r2)
    mm = m.reshape(B * H, W, C).to(torch.int64)
    for j in range(rows.shape[1]):
        seg = mm[:, 32 * j:32 * j + 32]                                      # [rows, <=32, C]
        w = (seg << torch.arange(seg.shape[1], device=dev, dtype=torch.int64)[None, :, None]).sum(1)
        assert torch.equal(rows[:, j].to(torch.int64) & 0xffffffff, w)


@pytest.mark.parametrize("groups_is_b", [True, False])
@pytest.mark.parametrize("relu,drop", [(True, True), (True, False), (False, False)])
@pytest.mark.parametrize("shape,reflect", [((2, 24, 18, 128), True), ((3, 13, 20, 64), False), ((1, 12, 12, 256), True),
                                           ((1, 14, 70, 64), True), ((2, 12, 41, 64), False),      # rows of 2-3 dropout words
                                           ((32, 64, 64, 256), True)])                              # the benchmark's shape
def test_input_transform_with_norm_apply_inside(groups_is_b, relu, drop, shape, reflect, dev):
    from mmhand_amd import lib as L, ops
    B, H, W, C = shape
    groups = B if groups_is_b else 1
    x = _mk(shape, 1, dev, 2.0, 0.5)
    scale = _mk((groups, C), 2, dev, 0.3, 1.0)
    shift = _mk((groups, C), 3, dev, 0.5)
    m = _mask(shape, 4, dev) if drop else None
    p = 0.5 if drop else 0.0
    a = ops.raw_scale_shift_act(x, scale, shift, None, relu, p, 0, m)
    tiles = B * (-(-H // 6)) * (-(-W // 6))
    V_ref = torch.empty((64, tiles, C), device=dev)
    L.call("mmh_wino_input", ops._ptr(a), B, H, W, C, int(reflect), 6, L.F32, ops._ptr(V_ref), ops._stream())
    drows = ops.raw_dropout_bits(shape, p, 0, m, dev, rows=True)[1] if drop else None
    V = torch.full((64, tiles, C), float("nan"), device=dev)
    L.call("mmh_wino_input_normact", ops._ptr(x), B, H, W, C, int(reflect), ops._ptr(V), ops._ptr(scale), ops._ptr(shift),
           groups, int(relu), p, ops._ptr(drows), ops._stream())
    assert torch.equal(V, V_ref)


@pytest.mark.parametrize("groups_is_b", [True, False])
@pytest.mark.parametrize("relu,drop", [(True, True), (True, False), (False, False)])
@pytest.mark.parametrize("shape", [(3, 14, 12, 128), (1, 13, 75, 64), (32, 64, 64, 256)])
def test_decide_again_backward_kernels_match_keep_bit_kernels(groups_is_b, relu, drop, shape, dev):
    from mmhand_amd import lib as L, ops
    B, H, W, C = shape
    groups = B if groups_is_b else 1
    rows = (B // groups) * H * W
    x = _mk(shape, 1, dev, 2.0, 0.5)
    g = _mk(shape, 5, dev)
    gamma = None if groups_is_b else _mk((C,), 6, dev, 0.1, 1.0)
    beta = None if groups_is_b else _mk((C,), 7, dev, 0.1)
    m = _mask(shape, 4, dev) if drop else None
    p = 0.5 if drop else 0.0
    mean, m2, _ = ops.raw_norm_stats(x, groups)
    scale, shift, invstd = ops.raw_norm_finalize(mean, m2, rows, gamma, beta, None, None)
    masked = 2 if (relu or drop) else 0
    kb = ops.raw_scale_shift_act(x, scale, shift, None, relu, p, 0, m, keep_bits=True)[1] if masked else None
    lib = L.load()
    ws = torch.empty(lib.mmh_norm_bwd_ws_bytes(groups, rows, C) // 4 + 4, device=dev)
    P, st = ops._ptr, ops._stream
    s1 = torch.empty((groups, C), device=dev); s2 = torch.empty_like(s1)
    L.call("mmh_norm_bwd_reduce", P(g), P(kb), P(x), P(mean), P(invstd), groups, rows, C, masked, p, P(s1), P(s2), P(ws),
           ws.numel() * 4, L.F32, L.F32, st())
    dx = torch.empty_like(x)
    L.call("mmh_norm_bwd_apply", P(g), P(kb), P(x), P(mean), P(invstd), P(gamma), P(s1), P(s2), float(rows), groups, rows, C,
           masked, p, P(dx), L.F32, L.F32, L.F32, st())
    dbits, drows = ops.raw_dropout_bits(shape, p, 0, m, dev, rows=True) if drop else (None, None)
    r1 = torch.empty_like(s1); r2 = torch.empty_like(s1)
    L.call("mmh_norm_bwd_reduce_rc", P(g), P(x), P(mean), P(invstd), P(scale), P(shift), P(dbits), groups, rows, C, int(relu),
           p, P(r1), P(r2), P(ws), ws.numel() * 4, st())
    assert torch.equal(r1, s1) and torch.equal(r2, s2)
    rdx = torch.empty_like(x)
    L.call("mmh_norm_bwd_apply_rc", P(g), P(x), P(mean), P(invstd), P(gamma), P(s1), P(s2), P(scale), P(shift), P(dbits),
           float(rows), groups, rows, C, int(relu), p, P(rdx), st())
    assert torch.equal(rdx, dx)
    # ... and the backward transform with the same arithmetic inside
    for fold in (0, 1) if ops._fold_same_grid(H, W) else (0,):
        tiles = B * (-(-H // 6)) * (-(-W // 6))
        V_ref = torch.empty((64, tiles, C), device=dev); Y_ref = torch.empty_like(V_ref)
        L.call("mmh_wino_input_dy", P(dx), B, H, W, C, 6, L.F32, P(V_ref), P(Y_ref), fold, st())
        V = torch.full_like(V_ref, float("nan")); Y = torch.full_like(V_ref, float("nan"))
        L.call("mmh_wino_input_dy_normbwd", P(g), P(x), B, H, W, C, P(V), P(Y), fold, P(mean), P(invstd), P(gamma), P(s1),
               P(s2), float(rows), P(scale), P(shift), P(drows), groups, int(relu), p, st())
        assert _same_but_for_ulps(V, V_ref) and _same_but_for_ulps(Y, Y_ref), fold


@pytest.mark.parametrize("norm", ["instance", "batch"])
@pytest.mark.parametrize("last_norm,frozen,shape", [(True, False, (2, 22, 16, 128)), (False, False, (2, 22, 16, 128)),
                                                    (True, True, (2, 22, 16, 128)),
                                                    (True, False, (1, 128, 130, 128))])   # configs[4]-sized maps, 5 row words
def test_two_conv_block_fused_vs_unfused(norm, last_norm, frozen, shape, dev, monkeypatch):
    """RP-conv-norm-ReLU-dropout-RP-conv(-norm): fusion on against fusion off, bit for bit; frozen: weights
    without gradients (the discriminators inside the generator step) - there the deferred gradient of the
    norm's input is materialised (mmh_norm_bwd_apply_rc) instead of computed inside the backward transform."""
    from mmhand_amd import networks, ops
    torch.manual_seed(0)

    class Net(networks._Net):
        def __init__(self):
            super().__init__(norm, True)
            self.blk = networks.Bag()
            self._conv(self.blk, 1, 128, 128, 3)
            self._normp(self.blk, 2, 128)
            self._conv(self.blk, 6, 128, 128, 3)
            self._normp(self.blk, 7, 128)

    net = Net().init_weights("normal", seed=3).to(dev)
    net.flatten_parameters()
    net.train()
    x = _mk(shape, 1, dev)
    gy = _mk(shape, 2, dev)
    mask = _mask(shape, 5, dev)
    if frozen:
        for p in net.parameters():
            p.requires_grad_(False)
    res = {}
    for fuse in (False, True):
        monkeypatch.setattr(ops, "USE_NORM_FUSION", fuse)
        net._mask_src = {"site": mask}
        xin = x.clone().requires_grad_(True)
        net.zero_grad()
        want = net._norm_fusion(net.blk[1], net.blk[6], xin)
        assert want == (2 if fuse else 0)      # BatchNorm convs have no bias, InstanceNorm ones a null bias gradient
        L_calls = []
        real = ops.L.call

        def spy(name, *a):
            L_calls.append(name)
            return real(name, *a)
        monkeypatch.setattr(ops.L, "call", spy)
        y = net.two_conv_block(net.blk, xin, "site", last_norm)
        y.backward(gy)
        monkeypatch.setattr(ops.L, "call", real)
        if fuse:
            assert "mmh_wino_input_normact" in L_calls and "mmh_norm_bwd_reduce_rc" in L_calls
            # the backward apply pass of BOTH norms (the last one is written forward as usual) inside the producing
            # conv's backward transform; materialised where that conv's weights are frozen
            n_sites = 2 if last_norm else 1
            assert L_calls.count("mmh_wino_input_dy_normbwd") == (0 if frozen else n_sites)
            assert L_calls.count("mmh_norm_bwd_apply_rc") == (n_sites if frozen else 0)
            assert "mmh_norm_bwd_apply" not in L_calls
            assert L_calls.count("mmh_scale_shift_act") == (1 if last_norm else 0)
        else:
            assert "mmh_wino_input_normact" not in L_calls
        res[fuse] = (y.detach().clone(), xin.grad.clone(), None if frozen else net.flat_grad.clone())
        assert not ops._nb_defer and not ops._lp_grads
    assert torch.equal(res[True][0], res[False][0])         # forward: bit for bit
    for a, b in zip(res[True][1:], res[False][1:]):
        if a is not None:
            # (few-ulp differences of the two fused backward transforms, carried through two norms: 4e-6 measured)
            assert float((a - b).abs().sum() / b.abs().sum()) < 1e-5, float((a - b).abs().max())


def test_deferred_gradient_on_a_foreign_edge_fails_loudly(dev):
    from mmhand_amd import ops
    g = ops.lp_proxy((1, 12, 12, 64), dev)
    with pytest.raises(RuntimeError, match="NormBwdDefer"):
        ops.norm_bwd_defer_in(g, "test")


def test_stats_merge_and_finalize_in_one_launch(dev):
    """InstanceNorm after a Winograd F(6x6,3x3) conv: the per-tile partials of the output transform merged and
    finalised by one kernel - the same fp32 mean / scale / shift / invstd as mmh_norm_stats_merge + mmh_norm_finalize,
    and the statistics of the plane itself."""
    from mmhand_amd import lib as L, ops
    x = _mk((2, 20, 16, 128), 1, dev)
    w = _mk((3, 3, 128, 128), 2, dev, 0.05)
    y = ops.raw_conv_fprop_wino(x, w, None, True, 0, 6)
    stats = ops._pending_stats[y.data_ptr()][0].clone()
    B, H, W, C = y.shape
    fast = ops.raw_norm_stats_finalize_pending(y, B)
    assert fast is not None and not ops._pending_stats
    mean, scale, shift, invstd, rows = fast
    P, st = ops._ptr, ops._stream
    m_ref = torch.empty((B, C), device=dev); m2_ref = torch.empty_like(m_ref)
    L.call("mmh_norm_stats_merge", P(stats), B, stats.shape[1], C, P(m_ref), P(m2_ref), st())
    sc_ref, sf_ref, is_ref = ops.raw_norm_finalize(m_ref, m2_ref, rows, None, None, None, None)
    assert torch.equal(mean, m_ref) and torch.equal(scale, sc_ref) and torch.equal(shift, sf_ref) and torch.equal(invstd, is_ref)
    yd = y.double().reshape(B, H * W, C)
    assert torch.allclose(mean.double(), yd.mean(1), atol=1e-6)
    assert torch.allclose(invstd.double(), 1.0 / torch.sqrt(yd.var(1, unbiased=False) + ops.EPS), rtol=1e-5)
    assert ops.raw_norm_stats_finalize_pending(y, B) is None        # consumed


@pytest.mark.parametrize("Cin,Cout,k,stride,pad,reflect,H", [(8, 64, 7, 1, 3, True, 16), (64, 128, 3, 2, 1, False, 32),
                                                            (128, 256, 3, 2, 1, False, 32), (4, 64, 3, 1, 1, False, 16)])
def test_direct_fprop_leaves_output_statistics(Cin, Cout, k, stride, pad, reflect, H, dev, monkeypatch):
    """The direct fp32 fprop (stems, stride-2 convs) writes per-tile count / mean / M2 of its output columns from its
    epilogue; the InstanceNorm behind it merges them instead of reading y: same normalised output as from a pass over y."""
    from mmhand_amd import lib as L, ops
    B = 2
    x = _mk((B, H, H, Cin), 1, dev)
    w = _mk((k, k, Cin, Cout), 2, dev, 0.05)
    bias = _mk((Cout,), 3, dev)
    calls = []
    real = L.call
    monkeypatch.setattr(L, "call", lambda n, *a: (calls.append(n), real(n, *a))[1])
    y = ops.raw_conv_fprop(x, w, bias, stride, pad, reflect, want_stats=True)
    assert "mmh_conv2d_fprop_stats" in calls and y.data_ptr() in ops._pending_stats
    y_plain = ops.raw_conv_fprop(x, w, bias, stride, pad, reflect)
    assert torch.equal(y, y_plain)
    ops._pending_stats.clear()
    ops._pending_stats[y.data_ptr()] = None      # placeholder, replaced below
    y2 = ops.raw_conv_fprop(x, w, bias, stride, pad, reflect, want_stats=True)
    fast = ops.raw_norm_stats_finalize_pending(y2, B)
    assert fast is not None
    mean, scale, shift, invstd, rows = fast
    yd = y2.double().reshape(B, -1, Cout)
    assert rows == yd.shape[1]
    assert torch.allclose(mean.double(), yd.mean(1), atol=2e-6)
    assert torch.allclose(invstd.double(), 1.0 / torch.sqrt(yd.var(1, unbiased=False) + ops.EPS), rtol=2e-5)
    # through the autograd function: statistics from the conv epilogue == statistics from a pass over y
    outs = []
    for fuse in (True, False):
        monkeypatch.setattr(ops, "FUSE_NORM_STATS", fuse)
        yy = ops.raw_conv_fprop(x, w, bias, stride, pad, reflect, want_stats=True)
        outs.append(ops.NormActFn.apply(yy, None, None, None, None, None, "instance", True, 0.0, 0, None, None))
    assert float((outs[0] - outs[1]).abs().max()) < 2e-5
