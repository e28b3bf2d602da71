"""Inference-only Generator (the reference's aug.py:26-58 path; BASELINE.json configs[3]).

eval-mode BatchNorm uses running statistics, so it is folded into the producing conv once
(w' = w*gamma/sqrt(var+eps), b' = beta - mean*that) and ReLU/Tanh run in the conv epilogue: the
whole forward becomes 66 conv launches + 9 gate launches + 3 pack launches with no normalisation
pass at all.  The static launch sequence is captured in a hipGraph (torch.cuda.CUDAGraph) and
replayed per batch.  InstanceNorm generators cannot be folded (per-sample statistics); they run
the regular eval forward under the same graph capture.
"""
import torch

from . import lib as L
from . import ops
from .networks import Generator
from .ops import pad4


def _fold(conv, norm, transposed=False):
    """(w', b') of conv followed by eval-mode BatchNorm `norm` (None -> conv as is)."""
    w = conv.weight.detach()
    b = conv.bias.detach() if conv.bias is not None else None
    if norm is None:
        return w.contiguous(), b
    scale = norm.weight.detach() / torch.sqrt(norm.running_var + ops.EPS)
    shift = norm.bias.detach() - norm.running_mean * scale
    co = scale.numel()
    if transposed:                       # physical [kh,kw,CoutT,CinT]
        w2 = w.clone()
        w2[:, :, :co, :] *= scale.view(1, 1, -1, 1)
    else:                                # physical [kh,kw,Cin,Cout]
        w2 = w.clone()
        w2[..., :co] *= scale.view(1, 1, 1, -1)
    b2 = torch.zeros(pad4(co), dtype=torch.float32, device=w.device)
    b2[:co] = shift if b is None else shift + b[:co] * scale
    return w2.contiguous(), b2


class InferenceGenerator:
    """Callable drop-in for ``Generator.eval()``: ``gen([H1, cat(P1,P2), cat(D1,D2)]) -> NCHW``."""

    def __init__(self, net: Generator, use_graph=True, bf16=False):
        assert isinstance(net, Generator)
        self.net = net.eval()
        self.bf16 = 2 if bf16 == 2 else bool(bf16)   # 16-bit MFMA compute (True bf16, 2 fp16) for the channel-%64 convs
        self.net.bf16 = self.bf16
        self.folded = net.norm == "batch"
        self.use_graph = use_graph
        self._graph = None
        self._key = None
        if self.folded:
            self._build_folded()

    # ------------------------------------------------------------------ folded weights
    def _build_folded(self):
        n, m = self.net, self.net.model
        i2 = 6 if n.use_dropout else 5
        f = {}
        for s in (1, 2, 3):
            d = m[f"stream{s}_down"]
            f[("down", s, 0)] = _fold(d[1], d[2])
            for i in range(n.n_down):
                f[("down", s, 1 + i)] = _fold(d[4 + 3 * i], d[5 + 3 * i])
        for b in range(n.n_blocks):
            blk = m["att"][b]
            for s in (1, 2, 3):
                cb = blk[f"conv_block_stream{s}"]
                f[("att", b, s, 0)] = _fold(cb[1], cb[2])
                f[("att", b, s, 1)] = _fold(cb[i2], cb[i2 + 1] if s == 1 else None)
        up = m["stream1_up"]
        for i in range(n.n_down):
            f[("up", i)] = _fold(up[3 * i], up[3 * i + 1], transposed=True)
        f[("head",)] = _fold(up[3 * n.n_down + 1], None)
        self.f = f
        self._graph = None              # a re-fold (BN recalibration) invalidates a captured graph
        ops.bump_weights_epoch()        # ... and every derived copy of the previous folded weights

    def refold(self):
        """Call after the network's weights or BN running statistics changed."""
        if self.folded:
            self._build_folded()
        else:
            self._graph = None
            ops.bump_weights_epoch()

    def _lp_chain_ok(self):
        """16-bit mode: every conv of the folded forward on the conv_lp16 kernels, activations handed from
        epilogue to loader in 16 bits (no twin conversions, no fp32 intermediate except the residual stream)"""
        n = self.net
        if not self.bf16 or n.n_down != 2 or n.n_blocks < 1:
            return False
        dim = n.ngf * 4
        d1 = ops.conv_desc(1, 16, 16, n.ngf, 2 * n.ngf, 3, 2, 1, False)
        d2 = ops.conv_desc(1, 8, 8, 2 * n.ngf, dim, 3, 2, 1, False)
        st = ops.conv_desc(1, 16, 16, 4, n.ngf, 7, 1, 3, True)
        return (ops.lp16_v2_ok(dim, dim, 3, 1, 1, 0) and ops.lp16_v2_ok(2 * dim, 2 * dim, 3, 1, 1, 0)
                and ops.lp16_v2_ok(2 * dim, dim, 3, 1, 1, 0) and ops.lp16g_ok(d1, 0, self.bf16) and ops.lp16g_ok(d2, 0, self.bf16)
                and ops.lp16_flat_ok(st, self.bf16) and ops.convT_lp16_ok(dim, 2 * n.ngf, self.bf16)
                and ops.convT_lp16_ok(2 * n.ngf, n.ngf, self.bf16))

    def _forward_folded_lp16(self, x1, x2, x3):
        """The folded forward with 16-bit activations between the kernels (BatchNorm folded, ReLU in the conv
        epilogues): stems on the flat-K kernel, stride-2 / transposed convs on conv_lp16g, the PATBlock convs
        on the halo kernel; the gate reads its two gates in 16 bits and writes the next block's concats in 16
        bits.  fp32 only: the residual stream x1 (its conv reads a twin) and the gate's s1."""
        n, f, lp = self.net, self.f, self.bf16
        xs = []
        for s, x in zip((1, 2, 3), (x1, x2, x3)):
            w, b = f[("down", s, 0)]
            B, H, W_, Cc = x.shape
            x = ops.raw_conv_lp16_flat(ops.conv_desc(B, H, W_, Cc, w.shape[3], 7, 1, 3, True), x, w, b, L.ACT_RELU, lp, out16=True)
            for i in range(n.n_down):
                w, b = f[("down", s, 1 + i)]
                B, H, W_, Cc = x.shape
                last = i + 1 == n.n_down
                x = ops.raw_conv_lp16g(ops.conv_desc(B, H, W_, Cc, w.shape[3], 3, 2, 1, False), 0, x, w, b, L.ACT_RELU, lp,
                                       out16=not (last and s == 1))        # stream 1 enters the fp32 residual stream
            xs.append(x)
        x1, x2, x3 = xs                     # x1 fp32, x2 / x3 16-bit
        for blk in range(n.n_blocks):
            ss = []
            for s, x in zip((1, 2, 3), (x1, x2, x3)):
                if x.dtype != torch.float32:
                    x16 = x
                elif s == 1 and blk > 0 and ops.USE_LP16_CAT_TWIN:
                    x16 = x2[..., x.shape[3]:]      # the second half of cat(s3, out): out in 16 bits, read in place
                else:
                    x16 = ops.lp16_twin(x, lp)
                w, b = f[("att", blk, s, 0)]
                y = ops.raw_conv3x3_lp16(x16, w, b, True, L.ACT_RELU, lp, 0, out16=True)
                w, b = f[("att", blk, s, 1)]
                ss.append(ops.raw_conv3x3_lp16(y, w, b, True, L.ACT_NONE, lp, 0, out16=s != 1))
            more = blk + 1 < n.n_blocks
            px = ops.lp_proxy(ss[1].shape, ss[1].device)
            res = ops.GateFn.apply(x1, ss[0], px, px, more, lp if more else 0, ss[1], ss[2])
            x1 = res[0]
            if more:
                x2, x3 = res[3], res[4]
        y = x1
        hw, hb = f[("head",)]
        Bh, Hh, Wh = x1.shape[0], x1.shape[1] << n.n_down, x1.shape[2] << n.n_down
        dh = ops.conv_desc(Bh, Hh, Wh, hw.shape[2], hw.shape[3], 7, 1, 3, True)
        head16 = hw.shape[3] == 4 and ops.conv7_n4_ok(dh, 0, lp)       # the head reads 16 bits too (conv7_n4.hip)
        for i in range(n.n_down):
            w, b = f[("up", i)]
            y = ops.raw_convT_fprop(y, w, b, L.ACT_RELU, lp, out16=i + 1 < n.n_down or head16)
        if head16:
            out = torch.empty((Bh, Hh, Wh, 4), dtype=torch.float32, device=y.device)
            return ops.raw_conv7_n4(dh, 0, y, hw, hb, out, L.ACT_TANH, lp)
        return ops.raw_conv_fprop(y, hw, hb, 1, 3, True, L.ACT_TANH, lp)

    def _forward_folded(self, x1, x2, x3):
        if self._lp_chain_ok():
            return self._forward_folded_lp16(x1, x2, x3)
        n, f = self.net, self.f
        xs = []
        for s, x in zip((1, 2, 3), (x1, x2, x3)):
            w, b = f[("down", s, 0)]
            x = ops.raw_conv_fprop(x, w, b, 1, 3, True, L.ACT_RELU, self.bf16)
            for i in range(n.n_down):
                w, b = f[("down", s, 1 + i)]
                x = ops.raw_conv_fprop(x, w, b, 2, 1, False, L.ACT_RELU, self.bf16)
            xs.append(x)
        x1, x2, x3 = xs
        for blk in range(n.n_blocks):
            ss = []
            for s, x in zip((1, 2, 3), (x1, x2, x3)):
                w, b = f[("att", blk, s, 0)]
                y = ops.raw_conv_fprop(x, w, b, 1, 1, True, L.ACT_RELU, self.bf16)
                w, b = f[("att", blk, s, 1)]
                ss.append(ops.raw_conv_fprop(y, w, b, 1, 1, True, L.ACT_NONE, self.bf16))
            x1, x2, x3 = ops.GateFn.apply(x1, ss[0], ss[1], ss[2], blk + 1 < n.n_blocks)
        y = x1
        for i in range(n.n_down):
            w, b = f[("up", i)]
            y = ops.raw_convT_fprop(y, w, b, L.ACT_RELU, self.bf16)
        w, b = f[("head",)]
        return ops.raw_conv_fprop(y, w, b, 1, 3, True, L.ACT_TANH, self.bf16)

    # ------------------------------------------------------------------ call
    def _eager(self, inputs):
        n = self.net
        xs = [ops.raw_pack([(x, True, nc)], x.shape[0], x.shape[2], x.shape[3], pad4(nc), x.device)
              for x, nc in zip(inputs, n.input_nc)]
        y = self._forward_folded(*xs) if self.folded else n.forward_nhwc(*xs)
        return ops.nhwc_to_nchw_view(y, n.output_nc)

    @torch.no_grad()
    def __call__(self, inputs):
        inputs = [x.float() for x in inputs]
        if not self.use_graph:
            return self._eager(inputs)
        key = tuple(tuple(x.shape) for x in inputs) + (inputs[0].device,)
        if self._graph is None or key != self._key:
            self._capture(inputs, key)
        for s, x in zip(self._static_in, inputs):
            s.copy_(x, non_blocking=True)
        self._graph.replay()
        return self._static_out

    def _capture(self, inputs, key):
        self._static_in = [x.clone() for x in inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                    # warm-up: one-time attribute calls, allocs
            for _ in range(2):
                self._eager(self._static_in)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._static_out = self._eager(self._static_in)
        # the captured launches read the derived weights (16-bit copies, flat-K stem copies, Winograd-domain
        # filters) made during the warm-up by raw pointer: hold ALL of them for as long as the graph lives
        # (ops drops its caches at every weights-epoch bump - any optimizer step, any re-fold)
        self._derived = ops.derived_weights_snapshot()
        self._graph, self._key = g, key
