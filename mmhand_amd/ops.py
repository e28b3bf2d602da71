"""PyTorch-ROCm shims over the C-ABI of libmmhand_hip.so.

Two layers:
  * ``raw_*`` functions: allocate outputs with the torch caching allocator and
    enqueue the HIP kernels on torch's current stream (tensors are passed as raw
    ``data_ptr()``; the library never allocates or synchronises);
  * ``torch.autograd.Function`` shims whose backward calls the matching
    dgrad/wgrad/backward kernels, so ``loss.backward()`` of the reference step
    (models/MMHandModel.py:294-308) runs entirely on the hand-written kernels.

All activations are physical NHWC fp32 tensors of shape [B, H, W, C] with C a
multiple of 4 (3/6/42-channel tensors are zero-padded to 4/8/44).  Conv weights
are physical [kh, kw, Cin, Cout]; the nn.Parameter objects the modules expose
are permuted *views* of them with the reference's logical OIHW / IOHW shape.
"""
import ctypes as C
import os
import weakref

import torch

from . import lib as L


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    # current stream of the CURRENT device: MMHandModel / the drivers set the device to the rank's
    # GPU before anything runs (one process per GPU), tensors are created on that same device.
    # Through torch's raw bindings: torch.cuda.current_stream() builds a Stream object behind three layers of device-index
    # helpers - 9 us per call, 1000 calls per step: 4.4 of the 40 ms the host needs to enqueue a 512x512 SyncBN step
    # (tools/probes/host_profile.py) - the raw pair is 0.3 us and returns the same handle (tests/test_pointwise_gpu.py)
    if _raw_stream is not None and _raw_device is not None:
        return C.c_void_p(_raw_stream(_raw_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t, name="tensor"):
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise RuntimeError(f"{name}: expected a contiguous fp32 CUDA tensor, got "
                           f"{t.dtype} {t.device} contiguous={t.is_contiguous()}")
    return t


def _chk16(t, name="tensor"):
    """fp32 or 16-bit contiguous device tensor"""
    if not (t.is_cuda and t.dtype in (torch.float32, torch.bfloat16, torch.float16) and t.is_contiguous()):
        raise RuntimeError(f"{name}: expected a contiguous fp32 / 16-bit CUDA tensor, got "
                           f"{t.dtype} {t.device} contiguous={t.is_contiguous()}")
    return t


def pad4(c):
    return (c + 3) // 4 * 4


def _empty(shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


def _ws(nbytes, like):
    return torch.empty(max(int(nbytes), 16) // 4 + 4, dtype=torch.float32, device=like.device)


# --------------------------------------------------------------------------- conv
fprop_timer = None   # optional bench hook: object with want(desc) / bracket() -> (event, event)
# optional bench hook: dict kind -> executed FLOPs ("mfma": matrix-core convolution arithmetic that the
# launches really perform - Winograd-domain GEMMs count their own multiplications, not the direct
# algorithm's; "valu": the thin 7x7 convs on the vector ALU).  None = not counting.
flop_meter = None


def _count(kind, flops):
    if flop_meter is not None:
        flop_meter[kind] = flop_meter.get(kind, 0.0) + float(flops)


def _count_desc(kind, d):
    if flop_meter is not None:
        _count(kind, 2.0 * d.B * d.Ho * d.Wo * d.Cin * d.Cout * d.kh * d.kw)

# bf16 MFMA path ("--opt_level O1/O2": the reference's apex AMP mixed precision becomes bf16
# compute with fp32 master weights, fp32 accumulation and fp32 tensors in HBM).  The bf16 weight
# copies are rebuilt lazily whenever the weights epoch moves (optimizer step, load_state_dict).
_weights_epoch = [0]
_bf16_cache = {}
_wino_cache = {}


def bump_weights_epoch(within=None):
    """Every derived copy of every weight (bf16 copies, Winograd-domain filters) is stale after an
    optimizer step / load_state_dict / broadcast: drop them all, so dead tensors' copies are freed
    too.  Entries additionally hold a weak reference to the tensor object they were made from - a new
    tensor that happens to land on a freed tensor's address never hits a stale entry.
    within (a tensor, e.g. one network's flat parameter buffer): only the copies of weights that live inside it are
    dropped - an optimizer step changes one network, and the Discriminators' copies made during the Generator step
    serve their own step too (96 -> 78 weight conversions per 16-bit iteration)."""
    _pending_stats.clear()
    if within is not None:
        lo = within.data_ptr()
        hi = lo + within.numel() * within.element_size()
        for c in _DERIVED_CACHES():
            for k in [k for k in c if lo <= k[0] < hi]:
                del c[k]
        for flag, batches, cls in ((USE_WINO_BATCH, _wino_batches, _WinoBatch), (USE_LP16_BATCH, _lp16_batches, _Lp16Batch)):
            if not flag:
                continue
            b = batches.get((lo, hi))
            if b is None:
                for k in [k for k in batches if k[0] < hi and lo < k[1]]:     # a dead model's buffer lived here
                    del batches[k]
                b = batches[(lo, hi)] = cls(lo, hi)
            b.stale = True
        return
    _weights_epoch[0] += 1
    for c in _DERIVED_CACHES():
        c.clear()
    _wino_batches.clear()        # load_state_dict / a new model: the batches learn their filters again
    _lp16_batches.clear()


def derived_weights_snapshot():
    """Every derived weight tensor alive right now (16-bit copies, flat-K copies, Winograd-domain filters), as
    one object to hold on to: a captured hipGraph reads these tensors by raw pointer, and bump_weights_epoch()
    - every optimizer step, every re-fold - empties the caches.  New caches must be added to _DERIVED_CACHES."""
    return tuple(dict(c) for c in _DERIVED_CACHES())


def _DERIVED_CACHES():
    return (_bf16_cache, _flat8_cache, _wino_cache, _stem16_cache)


def derived_batches_snapshot():
    """the tensors the per-network batches keep (the 16-bit copies of _Lp16Batch, the Winograd-domain filters of _WinoBatch,
    their device tables): a captured TRAINING step rewrites them in every replay, so its owner holds them like
    derived_weights_snapshot()'s"""
    keep = []
    for batches in (_lp16_batches, _wino_batches):
        for b in batches.values():
            keep.append(b.table)
            keep.extend(e[1] for e in b.entries.values())
    return keep


def _cache_get(cache, key, w):
    ent = cache.get(key)
    if ent is not None and ent[0] == _weights_epoch[0] and ent[1]() is w:
        return ent[2]
    return None


def _cache_put(cache, key, w, value):
    cache[key] = (_weights_epoch[0], weakref.ref(w), value)
    return value


def _lp(bf16):
    """The `bf16` argument of every conv entry point below selects the MFMA operand type:
    False / 0 = fp32, True / 1 = bf16, 2 = IEEE fp16 (apex O1's precision; needs loss scaling)."""
    return 0 if not bf16 else (2 if bf16 == 2 else 1)


def _dt(bf16):
    return (L.F32, L.BF16, L.FP16)[_lp(bf16)]


def _wd(bf16):
    return (torch.float32, torch.bfloat16, torch.float16)[_lp(bf16)]


def _lp_of(t):
    """the `bf16` argument value matching a tensor's element type (0 fp32 | True bf16 | 2 fp16)"""
    return {torch.float32: 0, torch.bfloat16: True, torch.float16: 2}[t.dtype]


# One launch for all 16-bit weight copies of a network (mmh_prep_weights_lp16_multi) - the 16-bit counterpart of
# _WinoBatch below, with the same life cycle: a batch belongs to one parameter buffer (the `within` of bump_weights_epoch),
# LEARNS the weights the network converts during one iteration (each still converted on its own) and from then on the
# first request after a step refreshes all of them, in the buffers the batch keeps, with one launch (78 launches of
# 5-15 us, 0.86 ms per 16-bit iteration, become three).  MMH_LP16_BATCH=0: every weight on its own.
USE_LP16_BATCH = os.environ.get("MMH_LP16_BATCH", "1") != "0"
# the Generator head's input gradient (64 <- 4 channels, 7x7 reflect) on the 16-bit stem kernel (MMH_HEAD_DGRAD16=0: the fp32
# implicit GEMM over the padded domain + fold pass)
USE_HEAD_DGRAD16 = os.environ.get("MMH_HEAD_DGRAD16", "1") != "0"
_lp16_batches = {}


class _Lp16Batch:
    def __init__(self, lo, hi):
        self.lo, self.hi = lo, hi
        self.entries = {}           # key -> [weakref(w), (wp, wt), taps, cin, cout, fp16]
        self.table = None           # device int64 [n][8], rebuilt when the set of entries changes
        self.order = []
        self.blocks = 0
        self.stale = False          # an optimizer step since the last batched conversion

    def learn(self, key, w, pair, taps, cin, cout, fp16):
        self.entries[key] = [weakref.ref(w), pair, taps, cin, cout, fp16]
        self.table = None

    def fetch(self, key, w):
        ent = self.entries.get(key)
        if ent is None or ent[0]() is not w or not self.stale:
            return None
        live = [(k, e) for k, e in self.entries.items() if e[0]() is not None]
        if len(live) != len(self.entries):
            self.entries = dict(live)
            self.table = None
        if self.table is None:
            rows, first = [], 0
            self.order = list(self.entries)
            for k in self.order:
                wr, (wp, wt), taps, ci, co, h16 = self.entries[k]
                rows.append([wr().data_ptr(), wp.data_ptr(), wt.data_ptr(), taps, ci, co, first, int(h16)])
                first += taps * ((ci + 63) // 64) * ((co + 63) // 64)
            self.blocks = first
            self.table = torch.tensor(rows, dtype=torch.int64).to(w.device)
        L.call("mmh_prep_weights_lp16_multi", _ptr(self.table), len(self.order), self.blocks, _stream())
        self.stale = False
        for k in self.order:
            wr, pair = self.entries[k][:2]
            _cache_put(_bf16_cache, k, wr(), pair)
        return ent[1]


def _lp16_batch_of(w):
    a = w.data_ptr()
    for b in _lp16_batches.values():
        if b.lo <= a < b.hi:
            return b
    return None


def bf16_weights(w, bf16=True):
    """(w_plain [k,k,Cin,Cout], w_t [k,k,Cout,Cin]) 16-bit copies of a physical fp32 weight; for
    Cin % 64 != 0 the second entry is w_flat [Cout, Kpad] (flat (tap, ci) contraction index)."""
    key = (w.data_ptr(), tuple(w.shape), _lp(bf16))
    ent = _cache_get(_bf16_cache, key, w)
    if ent is None:
        k, _, cin, cout = w.shape
        batch = _lp16_batch_of(w) if (USE_LP16_BATCH and cin % 64 == 0) else None
        if batch is not None:
            ent = batch.fetch(key, w)
            if ent is not None:
                return ent
        wd, fn = _wd(bf16), "mmh_prep_weights_" + ("fp16" if _lp(bf16) == 2 else "bf16")
        wp = torch.empty((k, k, cin, cout), dtype=wd, device=w.device)
        if cin % 64 == 0:
            wt = torch.empty((k, k, cout, cin), dtype=wd, device=w.device)
            L.call(fn, _ptr(w), k * k, cin, cout, _ptr(wp), _ptr(wt), _stream())
            if batch is not None:
                batch.learn(key, w, (wp, wt), k * k, cin, cout, _lp(bf16) == 2)
        else:
            kpad = (k * k * cin + 63) // 64 * 64
            wt = torch.empty((cout, kpad), dtype=wd, device=w.device)
            L.call(fn, _ptr(w), k * k, cin, cout, _ptr(wp), None, _stream())
            L.call(fn + "_flat", _ptr(w), k * k, cin, cout, _ptr(wt), _stream())
        ent = _cache_put(_bf16_cache, key, w, (wp, wt))
    return ent


def conv_desc(B, H, W, Cin, Cout, k, stride, pad, reflect, x_cs=None, y_cs=None):
    Ho = (H + 2 * pad - k) // stride + 1
    Wo = (W + 2 * pad - k) // stride + 1
    return L.ConvDesc(B, H, W, Cin, Cout, k, k, stride, pad,
                      L.PAD_REFLECT if reflect else L.PAD_ZERO, Ho, Wo,
                      x_cs or Cin, y_cs or Cout, L.F32)


# Winograd for the fp32 3x3 / stride-1 / pad-1 convs: F(6x6,3x3) (ragged tiles, any H and W;
# 5.06x fewer multiplications than the direct implicit GEMM, fp32 error 5e-6) from 128x128 channel
# pairs up, F(4x4,3x3) (4x fewer, H and W multiples of 4) from 64x64, F(2x2,3x3) (2.25x fewer)
# for even sizes from 256x256.  Below those sizes the direct kernel is faster (transform-bound).
# MMH_WINOGRAD=0 or ops.USE_WINOGRAD = False selects the direct kernels everywhere;
# MMH_WINOGRAD_TILE / ops.WINOGRAD_TILE = 4 | 2 caps the tile size.
USE_WINOGRAD = os.environ.get("MMH_WINOGRAD", "1") != "0"
# MMH_WINOGRAD=bwd (ops.set_winograd_mode("bwd"), train option --fp32_exact_grads): the gradient-exact fp32 hybrid.
# The FORWARD 3x3 convs run on the direct implicit-GEMM kernels, so activations - and with them every ReLU mask - are
# those of the direct path; dgrad and wgrad stay on F(6x6,3x3), whose rounding differences stay what they are (1e-5 on
# the gradients, profiles/r03_wino_grad_split.txt) instead of being amplified into mask flips (3e-3).  The backward
# computes the transformed input V itself (one extra input transform per conv) and runs the fused dy pass.
WINOGRAD_FPROP = os.environ.get("MMH_WINOGRAD", "1") != "bwd"


def set_winograd_mode(mode, direct_levels=None):
    """"all" (default: every pass of the eligible fp32 3x3 convs on Winograd), "bwd" (direct fprop, Winograd dgrad + wgrad:
    the gradient-exact hybrid), "off" (direct kernels everywhere).
    direct_levels (1 | 2; default 2 for "bwd" and "off", 1 for "all"; MMH_DIRECT_LEVELS overrides the default): summation
    levels of the direct fp32 fprop with more than 64 output channels (mmh_set_option("conv_levels")).  Two levels - a fresh
    MFMA chain per 32-deep k-step, folded by vector adds - take the full-size Generator's output from 2.9e-6 to 1.1e-6 of
    float64 (PyTorch's fp32 on the CPU: 1.2e-6) and with it the parameter gradients from a median 2.2e-3 to 9.5e-4 (PyTorch:
    7.6e-4): at 256x256 the gradients' distance follows the FORWARD's.  The modes whose point is accuracy get it;
    the all-Winograd headline keeps the one-level direct kernels for its stride-2 convs (its forward error is the
    Winograd transforms')."""
    global USE_WINOGRAD, WINOGRAD_FPROP
    if mode not in ("all", "bwd", "off"):
        raise ValueError(f"winograd mode {mode!r}: expected all | bwd | off")
    USE_WINOGRAD, WINOGRAD_FPROP = mode != "off", mode != "bwd"
    if direct_levels is None:
        direct_levels = int(os.environ.get("MMH_DIRECT_LEVELS", "2" if mode != "all" else "1"))
    L.call("mmh_set_option", b"conv_levels", 2 if int(direct_levels) == 2 else 1)
    bump_weights_epoch()
WINOGRAD_TILE = int(os.environ.get("MMH_WINOGRAD_TILE", "6"))
WINO6_MIN = 128 * 128      # F(6x6,3x3) from this Cin*Cout up (64 planes of filter transform per conv)
# keep the forward pass's transformed input for the wgrad pass (1.78x the activation's bytes per
# eligible conv at F(6,3), 2.25x at F(4,3); one input transform less per conv and step);
# MMH_WINOGRAD_KEEP_INPUT=0 re-transforms
KEEP_WINOGRAD_INPUT = os.environ.get("MMH_WINOGRAD_KEEP_INPUT", "1") != "0"
# 7x7 convs with <= 4 output columns (Generator head fprop, Discriminator-stem dgrad towards the
# generated image) on the vector-ALU kernel of conv_thin.hip; MMH_THIN=0 keeps them on the MFMA path
USE_THIN = os.environ.get("MMH_THIN", "1") != "0"
# fp32 wgrad of the 7x7 stems on the LDS-band kernel of conv_stem.hip; MMH_STEM_WGRAD=0: the generic direct wgrad
USE_STEM_WGRAD = os.environ.get("MMH_STEM_WGRAD", "1") != "0"
# F(6x6,3x3) backward: both transforms of dy (dgrad and wgrad operands) from one read of dy
FUSE_WINO6_BWD = os.environ.get("MMH_FUSE_WINO6_BWD", "1") != "0"
# --opt_level O1/O2: the 3x3 stride-1 convs with channels % 128 == 0 on bf16 Winograd F(2x2,3x3)
USE_WINOGRAD_BF16 = os.environ.get("MMH_WINOGRAD_BF16", "1") != "0"


# bf16 Winograd engages per pass where it was measured faster than the direct bf16 kernel at B=32,
# 64x64 (tools/bench_wino_bf16.py): fprop 512->512 1.50x, 512->256 1.13x, 256->256 0.90x; dgrad
# 1.31x / 0.97x / 0.83x; wgrad 1.91x / 1.49x / 1.34x.  Minimum Cin*Cout per pass:
WINO_BF16_MIN = {"fprop": 512 * 256, "dgrad": 512 * 512, "wgrad": 0}


def _wino_tile(B, H, W_, Cin, Cout, k, stride, pad, bf16, op="fprop"):
    """0 = direct kernel, else the Winograd output-tile size (2, 4 or 6)."""
    if not (USE_WINOGRAD and k == 3 and stride == 1 and pad == 1 and Cin % 32 == 0 and Cout % 32 == 0):
        return 0
    if op == "fprop" and not WINOGRAD_FPROP and not bf16:
        return 0
    if bf16:
        # fprop / dgrad of the 256- and 512-channel stack: the second-generation direct kernel
        # (conv_lp16.hip, 870-1000 TFLOP/s) beats F(2x2,3x3) + its HBM-bound transforms
        if op in ("fprop", "dgrad") and lp16_v2_ok(Cin, Cout, k, stride, pad, 0 if op == "fprop" else 1):
            return 0
        # bf16 MFMA path: F(2x2,3x3) with bf16 Winograd-domain tensors (F(4x4,3x3) would amplify
        # the bf16 rounding ~8x: 2e-2 vs 4e-3 relative); the bf16 GEMM kernels need channels % 128
        ok = (USE_WINOGRAD_BF16 and H % 2 == 0 and W_ % 2 == 0 and H >= 4 and W_ >= 4
              and Cin % 128 == 0 and Cout % 128 == 0 and Cin * Cout >= WINO_BF16_MIN[op])
        return 2 if ok else 0
    # measured on MI355X at B=32: F(4,3) wins from 64x64 channels up (1.1-2.8x), F(2,3) only from
    # 256x256 up (1.25-1.6x; transform-bound below); F(6,3) (ragged tiles, any size) has 21 % fewer
    # multiplications and a 21 % smaller Winograd domain than F(4,3)
    if WINOGRAD_TILE == 6 and H >= 12 and W_ >= 12 and Cin * Cout >= WINO6_MIN:
        return 6
    if WINOGRAD_TILE >= 4 and H % 4 == 0 and W_ % 4 == 0 and H >= 8 and W_ >= 8 and Cin * Cout >= 64 * 64:
        return 4
    if H % 2 == 0 and W_ % 2 == 0 and H >= 4 and W_ >= 4 and Cin * Cout >= 256 * 256:
        return 2
    return 0


def wino_weights(w, tile, flip_transpose=False, bf16=False):
    """Winograd-domain filter of a physical 3x3 weight (cached per weights epoch): fp32 U [P,K,N],
    bf16 U [P,N,K] (contraction index contiguous)."""
    key = (w.data_ptr(), tuple(w.shape), tile, bool(flip_transpose), _lp(bf16))
    U = _cache_get(_wino_cache, key, w)
    if U is None:
        _, _, cin, cout = w.shape
        batch = _wino_batch_of(w) if (USE_WINO_BATCH and tile == 6 and not bf16) else None
        if batch is not None:
            U = batch.fetch(key, w, cin, cout, bool(flip_transpose))
            if U is not None:
                return U
        P = (tile + 2) ** 2
        kn = (cout, cin) if flip_transpose else (cin, cout)        # (K, N)
        U = torch.empty((P, kn[1], kn[0]) if bf16 else (P, kn[0], kn[1]),
                        dtype=_wd(bf16), device=w.device)
        L.call("mmh_wino_weights", _ptr(w), cin, cout, int(flip_transpose), tile, _dt(bf16),
               _ptr(U), _stream())
        _cache_put(_wino_cache, key, w, U)
        if batch is not None:
            batch.learn(key, w, U, cin, cout, bool(flip_transpose))
    return U


# One launch for all F(6x6,3x3) filter transforms of a network (mmh_wino_weights_multi).  After an optimizer step every
# Winograd-domain filter of that network is stale; transforming them one by one as the forward pass reaches them is 74
# launches of 8-25 us that cannot fill the chip (1.9 ms per fp32 step).  A batch belongs to one parameter buffer (the `within`
# of bump_weights_epoch): it LEARNS the (weight, flip) pairs the network asks for during one iteration - each still
# transformed on its own - and from then on the first request after a step transforms all of them, into buffers the batch
# keeps, with one launch.  MMH_WINO_BATCH=0: every filter on its own.
USE_WINO_BATCH = os.environ.get("MMH_WINO_BATCH", "1") != "0"
_wino_batches = {}


class _WinoBatch:
    def __init__(self, lo, hi):
        self.lo, self.hi = lo, hi
        self.entries = {}           # key -> [weakref(w), U, cin, cout, flip]
        self.table = None           # device int64 [n][6], rebuilt when the set of entries changes
        self.order = []
        self.blocks = 0
        self.stale = False          # an optimizer step since the last batched transform

    def learn(self, key, w, U, cin, cout, flip):
        self.entries[key] = [weakref.ref(w), U, cin, cout, flip]
        self.table = None

    def fetch(self, key, w, cin, cout, flip):
        """The Winograd-domain filter of (w, flip) if this batch knows the pair: all of the batch's filters are
        transformed now if a step has gone by since the last time.  None: not known (the caller transforms it alone)."""
        ent = self.entries.get(key)
        if ent is None or ent[0]() is not w or not self.stale:
            return None
        live = [(k, e) for k, e in self.entries.items() if e[0]() is not None]
        if len(live) != len(self.entries):
            self.entries = dict(live)
            self.table = None
        if self.table is None:
            rows, first = [], 0
            self.order = list(self.entries)
            for k in self.order:
                wr, U, ci, co, fl = self.entries[k]
                rows.append([wr().data_ptr(), U.data_ptr(), ci, co, int(fl), first])
                first += (ci * co + 255) // 256
            self.blocks = first
            self.table = torch.tensor(rows, dtype=torch.int64).to(w.device)
        L.call("mmh_wino_weights_multi", _ptr(self.table), len(self.order), self.blocks, _stream())
        self.stale = False
        for k in self.order:
            wr, U, _, _, _ = self.entries[k]
            _cache_put(_wino_cache, k, wr(), U)
        return ent[1]


def _wino_batch_of(w):
    a = w.data_ptr()
    for b in _wino_batches.values():
        if b.lo <= a < b.hi:
            return b
    return None


# Reflect-fold dgrad for F(6x6,3x3): the gradient of a ReflectionPad2d(1) conv is taken on the padded
# (H+2) x (W+2) domain with the tile origin on the pad ring, where every ring pixel and the pixel it
# folds onto share one 6x6 tile - the output transform adds them in registers.  Replaces the eight
# border GEMMs + border_add per conv (16 ms of a 303 ms step).  MMH_WINO_FOLD=0: border-GEMM path.
USE_WINO_FOLD = os.environ.get("MMH_WINO_FOLD", "1") != "0"
# Summation of the fp32 Winograd dgrad GEMMs: 1 = one k-ordered chain (5-6 % faster), 2 = the two-level sum of the forward
# GEMMs, 0 = the library default.  The two-level sum exists because a forward rounding difference flips ReLU masks and moves
# this network's gradients by 3e-3; a rounding difference in a BACKWARD GEMM stays what it is (Winograd dgrad alone against
# the direct kernels: 6e-6 with two levels, profiles/r03_wino_grad_split.txt).
WINO_DGRAD_LEVELS = int(os.environ.get("MMH_WINO_DGRAD_LEVELS", "1"))


def _fold_ok(H, W_):
    return USE_WINO_FOLD and H >= 6 and W_ >= 6 and (H + 1) % 6 >= 2 and (W_ + 1) % 6 >= 2


def _fold_same_grid(H, W_):
    """fused backward transform: dgrad (padded-domain) and wgrad operands share one tile grid"""
    return _fold_ok(H, W_) and -(-(H + 2) // 6) == -(-H // 6) and -(-(W_ + 2) // 6) == -(-W_ // 6)


# Per-tile output statistics of the last Winograd F(6x6,3x3) conv outputs, keyed by the output's
# data pointer: NormActFn (instance norm) picks them up instead of re-reading the conv output.
FUSE_NORM_STATS = os.environ.get("MMH_FUSE_NORM_STATS", "1") != "0"
# ... also from the epilogues of the 16-bit 7x7 stems (conv_stem16.hip) and of the stride-2 kernel (conv_s2_lp16.hip): built,
# tested (tests/test_conv_s2_lp16_gpu.py) and measured - the 16-bit step runs 101.5 ms with them against 101.0 without
# (tools/ab_step.py ops:FUSE_NORM_STATS_NARROW): at 64 / 128 channels the statistics pass they save (33-59 us) costs less
# than the epilogue arithmetic on kernels that hold one or two workgroups per CU.  Off by default.
FUSE_NORM_STATS_NARROW = os.environ.get("MMH_FUSE_NORM_STATS_NARROW", "0") == "1"
_pending_stats = {}


def _park_stats(y, stats):
    """Partial statistics of a conv output y, left for the norm that reads y next (raw_norm_stats[_finalize_pending]).
    A conv asks for them only when its output goes straight into a norm (to_norm), so an entry lives until that norm
    takes it - the three generator streams run conv, conv, conv, norm, norm, norm under SyncBN packing - and every
    optimizer step / epoch bump empties the table, so no entry outlives its iteration.  The key is the output's address
    and shape; the table is kept tiny (a conv whose norm never came would otherwise leave an entry behind)."""
    if len(_pending_stats) >= 8:
        _pending_stats.clear()
    _pending_stats[y.data_ptr()] = (stats, tuple(y.shape), weakref.ref(y), y._version)


def _take_stats(x, peek=False):
    """the partials parked for the tensor that still lives at x's address (not for a dead conv output whose address the
    caching allocator handed to a new tensor: the producing conv's norm may never have run), else None"""
    pend = _pending_stats.get(x.data_ptr())
    if pend is None:
        return None
    y = pend[2]()
    if pend[1] != tuple(x.shape) or y is None or y._version != pend[3] or x._version != pend[3]:
        del _pending_stats[x.data_ptr()]
        return None
    if not peek:
        del _pending_stats[x.data_ptr()]
    return pend


def _wino_conv(x, U, bias, Cout, reflect, act, tile, time_it=False, keep_V=False, bf16=False, want_stats=False,
               fold=False, pro=None, levels=0):
    """input transform -> P batched GEMMs (one launch) -> output transform (+bias, activation).
    keep_V also returns the transformed input (the wgrad pass contracts exactly this tensor).
    bf16: V, U, M are bf16 (tile 2), the GEMMs run on the bf16 MFMA; x, y stay fp32.
    fold (tile 6, fp32): x is dy of a reflect-padded conv; tiles cover the padded domain and the
    output transform folds the pad ring back (reflect is ignored)."""
    B, H, W_, Cin = x.shape
    P = (tile + 2) ** 2
    if fold:
        tiles = B * (-(-(H + 2) // 6)) * (-(-(W_ + 2) // 6))
    else:
        tiles = B * (-(-H // tile)) * (-(-W_ // tile))      # F(6x6,3x3) tiles are ragged
    dt = _dt(bf16)
    wd = _wd(bf16)
    V = torch.empty((P, tiles, Cin), dtype=wd, device=x.device)
    M = torch.empty((P, tiles, Cout), dtype=wd, device=x.device)
    y = _empty((B, H, W_, Cout), x)
    timed = time_it and fprop_timer is not None and fprop_timer.want_gemm(P, tiles, Cin, Cout)
    if timed:       # HIP events on the launch stream (bench.py roofline): the GEMM launch alone, and
        e0, e1, o0, o1 = fprop_timer.bracket_op()       # the whole op (both transforms + GEMM)
        o0.record()
    if pro is not None:     # the norm-apply / ReLU / dropout in front of this conv, inside the transform (NormDefer)
        assert tile == 6 and not bf16 and not fold and tuple(pro.x.shape) == tuple(x.shape)
        L.call("mmh_wino_input_normact", _ptr(pro.x), B, H, W_, Cin, int(bool(reflect)), _ptr(V), _ptr(pro.scale),
               _ptr(pro.shift), pro.groups, int(pro.relu), float(pro.drop_p), _ptr(pro.drows), _stream())
    else:
        L.call("mmh_wino_input", _ptr(x), B, H, W_, Cin, 2 if fold else int(bool(reflect)), tile, dt, _ptr(V), _stream())
    if timed:
        e0.record()
    if levels and not bf16:
        L.call("mmh_wino_gemm_levels", _ptr(V), _ptr(U), _ptr(M), tiles, Cin, Cout, P, levels, _stream())
    else:
        L.call("mmh_wino_gemm", _ptr(V), _ptr(U), _ptr(M), tiles, Cin, Cout, P, dt, _stream())
    if timed:
        e1.record()
    _count("mfma", 2.0 * P * tiles * Cin * Cout)
    stats = None
    if want_stats and FUSE_NORM_STATS and tile == 6 and not bf16 and act == L.ACT_NONE:
        stats = _empty((B, tiles // B, 3, Cout), x)
    L.call("mmh_wino_output", _ptr(M), _ptr(y), _ptr(bias), B, H, W_, Cout, act, tile, dt, _ptr(stats),
           int(bool(fold)), _stream())
    if timed:
        o1.record()
    if stats is not None:
        _park_stats(y, stats)
    return (y, V) if keep_V else y


def raw_conv_fprop_wino(x, w, bias, reflect, act=L.ACT_NONE, tile=4, keep_V=False, bf16=False, pro=None):
    """3x3 / stride 1 / pad 1 conv by Winograd (fp32, or bf16 Winograd-domain tensors with tile 2).
    pro (NormDefer): x is only a shape carrier; the conv runs on dropout(relu(pro.x * scale + shift))."""
    if pro is None:
        _chk(x, "x")
    _chk(w, "w")
    return _wino_conv(x, wino_weights(w, tile, False, bf16), bias, w.shape[3], reflect, act, tile, time_it=True,
                      keep_V=keep_V, bf16=bf16, want_stats=True, pro=pro)


def raw_conv_dgrad_wino(dy, w, x_shape, reflect, tile=4, bf16=False):
    """fp32 3x3 / stride 1 / pad 1 dgrad (folded) by Winograd: the zero-padded correlation of dy
    with the flipped filter gives g on the real domain; reflect padding adds the eight border
    terms exactly as the direct path does.  (Running the border GEMMs on a second stream beside
    the transforms was measured 2.6 % SLOWER per step than in line: the two cross-stream waits
    per conv cost more than the ~100 us of latency-bound GEMMs they hide.)"""
    _chk(dy, "dy"); _chk(w, "w")
    B, H, W_, Cin = x_shape
    Cout = w.shape[3]
    if reflect and tile == 6 and not bf16 and _fold_ok(H, W_):
        return _wino_conv(dy, wino_weights(w, tile, True, bf16), None, Cin, False, L.ACT_NONE, tile, fold=True,
                          levels=WINO_DGRAD_LEVELS)
    dx = _wino_conv(dy, wino_weights(w, tile, True, bf16), None, Cin, False, L.ACT_NONE, tile, bf16=bf16, levels=WINO_DGRAD_LEVELS)
    if reflect:
        d = conv_desc(B, H, W_, Cin, Cout, 3, 1, 1, True)
        wb = w
        if bf16 and Cout % 64 == 0:         # border GEMMs on the bf16 MFMA too
            d.dtype = _dt(bf16)
            wb = bf16_weights(w, bf16)[0]
        ws = _ws(L.load().mmh_conv2d_dgrad_border_ws_bytes(C.byref(d)), dy)
        L.call("mmh_conv2d_dgrad_border", C.byref(d), _ptr(dy), _ptr(wb), _ptr(dx), _ptr(ws), ws.numel() * 4, 3,
               0, _stream())
    return dx


def raw_conv_wgrad_wino(x, dy, reflect, tile=4, V=None, bf16=False, out=None):
    """fp32 3x3 / stride 1 / pad 1 wgrad by Winograd: dw = G^T [sum_tiles (B^T d B).(A dY A^T)] G.
    V = the transformed input kept by the forward pass (x is then unused and may be None)."""
    _chk(dy, "dy")
    B, H, W_, Cout = dy.shape
    P = (tile + 2) ** 2
    tiles = B * (-(-H // tile)) * (-(-W_ // tile))      # F(6x6,3x3) tiles are ragged
    dt = _dt(bf16)
    wd = _wd(bf16)
    if V is None:
        _chk(x, "x")
        Cin = x.shape[3]
        V = torch.empty((P, tiles, Cin), dtype=wd, device=dy.device)
        L.call("mmh_wino_input", _ptr(x), B, H, W_, Cin, int(bool(reflect)), tile, dt, _ptr(V), _stream())
    else:
        Cin = V.shape[2]
        assert tuple(V.shape) == (P, tiles, Cin) and V.is_contiguous() and V.dtype == wd
    Yh = torch.empty((P, tiles, Cout), dtype=wd, device=dy.device)
    L.call("mmh_wino_dy", _ptr(dy), B, H, W_, Cout, tile, dt, _ptr(Yh), _stream())
    ws = _ws(L.load().mmh_wino_wgrad_gemm_ws_bytes(tiles, Cin, Cout, P), dy)
    dU = _empty((P, Cin, Cout), dy)
    L.call("mmh_wino_wgrad_gemm", _ptr(V), _ptr(Yh), tiles, Cin, Cout, P, dt, _ptr(ws), ws.numel() * 4, _ptr(dU),
           _stream())
    _count("mfma", 2.0 * P * tiles * Cin * Cout)
    dw = out if out is not None else _empty((3, 3, Cin, Cout), dy)
    L.call("mmh_wino_dw", _ptr(dU), Cin, Cout, tile, _ptr(dw), int(out is not None), _stream())
    return dw


def raw_conv_bwd_wino6(dy, w, x_shape, reflect, V, dw_out=None, nbd=None):
    """dgrad and wgrad of a 3x3 / stride 1 / pad 1 conv by Winograd F(6x6,3x3) with ONE pass over
    dy for both of its transforms (mmh_wino_input_dy).  V = the forward pass's transformed input.
    nbd (NormBwdDefer): dy does not exist - the transform computes it per element from the gradient
    of the norm that follows this conv (mmh_wino_input_dy_normbwd).  Returns (dx, dw)."""
    _chk(w, "w")
    B, H, W_, Cin = x_shape
    Cout = w.shape[3]
    P, tile = 64, 6
    tiles = B * (-(-H // tile)) * (-(-W_ // tile))
    assert tuple(V.shape) == (P, tiles, Cin) and V.is_contiguous() and V.dtype == torch.float32
    fold = bool(reflect) and _fold_same_grid(H, W_)
    if nbd is not None:
        dy = nbd.g
        assert tuple(dy.shape) == (B, H, W_, Cout) and tuple(nbd.x.shape) == tuple(dy.shape)
        Vd = _empty((P, tiles, Cout), dy)
        Yh = _empty((P, tiles, Cout), dy)
        L.call("mmh_wino_input_dy_normbwd", _ptr(nbd.g), _ptr(nbd.x), B, H, W_, Cout, _ptr(Vd), _ptr(Yh), int(fold),
               _ptr(nbd.mean), _ptr(nbd.invstd), _ptr(nbd.gamma), _ptr(nbd.s1), _ptr(nbd.s2), float(nbd.count),
               _ptr(nbd.scale), _ptr(nbd.shift), _ptr(nbd.drows), nbd.groups, int(nbd.relu), float(nbd.drop_p),
               _stream())
    else:
        _chk(dy, "dy")
        Vd = _empty((P, tiles, Cout), dy)
        Yh = _empty((P, tiles, Cout), dy)
        L.call("mmh_wino_input_dy", _ptr(dy), B, H, W_, Cout, tile, L.F32, _ptr(Vd), _ptr(Yh), int(fold), _stream())
    # dgrad: correlation with the flipped filter; reflect padding: folded in the output transform
    # (padded-domain tiles) or, where the size does not allow it, the eight border GEMMs
    M = _empty((P, tiles, Cin), dy)
    if WINO_DGRAD_LEVELS:
        L.call("mmh_wino_gemm_levels", _ptr(Vd), _ptr(wino_weights(w, tile, True)), _ptr(M), tiles, Cout, Cin, P, WINO_DGRAD_LEVELS,
               _stream())
    else:
        L.call("mmh_wino_gemm", _ptr(Vd), _ptr(wino_weights(w, tile, True)), _ptr(M), tiles, Cout, Cin, P, L.F32, _stream())
    _count("mfma", 2.0 * 2 * P * tiles * Cin * Cout)       # this GEMM and the wgrad GEMM below
    dx = _empty((B, H, W_, Cin), dy)
    L.call("mmh_wino_output", _ptr(M), _ptr(dx), None, B, H, W_, Cin, L.ACT_NONE, tile, L.F32, None, int(fold),
           _stream())
    if reflect and not fold:
        assert nbd is None, "the border GEMMs read dy: the caller materialises it (Conv2dFn.backward)"
        d = conv_desc(B, H, W_, Cin, Cout, 3, 1, 1, True)
        ws = _ws(L.load().mmh_conv2d_dgrad_border_ws_bytes(C.byref(d)), dy)
        L.call("mmh_conv2d_dgrad_border", C.byref(d), _ptr(dy), _ptr(w), _ptr(dx), _ptr(ws), ws.numel() * 4, 3,
               0, _stream())
    # wgrad
    ws = _ws(L.load().mmh_wino_wgrad_gemm_ws_bytes(tiles, Cin, Cout, P), dy)
    dU = _empty((P, Cin, Cout), dy)
    L.call("mmh_wino_wgrad_gemm", _ptr(V), _ptr(Yh), tiles, Cin, Cout, P, L.F32, _ptr(ws), ws.numel() * 4, _ptr(dU),
           _stream())
    dw = dw_out if dw_out is not None else _empty((3, 3, Cin, Cout), dy)
    L.call("mmh_wino_dw", _ptr(dU), Cin, Cout, tile, _ptr(dw), int(dw_out is not None), _stream())
    return dx, dw


def raw_conv_fprop(x, w, bias, stride, pad, reflect, act=L.ACT_NONE, bf16=False, want_stats=False):
    """want_stats: the output feeds an InstanceNorm directly - the direct fp32 kernel then leaves per-tile statistics
    of y for it (as the Winograd output transform does; _pending_stats)"""
    _chk(x, "x"); _chk(w, "w")
    B, H, W_, Cin = x.shape
    k, _, wc, Cout = w.shape
    assert wc == Cin, f"weight Cin {wc} != x channels {Cin}"
    wt = _wino_tile(B, H, W_, Cin, Cout, k, stride, pad, bf16)
    if wt:
        return raw_conv_fprop_wino(x, w, bias, reflect, act, wt, bf16=bf16)
    d = conv_desc(B, H, W_, Cin, Cout, k, stride, pad, reflect)
    y = _empty((B, d.Ho, d.Wo, Cout), x)
    if USE_THIN and k == 7 and stride == 1 and pad == 3 and Cout == 4 and Cin % 4 == 0:
        # the Generator head (64 -> 3): 4 output columns waste a 32-wide MFMA tile.  16-bit mode: the 16-column MFMA
        # over an LDS halo (conv7_n4.hip); fp32: vector-ALU kernel
        if bf16 and conv7_n4_ok(d, 0, bf16):
            return raw_conv7_n4(d, 0, lp16_twin(x, bf16), w, bias, y, act, bf16)
        L.call("mmh_conv7_thin_fprop", C.byref(d), _ptr(x), _ptr(w), _ptr(bias), _ptr(y), act, _stream())
        _count_desc("valu", d)
        return y
    if bf16 and lp16_v2_ok(Cin, Cout, k, stride, pad, 0):
        timed = fprop_timer is not None and fprop_timer.want(d)
        x16 = lp16_twin(x, bf16)
        if timed:
            e0, e1 = fprop_timer.bracket()
            e0.record()
        y = raw_conv3x3_lp16(x16, w, bias, reflect, act, bf16, 0)
        if timed:
            e1.record()
        return y
    if bf16 and lp16g_ok(d, 0, bf16):
        return raw_conv_lp16g(d, 0, lp16_twin(x, bf16), w, bias, act, bf16)
    if bf16 and lp16_flat_ok(d, bf16):
        timed = fprop_timer is not None and fprop_timer.want(d)
        if timed:
            e0, e1 = fprop_timer.bracket()
            e0.record()
        y = raw_conv_lp16_flat(d, x, w, bias, act, bf16)
        if timed:
            e1.record()
        return y
    if bf16:
        d.dtype = _dt(bf16)
        w = bf16_weights(w, bf16)[1]
    _count_desc("mfma", d)
    if (want_stats and FUSE_NORM_STATS and not bf16 and act == L.ACT_NONE and (d.Ho * d.Wo) % 128 == 0
            and not (fprop_timer is not None and fprop_timer.want(d))):
        chunks = L.load().mmh_conv2d_fprop_stats_chunks(C.byref(d))
        if chunks > 0:
            stats = _empty((B, chunks // B, 3, Cout), x)
            L.call("mmh_conv2d_fprop_stats", C.byref(d), _ptr(x), _ptr(w), _ptr(bias), _ptr(y), _ptr(stats), _stream())
            _park_stats(y, stats)
            return y
    if fprop_timer is not None and fprop_timer.want(d):
        e0, e1 = fprop_timer.bracket()      # HIP events on the launch stream (bench.py roofline)
        e0.record()
        L.call("mmh_conv2d_fprop", C.byref(d), _ptr(x), _ptr(w), _ptr(bias), _ptr(y), act, _stream())
        e1.record()
        return y
    L.call("mmh_conv2d_fprop", C.byref(d), _ptr(x), _ptr(w), _ptr(bias), _ptr(y), act, _stream())
    return y


USE_CONV7_N4 = os.environ.get("MMH_CONV7_N4", "1") != "0"


def conv7_n4_ok(d, mode, lp):
    """7x7 conv with <= 4 output channels from a 64-channel 16-bit tensor on the 16-column MFMA (mmh_conv7_n4_lp16):
    mode 0 the Generator head's fprop, 1 a Discriminator stem's gradient towards its first four input channels"""
    if not (USE_CONV7_N4 and lp):
        return False
    keep = d.dtype
    d.dtype = _dt(lp)
    ok = bool(L.load().mmh_conv7_n4_lp16_supported(C.byref(d), mode))
    d.dtype = keep
    return ok


def raw_conv7_n4(d, mode, x16, w, bias, y, act, lp):
    """y: the fp32 output (mode 0) / dx whose channels [0,4) are written (mode 1)"""
    assert x16.dtype == _wd(lp) and x16.is_contiguous()
    keep = d.dtype
    d.dtype = _dt(lp)
    nbytes = L.load().mmh_conv7_n4_lp16_ws_bytes(C.byref(d), mode)
    ws = torch.empty(nbytes // 4 + 4, dtype=torch.float32, device=x16.device)
    L.call("mmh_conv7_n4_lp16", C.byref(d), mode, _ptr(x16), _ptr(w), _ptr(bias), _ptr(y), act, _ptr(ws), ws.numel() * 4,
           _ptr(zero_page(x16.device)), _stream())
    d.dtype = keep
    _count("mfma", 2.0 * d.B * d.H * d.W * d.kh * d.kw * 64 * 16)      # executed: 16 columns, 4 of them meaningful
    return y


# The Generator head (ReflectionPad2d(3) + Conv2d(64, 3, 7) + Tanh, models/Generator.py:254-259) with a 16-bit input on all
# three passes: fprop mmh_conv7_n4_lp16, input gradient mmh_conv7_head_dgrad_lp16, weight gradient
# mmh_conv7_head_wgrad_lp16 - so the norm in front of it writes 16 bits only and takes a 16-bit gradient, as the 3x3 convs'
# neighbours do (under apex O1 this conv runs in fp16 as well).  MMH_HEAD16=0: fp32 input, fp32 vector-ALU weight gradient.
USE_HEAD16 = os.environ.get("MMH_HEAD16", "1") != "0"


def head16_ok(B, H, W_, Cin, Cout, k, stride, pad, reflect, bf16):
    if not (USE_HEAD16 and USE_THIN and USE_HEAD_DGRAD16 and bf16 and k == 7 and stride == 1 and pad == 3 and reflect
            and Cin == 64 and Cout == 4):
        return False
    d = conv_desc(B, H, W_, Cin, Cout, k, stride, pad, reflect)
    if not conv7_n4_ok(d, 0, bf16):
        return False
    d.dtype = _dt(bf16)
    lib = L.load()
    return bool(lib.mmh_conv7_head_dgrad_lp16_supported(C.byref(d)) and lib.mmh_conv7_head_wgrad_lp16_supported(C.byref(d)))


def raw_head_wgrad16(x16, g, bf16, out=None):
    """x16: the head's 16-bit input [B,H,W,64]; g: fp32 gradient of its output [B,H,W,4] -> dw [7,7,64,4] fp32 (added into `out`)"""
    B, H, W_, Cin = x16.shape
    assert x16.dtype == _wd(bf16) and x16.is_contiguous() and g.dtype == torch.float32 and g.is_contiguous()
    d = conv_desc(B, H, W_, Cin, 4, 7, 1, 3, True)
    d.dtype = _dt(bf16)
    nbytes = L.load().mmh_conv7_head_wgrad_lp16_ws_bytes(C.byref(d))
    ws = torch.empty(nbytes // 4 + 64, dtype=torch.float32, device=g.device)
    off = (-ws.data_ptr()) % 256 // 4
    dw = out if out is not None else torch.empty((7, 7, Cin, 4), dtype=torch.float32, device=g.device)
    L.call("mmh_conv7_head_wgrad_lp16", C.byref(d), _ptr(x16), _ptr(g), _ptr(dw), _ptr(ws[off:]), nbytes, int(out is not None),
           _ptr(zero_page(g.device)), _stream())
    _count("mfma", 2.0 * B * (H + 6) * (W_ + 6) * 49 * 64 * 8)      # executed: 8 columns, 3 of them meaningful
    return dw


def raw_conv_dgrad_thin(dy, w, x_shape, reflect):
    """dgrad of a 7x7 / stride 1 / pad 3 conv for the first 4 input channels only (the others are
    returned as zeros): the Discriminator stems seen from the generated image.  dy: fp32 or 16-bit."""
    _chk16(dy, "dy"); _chk(w, "w")
    B, H, W_, Cin = x_shape
    Cout = w.shape[3]
    d = conv_desc(B, H, W_, Cin, Cout, 7, 1, 3, reflect)
    dx = torch.zeros((B, H, W_, Cin), dtype=torch.float32, device=dy.device)
    if dy.dtype != torch.float32 and conv7_n4_ok(d, 1, _lp_of(dy)):
        return raw_conv7_n4(d, 1, dy, w, None, dx, L.ACT_NONE, _lp_of(dy))
    ws = _ws(L.load().mmh_conv7_thin_dgrad_ws_bytes(C.byref(d)), dy)
    L.call("mmh_conv7_thin_dgrad", C.byref(d), _ptr(dy), _ptr(w), _ptr(dx), _ptr(ws), ws.numel() * 4, _tdt(dy),
           _stream())
    _count("valu", 2.0 * B * H * W_ * 4 * Cout * 49)
    return dx


def _dgrad_takes_dy16(B, H, W_, Cin, Cout, k, stride, pad, reflect, bf16):
    """raw_conv_dgrad reads only the 16-bit dy for this conv (halo / general 16-bit kernels, the four-column kernel)"""
    if not bf16 or _wino_tile(B, H, W_, Cin, Cout, k, stride, pad, bf16, "dgrad"):
        return False
    d = conv_desc(B, H, W_, Cin, Cout, k, stride, pad, reflect)
    return bool(lp16_v2_ok(Cin, Cout, k, stride, pad, 1) or (not reflect and lp16g_ok(d, 1, bf16))
                or (k == 3 and Cin == 4 and conv7_n4_ok(d, 1, bf16)))


def _add_into(dx, addend):
    if addend is None:
        return dx
    _chk(addend, "addend")
    return dx.add_(addend)


def raw_conv_dgrad(dy, w, x_shape, stride, pad, reflect, bf16=False, dx_channels=0, dy16=None, out16=False, addend=None,
                   nbr=None):
    """dx_channels > 0: only the first dx_channels input channels need a gradient (the caller
    ignores the rest, which may come back as zeros).  dy16: an existing 16-bit twin of dy (dy itself
    may then be None); out16: return dx in 16 bits (conv_lp16 path only).  addend (fp32, dx's shape): returns
    dx + addend - in the halo kernel's epilogue where that kernel runs, else by one in-place add.
    nbr (NormBwdSite): dx is the output gradient of that norm - where the halo kernel can, it fills in the norm's backward
    sums (raw_conv3x3_lp16)."""
    if addend is not None:
        assert not out16 and not dx_channels
        _chk(addend, "addend")
        B_, H_, W__, Cin_ = x_shape
        k_, _, _, Cout_ = w.shape
        d_ = conv_desc(B_, H_, W__, Cin_, Cout_, k_, stride, pad, reflect)
        d_.dtype = _dt(bf16)
        fused = (bool(bf16) and lp16_v2_ok(Cin_, Cout_, k_, stride, pad, 1) and not _wino_tile(B_, H_, W__, Cin_, Cout_, k_, stride, pad, bf16, "dgrad")
                 and bool(L.load().mmh_conv3x3_lp16_dgrad_add_supported(C.byref(d_)))
                 # reflect padding off the in-kernel fold (images of one 16 x 16 tile either way): the border terms land after
                 # the kernel, so a fused addend would change the order of the sums - keep (main + border) + addend there
                 and (not reflect or (USE_LP16_FOLD and bool(L.load().mmh_conv3x3_lp16_fold_supported(C.byref(d_))))))
        if not fused:
            return _add_into(raw_conv_dgrad(dy, w, x_shape, stride, pad, reflect, bf16, 0, dy16, False), addend)
    _chk(w, "w")
    B, H, W_, Cin = x_shape
    if (USE_HEAD_DGRAD16 and bf16 and w.shape[0] == 7 and w.shape[3] == 4 and Cin == 64 and reflect and stride == 1 and pad == 3
            and dy is not None and dy.dtype == torch.float32 and not dx_channels):
        # the Generator head's input gradient as a stem-shaped 16-bit convolution (mmh_conv7_head_dgrad_lp16)
        d = conv_desc(B, H, W_, Cin, 4, 7, 1, 3, True)
        d.dtype = _dt(bf16)
        if L.load().mmh_conv7_head_dgrad_lp16_supported(C.byref(d)):
            nbytes = L.load().mmh_conv7_head_dgrad_lp16_ws_bytes(C.byref(d))
            ws = torch.empty(nbytes // 4 + 64, dtype=torch.float32, device=dy.device)
            off = (-ws.data_ptr()) % 256 // 4
            dx = torch.empty((B, H, W_, Cin), dtype=_wd(bf16) if out16 else torch.float32, device=dy.device)
            L.call("mmh_conv7_head_dgrad_lp16", C.byref(d), _ptr(dy), _ptr(w), _ptr(dx), int(out16), _ptr(ws[off:]), nbytes,
                   _ptr(zero_page(dy.device)), _stream())
            _count_desc("mfma", d)
            return dx
    if dy is None or out16:
        assert bf16 and (dy16 is not None or dy is not None) and _dgrad_takes_dy16(B, H, W_, Cin, w.shape[3], w.shape[0],
                                                                                  stride, pad, reflect, bf16)
    if dy is not None:
        _chk(dy, "dy")
    k, _, _, Cout = w.shape
    if (USE_THIN and 0 < dx_channels <= 4 and k == 7 and stride == 1 and pad == 3 and Cout % 4 == 0
            and Cin >= 4):
        return raw_conv_dgrad_thin(dy, w, x_shape, reflect)
    wt = _wino_tile(B, H, W_, Cin, Cout, k, stride, pad, bf16, "dgrad")
    if wt:
        return raw_conv_dgrad_wino(dy, w, x_shape, reflect, wt, bf16=bf16)
    d = conv_desc(B, H, W_, Cin, Cout, k, stride, pad, reflect)
    if bf16 and k == 3 and Cin == 4 and not out16 and conv7_n4_ok(d, 1, bf16):
        # VGG19 conv1_1 (3 -> 64, zero padding) seen from the perceptual loss: four gradient columns - the 16-column MFMA
        # kernel of the 7x7 stems with nine taps (the generic kernel ran this at 7 TFLOP/s)
        dx = torch.empty((B, H, W_, Cin), dtype=torch.float32, device=w.device)
        return raw_conv7_n4(d, 1, dy16 if dy16 is not None else lp16_twin(dy, bf16), w, None, dx, L.ACT_NONE, bf16)
    if bf16 and lp16_v2_ok(Cin, Cout, k, stride, pad, 1):
        # main term (zero-padded correlation with the flipped filter) on the v2 kernel; reflect padding
        # adds the eight border terms (small 16-bit GEMMs + border_add) exactly as the other paths do
        if dy16 is None:
            dy16 = lp16_twin(dy, bf16)
        d.dtype = _dt(bf16)
        if reflect and USE_LP16_FOLD and L.load().mmh_conv3x3_lp16_fold_supported(C.byref(d)):
            # H, W multiples of 16: the halo kernel folds the pad ring's gradient in itself (mode 2), no border call
            return raw_conv3x3_lp16(dy16, w, None, True, L.ACT_NONE, bf16, 2, out16=out16, addend=addend, nbr=nbr)
        # (with reflect padding off the in-kernel fold the border terms land after the kernel: no sums from its epilogue)
        dx = raw_conv3x3_lp16(dy16, w, None, False, L.ACT_NONE, bf16, 1, out16=out16, addend=addend,
                              nbr=None if reflect else nbr)
        if reflect:
            ws = torch.empty(L.load().mmh_conv2d_dgrad_border_ws_bytes(C.byref(d)) // 4 + 4, dtype=torch.float32,
                             device=dx.device)
            # border GEMMs read the 16-bit dy; io16 bit 0 = dy is 16-bit, bit 1 = dx is 16-bit
            L.call("mmh_conv2d_dgrad_border", C.byref(d), _ptr(dy16), _ptr(bf16_weights(w, bf16)[0]), _ptr(dx), _ptr(ws),
                   ws.numel() * 4, 3, 1 | (2 if out16 else 0), _stream())
        return dx
    if bf16 and not reflect and lp16g_ok(d, 1, bf16):
        return raw_conv_lp16g(d, 1, dy16 if dy16 is not None else lp16_twin(dy, bf16), w, None, L.ACT_NONE, bf16,
                              out16=out16)
    if bf16 and Cout % 64 == 0:
        d.dtype = _dt(bf16)
        w = bf16_weights(w, bf16)[0]
    dx = _empty((B, H, W_, Cin), dy)
    nbytes = L.load().mmh_conv2d_dgrad_folded_ws_bytes(C.byref(d))
    ws = _ws(nbytes, dy) if nbytes else None
    L.call("mmh_conv2d_dgrad_folded", C.byref(d), _ptr(dy), _ptr(w), _ptr(dx), _ptr(ws),
           ws.numel() * 4 if ws is not None else 0, _stream())
    _count_desc("mfma", d)
    return dx


def raw_conv_wgrad(x, dy, k, stride, pad, reflect, bf16=False, out=None):
    """x, dy: fp32, or (16-bit mode) tensors already held in 16 bits.  out: add dw into this tensor.
    x may be a channel slice of a wider 16-bit tensor (the gate's concat read in place): the nine-tap wgrad takes the
    pixel stride, every other kernel gets a contiguous copy."""
    B, H, W_, Cin = x.shape
    Cout = dy.shape[3]
    if not x.is_contiguous() and not (bf16 and x.dtype != torch.float32 and lp16_wgrad_ok(Cin, Cout, k, stride, pad)):
        x = x.contiguous()
    if x.is_contiguous():
        _chk16(x, "x")
    _chk16(dy, "dy")
    lp_in = x.dtype != torch.float32 or dy.dtype != torch.float32
    assert bf16 or not lp_in
    if bf16 and lp16_wgrad_ok(Cin, Cout, k, stride, pad):
        return raw_wgrad3x3_lp16(x if x.dtype != torch.float32 else lp16_twin(x, bf16),
                                 dy if dy.dtype != torch.float32 else lp16_twin(dy, bf16), reflect, bf16, out=out)
    if lp_in:
        d = conv_desc(B, H, W_, Cin, Cout, k, stride, pad, reflect)
        c8 = (Cin + 7) // 8 * 8
        if lp16_flat_wgrad_ok(d, c8, bf16):     # flat (tap, channel)-row second-generation kernel
            if x.dtype == torch.float32:
                x = lp16_pad8(x, bf16)
            elif x.shape[3] < c8:
                x = lp16_pad8(x.float(), bf16)
            dy = dy if dy.dtype != torch.float32 else lp16_twin(dy, bf16)
            dd = conv_desc(B, H, W_, Cin, Cout, k, stride, pad, reflect)
            return raw_wgrad_lp16_flat(dd, x, c8, dy, bf16, out=out)
        # first-generation 16-bit wgrad kernel reading the 16-bit tensors directly (both or neither)
        x = x if x.dtype != torch.float32 else lp16_twin(x, bf16)
        dy = dy if dy.dtype != torch.float32 else lp16_twin(dy, bf16)
        d.dtype = _dt(bf16)
        assert (d.Ho, d.Wo) == (dy.shape[1], dy.shape[2])
        ws = _ws(L.load().mmh_conv2d_wgrad_ws_bytes(C.byref(d)), x)
        dw = out if out is not None else _empty((k, k, Cin, Cout), x)
        L.call("mmh_conv2d_wgrad", C.byref(d), _ptr(x), _ptr(dy), _ptr(dw), _ptr(ws), ws.numel() * 4, int(out is not None),
               (1 if x.dtype != torch.float32 else 0) | (2 if dy.dtype != torch.float32 else 0), _stream())
        _count_desc("mfma", d)
        return dw
    wt = _wino_tile(B, H, W_, Cin, Cout, k, stride, pad, bf16, "wgrad")
    if wt:
        return raw_conv_wgrad_wino(x, dy, reflect, wt, bf16=bf16, out=out)
    d = conv_desc(B, H, W_, Cin, Cout, k, stride, pad, reflect)
    if USE_THIN and k == 7 and stride == 1 and pad == 3 and Cout == 4 and Cin % 64 == 0:
        # the Generator head: 4 output columns; fp32 vector-ALU kernel (also under --opt_level O1/O2)
        ws = _ws(L.load().mmh_conv7_thin_wgrad_ws_bytes(C.byref(d)), x)
        dw = out if out is not None else _empty((k, k, Cin, Cout), x)
        L.call("mmh_conv7_thin_wgrad", C.byref(d), _ptr(x), _ptr(dy), _ptr(dw), _ptr(ws), ws.numel() * 4,
               int(out is not None), _stream())
        _count_desc("valu", d)
        return dw
    if USE_STEM_WGRAD and not bf16 and L.load().mmh_conv7_stem_wgrad_supported(C.byref(d)):
        # fp32 7x7 stems (Cin <= 44 -> 64): the input band of a filter row staged in LDS once instead of a
        # per-tap gather from global memory (conv_stem.hip)
        ws = _ws(L.load().mmh_conv7_stem_wgrad_ws_bytes(C.byref(d)), x)
        dw = out if out is not None else _empty((k, k, Cin, Cout), x)
        L.call("mmh_conv7_stem_wgrad", C.byref(d), _ptr(x), _ptr(dy), _ptr(dw), _ptr(ws), ws.numel() * 4,
               int(out is not None), _stream())
        _count_desc("mfma", d)
        return dw
    if bf16:
        d.dtype = _dt(bf16)
    assert (d.Ho, d.Wo) == (dy.shape[1], dy.shape[2])
    nbytes = L.load().mmh_conv2d_wgrad_ws_bytes(C.byref(d))
    ws = _ws(nbytes, x)
    dw = out if out is not None else _empty((k, k, Cin, Cout), x)
    L.call("mmh_conv2d_wgrad", C.byref(d), _ptr(x), _ptr(dy), _ptr(dw), _ptr(ws),
           ws.numel() * 4, int(out is not None), 0, _stream())
    _count_desc("mfma", d)
    return dw


# --------------------------------------------------------------------------- 16-bit direct 3x3 (v2)
# Second-generation 16-bit kernel for the 3x3 / stride 1 / pad 1 stack (conv_lp16.hip): both operands
# 16-bit in HBM, LDS-DMA, 256x256x64 tiles.  Takes a 16-bit twin of the activation (mmh_cvt_lp16).
USE_LP16_V2 = os.environ.get("MMH_LP16_V2", "1") != "0"
# 16-bit mode: the tensors between a conv_lp16 convolution and its norm / gate neighbours (conv outputs,
# and the gradients on both sides) live in HBM in 16 bits only, as apex O1's fp16 conv outputs do.
USE_LP16_EDGES = os.environ.get("MMH_LP16_EDGES", "1") != "0"
_zero_pages = {}


def zero_page(dev):
    z = _zero_pages.get(dev)
    if z is None:
        z = _zero_pages[dev] = torch.zeros(1024, dtype=torch.uint8, device=dev)
    return z


def _pix_stride(t):
    """pixel stride (elements) of an NHWC tensor or of a channel-slice view of one (channels contiguous, pixels `cs`
    apart, rows and images dense in pixels): what the kernels take as x_cs"""
    B, H, W_, Cc = t.shape
    cs = t.stride(2)
    assert t.stride(3) == 1 and cs >= Cc and t.stride(1) == W_ * cs and t.stride(0) == H * W_ * cs, \
        f"not an NHWC tensor or channel slice of one: shape {tuple(t.shape)}, strides {t.stride()}"
    return cs


# the 16-bit twin of a PATBlock's stream-1 input is already in memory: the gate wrote cat(s3, out) in 16 bits, and its
# second half IS out - the halo kernel and the nine-tap wgrad read it in place (pixel stride 2 C) instead of a conversion
USE_LP16_CAT_TWIN = os.environ.get("MMH_LP16_CAT_TWIN", "1") != "0"


def lp16_twin(x, bf16=True):
    """16-bit copy (bf16 / fp16) of an fp32 NHWC tensor."""
    _chk(x, "x")
    out = torch.empty(x.shape, dtype=_wd(bf16), device=x.device)
    L.call("mmh_cvt_lp16", _ptr(x), x.numel(), _dt(bf16), _ptr(out), _stream())
    return out


# A tensor with two consumers - the block input x that both the first conv of a two-conv block and the residual add
# read (PATBlock: out = x1 + s1 * gates, models/Generator.py:115-130; ResnetBlock: out = x + conv_block(x),
# models/Discriminator.py:50) - gets two gradients, which autograd adds in a pass of its own (35 launches, 1.7 ms per
# 16-bit step).  With a ResidualToken shared by the two consumers the residual side's backward (which always runs first:
# the conv's gradient depends on it) parks its gradient in the token instead of returning it, and the conv's dgrad adds it
# in its epilogue (mmh_conv3x3_lp16_dgrad_add) or, on any other path, with one in-place add.
USE_RESIDUAL_TOKENS = os.environ.get("MMH_RESIDUAL_TOKENS", "1") != "0"


class ResidualToken:
    __slots__ = ("armed", "addend", "taken")

    def __init__(self):
        self.armed = False      # set by the conv's forward: it will compute an input gradient
        self.addend = None      # parked by the residual side's backward, taken by the conv's backward
        self.taken = False      # the conv's backward has run: a later park() must go through autograd instead

    def park(self, g):
        """residual side: True = the gradient is parked (return None to autograd).  False once the conv's backward has
        already taken what there was (the residual side ran late): autograd then accumulates the two gradients itself."""
        if not self.armed or g is None or self.taken:
            return False
        assert self.addend is None, "ResidualToken: a gradient is already parked (backward ran twice?)"
        self.addend = g
        return True

    def take(self):
        g, self.addend = self.addend, None
        self.taken = True
        return g


# dgrad of the reflect-padded 3x3 convs: border terms inside the halo kernel (mmh_conv3x3_lp16 mode 2) where it applies
USE_LP16_FOLD = os.environ.get("MMH_LP16_FOLD", "1") != "0"


def lp16_v2_ok(Cin, Cout, k, stride, pad, mode):
    """mode 0 fprop (N = Cout), 1 dgrad (N = Cin)"""
    n = Cout if mode == 0 else Cin
    return USE_LP16_V2 and k == 3 and stride == 1 and pad == 1 and Cin % 64 == 0 and Cout % 64 == 0 and n % 256 == 0


# The first norm of a two-conv block (conv -> norm -> ReLU -> Dropout -> pad -> conv, models/Generator.py:66-77) receives its
# output gradient from the second conv's dgrad: that kernel's epilogue takes the norm's backward sums of the values it stores
# (mmh_conv3x3_lp16_dgrad_nbr), and the norm's reduce pass over g, x and the keep bits is gone.  The norm's forward leaves a
# NormBwdSite under its 16-bit output; the consuming conv picks it up, its backward fills in g / s1 / s2, the norm's backward
# takes them.  MMH_FUSE_NORM_BWD_REDUCE=0: off.
USE_NBR = os.environ.get("MMH_FUSE_NORM_BWD_REDUCE", "1") != "0"
NBR_MIN_C = int(os.environ.get("MMH_NBR_MIN_C", "256"))      # channel counts from which the fused form is used


class NormBwdSite:
    __slots__ = ("x", "bits", "mean", "invstd", "groups", "drop_p", "g", "s1", "s2")

    def __init__(self, x, bits, mean, invstd, groups, drop_p):
        self.x, self.bits, self.mean, self.invstd, self.groups, self.drop_p = x, bits, mean, invstd, groups, drop_p
        self.g = self.s1 = self.s2 = None


_nbr_sites = {}


def nbr_site_put(a16, site):
    while len(_nbr_sites) >= 8:         # bounded: sites whose consumer was not a conv of this kind
        _nbr_sites.pop(next(iter(_nbr_sites)))
    _nbr_sites[a16.data_ptr()] = (a16, site)


def nbr_site_take(a16):
    ent = _nbr_sites.pop(a16.data_ptr(), None)
    return ent[1] if ent is not None and ent[0] is a16 else None


def raw_conv3x3_lp16(x16, w, bias, reflect, act, bf16, mode, out16=False, want_stats=False, addend=None, nbr=None):
    """mode 0: y = conv(x16, w) (+bias, act); mode 1: dx = zero-pad correlation of x16 (= dy) with the
    flipped filter (the caller adds the reflect border terms); mode 2: the complete dgrad of a
    ReflectionPad2d(1) conv (border terms folded in the kernel).  w: the fp32 physical weight.
    nbr (NormBwdSite; dgrad, 16-bit dx): where the kernel can, it also fills in the sums of the norm whose output gradient dx
    is (nbr.g = dx, nbr.s1, nbr.s2)."""
    B, H, W_, Cx = x16.shape
    _, _, Cin, Cout = w.shape
    assert x16.dtype == _wd(bf16) and Cx == (Cin if mode == 0 else Cout)
    xcs = _pix_stride(x16)          # fprop: x16 may be a channel slice of a wider 16-bit tensor
    assert mode == 0 or xcs == Cx, "a strided 16-bit operand is supported for the fprop input only"
    d = conv_desc(B, H, W_, Cin, Cout, 3, 1, 1, reflect, x_cs=xcs if mode == 0 else None)
    d.dtype = _dt(bf16)
    wp, wt = bf16_weights(w, bf16)
    N = Cout if mode == 0 else Cin
    y = torch.empty((B, H, W_, N), dtype=_wd(bf16) if out16 else torch.float32, device=x16.device)
    chunks = (L.load().mmh_conv3x3_lp16_stats_chunks(C.byref(d))
              if (want_stats and FUSE_NORM_STATS and mode == 0 and out16 and act == L.ACT_NONE) else 0)
    if chunks:
        # the InstanceNorm behind this conv merges these partials (raw_norm_stats_finalize_pending) instead of reading y
        stats = torch.empty((B, chunks, 3, N), dtype=torch.float32, device=x16.device)
        L.call("mmh_conv3x3_lp16_fprop_stats", C.byref(d), _ptr(x16), _ptr(wt), _ptr(bias), _ptr(y), _ptr(stats),
               _ptr(zero_page(x16.device)), _stream())
        _park_stats(y, stats)
    elif (nbr is not None and USE_NBR and mode != 0 and out16 and addend is None and tuple(nbr.x.shape) == tuple(y.shape)
          and nbr.x.dtype == y.dtype and nbr.x.is_contiguous()
          and L.load().mmh_conv3x3_lp16_dgrad_nbr_chunks(C.byref(d), mode) > 0):
        cpi = L.load().mmh_conv3x3_lp16_dgrad_nbr_chunks(C.byref(d), mode)
        ws = _ws(B * cpi * 2 * N * 4, y)
        s1 = torch.empty((nbr.groups, N), dtype=torch.float32, device=y.device)
        s2 = torch.empty((nbr.groups, N), dtype=torch.float32, device=y.device)
        L.call("mmh_conv3x3_lp16_dgrad_nbr", C.byref(d), mode, _ptr(x16), _ptr(wp), _ptr(y), _ptr(nbr.x), _ptr(nbr.bits),
               _ptr(nbr.mean), _ptr(nbr.invstd), nbr.groups, float(nbr.drop_p), _ptr(s1), _ptr(s2), _ptr(ws), ws.numel() * 4,
               _ptr(zero_page(x16.device)), _stream())
        nbr.g, nbr.s1, nbr.s2 = y, s1, s2
    elif addend is not None:        # dgrad: dx = dgrad(dy) + addend in the epilogue
        assert mode != 0 and not out16 and bias is None and act == L.ACT_NONE and tuple(addend.shape) == tuple(y.shape)
        L.call("mmh_conv3x3_lp16_dgrad_add", C.byref(d), mode, _ptr(x16), _ptr(wp), _ptr(addend), _ptr(y),
               _ptr(zero_page(x16.device)), _stream())
    else:
        L.call("mmh_conv3x3_lp16", C.byref(d), mode, _ptr(x16), _ptr(wt if mode == 0 else wp), _ptr(bias), _ptr(y),
               int(out16), act, _ptr(zero_page(x16.device)), _stream())
    _count_desc("mfma", d)
    return y


# the same kernel family for the other 3x3 / pad 1 convs (stride 2, ConvTranspose2d, 64 / 128 columns)
USE_LP16_G = os.environ.get("MMH_LP16_G", "1") != "0"


def lp16g_ok(d, mode, bf16):
    """conv_lp16g_kernel takes pass `mode` (0 fprop | 1 dgrad) of the conv described by d"""
    if not (bf16 and USE_LP16_V2 and USE_LP16_G):
        return False
    old = d.dtype
    d.dtype = _dt(bf16)
    ok = bool(L.load().mmh_conv_lp16_supported(C.byref(d), mode))
    d.dtype = old
    return ok


def raw_conv_lp16g(d, mode, x16, w, bias, act, bf16, out16=False, want_stats=False):
    """pass `mode` of conv d on the general 16-bit kernel; x16: the 16-bit input (mode 0) / output
    gradient (mode 1); w: the fp32 physical weight [3,3,Cin,Cout].  want_stats (fprop, 16-bit output, no activation):
    where the stride-2 kernel takes the shape its epilogue leaves the partial statistics for the norm behind the conv."""
    assert x16.dtype == _wd(bf16) and x16.is_contiguous()
    d.dtype = _dt(bf16)
    wp, wt = bf16_weights(w, bf16)
    shape = (d.B, d.Ho, d.Wo, d.Cout) if mode == 0 else (d.B, d.H, d.W, d.Cin)
    y = torch.empty(shape, dtype=_wd(bf16) if out16 else torch.float32, device=x16.device)
    if want_stats and FUSE_NORM_STATS and FUSE_NORM_STATS_NARROW and mode == 0 and out16 and act == L.ACT_NONE:
        chunks = L.load().mmh_conv_lp16_stats_chunks(C.byref(d))
        if chunks > 0:
            stats = torch.empty((d.B, chunks, 3, d.Cout), dtype=torch.float32, device=x16.device)
            L.call("mmh_conv_lp16_fprop_stats", C.byref(d), _ptr(x16), _ptr(wt), _ptr(bias), _ptr(y), _ptr(stats),
                   _ptr(zero_page(x16.device)), _stream())
            _park_stats(y, stats)
            _count_desc("mfma", d)
            return y
    L.call("mmh_conv_lp16", C.byref(d), mode, _ptr(x16), _ptr(wt if mode == 0 else wp), _ptr(bias), _ptr(y),
           int(out16), act, _ptr(zero_page(x16.device)), _stream())
    _count_desc("mfma", d)
    return y


# flat-K 16-bit fprop for the 7x7 stems (conv_lp16f_kernel)
USE_LP16_FLAT = os.environ.get("MMH_LP16_FLAT", "1") != "0"
_flat8_cache = {}


def lp16_flat_ok(d, bf16):
    # 7x7 / 5x5 stems, and 3x3 with <= 8 input channels (VGG19 conv1_1: nine taps x 8 padded channels = 72 -> 128 deep)
    if not (bf16 and USE_LP16_V2 and USE_LP16_FLAT) or d.Cin > 64 or (d.kh < 5 and not (d.kh == 3 and d.Cin <= 8)):
        return False
    old = d.dtype
    d.dtype = _dt(bf16)
    ok = bool(L.load().mmh_conv_lp16_flat_supported(C.byref(d), (d.Cin + 7) // 8 * 8))
    d.dtype = old
    return ok


def flat8_weights(w, bf16):
    """16-bit [Cout][roundup(k*k*C8, 64)] copy of a physical fp32 weight [k,k,Cin,Cout], k = tap*C8 + c"""
    key = (w.data_ptr(), tuple(w.shape), _lp(bf16))
    ent = _cache_get(_flat8_cache, key, w)
    if ent is None:
        k, _, cin, cout = w.shape
        c8 = (cin + 7) // 8 * 8
        kpad = (k * k * c8 + 63) // 64 * 64
        out = torch.empty((cout, kpad), dtype=_wd(bf16), device=w.device)
        L.call("mmh_prep_weights_lp16_flat8", _ptr(w), k * k, cin, cout, c8, _dt(bf16), _ptr(out), _stream())
        ent = _cache_put(_flat8_cache, key, w, out)
    return ent


# The stems' padded 16-bit inputs straight from the pack kernel (mmh_pack_nhwc_lp16 writes them beside - or instead of - the
# fp32 NHWC tensor): raw_pack(twin=lp) parks the copy here under the fp32 tensor's address and version, lp16_pad8 takes it
# instead of running mmh_lp16_pad_cvt over the tensor.  MMH_PACK_TWIN=0: off.
USE_PACK_TWIN = os.environ.get("MMH_PACK_TWIN", "1") != "0"
_pack_twins = {}


def pack_twin_put(x, x16p, keep=False):
    """x16p is what lp16_pad8(x) would return; x may be a zero-stride proxy (lp_proxy) that is never read.  keep: the entry
    survives its use (a model input that several steps read; dropped by pack_twin_drop), else the first lp16_pad8 takes it."""
    while len(_pack_twins) >= 16:       # bounded: entries of tensors that never reached a stem
        _pack_twins.pop(next(iter(_pack_twins)))
    # a DETACHED alias is held (same storage, same version counter): the address stays unique while the entry lives, and an
    # entry no stem consumes does not pin the autograd graph of the PackFn output it was parked under
    _pack_twins[x.data_ptr()] = (x.detach(), x._version, x16p, keep)


def pack_twin_drop(x):
    if x is not None:
        _pack_twins.pop(x.data_ptr(), None)


def pack_twin_get(x, bf16, pop=True):
    ent = _pack_twins.get(x.data_ptr())
    if ent is None:
        return None
    x0, ver, x16p, keep = ent
    B, H, W_, Cc = x.shape
    ok = ((x0 is x or (x0.shape == x.shape and x0.stride() == x.stride())) and x0._version == ver and x._version == ver
          and x16p.dtype == _wd(bf16) and tuple(x16p.shape) == (B, H, W_, (Cc + 7) // 8 * 8))
    if not ok or (pop and not keep):
        del _pack_twins[x.data_ptr()]
    return x16p if ok else None


def lp16_pad8(x, bf16, out=None):
    """fp32 NHWC [B,H,W,C] -> 16-bit [B,H,W,C8], channels zero-padded to a multiple of 8 (out: written there)"""
    B, H, W_, Cc = x.shape
    c8 = (Cc + 7) // 8 * 8
    twin = pack_twin_get(x, bf16, pop=out is None) if USE_PACK_TWIN else None
    if twin is not None:
        if out is None:
            return twin
        out.copy_(twin)
        return out
    _chk(x, "x")
    x16p = out if out is not None else torch.empty((B, H, W_, c8), dtype=_wd(bf16), device=x.device)
    assert x16p.is_contiguous() and x16p.dtype == _wd(bf16) and tuple(x16p.shape) == (B, H, W_, c8)
    L.call("mmh_lp16_pad_cvt", _ptr(x), B * H * W_, Cc, c8, _dt(bf16), _ptr(x16p), _stream())
    return x16p


# flat (tap, channel)-row 16-bit wgrad (wgrad_lp16f_kernel): stems, stride-2 convs, ConvTranspose2d
USE_LP16_FLAT_WGRAD = os.environ.get("MMH_LP16_FLAT_WGRAD", "1") != "0"


def lp16_flat_wgrad_ok(d, c8, bf16, any_cin=False):
    """the flat-row second-generation wgrad kernel takes conv d - and pays: with fewer than 64 channels per
    tap (the 7x7 stems, 64 output columns) its DMA moves the im2col matrix for a quarter of the MFMA work
    per byte and the first-generation kernel is 15-40 % faster (any_cin: ask for support only)"""
    if not (bf16 and USE_LP16_V2 and USE_LP16_FLAT_WGRAD) or (d.Cin < 64 and not any_cin):
        return False
    old = d.dtype
    d.dtype = _dt(bf16)
    ok = bool(L.load().mmh_wgrad_lp16_flat_supported(C.byref(d), c8))
    d.dtype = old
    return ok


def raw_wgrad_lp16_flat(d, x16, c8, dy16, bf16, out=None):
    """dw [k,k,d.Cin,d.Cout] fp32 of conv d from the 16-bit gathered tensor x16 [B,H,W,>=c8] (c8 channels per tap)
    and the 16-bit per-pixel tensor dy16 [B,Ho,Wo,Cout]; out: add into this tensor."""
    assert x16.dtype == _wd(bf16) and dy16.dtype == _wd(bf16) and x16.is_contiguous() and dy16.is_contiguous()
    d.dtype = _dt(bf16)
    ws = torch.empty(max(int(L.load().mmh_wgrad_lp16_flat_ws_bytes(C.byref(d), c8)), 16) // 4, dtype=torch.float32,
                     device=x16.device)
    dw = out if out is not None else torch.empty((d.kh, d.kw, d.Cin, d.Cout), dtype=torch.float32, device=x16.device)
    L.call("mmh_wgrad_lp16_flat", C.byref(d), _ptr(x16), c8, x16.shape[3], _ptr(dy16), _ptr(dw), _ptr(ws), ws.numel() * 4,
           int(out is not None), _ptr(zero_page(x16.device)), _stream())
    _count_desc("mfma", d)
    return dw


USE_STEM_WGRAD16 = os.environ.get("MMH_STEM_WGRAD16", "1") != "0"


def stem_wgrad16_ok(d, c8, bf16):
    if not (USE_STEM_WGRAD16 and bf16):
        return False
    keep = d.dtype
    d.dtype = _dt(bf16)
    ok = bool(L.load().mmh_wgrad_stem_lp16_supported(C.byref(d), c8))
    d.dtype = keep
    return ok


def raw_wgrad_stem_lp16(d, x16p, dy16, bf16, out=None):
    """wgrad of a 7x7 stem on wgrad_stem.hip; x16p [B,H,W,C8] the stem's padded 16-bit input, dy16 [B,H,W,64]"""
    c8 = x16p.shape[3]
    assert x16p.dtype == _wd(bf16) and dy16.dtype == _wd(bf16) and x16p.is_contiguous() and dy16.is_contiguous()
    keep = d.dtype
    d.dtype = _dt(bf16)
    nbytes = L.load().mmh_wgrad_stem_lp16_ws_bytes(C.byref(d), c8)
    ws = torch.empty(nbytes // 4 + 4, dtype=torch.float32, device=x16p.device)
    dw = out if out is not None else torch.empty((7, 7, d.Cin, d.Cout), dtype=torch.float32, device=x16p.device)
    L.call("mmh_wgrad_stem_lp16", C.byref(d), _ptr(x16p), c8, _ptr(dy16), _ptr(dw), _ptr(ws), ws.numel() * 4,
           int(out is not None), _ptr(zero_page(x16p.device)), _stream())
    d.dtype = keep
    _count_desc("mfma", d)
    return dw


def raw_conv_wgrad_lp16_gen1(x16, dy16, Cin, k, stride, pad, reflect, bf16, out=None):
    """first-generation 16-bit wgrad kernel on 16-bit tensors; x16 [B,H,W,x_cs >= Cin] (a stem's padded input)"""
    B, H, W_, xcs = x16.shape
    Cout = dy16.shape[3]
    d = conv_desc(B, H, W_, Cin, Cout, k, stride, pad, reflect, x_cs=xcs)
    d.dtype = _dt(bf16)
    ws = _ws(L.load().mmh_conv2d_wgrad_ws_bytes(C.byref(d)), x16)
    dw = out if out is not None else torch.empty((k, k, Cin, Cout), dtype=torch.float32, device=x16.device)
    L.call("mmh_conv2d_wgrad", C.byref(d), _ptr(x16), _ptr(dy16), _ptr(dw), _ptr(ws), ws.numel() * 4, int(out is not None),
           3, _stream())
    _count_desc("mfma", d)
    return dw


USE_STEM_FPROP16 = os.environ.get("MMH_STEM_FPROP16", "1") != "0"
_stem16_cache = {}


def stem16_weights(w, bf16):
    """the stem's weight [k,k,Cin,64] (k = 7, or 3 at C8 = 8) for conv_stem16.hip: 16-bit [k][64][32 ceil(k C8 / 32) + 8]
    (mmh_prep_weights_stem16_k)"""
    key = (w.data_ptr(), tuple(w.shape), _lp(bf16))
    ent = _cache_get(_stem16_cache, key, w)
    if ent is None:
        ks, cin = w.shape[0], w.shape[2]
        c8 = (cin + 7) // 8 * 8
        out = torch.empty(L.load().mmh_conv_stem16_weights_bytes_k(c8, ks) // 2, dtype=_wd(bf16), device=w.device)
        L.call("mmh_prep_weights_stem16_k", _ptr(w), cin, c8, ks, _dt(bf16), _ptr(out), _stream())
        ent = _cache_put(_stem16_cache, key, w, out)
    return ent


# VGG19's conv1_1 (3 -> 64, 3x3, zero padding, full resolution) on the stem kernel's 3x3 form instead of the flat-K kernel
# (72-deep contraction padded to 128, an im2col tile gathered per k-step).  MMH_STEM3=0: the flat-K kernel.
USE_STEM3 = os.environ.get("MMH_STEM3", "1") != "0"


def raw_conv_lp16_flat(d, x, w, bias, act, bf16, out16=False, x16p=None, want_stats=False):
    """fprop of a small-Cin 'same' conv on the flat-K 16-bit kernel; x: fp32 NHWC [B,H,W,Cin] (or its padded
    16-bit copy x16p = lp16_pad8(x)).  The 7x7 stems with 64 output channels take conv_stem16.hip (input halo resident
    in LDS, column taps flattened into the contraction) instead."""
    c8 = (d.Cin + 7) // 8 * 8
    if x16p is None:
        x16p = lp16_pad8(x, bf16)
    d.dtype = _dt(bf16)
    y = torch.empty((d.B, d.H, d.W, d.Cout), dtype=_wd(bf16) if out16 else torch.float32, device=x16p.device)
    if (USE_STEM_FPROP16 and (d.kh == 7 or (d.kh == 3 and USE_STEM3)) and w.shape[3] == 64
            and L.load().mmh_conv_stem16_supported(C.byref(d), c8)):
        chunks = (L.load().mmh_conv_stem16_stats_chunks(C.byref(d), c8)
                  if (want_stats and FUSE_NORM_STATS and FUSE_NORM_STATS_NARROW and out16 and act == L.ACT_NONE) else 0)
        if chunks > 0:      # the InstanceNorm behind the stem merges these partials instead of reading y
            stats = torch.empty((d.B, chunks, 3, d.Cout), dtype=torch.float32, device=x16p.device)
            L.call("mmh_conv_stem16_stats", C.byref(d), _ptr(x16p), c8, _ptr(stem16_weights(w, bf16)), _ptr(bias), _ptr(y),
                   _ptr(stats), _ptr(zero_page(x16p.device)), _stream())
            _park_stats(y, stats)
            _count_desc("mfma", d)
            return y
        L.call("mmh_conv_stem16", C.byref(d), _ptr(x16p), c8, _ptr(stem16_weights(w, bf16)), _ptr(bias), _ptr(y),
               int(out16), act, _ptr(zero_page(x16p.device)), _stream())
        _count_desc("mfma", d)
        return y
    L.call("mmh_conv_lp16_flat", C.byref(d), _ptr(x16p), c8, _ptr(flat8_weights(w, bf16)), _ptr(bias), _ptr(y),
           int(out16), act, _ptr(zero_page(x16p.device)), _stream())
    _count_desc("mfma", d)
    return y


def stem_lp16_ok(d, bf16, dx_channels):
    """a 7x7 stem may hand its output over in 16 bits: flat-K fprop, and its input gradient is either
    not needed or only that of the first <= 4 channels (thin dgrad)"""
    return (lp16_flat_ok(d, bf16) and d.kh == 7 and d.stride == 1 and d.pad == 3 and USE_THIN
            and 0 <= dx_channels <= 4 and d.Cout % 4 == 0)


def lp16_chain_ok(Cin, Cout, k, stride, pad, reflect=True, bf16=True):
    """All three passes of this conv read and write 16-bit tensors directly (fprop and dgrad on
    conv_lp16.hip, wgrad there or on the first-generation kernel with 16-bit sources): its neighbours may
    hand x over, and take y, in 16 bits.  reflect: the padding mode (decides the dgrad kernel)."""
    if not (USE_LP16_V2 and k == 3 and pad == 1):
        return False
    if (lp16_v2_ok(Cin, Cout, k, stride, pad, 0) and lp16_v2_ok(Cin, Cout, k, stride, pad, 1)
            and lp16_wgrad_ok(Cin, Cout, k, stride, pad)):
        return True
    d = conv_desc(1, 8, 8, Cin, Cout, k, stride, pad, reflect)
    fprop = lp16_v2_ok(Cin, Cout, k, stride, pad, 0) or lp16g_ok(d, 0, bf16)
    dgrad = lp16_v2_ok(Cin, Cout, k, stride, pad, 1) or (not reflect and lp16g_ok(d, 1, bf16))
    return fprop and dgrad and Cin % 4 == 0 and Cout % 4 == 0


def convT_lp16_ok(CinT, CoutT, bf16=True):
    """ConvTranspose2d(k3,s2,p1,op1) with 16-bit tensors on all three passes"""
    d = conv_desc(1, 8, 8, CoutT, CinT, 3, 2, 1, False)
    return USE_LP16_V2 and lp16g_ok(d, 0, bf16) and lp16g_ok(d, 1, bf16)


def _tdt(t):
    """mmh dtype code of a tensor's element type"""
    return {torch.float32: L.F32, torch.bfloat16: L.BF16, torch.float16: L.FP16}[t.dtype]


# Gradients of 16-bit edges.  autograd validates a gradient against the fp32 proxy on the edge, so the
# producer of a 16-bit gradient returns a fresh proxy and parks the real tensor here under the proxy's
# address; the consumer (the backward of the node on the other side of the edge) takes it out.  An edge
# with two consumers would make autograd add the proxies - the lookup then fails loudly.
_lp_grads = {}


def lp_grad_out(t16):
    p = lp_proxy(t16.shape, t16.device)
    _lp_grads[p.data_ptr()] = (p, t16)        # p is held so that its address stays unique
    return p


def lp_grad_in(g, what):
    ent = _lp_grads.pop(g.data_ptr(), None) if g is not None else None
    if ent is None or tuple(ent[1].shape) != tuple(g.shape):
        raise RuntimeError(f"{what}: the gradient of a 16-bit edge did not arrive through the 16-bit channel "
                           "(an edge with more than one consumer, or a foreign backward node)")
    return ent[1]


def lp_grads_reset():
    """drop gradients parked by a backward pass that did not finish"""
    _lp_grads.clear()
    _nb_defer.clear()
    _nbr_sites.clear()
    for k in [k for k, ent in _pack_twins.items() if not ent[3]]:      # kept entries belong to model inputs (pack_twin_drop)
        del _pack_twins[k]


# Norm-apply fused into the neighbouring convolutions (fp32, Winograd F(6x6,3x3) stack; models/Generator.py:66-77
# conv -> norm -> ReLU -> Dropout -> pad -> conv).  Forward: NormActFn(defer) only finalises the statistics
# and returns a NormDefer; the consuming Conv2dFn applies scale / shift / ReLU / dropout inside its input
# transform.  Backward: NormActFn runs the reduction pass and (defer == 2) parks a NormBwdDefer under the
# proxy gradient it returns; the producing Conv2dFn applies it inside its backward transform - or
# materialises it with mmh_norm_bwd_apply_rc where that path does not apply.  MMH_FUSE_NORMACT=0: off.
USE_NORM_FUSION = os.environ.get("MMH_FUSE_NORMACT", "1") != "0"


class NormDefer:
    """a normalised activation that exists only as (conv output x, scale, shift, dropout bits)"""
    __slots__ = ("x", "scale", "shift", "groups", "relu", "drop_p", "drows")

    def __init__(self, x, scale, shift, groups, relu, drop_p, drows):
        self.x, self.scale, self.shift, self.groups = x, scale, shift, groups
        self.relu, self.drop_p, self.drows = relu, drop_p, drows


class NormBwdDefer:
    """the gradient of a norm's input as (gradient g of its output, its input x, sums s1 / s2, ...)"""
    __slots__ = ("g", "x", "mean", "invstd", "gamma", "s1", "s2", "count", "scale", "shift", "dbits", "drows", "groups",
                 "rows", "relu", "drop_p")


_nb_defer = {}


def norm_bwd_defer_out(ent):
    p = lp_proxy(ent.x.shape, ent.x.device)
    _nb_defer[p.data_ptr()] = (p, ent)
    return p


def norm_bwd_defer_in(g, what):
    ent = _nb_defer.pop(g.data_ptr(), None) if g is not None else None
    if ent is None or tuple(ent[1].x.shape) != tuple(g.shape):
        raise RuntimeError(f"{what}: expected the deferred gradient of a fused norm (NormBwdDefer) on this edge")
    return ent[1]


def raw_norm_bwd_apply_rc(e):
    """materialise a NormBwdDefer: dx of the norm's input (fp32)"""
    dx = torch.empty_like(e.x)
    L.call("mmh_norm_bwd_apply_rc", _ptr(e.g), _ptr(e.x), _ptr(e.mean), _ptr(e.invstd), _ptr(e.gamma), _ptr(e.s1),
           _ptr(e.s2), _ptr(e.scale), _ptr(e.shift), _ptr(e.dbits), float(e.count), e.groups, e.rows, e.x.shape[3],
           int(e.relu), float(e.drop_p), _ptr(dx), _stream())
    return dx


def raw_dropout_bits(shape, drop_p, seed, mask, device, rows=False):
    """uint8 [numel/8]: bit e of byte i = element 8 i + e is kept (mmh_dropout_bits).  rows: also the same
    decisions as int32 row words [B*H, ceil(W/32), C] (mmh_dropout_bits_rows) -> (bits, rows)"""
    n = 1
    for d in shape:
        n *= d
    bits = torch.empty((n // 8,), dtype=torch.uint8, device=device)
    if not rows:
        L.call("mmh_dropout_bits", n, float(drop_p), seed, _ptr(mask), _ptr(bits), _stream())
        return bits
    B, H, W_, Cc = shape
    drows = torch.empty((B * H, (W_ + 31) // 32, Cc), dtype=torch.int32, device=device)
    L.call("mmh_dropout_bits_both", B * H, W_, Cc, float(drop_p), seed, _ptr(mask), _ptr(bits), _ptr(drows), _stream())
    return bits, drows


def norm_fusion_ok(C):
    """channel counts the fused / decide-again norm kernels take (C/8 a power of two <= 256)"""
    c8 = C // 8
    return USE_NORM_FUSION and C % 8 == 0 and 1 <= c8 <= 256 and (c8 & (c8 - 1)) == 0


# Weight / bias gradients written straight into the parameter's .grad view of the flat gradient buffer
# (the wgrad kernels' own accumulate path) instead of returning a temporary that autograd's AccumulateGrad
# adds in a separate kernel (239 add launches per step).  Off by default: MMHandModel turns it on for
# single-process training; under data parallelism the bucket hooks hang on AccumulateGrad.
ACCUM_PARAM_GRADS = False
# Conv biases that feed an InstanceNorm have an identically zero gradient (the norm removes the channel
# mean): autograd - the reference's and ours - produces rounding noise for them, which Adam turns into
# +-lr steps no output can see.  1: return the exact zero without the reduction pass (62 colsum launches);
# MMH_NULL_BIAS_GRAD=compute restores the literal behaviour.
EXACT_NULL_BIAS_GRAD = os.environ.get("MMH_NULL_BIAS_GRAD", "exact") != "compute"


def _grad_target(p):
    """p.grad when gradients are to be accumulated in place (see ACCUM_PARAM_GRADS), else None"""
    if ACCUM_PARAM_GRADS and p is not None and p.requires_grad and p.grad is not None and p.grad.is_contiguous() \
            and p.grad.dtype == torch.float32 and p.grad.shape == p.shape:
        return p.grad
    return None


# Data parallelism with in-place parameter gradients: the bucket all-reduce of dp.GradBuckets has to know when a
# parameter's gradient is complete, and an in-place accumulate never reaches autograd's AccumulateGrad (whose hook tells
# it otherwise).  The conv shims therefore report every use of a trainable parameter in their forward and every finished
# contribution - added in place, or none to add - in their backward.  No GradBuckets object alive: two list tests per conv.
def _dp_use(ctx_needs, w, bias):
    from . import dp
    if dp.tracking():
        if ctx_needs[1]:
            dp.param_use(w)
        if bias is not None and ctx_needs[2]:
            dp.param_use(bias)


def _dp_done(ctx_needs, w, bias, dw, db):
    from . import dp
    if dp.tracking():
        if ctx_needs[1] and dw is None:
            dp.param_done(w)
        if bias is not None and ctx_needs[2] and db is None:
            dp.param_done(bias)


DP_NO_COMM = os.environ.get("MMH_DP_NO_COMM") == "1"       # see dp.NO_COMM: collectives stubbed out (timing aid only)


def _finish_param_grad(g, target):
    """what a backward returns for a parameter whose gradient was added into `target` in place"""
    return None if target is not None else g


def lp_proxy(shape, device):
    """fp32 tensor of the given logical shape over ONE element of storage (all strides 0): the autograd
    edge of a tensor whose data travel in 16 bits beside it.  autograd casts a gradient to its input's
    dtype, so a 16-bit tensor on the edge would add two conversion passes to the backward; the proxy
    keeps the edge fp32 and costs 4 bytes."""
    return torch.empty_strided(tuple(shape), (0,) * len(shape), dtype=torch.float32, device=device)


def lp16_wgrad_ok(Cin, Cout, k, stride, pad):
    return USE_LP16_V2 and k == 3 and stride == 1 and pad == 1 and Cin % 256 == 0 and Cout % 256 == 0


def raw_wgrad3x3_lp16(x16, dy16, reflect, bf16, out=None):
    """dw [3,3,Cin,Cout] fp32 from the 16-bit twins of x and dy (conv_lp16.hip wgrad).  out: add into
    this tensor instead of returning a new one."""
    B, H, W_, Cin = x16.shape
    Cout = dy16.shape[3]
    assert x16.dtype == _wd(bf16) and dy16.dtype == _wd(bf16) and dy16.is_contiguous()
    d = conv_desc(B, H, W_, Cin, Cout, 3, 1, 1, reflect, x_cs=_pix_stride(x16))     # x16: possibly a channel slice
    d.dtype = _dt(bf16)
    ws = torch.empty(max(int(L.load().mmh_wgrad3x3_lp16_ws_bytes(C.byref(d))), 16) // 4, dtype=torch.float32,
                     device=x16.device)
    dw = out if out is not None else torch.empty((3, 3, Cin, Cout), dtype=torch.float32, device=x16.device)
    L.call("mmh_wgrad3x3_lp16", C.byref(d), _ptr(x16), _ptr(dy16), _ptr(dw), _ptr(ws), ws.numel() * 4,
           int(out is not None), _ptr(zero_page(x16.device)), _stream())
    _count_desc("mfma", d)
    return dw


def _convT_desc(x, w):
    """ConvTranspose2d(k3,s2,p1,op1) seen as the dgrad of a stride-2 conv."""
    B, h, w_, CinT = x.shape
    k, _, CoutT, wc = w.shape   # physical [kh, kw, Cout_T, Cin_T]
    assert wc == CinT and k == 3
    return conv_desc(B, 2 * h, 2 * w_, CoutT, CinT, 3, 2, 1, False)


def raw_convT_fprop(x, w, bias, act=L.ACT_NONE, bf16=False, out16=False):
    """x: fp32 or (16-bit mode) 16-bit; out16: 16-bit output (conv_lp16 path only)"""
    _chk16(x, "x"); _chk(w, "w")
    d = _convT_desc(x, w)
    if bf16 and lp16g_ok(d, 1, bf16):       # the adjoint of a stride-2 conv = its dgrad
        return raw_conv_lp16g(d, 1, x if x.dtype != torch.float32 else lp16_twin(x, bf16), w, bias, act, bf16,
                              out16=out16)
    assert x.dtype == torch.float32 and not out16
    y = _empty((d.B, d.H, d.W, d.Cin), x)
    if bf16 and d.Cout % 64 == 0:
        d.dtype = _dt(bf16)
        w = bf16_weights(w, bf16)[0]
    L.call("mmh_convT2d_fprop", C.byref(d), _ptr(x), _ptr(w), _ptr(bias), _ptr(y), d.Cin, act,
           _stream())
    _count_desc("mfma", d)
    return y


def raw_convT_dgrad(dy, w, x_shape, bf16=False, out16=False):
    _chk16(dy, "dy")
    B, h, w_, CinT = x_shape
    d = conv_desc(B, 2 * h, 2 * w_, w.shape[2], CinT, 3, 2, 1, False)
    if bf16 and lp16g_ok(d, 0, bf16):
        return raw_conv_lp16g(d, 0, dy if dy.dtype != torch.float32 else lp16_twin(dy, bf16), w, None, L.ACT_NONE,
                              bf16, out16=out16)
    assert dy.dtype == torch.float32 and not out16
    dx = _empty((B, h, w_, CinT), dy)
    if bf16 and d.Cin % 64 == 0:
        d.dtype = _dt(bf16)
        w = bf16_weights(w, bf16)[1]
    L.call("mmh_convT2d_dgrad", C.byref(d), _ptr(dy), _ptr(w), _ptr(dx), _stream())
    _count_desc("mfma", d)
    return dx


def raw_convT_wgrad(x, dy, bf16=False, out=None):
    """x, dy: fp32, or (16-bit mode) tensors already held in 16 bits.  out: add dw into this tensor."""
    _chk16(x, "x"); _chk16(dy, "dy")
    assert bf16 or (x.dtype == torch.float32 and dy.dtype == torch.float32)
    B, h, w_, CinT = x.shape
    CoutT = dy.shape[3]
    d = conv_desc(B, 2 * h, 2 * w_, CoutT, CinT, 3, 2, 1, False)
    if bf16:
        d.dtype = _dt(bf16)
    if bf16 and (x.dtype != torch.float32 or dy.dtype != torch.float32) and lp16_flat_wgrad_ok(d, CoutT, bf16):
        # flat-row second-generation kernel: the output gradient is the gathered tensor, x the per-pixel one
        x = x if x.dtype != torch.float32 else lp16_twin(x, bf16)
        dy = dy if dy.dtype != torch.float32 else lp16_twin(dy, bf16)
        dd = conv_desc(B, 2 * h, 2 * w_, CoutT, CinT, 3, 2, 1, False)
        return raw_wgrad_lp16_flat(dd, dy, CoutT, x, bf16, out=out)
    if x.dtype != dy.dtype:         # the kernel takes both tensors in 16 bits or neither
        x = x if x.dtype != torch.float32 else lp16_twin(x, bf16)
        dy = dy if dy.dtype != torch.float32 else lp16_twin(dy, bf16)
    nbytes = L.load().mmh_conv2d_wgrad_ws_bytes(C.byref(d))
    ws = _ws(nbytes, x)
    dw = out if out is not None else _empty((3, 3, CoutT, CinT), x)
    L.call("mmh_convT2d_wgrad", C.byref(d), _ptr(x), _ptr(dy), _ptr(dw), _ptr(ws),
           ws.numel() * 4, int(out is not None),
           (1 if x.dtype != torch.float32 else 0) | (2 if dy.dtype != torch.float32 else 0), _stream())
    _count_desc("mfma", d)
    return dw


def raw_colsum(x2d_rows, Ccols, x, out=None):
    """x: fp32 or 16-bit; the sums are fp32.  out: add into this tensor."""
    ws = torch.empty(L.load().mmh_colsum_ws_bytes(x2d_rows, Ccols) // 4 + 4, dtype=torch.float32, device=x.device)
    acc = out is not None
    if out is None:
        out = torch.empty((Ccols,), dtype=torch.float32, device=x.device)
    L.call("mmh_colsum", _ptr(x), x2d_rows, Ccols, Ccols, _ptr(out), _ptr(ws), ws.numel() * 4, int(acc), _tdt(x),
           _stream())
    return out


def raw_act_bwd_lp16(g, y, act, lp):
    """16-bit(g * act'(y)) in one pass (mmh_act_bwd_lp16)"""
    _chk(g, "g"); _chk(y, "y")
    out = torch.empty(g.shape, dtype=_wd(lp), device=g.device)
    L.call("mmh_act_bwd_lp16", _ptr(g), _ptr(y), g.numel(), act, _dt(lp), _ptr(out), _stream())
    return out


def raw_act_bwd(g, y, act):
    g = g.contiguous()
    dx = torch.empty_like(g)
    L.call("mmh_act_bwd", _ptr(g), _ptr(y), _ptr(dx), g.numel(), act, _stream())
    return dx


def _tok_add(dx, addend):
    """Conv2dFn.backward: the residual side's parked gradient that no kernel epilogue took, added in place"""
    if addend is None:
        return dx
    assert dx is not None and dx.dtype == torch.float32, "a parked residual gradient needs an fp32 input gradient to join"
    return dx.add_(addend)


class Conv2dFn(torch.autograd.Function):
    """nn.Conv2d (+ReflectionPad2d, +bias, +ReLU/Tanh epilogue) on the implicit-GEMM kernels."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, pad, reflect, act, bf16=False, dx_channels=0, x16=None, y_lp=False,
                null_bias_grad=False, pro=None, g_defer=False, res_tok=None, x_twin=None):
        """x16: the producer already wrote x in 16 bits (NormActFn out_lp / GateFn cat_lp); x is then
        the zero-stride fp32 proxy that carries the autograd edge (lp_proxy) and is never read.
        y_lp: hand the output over in 16 bits only -> returns (proxy, y16); the consumer (NormActFn /
        GateFn) sends the gradient back in 16 bits too (lp_grad_out)."""
        ctx.set_materialize_grads(False)    # no full-size zero "gradient" for the non-differentiable 16-bit output
        _dp_use(ctx.needs_input_grad, w, bias)
        # res_tok (ResidualToken): x has a second consumer whose gradient this conv's backward adds to its own
        ctx.res_tok = res_tok
        if res_tok is not None:
            assert x16 is None and pro is None, "a residual token belongs to an fp32 block input"
            res_tok.armed = bool(ctx.needs_input_grad[0])
            res_tok.taken = False
        # pro (NormDefer): x is the proxy of a normalised activation that was never written; the norm-apply runs
        # inside this conv's input transform.  g_defer: the norm behind this conv sends its input gradient
        # as a NormBwdDefer (see USE_NORM_FUSION)
        ctx.g_defer = bool(g_defer)
        # null_bias_grad: the output feeds an InstanceNorm directly - the bias gradient is identically zero
        ctx.skip_db = bool(null_bias_grad and EXACT_NULL_BIAS_GRAD)
        ctx.bias_p = bias
        B, H, W_, Cin = x.shape
        ctx.dx_channels = dx_channels
        wt = _wino_tile(B, H, W_, Cin, w.shape[3], w.shape[0], stride, pad, bf16)
        ctx.cfg = (stride, pad, reflect, act, bias is not None, bf16)
        ctx.x_shape = tuple(x.shape)
        ctx.wino_V = 0
        ctx.lp16 = False
        ctx.x_lp = x16 is not None
        ctx.y_lp = bool(y_lp)
        k = w.shape[0]
        ctx.stem16 = ctx.head16 = False
        ctx.nbr = nbr_site_take(x16) if (x16 is not None and torch.is_tensor(x16)) else None
        if pro is not None:
            assert wt == 6 and not bf16 and x16 is None and not y_lp, "a deferred norm needs the fp32 F(6x6,3x3) path"
            if KEEP_WINOGRAD_INPUT and ctx.needs_input_grad[1]:
                y, V = raw_conv_fprop_wino(x, w, bias, reflect, act, wt, keep_V=True, pro=pro)
                ctx.wino_V = wt
                ctx.save_for_backward(V, w, y if act != L.ACT_NONE else None)
                return y
            assert not ctx.needs_input_grad[1], "a deferred norm in front of a trainable conv needs MMH_WINOGRAD_KEEP_INPUT=1"
            y = raw_conv_fprop_wino(x, w, bias, reflect, act, wt, pro=pro)
            ctx.save_for_backward(None, w, y if act != L.ACT_NONE else None)
            return y
        chain = bool(bf16) and lp16_chain_ok(Cin, w.shape[3], k, stride, pad, reflect, bf16)
        head16 = bool(bf16) and k == 7 and not y_lp and not dx_channels and pro is None \
            and head16_ok(B, H, W_, Cin, w.shape[3], k, stride, pad, reflect, bf16)
        if x16 is not None:
            assert (chain or head16) and x16.dtype == _wd(bf16) and tuple(x16.shape) == tuple(x.shape), \
                "a 16-bit input needs 16-bit kernels for all three passes"
        ctx.stem16 = False
        ctx.head16 = head16
        if head16:
            # the Generator head: the 16-bit input (the producing norm's own output, or a twin of x) serves all three passes
            if x16 is None:
                x16 = lp16_twin(x, bf16)
            d = conv_desc(B, H, W_, Cin, w.shape[3], k, stride, pad, reflect)
            y = torch.empty((B, H, W_, w.shape[3]), dtype=torch.float32, device=x16.device)
            raw_conv7_n4(d, 0, x16, w, bias, y, act, bf16)
            ctx.save_for_backward(x16, w, y if act != L.ACT_NONE else None)
            return y
        if y_lp and not chain:
            # a 7x7 stem handing its output over in 16 bits: flat-K fprop with a 16-bit epilogue; the
            # backward takes the 16-bit gradient into the thin dgrad, the wgrad and the bias sum
            d = conv_desc(B, H, W_, Cin, w.shape[3], k, stride, pad, reflect)
            assert bf16 and act == L.ACT_NONE and x16 is None and stem_lp16_ok(d, bf16, dx_channels), \
                "a 16-bit output needs 16-bit kernels for all passes and no activation"
            x16p = lp16_pad8(x, bf16)
            y = raw_conv_lp16_flat(d, None, w, bias, act, bf16, out16=True, x16p=x16p, want_stats=bool(null_bias_grad))
            ctx.stem16 = True
            ctx.save_for_backward(x16p, w, None)        # the padded 16-bit input serves the wgrad too
            ctx.mark_non_differentiable(y)
            return lp_proxy(y.shape, y.device), y
        if y_lp:
            assert act == L.ACT_NONE, "a 16-bit output needs no activation"
        v2 = bool(bf16) and lp16_v2_ok(Cin, w.shape[3], k, stride, pad, 0)
        if v2 or chain:
            # 16-bit path: one 16-bit twin of x (or the producer's own 16-bit output) feeds the fprop and,
            # kept instead of x, the wgrad.  256 / 512-channel stride-1 stack: conv_lp16h2_kernel, the other
            # 3x3 convs (stride 2, 64 / 128 columns): conv_lp16g_kernel
            if x16 is None:
                # x_twin: a 16-bit copy of x that already exists (a channel slice of the gate's 16-bit concat)
                if x_twin is not None and v2 and USE_LP16_CAT_TWIN:
                    assert x_twin.dtype == _wd(bf16) and tuple(x_twin.shape) == tuple(x.shape)
                    x16 = x_twin
                else:
                    x16 = lp16_twin(x, bf16)
            d = conv_desc(B, H, W_, Cin, w.shape[3], k, stride, pad, reflect)
            timed = fprop_timer is not None and fprop_timer.want(d)
            if timed:
                e0, e1 = fprop_timer.bracket()
                e0.record()
            if v2:
                y = raw_conv3x3_lp16(x16, w, bias, reflect, act, bf16, 0, out16=bool(y_lp),
                                     want_stats=bool(null_bias_grad and y_lp))
            else:
                y = raw_conv_lp16g(d, 0, x16, w, bias, act, bf16, out16=bool(y_lp), want_stats=bool(null_bias_grad and y_lp))
            if timed:
                e1.record()
            ctx.lp16 = chain or lp16_wgrad_ok(Cin, w.shape[3], k, stride, pad)
            ctx.save_for_backward(x16 if ctx.lp16 else x, w, y if act != L.ACT_NONE else None)
            if y_lp:
                ctx.mark_non_differentiable(y)
                return lp_proxy(y.shape, y.device), y
            return y
        if wt and KEEP_WINOGRAD_INPUT and ctx.needs_input_grad[1]:
            # the wgrad pass contracts the same transformed input: keep it instead of x
            y, V = raw_conv_fprop_wino(x, w, bias, reflect, act, wt, keep_V=True, bf16=bf16)
            ctx.wino_V = wt
            ctx.save_for_backward(V, w, y if act != L.ACT_NONE else None)
            return y
        y = raw_conv_fprop(x, w, bias, stride, pad, reflect, act, bf16, want_stats=bool(null_bias_grad))
        ctx.save_for_backward(x, w, y if act != L.ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, g, _g16=None):
        out = Conv2dFn._backward(ctx, g, _g16)
        _dp_done(ctx.needs_input_grad, ctx.saved_tensors[1], ctx.bias_p, out[1], out[2])
        return out

    @staticmethod
    def _backward(ctx, g, _g16=None):
        addend = ctx.res_tok.take() if ctx.res_tok is not None else None    # the residual side's gradient of x, if parked
        x, w, y = ctx.saved_tensors
        stride, pad, reflect, act, has_bias, bf16 = ctx.cfg
        dx = dw = db = None
        if g is None:       # no gradient reaches the conv: what the residual side parked IS the gradient of x
            return (addend,) + (None,) * 15
        wt_ = _grad_target(w) if ctx.needs_input_grad[1] else None                       # in-place targets
        bt_ = _grad_target(ctx.bias_p) if (has_bias and ctx.needs_input_grad[2]) else None
        want_db = has_bias and ctx.needs_input_grad[2] and not ctx.skip_db
        if has_bias and ctx.needs_input_grad[2] and ctx.skip_db and (ctx.bias_p.grad is None or not ACCUM_PARAM_GRADS):
            # exact zero.  Returned as a tensor unless gradients are accumulated in place (then there is
            # nothing to add): under data parallelism the bucket hooks count one AccumulateGrad per parameter
            db = torch.zeros_like(ctx.bias_p)
        if (act != L.ACT_NONE and bf16 and ctx.needs_input_grad[0] and not ctx.needs_input_grad[1] and not want_db
                and not ctx.g_defer and not ctx.y_lp and not ctx.x_lp and not ctx.dx_channels
                and _dgrad_takes_dy16(*ctx.x_shape, w.shape[3], w.shape[0], stride, pad, reflect, bf16)):
            # frozen weights behind an activation epilogue (VGG19 conv1_1 / conv1_2 + ReLU under the perceptual loss):
            # nothing but the 16-bit dgrad reads g * act'(y) - one pass writes it in 16 bits
            g16 = raw_act_bwd_lp16(g.contiguous(), y, act, bf16)
            dx = raw_conv_dgrad(None, w, ctx.x_shape, stride, pad, reflect, bf16, 0, dy16=g16)
            return _tok_add(dx, addend), dw, db, None, None, None, None, None, None, None, None, None, None, None, None, None
        if ctx.head16:      # x is the head's 16-bit input; g the fp32 gradient of its four output columns
            g = g.contiguous()
            if act != L.ACT_NONE:
                g = raw_act_bwd(g, y, act)
            if ctx.needs_input_grad[0]:
                dx = raw_conv_dgrad(g, w, ctx.x_shape, stride, pad, reflect, bf16, 0, out16=ctx.x_lp)
                if ctx.x_lp:
                    dx = lp_grad_out(dx)
            if ctx.needs_input_grad[1]:
                dw = _finish_param_grad(raw_head_wgrad16(x, g, bf16, out=wt_), wt_)
            if want_db:
                db = _finish_param_grad(raw_colsum(g.numel() // g.shape[3], g.shape[3], g, out=bt_), bt_)
            return _tok_add(dx, addend), dw, db, None, None, None, None, None, None, None, None, None, None, None, None, None
        if ctx.stem16:      # x is the stem's padded 16-bit input [B,H,W,C8] saved by the forward pass
            g16 = lp_grad_in(g, "Conv2dFn (stem)")
            if ctx.needs_input_grad[0]:     # only the generated image inside the concat (stem_lp16_ok)
                dx = raw_conv_dgrad_thin(g16, w, ctx.x_shape, reflect)
            if ctx.needs_input_grad[1]:
                Bx, Hx, Wx, Cx = ctx.x_shape
                dd = conv_desc(Bx, Hx, Wx, Cx, w.shape[3], w.shape[0], stride, pad, reflect)
                if stem_wgrad16_ok(dd, x.shape[3], bf16) and g16.is_contiguous():
                    dw = _finish_param_grad(raw_wgrad_stem_lp16(dd, x, g16, bf16, out=wt_), wt_)
                elif lp16_flat_wgrad_ok(dd, x.shape[3], bf16):
                    dw = _finish_param_grad(raw_wgrad_lp16_flat(dd, x, x.shape[3], g16, bf16, out=wt_), wt_)
                else:       # small Cin: the first-generation kernel, reading the padded 16-bit input in place
                    dw = _finish_param_grad(raw_conv_wgrad_lp16_gen1(x, g16, Cx, w.shape[0], stride, pad, reflect, bf16,
                                                                     out=wt_), wt_)
            if want_db:
                db = _finish_param_grad(raw_colsum(g16.numel() // g16.shape[3], g16.shape[3], g16, out=bt_), bt_)
            return _tok_add(dx, addend), dw, db, None, None, None, None, None, None, None, None, None, None, None, None, None
        if ctx.y_lp:        # 16-bit edge on the output: the gradient arrives in 16 bits, no fp32 copy exists
            g16 = lp_grad_in(g, "Conv2dFn")
            if ctx.needs_input_grad[0]:
                fuse = addend is not None and not ctx.x_lp and not ctx.dx_channels
                dx = raw_conv_dgrad(None, w, ctx.x_shape, stride, pad, reflect, bf16, ctx.dx_channels, dy16=g16,
                                    out16=ctx.x_lp, addend=addend if fuse else None,
                                    nbr=ctx.nbr if (ctx.x_lp and not ctx.dx_channels) else None)
                if fuse:
                    addend = None       # added by the dgrad itself
                if ctx.x_lp:
                    dx = lp_grad_out(dx)
            if ctx.needs_input_grad[1]:
                dw = _finish_param_grad(raw_conv_wgrad(x, g16, w.shape[0], stride, pad, reflect, bf16, out=wt_), wt_)
            if want_db:
                db = _finish_param_grad(raw_colsum(g16.numel() // g16.shape[3], g16.shape[3], g16, out=bt_), bt_)
            return _tok_add(dx, addend), dw, db, None, None, None, None, None, None, None, None, None, None, None, None, None
        fused_bwd = (FUSE_WINO6_BWD and ctx.wino_V == 6 and not bf16 and ctx.needs_input_grad[0]
                     and ctx.needs_input_grad[1]
                     and _wino_tile(*ctx.x_shape, w.shape[3], 3, stride, pad, bf16, "dgrad") == 6)
        nbd = None
        if ctx.g_defer:     # the norm behind this conv parked its input gradient as a NormBwdDefer
            nbd = norm_bwd_defer_in(g, "Conv2dFn")
            Hx, Wx = ctx.x_shape[1], ctx.x_shape[2]
            if not (fused_bwd and act == L.ACT_NONE and not want_db and (not reflect or _fold_same_grid(Hx, Wx))):
                g = raw_norm_bwd_apply_rc(nbd)      # no fused transform on this path: materialise it
                nbd = None
        if nbd is None:
            g = g.contiguous()
            if act != L.ACT_NONE:
                g = raw_act_bwd(g, y, act)
        if fused_bwd:
            dx, dw = raw_conv_bwd_wino6(g, w, ctx.x_shape, reflect, x, dw_out=wt_, nbd=nbd)    # x is the saved V here
            dw = _finish_param_grad(dw, wt_)
            if want_db:
                db = _finish_param_grad(raw_colsum(g.numel() // g.shape[3], g.shape[3], g, out=bt_), bt_)
            return _tok_add(dx, addend), dw, db, None, None, None, None, None, None, None, None, None, None, None, None, None
        if ctx.lp16:        # x is the 16-bit twin saved by the forward pass; one twin of g serves both passes
            g16 = lp16_twin(g, bf16)
            if ctx.needs_input_grad[0]:
                dx = raw_conv_dgrad(g, w, ctx.x_shape, stride, pad, reflect, bf16, ctx.dx_channels, dy16=g16,
                                    out16=ctx.x_lp)
                if ctx.x_lp:
                    dx = lp_grad_out(dx)
            if ctx.needs_input_grad[1]:
                dw = _finish_param_grad(raw_conv_wgrad(x, g16, w.shape[0], stride, pad, reflect, bf16, out=wt_), wt_)
            if want_db:
                db = _finish_param_grad(raw_colsum(g.numel() // g.shape[3], g.shape[3], g, out=bt_), bt_)
            return _tok_add(dx, addend), dw, db, None, None, None, None, None, None, None, None, None, None, None, None, None
        if (not WINOGRAD_FPROP and FUSE_WINO6_BWD and not bf16 and not ctx.wino_V and not ctx.dx_channels
                and ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and x is not None and x.dtype == torch.float32
                and _wino_tile(*ctx.x_shape, w.shape[3], w.shape[0], stride, pad, bf16, "dgrad") == 6
                and _wino_tile(*ctx.x_shape, w.shape[3], w.shape[0], stride, pad, bf16, "wgrad") == 6):
            # gradient-exact hybrid: the forward ran on the direct kernel and kept x; its transform is made here
            Bx, Hx, Wx, Cx = ctx.x_shape
            V = _empty((64, Bx * (-(-Hx // 6)) * (-(-Wx // 6)), Cx), x)
            L.call("mmh_wino_input", _ptr(x), Bx, Hx, Wx, Cx, int(bool(reflect)), 6, L.F32, _ptr(V), _stream())
            dx, dw = raw_conv_bwd_wino6(g, w, ctx.x_shape, reflect, V, dw_out=wt_)
            dw = _finish_param_grad(dw, wt_)
            if want_db:
                db = _finish_param_grad(raw_colsum(g.numel() // g.shape[3], g.shape[3], g, out=bt_), bt_)
            return _tok_add(dx, addend), dw, db, None, None, None, None, None, None, None, None, None, None, None, None, None
        if ctx.needs_input_grad[0]:
            dx = raw_conv_dgrad(g, w, ctx.x_shape, stride, pad, reflect, bf16, ctx.dx_channels)
        if ctx.needs_input_grad[1]:
            if ctx.wino_V:
                dw = _finish_param_grad(raw_conv_wgrad_wino(None, g, reflect, ctx.wino_V, V=x, bf16=bf16, out=wt_), wt_)
            else:
                dw = _finish_param_grad(raw_conv_wgrad(x, g, w.shape[0], stride, pad, reflect, bf16, out=wt_), wt_)
        if want_db:
            db = _finish_param_grad(raw_colsum(g.numel() // g.shape[3], g.shape[3], g, out=bt_), bt_)
        return _tok_add(dx, addend), dw, db, None, None, None, None, None, None, None, None, None, None, None, None, None


class ConvT2dFn(torch.autograd.Function):
    """nn.ConvTranspose2d(k3,s2,p1,op1): fprop = stride-2 dgrad kernel, dgrad = stride-2 fprop."""

    @staticmethod
    def forward(ctx, x, w, bias, bf16=False, x16=None, y_lp=False, null_bias_grad=False):
        """x16 / y_lp: 16-bit edges as in Conv2dFn (convT_lp16_ok); null_bias_grad as there."""
        ctx.set_materialize_grads(False)
        _dp_use(ctx.needs_input_grad, w, bias)
        ctx.skip_db = bool(null_bias_grad and EXACT_NULL_BIAS_GRAD)
        ctx.bias_p = bias
        ctx.has_bias = bias is not None
        ctx.bf16 = bf16
        ctx.x_lp = x16 is not None
        ctx.y_lp = bool(y_lp)
        ctx.x_shape = tuple(x.shape)
        chain = bool(bf16) and convT_lp16_ok(x.shape[3], w.shape[2], bf16)
        if x16 is not None or y_lp:
            assert chain, "16-bit edges of a ConvTranspose2d need 16-bit kernels for all three passes"
        if chain:
            if x16 is None:
                x16 = lp16_twin(x, bf16)
            y = raw_convT_fprop(x16, w, bias, L.ACT_NONE, bf16, out16=bool(y_lp))
            ctx.lp16 = True
            ctx.save_for_backward(x16, w)
            if y_lp:
                ctx.mark_non_differentiable(y)
                return lp_proxy(y.shape, y.device), y
            return y
        ctx.lp16 = False
        y = raw_convT_fprop(x, w, bias, L.ACT_NONE, bf16)
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, g, _g16=None):
        out = ConvT2dFn._backward(ctx, g, _g16)
        _dp_done(ctx.needs_input_grad, ctx.saved_tensors[1], ctx.bias_p, out[1], out[2])
        return out

    @staticmethod
    def _backward(ctx, g, _g16=None):
        x, w = ctx.saved_tensors
        dx = dw = db = None
        if g is None:
            return (None,) * 7
        wt_ = _grad_target(w) if ctx.needs_input_grad[1] else None
        bt_ = _grad_target(ctx.bias_p) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        if ctx.has_bias and ctx.needs_input_grad[2] and ctx.skip_db and (ctx.bias_p.grad is None or not ACCUM_PARAM_GRADS):
            db = torch.zeros_like(ctx.bias_p)
        if ctx.lp16:
            g = lp_grad_in(g, "ConvT2dFn") if ctx.y_lp else lp16_twin(g.contiguous(), ctx.bf16)
            if ctx.needs_input_grad[0]:
                dx = raw_convT_dgrad(g, w, ctx.x_shape, ctx.bf16, out16=ctx.x_lp)
                if ctx.x_lp:
                    dx = lp_grad_out(dx)
        else:
            g = g.contiguous()
            if ctx.needs_input_grad[0]:
                dx = raw_convT_dgrad(g, w, ctx.x_shape, ctx.bf16)
        if ctx.needs_input_grad[1]:
            dw = _finish_param_grad(raw_convT_wgrad(x, g, ctx.bf16, out=wt_), wt_)
        if ctx.has_bias and ctx.needs_input_grad[2] and not ctx.skip_db:
            db = _finish_param_grad(raw_colsum(g.numel() // g.shape[3], g.shape[3], g, out=bt_), bt_)
        return dx, dw, db, None, None, None, None


# --------------------------------------------------------------------------- norm
EPS = 1e-5
_seed_counter = [0x5EED0001]


def next_dropout_seed():
    """Fresh 64-bit seed per dropout site per call (on-device counter-hash RNG)."""
    _seed_counter[0] = (_seed_counter[0] * 6364136223846793005 + 1442695040888963407) % (1 << 64)
    return _seed_counter[0]


def set_dropout_seed(seed):
    _seed_counter[0] = int(seed) % (1 << 64)


def raw_norm_stats(x, groups, out=None):
    """mean, M2 per (group, channel); groups = B (instance) or 1 (batch).  For instance norm right
    after a Winograd F(6x6,3x3) conv the per-tile partials written by its output transform are
    merged instead of reading x again.  out = (mean, m2): contiguous [groups, C] fp32 tensors to write into (slots of a
    SyncBN message buffer: no packing pass afterwards)."""
    B, H, W_, Cc = x.shape
    rows = (B // groups) * H * W_
    pend = _take_stats(x) if groups in (B, 1) else None
    if pend is not None:
        stats = pend[0]         # [B][chunks][3][C]; BatchNorm (groups == 1): the same array as one group of B * chunks
        mean = out[0] if out is not None else torch.empty((groups, Cc), dtype=torch.float32, device=x.device)
        m2 = out[1] if out is not None else torch.empty((groups, Cc), dtype=torch.float32, device=x.device)
        chunks = stats.shape[1] * (B // groups)
        if groups == 1 and B > 1 and chunks >= 128:
            # one group, B * chunks-per-image partials per channel: merge per image in parallel, then the B results
            ws = torch.empty(B * 3 * Cc, dtype=torch.float32, device=x.device)
            L.call("mmh_norm_stats_merge2", _ptr(stats), chunks, Cc, B, _ptr(ws), ws.numel() * 4, _ptr(mean), _ptr(m2),
                   _stream())
        else:
            L.call("mmh_norm_stats_merge", _ptr(stats), groups, chunks, Cc, _ptr(mean), _ptr(m2), _stream())
        return mean, m2, rows
    ws = torch.empty(L.load().mmh_norm_stats_ws_bytes(groups, rows, Cc) // 4 + 4, dtype=torch.float32, device=x.device)
    mean = out[0] if out is not None else torch.empty((groups, Cc), dtype=torch.float32, device=x.device)
    m2 = out[1] if out is not None else torch.empty((groups, Cc), dtype=torch.float32, device=x.device)
    L.call("mmh_norm_stats", _ptr(x), groups, rows, Cc, Cc, _ptr(mean), _ptr(m2), _ptr(ws),
           ws.numel() * 4, _tdt(x), _stream())
    return mean, m2, rows


def raw_norm_stats_finalize_pending(x, groups):
    """InstanceNorm right after a Winograd F(6x6,3x3) conv: merge the per-tile partials of its output transform AND
    finalise (scale = invstd, shift = -mean * invstd) in one launch -> (mean, scale, shift, invstd, rows), or None
    when no partials are pending for x"""
    B, H, W_, Cc = x.shape
    pend = _take_stats(x) if groups == B else None
    if pend is None:
        return None
    stats = pend[0]
    rows = H * W_
    mean = _empty((groups, Cc), x); m2 = _empty((groups, Cc), x)
    scale = torch.empty_like(mean); shift = torch.empty_like(mean); invstd = torch.empty_like(mean)
    L.call("mmh_norm_stats_merge_finalize", _ptr(stats), groups, stats.shape[1], Cc, float(rows), EPS, _ptr(mean), _ptr(m2),
           _ptr(scale), _ptr(shift), _ptr(invstd), _stream())
    return mean, scale, shift, invstd, rows


def raw_norm_finalize(mean, m2, count, gamma, beta, running_mean, running_var, momentum=0.1):
    groups, Cc = mean.shape
    scale = torch.empty_like(mean); shift = torch.empty_like(mean); invstd = torch.empty_like(mean)
    L.call("mmh_norm_finalize", _ptr(mean), _ptr(m2), float(count), _ptr(gamma), _ptr(beta), EPS,
           groups, Cc, _ptr(scale), _ptr(shift), _ptr(invstd), _ptr(running_mean),
           _ptr(running_var), momentum, _stream())
    return scale, shift, invstd


def raw_scale_shift_act(x, scale, shift, residual, relu, drop_p, seed, mask, keep_bits=False, out_lp=0, twin_lp=0):
    """keep_bits: also return the uint8 [B,H,W,C/4] array of surviving-lane bits (4 per byte), the
    only thing the norm backward needs of `out`.  out_lp: 0 fp32 | True bf16 | 2 fp16 output.
    twin_lp (True bf16 | 2 fp16): also return the same values as a 16-bit tensor, written by the same pass
    (mmh_scale_shift_act_twin) -> out [, keep bits] [, twin]."""
    B, H, W_, Cc = x.shape
    groups = scale.shape[0]
    rows = (B // groups) * H * W_
    out = torch.empty(x.shape, dtype=_wd(out_lp), device=x.device)
    kb = torch.empty((B, H, W_, Cc // 4), dtype=torch.uint8, device=x.device) if keep_bits else None
    if twin_lp:
        twin = torch.empty(x.shape, dtype=_wd(twin_lp), device=x.device)
        L.call("mmh_scale_shift_act_twin", _ptr(x), _ptr(scale), _ptr(shift), _ptr(residual), _ptr(out),
               groups, rows, Cc, int(relu), float(drop_p), seed, _ptr(mask), _ptr(kb), _tdt(x), _dt(out_lp), _ptr(twin),
               _dt(twin_lp), _stream())
        return (out, kb, twin) if keep_bits else (out, twin)
    L.call("mmh_scale_shift_act", _ptr(x), _ptr(scale), _ptr(shift), _ptr(residual), _ptr(out),
           groups, rows, Cc, int(relu), float(drop_p), seed, _ptr(mask), _ptr(kb), _tdt(x), _dt(out_lp), _stream())
    return (out, kb) if keep_bits else out


def _sync_stats(mean, m2, rows, group):
    """SyncBN: merge per-rank (mean, M2) with Chan's formula (replaces apex SyncBatchNorm,
    models/MMHandModel.py:109-116).  One all_gather of [3C] floats per norm site.  On the device the
    gathered (count, mean, M2) triples of the ranks are exactly the partial layout
    [groups=1][chunks=world][3][C] that mmh_norm_stats_merge reduces: one launch, no torch glue."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if DP_NO_COMM:      # timing aid: this rank's statistics stand in for the global ones
        return mean, m2, rows
    collective_counter["all_gather"] = collective_counter.get("all_gather", 0) + 1
    if mean.is_cuda:
        packed = torch.cat([torch.full_like(mean, float(rows)), mean, m2], 0)          # [3, C]
        # flat [world*3, C]: the shape both RCCL and gloo accept for all_gather_into_tensor
        gathered = torch.empty((world * 3, packed.shape[1]), dtype=packed.dtype, device=packed.device)
        dist.all_gather_into_tensor(gathered, packed, group=group)
        if USE_SYNCBN_FUSED:
            return _GatheredStats(gathered.view(world, 3 * packed.shape[1]), 0, world, mean.shape[1], rows * world), None, rows * world
        gmean = torch.empty_like(mean)
        gm2 = torch.empty_like(m2)
        L.call("mmh_norm_stats_merge", _ptr(gathered), 1, world, mean.shape[1], _ptr(gmean), _ptr(gm2), _stream())
        return gmean, gm2, rows * world
    packed = torch.cat([mean, m2], 0)          # [2, C]   (host-logic path of the CPU gloo test)
    gathered = [torch.empty_like(packed) for _ in range(world)]
    dist.all_gather(gathered, packed, group=group)
    means = torch.stack([t[0] for t in gathered])      # [world, C]
    m2s = torch.stack([t[1] for t in gathered])
    gmean = means.mean(0, keepdim=True)
    gm2 = m2s.sum(0, keepdim=True) + rows * ((means - gmean) ** 2).sum(0, keepdim=True)
    return gmean.contiguous(), gm2.contiguous(), rows * world


# collectives issued by the norm layers (SyncBN), for tests / diagnostics: {"all_gather": n, "all_reduce": n}
collective_counter = {}


class _GatheredStats:
    """One site's (count, mean, M2) triples as the all-gather delivered them: rank r's at buf[r, off : off + 3 C].  The merge
    over the ranks happens together with the finalisation (mmh_syncbn_merge_finalize: one launch per site instead of a block
    copy, a merge and a finalise - 101 sites per iteration).  MMH_SYNCBN_FUSED=0: merge here, finalise later (the round-4
    path; same values)."""
    __slots__ = ("buf", "off", "world", "C", "count")

    def __init__(self, buf, off, world, Cc, count):
        self.buf, self.off, self.world, self.C, self.count = buf, off, world, Cc, count

    def finalize(self, gamma, beta, running_mean, running_var, momentum=0.1):
        dev = self.buf.device
        mean = torch.empty((1, self.C), dtype=torch.float32, device=dev)
        m2 = torch.empty_like(mean); scale = torch.empty_like(mean); shift = torch.empty_like(mean); invstd = torch.empty_like(mean)
        L.call("mmh_syncbn_merge_finalize", C.c_void_p(self.buf.data_ptr() + 4 * self.off), self.world, self.buf.stride(0), self.C,
               float(self.count), EPS, _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(m2), _ptr(scale), _ptr(shift), _ptr(invstd),
               _ptr(running_mean), _ptr(running_var), momentum, _stream())
        return mean, m2, scale, shift, invstd


USE_SYNCBN_FUSED = os.environ.get("MMH_SYNCBN_FUSED", "1") != "0"
# the statistics / backward-sum kernels of a packed SyncBN node write straight into the collective's message buffer
# (stats_message / sums_message); MMH_SYNCBN_MSG=0: separate tensors, packed by torch.cat afterwards (the round-4 path)
USE_SYNCBN_MSG = os.environ.get("MMH_SYNCBN_MSG", "1") != "0"


def stats_message(sizes, rows, device):
    """The all-gather message of a pack of SyncBN sites, allocated BEFORE the sites compute their statistics: per site
    [count | mean | M2], C floats each.  The count slots are filled here (one launch when the sites share a row count, as
    the packs of the step do), the statistics kernels write mean / M2 straight into their slots (raw_norm_stats out=): no
    full_like / cat / cat per site afterwards (~6 tiny torch kernels and 60 us of host time per site).
    -> (buffer, [(mean view, m2 view) per site])"""
    total = 3 * sum(sizes)
    if len(set(rows)) == 1:
        buf = torch.full((total,), float(rows[0]), dtype=torch.float32, device=device)
    else:
        buf = torch.empty((total,), dtype=torch.float32, device=device)
    views, off = [], 0
    for Cc, r in zip(sizes, rows):
        if len(set(rows)) != 1:
            buf[off:off + Cc].fill_(float(r))
        views.append((buf[off + Cc:off + 2 * Cc].view(1, Cc), buf[off + 2 * Cc:off + 3 * Cc].view(1, Cc)))
        off += 3 * Cc
    return buf, views


def _sync_stats_multi(items, group, packed=None):
    """SyncBN statistics of several norm sites with ONE collective: items = [(mean [1,C], m2 [1,C], rows), ...] ->
    [(gmean, gm2, count), ...].  The sites' (count, mean, M2) triples travel as one flat message (all_gather of
    sum 3 C_i floats per rank); each site then merges its [world][3][C] block with Chan's formula on the device.
    packed: the message, already holding every site's triple (stats_message)."""
    import torch.distributed as dist
    if len(items) == 1 and packed is None:
        return [_sync_stats(*items[0], group)]
    if DP_NO_COMM:
        return [(m, m2, rows) for m, m2, rows in items]
    world = dist.get_world_size(group)
    collective_counter["all_gather"] = collective_counter.get("all_gather", 0) + 1
    if len(items) > 1:
        collective_counter["packed_sites"] = collective_counter.get("packed_sites", 0) + len(items)
    if packed is None:
        packed = torch.cat([torch.cat([torch.full_like(m, float(rows)), m, m2], 0).reshape(-1) for m, m2, rows in items])
    gathered = torch.empty((world, packed.numel()), dtype=packed.dtype, device=packed.device)
    if packed.is_cuda:
        dist.all_gather_into_tensor(gathered.view(-1), packed, group=group)
    else:
        parts = [torch.empty_like(packed) for _ in range(world)]
        dist.all_gather(parts, packed, group=group)
        gathered = torch.stack(parts)
    out, off = [], 0
    for m, m2, rows in items:
        Cc = m.shape[1]
        if m.is_cuda and USE_SYNCBN_FUSED:
            out.append((_GatheredStats(gathered, off, world, Cc, rows * world), None, rows * world))
            off += 3 * Cc
            continue
        blk = gathered[:, off:off + 3 * Cc].reshape(world * 3, Cc).contiguous()        # [world][3][C]
        off += 3 * Cc
        if m.is_cuda:
            gmean, gm2 = torch.empty_like(m), torch.empty_like(m2)
            L.call("mmh_norm_stats_merge", _ptr(blk), 1, world, Cc, _ptr(gmean), _ptr(gm2), _stream())
        else:
            t = blk.view(world, 3, Cc)
            gmean = t[:, 1].mean(0, keepdim=True)
            gm2 = t[:, 2].sum(0, keepdim=True) + rows * ((t[:, 1] - gmean) ** 2).sum(0, keepdim=True)
        out.append((gmean.contiguous(), gm2.contiguous(), rows * world))
    return out


def sums_message(sizes, device):
    """The all-reduce message of a pack of SyncBN sites' backward sums, allocated before the reduce kernels run: per site
    [s1 | s2], C floats each; the kernels write into their slots (_norm_bwd_local sums_out=).  -> (buffer, [(s1, s2) views])"""
    buf = torch.empty((2 * sum(sizes),), dtype=torch.float32, device=device)
    views, off = [], 0
    for Cc in sizes:
        views.append((buf[off:off + Cc].view(1, Cc), buf[off + Cc:off + 2 * Cc].view(1, Cc)))
        off += 2 * Cc
    return buf, views


def _sync_bwd_sums_multi(pairs, group):
    """the backward sums (s1, s2) of several norm sites all-reduced as ONE message; pairs = [(s1, s2), ...]"""
    import torch.distributed as dist
    if DP_NO_COMM:
        return list(pairs)
    collective_counter["all_reduce"] = collective_counter.get("all_reduce", 0) + 1
    if len(pairs) > 1:
        collective_counter["packed_sites"] = collective_counter.get("packed_sites", 0) + len(pairs)
    packed = torch.cat([torch.cat([s1, s2], 0).reshape(-1) for s1, s2 in pairs])
    dist.all_reduce(packed, group=group)
    out, off = [], 0
    for s1, s2 in pairs:
        n = s1.numel()
        out.append((packed[off:off + n].view_as(s1).contiguous(), packed[off + n:off + 2 * n].view_as(s2).contiguous()))
        off += 2 * n
    return out


def _sync_bwd_sums_inplace(packed, pairs, group):
    """As _sync_bwd_sums_multi when the pairs already ARE the slots of the message `packed` (sums_message): the local sums are
    kept in ONE clone (the affine parameters' gradients want them), the all-reduce runs in place and the global sums are the
    same views -> (local pairs, global pairs)."""
    import torch.distributed as dist
    if DP_NO_COMM:
        return list(pairs), list(pairs)
    collective_counter["all_reduce"] = collective_counter.get("all_reduce", 0) + 1
    if len(pairs) > 1:
        collective_counter["packed_sites"] = collective_counter.get("packed_sites", 0) + len(pairs)
    local = packed.clone()
    dist.all_reduce(packed, group=group)
    loc, off = [], 0
    for s1, s2 in pairs:
        n = s1.numel()
        loc.append((local[off:off + n].view_as(s1), local[off + n:off + 2 * n].view_as(s2)))
        off += 2 * n
    return loc, list(pairs)


class _SiteCtx:
    """What the norm forward / backward phases below need of an autograd ctx, for ONE site of a multi-site node"""

    def __init__(self):
        self.saved_tensors, self.nondiff = (), []

    def save_for_backward(self, *t):
        self.saved_tensors = t

    def mark_non_differentiable(self, *t):
        self.nondiff += list(t)


# The norm node in four phases, so that several sites can share their collectives (NormActMultiFn):
#   forward:  _norm_fwd_local (statistics of this rank) -> [SyncBN: all-gather + merge] -> _norm_fwd_finish
#   backward: _norm_bwd_local (sums of this rank)       -> [SyncBN: all-reduce]         -> _norm_bwd_finish
def _norm_fwd_local(ctx, x, gamma, beta, mode, relu, drop_p, out_lp, x16, stat_out=None):
    ctx.in_lp = x16 is not None
    ctx.out_lp = bool(out_lp)
    if x16 is not None:
        assert tuple(x16.shape) == tuple(x.shape) and x16.is_contiguous()
        x = x16
    else:
        _chk(x, "x")
    if drop_p > 0 and not relu:
        # the keep bits are (out > 0): exact only behind a ReLU (the reference never drops without one)
        raise RuntimeError("NormActFn: dropout without a preceding ReLU is not supported")
    B = x.shape[0]
    groups = B if mode == "instance" else 1
    fast = raw_norm_stats_finalize_pending(x, groups) if (mode == "instance" and gamma is None and beta is None) else None
    if fast is not None:
        return x, groups, fast, None, None, fast[4]
    mean, m2, rows = raw_norm_stats(x, groups, out=stat_out)
    return x, groups, None, mean, m2, rows


# A norm output that stays fp32 for a non-conv consumer (a residual add) AND feeds a 16-bit 3x3 conv: the apply pass writes the
# conv's 16-bit operand beside the fp32 tensor (mmh_scale_shift_act_twin) instead of leaving a conversion pass to the conv.
USE_NORM_TWIN = os.environ.get("MMH_NORM_TWIN", "1") != "0"


def norm_twin_ok(C):
    c8 = C // 8
    return USE_NORM_TWIN and USE_LP16_CAT_TWIN and C % 8 == 0 and 1 <= c8 <= 256 and (c8 & (c8 - 1)) == 0


def _norm_fwd_finish(ctx, st, synced, gamma, beta, residual, running_mean, running_var, relu, drop_p, seed, mask,
                     sync_group, out_lp, x16, defer, want_twin=0):
    x, groups, fast, mean, m2, rows = st
    if fast is not None:
        mean, scale, shift, invstd, rows = fast
        count = rows
    else:
        count = rows
        if synced is not None:
            mean, m2, count = synced
        if isinstance(mean, _GatheredStats):        # SyncBN: merge over the ranks and finalise in one launch
            mean, m2, scale, shift, invstd = mean.finalize(gamma, beta, running_mean, running_var)
        else:
            scale, shift, invstd = raw_norm_finalize(mean, m2, count, gamma, beta, running_mean, running_var)
    ctx.defer = int(defer)
    if defer == 3:
        # defer 3 (a block's last norm: no ReLU, no dropout; its output feeds a gate / residual add and is
        # written as usual): only the BACKWARD apply pass goes to the producing conv's backward transform
        assert (x16 is None and not out_lp and not relu and drop_p == 0 and x.dtype == torch.float32
                and norm_fusion_ok(x.shape[3]))
        out = raw_scale_shift_act(x, scale, shift, residual, False, 0.0, 0, None)
        ctx.cfg = (groups, rows, count, False, 0.0, sync_group, residual is not None)
        ctx.save_for_backward(x, None, mean, invstd, gamma, scale, shift, None)
        return (out,)
    if defer:
        # defer 1: the apply pass runs inside the consuming conv's input transform (the caller builds the
        # NormDefer from the returned scale / shift / dropout bits); no output, no keep bits: the backward
        # decides again.  defer 2: the backward's apply pass goes to the producing conv too.
        assert x16 is None and not out_lp and residual is None and x.dtype == torch.float32 and norm_fusion_ok(x.shape[3])
        dbits, drows = raw_dropout_bits(x.shape, drop_p, seed, mask, x.device, rows=True) if drop_p > 0 else (None, None)
        ctx.cfg = (groups, rows, count, bool(relu), float(drop_p), sync_group, False)
        ctx.save_for_backward(x, dbits, mean, invstd, gamma, scale, shift, drows)
        ctx.mark_non_differentiable(scale, shift)
        if drows is not None:
            ctx.mark_non_differentiable(drows)
        return lp_proxy(x.shape, x.device), scale, shift, drows
    masked = bool(relu or drop_p > 0)
    twin_lp = want_twin if (want_twin and not out_lp and norm_twin_ok(x.shape[3])) else 0
    r = raw_scale_shift_act(x, scale, shift, residual, relu, drop_p, seed, mask, keep_bits=masked, out_lp=out_lp,
                            twin_lp=twin_lp)        # masked: the backward needs only which lanes survived (4 bits per float4)
    r = r if isinstance(r, tuple) else (r,)
    out, kb, twin = r[0], (r[1] if masked else None), (r[-1] if twin_lp else None)
    ctx.cfg = (groups, rows, count, bool(relu), float(drop_p), sync_group,
               residual is not None)
    ctx.save_for_backward(x, kb, mean, invstd, gamma)
    ctx.nbr_site = None
    if (USE_NBR and out_lp and masked and residual is None and x.dtype != torch.float32 and x.shape[3] >= NBR_MIN_C
            and x.shape[3] % 256 == 0):
        # the conv that consumes `out` may take this norm's backward sums in its dgrad epilogue (NormBwdSite)
        ctx.nbr_site = NormBwdSite(x, kb, mean, invstd, groups, float(drop_p))
        nbr_site_put(out, ctx.nbr_site)
    if want_twin and not out_lp:        # (out, twin | None): the caller asked for the pair
        if twin is not None:
            ctx.mark_non_differentiable(twin)
        return out, twin
    if out_lp:
        if residual is not None:
            raise RuntimeError("NormActFn: a 16-bit output together with a residual is not supported")
        ctx.mark_non_differentiable(out)
        return lp_proxy(x.shape, x.device), out
    return (out,)


# InstanceNorm backward in ONE pass where a (sample, channel) plane fits a workgroup (mmh_norm_bwd_fused: 16-bit x, rows <=
# 8192 / 8-channel lane groups - the 64x64 feature maps of the 256x256 configurations): g, x and the keep bits are read once
# instead of twice, the reduce launch is gone.  Not under SyncBN (the sums cross ranks between the passes).
# MMH_NORM_BWD_PLANE=0: always reduce + apply.
USE_NORM_BWD_PLANE = os.environ.get("MMH_NORM_BWD_PLANE", "0") == "1"


def _norm_bwd_local(ctx, g, sums_out=None):
    """this rank's sums (s1 = sum dz, s2 = sum dz * xhat per plane) -> (g, s1, s2); sums_out = (s1, s2): slots of a SyncBN
    message buffer to write into (groups == 1)"""
    groups, rows, count, relu, drop_p, sync_group, has_res = ctx.cfg
    ctx.fused_dx = None
    if ctx.defer:
        x, dbits, mean, invstd, gamma, scale, shift, drows = ctx.saved_tensors
        g = g.contiguous()
        Cc = x.shape[3]
        ws = _ws(L.load().mmh_norm_bwd_ws_bytes(groups, rows, Cc), x)
        s1, s2 = sums_out if sums_out is not None else (_empty((groups, Cc), x), _empty((groups, Cc), x))
        L.call("mmh_norm_bwd_reduce_rc", _ptr(g), _ptr(x), _ptr(mean), _ptr(invstd), _ptr(scale), _ptr(shift),
               _ptr(dbits), groups, rows, Cc, int(relu), drop_p, _ptr(s1), _ptr(s2), _ptr(ws), ws.numel() * 4, _stream())
        return g, s1, s2
    x, out, mean, invstd, gamma = ctx.saved_tensors     # `out` here = the keep-bits array (or None)
    if has_res and relu:
        raise RuntimeError("NormActFn: residual together with ReLU is not on the reference path")
    # a 16-bit edge on the output: its gradient comes from a 16-bit convolution's dgrad, in 16 bits
    g = lp_grad_in(g, "NormActFn") if ctx.out_lp else g.contiguous()
    Cc = x.shape[3]
    masked = 2 if (relu or drop_p > 0) else 0
    site = getattr(ctx, "nbr_site", None)
    if site is not None and site.g is g and site.s1 is not None:
        # the dgrad that produced g took the sums in its epilogue (mmh_conv3x3_lp16_dgrad_nbr): no reduce pass
        s1, s2 = site.s1, site.s2
        site.g = site.s1 = site.s2 = None
        if sums_out is not None:
            sums_out[0].copy_(s1.view_as(sums_out[0]))
            sums_out[1].copy_(s2.view_as(sums_out[1]))
            s1, s2 = sums_out
        return g, s1, s2
    s1, s2 = sums_out if sums_out is not None else (_empty((groups, Cc), x), _empty((groups, Cc), x))
    if (USE_NORM_BWD_PLANE and sync_group is None and x.dtype != torch.float32 and g.dtype in (torch.float32, x.dtype)
            and L.load().mmh_norm_bwd_fused_supported(groups, rows, Cc, masked, _tdt(g), _tdt(x))):
        dx = torch.empty_like(x)
        L.call("mmh_norm_bwd_fused", _ptr(g), _ptr(out), _ptr(x), _ptr(mean), _ptr(invstd), _ptr(gamma), float(count),
               groups, rows, Cc, masked, drop_p, _ptr(s1), _ptr(s2), _ptr(dx), _tdt(g), _tdt(x), _tdt(dx), _stream())
        ctx.fused_dx = dx
        return g, s1, s2
    ws = _ws(L.load().mmh_norm_bwd_ws_bytes(groups, rows, Cc), x)
    L.call("mmh_norm_bwd_reduce", _ptr(g), _ptr(out), _ptr(x), _ptr(mean), _ptr(invstd), groups,
           rows, Cc, masked, drop_p, _ptr(s1), _ptr(s2), _ptr(ws), ws.numel() * 4, _tdt(g), _tdt(x), _stream())
    return g, s1, s2


def _norm_bwd_finish(ctx, g, s1l, s2l, s1, s2):
    """s1l / s2l: this rank's sums (the affine parameters' gradients), s1 / s2: the global ones -> (dx, dgamma, dbeta, dres)"""
    groups, rows, count, relu, drop_p, sync_group, has_res = ctx.cfg
    gamma = ctx.saved_tensors[4]
    dgamma = dbeta = None
    if gamma is not None:
        # autograd may keep what it is handed as the parameter's .grad and add later contributions into it: a copy, unless
        # nobody reads the local sums after this call - under SyncBN the apply pass takes the all-reduced ones (separate
        # tensors), and an apply pass that is not deferred is enqueued below, ahead of any later writer (202 copies per
        # --norm batch step otherwise)
        own = (s1l is not s1 and s2l is not s2) or not ctx.defer
        dgamma = s2l.sum(0) if groups > 1 else (s2l.reshape(-1) if own else s2l.reshape(-1).clone())
        dbeta = s1l.sum(0) if groups > 1 else (s1l.reshape(-1) if own else s1l.reshape(-1).clone())
    if ctx.defer:
        x, dbits, mean, invstd, gamma, scale, shift, drows = ctx.saved_tensors
        e = NormBwdDefer()
        e.g, e.x, e.mean, e.invstd, e.gamma, e.s1, e.s2, e.count = g, x, mean, invstd, gamma, s1, s2, count
        e.scale, e.shift, e.dbits, e.drows = scale, shift, dbits, drows
        e.groups, e.rows, e.relu, e.drop_p = groups, rows, relu, drop_p
        dx = norm_bwd_defer_out(e) if ctx.defer >= 2 else raw_norm_bwd_apply_rc(e)
        return dx, dgamma, dbeta, (g if has_res else None)
    x, out, mean, invstd, gamma = ctx.saved_tensors
    Cc = x.shape[3]
    masked = 2 if (relu or drop_p > 0) else 0
    dx = getattr(ctx, "fused_dx", None)
    ctx.fused_dx = None
    if dx is None:
        dx = torch.empty_like(x)        # a 16-bit x came from a 16-bit convolution: its gradient goes back in 16 bits
        L.call("mmh_norm_bwd_apply", _ptr(g), _ptr(out), _ptr(x), _ptr(mean), _ptr(invstd),
               _ptr(gamma), _ptr(s1), _ptr(s2), float(count), groups, rows, Cc, masked, drop_p,
               _ptr(dx), _tdt(g), _tdt(x), _tdt(dx), _stream())
    dres = g if has_res else None
    if ctx.in_lp:
        dx = lp_grad_out(dx)
    return dx, dgamma, dbeta, dres


class NormActFn(torch.autograd.Function):
    """[Batch|Instance]Norm2d (training statistics) -> ReLU -> Dropout (+ residual add).

    mode: 'batch' | 'instance'.  ``mask`` (uint8, test hook) replaces the on-device RNG.
    """

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, running_mean, running_var, mode, relu, drop_p,
                seed, mask, sync_group, out_lp=0, x16=None, defer=0, res_tok=None, want_twin=0):
        """out_lp (True bf16 | 2 fp16): the output is written in that 16-bit type only - it feeds a
        16-bit convolution (conv_lp16.hip) and nothing else, so no fp32 copy and no conversion pass.
        x16: the producing convolution wrote x in 16 bits only (Conv2dFn y_lp); x is then the proxy on
        the autograd edge.  Statistics and all arithmetic are fp32 either way (apex O1 keeps
        batch_norm in fp32 on fp16 conv outputs).
        want_twin (True bf16 | 2 fp16): the fp32 output also feeds a 16-bit conv - returns (out, twin16 | None).
        Returns out | (proxy, out16) with out_lp | (proxy, scale, shift, drows) with defer 1 / 2."""
        ctx.set_materialize_grads(False)
        ctx.res_tok = res_tok if residual is not None else None    # `residual` is also a conv's input (ResidualToken)
        st = _norm_fwd_local(ctx, x, gamma, beta, mode, relu, drop_p, out_lp, x16)
        synced = None
        if mode == "batch" and sync_group is not None and st[2] is None:
            synced = _sync_stats(st[3], st[4], st[5], sync_group)
        outs = _norm_fwd_finish(ctx, st, synced, gamma, beta, residual, running_mean, running_var, relu, drop_p, seed,
                                mask, sync_group, out_lp, x16, defer, want_twin if not defer else 0)
        if want_twin and not defer and not out_lp and outs[1] is None:
            return outs[0], None        # (a None output is not a tensor: autograd passes it through)
        return outs[0] if len(outs) == 1 else outs

    @staticmethod
    def backward(ctx, g, _g16=None, _a=None, _b=None):
        if g is None:
            return (None,) * 17
        sync_group = ctx.cfg[5]
        g, s1l, s2l = _norm_bwd_local(ctx, g)
        s1, s2 = s1l, s2l
        if sync_group is not None:
            (s1, s2), = _sync_bwd_sums_multi([(s1l, s2l)], sync_group)
        dx, dgamma, dbeta, dres = _norm_bwd_finish(ctx, g, s1l, s2l, s1, s2)
        if ctx.res_tok is not None and ctx.res_tok.park(dres):
            dres = None         # joins the block input's gradient inside its first conv's dgrad
        return dx, dgamma, dbeta, dres, None, None, None, None, None, None, None, None, None, None, None, None, None


NORM_SITE_ARGS = 14     # per site: x, gamma, beta, residual, running_mean, running_var, relu, drop_p, seed, mask, out_lp, x16, defer, _
NORM_SITE_OUTS = 4


class NormActMultiFn(torch.autograd.Function):
    """Several BatchNorm sites that do not depend on each other - the three generator streams at one depth, the real and
    the fake pass of a discriminator - as ONE autograd node, so that under SyncBN (apex convert_syncbn_model,
    models/MMHandModel.py:109-116) their statistics travel in one all-gather and their backward sums in one all-reduce
    instead of one collective per site (101 + 101 latency-bound collectives per iteration).
    Arithmetic per site is NormActFn's, phase by phase.

    apply(sync_group, n, *flat): flat = n x NORM_SITE_ARGS entries
        (x, gamma, beta, residual, running_mean, running_var, relu, drop_p, seed, mask, out_lp, x16, defer, None)
    returns n x NORM_SITE_OUTS entries: per site NormActFn's outputs padded with None."""

    @staticmethod
    def forward(ctx, sync_group, n, *flat):
        ctx.set_materialize_grads(False)
        assert len(flat) == n * NORM_SITE_ARGS
        sites, args, states = [], [], []
        # SyncBN on the device: the all-gather's message exists first and the statistics kernels write into it (stats_message)
        msg = slots = None
        if sync_group is not None and USE_SYNCBN_MSG and not DP_NO_COMM and all(flat[i * NORM_SITE_ARGS].is_cuda for i in range(n)):
            xs = [flat[i * NORM_SITE_ARGS] for i in range(n)]
            msg, slots = stats_message([x.shape[3] for x in xs], [x.shape[0] * x.shape[1] * x.shape[2] for x in xs], xs[0].device)
        for i in range(n):
            a = flat[i * NORM_SITE_ARGS:(i + 1) * NORM_SITE_ARGS]
            x, gamma, beta, residual, rmean, rvar, relu, drop_p, seed, mask, out_lp, x16, defer, _ = a
            sc = _SiteCtx()
            states.append(_norm_fwd_local(sc, x, gamma, beta, "batch", relu, drop_p, out_lp, x16,
                                          stat_out=slots[i] if slots is not None else None))
            sites.append(sc)
            args.append(a)
        synced = [None] * n
        if sync_group is not None:
            synced = _sync_stats_multi([(st[3], st[4], st[5]) for st in states], sync_group, packed=msg)
        outs, saved, counts, nondiff = [], [], [], []
        for sc, st, sy, a in zip(sites, states, synced, args):
            x, gamma, beta, residual, rmean, rvar, relu, drop_p, seed, mask, out_lp, x16, defer, _ = a
            o = _norm_fwd_finish(sc, st, sy, gamma, beta, residual, rmean, rvar, relu, drop_p, seed, mask, sync_group,
                                 out_lp, x16, defer)
            outs += list(o) + [None] * (NORM_SITE_OUTS - len(o))
            saved += list(sc.saved_tensors)
            counts.append(len(sc.saved_tensors))
            nondiff += sc.nondiff
            sc.saved_tensors = ()
        if nondiff:     # ONE call: mark_non_differentiable replaces, it does not append
            ctx.mark_non_differentiable(*nondiff)
        ctx.save_for_backward(*saved)
        ctx.sites, ctx.counts, ctx.sync_group = sites, counts, sync_group
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        n = len(ctx.sites)
        saved, off = ctx.saved_tensors, 0
        for sc, c in zip(ctx.sites, ctx.counts):
            sc.saved_tensors = saved[off:off + c]
            off += c
        live = [i for i in range(n) if grads[i * NORM_SITE_OUTS] is not None]
        # SyncBN on the device: the reduce kernels write their sums into the all-reduce's message (sums_message)
        msg = slots = None
        if (ctx.sync_group is not None and live and USE_SYNCBN_MSG and not DP_NO_COMM
                and all(grads[i * NORM_SITE_OUTS].is_cuda and ctx.sites[i].cfg[0] == 1 for i in live)):
            msg, sl = sums_message([ctx.sites[i].saved_tensors[0].shape[3] for i in live], grads[live[0] * NORM_SITE_OUTS].device)
            slots = dict(zip(live, sl))
        loc = {i: _norm_bwd_local(ctx.sites[i], grads[i * NORM_SITE_OUTS], sums_out=slots[i] if slots is not None else None)
               for i in live}
        glob = {i: (loc[i][1], loc[i][2]) for i in live}
        if ctx.sync_group is not None and live:
            if msg is not None:
                lo, red = _sync_bwd_sums_inplace(msg, [glob[i] for i in live], ctx.sync_group)
                loc = {i: (loc[i][0], l1, l2) for i, (l1, l2) in zip(live, lo)}
            else:
                red = _sync_bwd_sums_multi([glob[i] for i in live], ctx.sync_group)
            glob = dict(zip(live, red))
        out = [None, None]
        for i in range(n):
            if i in loc:
                g, s1l, s2l = loc[i]
                dx, dgamma, dbeta, dres = _norm_bwd_finish(ctx.sites[i], g, s1l, s2l, *glob[i])
            else:
                dx = dgamma = dbeta = dres = None
            out += [dx, dgamma, dbeta, dres] + [None] * (NORM_SITE_ARGS - 4)
            ctx.sites[i].saved_tensors = ()
        return tuple(out)


# VGG19 conv1_1 + ReLU + conv1_2 + ReLU (losses/L1_plus_perceptualLoss.py:22-27 with the shipped --perceptual_layers 3) as ONE
# node in 16-bit mode, with the edge between the two convs 16-bit: conv1_1 writes its ReLU output in 16 bits (as separate nodes
# it wrote fp32 and a conversion pass made conv1_2's operand), conv1_2's dgrad hands its gradient back in 16 bits and conv1_1's
# ReLU backward reads both in 16 bits.  Same values as the two nodes (rounding and masking commute); the weights are frozen:
# only the image gradient goes back.  MMH_VGG_PAIR=0: two Conv2dFn nodes.
USE_VGG_PAIR = os.environ.get("MMH_VGG_PAIR", "1") != "0"


def vgg_pair_ok(x, w1, w2, bf16):
    if not (USE_VGG_PAIR and bf16 and x.dim() == 4 and x.shape[3] == 4 and tuple(w1.shape) == (3, 3, 4, 64)
            and tuple(w2.shape) == (3, 3, 64, 64)):
        return False
    B, H, W_, _ = x.shape
    d1 = conv_desc(B, H, W_, 4, 64, 3, 1, 1, False)
    d2 = conv_desc(B, H, W_, 64, 64, 3, 1, 1, False)
    return bool(lp16_flat_ok(d1, bf16) and conv7_n4_ok(d1, 1, bf16) and lp16g_ok(d2, 0, bf16) and lp16g_ok(d2, 1, bf16))


class VggPairFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, bf16):
        B, H, W_, _ = x.shape
        y1 = raw_conv_lp16_flat(conv_desc(B, H, W_, 4, 64, 3, 1, 1, False), x, w1, b1, L.ACT_RELU, bf16, out16=True)
        y2 = raw_conv_lp16g(conv_desc(B, H, W_, 64, 64, 3, 1, 1, False), 0, y1, w2, b2, L.ACT_RELU, bf16, out16=False)
        ctx.bf16 = bf16
        ctx.x_shape = tuple(x.shape)
        if ctx.needs_input_grad[0]:
            ctx.save_for_backward(y1, y2, w1, w2)
        return y2

    @staticmethod
    def backward(ctx, g):
        y1, y2, w1, w2 = ctx.saved_tensors
        bf16 = ctx.bf16
        B, H, W_, _ = ctx.x_shape
        g2 = raw_act_bwd_lp16(g.contiguous(), y2, L.ACT_RELU, bf16)                 # 16-bit (g * [y2 > 0])
        dy1 = raw_conv_lp16g(conv_desc(B, H, W_, 64, 64, 3, 1, 1, False), 1, g2, w2, None, L.ACT_NONE, bf16, out16=True)
        g1 = torch.empty_like(dy1)
        L.call("mmh_act_bwd_lp16_io", _ptr(dy1), 1, _ptr(y1), 1, dy1.numel(), L.ACT_RELU, _dt(bf16), _ptr(g1), _stream())
        dx = torch.empty((B, H, W_, 4), dtype=torch.float32, device=g.device)
        raw_conv7_n4(conv_desc(B, H, W_, 4, 64, 3, 1, 1, False), 1, g1, w1, None, dx, L.ACT_NONE, bf16)
        return dx, None, None, None, None, None


USE_VGG_L1_LP16 = os.environ.get("MMH_VGG_L1_LP16", "1") != "0"


class VggL1Fn(torch.autograd.Function):
    """lambda * mean|vgg(x_fake) - vgg(x_real)| for the shipped slice (conv1_1, ReLU, conv1_2, ReLU) in 16-bit mode, as ONE
    node whose feature maps are 16-bit - what apex O1 does to losses/L1_plus_perceptualLoss.py:60-66 (the convolutions
    return fp16, F.l1_loss takes them as they are).  Against VggPairFn + L1MeanFn: the two 64-channel feature maps are
    written and read in 2 bytes, and the backward starts from one pass (mmh_l1_relu_bwd_lp16: L1 gradient x ReLU mask ->
    16 bits) instead of the fp32 L1 gradient followed by the mask pass.  The loss differs from the fp32-feature form by
    the rounding of the features (relative 2^-9 per element, unbiased)."""

    @staticmethod
    def forward(ctx, x_fake, x_real, w1, b1, w2, b2, weight, bf16):
        B, H, W_, _ = x_fake.shape
        d1, d2 = conv_desc(B, H, W_, 4, 64, 3, 1, 1, False), conv_desc(B, H, W_, 64, 64, 3, 1, 1, False)
        y1 = raw_conv_lp16_flat(d1, x_fake, w1, b1, L.ACT_RELU, bf16, out16=True)
        f = raw_conv_lp16g(d2, 0, y1, w2, b2, L.ACT_RELU, bf16, out16=True)
        r = raw_conv_lp16g(d2, 0, raw_conv_lp16_flat(d1, x_real, w1, b1, L.ACT_RELU, bf16, out16=True), w2, b2,
                           L.ACT_RELU, bf16, out16=True)
        n = f.numel()
        ws = _ws(L.load().mmh_reduce_ws_bytes(n), x_fake)
        out = _empty((), x_fake)
        L.call("mmh_l1_fwd_lp16", _ptr(f), _ptr(r), n, float(weight), float(n), _dt(bf16), _ptr(out), _ptr(ws),
               ws.numel() * 4, _stream())
        ctx.cfg = (float(weight), bf16, tuple(x_fake.shape))
        if ctx.needs_input_grad[0]:
            ctx.save_for_backward(y1, f, r, w1, w2)
        return out

    @staticmethod
    def backward(ctx, g):
        y1, f, r, w1, w2 = ctx.saved_tensors
        weight, bf16, (B, H, W_, _) = ctx.cfg
        g = g.contiguous().float()
        g2 = torch.empty_like(f)
        L.call("mmh_l1_relu_bwd_lp16", _ptr(f), _ptr(r), f.numel(), weight, float(f.numel()), _ptr(g), _dt(bf16), _ptr(g2),
               _stream())
        dy1 = raw_conv_lp16g(conv_desc(B, H, W_, 64, 64, 3, 1, 1, False), 1, g2, w2, None, L.ACT_NONE, bf16, out16=True)
        g1 = g2                                               # g2 is dead once the dgrad has read it
        L.call("mmh_act_bwd_lp16_io", _ptr(dy1), 1, _ptr(y1), 1, dy1.numel(), L.ACT_RELU, _dt(bf16), _ptr(g1), _stream())
        dx = torch.empty((B, H, W_, 4), dtype=torch.float32, device=g.device)
        raw_conv7_n4(conv_desc(B, H, W_, 4, 64, 3, 1, 1, False), 1, g1, w1, None, dx, L.ACT_NONE, bf16)
        return dx, None, None, None, None, None, None, None


class AffineActFn(torch.autograd.Function):
    """out = relu?(x*scale[c] + shift[c]) with fixed per-channel scale/shift: eval-mode
    BatchNorm (aug.py:38-39) and the ImageNet pre-normalisation of the perceptual loss
    (losses/L1_plus_perceptualLoss.py:40-58)."""

    @staticmethod
    def forward(ctx, x, scale, shift, relu):
        out = raw_scale_shift_act(x, scale.reshape(1, -1), shift.reshape(1, -1), None, relu, 0.0,
                                  0, None)
        ctx.relu = relu
        ctx.save_for_backward(scale, out if relu else None)
        return out

    @staticmethod
    def backward(ctx, g):
        scale, out = ctx.saved_tensors
        g = g.contiguous()
        if ctx.relu:
            g = raw_act_bwd(g, out, L.ACT_RELU)
        zero = torch.zeros_like(scale).reshape(1, -1)
        dx = raw_scale_shift_act(g, scale.reshape(1, -1), zero, None, False, 0.0, 0, None)
        return dx, None, None, None


# --------------------------------------------------------------------------- gate
class GateFn(torch.autograd.Function):
    """PATBlock tail (models/Generator.py:115-130): out = x1 + s1*sig(s2)*sig(s3),
    x2n = cat(s3, out), x3n = cat(s2, out) — the concat is written by the same kernel."""

    @staticmethod
    def forward(ctx, x1, s1, s2, s3, want_cat, cat_lp=0, s2_16=None, s3_16=None, res_tok=None):
        """cat_lp (True bf16 | 2 fp16): cat(s3,out) / cat(s2,out) are written in that 16-bit type only
        (they feed the next block's 16-bit convolutions and nothing else); `out` stays fp32.
        s2_16 / s3_16: s2 and s3 were written in 16 bits only by their convolutions (Conv2dFn y_lp);
        s2 / s3 are then the proxies on the autograd edges and the gradients go back in 16 bits."""
        ctx.set_materialize_grads(False)
        ctx.res_tok = res_tok       # x1 is also the input of this block's first stream-1 conv (ResidualToken)
        ctx.s_lp = s2_16 is not None
        ctx.cat_lp = bool(want_cat and cat_lp)
        if ctx.s_lp:
            assert s3_16 is not None and s2_16.dtype == s3_16.dtype and s2_16.is_contiguous() and s3_16.is_contiguous()
            s2, s3 = s2_16, s3_16
        for t in (x1, s1) + (() if ctx.s_lp else (s2, s3)):
            _chk(t)
        B, H, W_, Cc = x1.shape
        out = torch.empty_like(x1)
        x2n = x3n = None
        if want_cat:
            x2n = torch.empty((B, H, W_, 2 * Cc), dtype=_wd(cat_lp), device=x1.device)
            x3n = torch.empty((B, H, W_, 2 * Cc), dtype=_wd(cat_lp), device=x1.device)
        L.call("mmh_patblock_gate_fwd", _ptr(x1), _ptr(s1), _ptr(s2), _ptr(s3), _ptr(out),
               _ptr(x2n), _ptr(x3n), B * H * W_, Cc, _dt(cat_lp), _tdt(s2), _stream())
        ctx.save_for_backward(s1, s2, s3)
        ctx.want_cat = want_cat
        if want_cat and cat_lp:
            ctx.mark_non_differentiable(x2n, x3n)
            return out, lp_proxy(x2n.shape, x1.device), lp_proxy(x3n.shape, x1.device), x2n, x3n
        if want_cat:
            return out, x2n, x3n
        return out, None, None

    @staticmethod
    def backward(ctx, g_out, g_x2n, g_x3n, _a=None, _b=None):
        s1, s2, s3 = ctx.saved_tensors
        B, H, W_, Cc = s1.shape
        g_out = None if g_out is None else g_out.contiguous()
        if ctx.cat_lp:      # the cats fed 16-bit convolutions: their gradients come back in 16 bits
            g_x2n = lp_grad_in(g_x2n, "GateFn (cat(s3,out))")
            g_x3n = lp_grad_in(g_x3n, "GateFn (cat(s2,out))")
        else:
            g_x2n = None if g_x2n is None else g_x2n.contiguous()
            g_x3n = None if g_x3n is None else g_x3n.contiguous()
        gx1 = torch.empty_like(s1); gs1 = torch.empty_like(s1)
        gs2 = torch.empty_like(s2); gs3 = torch.empty_like(s3)
        L.call("mmh_patblock_gate_bwd", _ptr(g_out), _ptr(g_x2n), _ptr(g_x3n), _ptr(s1), _ptr(s2),
               _ptr(s3), _ptr(gx1), _ptr(gs1), _ptr(gs2), _ptr(gs3), B * H * W_, Cc,
               L.F32 if g_x2n is None else _tdt(g_x2n), _tdt(s2), _tdt(gs2), _stream())
        if ctx.s_lp:
            gs2, gs3 = lp_grad_out(gs2), lp_grad_out(gs3)
        if ctx.res_tok is not None and ctx.res_tok.park(gx1):
            gx1 = None          # joins the gradient of x1 inside the stream-1 conv's dgrad
        return gx1, gs1, gs2, gs3, None, None, None, None, None


# The PATBlock gate with the block's last InstanceNorm inside (16-bit mode): the norm's output s1 has one reader, the gate,
# so it is never written - the gate applies scale / shift to the conv output itself, and its backward kernel leaves the norm
# backward's plane sums (mmh_patblock_gate_norm_fwd / _bwd: bit-identical to the separate launches).  MMH_GATE_NORM=0: off.
USE_GATE_NORM = os.environ.get("MMH_GATE_NORM", "1") != "0"


def gate_norm_ok(B, rows, C):
    return bool(USE_GATE_NORM and L.load().mmh_patblock_gate_norm_supported(B, rows, C))


class GateNormFn(torch.autograd.Function):
    """GateFn with s1 = InstanceNorm(y2) computed inside: y2 is the proxy of the stream-1 conv output that exists in 16 bits
    only (y2_16, Conv2dFn y_lp).  Same outputs and conventions as GateFn."""

    @staticmethod
    def forward(ctx, x1, y2, s2, s3, want_cat, cat_lp, y2_16, s2_16, s3_16, res_tok):
        ctx.set_materialize_grads(False)
        ctx.res_tok = res_tok
        ctx.s_lp = s2_16 is not None
        ctx.cat_lp = bool(want_cat and cat_lp)
        if ctx.s_lp:
            assert s3_16 is not None and s2_16.dtype == s3_16.dtype and s2_16.is_contiguous() and s3_16.is_contiguous()
            s2, s3 = s2_16, s3_16
        _chk(x1, "x1")
        assert y2_16.dtype != torch.float32 and y2_16.is_contiguous() and tuple(y2_16.shape) == tuple(x1.shape)
        B, H, W_, Cc = x1.shape
        rows = H * W_
        fast = raw_norm_stats_finalize_pending(y2_16, B)
        if fast is not None:
            mean, scale, shift, invstd, _ = fast
        else:
            mean, m2, _ = raw_norm_stats(y2_16, B)
            scale, shift, invstd = raw_norm_finalize(mean, m2, rows, None, None, None, None)
        out = torch.empty_like(x1)
        x2n = x3n = None
        if want_cat:
            x2n = torch.empty((B, H, W_, 2 * Cc), dtype=_wd(cat_lp), device=x1.device)
            x3n = torch.empty((B, H, W_, 2 * Cc), dtype=_wd(cat_lp), device=x1.device)
        L.call("mmh_patblock_gate_norm_fwd", _ptr(x1), _ptr(y2_16), _ptr(scale), _ptr(shift), _ptr(s2), _ptr(s3), _ptr(out),
               _ptr(x2n), _ptr(x3n), B, rows, Cc, _tdt(y2_16), _dt(cat_lp), _tdt(s2), _stream())
        ctx.save_for_backward(y2_16, mean, invstd, scale, shift, s2, s3)
        ctx.want_cat = want_cat
        if want_cat and cat_lp:
            ctx.mark_non_differentiable(x2n, x3n)
            return out, lp_proxy(x2n.shape, x1.device), lp_proxy(x3n.shape, x1.device), x2n, x3n
        if want_cat:
            return out, x2n, x3n
        return out, None, None

    @staticmethod
    def backward(ctx, g_out, g_x2n, g_x3n, _a=None, _b=None):
        y2, mean, invstd, scale, shift, s2, s3 = ctx.saved_tensors
        B, H, W_, Cc = y2.shape
        rows = H * W_
        g_out = None if g_out is None else g_out.contiguous()
        if ctx.cat_lp:
            g_x2n = lp_grad_in(g_x2n, "GateNormFn (cat(s3,out))")
            g_x3n = lp_grad_in(g_x3n, "GateNormFn (cat(s2,out))")
        else:
            g_x2n = None if g_x2n is None else g_x2n.contiguous()
            g_x3n = None if g_x3n is None else g_x3n.contiguous()
        gx1 = torch.empty((B, H, W_, Cc), dtype=torch.float32, device=y2.device)
        gs1 = torch.empty_like(gx1)
        gs2 = torch.empty_like(s2); gs3 = torch.empty_like(s3)
        sum1 = _empty((B, Cc), gx1); sum2 = _empty((B, Cc), gx1)
        ws = _ws(L.load().mmh_norm_bwd_ws_bytes(B, rows, Cc), gx1)
        L.call("mmh_patblock_gate_norm_bwd", _ptr(g_out), _ptr(g_x2n), _ptr(g_x3n), _ptr(y2), _ptr(scale), _ptr(shift),
               _ptr(mean), _ptr(invstd), _ptr(s2), _ptr(s3), _ptr(gx1), _ptr(gs1), _ptr(gs2), _ptr(gs3), _ptr(sum1), _ptr(sum2),
               _ptr(ws), ws.numel() * 4, B, rows, Cc, _tdt(y2), L.F32 if g_x2n is None else _tdt(g_x2n), _tdt(s2), _stream())
        dy2 = torch.empty_like(y2)      # the norm backward's apply pass: the gradient of the conv output, in 16 bits
        L.call("mmh_norm_bwd_apply", _ptr(gs1), None, _ptr(y2), _ptr(mean), _ptr(invstd), None, _ptr(sum1), _ptr(sum2),
               float(rows), B, rows, Cc, 0, 0.0, _ptr(dy2), L.F32, _tdt(y2), _tdt(dy2), _stream())
        if ctx.s_lp:
            gs2, gs3 = lp_grad_out(gs2), lp_grad_out(gs3)
        if ctx.res_tok is not None and ctx.res_tok.park(gx1):
            gx1 = None
        return gx1, lp_grad_out(dy2), gs2, gs3, None, None, None, None, None, None


# --------------------------------------------------------------------------- losses
class BCEWithLogitsConstFn(torch.autograd.Function):
    """weight * mean(BCEWithLogits(x, target)) for a constant target (GANLoss,
    models/network_utils.py:129-163)."""

    @staticmethod
    def forward(ctx, x, target, weight):
        _chk(x, "x")
        n = x.numel()
        ws = _ws(L.load().mmh_reduce_ws_bytes(n), x)
        out = _empty((), x)
        L.call("mmh_bce_logits_fwd", _ptr(x), n, float(target), float(weight), float(n), _ptr(out),
               _ptr(ws), ws.numel() * 4, _stream())
        ctx.cfg = (float(target), float(weight))
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        target, weight = ctx.cfg
        g = g.contiguous().float()
        dx = torch.empty_like(x)
        L.call("mmh_bce_logits_bwd", _ptr(x), x.numel(), target, weight, float(x.numel()), _ptr(g),
               _ptr(dx), _stream())
        return dx, None, None


class BCEWithLogitsHalvesFn(torch.autograd.Function):
    """GANLoss of a discriminator that ran ONCE on cat(real batch, fake batch) (MMHandModel.backward_D_basic under
    InstanceNorm): (weight * mean BCE(x[:B], 1), weight * mean BCE(x[B:], 0)) straight from the two halves of the logits
    and, backward, both halves of dx written in place - no slice views, so no zero-fill / copy / add passes of autograd's
    slice backward over the [2B, H/4, W/4, 256] map."""

    @staticmethod
    def forward(ctx, x, weight):
        _chk(x, "x")
        assert x.shape[0] % 2 == 0
        n = x.numel() // 2
        ws = _ws(L.load().mmh_reduce_ws_bytes(n), x)
        out = torch.empty((2,), dtype=torch.float32, device=x.device)
        for h, target in ((0, 1.0), (1, 0.0)):
            L.call("mmh_bce_logits_fwd", C.c_void_p(x.data_ptr() + 4 * n * h), n, target, float(weight), float(n),
                   C.c_void_p(out.data_ptr() + 4 * h), _ptr(ws), ws.numel() * 4, _stream())
        ctx.weight = float(weight)
        ctx.save_for_backward(x)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_real, g_fake):
        (x,) = ctx.saved_tensors
        n = x.numel() // 2
        dx = torch.empty_like(x)
        for h, (target, g) in enumerate(((1.0, g_real), (0.0, g_fake))):
            if g is None:
                dx[h * (x.shape[0] // 2):(h + 1) * (x.shape[0] // 2)].zero_()
                continue
            g = g.contiguous().float()
            L.call("mmh_bce_logits_bwd", C.c_void_p(x.data_ptr() + 4 * n * h), n, target, ctx.weight, float(n), _ptr(g),
                   C.c_void_p(dx.data_ptr() + 4 * n * h), _stream())
        return dx, None


def _pair_loss_fwd(kind, ctx, a, b, weight, denom):
    _chk(a, "a"); _chk(b, "b")
    n = a.numel()
    ws = _ws(L.load().mmh_reduce_ws_bytes(n), a)
    out = _empty((), a)
    L.call(f"mmh_{kind}_fwd", _ptr(a), _ptr(b), n, float(weight), float(denom), _ptr(out), _ptr(ws),
           ws.numel() * 4, _stream())
    ctx.cfg = (float(weight), float(denom))
    ctx.save_for_backward(a, b)
    return out


def _pair_loss_bwd(kind, ctx, g):
    a, b = ctx.saved_tensors
    weight, denom = ctx.cfg
    g = g.contiguous().float()
    da = torch.empty_like(a)
    L.call(f"mmh_{kind}_bwd", _ptr(a), _ptr(b), a.numel(), weight, denom, _ptr(g), _ptr(da), _stream())
    return da, None, None, None


class L1MeanFn(torch.autograd.Function):
    """weight * sum|a-b| / denom (F.l1_loss, losses/L1_plus_perceptualLoss.py:37,66-67).
    ``denom`` is the logical element count (zero pad lanes contribute nothing)."""

    @staticmethod
    def forward(ctx, a, b, weight, denom):
        return _pair_loss_fwd("l1", ctx, a, b, weight, denom)

    @staticmethod
    def backward(ctx, g):
        return _pair_loss_bwd("l1", ctx, g)


class MSEMeanFn(torch.autograd.Function):
    """weight * sum (a-b)^2 / denom (F.mse_loss: the --percep_is_l1 0 branch,
    losses/L1_plus_perceptualLoss.py:68-71)."""

    @staticmethod
    def forward(ctx, a, b, weight, denom):
        return _pair_loss_fwd("mse", ctx, a, b, weight, denom)

    @staticmethod
    def backward(ctx, g):
        return _pair_loss_bwd("mse", ctx, g)


class MaxPool2x2Fn(torch.autograd.Function):
    """nn.MaxPool2d(2, 2) on NHWC fp32: the pooling layers of vgg19.features beyond index 3
    (losses/L1_plus_perceptualLoss.py:22-27, --perceptual_layers)."""

    @staticmethod
    def forward(ctx, x):
        _chk(x, "x")
        B, H, W_, Cc = x.shape
        y = _empty((B, H // 2, W_ // 2, Cc), x)
        L.call("mmh_maxpool2x2_fwd", _ptr(x), B, H, W_, Cc, _ptr(y), _stream())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        B, H, W_, Cc = x.shape
        dx = torch.empty_like(x)
        L.call("mmh_maxpool2x2_bwd", _ptr(x), _ptr(g.contiguous()), B, H, W_, Cc, _ptr(dx), _stream())
        return dx


# --------------------------------------------------------------------------- layout
def _plane(t, nchw):
    """PlaneSrc for a logical-NCHW tensor (any strides) or a physical NHWC tensor."""
    if nchw:
        sb, sc, sh, sw = t.stride()
        Cc = t.shape[1]
    else:
        sb, sh, sw, sc = t.stride()
        Cc = t.shape[3]
    return L.PlaneSrc(t.data_ptr(), Cc, sb, sc, sh, sw)


def raw_pack(srcs, B, H, W_, Cd, device, out=None, twin=0, twin_out=None, only16=False, keep_twin=False):
    """srcs: list of (tensor, is_nchw, n_channels).  Returns NHWC [B,H,W,Cd] (zero padded); out: write into this
    contiguous [B,H,W,Cd] tensor (e.g. one half of a two-batch buffer) instead of a new one.
    twin (operand type of the 16-bit mode): the same pass also writes the 16-bit copy with channels padded to a multiple of 8
    that a 16-bit stem reads (lp16_pad8 of the result) - into twin_out, else parked for lp16_pad8 (pack_twin_put).
    only16: the fp32 tensor is not written at all; returns the 16-bit copy."""
    arr = (L.PlaneSrc * len(srcs))()
    for i, (t, nchw, nch) in enumerate(srcs):
        assert t.dtype == torch.float32 and t.is_cuda
        p = _plane(t, nchw)
        p.C = nch
        arr[i] = p
    c8 = (Cd + 7) // 8 * 8
    tiled = USE_PACK_TWIN and Cd <= 56
    if only16:
        assert twin and out is None and tiled, "a 16-bit-only pack needs the tiled kernel (Cd <= 56, MMH_PACK_TWIN)"
    elif out is None:
        out = torch.empty((B, H, W_, Cd), dtype=torch.float32, device=device)
    else:
        assert tuple(out.shape) == (B, H, W_, Cd) and out.is_contiguous() and out.dtype == torch.float32
    if not tiled:
        L.call("mmh_pack_nhwc", arr, len(srcs), _ptr(out), B, H, W_, Cd, 0, _stream())
        return out
    x16p = None
    if twin:
        x16p = twin_out if twin_out is not None else torch.empty((B, H, W_, c8), dtype=_wd(twin), device=device)
        assert x16p.is_contiguous() and x16p.dtype == _wd(twin) and tuple(x16p.shape) == (B, H, W_, c8)
    L.call("mmh_pack_nhwc_lp16", arr, len(srcs), _ptr(out), _ptr(x16p), B, H, W_, Cd, c8, _dt(twin) if twin else L.BF16,
           _stream())
    if only16:
        return x16p
    if twin and twin_out is None:
        pack_twin_put(out, x16p, keep=keep_twin)
    return out


def raw_unpack(nhwc, dsts):
    """dsts: list of (tensor or None, is_nchw, n_channels): scatter channels of nhwc into them."""
    B, H, W_, Cd = nhwc.shape
    arr = (L.PlaneSrc * len(dsts))()
    for i, (t, nchw, nch) in enumerate(dsts):
        if t is None:
            arr[i] = L.PlaneSrc(None, nch, 0, 0, 0, 0)
        else:
            p = _plane(t, nchw)
            p.C = nch
            arr[i] = p
    L.call("mmh_pack_nhwc", arr, len(dsts), _ptr(nhwc), B, H, W_, Cd, 1, _stream())


class PackFn(torch.autograd.Function):
    """cat(channels) + zero pad to a multiple of 4 + (NCHW|NHWC) -> NHWC, one kernel; replaces
    torch.cat at models/MMHandModel.py:216-220,238,242,278-289.  Sources are given as
    (tensor, is_nchw) pairs flattened: PackFn.apply(Cd, t0, nchw0, c0, t1, nchw1, c1, ...)."""

    @staticmethod
    def forward(ctx, Cd, *args):
        """Cd: the packed channel count, or (Cd, lp): also leave the padded 16-bit copy for a 16-bit stem (raw_pack twin)"""
        Cd, twin = Cd if isinstance(Cd, tuple) else (Cd, 0)
        srcs = [(args[i], args[i + 1], args[i + 2]) for i in range(0, len(args), 3)]
        t0, nchw0, _ = srcs[0]
        if nchw0:
            B, _, H, W_ = t0.shape
        else:
            B, H, W_, _ = t0.shape
        ctx.meta = [(tuple(t.shape), nchw, nch) for t, nchw, nch in srcs]
        return raw_pack(srcs, B, H, W_, Cd, t0.device, twin=twin)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        grads = [None]
        dsts = []
        for i, (shape, nchw, nch) in enumerate(ctx.meta):
            if ctx.needs_input_grad[1 + 3 * i]:
                t = torch.zeros(shape, dtype=torch.float32, device=g.device)
                dsts.append((t, nchw, nch))
                grads += [t, None, None]
            else:
                dsts.append((None, nchw, nch))
                grads += [None, None, None]
        raw_unpack(g, dsts)
        return tuple(grads)


def nhwc_to_nchw_view(t, Cc=None):
    """Zero-copy logical-NCHW view of a physical NHWC tensor (first Cc channels)."""
    v = t.permute(0, 3, 1, 2)
    return v if Cc is None or Cc == t.shape[3] else v[:, :Cc]


# --------------------------------------------------------------------------- optimiser
def adam_step(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0, skip_flag=None, loss_scale=None):
    """torch.optim.Adam.step (models/MMHandModel.py:90-98) on flat buffers, one launch.
    skip_flag: int32 device scalar; non-zero makes the launch a no-op (overflow skip).
    loss_scale: fp32 device scalar the gradient is divided by (dynamic loss scaling)."""
    for t in (p, g, m, v):
        _chk(t)
    if skip_flag is not None:
        assert skip_flag.dtype == torch.int32 and skip_flag.is_cuda
    if loss_scale is not None:
        assert loss_scale.dtype == torch.float32 and loss_scale.is_cuda
    L.call("mmh_adam_step", _ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), float(lr), float(beta1),
           float(beta2), float(eps), int(step), float(grad_scale), _ptr(skip_flag), _ptr(loss_scale), _stream())


def grad_nonfinite(g, flag_out, flag_in=None, own_out=None):
    """flag_out = (flag_in or 0) | any(!isfinite(g)) on the device (MMHandModel.loss_backward,
    models/MMHandModel.py:294-308); own_out = any(!isfinite(g)) alone; int32 one-element tensors."""
    _chk(g)
    for f in (flag_out, flag_in, own_out):
        assert f is None or (f.dtype == torch.int32 and f.is_cuda and f.numel() == 1)
    L.call("mmh_grad_nonfinite", _ptr(g), g.numel(), _ptr(flag_in), _ptr(flag_out), _ptr(own_out), _stream())


# apex.amp dynamic loss scaler defaults (apex/amp/scaler.py: init 2**16, factor 2, window 2000, max 2**24)
LOSS_SCALE_INIT, LOSS_SCALE_GROWTH, LOSS_SCALE_BACKOFF = 65536.0, 2.0, 0.5
LOSS_SCALE_WINDOW, LOSS_SCALE_MIN, LOSS_SCALE_MAX = 2000, 1.0, 2.0 ** 24


def loss_scale_update(state, overflow, window=None):
    """state: fp32 [2] = {scale, clean steps}; overflow: int32 [1] (this loss's own flag)."""
    assert state.dtype == torch.float32 and state.is_cuda and state.numel() == 2 and state.is_contiguous()
    assert overflow.dtype == torch.int32 and overflow.is_cuda and overflow.numel() == 1
    L.call("mmh_loss_scale_update", _ptr(state), _ptr(overflow), LOSS_SCALE_GROWTH, LOSS_SCALE_BACKOFF,
           int(window or LOSS_SCALE_WINDOW), LOSS_SCALE_MIN, LOSS_SCALE_MAX, _stream())


# --------------------------------------------------------------------------- pose maps
def pose_heatmaps(uv, H, W_, sigma=6.0):
    """uv: float64 CUDA tensor [n_maps, 2] (x, y) -> fp32 [n_maps, H, W] (generic_dataset.py:191-242)."""
    assert uv.dtype == torch.float64 and uv.is_cuda and uv.is_contiguous()
    n = uv.shape[0]
    out = torch.empty((n, H, W_), dtype=torch.float32, device=uv.device)
    L.call("mmh_pose_heatmaps", _ptr(uv), n, H, W_, float(sigma), _ptr(out), _stream())
    return out


def map_to_cord(maps, threshold=0.1):
    """maps: fp32 CUDA [n_maps, H, W] -> int32 [n_maps, 2] (y, x) or -1 (util/util.py:94-114)."""
    _chk(maps, "maps")
    n, H, W_ = maps.shape
    out = torch.empty((n, 2), dtype=torch.int32, device=maps.device)
    L.call("mmh_map_to_cord", _ptr(maps), n, H, W_, float(threshold), _ptr(out), _stream())
    return out


def decode_inputs(img1, img2, dep1, dep2, uv1, uv2, sigma=6.0):
    """On-device input pipeline (data/generic_dataset.py:133-180): uint8 BGR images + depth PNGs
    [B,H,W,3] and float64 joints [B,21,2] -> the stems' NHWC buffers (x_H1, x_H2, x_P, x_D)."""
    B, H, W_, _ = img1.shape
    for t in (img1, img2, dep1, dep2):
        assert t.dtype == torch.uint8 and t.is_cuda and t.is_contiguous() and tuple(t.shape) == (B, H, W_, 3)
    for t in (uv1, uv2):
        assert t.dtype == torch.float64 and t.is_cuda and t.is_contiguous() and tuple(t.shape) == (B, 21, 2)
    dev = img1.device
    xh1 = torch.empty((B, H, W_, 4), dtype=torch.float32, device=dev)
    xh2 = torch.empty((B, H, W_, 4), dtype=torch.float32, device=dev)
    xp = torch.empty((B, H, W_, 44), dtype=torch.float32, device=dev)
    xd = torch.empty((B, H, W_, 8), dtype=torch.float32, device=dev)
    L.call("mmh_decode_inputs", _ptr(img1), _ptr(img2), _ptr(dep1), _ptr(dep2), _ptr(uv1), _ptr(uv2),
           B, H, W_, float(sigma), _ptr(xh1), _ptr(xh2), _ptr(xp), _ptr(xd), _stream())
    return xh1, xh2, xp, xd
