"""conv_lp16q_kernel (lp16_shape 20: one wave per SIMD, 512 registers, wave tile 8 rows x 16 pixels x 128 channels) against
conv_lp16h2_kernel (19: two waves per SIMD): results bit for bit (same k order, same MFMA shape), then interleaved timings
of every entry point of the 16-bit 3x3 stack - fprop (16-bit epilogue, bias), fprop + statistics, zero-pad dgrad, reflect-fold
dgrad, dgrad + addend (fp32) - on the step's shapes.  MMH_Q_ABLATE=1 adds the timing-only switches (lp16_dbg; results wrong)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd import ops, lib
L = lib.load(); dev = torch.device("cuda:0")
def timeit(fn, iters=10):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
def setopt(k, v): lib.check(L.mmh_set_option(k.encode(), v), "set")
shapes = ((32, 64, 256, 256), (32, 64, 512, 512), (32, 64, 512, 256), (4, 128, 256, 256), (2, 32, 256, 256), (1, 48, 256, 512))
ok_all = True
for (B, H, Cin, Cout) in shapes:
    x = torch.randn(B, H, H, Cin, device=dev); dy = torch.randn(B, H, H, Cout, device=dev)
    w = torch.randn(3, 3, Cin, Cout, device=dev) * 0.05; bias = torch.randn(Cout, device=dev)
    addend = torch.randn(B, H, H, Cin, device=dev)
    xb, dyb = ops.lp16_twin(x, True), ops.lp16_twin(dy, True)
    flop = 2.0 * B * H * H * Cin * Cout * 9
    fold_ok = H % 16 == 0 and H >= 32 and Cin % 256 == 0
    fns = {"fprop": lambda: ops.raw_conv3x3_lp16(xb, w, bias, True, 0, True, 0, out16=True),
           "fprop f32 relu": lambda: ops.raw_conv3x3_lp16(xb, w, bias, True, 1, True, 0),
           "dgrad": lambda: ops.raw_conv3x3_lp16(dyb, w, None, False, 0, True, 1, out16=True)}
    if H % 16 == 0:
        def fps():
            y = ops.raw_conv3x3_lp16(xb, w, bias, True, 0, True, 0, out16=True, want_stats=True)
            st = ops._pending_stats_take(y) if hasattr(ops, "_pending_stats_take") else None
            return y if st is None else (y, st)
        fns["fprop+stats"] = lambda: ops.raw_conv3x3_lp16(xb, w, bias, True, 0, True, 0, out16=True, want_stats=True)
    if fold_ok:
        fns["fold dgrad"] = lambda: ops.raw_conv3x3_lp16(dyb, w, None, True, 0, True, 2, out16=True)
        fns["fold dgrad + addend"] = lambda: ops.raw_conv3x3_lp16(dyb, w, None, True, 0, True, 2, addend=addend)
    line = []
    for name, fn in fns.items():
        setopt("lp16_shape", 19); ref = fn()
        setopt("lp16_shape", 20); outs = [fn() for _ in range(3)]
        same = all(torch.equal(o, ref) for o in outs)
        ok_all &= same
        if not same:
            d = max(float((o.float() - ref.float()).abs().max()) for o in outs)
            line.append(f"{name}: MISMATCH max|d|={d:.3e} (ref max {float(ref.float().abs().max()):.3e})")
        else:
            line.append(f"{name}: bit-identical")
    print(f"B{B} {H}x{H} {Cin}->{Cout}: " + " | ".join(line), flush=True)
    if B * H * H < 32 * 64 * 64 and not (B == 4 and H == 128):
        continue
    def mk(shape, fn, dbg=0):
        def run():
            setopt("lp16_shape", shape)
            if dbg: setopt("lp16_dbg", dbg)
            r = fn()
            if dbg: setopt("lp16_dbg", 0)
            return r
        return run
    variants = {}
    for name, fn in fns.items():
        variants[f"h2 {name}"] = mk(19, fn); variants[f"q {name}"] = mk(20, fn)
    if os.environ.get("MMH_Q_ABLATE") == "1":
        for dbg, what in ((1, "no weight DMA"), (2, "no halo DMA"), (3, "no DMA"), (32, "all DMA behind the barrier")):
            variants[f"q fprop [{what}]"] = mk(20, fns["fprop"], dbg)
            variants[f"h2 fprop [{what}]"] = mk(19, fns["fprop"], dbg)
        for nm in ("fprop", "dgrad", "fold dgrad", "fold dgrad + addend", "fprop+stats"):
            if nm in fns:
                variants[f"h2 {nm} [round-4 barrier: full LDS drain + fence]"] = mk(19, fns[nm], 1024)
        for dbg, what in ((3 + 64, "no DMA, no fragment reads"), (3 + 256, "no DMA, no epilogue"), (3 + 512, "no DMA, no barrier"),
                          (3 + 64 + 256 + 512, "MFMAs and loop bookkeeping only"), (256, "no epilogue")):
            variants[f"q fprop [{what}]"] = mk(20, fns["fprop"], dbg)
    res = {k: [] for k in variants}
    for f in variants.values(): f()
    torch.cuda.synchronize()
    for r in range(5):
        for k, f in variants.items(): res[k].append(timeit(f))
    for k, v in res.items():
        m = statistics.median(v)
        print(f"    {k}: {m*1e3:.0f} us ({flop/m/1e9:.0f} TF)", flush=True)
setopt("lp16_shape", 19)
print("ALL BIT-IDENTICAL" if ok_all else "MISMATCHES", flush=True)
