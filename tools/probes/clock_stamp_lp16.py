"""The clock the chip holds under conv_lp16h2_kernel (MI355X_MICROARCH.md 'DVFS give-back' item 6): a diagnostic
build (make AB=1 ... OUT=../libmmhand_hip_ab.so) stamps s_memtime / s_memrealtime around the kernel body in wave 0 of every
workgroup (mmh_set_option("lp16_dbg", 4096)); delta s_memtime / delta s_memrealtime x 100 MHz, median over workgroups, after
>= 2 s of back-to-back launches on random data.  Rows: the kernel as built and its timing-only ablations (DESIGN 4.3d's table:
433 us as built, 392 us for the MFMAs and the loop alone), so that the wall-time ratio can be split into cycles and clock.

    MMH_LIB_PATH=mmhand_amd/libmmhand_hip_ab.so python tools/probes/clock_stamp_lp16.py"""
import ctypes as C
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch                                              # noqa: E402
from mmhand_amd import lib, ops                           # noqa: E402

L = lib.load()
dev = torch.device("cuda:0")
B, H, Cin, Cout = 32, 64, 512, 512
x16 = torch.randn(B, H, H, Cin, device=dev).bfloat16()
w = torch.randn(3, 3, Cin, Cout, device=dev) * 0.05
dy16 = torch.randn(B, H, H, Cout, device=dev).bfloat16()
ops.bump_weights_epoch()
STAMP = 4096
NWG = 256
buf = (C.c_uint64 * (2 * NWG))()


def run(dbg, mode, seconds=2.5, what=""):
    src = x16 if mode == 0 else dy16
    fn = lambda: ops.raw_conv3x3_lp16(src, w, None, True, lib.ACT_NONE, True, mode, out16=True)    # noqa: E731
    lib.check(L.mmh_set_option(b"lp16_dbg", dbg | STAMP), "opt")
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            fn()
        n += 50
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    got = L.mmh_lp16_clock_stamps(buf, NWG)
    if got <= 0:
        raise SystemExit("no stamps: this is not an A/B build (MMH_LIB_PATH=.../libmmhand_hip_ab.so, make AB=1)")
    ghz = [buf[2 * i] / buf[2 * i + 1] * 0.1 for i in range(got) if buf[2 * i + 1] > 0]
    cyc = [buf[2 * i] for i in range(got) if buf[2 * i + 1] > 0]
    lib.check(L.mmh_set_option(b"lp16_dbg", 0), "opt")
    fl = 2.0 * B * H * H * Cin * Cout * 9
    print(f"{what:58s} {us:7.1f} us  {fl / us / 1e6:6.0f} TF   clock {statistics.median(ghz):.3f} GHz (min {min(ghz):.3f} max {max(ghz):.3f}, "
          f"{len(ghz)} workgroups)   shader cycles per workgroup {statistics.median(cyc) / 1e3:.0f} k   [{n} launches before]", flush=True)


print(f"conv_lp16h2_kernel, {Cin}->{Cout} @ {H}x{H}, B={B}, bf16, 16-bit output; stamps around the whole kernel body")
run(0, 0, what="fprop as built")
run(1 | 2, 0, what="fprop without any DMA (timing only)")
run(1 | 2 | 64, 0, what="fprop without DMA and fragment reads (timing only)")
run(1 | 2 | 64 | 256 | 512, 0, what="fprop: MFMAs and loop bookkeeping only (timing only)")
run(256, 0, what="fprop as built, no epilogue (timing only)")
run(0, 2, what="reflect-fold dgrad as built")
run(0, 0, what="fprop as built (again, last)")
