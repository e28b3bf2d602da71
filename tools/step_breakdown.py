"""Per-call GPU time of one optimize_parameters(): every C-ABI call is bracketed with events and the
times are summed per (entry point, shape).  Also times the torch kernels in between as 'torch glue'
(step time - sum of bracketed calls).

    python tools/step_breakdown.py [--dtype bf16] [--batch 32] [--size 256] [--top 60]
"""
import argparse
import collections
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmhand_amd import lib as L                                   # noqa: E402
from mmhand_amd.mmhand_model import MMHandModel                   # noqa: E402
from mmhand_amd.options import default_train_opt                  # noqa: E402


def describe(name, args):
    for a in args:
        d = getattr(a, "_obj", None)
        if isinstance(d, L.ConvDesc):
            mode = ""
            if name in ("mmh_conv3x3_lp16", "mmh_conv3x3_lp16_dgrad_add"):
                mode = " mode%d" % args[1]
            return "B%d %dx%d %d->%d k%d s%d%s" % (d.B, d.H, d.W, d.Cin, d.Cout, d.kh, d.stride, mode)
    ints = [str(a) for a in args if isinstance(a, int) and not isinstance(a, bool) and 0 < a < 10 ** 9]
    return " ".join(ints[:6])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--top", type=int, default=70)
    ap.add_argument("--norm", default="instance", choices=["instance", "batch"])
    ap.add_argument("--set", action="append", default=[], help="mmh_set_option key=value (repeatable)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    for kv in a.set:
        k, v = kv.split("=")
        L.check(L.load().mmh_set_option(k.encode(), int(v)), "mmh_set_option")
    opt = default_train_opt(batchSize=a.batch, norm=a.norm, name="breakdown", checkpoints_dir="/tmp/mmh_bench",
                            opt_level={"f32": "O0", "bf16": "O1", "fp16": "O1_FP16"}[a.dtype])
    model = MMHandModel(opt)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import synthetic_batch_gpu
    model.set_input(synthetic_batch_gpu(a.batch, a.size, a.size, 49, dev))
    for _ in range(3):
        model.optimize_parameters()
    torch.cuda.synchronize()

    rec = []
    real_call = L.call

    def timed_call(name, *args):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        real_call(name, *args)
        e1.record()
        rec.append((name, describe(name, args), e0, e1))

    s0 = torch.cuda.Event(enable_timing=True)
    s1 = torch.cuda.Event(enable_timing=True)
    L.call = timed_call
    import mmhand_amd.ops as ops
    s0.record()
    model.optimize_parameters()
    s1.record()
    L.call = real_call
    torch.cuda.synchronize()
    step_ms = s0.elapsed_time(s1)
    agg = collections.OrderedDict()
    by_name = collections.Counter()
    for name, desc, e0, e1 in rec:
        t = e0.elapsed_time(e1)
        k = (name, desc)
        c = agg.setdefault(k, [0, 0.0])
        c[0] += 1
        c[1] += t
        by_name[name] += t
    tot = sum(v[1] for v in agg.values())
    print("step %.1f ms (with %d event pairs); bracketed calls %.1f ms; torch glue + gaps %.1f ms" %
          (step_ms, len(rec), tot, step_ms - tot))
    print("\n-- by entry point")
    for n, t in by_name.most_common():
        print("%-34s %8.2f ms" % (n, t))
    print("\n-- by (entry point, shape)")
    for (name, desc), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:a.top]:
        print("%-30s %-34s x%-3d %8.2f ms  %8.1f us" % (name, desc, n, t, 1e3 * t / n))


if __name__ == "__main__":
    main()
