"""Does it train: N iterations of optimize_parameters() on ONE fixed synthetic batch (overfit), the L1 and GAN
losses at the start and the end - fp32 with the norm fused into the Winograd transforms, the same unfused, and
the 16-bit mode.

    python tools/train_sanity.py [--iters 40] [--batch 8]
"""
import argparse
import os
import random
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(a):
    import torch
    from bench import synthetic_batch_gpu
    from mmhand_amd.mmhand_model import MMHandModel
    from mmhand_amd.options import default_train_opt
    random.seed(0)
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    opt = default_train_opt(batchSize=a.batch, norm=a.norm, name="sanity", checkpoints_dir="/tmp/mmh_sanity",
                            opt_level={"f32": "O0", "bf16": "O1", "fp16": "O1_FP16"}[a.dtype], graph_step=bool(a.graph))
    model = MMHandModel(opt)
    model.set_input(synthetic_batch_gpu(a.batch, 256, 256, 49, dev))
    hist = []
    for it in range(a.iters):
        model.optimize_parameters()
        if it % 10 == 0 or it == a.iters - 1:
            e = model.get_current_errors()
            hist.append((it, {k: round(float(v), 4) for k, v in e.items()}))
    torch.cuda.synchronize()
    for it, e in hist:
        print(f"  it {it:3d}: {e}")
    first, last = hist[0][1], hist[-1][1]
    key = next(k for k in first if "L1" in k)
    ok = all(v == v and abs(v) < 1e4 for v in last.values()) and last[key] < first[key]
    if a.graph:
        print(f"  --graph_step: {model.graph_replays} replays of the captured iteration, graph_error = {model.graph_error}")
        ok = ok and model.graph_error is None and model.graph_replays == a.iters - model._graph_warm
    print(f"  {key}: {first[key]} -> {last[key]}  {'OK' if ok else 'NOT DECREASING'}")
    sys.exit(0 if ok else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--norm", default="instance")
    ap.add_argument("--dtype", default=None)
    ap.add_argument("--graph", type=int, default=0, help="1: --graph_step (the iteration replayed from a captured hipGraph)")
    ap.add_argument("--graph-soak", action="store_true",
                    help="the three precisions with --graph_step beside their eager runs (round 6: a soak of the replayed step)")
    a = ap.parse_args()
    if a.dtype is not None:
        return child(a)
    rc = 0
    runs = [("f32", {}, 0), ("f32", {"MMH_FUSE_NORMACT": "0"}, 0), ("bf16", {}, 0), ("fp16", {}, 0)]
    if a.graph_soak:
        runs = [("f32", {}, 0), ("f32", {}, 1), ("bf16", {}, 0), ("bf16", {}, 1), ("fp16", {}, 0), ("fp16", {}, 1)]
    for dtype, env, graph in runs:
        print(f"== {dtype} {env or ''}{' --graph_step' if graph else ''}", flush=True)
        r = subprocess.run([sys.executable, __file__, "--dtype", dtype, "--iters", str(a.iters), "--batch", str(a.batch),
                            "--norm", a.norm, "--graph", str(graph)], env={**os.environ, **env})
        rc |= r.returncode
    sys.exit(rc)


if __name__ == "__main__":
    main()
