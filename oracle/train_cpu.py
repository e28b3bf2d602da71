"""BASELINE.json configs[0]: "RHD 256x256, batch=2, fp32, reference train.py on CPU for 10 iters".

The reference's own train.py cannot run on a CPU (apex + CUDA asserts); this is
the oracle's counterpart: StepOracle (pinned against the reference leaf modules) driven for N
iterations on RHD-shaped synthetic batches, with the loss_log line format of
util/visualizer.py:116-123.  TEST INFRASTRUCTURE / CPU baseline only.

    python -m oracle.train_cpu [--iters 10] [--batch 2] [--size 256] [--norm batch] [--threads 16]
"""
import argparse
import os
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--norm", default="batch", choices=["batch", "instance"])
    ap.add_argument("--threads", type=int, default=min(16, os.cpu_count() or 1))
    ap.add_argument("--ngf", type=int, default=64)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    from mmhand_amd.networks import Discriminator, Generator, VGGHead   # parameter containers only
    from oracle import mmhand_ref as O
    g = Generator([3, 42, 6], 3, a.ngf, a.norm, True, 9).init_weights("normal", 49)
    dpb = Discriminator(24, a.ngf, a.norm, True, 3).init_weights("normal", 50)
    dpp = Discriminator(6, a.ngf, a.norm, True, 3).init_weights("normal", 51)
    orc = O.StepOracle(g.state_dict(), dpb.state_dict(), dpp.state_dict(), VGGHead().init_random().state_dict(),
                       a.norm, True, True, 9, 3, rng=random.Random(49))
    print(f"oracle CPU training: {a.iters} iters, B={a.batch}, {a.size}x{a.size}, --norm {a.norm}, "
          f"{a.threads} threads of {os.cpu_count()} logical CPUs")
    times = []
    for it in range(a.iters):
        batch = O.synthetic_batch(a.batch, a.size, a.size, seed=49 + it)
        t0 = time.time()
        errs = orc.step(batch)
        dt = time.time() - t0
        times.append(dt)
        print("(epoch: 1, iters: %d, time: %.3f) " % ((it + 1) * a.batch, dt / a.batch) +
              "".join("%s: %.3f " % kv for kv in errs.items()), flush=True)
    steady = times[2:] or times
    print("mean step %.2f s (after 2 warm-up iters) = %.4f images/s" %
          (sum(steady) / len(steady), a.batch * len(steady) / sum(steady)))


if __name__ == "__main__":
    main()
