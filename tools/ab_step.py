"""Interleaved A/B of ONE switch over whole optimize_parameters() steps in one process (one box, one clock state): the model is
built once, the switch flips every few steps.  Box-to-box variance of this pool is +-3 %, so a 1 % change only shows here.

    python tools/ab_step.py opt:lp16_persist 0 1            # an mmh_set_option key
    python tools/ab_step.py ops:USE_NORM_TWIN 0 1 [--dtype bf16] [--size 256] [--batch 32] [--rounds 6] [--steps 4]
    python tools/ab_step.py model:MERGE_D_PASSES 0 1        # a module attribute of mmhand_amd.ops / mmhand_amd.mmhand_model
"""
import argparse, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ap = argparse.ArgumentParser()
ap.add_argument("switch"); ap.add_argument("values", nargs="+")
ap.add_argument("--dtype", default="bf16"); ap.add_argument("--size", type=int, default=256); ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--norm", default="instance"); ap.add_argument("--rounds", type=int, default=6); ap.add_argument("--steps", type=int, default=4)
a = ap.parse_args()
from bench import synthetic_batch_gpu
from mmhand_amd import lib, ops, mmhand_model
from mmhand_amd.mmhand_model import MMHandModel
from mmhand_amd.options import default_train_opt
kind, key = a.switch.split(":")
def setv(v):
    if kind == "opt": lib.check(lib.load().mmh_set_option(key.encode(), int(v)), "mmh_set_option")
    else:
        mod = ops if kind == "ops" else mmhand_model
        cur = getattr(mod, key)
        setattr(mod, key, type(cur)(int(v)) if isinstance(cur, (bool, int)) else v)
        ops.bump_weights_epoch()
dev = torch.device("cuda:0")
model = MMHandModel(default_train_opt(batchSize=a.batch, norm=a.norm, name="ab", checkpoints_dir="/tmp/mmh_bench",
                                      opt_level="O1" if a.dtype == "bf16" else "O0"))
model.set_input(synthetic_batch_gpu(a.batch, a.size, a.size, 49, dev))
res = {v: [] for v in a.values}
for v in a.values:
    setv(v)
    for _ in range(2): model.optimize_parameters()
torch.cuda.synchronize()
for r in range(a.rounds):
    for v in a.values:
        setv(v); model.optimize_parameters(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps): model.optimize_parameters()
        torch.cuda.synchronize()
        res[v].append((time.perf_counter() - t0) / a.steps * 1e3)
print(f"{a.switch} ({a.dtype}, {a.size}x{a.size}, B={a.batch}, --norm {a.norm}): " +
      " | ".join(f"{v}: {statistics.median(res[v]):.2f} ms/step (min {min(res[v]):.2f})" for v in a.values), flush=True)
