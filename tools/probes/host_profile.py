"""Where the HOST's time for one optimize_parameters() goes (cProfile, GPU idle at the start of every call): is it the C-ABI
crossings or the Python around them?   python tools/probes/host_profile.py [--norm batch] [--size 512 --batch 4]"""
import argparse, cProfile, os, pstats, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ap = argparse.ArgumentParser()
ap.add_argument("--norm", default="batch"); ap.add_argument("--dtype", default="bf16")
ap.add_argument("--size", type=int, default=512); ap.add_argument("--batch", type=int, default=4)
a = ap.parse_args()
import torch
from bench import synthetic_batch_gpu
from mmhand_amd import lib as L
from mmhand_amd.mmhand_model import MMHandModel
from mmhand_amd.options import default_train_opt
dev = torch.device("cuda:0")
model = MMHandModel(default_train_opt(batchSize=a.batch, norm=a.norm, name="hp", checkpoints_dir="/tmp/mmh_bench",
                                      opt_level="O1" if a.dtype == "bf16" else "O0"))
model.set_input(synthetic_batch_gpu(a.batch, a.size, a.size, 49, dev))
for _ in range(3): model.optimize_parameters()
torch.cuda.synchronize()
import gc; gc.collect(); gc.freeze()
pr = cProfile.Profile()
N = 5
for _ in range(N):
    torch.cuda.synchronize()
    pr.enable(); model.optimize_parameters(); pr.disable()
torch.cuda.synchronize()
s = io.StringIO(); ps = pstats.Stats(pr, stream=s); ps.sort_stats("tottime").print_stats(28)
txt = s.getvalue()
print(f"--norm {a.norm} --dtype {a.dtype} {a.size}x{a.size} B={a.batch}: {N} calls profiled (tottime = time inside the function itself, all {N} calls)")
print(txt[txt.index("ncalls"):])
