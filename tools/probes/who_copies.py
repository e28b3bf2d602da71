"""Which Python call sites issue device-to-device copies / fills in one 16-bit optimize_parameters()?"""
import os, sys, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import synthetic_batch_gpu
from mmhand_amd.mmhand_model import MMHandModel
from mmhand_amd.options import default_train_opt
dev = torch.device("cuda:0")
model = MMHandModel(default_train_opt(batchSize=8, norm="instance", name="who", checkpoints_dir="/tmp/mmh_bench", opt_level="O1"))
model.set_input(synthetic_batch_gpu(8, 256, 256, 49, dev))
for _ in range(2): model.optimize_parameters()
torch.cuda.synchronize()
from torch.utils._python_dispatch import TorchDispatchMode
cnt = collections.Counter()
class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(k in name for k in ("copy_", "clone", "fill_", "zero_", "zeros", "cat", "contiguous", "_to_copy", "add", "mul", "sum")):
            st = [f for f in traceback.extract_stack() if "/mmhand_amd/" in f.filename or "bench.py" in f.filename]
            where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in st[-3:][::-1])
            cnt[(name, where)] += 1
        return func(*args, **(kwargs or {}))
with Spy():
    model.optimize_parameters()
torch.cuda.synchronize()
for (name, where), n in cnt.most_common(60):
    print(f"{n:4d}  {name:32s} {where}")
