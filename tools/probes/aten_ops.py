"""Which ATen ops (copies, fills, adds - everything that is not a C-ABI call) one bf16 optimize_parameters() runs, by Python call site."""
import os, sys, traceback
from collections import Counter
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from bench import synthetic_batch_gpu
from mmhand_amd.mmhand_model import MMHandModel
from mmhand_amd.options import default_train_opt
size = int(os.environ.get("SIZE", "256")); B = int(os.environ.get("BATCH", "32"))
dev = torch.device("cuda:0")
model = MMHandModel(default_train_opt(batchSize=B, norm=os.environ.get("NORM", "instance"), name="aten", checkpoints_dir="/tmp/mmh_bench", opt_level="O1"))
model.set_input(synthetic_batch_gpu(B, size, size, 49, dev))
for _ in range(3): model.optimize_parameters()
torch.cuda.synchronize()
seen = Counter()
SKIP = ("aten.view", "aten.reshape", "aten.detach", "aten.empty", "aten.as_strided", "aten.slice", "aten.select", "aten.alias",
        "aten._unsafe_view", "aten.unsqueeze", "aten.squeeze", "aten.t.", "aten.expand", "aten.permute", "aten.transpose",
        "aten.is_", "aten.stride", "aten.sym_", "aten.size", "aten.dim", "aten.numel", "aten.storage_offset", "aten._local_scalar",
        "aten.lift_fresh", "aten.new_empty", "aten.narrow", "aten.split", "aten.unbind", "aten.contiguous")
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            fr = [f for f in traceback.extract_stack() if "mmhand_amd" in f.filename or f.filename.endswith("bench.py")]
            site = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno} {fr[-1].name}" if fr else "(autograd engine)"
            shp = next((tuple(a.shape) for a in args if isinstance(a, torch.Tensor)), ())
            seen[(name, site, shp if len(shp) < 2 else "nd")] += 1
        return func(*args, **(kwargs or {}))
with Log():
    model.optimize_parameters()
torch.cuda.synchronize()
for (name, site, shp), c in sorted(seen.items(), key=lambda kv: -kv[1])[:60]:
    print(f"{c:4d}  {name:34s} {str(shp):14s} {site}")
