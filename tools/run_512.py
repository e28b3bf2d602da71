"""configs[4]'s per-GPU shape as one process for rocprofv3: 512x512, batch 4, bf16 (--opt_level O1), 2 warm-up + 5 steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synthetic_batch_gpu
from mmhand_amd.mmhand_model import MMHandModel
from mmhand_amd.options import default_train_opt
dev = torch.device("cuda:0")
model = MMHandModel(default_train_opt(batchSize=4, norm="instance", name="p512", checkpoints_dir="/tmp/mmh_bench", opt_level="O1"))
model.set_input(synthetic_batch_gpu(4, 512, 512, 49, dev))
for _ in range(2):
    model.optimize_parameters()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    model.optimize_parameters()
torch.cuda.synchronize()
print("512x512 B=4 bf16: %.2f ms per step" % ((time.perf_counter() - t0) / 5 * 1e3))
