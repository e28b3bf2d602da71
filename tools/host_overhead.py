"""Host-side enqueue time of one optimize_parameters() vs its GPU time (is the step launch-bound?)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synthetic_batch_gpu
from mmhand_amd.mmhand_model import MMHandModel
from mmhand_amd.options import default_train_opt
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
model = MMHandModel(default_train_opt(batchSize=B, norm="instance", name="bench", checkpoints_dir="/tmp/mmh_bench"))
model.set_input(synthetic_batch_gpu(B, 256, 256, 49, dev))
for _ in range(3):
    model.optimize_parameters()
torch.cuda.synchronize()
host, total = [], []
for _ in range(5):
    t0 = time.perf_counter()
    model.optimize_parameters()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append(t1 - t0); total.append(t2 - t0)
print(f"B={B}: host enqueue {1e3 * sum(host) / 5:.1f} ms/step, step (host+GPU drain) {1e3 * sum(total) / 5:.1f} ms")
