"""C-ABI calls of ONE optimize_parameters() by entry point (which calls make the SyncBN path 1060 per step)
    python tools/probes/call_histogram.py [--norm batch] [--dtype bf16] [--size 512 --batch 4] [--dp]"""
import argparse, collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ap = argparse.ArgumentParser()
ap.add_argument("--norm", default="batch"); ap.add_argument("--dtype", default="bf16")
ap.add_argument("--size", type=int, default=512); ap.add_argument("--batch", type=int, default=4); ap.add_argument("--dp", action="store_true")
a = ap.parse_args()
import torch, torch.distributed as dist
if a.dp:
    os.environ.update(MMH_FORCE_DP="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", NCCL_SOCKET_IFNAME="lo")
    os.environ.setdefault("MASTER_PORT", "29741"); os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="env://", device_id=torch.device("cuda", 0))
from bench import synthetic_batch_gpu
from mmhand_amd import lib as L
from mmhand_amd.mmhand_model import MMHandModel
from mmhand_amd.options import default_train_opt
dev = torch.device("cuda:0")
model = MMHandModel(default_train_opt(batchSize=a.batch, norm=a.norm, name="hist", checkpoints_dir="/tmp/mmh_bench",
                                      distributed=bool(a.dp), opt_level="O1" if a.dtype == "bf16" else "O0"))
model.set_input(synthetic_batch_gpu(a.batch, a.size, a.size, 49, dev))
for _ in range(3): model.optimize_parameters()
torch.cuda.synchronize()
hist = collections.Counter(); real = L.call
def counting(name, *args):
    hist[name] += 1
    return real(name, *args)
L.call = counting
model.optimize_parameters()
L.call = real
torch.cuda.synchronize()
print(f"--norm {a.norm} --dtype {a.dtype} {a.size}x{a.size} B={a.batch} dp={a.dp}: {sum(hist.values())} C-ABI calls per step")
for k, v in hist.most_common():
    print(f"  {v:5d}  {k}")
if a.dp: dist.destroy_process_group()
