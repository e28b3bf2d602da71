"""--graph_step against the eager step, same process, same box: images/s, host enqueue per iteration, C-ABI calls.
    python tools/probes/graph_step_bench.py [bf16|f32|512]..."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch                                              # noqa: E402
import bench                                              # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
which = sys.argv[1:] or ["bf16", "512", "f32"]
cfg = {"bf16": dict(B=32, size=256, opt_level="O1"), "512": dict(B=4, size=512, opt_level="O1"),
       "f32": dict(B=32, size=256, opt_level="O0")}
for w in which:
    c = cfg[w]
    for graph in (False, True, False, True):
        r = bench.side_train_run(dev, c["B"], c["size"], 8, warmup=6, opt_level=c["opt_level"], graph_step=graph)
        print(w, "graph" if graph else "eager", json.dumps({k: r.get(k) for k in (
            "images_per_s", "ms_per_step", "host_enqueue_ms", "c_abi_calls_per_step", "host_enqueue_over_step", "graph_step",
            "graph_replays", "graph_error", "losses_finite")}), "peak GiB %.1f" % (torch.cuda.max_memory_allocated() / 2 ** 30), flush=True)
        torch.cuda.reset_peak_memory_stats()
