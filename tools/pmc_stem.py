"""Launch the fp32 7x7 stem fprop (44->64, reflect pad 3, B=32 @256x256) a few times; run under
`rocprofv3 --pmc ...` (tools/traffic_stem.sh).  MMH_OPTS=key=value,... sets library options."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmhand_amd import lib, ops      # noqa: E402

for kv in os.environ.get("MMH_OPTS", "").split(","):
    if kv:
        k, v = kv.split("=")
        lib.check(lib.load().mmh_set_option(k.encode(), int(v)), "opt")
dev = torch.device("cuda:0")
x = torch.randn(32, 256, 256, 44, device=dev)
w = torch.randn(7, 7, 44, 64, device=dev) * 0.05
for _ in range(4):
    ops.raw_conv_fprop(x, w, None, 1, 3, True, 0)
torch.cuda.synchronize()
