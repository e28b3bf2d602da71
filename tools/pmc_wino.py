"""Launch the Winograd F(6x6,3x3) (TILE=4: F(4x4,3x3)) passes of the dominant conv (3x3 reflect 512->512 @64x64, B=32) a
few times; run under `rocprofv3 --pmc ...` to collect counters per dispatch (tools/traffic_wino.sh)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd import ops, lib
for kv in os.environ.get('MMH_OPTS', '').split(','):
    if kv:
        k, v = kv.split('=')
        lib.check(lib.load().mmh_set_option(k.encode(), int(v)), 'opt')
dev = torch.device("cuda:0")
TILE = int(os.environ.get("TILE", "6"))
B, H, W, Cin, Cout = 32, 64, 64, 512, 512
x = torch.randn(B, H, W, Cin, device=dev); w = torch.randn(3, 3, Cin, Cout, device=dev) * 0.05
y = ops.raw_conv_fprop_wino(x, w, None, True, 0, TILE); dy = torch.randn_like(y)
for _ in range(3):
    ops.raw_conv_fprop_wino(x, w, None, True, 0, TILE)
    ops.raw_conv_dgrad_wino(dy, w, x.shape, True, TILE)
    ops.raw_conv_wgrad_wino(x, dy, True, TILE)
torch.cuda.synchronize()
