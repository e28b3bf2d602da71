"""a few launches of the stride-2 dgrad (conv_s2d_kernel, B=32 128x128x128 -> 256x256x64, 16-bit), the stride-2 fprop
(conv_s2f_kernel), the stride-2 weight gradient (wgrad_lp16t_kernel<., 2>) and a plain fill of the dgrad's output size, for
rocprofv3 --pmc (tools/probes/pmc_s2d_r05.sh, tools/traffic_r05.sh)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mmhand_amd import ops
dev = torch.device("cuda:0")
B, H, Cin, Cout = 32, 256, 64, 128
dy16 = torch.randn(B, H // 2, H // 2, Cout, device=dev).bfloat16(); w = torch.randn(3, 3, Cin, Cout, device=dev) * 0.05
x16 = torch.randn(B, H, H, Cin, device=dev).bfloat16()
d = ops.conv_desc(B, H, H, Cin, Cout, 3, 2, 1, False)
for _ in range(4):
    y = ops.raw_conv_lp16g(d, 1, dy16, w, None, 0, True, out16=True)
    z = ops.raw_conv_lp16g(d, 0, x16, w, None, 0, True, out16=True)
    dw = ops.raw_wgrad_lp16_flat(ops.conv_desc(B, H, H, Cin, Cout, 3, 2, 1, False), x16, Cin, dy16, True)
    y.fill_(1.0)
torch.cuda.synchronize()
