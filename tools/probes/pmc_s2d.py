"""a few launches of the stride-2 dgrad (conv_s2d_kernel, B=32 128x128x128 -> 256x256x64, 16-bit) for rocprofv3 --pmc"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mmhand_amd import ops
dev = torch.device("cuda:0")
B, H, Cin, Cout = 32, 256, 64, 128
dy16 = torch.randn(B, H // 2, H // 2, Cout, device=dev).bfloat16(); w = torch.randn(3, 3, Cin, Cout, device=dev) * 0.05
for _ in range(5):
    y = ops.raw_conv_lp16g(ops.conv_desc(B, H, H, Cin, Cout, 3, 2, 1, False), 1, dy16, w, None, 0, True, out16=True)
torch.cuda.synchronize()
