"""Round-3 PMC target: the 16-bit kernels of the 3x3 512->512 conv at 64x64, B=32 - conv_lp16h2_kernel (fprop, dgrad),
wgrad_lp16t_kernel - a few dispatches each.  Run under `rocprofv3 --pmc <counters>` (one counter set per pass):
    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES ... -d out -- python3 tools/pmc_r03.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd import ops, lib
dev = torch.device("cuda:0")
B, H, W, Cin, Cout = 32, 64, 64, 512, 512
x = torch.randn(B, H, W, Cin, device=dev); w = torch.randn(3, 3, Cin, Cout, device=dev) * 0.05
dy = torch.randn(B, H, W, Cout, device=dev)
xb = ops.lp16_twin(x, True); dyb = ops.lp16_twin(dy, True)
for _ in range(3):
    ops.raw_conv3x3_lp16(xb, w, None, True, 0, True, 0, out16=True)                     # fprop
    ops.raw_conv3x3_lp16(xb, w, None, True, 0, True, 0, out16=True, want_stats=True)    # fprop + statistics epilogue
    ops.raw_conv3x3_lp16(dyb, w, None, False, 0, True, 1, out16=True)                   # zero-pad dgrad
    ops.raw_conv_dgrad(None, w, (B, H, W, Cin), 1, 1, True, bf16=True, dy16=dyb, out16=True)    # reflect dgrad, ring folded
    ops.raw_wgrad3x3_lp16(xb, dyb, True, True)
torch.cuda.synchronize()
