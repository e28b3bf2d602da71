"""Round-2 PMC target: the dominant kernels at their bench shapes, a few dispatches each.
   fp32: Winograd F(6x6,3x3) fprop of 3x3 reflect 512->512 @64x64, B=32 (wino_gemm_kernel<128,2> + transforms),
         reflect-fold dgrad (input_dy / gemm / output(fold));
   bf16: conv_lp16s_kernel fprop, conv_lp16s dgrad, wgrad_lp16_kernel of the same conv.
Run under `rocprofv3 --pmc <counters>` (one counter set per pass)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd import ops, lib
dev = torch.device("cuda:0")
B, H, W, Cin, Cout = 32, 64, 64, 512, 512
x = torch.randn(B, H, W, Cin, device=dev); w = torch.randn(3, 3, Cin, Cout, device=dev) * 0.05
dy = torch.randn(B, H, W, Cout, device=dev)
which = os.environ.get("PMC_WHICH", "both")
for _ in range(3):
    if which in ("both", "f32"):
        y, V = ops.raw_conv_fprop_wino(x, w, None, True, 0, 6, keep_V=True)
        ops.raw_conv_bwd_wino6(dy, w, x.shape, True, V)
    if which in ("both", "bf16"):
        xb = ops.lp16_twin(x, True); dyb = ops.lp16_twin(dy, True)
        ops.raw_conv3x3_lp16(xb, w, None, True, 0, True, 0)
        ops.raw_conv3x3_lp16(dyb, w, None, False, 0, True, 1)
        ops.raw_wgrad3x3_lp16(xb, dyb, True, True)
torch.cuda.synchronize()
