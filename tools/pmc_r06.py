"""Round-6 PMC target: a few launches each of the 16-bit kernels whose counters DESIGN quotes - the halo
kernel's fprop (conv_lp16h2_kernel), its reflect-fold dgrad, the dgrad with the norm-backward sums in its epilogue
(conv_lp16h2_nbr_kernel), the nine-tap weight gradient (wgrad_lp16t_kernel, stride 1 and stride 2), the general stride-2
fprop 128 -> 256 (conv_lp16g_kernel) and the weights-stationary stride-2 fprop 64 -> 128 (conv_s2f_kernel) - and the fp32
headline's wino_gemm_kernel<128,2>.  Run directly behind `rocprofv3 ... --` (tools/pmc_r06.sh)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                              # noqa: E402
from mmhand_amd import lib, ops                           # noqa: E402

dev = torch.device("cuda:0")
N = 4
B, H = 32, 64
x16 = torch.randn(B, H, H, 512, device=dev).bfloat16()
w = torch.randn(3, 3, 512, 512, device=dev) * 0.05
dy16 = torch.randn(B, H, H, 512, device=dev).bfloat16()
bits = torch.randint(-32768, 32767, (B * H * H * 512 // 8,), device=dev, dtype=torch.int16)
site = ops.NormBwdSite(x16, bits, torch.zeros(B, 512, device=dev), torch.ones(B, 512, device=dev), B, 0.5)
ops.bump_weights_epoch()
for _ in range(N):
    ops.raw_conv3x3_lp16(x16, w, None, True, lib.ACT_NONE, True, 0, out16=True)                       # fprop
for _ in range(N):
    ops.raw_conv3x3_lp16(dy16, w, None, True, lib.ACT_NONE, True, 2, out16=True)                      # reflect-fold dgrad
for _ in range(N):
    ops.raw_conv3x3_lp16(dy16, w, None, True, lib.ACT_NONE, True, 2, out16=True, nbr=site)            # ... + norm sums
for _ in range(N):
    ops.raw_wgrad3x3_lp16(x16, dy16, True, True)                                                       # nine-tap wgrad
for Hs, Cin, Cout in ((256, 64, 128), (128, 128, 256)):
    xs = torch.randn(B, Hs, Hs, Cin, device=dev).bfloat16()
    ws = torch.randn(3, 3, Cin, Cout, device=dev) * 0.05
    dys = torch.randn(B, Hs // 2, Hs // 2, Cout, device=dev).bfloat16()
    for _ in range(N):
        ops.raw_conv_lp16g(ops.conv_desc(B, Hs, Hs, Cin, Cout, 3, 2, 1, False), 0, xs, ws, None, 0, True, out16=True)
    for _ in range(N):
        ops.raw_wgrad_lp16_flat(ops.conv_desc(B, Hs, Hs, Cin, Cout, 3, 2, 1, False), xs, Cin, dys, True)
xf = torch.randn(B, H, H, 512, device=dev)
for _ in range(N):
    ops.raw_conv_fprop_wino(xf, w, None, True, tile=6)                                                 # fp32 headline GEMM
torch.cuda.synchronize()
