"""Round-3 PMC target, fp32: the kernels this round added at the training shapes (B=32) - dgrad_s2_kernel (64->128 @256x256,
128->256 @128x128), wgrad_s2_kernel (the same two), conv_stem_f32_kernel (24->64 @256x256), wino_wgrad_dma_kernel (64 planes,
3872 tiles, 512x512 and 256x256) - three dispatches each.  Run under `rocprofv3 --pmc <counters>` (one counter set per pass):
    rocprofv3 --pmc FETCH_SIZE -d out -- python3 tools/pmc_r03_f32.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd import ops, lib
dev = torch.device("cuda:0"); B = 32
L = lib.load()
st = lambda: torch.cuda.current_stream().cuda_stream
for H, Cin, Cout in ((256, 64, 128), (128, 128, 256)):
    w = torch.randn(3, 3, Cin, Cout, device=dev) * 0.05
    dy = torch.randn(B, H // 2, H // 2, Cout, device=dev); x = torch.randn(B, H, H, Cin, device=dev)
    for _ in range(3):
        ops.raw_conv_dgrad(dy, w, (B, H, H, Cin), 2, 1, False)
        ops.raw_conv_wgrad(x, dy, 3, 2, 1, False)
    del w, dy, x
x = torch.randn(B, 256, 256, 24, device=dev); w7 = torch.randn(7, 7, 24, 64, device=dev) * 0.05; b7 = torch.randn(64, device=dev)
for _ in range(3):
    ops.raw_conv_fprop(x, w7, b7, 1, 3, True, 0)
del x
P, T = 64, 3872
for Cin, Cout in ((512, 512), (256, 256)):
    V = torch.randn(P, T, Cin, device=dev); Y = torch.randn(P, T, Cout, device=dev); dU = torch.empty(P, Cin, Cout, device=dev)
    nws = L.mmh_wino_wgrad_gemm_ws_bytes(T, Cin, Cout, P); ws = torch.empty(nws // 4 + 4, device=dev)
    for _ in range(3):
        lib.call("mmh_wino_wgrad_gemm", V.data_ptr(), Y.data_ptr(), T, Cin, Cout, P, lib.F32, ws.data_ptr(), nws, dU.data_ptr(), st())
    del V, Y, dU, ws
torch.cuda.synchronize()
