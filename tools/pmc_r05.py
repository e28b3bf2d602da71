"""Round-5 PMC target: the 512 -> 512 @64x64 B=32 16-bit fprop on the halo kernel's two forms - conv_lp16h2_kernel (lp16_shape 19)
and, in an A/B build (MMH_LIB_PATH=.../libmmhand_hip_ab.so), conv_lp16q_kernel (20) - and the stride-2 dgrad conv_s2d_kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd import ops, lib
L = lib.load(); dev = torch.device("cuda:0")
B, H, W, Cin, Cout = 32, 64, 64, 512, 512
x = torch.randn(B, H, W, Cin, device=dev); w = torch.randn(3, 3, Cin, Cout, device=dev) * 0.05
xb = ops.lp16_twin(x, True)
dy16 = torch.randn(B, 128, 128, 128, device=dev).bfloat16(); w2 = torch.randn(3, 3, 64, 128, device=dev) * 0.05
d2 = ops.conv_desc(B, 256, 256, 64, 128, 3, 2, 1, False)
for shape in (19, 20):
    lib.check(L.mmh_set_option(b"lp16_shape", shape), "set")
    try:
        for _ in range(4):
            ops.raw_conv3x3_lp16(xb, w, None, True, 0, True, 0, out16=True)
    except RuntimeError as e:
        print("lp16_shape", shape, "not in this build:", str(e)[:80], file=sys.stderr)
lib.check(L.mmh_set_option(b"lp16_shape", 19), "set")
for _ in range(4):
    ops.raw_conv_lp16g(d2, 1, dy16, w2, None, 0, True, out16=True)
torch.cuda.synchronize()
