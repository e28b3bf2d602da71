"""mmh_wino_gemm against torch.bmm (fp64) for a few shapes; prints where the error sits."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd import ops, lib
L = lib
dev = torch.device("cuda:0")
shapes = [(36, 32, 64, 128), (36, 128, 64, 128), (36, 256, 64, 128), (16, 200, 96, 64), (36, 8192, 512, 512),
          (36, 1000, 256, 160), (4, 8, 32, 64)]
for v2 in (0, 1):
    lib.check(lib.load().mmh_set_option(b"wino_gemm_v2", v2), "opt")
    for (P, M, K, N) in shapes:
        V = torch.randn(P, M, K, device=dev); U = torch.randn(P, K, N, device=dev)
        out = torch.full((P, M, N), 7.0, device=dev)
        lib.call("mmh_wino_gemm", V.data_ptr(), U.data_ptr(), out.data_ptr(), M, K, N, P, lib.F32, torch.cuda.current_stream().cuda_stream)
        ref = torch.bmm(V.double(), U.double())
        err = (out.double() - ref).abs()
        rel = err.sum() / ref.abs().sum()
        msg = f"v2={v2} P={P} M={M} K={K} N={N}: rel L1 {rel:.2e}"
        if rel > 1e-5:
            bad = (err > 1e-2)
            msg += f" bad planes {bad.any(2).any(1).nonzero().flatten().tolist()[:8]} rows {bad.any(2).any(0).nonzero().flatten().tolist()[:12]} cols {bad.any(1).any(0).nonzero().flatten().tolist()[:12]}"
        print(msg, flush=True)
