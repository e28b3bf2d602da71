"""Stage-level check of mmh_wino_gemm (the persistent Winograd-domain GEMM kernel and the generic
batched kernel behind the same entry point) against an fp64 matrix product of the same operands.
Tolerance: 2e-6 relative L1 (fp32 accumulation over K <= 512)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 2e-6

# (planes, rows, K, N): row tails, column-group tails, one- and many-item work lists, K = 1 k-step
SHAPES = [(36, 32, 64, 128), (36, 128, 64, 128), (36, 256, 64, 128), (16, 200, 96, 64),
          (36, 1000, 256, 160), (4, 8, 32, 64), (36, 2048, 512, 512), (16, 4100, 128, 96)]


@pytest.mark.parametrize("v2", [0, 1])
@pytest.mark.parametrize("shape", SHAPES)
def test_wino_gemm_matches_fp64_product(shape, v2):
    from mmhand_amd import lib
    P, M, K, N = shape
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(P * 1000 + M)
    V = torch.randn(P, M, K, generator=g).to(dev)
    U = torch.randn(P, K, N, generator=g).to(dev)
    out = torch.full((P, M, N), 7.0, device=dev)
    lib.check(lib.load().mmh_set_option(b"wino_gemm_v2", v2), "mmh_set_option")
    try:
        lib.call("mmh_wino_gemm", V.data_ptr(), U.data_ptr(), out.data_ptr(), M, K, N, P, lib.F32,
                 torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    finally:
        lib.check(lib.load().mmh_set_option(b"wino_gemm_v2", 1), "mmh_set_option")
    ref = torch.bmm(V.double().cpu(), U.double().cpu())
    rel = (out.double().cpu() - ref).abs().sum() / ref.abs().sum()
    assert rel < TOL, rel


# (planes, tiles, Cin, Cout): split tails (tiles not a multiple of 32), a single k-step, many splits
WGRAD_SHAPES = [(36, 32, 128, 128), (36, 200, 128, 256), (16, 1000, 256, 128), (36, 4096, 256, 256),
                (36, 37, 128, 128), (4, 8192, 512, 512), (36, 300, 64, 96)]


@pytest.mark.parametrize("v2", [0, 1, 2])
@pytest.mark.parametrize("shape", WGRAD_SHAPES + [(64, 3872, 256, 256), (64, 105, 128, 512), (16, 64, 256, 256)])
def test_wino_wgrad_gemm_matches_fp64_product(shape, v2):
    """dU[xi] = V[xi]^T . Yh[xi] (contraction over tiles, split-K + fixed-order slab reduction):
    the persistent TN kernel (v2 = 1, channel counts % 128 == 0), the generic wgrad kernel (0) and the LDS-DMA ring kernel
    (2: wino_wgrad_dma.hip, Cin % 128 == 0, Cout % 256 == 0, >= 32 tiles; other shapes fall through to v2 = 1): whole and
    ragged last stages, with and without a split tile range."""
    from mmhand_amd import lib
    P, T, Cin, Cout = shape
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(P * 1000 + T)
    V = torch.randn(P, T, Cin, generator=g).to(dev)
    Y = torch.randn(P, T, Cout, generator=g).to(dev)
    L = lib.load()
    lib.check(L.mmh_set_option(b"wino_wgrad_v2", min(v2, 1)), "mmh_set_option")
    lib.check(L.mmh_set_option(b"wino_wgrad_dma", int(v2 == 2)), "mmh_set_option")
    try:
        nws = L.mmh_wino_wgrad_gemm_ws_bytes(T, Cin, Cout, P)
        ws = torch.full((nws // 4 + 4,), float("nan"), device=dev)
        dU = torch.full((P, Cin, Cout), 7.0, device=dev)
        lib.call("mmh_wino_wgrad_gemm", V.data_ptr(), Y.data_ptr(), T, Cin, Cout, P, lib.F32, ws.data_ptr(), nws,
                 dU.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    finally:
        lib.check(L.mmh_set_option(b"wino_wgrad_v2", 1), "mmh_set_option")
    ref = torch.bmm(V.double().cpu().transpose(1, 2), Y.double().cpu())
    rel = (dU.double().cpu() - ref).abs().sum() / ref.abs().sum()
    assert rel < 5e-6, rel       # fp32 accumulation over up to 8192 tiles


BF16_GEMM_SHAPES = [(16, 32, 128, 128), (16, 200, 128, 256), (16, 1000, 256, 160), (4, 8, 64, 64),
                    (16, 4096, 512, 512), (16, 131, 192, 96)]


@pytest.mark.parametrize("shape", BF16_GEMM_SHAPES)
def test_wino_gemm_bf16(shape):
    """bf16 NT GEMM M[xi] = V[xi] . U[xi]^T (U stored [N][K]): fp32 accumulation of bf16 operands,
    result rounded once to bf16 -> within 2^-8 of the fp64 product of the same bf16 operands."""
    from mmhand_amd import lib
    P, M, K, N = shape
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(P * 1000 + M)
    V = torch.randn(P, M, K, generator=g).bfloat16().to(dev)
    U = torch.randn(P, N, K, generator=g).bfloat16().to(dev)
    out = torch.full((P, M, N), 7.0, dtype=torch.bfloat16, device=dev)
    lib.call("mmh_wino_gemm", V.data_ptr(), U.data_ptr(), out.data_ptr(), M, K, N, P, lib.BF16,
             torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ref = torch.bmm(V.double().cpu(), U.double().cpu().transpose(1, 2))
    err = (out.double().cpu() - ref).abs()
    assert (err <= 2.0 ** -8 * ref.abs() + 1e-3).all(), float((err / (ref.abs() + 1e-3)).max())
    assert err.sum() / ref.abs().sum() < 2e-3


BF16_WGRAD_SHAPES = [(16, 64, 128, 128), (16, 200, 128, 256), (16, 1000, 256, 128), (16, 4096, 256, 256),
                     (16, 37, 128, 128), (4, 8192, 512, 512)]


@pytest.mark.parametrize("shape", BF16_WGRAD_SHAPES)
def test_wino_wgrad_gemm_bf16(shape):
    """bf16 TN GEMM dU[xi] = V[xi]^T . Yh[xi] (transposed LDS reads), fp32 accumulation and output."""
    from mmhand_amd import lib
    P, T, Cin, Cout = shape
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(P * 1000 + T)
    V = torch.randn(P, T, Cin, generator=g).bfloat16().to(dev)
    Y = torch.randn(P, T, Cout, generator=g).bfloat16().to(dev)
    L = lib.load()
    nws = L.mmh_wino_wgrad_gemm_ws_bytes(T, Cin, Cout, P)
    ws = torch.full((nws // 4 + 4,), float("nan"), device=dev)
    dU = torch.full((P, Cin, Cout), 7.0, device=dev)
    lib.call("mmh_wino_wgrad_gemm", V.data_ptr(), Y.data_ptr(), T, Cin, Cout, P, lib.BF16, ws.data_ptr(), nws,
             dU.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ref = torch.bmm(V.double().cpu().transpose(1, 2), Y.double().cpu())
    rel = (dU.double().cpu() - ref).abs().sum() / ref.abs().sum()
    assert rel < 5e-6, rel
