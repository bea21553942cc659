"""Every GEMM kernel variant (register-staged / LDS-DMA staged, 64x128 / 128x128 / 256x128 tiles)
against fp32 PyTorch on the same bf16-rounded operands, incl. ragged M/N edges and fused epilogues."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    import devqa_amd  # noqa: F401
    from devqa_amd import lib
    lib.load()
    yield lib
    lib.gemm_set_mode(0)


@pytest.mark.parametrize("mode", [0, 1, 2, 3, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28])
@pytest.mark.parametrize("M,N,K,act,res", [
    (4099, 4224, 1408, 0, False),     # 128x128 (mode 0/1) or 256x128 (mode 2); ragged M
    (16448, 1408, 6144, 0, True),     # ViT fc2 shape, in-place residual
    (2500, 10240, 2560, 1, False),    # OPT fc1 + ReLU
    (1000, 2568, 640, 2, False),      # ragged N (not a multiple of the tile), GELU
    (70, 2560, 2560, 0, True),        # 64x128 variant
])
def test_gemm_variants(L, mode, M, N, K, act, res):
    L.gemm_set_mode(mode)
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, generator=g) * 0.1
    r = torch.randn(M, N, generator=g) if res else None
    ref = a.float() @ w.float().T + b
    if act == 1:
        ref = torch.relu(ref)
    elif act == 2:
        ref = torch.nn.functional.gelu(ref)
    rd = r.cuda() if res else None
    if res:
        ref = ref + r
    out = L.gemm(a.cuda(), w.cuda(), b.cuda(), 1.0, act, rd, out_f32=rd if res else None, want="f32")
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), atol=3e-4, rtol=3e-4)


# Last column tile at most 128 columns wide: the ping-pong kernel's half-width schedule (2 phases per K-tile, three 48-KiB K-tile
# buffers, `pp_mainloop_half`); nk = 1, 2, 3 walk its prologue / tail waits, every epilogue kind its 128 x 32 wave tiles.
@pytest.mark.parametrize("M,N,K,act,res,want", [
    (700, 128, 64, 0, False, "bf16"),       # one half tile, one K-tile
    (513, 384, 128, 1, False, "bf16"),      # 1.5 column tiles, two K-tiles, ragged M
    (256, 1408, 192, 0, True, "f32"),       # 5.5 column tiles, three K-tiles, in-place fp32 residual
    (1000, 96, 1408, 2, False, "bf16"),     # fewer than 128 columns, GELU
    (300, 1404, 256, 0, False, "bf16"),     # N % 8 != 0: fp32 transposition on the half tile
    (300, 1404, 256, 0, False, "f32"),
    (2048, 4224, 1408, 0, False, "bf16"),   # ViT QKV: 16.5 column tiles
    (1030, 1408, 1408, 0, True, "f32"),     # ViT projection
    (600, 200, 320, 3, True, "both"),       # both outputs + residual, quick-GELU, ragged half tile (72 columns)
])
@pytest.mark.parametrize("mode", [22, 28])
def test_gemm_half_width_column_tile(L, M, N, K, act, res, want, mode):
    L.gemm_set_mode(mode)
    g = torch.Generator().manual_seed(M + 3 * N + K)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, generator=g) * 0.1
    r = torch.randn(M, N, generator=g) if res else None
    ref = a.float() @ w.float().T + b
    ref = {0: lambda t: t, 1: torch.relu, 2: torch.nn.functional.gelu, 3: lambda t: t * torch.sigmoid(1.702 * t)}[act](ref)
    if res:
        ref = ref + r
    guard = torch.full((M + 2, N), 7.0, dtype=torch.bfloat16, device="cuda")
    gf = torch.full((M + 2, N), 7.0, device="cuda")
    if res:
        gf[1:M + 1] = r.cuda()
    ob = guard[1:M + 1] if want in ("bf16", "both") else None
    of = gf[1:M + 1] if (want in ("f32", "both") or res) else None
    L.gemm(a.cuda(), w.cuda(), b.cuda(), 1.0, act, of if res else None, out_bf16=ob, out_f32=of)
    torch.cuda.synchronize()
    L.gemm_set_mode(0)
    assert float(guard[0].float().min()) == 7.0 and float(guard[M + 1].float().max()) == 7.0
    assert float(gf[0].min()) == 7.0 and float(gf[M + 1].max()) == 7.0
    if of is not None:
        np.testing.assert_allclose(of.cpu().numpy(), ref.numpy(), atol=3e-4, rtol=3e-4)
    if ob is not None:
        np.testing.assert_allclose(ob.float().cpu().numpy(), ref.numpy(), atol=1.2e-2, rtol=8e-3)


@pytest.mark.parametrize("M,N,K,act,res,want", [
    (48, 10240, 2560, 1, False, "bf16"),    # OPT fc1 at B = 1: auto split-K (80 column tiles -> 6 K-slices) + fused epilogue
    (33, 2560, 10240, 0, True, "f32"),      # OPT fc2 with the in-place fp32 residual
    (64, 7680, 2560, 0, False, "bf16"),     # fused qkv
    (17, 4096, 11008, 0, True, "f32"),      # LLaMA down_proj
    (48, 50272, 2560, 0, False, "f32"),     # lm_head rows: enough column tiles, no split
    (257, 1408, 6144, 0, True, "f32"),      # single-image ViT fc2: 3 x 11 tiles of 128x128 -> 15 K-slices, in-place fp32 residual
    (257, 4224, 1408, 0, False, "bf16"),    # single-image ViT QKV: 99 tiles -> 5 K-slices
    (300, 1408, 1408, 0, True, "f32"),      # projection, ragged M
    (512, 2560, 2560, 1, False, "bf16"),    # upper end of the split range (4 x 20 tiles)
    (180, 2560, 1088, 0, False, "f32"),     # compacted-column fc2 rows of the FT loop
])
def test_small_m_splitk_epilogue(L, M, N, K, act, res, want):
    L.gemm_set_mode(0)
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, generator=g) * 0.1
    r = torch.randn(M, N, generator=g) if res else None
    ref = a.float() @ w.float().T + b
    if act == 1:
        ref = torch.relu(ref)
    if res:
        ref = ref + r
    rd = r.cuda() if res else None
    out = L.gemm(a.cuda(), w.cuda(), b.cuda(), 1.0, act, rd, out_f32=rd if res else None, want=want)
    torch.cuda.synchronize()
    tol = 3e-4 if want == "f32" else 1.5e-2
    np.testing.assert_allclose(out.float().cpu().numpy(), ref.numpy(), atol=tol, rtol=tol)


@pytest.mark.parametrize("M,N,K,act,bias,alpha", [
    (4099, 4224, 1408, 0, True, 1.0),      # ragged M, ViT QKV
    (1000, 2568, 640, 2, True, 1.0),       # ragged N (2568 = 10 tiles + 8 columns), GELU
    (2500, 10240, 2560, 1, True, 1.0),     # OPT fc1 + ReLU
    (777, 264, 128, 3, False, 0.5),        # one ragged tile column, no bias, alpha, quick-GELU
    (300, 4100, 256, 0, True, 1.0),        # N % 8 != 0: falls back to the fp32 transposition
])
def test_gemm_bf16_output_epilogues(L, M, N, K, act, bias, alpha):
    """bf16-only outputs take pp_epilogue_bf16 (packed-bf16 LDS transposition, 16-byte stores); mode 26 forces the fp32
    transposition: both must give the same bits, equal to the fp32 reference rounded to bf16 (up to one bf16 ulp where the
    fp32 accumulation order matters)."""
    g = torch.Generator().manual_seed(M * 7 + N)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).cuda()
    b = (torch.randn(N, generator=g) * 0.1).cuda() if bias else None
    ref = a.float() @ w.float().T
    if bias:
        ref = ref + b
    ref = ref * alpha
    ref = {0: lambda t: t, 1: torch.relu, 2: torch.nn.functional.gelu, 3: lambda t: t * torch.sigmoid(1.702 * t)}[act](ref)
    outs = {}
    for mode in (22, 24, 26, 28):
        L.gemm_set_mode(mode)
        guard = torch.full((M + 2, N), 7.0, dtype=torch.bfloat16, device="cuda")      # rows before / after must stay untouched
        o = guard[1:M + 1]
        L.gemm(a, w, b, alpha, act, None, out_bf16=o)
        torch.cuda.synchronize()
        assert float(guard[0].float().min()) == 7.0 and float(guard[M + 1].float().max()) == 7.0
        outs[mode] = o.clone()
    L.gemm_set_mode(0)
    # 22 = production (packed bf16 epilogue), 26 = fp32 transposition epilogue: bit-identical.  24 = the opt-in packed-polynomial GELU
    # (|Phi error| <= 1.3e-5, csrc/gemm_bf16_pp.hip): equal up to one bf16 rounding step of the stored value, on < 2 % of the values
    assert torch.equal(outs[22], outs[26])
    assert torch.equal(outs[22], outs[28])      # 28 = the persistent form (a workgroup per CU, the successor's K-tile 0 staged under the epilogue)
    if act == 2:
        d = (outs[24].float() - outs[26].float()).abs()
        assert float((d - 2.0 ** -7 * outs[26].float().abs()).max()) <= 6e-5, float(d.max())
        assert float((d > 0).float().mean()) < 0.02
    else:
        assert torch.equal(outs[24], outs[26])
    np.testing.assert_allclose(outs[22].float().cpu().numpy(), ref.cpu().numpy(), atol=6e-3, rtol=8e-3)


@pytest.mark.parametrize("M,N,K,act,res", [
    (48, 1920, 12800, 0, False),     # MEND hyper-network v-GEMM: 30 tiles -> 17 K-slices
    (64, 2560, 10240, 0, True),      # OPT fc2 rows, in-place residual, split
    (100, 200, 4096, 1, False),      # two row tiles, ragged N, ReLU, split
    (48, 12800, 1920, 0, False),     # 200 tiles: no split
    (7, 40, 640, 2, True),           # tiny: no split (K < 2048)
])
def test_gemm_f32_exact_and_splitk(L, M, N, K, act, res):
    """The exact-fp32 GEMM (fp32 operands through lib.gemm), with and without its automatic split-K, against float64."""
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g) * 0.1
    r = torch.randn(M, N, generator=g) if res else None
    ref = a.double() @ w.double().T + b.double()
    if act == 1:
        ref = torch.relu(ref)
    elif act == 2:
        ref = torch.nn.functional.gelu(ref)
    if res:
        ref = ref + r.double()
    rd = r.cuda() if res else None
    out = L.gemm(a.cuda(), w.cuda(), b.cuda(), 1.0, act, rd, out_f32=rd if res else None, want="f32")
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), ref.float().numpy(), atol=2e-5, rtol=2e-5)
