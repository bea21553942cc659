"""Op-level parity of every C-ABI entry point against plain fp32 PyTorch (CPU) on the same
(bf16-rounded) inputs.  Integer outputs (argmax, top-k ids) must match exactly; floating point
tolerances are written per test."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    import devqa_amd  # noqa: F401
    from devqa_amd import lib
    lib.load()
    return lib


def dev(t):
    return t.to("cuda")


def bf(t):  # round to bf16 and back (fp32 CPU reference sees the same operand values)
    return t.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize("M,N,K,act,res,alpha", [
    (257, 4224, 1408, 0, False, 1.0),
    (1028, 6144, 1408, 2, False, 1.0),
    (48, 2560, 2560, 0, True, 1.0),
    (48, 2560, 2560, 0, False, 80 ** -0.5),
    (3, 50272, 2560, 0, False, 1.0),
    (130, 136, 40, 1, True, 1.0),
    (33, 48, 608, 0, False, 1.0),
    (64, 10240, 2560, 1, False, 1.0),
    (1000, 2560, 10240, 0, True, 1.0),
])
def test_gemm(L, M, N, K, act, res, alpha):
    g = torch.Generator().manual_seed(M * 7 + N)
    a = bf(torch.randn(M, K, generator=g))
    w = bf(torch.randn(N, K, generator=g) / K ** 0.5)
    b = torch.randn(N, generator=g) * 0.1
    r = torch.randn(M, N, generator=g) if res else None
    ref = (a @ w.T + b) * alpha
    if act == 1:
        ref = torch.relu(ref)
    elif act == 2:
        ref = torch.nn.functional.gelu(ref)
    if res:
        ref = ref + r
    ob, of = L.gemm(dev(a).to(torch.bfloat16), dev(w).to(torch.bfloat16), dev(b), alpha, act,
                    dev(r) if res else None, want="both")
    torch.cuda.synchronize()
    np.testing.assert_allclose(of.cpu().numpy(), ref.numpy(), atol=2e-4, rtol=2e-4)
    np.testing.assert_allclose(ob.float().cpu().numpy(), ref.numpy(), atol=2e-2, rtol=1e-2)


def test_gemm_strided_a_and_inplace_residual(L):
    g = torch.Generator().manual_seed(5)
    big = bf(torch.randn(100, 3 * 64, generator=g))
    w = bf(torch.randn(96, 64, generator=g))
    a_view = dev(big).to(torch.bfloat16)[:, 64:128]  # row stride 192, unit inner stride
    resid = dev(torch.randn(100, 96, generator=g))
    ref = big[:, 64:128] @ w.T + resid.cpu()
    L.gemm(a_view, dev(w).to(torch.bfloat16), residual=resid, out_f32=resid)
    torch.cuda.synchronize()
    np.testing.assert_allclose(resid.cpu().numpy(), ref.numpy(), atol=1e-4, rtol=1e-4)


def test_gemm_rejects_bad_shapes(L):
    a = torch.zeros(4, 12, dtype=torch.bfloat16, device="cuda")
    w = torch.zeros(8, 12, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(L.DevqaError):
        L.gemm(a, w)  # K % 8 != 0


@pytest.mark.parametrize("M,D,eps", [(257, 1408, 1e-6), (32, 768, 1e-12), (48, 2560, 1e-5), (5, 40, 1e-5)])
def test_layernorm_fwd_bwd(L, M, D, eps):
    g = torch.Generator().manual_seed(D)
    x = torch.randn(M, D, generator=g) * 3 + 1
    add = torch.randn(M, D, generator=g)
    gamma = 1 + 0.1 * torch.randn(D, generator=g)
    beta = 0.1 * torch.randn(D, generator=g)
    ref = torch.nn.functional.layer_norm(x + add, (D,), gamma, beta, eps)
    ob, of = L.layernorm(dev(x), dev(gamma), dev(beta), eps, add=dev(add), want="both")
    np.testing.assert_allclose(of.cpu().numpy(), ref.numpy(), atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(ob.float().cpu().numpy(), ref.numpy(), atol=3e-2, rtol=1e-2)
    xr = x.clone().requires_grad_(True)
    dy = torch.randn(M, D, generator=g)
    torch.nn.functional.layer_norm(xr, (D,), gamma, beta, eps).backward(dy)
    dx = L.layernorm_bwd_dx(dev(x), dev(gamma), dev(dy), eps)
    np.testing.assert_allclose(dx.cpu().numpy(), xr.grad.numpy(), atol=2e-5, rtol=1e-4)


def _ref_attention(q, k, v, desc, H, dh, scale, causal):
    out = torch.zeros(q.shape[0], H * dh)
    for (qs, ql, kps, kpl, kos, kol) in desc:
        for h in range(H):
            qq = q[qs:qs + ql, h * dh:(h + 1) * dh]
            kk = torch.cat([k[kps:kps + kpl, h * dh:(h + 1) * dh], k[kos:kos + kol, h * dh:(h + 1) * dh]])
            vv = torch.cat([v[kps:kps + kpl, h * dh:(h + 1) * dh], v[kos:kos + kol, h * dh:(h + 1) * dh]])
            s = (qq @ kk.T) * scale
            if causal:
                qi = torch.arange(ql)[:, None] + (kol - ql)
                kj = torch.arange(kol)[None, :]
                mask = torch.cat([torch.ones(ql, kpl, dtype=torch.bool), kj <= qi], 1)
                s = s.masked_fill(~mask, float("-inf"))
            out[qs:qs + ql, h * dh:(h + 1) * dh] = torch.softmax(s, -1) @ vv
    return out


@pytest.mark.parametrize("name", ["vit", "qformer_cross", "opt_causal", "opt_prefix", "opt_pack_short", "llama_pack_short", "tiny_heads",
                                  "clip_l_577", "llama_dh128", "llama_prefix_576", "opt_pack_32", "llama_pack_32", "dh64_pack_32"])
def test_attention(L, name):
    g = torch.Generator().manual_seed(11)
    keep = slice(None)
    if name == "vit":  # 2 images x 257 tokens, fused qkv buffer
        H, dh, n = 16, 88, 257
        qkv = bf(torch.randn(2 * n, 3 * H * dh, generator=g))
        q, k, v = qkv[:, :H * dh], qkv[:, H * dh:2 * H * dh], qkv[:, 2 * H * dh:]
        desc = [(0, n, 0, n, 0, 0), (n, n, n, n, 0, 0)]
        causal, scale = 0, dh ** -0.5
        dq = dev(qkv).to(torch.bfloat16)
        views = (dq[:, :H * dh], dq[:, H * dh:2 * H * dh], dq[:, 2 * H * dh:])
    else:
        if name == "qformer_cross":
            H, dh = 12, 64
            q = bf(torch.randn(64, H * dh, generator=g))
            k = bf(torch.randn(514, H * dh, generator=g))
            v = bf(torch.randn(514, H * dh, generator=g))
            desc = [(0, 32, 0, 257, 0, 0), (32, 32, 257, 257, 0, 0)]
            causal, scale = 0, dh ** -0.5
        elif name == "opt_causal":
            H, dh = 32, 80
            T = [48, 17, 70]
            tot = sum(T)
            q = bf(torch.randn(tot, H * dh, generator=g))
            k = bf(torch.randn(tot, H * dh, generator=g))
            v = bf(torch.randn(tot, H * dh, generator=g))
            st = np.cumsum([0] + T)
            desc = [(int(st[i]), T[i], 0, 0, int(st[i]), T[i]) for i in range(3)]
            causal, scale = 1, 1.0
        elif name == "clip_l_577":   # CLIP ViT-L/14-336 of LLaVA-1.5: 576 patches + CLS, 16 heads x 64, own-range descriptors
            H, dh, n = 16, 64, 577
            q = bf(torch.randn(2 * n, H * dh, generator=g))
            k = bf(torch.randn(2 * n, H * dh, generator=g))
            v = bf(torch.randn(2 * n, H * dh, generator=g))
            desc = [(0, n, 0, 0, 0, n), (n, n, 0, 0, n, n)]
            causal, scale = 0, dh ** -0.5
        elif name == "llama_dh128":  # Vicuna-7B heads (32 x 128), causal, ~600-token sequences (576 image tokens + text)
            H, dh = 32, 128
            T = [598, 46, 130]
            tot = sum(T)
            q = bf(torch.randn(tot, H * dh, generator=g))
            k = bf(torch.randn(tot, H * dh, generator=g))
            v = bf(torch.randn(tot, H * dh, generator=g))
            st = np.cumsum([0] + T)
            desc = [(int(st[i]), T[i], 0, 0, int(st[i]), T[i]) for i in range(3)]
            causal, scale = 1, dh ** -0.5
        elif name == "llama_prefix_576":  # a shared 576-row image prefix (9 chunks) + two texts attending to it, head dim 128
            H, dh = 8, 128
            q = bf(torch.randn(576 + 20 + 70, H * dh, generator=g))
            k = bf(torch.randn(576 + 20 + 70, H * dh, generator=g))
            v = bf(torch.randn(576 + 20 + 70, H * dh, generator=g))
            desc = [(0, 576, 0, 0, 0, 576), (576, 20, 0, 576, 576, 20), (596, 70, 0, 576, 596, 70)]
            causal, scale = 1, dh ** -0.5
        elif name in ("opt_pack_32", "llama_pack_32", "dh64_pack_32"):
            # the bench's decoder pack: 32-row image prefixes, texts of 1..32 rows behind them (<= 64 keys: one chunk) or alone, one text behind a
            # LONG prefix (3 chunks) -- the wave-per-item pack kernel (opt-in) is run on it below
            H, dh = {"opt_pack_32": (32, 80), "llama_pack_32": (8, 128), "dh64_pack_32": (12, 64)}[name]
            lens = [32, 5, 17, 31, 32, 25, 1, 16, 32, 9, 150, 20]
            pre = [None, 0, 0, 0, None, 4, 4, 4, None, None, None, 10]
            st = np.cumsum([0] + lens)
            tot = int(st[-1])
            q = bf(torch.randn(tot, H * dh, generator=g))
            k = bf(torch.randn(tot, H * dh, generator=g))
            v = bf(torch.randn(tot, H * dh, generator=g))
            desc = [(int(st[i]), lens[i], int(st[pre[i]]) if pre[i] is not None else 0, lens[pre[i]] if pre[i] is not None else 0,
                     int(st[i]), lens[i]) for i in range(len(lens))]
            desc[10] = (int(st[10]), 0, 0, 0, int(st[10]), 0)        # the 150 rows are only ever a prefix (no queries of their own: q_len 0)
            keep = torch.ones(tot, dtype=torch.bool)
            keep[int(st[10]):int(st[11])] = False             # ... and no output rows
            causal, scale = 1, (1.0 if dh == 80 else dh ** -0.5)
        elif name in ("opt_pack_short", "llama_pack_short"):
            # a decoder probe pack: two 32-row prefixes + texts of 1..64 rows behind them (one or two key chunks, one or two 32-query
            # tiles) and texts without a prefix -- the default for it is the single-image two-wave LDS-DMA kernel
            H, dh = (32, 80) if name == "opt_pack_short" else (8, 128)
            lens = [32, 5, 17, 31, 32, 25, 1, 64, 33, 9]
            pre = [None, 0, 0, 0, None, 4, 4, 4, None, None]      # index of the prefix sequence each one sees
            st = np.cumsum([0] + lens)
            tot = int(st[-1])
            q = bf(torch.randn(tot, H * dh, generator=g))
            k = bf(torch.randn(tot, H * dh, generator=g))
            v = bf(torch.randn(tot, H * dh, generator=g))
            desc = [(int(st[i]), lens[i], int(st[pre[i]]) if pre[i] is not None else 0, lens[pre[i]] if pre[i] is not None else 0,
                     int(st[i]), lens[i]) for i in range(len(lens))]
            causal, scale = 1, (1.0 if dh == 80 else dh ** -0.5)
        elif name == "opt_prefix":  # prefix seq (32 rows) + two text seqs attending to it
            H, dh = 32, 80
            q = bf(torch.randn(32 + 16 + 21, H * dh, generator=g))
            k = bf(torch.randn(32 + 16 + 21, H * dh, generator=g))
            v = bf(torch.randn(32 + 16 + 21, H * dh, generator=g))
            desc = [(0, 32, 0, 0, 0, 32), (32, 16, 0, 32, 32, 16), (48, 21, 0, 32, 48, 21)]
            causal, scale = 1, 1.0
        else:
            H, dh = 5, 8
            q = bf(torch.randn(29, H * dh, generator=g))
            k = bf(torch.randn(29, H * dh, generator=g))
            v = bf(torch.randn(29, H * dh, generator=g))
            desc = [(0, 29, 0, 0, 0, 29)]
            causal, scale = 1, 0.7
        views = tuple(dev(t).to(torch.bfloat16) for t in (q, k, v))
    ref = _ref_attention(q, k, v, desc, H, dh, scale, causal)
    d = torch.tensor(desc, dtype=torch.int32, device="cuda")
    out = L.attention(views[0], views[1], views[2], d, len(desc), max(x[1] for x in desc), H, dh, scale, causal)
    np.testing.assert_allclose(out.float().cpu()[keep].numpy(), ref[keep].numpy(), atol=2e-2, rtol=2e-2)
    # both stagings of the chunked kernel on every case, whatever the launcher's default for the shape: register-staged (the default
    # for short sequences) and LDS-DMA with 64- and 128-query tiles (the default for long ones) -- same arithmetic, bit-identical
    # (the launcher's default may be the ring kernel with the last key folded -- non-causal, >= 224 queries: same tolerance against the reference
    # above, not the same bits; DEVQA_ATTENTION_FOLD=0 gives the unfolded form)
    import os
    outs = {}
    base_env = {"DEVQA_ATTENTION_FOLD": "0", "DEVQA_ATTENTION_PACK": "0"}
    for var in ({}, {"DEVQA_ATTENTION_DMA": "0"}, {"DEVQA_ATTENTION_DMA": "1", "DEVQA_ATTENTION_NW": "4"},
                {"DEVQA_ATTENTION_DMA": "1", "DEVQA_ATTENTION_NW": "8"}, {"DEVQA_ATTENTION_DMA": "1", "DEVQA_ATTENTION_NW": "9"},
                {"DEVQA_ATTENTION_NBUF": "3"}, {"DEVQA_ATTENTION_RING": "0"}, {"DEVQA_ATTENTION_SHORT": "0"}, {"DEVQA_ATTENTION_SHORT": "1"}):
        var = dict(base_env, **var)
        os.environ.update(var)
        try:
            o = L.attention(views[0], views[1], views[2], d, len(desc), max(x[1] for x in desc), H, dh, scale, causal)
        finally:
            for k_ in var:
                del os.environ[k_]
        np.testing.assert_allclose(o.float().cpu()[keep].numpy(), ref[keep].numpy(), atol=2e-2, rtol=2e-2, err_msg=str(var))
        outs[tuple(sorted(var.items()))] = o.cpu()[keep]
    vals = list(outs.values())
    assert all(torch.equal(vals[0], x) for x in vals[1:])
    for var in ({"DEVQA_ATTENTION_PACK": "1"}, {"DEVQA_ATTENTION_NBUF": "3"}, {"DEVQA_ATTENTION_NW": "8"}, {"DEVQA_ATTENTION_NW": "9"}):   # pack kernel up to 64 queries; ring forms with the fold
        os.environ.update(var)
        try:
            o = L.attention(views[0], views[1], views[2], d, len(desc), max(x[1] for x in desc), H, dh, scale, causal)
        finally:
            for k_ in var:
                del os.environ[k_]
        np.testing.assert_allclose(o.float().cpu()[keep].numpy(), ref[keep].numpy(), atol=2e-2, rtol=2e-2, err_msg=str(var))
    if name == "vit":   # the opt-in K/V-resident kernel (self_full promise; enough (sequence, head) pairs to be selected: 8 x 16)
        qkv8 = bf(torch.randn(8 * n, 3 * H * dh, generator=g))
        d8 = [(i * n, n, 0, 0, i * n, n) for i in range(8)]
        ref8 = _ref_attention(qkv8[:, :H * dh], qkv8[:, H * dh:2 * H * dh], qkv8[:, 2 * H * dh:], d8, H, dh, scale, 0)
        dq8 = dev(qkv8).to(torch.bfloat16)
        dd8 = torch.tensor(d8, dtype=torch.int32, device="cuda")
        import os
        os.environ["DEVQA_ATTENTION_RESIDENT"] = "1"
        try:
            o8 = L.attention(dq8[:, :H * dh], dq8[:, H * dh:2 * H * dh], dq8[:, 2 * H * dh:], dd8, 8, n, H, dh, scale, 0, self_full=True)
        finally:
            del os.environ["DEVQA_ATTENTION_RESIDENT"]
        np.testing.assert_allclose(o8.float().cpu().numpy(), ref8.numpy(), atol=2e-2, rtol=2e-2)
        o8c = L.attention(dq8[:, :H * dh], dq8[:, H * dh:2 * H * dh], dq8[:, 2 * H * dh:], dd8, 8, n, H, dh, scale, 0)
        np.testing.assert_allclose(o8.float().cpu().numpy(), o8c.float().cpu().numpy(), atol=1e-2, rtol=1e-2)
    if name == "vit":   # the experimental 128-query-tile instantiation (two query blocks per wave) on the same inputs
        import os
        os.environ["DEVQA_ATTENTION_QB"] = "2"
        try:
            out2 = L.attention(views[0], views[1], views[2], d, len(desc), max(x[1] for x in desc), H, dh, scale, causal)
        finally:
            del os.environ["DEVQA_ATTENTION_QB"]
        np.testing.assert_allclose(out2.float().cpu().numpy(), ref.numpy(), atol=2e-2, rtol=2e-2)


def test_patch_embed_and_assemble(L):
    g = torch.Generator().manual_seed(2)
    B, S, P, D = 2, 28, 14, 48
    pix = torch.randn(B, 3, S, S, generator=g)
    w = bf(torch.randn(D, 3, P, P, generator=g) * 0.05)
    b = torch.randn(D, generator=g) * 0.1
    ref = torch.nn.functional.conv2d(bf(pix), w, b, stride=P).flatten(2).transpose(1, 2)
    Kreal, Kpad = 3 * P * P, ((3 * P * P + 31) // 32) * 32
    wp = torch.zeros(D, Kpad)
    wp[:, :Kreal] = w.reshape(D, -1)
    cols = L.im2col_patches(dev(pix), P, Kpad)
    out = L.gemm(cols, dev(wp).to(torch.bfloat16), dev(b), want="f32")
    np.testing.assert_allclose(out.cpu().numpy().reshape(B, -1, D), ref.numpy(), atol=1e-4, rtol=1e-4)
    cls = torch.randn(D, generator=g)
    pos = torch.randn(5, D, generator=g)
    x = L.vit_assemble(out, dev(cls), dev(pos), B, 4, D).cpu().reshape(B, 5, D)
    refx = torch.cat([cls.expand(B, 1, D), ref], 1) + pos
    np.testing.assert_allclose(x.numpy(), refx.numpy(), atol=1e-4, rtol=1e-4)


def test_embed_rows_and_gather(L):
    g = torch.Generator().manual_seed(4)
    V, D, npos = 640, 40, 130
    emb = bf(torch.randn(V, D, generator=g))
    post = bf(torch.randn(npos, D, generator=g))
    img = torch.randn(8, D, generator=g)
    token = torch.tensor([0] * 8 + [2, 17, 639, 5], dtype=torch.int32)
    src = torch.tensor(list(range(8)) + [-1] * 4, dtype=torch.int32)
    pos = torch.arange(12, dtype=torch.int32)
    out = L.embed_rows(dev(token), dev(src), dev(pos), dev(emb).to(torch.bfloat16), dev(img), dev(post).to(torch.bfloat16))
    ref = torch.cat([img, emb[token[8:].long()]]) + post[pos.long() + 2]
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), atol=1e-6)
    idx = torch.tensor([3, 0, 11, 3], dtype=torch.int32)
    got = L.gather_rows(out, dev(idx))
    np.testing.assert_array_equal(got.cpu().numpy(), out.cpu().numpy()[idx.numpy()])
    c = L.cast_f32_bf16(out)
    np.testing.assert_array_equal(c.float().cpu().numpy(), out.cpu().to(torch.bfloat16).float().numpy())


@pytest.mark.parametrize("R,V", [(7, 50272), (3, 640), (2, 1001 * 4)])
def test_vocab_rows(L, R, V):
    g = torch.Generator().manual_seed(V)
    logits = torch.randn(R, V, generator=g) * 4
    logits[0, 5] = logits[0, 900 % V] = logits[0].max() + 1  # tie -> first index wins
    labels = torch.randint(0, V, (R,), generator=g).to(torch.int32)
    coef = torch.rand(R, generator=g)
    am, nll, dl = L.vocab_rows(dev(logits), dev(labels), dev(coef), True, True, True)
    np.testing.assert_array_equal(am.cpu().numpy(), logits.argmax(-1).numpy())
    lp = torch.log_softmax(logits, -1)
    ref_nll = -lp.gather(-1, labels.long()[:, None])[:, 0]
    np.testing.assert_allclose(nll.cpu().numpy(), ref_nll.numpy(), atol=2e-4, rtol=1e-5)
    ref_dl = torch.softmax(logits, -1)
    ref_dl[torch.arange(R), labels.long()] -= 1
    ref_dl *= coef[:, None]
    np.testing.assert_allclose(dl.float().cpu().numpy(), ref_dl.numpy(), atol=1e-5, rtol=1e-2)


# Lmax > 16: the wide kernel (a-rows re-read per column step), 17..32 and 33..64 instantiations.  form: "matrix" = devqa_ft_adamw_step (first-moment
# matrix), "factored" = devqa_ft_adamw_step_fm (the form the path runs: the first moment is the EMA of dy times the constant a-rows, no matrix).
# Din 80: the lane-group / wave-per-row kernels of column-compacted matrices; Din 1100: the workgroup-per-row-block kernel of dense ones.
@pytest.mark.parametrize("form", ["matrix", "factored"])
@pytest.mark.parametrize("Lmax,wd,clamp,Din", [(1, 0.0, -1.0, 80), (3, 0.0, -1.0, 80), (2, 0.1, -1.0, 80), (3, 0.0, 2.5e-3, 80), (6, 0.0, -1.0, 80),
                                               (13, 0.0, -1.0, 80), (17, 0.0, -1.0, 80), (29, 0.1, 2.5e-3, 80), (40, 0.0, -1.0, 80), (64, 0.0, 2.5e-3, 80),
                                               (1, 0.0, -1.0, 1100), (3, 0.1, 2.5e-3, 1100), (12, 0.0, -1.0, 1100), (20, 0.0, -1.0, 1100)])
def test_ft_adamw_step_matches_torch_adamw(L, Lmax, wd, clamp, Din, form):
    g = torch.Generator().manual_seed(Lmax)
    E, Dout = 3, 40
    lr = 1e-3
    w0 = torch.randn(Dout, Din, generator=g) * 0.05
    a = torch.relu(torch.randn(E, Lmax, Din, generator=g))
    # factored form: edit 2 is a ONE-ROW edit (only slot 0 carries a loss row): flagged `single`, its second moment is factored as well and
    # its v matrix is never touched
    one_row = form == "factored"
    if one_row:
        a[2, 1:] = 0
    w = torch.zeros(E, Dout, Din, device="cuda")
    m = torch.zeros_like(w) if form == "matrix" else torch.full((E, Lmax + 1, Dout), float("nan"), device="cuda")   # (the state is ignored at the first update)
    v = torch.zeros_like(w)
    if one_row:
        v[2] = 7.0
    single = torch.tensor([0, 0, 1], dtype=torch.int32, device="cuda") if one_row else None
    y = torch.zeros(E, Lmax, Dout, device="cuda")
    do_update = torch.tensor([1, 0, 1], dtype=torch.int32, device="cuda")  # edit 1 never updates
    adam_t = torch.zeros(E, dtype=torch.int32, device="cuda")
    params = [w0.clone().requires_grad_(True) for _ in range(E)]
    opts = [torch.optim.AdamW([p], lr=lr, weight_decay=wd) for p in params]
    for step in range(5):
        dy = torch.randn(E, Lmax, Dout, generator=g) * 0.1
        if one_row:
            dy[2, 1:] = 0
        if step == 3:
            do_update[0] = 0           # a skipped step in the middle (loss under the floor): state and outputs of edit 0 stay
        elif step == 4:
            do_update[0] = 1
        adam_t += do_update
        if form == "matrix":
            L.ft_adamw_step(w, m, v, dev(w0), dev(a), dev(dy), y, do_update, adam_t, lr, 0.9, 0.999, 1e-8, wd, clamp)
        else:
            L.ft_adamw_step_fm(w, m, v, dev(w0), dev(a), dev(dy), y, do_update, adam_t, lr, 0.9, 0.999, 1e-8, wd, clamp, single=single)
        for e in range(E):
            if int(do_update[e]) == 0:
                continue
            params[e].grad = torch.einsum("lo,li->oi", dy[e], a[e])
            opts[e].step()
            if clamp >= 0:
                with torch.no_grad():
                    params[e][...] = torch.clamp(params[e], w0 - clamp, w0 + clamp)
        torch.cuda.synchronize()
        for e in (0, 2):
            if form == "matrix":
                np.testing.assert_allclose(w[e].cpu().numpy(), params[e].detach().numpy(), atol=2e-6, rtol=1e-5)
            else:
                # m and v now come from differently rounded sums: where the L terms of a gradient element cancel to ~1e-4 of their size the
                # ratio m / sqrt(v) is ill-conditioned (in the reference too) and a weight may move by a few 1e-6 -- a handful of elements
                # of 44000 with 20 random rows, none at the path's L <= 3 (tests/test_realdim_batched_gpu.py: both forms 9.2e-6 of the reference)
                err = (w[e].cpu() - params[e].detach()).abs()
                assert float(err.max()) < 3e-5 and float((err > 2e-6 + 1e-5 * params[e].detach().abs()).float().mean()) < 1e-3, float(err.max())
            ref_y = a[e] @ params[e].detach().T
            np.testing.assert_allclose(y[e].cpu().numpy(), ref_y.numpy(), atol=(1e-4 if form == "matrix" else 2e-4) * (Din / 80) ** 0.5, rtol=1e-4)
            ref_m = opts[e].state[params[e]]["exp_avg"]
            got_m = m[e].cpu() if form == "matrix" else torch.einsum("lo,li->oi", m[e, :Lmax].cpu(), a[e])
            np.testing.assert_allclose(got_m.numpy(), ref_m.numpy(), atol=1e-6, rtol=1e-4)
            ref_v = opts[e].state[params[e]]["exp_avg_sq"]
            if one_row and e == 2:      # v_t = e_t (x) a[0]^2, the matrix untouched
                np.testing.assert_allclose(torch.outer(m[e, Lmax].cpu(), a[e, 0] ** 2).numpy(), ref_v.numpy(), atol=1e-9, rtol=1e-4)
                assert float((v[e] - 7.0).abs().max()) == 0.0
            else:
                np.testing.assert_allclose(v[e].cpu().numpy(), ref_v.numpy(), atol=1e-9, rtol=1e-4)
    assert float(y[1].abs().sum()) == 0.0  # inactive edit untouched


@pytest.mark.parametrize("Lr", [16, 17, 37, 64])
def test_rows_matvec_many_rows(L, Lr):
    """More than 16 rows per edit go 16 at a time with the edit stride of the whole block."""
    g = torch.Generator().manual_seed(Lr)
    E, Dout, Din = 3, 42, 72
    w = torch.randn(E, Dout, Din, generator=g)
    a = torch.randn(E, Lr, Din, generator=g)
    b = torch.randn(Dout, generator=g)
    r = torch.randn(E, Lr, Dout, generator=g)
    y = L.rows_matvec(dev(w), dev(a), dev(b), dev(r))
    np.testing.assert_allclose(y.cpu().numpy(), (torch.einsum("eoi,eli->elo", w, a) + b + r).numpy(), atol=1e-4, rtol=1e-4)
    y2 = L.rows_matvec(dev(w[1]), dev(a), shared=True)
    np.testing.assert_allclose(y2.cpu().numpy(), torch.einsum("oi,eli->elo", w[1], a).numpy(), atol=1e-4, rtol=1e-4)


def test_rows_matvec_delta_and_control(L):
    g = torch.Generator().manual_seed(9)
    E, Lr, Dout, Din = 2, 5, 40, 80
    w = torch.randn(E, Dout, Din, generator=g)
    a = torch.randn(E, Lr, Din, generator=g)
    b = torch.randn(Dout, generator=g)
    r = torch.randn(E, Lr, Dout, generator=g)
    y = L.rows_matvec(dev(w), dev(a), dev(b), dev(r))
    ref = torch.einsum("eoi,eli->elo", w, a) + b + r
    np.testing.assert_allclose(y.cpu().numpy(), ref.numpy(), atol=1e-4, rtol=1e-4)
    y2 = L.rows_matvec(dev(w[0]), dev(a), shared=True)
    np.testing.assert_allclose(y2.cpu().numpy(), torch.einsum("oi,eli->elo", w[0], a).numpy(), atol=1e-4, rtol=1e-4)
    # delta ops
    w0 = dev(w[0].contiguous())
    wc = dev(w[1].contiguous())
    d = torch.empty_like(wc)
    L.delta_op(0, wc, w0, d)
    np.testing.assert_array_equal(d.cpu().numpy(), (w[1] - w[0]).numpy())
    L.delta_op(2, wc, w0, None)
    np.testing.assert_array_equal(wc.cpu().numpy(), w[0].numpy())
    L.delta_op(1, wc, None, d)
    np.testing.assert_array_equal(wc.cpu().numpy(), (w[0] + (w[1] - w[0])).numpy())
    # loop control
    nll = dev(torch.tensor([[2.0, 4.0, 9.0], [0.004, 0.006, 9.0], [1.0, 1.0, 1.0]]))
    mask = dev(torch.tensor([[1.0, 1.0, 0.0], [1.0, 1.0, 0.0], [1.0, 0.0, 0.0]]))
    active = torch.tensor([1, 1, 0], dtype=torch.int32, device="cuda")
    do_update = torch.zeros(3, dtype=torch.int32, device="cuda")
    n_steps = torch.zeros(3, dtype=torch.int32, device="cuda")
    adam_t = torch.zeros(3, dtype=torch.int32, device="cuda")
    losses = torch.zeros(3, 25, device="cuda")
    L.ft_step_control(nll, mask, 0, 25, 1e-2, active, do_update, n_steps, adam_t, losses)
    assert active.tolist() == [1, 0, 0] and do_update.tolist() == [1, 0, 0]
    assert n_steps.tolist() == [1, 1, 0] and adam_t.tolist() == [1, 0, 0]
    np.testing.assert_allclose(losses[:, 0].cpu().numpy(), [3.0, 0.005, 0.0], rtol=1e-6)


# Q <= 4 takes the few-query path (row-wise scores + wave-level selection), more queries the tiled one
@pytest.mark.parametrize("N,Q,D,k", [(15000, 100, 384, 5), (19035, 7, 384, 32), (300, 3, 64, 5), (15000, 1, 384, 5), (15000, 1, 384, 32),
                                     (19035, 4, 384, 32), (64, 2, 64, 5), (130, 1, 768, 1)])
def test_cosine_topk_exact_indices(L, N, Q, D, k):
    from oracle import devqa_oracle as O
    rng = np.random.default_rng(N + Q)
    c = rng.standard_normal((N, D)).astype(np.float32)
    q = rng.standard_normal((Q, D)).astype(np.float32)
    q[0] = c[17]  # exact hit
    c[41] = c[17]  # duplicate row: tie -> lowest id first
    ridx, rsc = O.cosine_topk(c, q, k)
    idx, sc = L.cosine_topk(dev(torch.from_numpy(c)), dev(torch.from_numpy(q)), k)
    np.testing.assert_array_equal(idx.cpu().numpy(), ridx)
    np.testing.assert_allclose(sc.cpu().numpy(), rsc, atol=1e-6)
    assert idx[0, 0].item() == 17 and (k < 2 or idx[0, 1].item() == 41)
    # raw dot score (no normalisation), as semantic_search(dot_score) on pre-normalised data
    ridx2, _ = O.cosine_topk(c, q, k, False, False)
    idx2, _ = L.cosine_topk(dev(torch.from_numpy(c)), dev(torch.from_numpy(q)), k, False, False)
    np.testing.assert_array_equal(idx2.cpu().numpy(), ridx2)


@pytest.mark.parametrize("N,Q,k", [(15000, 1, 5), (15000, 1, 32), (15000, 3, 5), (20480, 4, 32), (19035, 2, 5), (300, 1, 32), (20, 2, 32), (33, 1, 1)])
def test_cosine_topk_few_queries_one_launch(L, N, Q, k):
    """The Q <= 4 path (finds_sim / IKE_VL / LTE_VL retrieval; ONE launch: scores + radix select + fp64 re-score by the last-arriving
    workgroup): exact indices vs the float64 brute force, with MASSIVE ties (blocks of identical rows -> the lowest ids must win, more
    equal keys than places), with the corpus' inverse norms cached (the product call sites) and without, twice in a row (the counter
    the workgroups meet on must come back to zero), and against the many-query tiled path on the same inputs."""
    from oracle import devqa_oracle as O
    D = 384
    rng = np.random.default_rng(N * 7 + Q + k)
    c = rng.standard_normal((N, D)).astype(np.float32)
    q = rng.standard_normal((Q, D)).astype(np.float32)
    if N >= 300:
        q[0] = c[123] * 2.5
        for j in range(124, 124 + 60):          # 61 identical best rows for query 0: ties far beyond k + 8
            c[j] = c[123]
        c[N - 1] = c[5]                         # a far-apart duplicate pair
    def pad(idx_, sc_=None):       # a corpus smaller than k: the product pads the id row with -1
        n = idx_.shape[1]
        if n == k:
            return idx_, sc_
        return (np.concatenate([idx_, np.full((idx_.shape[0], k - n), -1, np.int64)], 1),
                None if sc_ is None else np.concatenate([sc_, np.full((idx_.shape[0], k - n), -np.inf)], 1))
    ridx, rsc = pad(*O.cosine_topk(c, q, k))
    ct, qt = dev(torch.from_numpy(c)), dev(torch.from_numpy(q))
    inv = L.row_inv_norm(ct)
    np.testing.assert_allclose(inv.cpu().numpy(), 1.0 / np.linalg.norm(c.astype(np.float64), axis=1), rtol=2e-6)
    for cached in (None, inv, None, inv):
        idx, sc = L.cosine_topk(ct, qt, k, True, True, corpus_inv_norm=cached)
        torch.cuda.synchronize()
        got, gsc = idx.cpu().numpy(), sc.cpu().numpy()
        valid = ridx >= 0
        np.testing.assert_array_equal(got, ridx)
        np.testing.assert_allclose(gsc[valid], rsc[valid], atol=1e-6)
    if N >= 300:
        assert got[0, :min(k, 61)].tolist() == list(range(123, 123 + min(k, 61)))
    # the tiled many-query path (Q > 4) on the same queries, repeated
    q5 = np.concatenate([q, q, q, q, q])[:max(5, Q)]
    idx5, _ = L.cosine_topk(ct, dev(torch.from_numpy(q5)), k, True, True)
    np.testing.assert_array_equal(idx5.cpu().numpy()[:Q], ridx)
    # raw dot product (no normalisation on either side)
    ridx2, _ = pad(O.cosine_topk(c, q, k, False, False)[0])
    idx2, _ = L.cosine_topk(ct, qt, k, False, False)
    np.testing.assert_array_equal(idx2.cpu().numpy(), ridx2)


def test_column_compaction(L):
    g = torch.Generator().manual_seed(21)
    E, Lr, Din, Dout = 3, 2, 10240, 24
    a = torch.relu(torch.randn(E, Lr, Din, generator=g) - 1.8)  # ~3.6% active
    a[1] = 0
    a[1, 0, 5] = 1.0
    a[1, 1, 10239] = 2.0
    idx, cnt = L.active_columns(dev(a))
    for e in range(E):
        ref = torch.nonzero((a[e] != 0).any(0))[:, 0]
        assert int(cnt[e]) == len(ref)
        np.testing.assert_array_equal(idx[e, :len(ref)].cpu().numpy(), ref.numpy())
    npad = (int(cnt.max()) + 7) // 8 * 8
    w = torch.randn(Dout, Din, generator=g)
    wc = L.gather_cols(dev(w), idx, cnt, npad, per_edit=False).cpu()
    ac = L.gather_cols(dev(a), idx, cnt, npad, per_edit=True).cpu()
    for e in range(E):
        n = int(cnt[e])
        j = idx[e, :n].cpu().long()
        np.testing.assert_array_equal(wc[e, :, :n].numpy(), w[:, j].numpy())
        assert float(wc[e, :, n:].abs().sum()) == 0.0
        np.testing.assert_array_equal(ac[e, :, :n].numpy(), a[e][:, j].numpy())
        # the compacted product carries all of W.a
        np.testing.assert_allclose((ac[e] @ wc[e].T).numpy(), (a[e] @ w.T).numpy(), atol=1e-4, rtol=1e-4)
    ab = dev(a[0]).to(torch.bfloat16).contiguous()
    gb = L.gather_cols(ab, idx[0:1], cnt[0:1], npad, per_edit=False)[0].float().cpu()
    n0 = int(cnt[0])
    np.testing.assert_array_equal(gb[:, :n0].numpy(), ab.float().cpu()[:, idx[0, :n0].cpu().long()].numpy())
    dense = torch.zeros(Dout, Din, device="cuda")
    L.scatter_cols_add(dev(wc[0].contiguous()), idx[0], cnt[0:1], dense)
    ref = torch.zeros(Dout, Din)
    ref[:, idx[0, :n0].cpu().long()] = w[:, idx[0, :n0].cpu().long()]
    np.testing.assert_array_equal(dense.cpu().numpy(), ref.numpy())


def test_attention_is_deterministic_and_variants_agree(L):
    """Multi-chunk causal sequences (> 64 keys: the K/V refill path) several times over, and the opt-in instantiations: bit-identical.
    Guards the software-managed MFMA -> VALU read hazard of the inline-asm max chain (csrc/attention_mfma.hip, am_max16): without its
    wait states the running max was read stale now and then -- results within tolerance but different from run to run.
    The non-causal case takes the ring kernel by default (9-wave tiles; the last key of 65 / 129 / 257 is processed from LDS after the chunk loop
    instead of as a chunk of its own): deterministic; bit-identical to the other stagings with the fold off; within bf16 rounding of them with
    it on.  Sequences of 1, 2 and 63 keys share the launch (no chunk to fold into, a lone key, a short chunk)."""
    import os
    torch.manual_seed(3)
    lens = [104, 30, 70, 129, 64, 65, 257, 1, 2, 63]
    starts = np.cumsum([0] + lens[:-1]).tolist()
    R = sum(lens)
    for H, dh, causal in ((5, 8, 1), (32, 80, 1), (16, 88, 0), (16, 64, 0)):
        d = H * dh
        qkv = torch.randn(R, 3 * d, device="cuda").to(torch.bfloat16)
        desc = torch.tensor([[s, n, 0, 0, s, n] for s, n in zip(starts, lens)], dtype=torch.int32, device="cuda")
        q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]

        def run():
            out = torch.full((R, d), 7.0, device="cuda", dtype=torch.bfloat16)
            L.attention(q, k, v, desc, len(lens), max(lens), H, dh, dh ** -0.5, causal, out=out)
            torch.cuda.synchronize()
            return out
        folded = run()
        for _ in range(5):
            assert torch.equal(run(), folded)
        os.environ["DEVQA_ATTENTION_FOLD"] = "0"
        try:
            base = run()
            for _ in range(5):
                assert torch.equal(run(), base)
            if causal:
                assert torch.equal(base, folded)          # only the non-causal ring kernel folds
            else:
                err = (base.float() - folded.float()).abs().max().item()
                assert err < 4e-2, err                     # (the folded key's fp32 score comes from a VALU fma chain: an output may move by a bf16 ulp)
            for var in ("DEVQA_ATTENTION_DBUF", "DEVQA_ATTENTION_QB"):
                os.environ[var] = "1" if var.endswith("DBUF") else "2"
                try:
                    assert torch.equal(run(), base), var
                finally:
                    del os.environ[var]
            for dma, nw in (("0", "4"), ("1", "4"), ("1", "8"), ("1", "9")):      # both stagings, the tile sizes of the LDS-DMA / ring kernels
                os.environ["DEVQA_ATTENTION_DMA"], os.environ["DEVQA_ATTENTION_NW"] = dma, nw
                try:
                    for _ in range(3):
                        assert torch.equal(run(), base), (dma, nw)
                finally:
                    del os.environ["DEVQA_ATTENTION_DMA"], os.environ["DEVQA_ATTENTION_NW"]
            os.environ["DEVQA_ATTENTION_RING"] = "0"       # the two-image LDS-DMA kernel on the same sequences
            try:
                assert torch.equal(run(), base), "ring off"
            finally:
                del os.environ["DEVQA_ATTENTION_RING"]
        finally:
            del os.environ["DEVQA_ATTENTION_FOLD"]


def test_layernorm_param_grads_and_colsum(L):
    """dgamma / dbeta of LayerNorm over packed rows (+ accumulate) and column sums, vs float64 autograd; deterministic."""
    g = torch.Generator().manual_seed(9)
    for M, D in ((1, 40), (37, 40), (300, 2560), (129, 1408)):
        x = torch.randn(M, D, generator=g) * 2 + 0.3
        add = torch.randn(M, D, generator=g)
        dy = torch.randn(M, D, generator=g)
        xd = (x + add).double().requires_grad_(False)
        gam = torch.ones(D, dtype=torch.float64, requires_grad=True)
        bet = torch.zeros(D, dtype=torch.float64, requires_grad=True)
        y = torch.nn.functional.layer_norm(xd, (D,), gam, bet, 1e-5)
        (y * dy.double()).sum().backward()
        dg = torch.full((D,), 0.5, device="cuda")
        db = torch.full((D,), -0.25, device="cuda")
        L.layernorm_bwd_params(dev(x), dev(dy), 1e-5, dg, db, add=dev(add), accumulate=True)
        np.testing.assert_allclose(dg.cpu().numpy() - 0.5, gam.grad.float().numpy(), rtol=2e-4, atol=2e-4 * M ** 0.5)
        np.testing.assert_allclose(db.cpu().numpy() + 0.25, bet.grad.float().numpy(), rtol=2e-4, atol=2e-4 * M ** 0.5)
        dg2, db2 = torch.empty(D, device="cuda"), torch.empty(D, device="cuda")
        L.layernorm_bwd_params(dev(x), dev(dy), 1e-5, dg2, db2, add=dev(add), accumulate=False)
        dg3, db3 = torch.empty(D, device="cuda"), torch.empty(D, device="cuda")
        L.layernorm_bwd_params(dev(x), dev(dy), 1e-5, dg3, db3, add=dev(add), accumulate=False)
        assert torch.equal(dg2, dg3) and torch.equal(db2, db3)
        # LlamaRMSNorm: xhat = x * rsqrt(mean(x^2) + eps), weight only
        wr = torch.ones(D, dtype=torch.float64, requires_grad=True)
        yr = xd * torch.rsqrt((xd * xd).mean(-1, keepdim=True) + 1e-6) * wr
        (yr * dy.double()).sum().backward()
        dgr = torch.zeros(D, device="cuda")
        L.layernorm_bwd_params(dev(x), dev(dy), 1e-6, dgr, None, add=dev(add), accumulate=True, rms=True)
        np.testing.assert_allclose(dgr.cpu().numpy(), wr.grad.float().numpy(), rtol=2e-4, atol=2e-4 * M ** 0.5)
        cs = torch.zeros(D, device="cuda")
        L.colsum_(dev(dy), cs, True)
        L.colsum_(dev(dy), cs, True)
        np.testing.assert_allclose(cs.cpu().numpy(), 2 * dy.double().sum(0).float().numpy(), rtol=1e-5, atol=1e-5 * M ** 0.5)


@pytest.mark.parametrize("dh,H", [(128, 64), (64, 24), (32, 5), (96, 8), (8, 6)])
def test_rope_bf16_vector_form_is_the_elementwise_form(L, dh, H):
    """Rotary embedding (HF rotate_half convention) on q | k heads of a fused buffer: the 16-byte form (dh % 32 == 0) against the element-wise
    kernel -- reached through a row stride that is not a multiple of 8 -- bit for bit, and both against a float64 restatement."""
    g = torch.Generator().manual_seed(dh)
    R, theta = 37, 10000.0
    pos = torch.randint(0, 700, (R,), generator=g).to(torch.int32)
    x = (torch.randn(R, H * dh, generator=g) * 2).to(torch.bfloat16)
    a = x.clone().cuda()
    L.rope_(a, pos.cuda(), H, dh, theta)
    wide = torch.zeros(R, H * dh + 4, dtype=torch.bfloat16, device="cuda")
    b = wide[:, :H * dh]
    b.copy_(x.cuda())
    L.rope_(b, pos.cuda(), H, dh, theta)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    half = dh // 2
    inv = theta ** (-torch.arange(half, dtype=torch.float64) * 2 / dh)
    ang = pos.double()[:, None] * inv[None, :]
    xd = x.double().view(R, H, dh)
    ref = torch.cat([xd[..., :half] * ang.cos()[:, None] - xd[..., half:] * ang.sin()[:, None],
                     xd[..., half:] * ang.cos()[:, None] + xd[..., :half] * ang.sin()[:, None]], -1).view(R, H * dh)
    np.testing.assert_allclose(a.float().cpu().numpy(), ref.numpy(), atol=6e-2, rtol=1e-2)


@pytest.mark.parametrize("M,F,K", [(4096, 2048, 512), (3000, 2560, 256)])
def test_gemm_fused_swiglu(L, M, F, K):
    """silu(a . gate^T) * (a . up^T) in the GEMM's epilogue (interleaved [gate | up] rows, DEVQA_ACT_SWIGLU_IL16) against the two-pass form it
    replaces (GEMM to a bf16 [M, 2F] buffer, then the SwiGLU kernel) and a float64 restatement.  The fused form multiplies the fp32 accumulators
    -- it skips the bf16 rounding of gate and up -- so it agrees with the two-pass form to bf16 rounding and is the closer one to float64."""
    g = torch.Generator().manual_seed(F)
    a = (torch.randn(M, K, generator=g)).to(torch.bfloat16)
    w = (torch.randn(2 * F, K, generator=g) / K ** 0.5).to(torch.bfloat16)
    assert L.gemm_swiglu_supported(M, 2 * F, K) and not L.gemm_swiglu_supported(48, 2 * F, K) and not L.gemm_swiglu_supported(M, 2 * F + 64, K)
    fused = L.gemm_swiglu(a.cuda(), L.interleave_gate_up(w.cuda()))
    gu = L.gemm(a.cuda(), w.cuda())
    two = torch.empty(M, F, dtype=torch.bfloat16, device="cuda")
    L._chk(L.load().devqa_swiglu_bf16(L._p(gu), M, F, L._p(two), L._stream()), "swiglu")
    acc = a.double() @ w.double().T
    ref = torch.nn.functional.silu(acc[:, :F]) * acc[:, F:]
    e_f = (fused.double().cpu() - ref).abs().max().item()
    e_t = (two.double().cpu() - ref).abs().max().item()
    scale = ref.abs().max().item()
    print("fused SwiGLU: max |err| %.3g, two-pass %.3g (values up to %.3g)" % (e_f, e_t, scale))
    assert e_f < 8e-3 * scale and e_f <= e_t * 1.05
    np.testing.assert_allclose(fused.float().cpu().numpy(), two.float().cpu().numpy(), atol=2e-2 * scale, rtol=2e-2)



def test_gelu_forward_and_backward(L):
    """devqa_gelu_f32 / devqa_gelu_bwd_f32 (the Q-Former FFN under FT_VL's Q-Former selection) against torch's erf GELU and its autograd"""
    torch.manual_seed(11)
    x = (torch.randn(77, 130, device="cuda") * 2.5).contiguous()
    g = torch.randn_like(x)
    xr = x.double().requires_grad_(True)
    yr = torch.nn.functional.gelu(xr)
    yr.backward(g.double())
    assert (L.gelu(x, want="f32").double() - yr.detach()).abs().max().item() < 2e-6
    assert torch.equal(L.gelu(x, want="bf16"), L.gelu(x, want="f32").to(torch.bfloat16))
    assert (L.gelu_bwd(x, g).double() - xr.grad).abs().max().item() < 5e-6


@pytest.mark.parametrize("kind", ["self_causal", "self_full", "cross"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attention_bwd_vs_autograd(L, kind, dtype):
    """dq / dk / dv of the packed attention against float64 autograd: causal own keys (the decoders), full own keys and CROSS attention --
    32 queries over 70 keys of another row set, named through the descriptor's own-key fields (the Q-Former's cross-attention under
    FT_VL's Q-Former selection)."""
    torch.manual_seed(5)
    H, dh = 3, 16
    d = H * dh
    if kind == "cross":
        qn, kn = [32, 32], [70, 70]
    else:
        qn = kn = [37, 5, 64]
    q = torch.randn(sum(qn), d, device="cuda")
    k = torch.randn(sum(kn), d, device="cuda")
    v = torch.randn(sum(kn), d, device="cuda")
    go = torch.randn(sum(qn), d, device="cuda")
    qs, ks = np.cumsum([0] + qn[:-1]).tolist(), np.cumsum([0] + kn[:-1]).tolist()
    desc_b = torch.tensor([[a, n, 0, 0, b, m] for a, n, b, m in zip(qs, qn, ks, kn)], dtype=torch.int32, device="cuda")
    causal = int(kind == "self_causal")
    qc, kc, vc, gc = (t.to(dtype).contiguous() for t in (q, k, v, go))
    refs = [t.to(dtype).double().requires_grad_(True) for t in (q, k, v)]
    outs = []
    for a, n, b, m in zip(qs, qn, ks, kn):
        qq, kk, vv = (t[s:s + ln].view(ln, H, dh).transpose(0, 1) for t, s, ln in ((refs[0], a, n), (refs[1], b, m), (refs[2], b, m)))
        sc = qq @ kk.transpose(1, 2) * dh ** -0.5
        if causal:
            sc = sc + torch.full((n, m), float("-inf"), device="cuda", dtype=torch.float64).triu(1)
        outs.append((torch.softmax(sc, -1) @ vv).transpose(0, 1).reshape(n, d))
    o = torch.cat(outs, 0)
    o.backward(gc.double())
    dq, dk, dv = L.attention_bwd(qc, kc, vc, o.detach().to(dtype).contiguous(), gc, desc_b, len(qn), max(max(qn), max(kn)), H, dh, dh ** -0.5, causal)
    assert dk.shape == k.shape and dv.shape == v.shape
    tol = 2e-5 if dtype == torch.float32 else 6e-2
    for got, ref in zip((dq, dk, dv), refs):
        err = (got.double() - ref.grad).abs().max().item() / max(ref.grad.abs().max().item(), 1e-9)
        assert err < tol, (kind, dtype, err)
