"""GPU: the PATH-LEVEL C ABI (include/devqa.h "PATH LEVEL", csrc/path_ctx.hip).

1. One complete tiny edit+eval cycle driven through the path-level entry points ONLY -- devqa_ctx_create, devqa_vision_encode,
   devqa_llm_prefix / devqa_llm_layers (the frozen prefix), devqa_llm_forward, devqa_llm_head, devqa_ft_edit, devqa_ctx_bind_edit_target,
   devqa_apply_delta, devqa_restore, devqa_token_acc -- plus op-level packing helpers (embed_rows / gather_rows / gemm), with NO use of
   engine.py / batched.py / the evaluator: what a non-Python host would write.  Checked against the REFERENCE goldens: G4 (per-step
   losses, step count, weight delta of `FTvl.execute_ft`) and G5 (`results.json` of `evaluate_sequential_edit`).
2. The product's engine, which calls the same entry points, equals the Python-ordered op-level schedule (DEVQA_PATH_ABI=0) bit for bit.
3. Error behaviour: status codes + devqa_last_error, never an exception from C."""
import ctypes
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
WNAME = "language_model.model.decoder.layers.1.fc2.weight"
LOC = ["text_loc", "t3i3", "t1i4", "t2i4", "t1i2", "t1i3", "t2i1", "t2i2", "t3i1"]


@pytest.fixture(scope="module", params=["fp32", "bf16"])
def tiny(gold_dir, request):
    import devqa_amd  # noqa: F401
    from devqa_amd.editor.vllms_for_edit.blip2.blip2 import BLIP2OPTForEdit
    vllm = BLIP2OPTForEdit(os.path.join(gold_dir, "tiny_blip2"), "cuda:0", dtype=request.param)
    vllm.model.promote_to_fp32(WNAME)
    j = json.load(open(os.path.join(gold_dir, "tiny_goldens.json")))
    z = np.load(os.path.join(gold_dir, "tiny_goldens.npz"))
    rec = json.load(open(os.path.join(gold_dir, "evqa8_records.json")))["records"]
    return vllm, j, z, rec, request.param


class Host:
    """Everything a host keeps in its own language: tokenisation, image decoding, label/mask bookkeeping, sequence packing."""

    def __init__(self, vllm):
        from devqa_amd import lib
        self.v, self.lib = vllm, lib
        self.m = vllm.model
        self.ctx = vllm.engine.path_ctx()           # devqa_ctx_create over the model's weight table
        assert self.ctx is not None
        self.Q = vllm.get_img_token_n()
        self.dev = torch.device("cuda:0")

    def probe(self, prompt, image, target):
        strs, y, m, _ = self.v.xym_token_bookkeeping([prompt], [target])
        return {"ids": self.v.tokenizer(strs[0])["input_ids"], "y": y[0], "m": m[0], "image": image}

    def pack(self, probes):
        """-> (x fp32 [R, d], desc int32 [n, 6], max_len, per-probe (start, length)); images encoded by devqa_vision_encode"""
        lib = self.lib
        imgs = list(dict.fromkeys(p["image"] for p in probes if p["image"] is not None))
        rows = None
        if imgs:
            pix = torch.from_numpy(np.stack([self.v.load_pixels(i) for i in imgs])).to(self.dev)
            rows = self.ctx.vision_encode(pix).reshape(-1, self.ctx.desc.t_hidden).contiguous()
        tok, src, pos, desc, spans, r = [], [], [], [], [], 0
        for p in probes:
            n0 = 0
            if p["image"] is not None:
                k = imgs.index(p["image"])
                tok += [0] * self.Q
                src += list(range(k * self.Q, (k + 1) * self.Q))
                pos += list(range(self.Q))
                n0 = self.Q
            tok += p["ids"]
            src += [-1] * len(p["ids"])
            pos += list(range(n0, n0 + len(p["ids"])))
            n = n0 + len(p["ids"])
            desc.append([r, n, 0, 0, r, n])
            spans.append((r, n))
            r += n
        t = lambda a: torch.tensor(a, dtype=torch.int32, device=self.dev)
        x = lib.embed_rows(t(tok), t(src), t(pos), self.m.get("language_model.model.decoder.embed_tokens.weight"), rows,
                           self.m.get("language_model.model.decoder.embed_positions.weight"))
        return x, t(desc), max(n for _, n in spans), spans

    def label_row_logits(self, probes):
        """decoder forward of the packed probes; logits of every probe's last-L rows (devqa_llm_forward)"""
        x, desc, max_len, spans = self.pack(probes)
        want, cuts = [], []
        for (s, n), p in zip(spans, probes):
            L = p["y"].numel()
            cuts.append((len(want), L))
            want += list(range(s + n - L, s + n))
        logits = self.ctx.llm_forward(x, desc, len(probes), max_len, True, torch.tensor(want, dtype=torch.int32, device=self.dev))
        return [logits[a:a + L] for a, L in cuts]


def test_one_cycle_through_path_level_calls_only(tiny, in_gold_dir):
    vllm, j, z, rec, mode = tiny
    lib = Host(vllm).lib
    H = Host(vllm)
    ctx, m, dev = H.ctx, H.m, H.dev
    tok = vllm.tokenizer
    d0 = rec[0]
    g4, g5 = j["g4"][0], j["g5_results_sen1"][0][0]
    assert g4["request"] == d0["requests"][0] and g4["weight"] == WNAME
    w_master = m.get(WNAME)
    w_orig = w_master.clone()
    # ---- prepare: 9 pre-edit locality probes (vllm_editor_eval.py:102-113) ----
    loc = [H.probe(d0["locality"][n][0]["prompt"], d0["locality"][n][0]["image"], d0["locality"][n][0]["target"]) for n in LOC]
    before = []
    for p, lg in zip(loc, H.label_row_logits(loc)):
        zero = torch.zeros(lg.shape[0], dtype=torch.int32, device=dev)
        _, pred = lib.token_acc(lg, zero, torch.ones(lg.shape[0], device=dev))
        before.append(pred)
    # ---- edit: the frozen prefix once, then devqa_ft_edit (ft_vl.py:66-158) ----
    req = d0["requests"][0]
    e = H.probe(req["prompt"], req["image"], " " + req["target_new"])          # the leading space FT_VL forces (ft_vl.py:73-75)
    x, desc, max_len, spans = H.pack([e])
    x_chk = x.clone()
    a = ctx.llm_prefix(x, desc, 1, max_len, True)                                # devqa_llm_prefix; x: residual before the edited layer's FFN add
    a_chk = ctx.llm_layers(x_chk, desc, 1, max_len, True, -1, stop_before_fc2=True)     # the same call under its general name
    assert torch.equal(a, a_chk) and torch.equal(x, x_chk)
    L = e["y"].numel()
    rows = [spans[0][1] - L + i for i in range(L) if int(e["m"][i]) != 0]
    ridx = torch.tensor(rows, dtype=torch.int32, device=dev)
    a_rows = lib.gather_rows(a, ridx).to(torch.float32).unsqueeze(0).contiguous()                       # [1, k, ffn]
    resid = (lib.gather_rows(x, ridx) + m.get(WNAME.replace("weight", "bias"))).contiguous()           # [k, d] (+ fc2 bias)
    labels = torch.tensor([int(e["y"][i]) for i in range(L) if int(e["m"][i]) != 0], dtype=torch.int32, device=dev)
    mask = torch.ones((1, len(rows)), dtype=torch.float32, device=dev)
    delta, losses, steps, updates = ctx.ft_edit(w_master.data, a_rows, resid, labels, mask, 25, 1e-3, 0.0, -1.0)
    n = int(steps[0])
    tol = 1e-3 if mode == "fp32" else 1e-2
    got_l, ref_l = losses[0, :n].cpu().numpy(), np.asarray(g4["losses"])
    print(mode, "steps", n, "ref", g4["steps"], "max loss err %.3g" % np.abs(got_l - ref_l[:n]).max())
    assert n == g4["steps"] and int(updates[0]) <= n
    assert (np.abs(got_l - ref_l) / np.maximum(ref_l, 1.0)).max() < tol
    gold = z["g4_delta_0"]
    rel = np.linalg.norm(delta[0].cpu().numpy() - gold) / np.linalg.norm(gold)
    print(mode, "delta rel_l2 %.3g" % rel)
    assert rel < (1e-3 if mode == "fp32" else 4e-2)       # bf16 on the 40x80 tiny matrix: see tests/test_blip2_gpu.py
    # ---- apply, test (12 probes), restore ----
    ctx.bind_edit_target(WNAME)
    ctx.apply_delta(delta[0].contiguous())
    assert float((w_master - w_orig - delta[0]).abs().max()) < 1e-7
    post = [H.probe(req["prompt"], req["image"], req["target_new"])]
    post += [H.probe(d0["generality"][g][0]["prompt"], d0["generality"][g][0]["image"], d0["generality"][g][0]["target"])
             for g in ("text_rephrase", "image_rephrase")]
    post += loc
    refs = [g5["reliability"][0]] + [g5["generality"][g][0] for g in ("text_rephrase", "image_rephrase")] + [g5["locality"][n][0] for n in LOC]
    same = 0
    for i, (p, lg, ref) in enumerate(zip(post, H.label_row_logits(post), refs)):
        lab = p["y"].to(dev, torch.int32) if i < 3 else before[i - 3]             # locality: agreement with the PRE-edit argmax
        acc, pred = lib.token_acc(lg, lab.contiguous(), p["m"].to(dev, torch.float32).contiguous())
        keep = p["m"].bool()
        text = tok.decode(pred.cpu().long()[keep])
        ok = abs(round(float(acc), 4) - ref["acc"]) < 1e-9 and text == ref["predict_after_edit"]
        if i >= 3:
            ok = ok and tok.decode(before[i - 3].cpu().long()[keep]) == ref["predict_before_edit"]
        same += ok
    print(mode, "probes equal to the reference's results.json: %d/12" % same)
    assert same == 12 if mode == "fp32" else same >= 10
    ctx.restore()
    assert torch.equal(w_master, w_orig)                                             # restore is bit exact
    if mode == "bf16":
        assert torch.equal(m.weight_for_gemm(WNAME), w_orig.to(torch.bfloat16))      # ... and the shadow followed


def test_engine_over_the_context_equals_python_schedule(tiny, in_gold_dir):
    """engine.encode_images / decoder_layers / lm_head / the batched FT loop run devqa_vision_encode / devqa_llm_layers /
    devqa_llm_head / devqa_ft_edit; DEVQA_PATH_ABI=0 runs the same kernels ordered from Python.  Bit-identical."""
    from copy import deepcopy
    from devqa_amd.batched import BatchedEditEval
    from devqa_amd.editor.vllm_editors.ft_vl.ft_vl import FTvl, FTvlConfig
    vllm, j, z, rec, mode = tiny
    eng = vllm.engine
    pix = torch.from_numpy(np.stack([vllm.load_pixels(rec[i]["requests"][0]["image"]) for i in range(3)])).cuda()
    cfg = FTvlConfig(edit_model_name="blip2-opt-2.7b", rewrite_module_tmp="language_model.model.decoder.layers.{}.fc2.weight",
                     layers=[1], num_steps=25, lr=1e-3, weight_decay=0, norm_constraint=False, batch_size=1)
    ed = FTvl(vllm, cfg, "cuda:0")

    def run():
        it = eng.encode_images(pix)
        be = BatchedEditEval(ed, cycles_per_batch=3)
        be.keep_debug = True
        res = be.run([[deepcopy(r)] for r in rec[:3]], [[deepcopy(r)] for r in rec[:3]])
        torch.cuda.synchronize()
        return it, be.debug["pre_logits"].clone(), be.debug["post_logits"].clone(), be.debug["delta"][0].clone(), be.last_losses.copy(), res
    assert eng.path_ctx() is not None
    a = run()
    os.environ["DEVQA_PATH_ABI"] = "0"
    try:
        assert eng.path_ctx() is None
        b = run()
    finally:
        del os.environ["DEVQA_PATH_ABI"]
    for k in range(4):
        assert torch.equal(a[k], b[k]), k
    assert np.array_equal(a[4], b[4])
    assert json.dumps(a[5], sort_keys=True, default=str).replace("edit_time", "") != ""     # results exist
    for r1, r2 in zip(a[5], b[5]):
        for sec in ("generality", "locality"):
            for sub in r1[0][sec]:
                assert r1[0][sec][sub][0]["acc"] == r2[0][sec][sub][0]["acc"]


def test_context_follows_reassigned_parameter_storage(gold_dir):
    """The path-level context caches raw device pointers of the weight table.  A parameter whose STORAGE is replaced (an editor
    assigning `p.data = ...`, a reload into fresh buffers) must be seen by the next call: the engine keys the context on the
    model's storage fingerprint and rebuilds it.  Checked against the Python-ordered schedule, which looks parameters up by name
    on every call (DEVQA_PATH_ABI=0)."""
    import devqa_amd  # noqa: F401
    from devqa_amd.editor.vllms_for_edit.blip2.blip2 import BLIP2OPTForEdit
    vllm = BLIP2OPTForEdit(os.path.join(gold_dir, "tiny_blip2"), "cuda:0", dtype="bf16")
    eng, m = vllm.engine, vllm.model
    rec = json.load(open(os.path.join(gold_dir, "evqa8_records.json")))["records"]
    old = os.getcwd()
    os.chdir(gold_dir)
    try:
        pix = torch.from_numpy(np.stack([vllm.load_pixels(rec[0]["requests"][0]["image"])])).cuda()
    finally:
        os.chdir(old)
    seqs = [(0, [2, 17, 33, 21, 9])]

    def logits():
        it = eng.encode_images(pix)
        ps = eng.pack_from_tokens(seqs, it)
        return eng.full_logits(ps).clone()
    base = logits()
    ctx0 = eng.path_ctx()
    assert ctx0 is not None and eng.path_ctx() is ctx0           # unchanged storage: the context is kept
    names = ["language_model.model.decoder.layers.0.fc1.weight", "vision_model.encoder.layers.1.mlp.fc2.weight",
             "language_model.model.decoder.layers.1.final_layer_norm.bias"]
    for n in names:                                              # out-of-place: new storage, new values
        p = m.get(n)
        p.data = (p.data.float() * 1.5 + 0.01).to(p.dtype)
    got = logits()
    assert eng.path_ctx() is not ctx0                            # rebuilt
    os.environ["DEVQA_PATH_ABI"] = "0"
    try:
        want = logits()
    finally:
        del os.environ["DEVQA_PATH_ABI"]
    assert torch.equal(got, want)
    assert not torch.equal(got, base)
    # in-place writes keep the addresses: no rebuild, values still followed
    ctx1 = eng.path_ctx()
    m.get(names[0]).mul_(0.5)
    got2 = logits()
    assert eng.path_ctx() is ctx1 and not torch.equal(got2, got)


def test_path_level_error_behaviour(tiny):
    from devqa_amd import lib
    vllm = tiny[0]
    L = lib.load()
    ctx = vllm.engine.path_ctx()
    # a dead / foreign handle: status code + message, no crash
    assert L.devqa_vision_encode_workspace(ctypes.c_uint64(12345), 1) == -1
    rc = L.devqa_restore(ctypes.c_uint64(12345), None)
    assert rc == -5 and b"context handle" in L.devqa_last_error()                    # DEVQA_E_STATE
    # restore before an edit target was bound
    fresh = lib.PathContext(0, ctx.desc, vllm.model.weight_table())
    rc = L.devqa_restore(ctypes.c_uint64(fresh.h), None)
    assert rc == -5 and b"no edit target bound" in L.devqa_last_error()
    with pytest.raises(lib.DevqaError, match="no entry"):
        fresh.bind_edit_target("no.such.weight")
    # workspace too small
    pix = torch.zeros((1, 3, ctx.desc.image_size, ctx.desc.image_size), device="cuda")
    out = torch.empty((1, ctx.desc.num_query_tokens, ctx.desc.t_hidden), device="cuda")
    ws = torch.empty(4096 + 256, dtype=torch.uint8, device="cuda")
    p = ws.data_ptr() + (-ws.data_ptr()) % 256
    rc = L.devqa_vision_encode(ctypes.c_uint64(fresh.h), pix.data_ptr(), 1, out.data_ptr(), ctypes.c_void_p(p), 4096, None)
    assert rc == -2 and b"too small" in L.devqa_last_error()                         # DEVQA_E_SHAPE
    # a table without a weight the schedule needs
    t = dict(vllm.model.weight_table())
    t.pop("language_projection.weight")
    broken = lib.PathContext(0, ctx.desc, t)
    with pytest.raises(lib.DevqaError, match="language_projection.weight"):
        broken.vision_encode(pix)
    broken.close()
    fresh.close()
    assert L.devqa_ctx_destroy(ctypes.c_uint64(fresh.h or 0)) == -5                  # double destroy is refused, not a crash


def test_gather_scores_single_rank():
    """devqa_comm_* / devqa_gather_scores with a one-rank communicator (the N > 1 launch is the driver's; rank logic: test_dist_cpu)."""
    from devqa_amd import lib
    comm = lib.ScoreComm(0, 1, lib.comm_unique_id(), 0)
    rows = torch.arange(5 * 16, dtype=torch.float32, device="cuda").view(5, 16).contiguous()
    out = comm.gather_scores(rows)
    torch.cuda.synchronize()
    assert torch.equal(out, rows)
    h = comm.h
    comm.close()
    # a destroyed (or never created) communicator handle is an error status, not a dereference
    L = lib.load()
    assert L.devqa_gather_scores(ctypes.c_uint64(h), ctypes.c_void_p(rows.data_ptr()), 5, ctypes.c_void_p(out.data_ptr()), None) == -5
    assert b"communicator" in L.devqa_last_error()
    assert L.devqa_comm_destroy(ctypes.c_uint64(h)) == -5
    assert L.devqa_gather_scores(ctypes.c_uint64(0xdeadbeef), ctypes.c_void_p(rows.data_ptr()), 5, ctypes.c_void_p(out.data_ptr()), None) == -5


def test_mend_transform_and_apply_entry_points():
    """devqa_mend_transform (K16) against the op-level composition it sequences and a float64 restatement of
    GradientTransform / LRLinear (auxiliary_networks.py:112-151, 62-83); devqa_mend_apply (K17) against h @ (xt^T dt)."""
    import devqa_amd  # noqa: F401
    from devqa_amd import lib
    g = torch.Generator().manual_seed(5)
    R, du, dv, rank, nl = 9, 40, 24, 16, 2
    D = du + dv
    x, dl = torch.randn(R, du, generator=g), torch.randn(R, dv, generator=g)
    idx = torch.tensor([0, 2, 3, 7], dtype=torch.int32)
    stats = [torch.randn(du, generator=g), torch.rand(du, generator=g) + 0.5, torch.randn(dv, generator=g), torch.rand(dv, generator=g) + 0.5]
    layers = [{"u": torch.randn(D, rank, generator=g) * 0.2, "v": torch.randn(rank, D, generator=g) * 0.2, "bias": torch.randn(D, generator=g) * 0.1,
               "mode_scale": torch.rand(D, generator=g) + 0.5, "mode_shift": torch.randn(D, generator=g) * 0.1} for _ in range(nl)]
    cu = lambda t: t.cuda().contiguous()  # noqa: E731
    ox, od = lib.mend_transform(cu(x), cu(dl), cu(idx), [{k: cu(v) for k, v in L_.items()} for L_ in layers], [cu(t) for t in stats])
    # float64 restatement
    sel = idx.long()
    inp = torch.cat([(x[sel].double() - stats[0].double()) / (stats[1].double() + 1e-7),
                     (dl[sel].double() - stats[2].double()) / (stats[3].double() + 1e-7)], 1)
    for L_ in layers:
        pre = inp @ L_["v"].double().T @ L_["u"].double().T + L_["bias"].double()
        inp = inp + torch.relu(pre) * L_["mode_scale"].double() + L_["mode_shift"].double()
    # the epilogue's exact formula is the op-level kernel's (tests/test_mend_gpu.py pins it to the reference); here: same composition
    inp32 = lib.mend_normalize_concat(cu(x), cu(dl), cu(idx), *[cu(t) for t in stats], 1e-7)
    for L_ in layers:
        prea = lib.gemm(lib.gemm(inp32, cu(L_["v"])), cu(L_["u"]))
        inp32 = lib.mend_lrlinear_epilogue(prea, cu(L_["bias"]), cu(L_["mode_scale"]), cu(L_["mode_shift"]), inp32)
    assert torch.equal(ox, inp32[:, :du].contiguous()) and torch.equal(od, inp32[:, du:].contiguous())
    # DEVQA_MEND_SPLIT_BF16 (the bf16 compute mode's form: three bf16 MFMA products of split operands per GEMM) against the exact-fp32 form
    sx, sd = lib.mend_transform(cu(x), cu(dl), cu(idx), [{k: cu(v) for k, v in L_.items()} for L_ in layers], [cu(t) for t in stats], split_bf16=True)
    ex_small = torch.cat([ox, od], 1).double()
    assert float((torch.cat([sx, sd], 1).double() - ex_small).abs().max()) < 1e-4 * float(ex_small.abs().max()) and not torch.equal(sx, ox)
    # ... and at the BLIP-2 hyper-network's width (D = 12800, rank 1920), where K is long enough for the dropped lo.lo term to show
    g2 = torch.Generator().manual_seed(6)
    n2, du2, dv2, rank2 = 37, 10240, 2560, 1920
    x2, d2 = torch.randn(n2, du2, generator=g2), torch.randn(n2, dv2, generator=g2)
    lay2 = {"u": torch.randn(du2 + dv2, rank2, generator=g2) * 0.02, "v": torch.randn(rank2, du2 + dv2, generator=g2) * 0.01,
            "bias": torch.randn(du2 + dv2, generator=g2) * 0.1, "mode_scale": torch.rand(du2 + dv2, generator=g2) + 0.5,
            "mode_shift": torch.randn(du2 + dv2, generator=g2) * 0.1}
    ex = torch.cat(lib.mend_transform(cu(x2), cu(d2), None, [{k: cu(v) for k, v in lay2.items()}], None), 1).double()
    sp = torch.cat(lib.mend_transform(cu(x2), cu(d2), None, [{k: cu(v) for k, v in lay2.items()}], None, split_bf16=True), 1).double()
    pre64 = (torch.cat([x2, d2], 1).double() @ lay2["v"].double().T) @ lay2["u"].double().T        # what the two GEMMs compute, in float64
    err = float((sp - ex).abs().max())
    print("mend_transform at D 12800 / rank 1920: max |split bf16 - exact fp32| %.3g (pre-activations up to %.3g, outputs up to %.3g)"
          % (err, float(pre64.abs().max()), float(ex.abs().max())))
    assert err < 1e-4 * float(pre64.abs().max())
    for mode in (torch.float32, torch.bfloat16):
        npad, din, dout, Rr = 64, 48, 40, 13
        h = torch.randn(Rr, din, generator=g).to(mode).cuda()
        xt = torch.zeros(npad, din)
        dt = torch.zeros(npad, dout)
        xt[:5], dt[:5] = torch.randn(5, din, generator=g), torch.randn(5, dout, generator=g)
        y0 = torch.randn(Rr, dout, generator=g).cuda()
        y = y0.clone()
        lib.mend_apply_(h, xt.to(mode).cuda(), dt.t().contiguous().to(mode).cuda(), y)
        ref = y0.double().cpu() + (h.double().cpu() @ xt.to(mode).double().T) @ dt.to(mode).double()
        tol = 1e-5 if mode == torch.float32 else 3e-2
        assert float((y.double().cpu() - ref).abs().max() / ref.abs().max()) < tol
