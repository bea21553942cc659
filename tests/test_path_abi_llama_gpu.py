"""GPU: the path-level C ABI for the LLaMA-family models (include/devqa.h: DEVQA_FAMILY_LLAVA, DEVQA_FAMILY_MINIGPT4) --
devqa_vision_encode (CLIP ViT + projector / EVA ViT-g + Q-Former + llama_proj), devqa_llm_layers_ex (RMSNorm, rotary q / k, causal
attention, SwiGLU; positions + first_layer), devqa_llm_head, devqa_llm_forward_ex, devqa_ft_edit.

1. The engines, which run through these entry points, equal the Python-ordered op-level schedule (DEVQA_PATH_ABI=0) BIT FOR BIT on image
   encoding, a full forward and a batched FT_VL evaluation; the goldens of tests/test_llava_gpu.py / test_minigpt4_gpu.py (HF LLaVA, the
   reference's own eva_vit.py / Qformer.py, the oracles) are then met THROUGH the path-level calls.
2. One LLaVA forward driven by path-level calls alone (what a non-Python host writes) against the HF-LLaVA golden logits.
3. Error behaviour: a LLaMA-family context refuses a call without rotary positions."""
import ctypes
import json
import os
from copy import deepcopy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
SEED = 5


def _llava(gold_dir, mode):
    import devqa_amd  # noqa: F401
    from devqa_amd.editor.vllms_for_edit.llava.llava import LlavaForEdit
    return LlavaForEdit(os.path.join(gold_dir, "tiny_llava"), "cuda:0", True, dtype=mode)


def _minigpt4(gold_dir, mode):
    import devqa_amd  # noqa: F401
    from transformers import AutoTokenizer
    from devqa_amd import minigpt4_spec as S
    from devqa_amd.editor.vllms_for_edit.minigpt4.minigpt4 import MiniGPT4ForEdit
    from devqa_amd.editor.vllms_for_edit.minigpt4.modeling import MiniGPT4Native
    model = MiniGPT4Native.from_synth(S.TINY_MINIGPT4, SEED, "unit", "cuda:0", mode)
    tok = AutoTokenizer.from_pretrained(os.path.join(gold_dir, "tiny_llava"))
    return MiniGPT4ForEdit(None, "cuda:0", True, model=model, tokenizer=tok, dtype=mode)


def _ft_editor(vllm, name):
    from devqa_amd.editor.vllm_editors.ft_vl.ft_vl import FTvl, FTvlConfig
    cfg = FTvlConfig(edit_model_name=name, rewrite_module_tmp=vllm.engine.edit_target(), layers=[1], num_steps=25, lr=1e-3, weight_decay=0,
                     norm_constraint=False, batch_size=1)
    return FTvl(vllm, cfg, "cuda:0")


@pytest.mark.parametrize("family", ["llava", "minigpt4"])
@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_engine_over_the_context_equals_python_schedule(gold_dir, in_gold_dir, family, mode):
    from devqa_amd.batched import BatchedEditEval
    vllm = (_llava if family == "llava" else _minigpt4)(gold_dir, mode)
    eng = vllm.engine
    rec = json.load(open(os.path.join(gold_dir, "evqa8_records.json")))["records"]
    pix = torch.from_numpy(np.stack([vllm.load_pixels(rec[i]["requests"][0]["image"]) for i in range(3)])).cuda()
    ed = _ft_editor(vllm, "llava-v1.5-7b" if family == "llava" else "minigpt-4-vicuna-7b")

    def run():
        it = eng.encode_images(pix)
        r0 = rec[0]["requests"][0]
        (x, vt), y, m = vllm.prompts_imgs_target_to_xym([r0["prompt"]], [r0["image"]], [r0["target_new"]])
        full = vllm.get_llm_outpt(x, vt).logits.clone()
        be = BatchedEditEval(ed, cycles_per_batch=3)
        be.keep_debug = True
        res = be.run([[deepcopy(r)] for r in rec[:3]], [[deepcopy(r)] for r in rec[:3]])
        torch.cuda.synchronize()
        return it.clone(), full, be.debug["pre_logits"].clone(), be.debug["post_logits"].clone(), be.debug["delta"][0].clone(), be.last_losses.copy(), res
    ctx = eng.path_ctx()
    assert ctx is not None and ctx.desc.family == (2 if family == "llava" else 3)
    a = run()
    os.environ["DEVQA_PATH_ABI"] = "0"
    try:
        assert eng.path_ctx() is None
        b = run()
    finally:
        del os.environ["DEVQA_PATH_ABI"]
    for k in range(5):
        assert torch.equal(a[k], b[k]), (family, mode, k, float((a[k].float() - b[k].float()).abs().max()))
    assert np.array_equal(a[5], b[5])
    for r1, r2 in zip(a[6], b[6]):
        for sec in ("generality", "locality"):
            for sub in r1[0][sec]:
                assert r1[0][sec][sub][0]["acc"] == r2[0][sec][sub][0]["acc"]


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_llava_forward_through_path_level_calls_only(gold_dir, in_gold_dir, mode):
    """devqa_vision_encode -> host-side splice at the <image> token (op-level embed_rows) -> devqa_llm_forward_ex, against the logits of
    HF LlavaForConditionalGeneration (tiny_llava_goldens g3)."""
    from devqa_amd import lib
    vllm = _llava(gold_dir, mode)
    ctx = vllm.engine.path_ctx()
    j = json.load(open(os.path.join(gold_dir, "tiny_llava_goldens.json")))
    z = np.load(os.path.join(gold_dir, "tiny_llava_goldens.npz"))
    dev = torch.device("cuda:0")
    tol = 2e-5 if mode == "fp32" else 3e-2
    n_img = vllm.get_img_token_n()
    checked = 0
    for i, g in enumerate(j["g1"]):
        if g["image"] is None:
            continue
        strs, y, m, _ = vllm.xym_token_bookkeeping([g["prompt"]], [g["target"]])
        ids = vllm.batched_token_ids(strs[0], True)
        p = ids.index(vllm.get_img_special_token_id())
        pix = torch.from_numpy(vllm.load_pixels(g["image"])[None]).to(dev)
        feats = ctx.vision_encode(pix).reshape(-1, ctx.desc.t_hidden).contiguous()            # [n_img, d]
        tok = ids[:p] + [0] * n_img + ids[p + 1:]
        src = [-1] * p + list(range(n_img)) + [-1] * (len(ids) - p - 1)
        R = len(tok)
        emb = vllm.engine.embed_table()
        x = lib.embed_rows(lib.h2d(tok, torch.int32, dev), lib.h2d(src, torch.int32, dev), torch.full((R,), -2, dtype=torch.int32, device=dev), emb,
                           feats, torch.zeros((1, emb.shape[1]), dtype=emb.dtype, device=dev))
        pos = torch.arange(R, dtype=torch.int32, device=dev)
        desc = lib.h2d([[0, R, 0, 0, 0, R]], torch.int32, dev)
        want = torch.arange(R, dtype=torch.int32, device=dev)
        logits = ctx.llm_forward(x, desc, 1, R, True, want, positions=pos)
        gl = z["g3_logits_%d" % i][0]
        err = float(np.abs(logits.cpu().numpy() - gl).max() / np.abs(gl).max())
        print(mode, i, "path-level LLaVA forward vs HF logits: rel err %.3g" % err)
        assert logits.shape == gl.shape and err < tol
        checked += 1
    assert checked >= 2


def test_llama_context_needs_positions(gold_dir, in_gold_dir):
    from devqa_amd import lib
    vllm = _llava(gold_dir, "bf16")
    ctx = vllm.engine.path_ctx()
    L = lib.load()
    x = torch.zeros((4, ctx.desc.t_hidden), dtype=torch.float32, device="cuda")
    desc = torch.tensor([[0, 4, 0, 0, 0, 4]], dtype=torch.int32, device="cuda")
    n = L.devqa_llm_layers_workspace(ctx.h, 4, 0)
    ws = torch.empty((n + 256,), dtype=torch.uint8, device="cuda")
    wp = ctypes.c_void_p(ws.data_ptr() + (-ws.data_ptr()) % 256)
    rc = L.devqa_llm_layers(ctx.h, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(desc.data_ptr()), 1, 4, 4, 1, -1, 0, None, wp, n, None)
    assert rc == -1 and b"rotary positions" in L.devqa_last_error()
    # first_layer / n_layers outside the model are refused
    pos = torch.arange(4, dtype=torch.int32, device="cuda")
    rc = L.devqa_llm_layers_ex(ctx.h, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(pos.data_ptr()), ctypes.c_void_p(desc.data_ptr()), 1, 4, 4, 1,
                               ctx.desc.t_layers, 1, 0, None, wp, n, None)
    assert rc == -2


def test_fused_swiglu_operand_follows_an_edited_gate_proj(gold_dir, in_gold_dir):
    """The interleaved [gate | up] copy the fused-SwiGLU GEMM reads (blip2/modeling.py: gu_interleaved) is a DERIVED buffer: when an editor writes
    a gate_proj / up_proj fp32 master through raw pointers (FT_VL's general form, devqa_adam_step) the bf16 row block of the fused operand is
    refreshed by a HIP cast, which torch's version counter does not see -- the copy must follow all the same.  Forced onto the 256 x 256 kernel
    (gemm mode 22) so that a tiny decoder takes the fused epilogue; checked on the buffer itself and on the logits against the two-pass form."""
    import devqa_amd  # noqa: F401
    from transformers import AutoTokenizer
    from devqa_amd import lib, minigpt4_spec as S
    from devqa_amd.editor.vllms_for_edit.minigpt4.minigpt4 import MiniGPT4ForEdit
    from devqa_amd.editor.vllms_for_edit.minigpt4.modeling import MiniGPT4Native
    spec = deepcopy(S.TINY_MINIGPT4)
    spec["text_config"]["intermediate_size"] = 128          # [gate | up] = 256 rows: one column tile of the fused form
    model = MiniGPT4Native.from_synth(spec, SEED, "unit", "cuda:0", "bf16")
    tok = AutoTokenizer.from_pretrained(os.path.join(gold_dir, "tiny_llava"))
    vllm = MiniGPT4ForEdit(None, "cuda:0", True, model=model, tokenizer=tok, dtype="bf16")
    eng = vllm.engine
    rec = json.load(open(os.path.join(gold_dir, "evqa8_records.json")))["records"]
    reqs = [rec[i]["requests"][0] for i in range(6)]
    (x, vt), _, _ = vllm.prompts_imgs_target_to_xym([r["prompt"] for r in reqs], [r["image"] for r in reqs], [r["target_new"] for r in reqs])
    rows = int(x["attention_mask"].sum())
    lib.gemm_set_mode(22)
    try:
        assert eng._fuse_swiglu() and lib.gemm_swiglu_supported(rows, 256, 64), rows

        def logits(fused=True):
            if not fused:
                os.environ["DEVQA_SWIGLU_FUSED"] = "0"
            try:
                out = vllm.get_llm_outpt(x, vt).logits.float().clone()
            finally:
                os.environ.pop("DEVQA_SWIGLU_FUSED", None)
            msk = x["attention_mask"].bool()
            return out[msk]
        a0, b0 = logits(), logits(False)
        name = "llama_model.model.layers.0.mlp.gate_proj.weight"
        master = model.promote_to_fp32(name)
        master.data.mul_(-1.0)                  # as a kernel would: in place behind torch's version counter ...
        model.mark_dirty(name)                  # ... and announced the way FTvl announces it
        a1, b1 = logits(), logits(False)
        torch.cuda.synchronize()
        assert torch.equal(model.gu_interleaved()[0], lib.interleave_gate_up(model.fused_w["llama_gu.0"]))
        gate = model.fused_w["llama_gu.0"][:128].float()
        assert torch.equal(gate, master.data.to(torch.bfloat16).float())
        tol = 4.0 * float((a0 - b0).abs().max()) + 1e-3          # fused vs two-pass: one bf16 rounding of gate and up apart
        moved = float((b1 - b0).abs().max())
        assert moved > 10 * tol, (moved, tol)                        # the edit is visible ...
        assert float((a1 - b1).abs().max()) <= tol, (float((a1 - b1).abs().max()), tol, moved)     # ... and the fused path sees it too
    finally:
        lib.gemm_set_mode(0)
