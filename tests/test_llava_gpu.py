"""GPU: native LLaVA wrapper + FT_VL + evaluator (generic and batched) against goldens from HF
LlavaForConditionalGeneration + the reference's FTvl / VLLMEditorEvaluation (tools/make_goldens_llava.py)."""
import json
import os
from copy import deepcopy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = {"fp32": dict(fwd=1e-3, loss=1e-3, delta=1e-3), "bf16": dict(fwd=2e-2, loss=3e-2, delta=1e-1)}


@pytest.fixture(scope="module", params=["fp32", "bf16"])
def lv(gold_dir, request):
    import devqa_amd  # noqa: F401
    from devqa_amd.editor.vllms_for_edit.llava.llava import LlavaForEdit
    vllm = LlavaForEdit(os.path.join(gold_dir, "tiny_llava"), "cuda:0", True, dtype=request.param)
    vllm.tol = TOL[request.param]
    vllm.strict = request.param == "fp32"
    j = json.load(open(os.path.join(gold_dir, "tiny_llava_goldens.json")))
    z = np.load(os.path.join(gold_dir, "tiny_llava_goldens.npz"))
    rec = json.load(open(os.path.join(gold_dir, "evqa8_records.json")))
    return vllm, j, z, rec


def _editor(vllm):
    from devqa_amd.editor.vllm_editors.ft_vl.ft_vl import FTvl, FTvlConfig
    cfg = FTvlConfig(edit_model_name="llava-v1.5-7b", rewrite_module_tmp="language_model.model.layers.1.mlp.down_proj.weight",
                     layers=[1], num_steps=25, lr=1e-3, weight_decay=0, norm_constraint=False, batch_size=1)
    return FTvl(vllm, cfg, "cuda:0")


def test_llava_forward(lv, in_gold_dir):
    vllm, j, z, rec = lv
    np.testing.assert_allclose(vllm.load_pixels(rec["odd_image"])[None], z["pixel_values_odd"], atol=1e-6)
    for i, g in enumerate(j["g1"]):
        (x, vt), y, m = vllm.prompts_imgs_target_to_xym([g["prompt"]], [g["image"]], [g["target"]])
        logits = vllm.get_llm_outpt(x, vt).logits
        assert vt == g["vt_range"] and list(x["inputs_embeds"].shape) == g["embeds_shape"]
        assert y.tolist() == g["label_ids"] and m.tolist() == g["label_masks"]
        gl = z["g3_logits_%d" % i]
        err = np.abs(logits.cpu().numpy() - gl).max() / np.abs(gl).max()
        e2 = np.abs(x["inputs_embeds"].cpu().numpy() - z["g2_embeds_%d" % i]).max() / np.abs(z["g2_embeds_%d" % i]).max()
        print(i, "logits rel err %.3g embeds %.3g" % (err, e2))
        assert err < vllm.tol["fwd"] and e2 < vllm.tol["fwd"]


def test_llava_ft(lv, in_gold_dir):
    vllm, j, z, rec = lv
    ed = _editor(vllm)
    for i, g in enumerate(j["g4"]):
        d = ed.execute_ft([g["request"]])[g["weight"]].cpu().numpy()
        assert len(ed.last_losses) == g["steps"]
        np.testing.assert_allclose(ed.last_losses, g["losses"], rtol=vllm.tol["loss"], atol=vllm.tol["loss"])
        gold = z["g4_delta_%d" % i]
        rel = np.linalg.norm(d - gold) / np.linalg.norm(gold)
        print(i, "delta rel_l2 %.3g" % rel)
        assert rel < vllm.tol["delta"]


def test_llava_evaluator_generic_and_batched(lv, in_gold_dir, tmp_path):
    from devqa_amd.dataset.vllm import BaseVLLMEditData
    from devqa_amd.evaluation.vllm_editor_eval import VLLMEditorEvaluation
    vllm, j, z, rec = lv

    class Data(BaseVLLMEditData):
        def dataset_name(self):
            return "EVQA"
    ed = _editor(vllm)
    gold = j["g5_results_sen1"]

    def flat(results):
        out = []
        for split in results:
            r = split[0]
            out.append(("rel", round(r["reliability"][0]["acc"], 4), r["reliability"][0]["predict_after_edit"], None))
            for sec in ("generality", "locality"):
                for sub in r[sec]:
                    it = r[sec][sub][0]
                    out.append((sub, round(it["acc"], 4), it["predict_after_edit"], it.get("predict_before_edit")))
        return out
    fg = flat(gold)
    for batched in (False, True):
        data = Data(deepcopy(rec["records"][:4]), deepcopy(rec["records"][:4]))
        res = VLLMEditorEvaluation(ed, data, "EVQA", str(tmp_path / str(batched))).evaluate_sequential_edit(1, False, None, batched=batched)
        fr = flat(res)
        same = sum(a == b for a, b in zip(fr, fg))
        print("batched" if batched else "generic", "== golden %d/%d" % (same, len(fg)))
        assert len(fr) == len(fg) == 48
        if vllm.strict:
            assert same == 48
        else:
            assert same >= 36
