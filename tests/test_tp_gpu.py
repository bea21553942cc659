"""GPU: TP_VL (T-Patcher) on the HIP engine against goldens produced by the reference's own TPvl
(tools/make_goldens_tp.py): patch neurons after one / two sequential edits, post-edit and restored logits, evaluator -- the OPT
FFN on the tiny BLIP-2 and the gated LLaMA FFN (gate_proj + up_proj -> down_proj) on the tiny LLaVA."""
import json
import os
from copy import deepcopy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class Draws:
    def __init__(self, seq):
        self.seq, self.i = list(seq), 0

    def choice(self, n, k):
        v = self.seq[self.i]
        self.i += 1
        return np.array([v])


@pytest.fixture(scope="module", params=["fp32", "bf16", "llava-fp32", "llava-bf16"])
def tp(gold_dir, request):
    import devqa_amd  # noqa: F401
    from devqa_amd.editor.vllm_editors.tp_vl.tp_vl import TPvl, TPvlConfig
    mode = request.param.split("-")[-1]
    if request.param.startswith("llava"):
        from devqa_amd.editor.vllms_for_edit.llava.llava import LlavaForEdit
        tag = "tiny_tp_llava"
        vllm = LlavaForEdit(os.path.join(gold_dir, "tiny_llava"), "cuda:0", True, dtype=mode)
    else:
        from devqa_amd.editor.vllms_for_edit.blip2.blip2 import BLIP2OPTForEdit
        tag = "tiny_tp"
        vllm = BLIP2OPTForEdit(os.path.join(gold_dir, "tiny_blip2"), "cuda:0", dtype=mode)
    j = json.load(open(os.path.join(gold_dir, tag + "_goldens.json")))
    z = np.load(os.path.join(gold_dir, tag + "_goldens.npz"))
    ed = TPvl(vllm, TPvlConfig.from_yaml(os.path.join(gold_dir, tag + "_cfg.yaml")), "cuda:0", locality_texts=j["sentences"],
              rng=Draws(j["draws_edits"]))
    assert list(ed.locality_data) == j["sentences"]
    return vllm, ed, j, z, mode


def _rel(a, g):
    return float(np.abs(a - g).max() / max(np.abs(g).max(), 1e-30))


def test_tp_edits(tp, in_gold_dir):
    vllm, ed, j, z, mode = tp
    pr = j["probe"]

    def logits():
        (x, vt), y, m = vllm.prompts_imgs_target_to_xym([pr["prompt"]], [pr["image"]], [pr["target"]])
        return vllm.get_llm_outpt(x, vt).logits.float().cpu().numpy()
    ltol = 1e-3 if mode == "fp32" else 6e-2
    assert _rel(logits(), z["pre_logits"]) < ltol
    for tag, r in zip("ab", j["requests"]):
        ed.edit_one_piece(deepcopy(r))
        if ed.gated:
            got = {"kg": ed.K[:, 0].t().cpu().numpy(), "bg": ed.B[:, 0].cpu().numpy(), "ku": ed.K[:, 1].t().cpu().numpy(),
                   "bu": ed.B[:, 1].cpu().numpy(), "v": ed.V.cpu().numpy()}
        else:
            got = {"k": ed.K.t().cpu().numpy(), "b": ed.B.cpu().numpy(), "v": ed.V.cpu().numpy()}
        errs = {key: _rel(got[key], z["%s_%s" % (tag, key)]) for key in got}
        e = _rel(logits(), z[tag + "_post_logits"])
        print(mode, tag, {k_: "%.2e" % v_ for k_, v_ in errs.items()}, "post-edit logits %.2e" % e, "losses[0], [-1]:", ed.last_losses[0], ed.last_losses[-1])
        for key in got:
            assert got[key].shape == z["%s_%s" % (tag, key)].shape
        if mode == "fp32":      # Adam normalises every coordinate's step to ~lr: element-wise agreement is an fp32 property
            assert max(errs.values()) < 5e-3
        else:                   # bf16: direction of the neuron + its effect on the logits
            for key in (("kg", "ku", "v") if ed.gated else ("k", "v")):
                g = z["%s_%s" % (tag, key)]
                cos = float((got[key] * g).sum() / (np.linalg.norm(got[key]) * np.linalg.norm(g)))
                assert cos > 0.9, (tag, key, cos)
        assert e < ltol * 2
    ed.restore_to_original_model()
    assert _rel(logits(), z["restored_logits"]) < ltol


def test_tp_evaluator(tp, in_gold_dir, gold_dir, tmp_path):
    from devqa_amd.dataset.vllm import BaseVLLMEditData
    from devqa_amd.evaluation.vllm_editor_eval import VLLMEditorEvaluation
    vllm, ed, j, z, mode = tp
    rec = json.load(open(os.path.join(gold_dir, "evqa8_records.json")))

    class Data(BaseVLLMEditData):
        def dataset_name(self):
            return "EVQA"
    ed.restore_to_original_model()
    ed.rng = Draws(j["draws_eval"])
    data = Data(deepcopy(rec["records"][:3]), deepcopy(rec["records"][:3]))
    res = VLLMEditorEvaluation(ed, data, "EVQA", str(tmp_path)).evaluate_sequential_edit(1, False, None)
    n = same = 0
    for rs, gs in zip(res, j["results_sen1"]):
        r, g = rs[0], gs[0]
        pairs = [(r["reliability"][0], g["reliability"][0])]
        for sec in ("generality", "locality"):
            for sub in g[sec]:
                pairs.append((r[sec][sub][0], g[sec][sub][0]))
        for a, b in pairs:
            n += 1
            same += int(round(a["acc"], 4) == round(b["acc"], 4) and a["predict_after_edit"] == b["predict_after_edit"])
    print(mode, "evaluator == golden %d/%d" % (same, n))
    assert n == 36
    assert same == 36 if mode == "fp32" else same >= 27
