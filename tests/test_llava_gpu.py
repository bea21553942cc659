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


# ---- MEND_VL on the LLaMA decoder (gate / up / down projections of both tiny layers) -------------------------------------
# The reference's MENDvl cannot be driven on LLaVA here (LlavaForEdit does not run on the installed transformers): PARITY
# UNPINNED by the reference for this combination.  The checker is the MEND oracle (pinned on BLIP-2 by the reference's own
# goldens) over the LLaVA oracle (pinned by HF goldens), on a seeded hyper-network state.
def test_llava_mend_vs_oracle(lv, gold_dir, in_gold_dir):
    from devqa_amd.editor.vllm_editors.mend_vl.mend_vl import MENDvl, MENDvlConfig
    from devqa_amd.synth import mend_aux_init
    from oracle.llava_oracle import OracleLlava
    from oracle.mend_oracle import OracleMENDvl
    vllm, j, z, rec = lv
    mods = ["language_model.model.layers.%d.mlp.%s" % (l, k) for l in (0, 1) for k in ("gate_proj", "up_proj", "down_proj")]
    aux = dict(n_hidden=1, hidden_dim=None, init="id", norm=True, act="relu", rank=16, shared=True, lr=1e-6)
    cfg = MENDvlConfig(edit_modules=mods, init_edit_lr=1e-2, edit_lr_lr=1e-4, aux_model=MENDvlConfig.AuxModelConfig(**aux),
                       edit_model_name="llava-v1.5-7b", relia_lambda=0.1, gen_lambda=0.1, loc_lambda=0.1)
    d, F = 64, 96
    tm = {"aux_models": {}, "edit_lrs": {str(i): torch.tensor(float(mend_aux_init("edit_lrs.%d" % i, (), 7))) for i in range(6)}}
    for (du, dv), n_modes in (((d, F), 4), ((F, d), 2)):
        key, D = str((du, dv)), du + dv
        shapes = {"u_mean": (du,), "u_std": (du,), "v_mean": (dv,), "v_std": (dv,), "u_s": (du,), "v_s": (dv,), "k": (1,)}
        for l in range(2):
            shapes.update({"mlp.layers.%d.u" % l: (D, 16), "mlp.layers.%d.v" % l: (16, D), "mlp.layers.%d.bias" % l: (D,),
                           "mlp.layers.%d.mode_shift.weight" % l: (n_modes, D), "mlp.layers.%d.mode_scale.weight" % l: (n_modes, D)})
        for leaf, shp in shapes.items():
            tm["aux_models"]["%s.%s" % (key, leaf)] = torch.from_numpy(mend_aux_init("aux_models.%s.%s" % (key, leaf), shp, 7))
    ed = MENDvl(vllm, cfg, "cuda:0", train_modules=tm)
    orc = OracleLlava.from_pretrained_dir(os.path.join(gold_dir, "tiny_llava"))
    oed = OracleMENDvl(orc, dict(edit_modules=mods, aux_model=aux), tm)
    assert [m["name"] for m in ed.modules] == [m["name"] for m in oed.modules]
    pr = rec["records"][2]["generality"]["text_rephrase"][0]

    def logits():
        (x, vt), y, m = vllm.prompts_imgs_target_to_xym([pr["prompt"]], [pr["image"]], [pr["target"]])
        with torch.no_grad():
            (ox, ovt), _, _ = orc.prompts_imgs_target_to_xym([pr["prompt"]], [pr["image"]], [pr["target"]])
            return vllm.get_llm_outpt(x, vt).logits.float().cpu().numpy(), orc.get_llm_outpt(ox, ovt).numpy()
    tol = dict(fac=1e-3, dw=1e-3, lg=1e-3) if vllm.strict else dict(fac=1e-1, dw=1.5e-1, lg=5e-2)
    for step, r in enumerate((rec["records"][0]["requests"][0], rec["records"][1]["requests"][0])):   # 2nd edit: running mean + delta branch
        ed.edit_one_piece(deepcopy(r))
        oed.edit_one_piece(deepcopy(r))
        worst = {}
        for m in ed.modules:
            got, ref = ed.last[m["name"]], oed.last[m["name"]]
            assert got["xt"].shape == tuple(ref["xt"].shape), (m["name"], got["xt"].shape, ref["xt"].shape)
            for key in ("xt", "dt"):
                e = float((got[key].cpu() - ref[key]).abs().max() / ref[key].abs().max())
                worst[key] = max(worst.get(key, 0), e)
            dw = ed.delta_weight(m["name"]).cpu()
            worst["dw"] = max(worst.get("dw", 0), float((dw - ref["dw"].detach()).abs().max() / ref["dw"].abs().max()))
        a, b = logits()
        e = float(np.abs(a - b).max() / np.abs(b).max())
        print("llava mend edit %d" % step, {k: "%.2e" % v for k, v in worst.items()}, "post-edit logits %.2e" % e)
        assert worst["xt"] < tol["fac"] and worst["dt"] < tol["fac"] and worst["dw"] < tol["dw"] and e < tol["lg"]
    ed.restore_to_original_model()
    oed.restore_to_original_model()
    a, b = logits()
    assert float(np.abs(a - b).max() / np.abs(b).max()) < tol["lg"]
