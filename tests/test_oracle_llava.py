"""CPU: LLaVA oracle against goldens from HF LlavaForConditionalGeneration + the reference's FTvl/evaluator."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import devqa_oracle as O
from oracle.llava_oracle import OracleLlava


@pytest.fixture(scope="module")
def lv(gold_dir):
    m = OracleLlava.from_pretrained_dir(os.path.join(gold_dir, "tiny_llava"))
    j = json.load(open(os.path.join(gold_dir, "tiny_llava_goldens.json")))
    z = np.load(os.path.join(gold_dir, "tiny_llava_goldens.npz"))
    rec = json.load(open(os.path.join(gold_dir, "evqa8_records.json")))
    return m, j, z, rec


def test_llava_forward(lv, in_gold_dir):
    m, j, z, rec = lv
    np.testing.assert_allclose(m.preprocess_image(rec["odd_image"]).numpy(), z["pixel_values_odd"], atol=1e-6)
    for i, g in enumerate(j["g1"]):
        with torch.no_grad():
            (x, vt), y, msk = m.prompts_imgs_target_to_xym([g["prompt"]], [g["image"]], [g["target"]])
            logits = m.get_llm_outpt(x, vt)
        assert vt == g["vt_range"] and list(x["inputs_embeds"].shape) == g["embeds_shape"]
        assert y.tolist() == g["label_ids"] and msk.tolist() == g["label_masks"]
        np.testing.assert_allclose(x["inputs_embeds"].numpy(), z["g2_embeds_%d" % i], atol=2e-5, rtol=1e-4)
        np.testing.assert_allclose(logits.numpy(), z["g3_logits_%d" % i], atol=2e-4, rtol=1e-4)
        assert abs(float(O.label_loss(logits, y, msk)) - g["label_loss"]) < 1e-4


def test_llava_ft_and_eval(lv, in_gold_dir):
    m, j, z, rec = lv
    ed = O.OracleFTvl(m, [1], "language_model.model.layers.1.mlp.down_proj.weight")
    for i, g in enumerate(j["g4"]):
        d = ed.execute_ft([g["request"]])[g["weight"]].numpy()
        assert len(ed.last_losses) == g["steps"]
        np.testing.assert_allclose(ed.last_losses, g["losses"], atol=2e-4, rtol=1e-4)
        np.testing.assert_allclose(d, z["g4_delta_%d" % i], atol=1e-5)
    res, _ = O.evaluate_sequential_edit(m, ed, rec["records"][:4], 1)
    for sr, sg in zip(res, j["g5_results_sen1"]):
        r, g = sr[0], sg[0]
        rr = O.round4(json.loads(json.dumps(r, default=lambda o: None)))
        for sec in ("generality", "locality"):
            for sub in g[sec]:
                a = {k: v for k, v in rr[sec][sub][0].items() if k != "before_edit_ids"}
                assert a == g[sec][sub][0], (sec, sub)
        assert rr["reliability"][0] == g["reliability"][0]
