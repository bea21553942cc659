"""CPU: the MEND_VL oracle restatement against goldens produced by the reference's own MENDvl
(tools/make_goldens_mend.py) -- hooked x / delta, transformed factors, delta weights (single, sequential
running mean, batch), post-edit logits, logit_KL_loss known answers, evaluator results."""
import json
import os
from copy import deepcopy

import numpy as np
import pytest
import torch
import yaml


@pytest.fixture(scope="module")
def mend(gold_dir):
    from oracle.devqa_oracle import OracleBlip2
    from oracle.mend_oracle import OracleMENDvl
    model = OracleBlip2.from_pretrained_dir(os.path.join(gold_dir, "tiny_blip2"))
    cfg = yaml.safe_load(open(os.path.join(gold_dir, "tiny_mend_cfg.yaml")))
    ck = torch.load(os.path.join(gold_dir, "tiny_mend_ckpt.pt"), map_location="cpu", weights_only=True)
    ed = OracleMENDvl(model, cfg, ck["train_modules"])
    j = json.load(open(os.path.join(gold_dir, "tiny_mend_goldens.json")))
    z = np.load(os.path.join(gold_dir, "tiny_mend_goldens.npz"))
    return model, ed, j, z


def _check(ed, z, tag, tol=2e-4):
    for i, m in enumerate(ed.modules):
        got = ed.last[m["name"]]
        for key in ("x", "delta", "xt", "dt", "dw"):
            g = z["%s_%s_%d" % (tag, key, i)]
            a = got[key].detach().numpy()
            assert a.shape == g.shape, (tag, key, i, a.shape, g.shape)
            err = np.abs(a - g).max() / max(np.abs(g).max(), 1e-30)
            assert err < tol, (tag, key, i, err)


def test_mend_oracle_edits(mend, in_gold_dir):
    model, ed, j, z = mend
    pr = j["probe"]

    def logits():
        with torch.no_grad():
            (x, vt), y, m = model.prompts_imgs_target_to_xym([pr["prompt"]], [pr["image"]], [pr["target"]])
            return model.get_llm_outpt(x, vt).numpy()
    ed.restore_to_original_model()
    np.testing.assert_allclose(logits(), z["pre_logits"], atol=2e-4)
    a, b, c = j["cases"]
    ed.edit_one_piece(deepcopy(a["requests"][0]))
    _check(ed, z, "a")
    np.testing.assert_allclose(logits(), z["a_post_logits"], atol=3e-3, rtol=1e-3)
    ed.edit_one_piece(deepcopy(b["requests"][1]))
    _check(ed, z, "b")
    np.testing.assert_allclose(logits(), z["b_post_logits"], atol=3e-3, rtol=1e-3)
    ed.restore_to_original_model()
    ed.edit_batch(deepcopy(c["requests"]))
    _check(ed, z, "c")
    np.testing.assert_allclose(logits(), z["c_post_logits"], atol=3e-3, rtol=1e-3)
    ed.restore_to_original_model()
    np.testing.assert_allclose(logits(), z["restored_logits"], atol=2e-4)


def test_kl_known_answer(mend):
    from oracle.devqa_oracle import logit_KL_loss
    _, _, j, z = mend
    l1, l2, mk = torch.from_numpy(z["kl_l1"]), torch.from_numpy(z["kl_l2"]), torch.from_numpy(z["kl_mask"])
    assert abs(float(logit_KL_loss(l1, l2, mk)) - j["kl"]) < 1e-6
    assert abs(float(logit_KL_loss(l1, l2, mk, average=False)) - j["kl_sum"]) < 1e-5


def test_mend_oracle_evaluator(mend, in_gold_dir, gold_dir):
    from oracle.devqa_oracle import evaluate_sequential_edit
    model, ed, j, z = mend
    rec = json.load(open(os.path.join(gold_dir, "evqa8_records.json")))
    ed.restore_to_original_model()
    res, _ = evaluate_sequential_edit(model, ed, deepcopy(rec["records"][:4]), 1)
    gold = j["results_sen1"]
    n = same = 0
    for rs, gs in zip(res, gold):
        r, g = rs[0], gs[0]
        pairs = [(r["reliability"][0], g["reliability"][0])]
        for sec in ("generality", "locality"):
            for sub in g[sec]:
                pairs.append((r[sec][sub][0], g[sec][sub][0]))
        for a, b in pairs:
            n += 1
            same += int(round(a["acc"], 4) == round(b["acc"], 4) and a["predict_after_edit"] == b["predict_after_edit"])
    assert n == 48 and same == 48, (same, n)
