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


def _organise(model, d):
    e = model.prompts_imgs_target_to_xym([d["requests"][0]["prompt"]], [d["requests"][0]["image"]], [d["requests"][0]["target_new"]])
    g = {k: model.prompts_imgs_target_to_xym([d["generality"][k][0]["prompt"]], [d["generality"][k][0]["image"]],
                                             [d["generality"][k][0]["target"]]) for k in d["generality"]}
    l = {k: model.prompts_imgs_target_to_xym([d["locality"][k][0]["prompt"]], [d["locality"][k][0]["image"]],
                                             [d["locality"][k][0]["target"]]) for k in d["locality"]}
    return e, g, l


def test_mend_oracle_training_steps(gold_dir, in_gold_dir):
    """Two train_a_batch steps against the reference's MENDvl.train_a_batch (G6b): losses, log dict, clipped
    gradients, hyper-network parameters / normalisation buffers / edit lrs after each Adam step."""
    from oracle.devqa_oracle import OracleBlip2
    from oracle.mend_oracle import OracleMENDvl
    model = OracleBlip2.from_pretrained_dir(os.path.join(gold_dir, "tiny_blip2"))
    cfg = yaml.safe_load(open(os.path.join(gold_dir, "tiny_mend_cfg.yaml")))
    ck = torch.load(os.path.join(gold_dir, "tiny_mend_ckpt.pt"), map_location="cpu", weights_only=True)
    ed = OracleMENDvl(model, cfg, ck["train_modules"])
    j = json.load(open(os.path.join(gold_dir, "tiny_mend_train_goldens.json")))
    z = np.load(os.path.join(gold_dir, "tiny_mend_train_goldens.npz"))
    rec = json.load(open(os.path.join(gold_dir, "evqa8_records.json")))["records"]
    ed.set_train(j["aux_lr"], j["edit_lr_lr"])
    for si, g in enumerate(j["steps"]):
        with torch.no_grad():
            batch = _organise(model, deepcopy(rec[g["sample"]]))
        loss, log = ed.train_a_batch(batch)
        assert abs(loss - g["loss"]) < 2e-4 * abs(g["loss"]), (loss, g["loss"])
        assert abs(log["Grad-Norm"] - g["log"]["Grad-Norm"]) < 2e-3 * g["log"]["Grad-Norm"]
        for k, v in g["log"]["Locality loss"].items():
            assert abs(log["Locality loss"][k] - v) < 2e-4 * max(abs(v), 1e-3)
        for n, gr in ed.last_grads.items():
            gold = z["s%d_grad_%s" % (si, n)]
            err = np.abs(gr.numpy() - gold).max() / max(np.abs(gold).max(), 1e-30)
            assert err < 2e-3, (si, n, err)
        for n, p in ed.train_params.items():
            gold = z["s%d_state_%s" % (si, n)]
            assert np.abs(p.detach().numpy() - gold).max() < 2e-5 + 1e-4 * np.abs(gold).max(), (si, n)
        for key in ("(40, 80)", "(80, 40)"):
            for leaf in ("u_mean", "u_std", "v_mean", "v_std", "k"):
                gold = z["s%d_state_aux_models.%s.%s" % (si, key, leaf)]
                got = ed.aux["%s.%s" % (key, leaf)].numpy()
                assert np.abs(got - gold).max() < 1e-4 * max(np.abs(gold).max(), 1e-6) + 1e-7, (si, key, leaf)
