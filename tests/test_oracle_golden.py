"""Pin the CPU oracle (oracle/devqa_oracle.py) against golden vectors produced by the
reference itself (tools/make_goldens.py, build container only)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import devqa_oracle as O

TOL = 2e-5  # fp32 CPU vs fp32 CPU, different summation orders


@pytest.fixture(scope="module")
def tiny(gold_dir):
    m = O.OracleBlip2.from_pretrained_dir(os.path.join(gold_dir, "tiny_blip2"))
    j = json.load(open(os.path.join(gold_dir, "tiny_goldens.json")))
    z = np.load(os.path.join(gold_dir, "tiny_goldens.npz"))
    rec = json.load(open(os.path.join(gold_dir, "evqa8_records.json")))
    return m, j, z, rec


def test_g1_xym_and_logits(tiny, in_gold_dir):
    m, j, z, _ = tiny
    for i, g in enumerate(j["g1"]):
        with torch.no_grad():
            (x, vt), y, msk = m.prompts_imgs_target_to_xym([g["prompt"]], [g["image"]], [g["target"]])
            logits = m.get_llm_outpt(x, vt)
        assert vt == g["vt_range"]
        assert list(x["inputs_embeds"].shape) == g["embeds_shape"]
        assert y.tolist() == g["label_ids"] and msk.tolist() == g["label_masks"]
        assert x["attention_mask"].tolist() == g["attention_mask"]
        np.testing.assert_allclose(x["inputs_embeds"].numpy(), z["g2_embeds_%d" % i], atol=TOL, rtol=1e-4)
        np.testing.assert_allclose(logits.numpy(), z["g3_logits_%d" % i], atol=2e-4, rtol=1e-4)
        assert abs(float(O.label_loss(logits, y, msk)) - g["label_loss"]) < 1e-4
        assert abs(g["label_loss"] - g["label_loss_vllm"]) < 1e-6


def test_g1_batch_padding(tiny):
    m, j, _, _ = tiny
    g = j["g1_batch"]
    with torch.no_grad():
        (x, vt), y, msk = m.prompts_imgs_target_to_xym(g["prompts"], [None, None], g["targets"])
        logits = m.get_llm_outpt(x, vt)
    assert y.tolist() == g["label_ids"] and msk.tolist() == g["label_masks"]
    assert x["attention_mask"].tolist() == g["attention_mask"]
    assert abs(float(O.label_loss(logits, y, msk)) - g["label_loss"]) < 1e-4
    assert abs(float(O.logit_KL_loss(logits, logits * 0.5, msk)) - g["kl_self"]) < 1e-4


def test_pixel_values_resize(tiny, in_gold_dir):
    m, j, z, rec = tiny
    pv = m.preprocess_image(rec["odd_image"])
    np.testing.assert_allclose(pv.numpy(), z["pixel_values_odd"], atol=1e-6)


def _ft(m, cfg=None):
    kw = dict(layers=[1], rewrite_module_tmp="language_model.model.decoder.layers.{}.fc2.weight",
              num_steps=25, lr=1e-3, weight_decay=0, norm_constraint=False, batch_size=1)
    kw.update(cfg or {})
    return O.OracleFTvl(m, **kw)


def test_g4_ft_losses_and_delta(tiny, in_gold_dir):
    m, j, z, _ = tiny
    ed = _ft(m)
    for i, g in enumerate(j["g4"]):
        deltas = ed.execute_ft([g["request"]])
        d = deltas[g["weight"]].numpy()
        assert len(ed.last_losses) == g["steps"]
        np.testing.assert_allclose(ed.last_losses, g["losses"], atol=2e-4, rtol=1e-4)
        np.testing.assert_allclose(d, z["g4_delta_%d" % i], atol=1e-5)
        # invariant: model restored
        assert torch.equal(m.w[g["weight"]], ed.original_w[g["weight"]])


def test_g4b_ft_variants(tiny, in_gold_dir):
    m, j, z, _ = tiny
    per = {}
    for g in j["g4b"]:
        key = json.dumps(g["cfg"], sort_keys=True)
        vi = list(per.keys()).index(key) if key in per else len(per)
        ri = per.setdefault(key, 0)
        per[key] += 1
        ed = _ft(m, g["cfg"])
        d = ed.execute_ft([g["request"]])["language_model.model.decoder.layers.1.fc2.weight"].numpy()
        assert len(ed.last_losses) == g["steps"], (g["cfg"], ed.last_losses)
        np.testing.assert_allclose(ed.last_losses, g["losses"], atol=5e-4, rtol=2e-3)
        np.testing.assert_allclose(d, z["g4b_delta_%d_%d" % (vi, ri)], atol=2e-5)


@pytest.mark.parametrize("edit_n", [1, 3])
def test_g5_evaluator(tiny, in_gold_dir, edit_n):
    m, j, _, rec = tiny
    ed = _ft(m)
    res, ns = O.evaluate_sequential_edit(m, ed, rec["records"], edit_n)
    gold = j["g5_results_sen%d" % edit_n]
    assert len(res) == len(gold)  # incomplete tail split dropped
    for sr, sg in zip(res, gold):
        for r, g in zip(sr, sg):
            rr = O.round4(json.loads(json.dumps(r, default=lambda o: None)))
            for sec in ("generality", "locality"):
                for sub in g[sec]:
                    for a, b in zip(rr[sec][sub], g[sec][sub]):
                        b2 = dict(b)
                        a2 = {k: v for k, v in a.items() if k != "before_edit_ids"}
                        assert a2 == b2, (sec, sub, a2, b2)
            for a, b in zip(rr["reliability"], g["reliability"]):
                assert a == b
    mean = O.get_mean_results([r for sr in res for r in sr])
    gm = j["g5_mean_sen%d" % edit_n]["total_mean"]
    mean = O.round4(mean)
    for sec in ("generality", "locality"):
        assert mean[sec] == gm[sec]
    assert abs(mean["reliability"]["acc"] - gm["reliability"]["acc"]) < 1e-9
    assert sum(ns) == gm["total_edit_n"]


def test_cosine_topk_kat():
    rng = np.random.default_rng(3)
    c = rng.standard_normal((500, 384)).astype(np.float32)
    q = c[[5, 77]] + 0.01 * rng.standard_normal((2, 384)).astype(np.float32)
    idx, sc = O.cosine_topk(c, q, 5)
    assert idx[0, 0] == 5 and idx[1, 0] == 77
    assert np.all(np.diff(sc, axis=1) <= 0)
    # ties -> lowest id first
    c2 = np.vstack([c[:3], c[:3]])
    idx2, _ = O.cosine_topk(c2, c[:1], 2)
    assert idx2.tolist() == [[0, 3]]
    assert O.finds_sim_select([4, 2, 9], {4: ("p", "yes"), 2: ("q", "no"), 9: ("r", "no")}, "yes") == 2
    assert O.finds_sim_select([4, 2], {4: ("p", "yes"), 2: ("q", "yes")}, "yes") == 2
