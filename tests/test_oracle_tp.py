"""CPU: the TP_VL oracle restatement against goldens produced by the reference's own TPvl (tools/make_goldens_tp.py): the OPT
FFN on the tiny BLIP-2 (fc1 -> fc2) and the gated LLaMA FFN on the tiny LLaVA (gate_proj + up_proj -> down_proj)."""
import json
import os
from copy import deepcopy

import numpy as np
import pytest
import torch
import yaml


class Draws:
    def __init__(self, seq):
        self.seq, self.i = list(seq), 0

    def choice(self, n, k):
        v = self.seq[self.i]
        self.i += 1
        return np.array([v])


@pytest.fixture(scope="module", params=["blip2", "llava"])
def tp(gold_dir, request):
    from oracle.tp_oracle import OracleTPvl
    if request.param == "blip2":
        from oracle.devqa_oracle import OracleBlip2
        model, tag = OracleBlip2.from_pretrained_dir(os.path.join(gold_dir, "tiny_blip2")), "tiny_tp"
    else:
        from oracle.llava_oracle import OracleLlava
        model, tag = OracleLlava.from_pretrained_dir(os.path.join(gold_dir, "tiny_llava")), "tiny_tp_llava"
    cfg = yaml.safe_load(open(os.path.join(gold_dir, tag + "_cfg.yaml")))
    j = json.load(open(os.path.join(gold_dir, tag + "_goldens.json")))
    z = np.load(os.path.join(gold_dir, tag + "_goldens.npz"))
    return model, OracleTPvl(model, cfg, j["sentences"], Draws(j["draws_edits"])), j, z


def test_tp_oracle_edits(tp, in_gold_dir):
    model, ed, j, z = tp
    pr = j["probe"]

    def logits():
        with torch.no_grad():
            (x, vt), y, m = model.prompts_imgs_target_to_xym([pr["prompt"]], [pr["image"]], [pr["target"]])
            return model.get_llm_outpt(x, vt).numpy()
    np.testing.assert_allclose(logits(), z["pre_logits"], atol=2e-4)
    for tag, r in zip("ab", j["requests"]):
        ed.edit_one_piece(deepcopy(r))
        parts = ((("kg", ed.K[0]), ("bg", ed.B[0]), ("ku", ed.K[1]), ("bu", ed.B[1]), ("v", ed.V)) if ed.gated
                 else (("k", ed.K[0]), ("b", ed.B[0]), ("v", ed.V)))
        for key, got in parts:
            g = z["%s_%s" % (tag, key)]
            assert got.shape == g.shape
            assert np.abs(got.numpy() - g).max() < 2e-3 * np.abs(g).max(), (tag, key, np.abs(got.numpy() - g).max())
        gl = z[tag + "_post_logits"]
        assert np.abs(logits() - gl).max() < 2e-3 * np.abs(gl).max()
    ed.restore_to_original_model()
    np.testing.assert_allclose(logits(), z["restored_logits"], atol=2e-4)


def test_tp_oracle_evaluator(tp, in_gold_dir, gold_dir):
    from oracle.devqa_oracle import evaluate_sequential_edit
    model, ed, j, z = tp
    rec = json.load(open(os.path.join(gold_dir, "evqa8_records.json")))
    ed.restore_to_original_model()
    ed.rng = Draws(j["draws_eval"])
    res, _ = evaluate_sequential_edit(model, ed, deepcopy(rec["records"][:3]), 1)
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
    assert n == 36 and same == 36, (same, n)
