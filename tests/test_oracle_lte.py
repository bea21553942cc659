"""CPU: the LTE_VL oracle (oracle/lte_oracle.py) on the tiny BLIP-2 -- hook pass-through with an empty pool, prefix
retrieval on both sides of the threshold, arg-max among several edits, prefix rows dropped, restore, and the oracle
evaluator driving it; and the same composition against goldens captured from the REFERENCE's own LTEvl
(tools/make_goldens_lte.py: prefixes, retrieval pool and decisions, hook logits, results.json for edit_n = 1 and 2)."""
import json
import os
from copy import deepcopy

import numpy as np
import torch

from lte_common import DIM, bow_encode


def _model(gold_dir):
    from oracle.devqa_oracle import OracleBlip2
    return OracleBlip2.from_pretrained_dir(os.path.join(gold_dir, "tiny_blip2"))


def test_lte_oracle(gold_dir, in_gold_dir):
    from oracle import devqa_oracle as O
    from oracle.lte_oracle import OracleLTEvl
    model = _model(gold_dir)
    rec = json.load(open(os.path.join(gold_dir, "evqa8_records.json")))["records"]
    ed = OracleLTEvl(model, bow_encode, 0.3, DIM)
    r0, r1 = rec[0]["requests"][0], rec[1]["requests"][0]

    def logits(prompt, image, target):
        (x, vt), y, m = model.prompts_imgs_target_to_xym([prompt], [image], [target])
        x["query_triple"] = (prompt, image, target)
        return model.get_llm_outpt(x, vt), x

    with torch.no_grad():
        plain, x0 = logits(r0["prompt"], r0["image"], r0["target_new"])
        assert torch.equal(plain, ed.plain_get_llm_outpt(x0, None))            # empty pool: pass-through
        ed.edit_one_piece(deepcopy(r0))
        ed.edit_one_piece(deepcopy(r1))
        assert ed.pool.shape == (2, DIM) and len(ed.prefixes) == 2
        for i, r in enumerate((r0, r1)):                                       # the edit's own prompt retrieves its own prefix
            req, pfx, sim = ed.retrieval([r["prompt"]])
            assert req is ed.requests[i] and int(sim.argmax()) == i and float(sim.max()) > 0.3
        got, x = logits(r0["prompt"], r0["image"], r0["target_new"])
        P = ed.prefixes[0]["inputs_embeds"].shape[1]
        cat = {"inputs_embeds": torch.cat([ed.prefixes[0]["inputs_embeds"], x["inputs_embeds"]], 1),
               "attention_mask": torch.cat([ed.prefixes[0]["attention_mask"], x["attention_mask"]], 1)}
        want = ed.plain_get_llm_outpt(cat, None)[:, P:]
        assert got.shape == plain.shape and torch.equal(got, want) and not torch.allclose(got, plain)
        # the prefix carries the edit's image tokens and the two signs
        n_img = model.cfg["num_query_tokens"]
        txt = ed.edit_sign + r0["prompt"] + " " + r0["target_new"] + ed.query_sign
        assert P == n_img + len(model.tok.encode(txt))
        # an unrelated prompt stays below the threshold -> plain path
        far = "zzqx vvk"
        assert ed.retrieval([far])[0] is None
        lf, xf = logits(far, r0["image"], "yes")
        assert torch.equal(lf, ed.plain_get_llm_outpt(xf, None))
        ed.restore_to_original_model()
        assert ed.pool.shape == (0, DIM) and torch.equal(logits(r0["prompt"], r0["image"], r0["target_new"])[0], plain)
    # through the oracle evaluator: in-context edits must not lower reliability on the tiny model
    res, ns = O.evaluate_sequential_edit(model, ed, rec[:2], 1)
    assert ns == [1, 1] and len(res) == 2 and 0.0 <= res[0][0]["reliability"][0]["acc"] <= 1.0
    ed.unhook()
    assert "get_llm_outpt" not in model.__dict__


def _flat(results):
    out = []
    for split in results:
        for d in split:
            out += [(round(x["acc"], 4), x["predict_after_edit"]) for x in d["reliability"]]
            out += [(round(x["acc"], 4), x["predict_after_edit"]) for g in d["generality"] for x in d["generality"][g]]
            out += [(round(x["acc"], 4), x["predict_before_edit"] + "|" + x["predict_after_edit"]) for l in d["locality"] for x in d["locality"][l]]
    return out


def test_lte_oracle_matches_reference_goldens(gold_dir, in_gold_dir):
    """The reference's LTEvl (sentence encoder stubbed by the same bag-of-words function) on the reference's BLIP-2 wrapper."""
    from oracle import devqa_oracle as O
    from oracle.lte_oracle import OracleLTEvl
    j = json.load(open(os.path.join(gold_dir, "tiny_lte_goldens.json")))
    z = np.load(os.path.join(gold_dir, "tiny_lte_goldens.npz"))
    rec = json.load(open(os.path.join(gold_dir, "evqa8_records.json")))["records"]
    model = _model(gold_dir)
    ed = OracleLTEvl(model, bow_encode, j["sim_threshold"], DIM)
    with torch.no_grad():
        for r in j["inf_requests"]:
            ed.edit_one_piece(deepcopy(r))
        np.testing.assert_array_equal(ed.pool.numpy(), z["inf_pool"])
        for i, pf in enumerate(ed.prefixes):
            assert pf["attention_mask"].tolist() == z["inf_prefix_mask_%d" % i].tolist()
            np.testing.assert_allclose(pf["inputs_embeds"].numpy(), z["inf_prefix_embeds_%d" % i], rtol=1e-4, atol=1e-5)
        kinds = set()
        for pr in j["inf_probes"]:
            (x, vt), y, m = model.prompts_imgs_target_to_xym([pr["prompt"]], [pr["image"]], [pr["target"]])
            x["query_triple"] = (pr["prompt"], pr["image"], pr["target"])
            got = model.get_llm_outpt(x, vt).numpy()
            req, pfx, sim = ed.retrieval([pr["prompt"]])
            assert (None if req is None else [q["prompt"] for q in j["inf_requests"]].index(req["prompt"])) == pr["retrieved"]
            np.testing.assert_allclose(sim.numpy(), np.asarray(pr["sim"]), atol=1e-6)
            gold = z["inf_logits_" + pr["name"]]
            assert list(got.shape) == pr["logits_shape"]
            assert np.abs(got - gold).max() / np.abs(gold).max() < 1e-4, pr["name"]
            kinds.add(pr["retrieved"] is None)
        assert kinds == {True, False}                 # both sides of the threshold are in the fixture
        ed.restore_to_original_model()
        pr = j["inf_probes"][0]
        (x, vt), y, m = model.prompts_imgs_target_to_xym([pr["prompt"]], [pr["image"]], [pr["target"]])
        x["query_triple"] = (pr["prompt"], pr["image"], pr["target"])
        gold = z["inf_logits_restored"]
        assert np.abs(model.get_llm_outpt(x, vt).numpy() - gold).max() / np.abs(gold).max() < 1e-4
    for sen in (1, 2):
        res, _ = O.evaluate_sequential_edit(model, ed, rec[:4], sen)
        want = j["eval"]["sen%d" % sen]
        a, b = _flat(res), _flat(want)
        assert len(a) == len(b) == 48
        assert [x[1] for x in a] == [x[1] for x in b]
        assert [x[0] for x in a] == [x[0] for x in b]


def test_lte_oracle_training_matches_reference_goldens(gold_dir, in_gold_dir):
    """oracle.lte_oracle.OracleLTETrainer against two steps of the reference's own `train_a_batch` (tools/make_goldens_lte.py)."""
    from oracle.lte_oracle import OracleLTETrainer
    j = json.load(open(os.path.join(gold_dir, "tiny_lte_goldens.json")))
    z = np.load(os.path.join(gold_dir, "tiny_lte_goldens.npz"))
    rec = json.load(open(os.path.join(gold_dir, "evqa8_records.json")))["records"]
    model = _model(gold_dir)
    w0 = {k: v.detach().clone() for k, v in model.w.items()}
    tr = OracleLTETrainer(model, "language_model.", j["lr"])
    for step, g in enumerate(j["train"]):
        loss, log = tr.train_a_batch(tr.organize_batch_data(deepcopy(rec[g["record"]])))
        assert abs(loss - g["loss"]) < 1e-4 * g["loss"]
        for k in g["log"]["Locality loss"]:
            assert abs(log["Locality loss"][k] - g["log"]["Locality loss"][k]) < 1e-4 * max(g["log"]["Locality loss"][k], 1e-2)
        for n in j["train_param_names"]:
            full = "language_model." + n
            key = "train_s%d_w_%s" % (step, n)
            if key not in z.files or n.endswith("k_proj.bias"):        # k_proj.bias: exactly-zero gradient, noise-driven Adam drift
                continue
            d_ref = z[key].astype(np.float64) - w0[full].numpy().reshape(-1)[:4096]
            d_got = model.w[full].detach().numpy().reshape(-1)[:4096].astype(np.float64) - w0[full].numpy().reshape(-1)[:4096]
            assert np.abs(d_got - d_ref).max() < 0.06 * (step + 1) * j["lr"], n
