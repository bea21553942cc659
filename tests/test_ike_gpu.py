"""GPU: IKE plugin through the evaluator on the tiny BLIP-2 (retrieval == float64 brute force, context is
installed / removed, weights untouched)."""
import json
import os
from copy import deepcopy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _encode(sentences):  # deterministic stand-in sentence encoder (hash -> 64-d unit-ish vector)
    out = []
    for s in sentences:
        rng = np.random.default_rng(abs(hash(s)) % (2 ** 32))
        base = np.random.default_rng(len(s) % 7).standard_normal(64)
        out.append((base + 0.5 * rng.standard_normal(64)).astype(np.float32))
    return np.stack(out)


def test_ike_plugin_end_to_end(gold_dir, in_gold_dir, tmp_path):
    import devqa_amd  # noqa: F401
    from oracle import devqa_oracle as O
    from devqa_amd.dataset.vllm import BaseVLLMEditData
    from devqa_amd.editor.vllm_editors.ike_vl.ike_vl import IKEvl, IKEvlConfig, build_ike_corpus
    from devqa_amd.editor.vllms_for_edit.blip2.blip2 import BLIP2OPTForEdit
    from devqa_amd.evaluation.vllm_editor_eval import VLLMEditorEvaluation
    rec = json.load(open(os.path.join(gold_dir, "evqa8_records.json")))
    train = [{"prompt": r["src"], "target": r["alt"], "rephrase_prompt": r["rephrase"], "locality_prompt": r["loc"],
              "locality_ground_truth": r["loc_ans"], "image_path": r["image"], "rephrase_image_path": r["image_rephrase"],
              "locality_image_path": r["m_loc"]} for r in rec["raw"]]
    corpus = build_ike_corpus(train, _encode)
    vllm = BLIP2OPTForEdit(os.path.join(gold_dir, "tiny_blip2"), "cuda:0", dtype="fp32")
    ed = IKEvl(vllm, IKEvlConfig("blip2-opt-2.7b", k=4), "cuda:0", corpus, _encode)
    req = rec["records"][0]["requests"][0]
    w_before = {n: p.clone() for n, p in vllm.model.named_parameters()}
    # retrieval == float64 brute force on the same vectors
    icl = ed.retrieve(req["prompt"], req["target_new"])
    q = _encode(["New Fact: %s %s\nPrompt: %s %s\n\n" % (req["prompt"], req["target_new"], req["prompt"], req["target_new"])])
    ridx, _ = O.cosine_topk(corpus["embeddings"], q, 4)
    assert icl[:4] == [corpus["sentences"][i] for i in ridx[0]] and icl[4].startswith("New Fact: ")
    # the edit installs the context; logits of a probe change; restore removes it
    probe = rec["records"][0]["locality"]["t1i4"][0]
    (x0, vt), y, m = vllm.prompts_imgs_target_to_xym([probe["prompt"]], [probe["image"]], [probe["target"]])
    ed.edit_one_piece(req)
    (x1, vt1), y1, m1 = vllm.prompts_imgs_target_to_xym([probe["prompt"]], [probe["image"]], [probe["target"]])
    assert x1["inputs_embeds"].shape[1] > x0["inputs_embeds"].shape[1] and y1.tolist() == y.tolist()
    l1 = vllm.get_llm_outpt(x1, vt1).logits
    assert l1.shape[1] == x1["inputs_embeds"].shape[1] and torch.isfinite(l1).all()
    ed.restore_to_original_model()
    (x2, _), _, _ = vllm.prompts_imgs_target_to_xym([probe["prompt"]], [probe["image"]], [probe["target"]])
    assert torch.equal(x2["inputs_embeds"], x0["inputs_embeds"])
    for n, p in vllm.model.named_parameters():
        assert torch.equal(p, w_before[n])
    # through the evaluator (generic path), 2 samples
    class Data(BaseVLLMEditData):
        def dataset_name(self):
            return "EVQA"
    data = Data(deepcopy(rec["records"][:2]), deepcopy(rec["records"][:2]))
    res = VLLMEditorEvaluation(ed, data, "EVQA", str(tmp_path)).evaluate_sequential_edit(1, False, None)
    assert len(res) == 2 and 0.0 <= res[0][0]["reliability"][0]["acc"] <= 1.0
    assert os.path.exists(os.path.join(str(tmp_path), "ike_vl", "blip2-opt-2.7b", "EVQA", "sequential_edit_1", "mean_results.json"))
    # the evaluator's look-ahead (a split's probes queued, read back after the next split is prepared; one vision call per group) against the
    # split-by-split order: same records, with three splits so that a queued split is completed both inside the loop and at its end
    def strip(results):
        out = deepcopy(results)
        for sp in out:
            for r in sp:
                for x in r["reliability"]:
                    x.pop("edit_time", None)
        return out
    runs = {}
    for la in ("1", "0"):
        os.environ["DEVQA_EVAL_LOOKAHEAD"] = la
        try:
            d3 = Data(deepcopy(rec["records"][:3]), deepcopy(rec["records"][:3]))
            runs[la] = strip(VLLMEditorEvaluation(ed, d3, "EVQA", str(tmp_path)).evaluate_sequential_edit(1, False, None, save=False))
        finally:
            del os.environ["DEVQA_EVAL_LOOKAHEAD"]
    assert runs["1"] == runs["0"] and len(runs["1"]) == 3

