#!/usr/bin/env python3
"""LTE_VL goldens from the REFERENCE's own `LTEvl` (R/editor/vllm_editors/lte_vl/lte_vl.py) on the tiny BLIP-2.  Build container only.

The reference module imports `sentence_transformers.SentenceTransformer` at load; that package is absent here and irrelevant to
the arithmetic under test, so it is replaced in-process by a stub whose `encode` is the deterministic bag-of-words encoder of
tests/lte_common.py (the product's LTEvl takes the same function as its `encode` argument).  Everything else -- the stored
edit prefixes, the retrieval decision, the hook on `get_llm_outpt`, the evaluator, `organize_batch_data` and `train_a_batch`
(full language-model fine-tuning with Adam, lte_vl.py:152-233) -- is the reference's code running on the reference's wrapper.

Writes tests/golden/tiny_lte_goldens.{json,npz} (data only):
  inf_*    prefixes (embeds + mask) and retrieval pool after two edits; per-probe hook logits, retrieved index and similarity for
           probes on both sides of sim_threshold; logits after restore
  eval     results.json of evaluate_sequential_edit(1) and (2) on 4 records
  train_*  two `train_a_batch` steps on records 0 and 1 from the committed tiny weights: losses, log dicts, and slices / checksums of
           every fine-tuned parameter after each step
"""
import json
import os
import shutil
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_goldens as MG  # noqa: E402  (import stubs, reference on sys.path)

sys.path.insert(0, os.path.join(MG.ROOT, "tests"))
from lte_common import DIM, bow_encode  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

_sts = types.ModuleType("sentence_transformers.SentenceTransformer")


class _StubSentenceTransformer:
    def __init__(self, path=None, device=None):
        self.path = path

    def encode(self, sentences):
        return bow_encode(list(sentences))


_sts.SentenceTransformer = _StubSentenceTransformer
sys.modules["sentence_transformers.SentenceTransformer"] = _sts
sys.modules["sentence_transformers"].SentenceTransformer = _sts

GOLD = MG.GOLD
t2n = MG.t2n


def main():
    from copy import deepcopy
    from editor.vllms_for_edit.blip2.blip2 import BLIP2OPTForEdit
    from editor.vllm_editors.lte_vl import lte_vl as ref_lte
    from evaluation.vllm_editor_eval import VLLMEditorEvaluation
    from dataset.vllm import BaseVLLMEditData

    os.chdir(GOLD)
    records = json.load(open(os.path.join(GOLD, "evqa8_records.json")))["records"]
    cfg = ref_lte.LTEvlConfig.from_yaml(os.path.join(MG.REF, "configs", "lte_vl", "blip2-opt-2.7b.yaml"))
    assert cfg.retrieval_embed_dim == DIM and cfg.fine_tune_modules_path == "language_model"
    vllm = BLIP2OPTForEdit(os.path.join(GOLD, "tiny_blip2"), "cpu")
    vllm_pd = BLIP2OPTForEdit(os.path.join(GOLD, "tiny_blip2"), "cpu")     # the frozen copy that prepares the training batches
    ed = ref_lte.LTEvl(vllm, cfg, "cpu", vllm_proc_data=vllm_pd, device_proc_data="cpu")
    npz, js = {}, {"sim_threshold": cfg.sim_threshold, "lr": cfg.train_config.lr}

    # ---------------- inference ----------------
    reqs = [deepcopy(records[0]["requests"][0]), deepcopy(records[1]["requests"][0])]
    for r in reqs:
        ed.edit_one_piece(deepcopy(r))
    for i, pf in enumerate(ed.edit_prefix_pool):
        npz["inf_prefix_embeds_%d" % i] = t2n(pf["inputs_embeds"]).astype(np.float32)
        npz["inf_prefix_mask_%d" % i] = t2n(pf["attention_mask"]).astype(np.int64)
    npz["inf_pool"] = t2n(ed.text_retr_pool).astype(np.float32)
    probes = [("rel0", records[0]["requests"][0]["prompt"], records[0]["requests"][0]["image"], records[0]["requests"][0]["target_new"]),
              ("gen1", records[1]["generality"]["text_rephrase"][0]["prompt"], records[1]["generality"]["text_rephrase"][0]["image"],
               records[1]["generality"]["text_rephrase"][0]["target"]),
              ("loc0", records[0]["locality"]["text_loc"][0]["prompt"], records[0]["locality"]["text_loc"][0]["image"],
               records[0]["locality"]["text_loc"][0]["target"]),
              ("loc3", records[3]["locality"]["t1i4"][0]["prompt"], records[3]["locality"]["t1i4"][0]["image"],
               records[3]["locality"]["t1i4"][0]["target"])]
    js["inf_probes"] = []
    for name, p, img, tgt in probes:
        with torch.no_grad():
            (x, vt), y, m = vllm.prompts_imgs_target_to_xym([p], [img], [tgt])
            x["query_triple"] = (p, img, tgt)
            logits = vllm.get_llm_outpt(x, vt).logits
            rr, pref, sim = ed.retrieval([p])
        npz["inf_logits_" + name] = t2n(logits).astype(np.float32)
        js["inf_probes"].append({"name": name, "prompt": p, "image": img, "target": tgt, "sim": t2n(sim).tolist(),
                                 "retrieved": None if rr[0] is None else [q["prompt"] for q in reqs].index(rr[0]["prompt"]),
                                 "logits_shape": list(logits.shape)})
        print(name, js["inf_probes"][-1]["retrieved"], js["inf_probes"][-1]["sim"], flush=True)
    ed.restore_to_original_model()
    with torch.no_grad():
        name, p, img, tgt = probes[0]
        (x, vt), y, m = vllm.prompts_imgs_target_to_xym([p], [img], [tgt])
        x["query_triple"] = (p, img, tgt)
        npz["inf_logits_restored"] = t2n(vllm.get_llm_outpt(x, vt).logits).astype(np.float32)
    js["inf_requests"] = reqs

    # ---------------- evaluator ----------------
    class _Data(BaseVLLMEditData):
        def dataset_name(self):
            return "EVQA"
    js["eval"] = {}
    for sen in (1, 2):
        root = "/tmp/devqa_gold_eval_lte_%d" % sen
        shutil.rmtree(root, ignore_errors=True)
        data = _Data(deepcopy(records[:4]), deepcopy(records[:4]))
        VLLMEditorEvaluation(ed, data, "EVQA", root).evaluate_sequential_edit(sen, False, None)
        dd = os.path.join(root, "lte_vl", "blip2-opt-2.7b", "EVQA", "sequential_edit_%d" % sen)
        res = json.load(open(os.path.join(dd, "results.json")))
        for split in res:
            for r in split:
                for rr in r["reliability"]:
                    rr.pop("edit_time", None)
        js["eval"]["sen%d" % sen] = res
        shutil.rmtree(root, ignore_errors=True)

    # ---------------- training: two steps of the reference's loop body ----------------
    ed.restore_to_original_model()
    ed.set_train(True)
    opt = ed.get_a_new_optimizer()
    ed.opt = opt
    names = [n for n, _ in ed.get_modules_for_training()["llm"].named_parameters()]
    js["train_param_names"] = names
    js["train"] = []
    step = -1
    for ri in range(len(records)):
        if step == 1:
            break
        batch = ed.organize_batch_data([deepcopy(records[ri])])
        try:
            loss, log = ed.train_a_batch(batch)
        except IndexError:      # prefix ++ locality probe longer than the tiny model's 128 positions (no update has happened)
            print("record", ri, "skipped: too long for the tiny position table", flush=True)
            continue
        step += 1
        entry = {"record": ri, "loss": loss, "log": log}
        print("train step", step, loss, log, flush=True)
        sd = dict(ed.get_modules_for_training()["llm"].named_parameters())
        for n in names:
            a = t2n(sd[n]).astype(np.float64)
            npz["train_s%d_sum_%s" % (step, n)] = np.asarray([a.sum(), np.abs(a).sum(), (a * a).sum()])
        for n in ("model.decoder.embed_tokens.weight", "model.decoder.embed_positions.weight", "model.decoder.final_layer_norm.weight",
                  "model.decoder.final_layer_norm.bias", "model.decoder.layers.0.self_attn.q_proj.weight",
                  "model.decoder.layers.0.self_attn.k_proj.bias", "model.decoder.layers.0.self_attn.out_proj.weight",
                  "model.decoder.layers.0.self_attn_layer_norm.weight", "model.decoder.layers.1.fc1.weight",
                  "model.decoder.layers.1.fc1.bias", "model.decoder.layers.1.fc2.weight", "model.decoder.layers.1.final_layer_norm.bias"):
            if n in sd:
                npz["train_s%d_w_%s" % (step, n)] = t2n(sd[n]).astype(np.float32).reshape(-1)[:4096]
        js["train"].append(entry)
    ed.set_train(False)
    np.savez_compressed(os.path.join(GOLD, "tiny_lte_goldens.npz"), **npz)
    json.dump(js, open(os.path.join(GOLD, "tiny_lte_goldens.json"), "w"), indent=1, default=str)
    print("lte goldens written:", len(npz), "arrays")


if __name__ == "__main__":
    main()
