#!/usr/bin/env python3
"""Goldens for TP_VL (T-Patcher): runs the REFERENCE's TPvl (editor/vllm_editors/tp_vl/tp_vl.py) on the tiny BLIP-2
fixture in this build container.  The reference loads its memory-loss texts with `datasets.load_dataset` from a local
wikitext directory that does not exist offline: `load_dataset` is replaced IN THIS GENERATOR by a stand-in returning a
committed list of synthetic sentences, and the editor's unseeded `rng` by a seeded one (the draws are stored).
Stores data only: the sentences, the drawn indices, the patch neurons after one / two sequential edits, post-edit
logits, evaluator results.

`--llava`: the same on the tiny LLaVA (HF LlavaForConditionalGeneration behind the adapter of tools/make_goldens_llava.py, see
there why), with gate_proj AND up_proj as in-layers and down_proj as out-layer, as R/configs/tp_vl/llava-v1.5-7b.yaml selects.
The committed config names the modules by the reference's (old) paths; the generator hands the reference the transformers-5.15
paths of the same modules.
"""
import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_goldens as MG  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402
import yaml  # noqa: E402

GOLD = MG.GOLD
t2n = MG.t2n


def sentences(n=40):
    rng = np.random.default_rng(17)
    words = ("the a of and to in is was for on with as by at from that this it are were be have has had not but or an which "
             "their its also one two first new time year city river school music game team world war state north south").split()
    return [" ".join(rng.choice(words, int(rng.integers(24, 40)))) + " ." for _ in range(n)]


def main():
    from copy import deepcopy
    import datasets
    sents = sentences()
    datasets.load_dataset = lambda *a, **k: {"text": sents}
    from editor.vllms_for_edit.blip2.blip2 import BLIP2OPTForEdit
    import editor.vllm_editors.tp_vl.tp_vl as ref_tp
    ref_tp.load_dataset = datasets.load_dataset
    from evaluation.vllm_editor_eval import VLLMEditorEvaluation
    from dataset.vllm import BaseVLLMEditData
    os.chdir(GOLD)
    rec = json.load(open(os.path.join(GOLD, "evqa8_records.json")))
    records = rec["records"]
    cfg_d = {"edit_model_name": "blip2-opt-2.7b", "edit_layer": 1, "num_steps": 25, "lr": 1.0e-2, "loss_a_lambda": 1.0e-4,
             "loss_m_lambda": 1.0e-4, "weight_decay": 0,
             "mlp_in_module_tmps": ["language_model.model.decoder.layers.{}.fc1"],
             "mlp_out_module_tmps": ["language_model.model.decoder.layers.{}.fc2"]}
    yaml.safe_dump(cfg_d, open(os.path.join(GOLD, "tiny_tp_cfg.yaml"), "w"))
    cfg = ref_tp.TPvlConfig.from_yaml(os.path.join(GOLD, "tiny_tp_cfg.yaml"))
    vllm = BLIP2OPTForEdit(os.path.join(GOLD, "tiny_blip2"), "cpu")
    ed = ref_tp.TPvl(vllm, cfg, "cpu")

    class Rng:   # records what the editor draws
        def __init__(self, seed):
            self.g, self.draws = np.random.default_rng(seed), []

        def choice(self, n, k):
            v = self.g.choice(n, k)
            self.draws.append(int(v[0]))
            return v
    ed.rng = Rng(5)
    probe = records[2]["generality"]["text_rephrase"][0]

    def probe_logits():
        with torch.no_grad():
            (x, vt), y, m = vllm.prompts_imgs_target_to_xym([probe["prompt"]], [probe["image"]], [probe["target"]])
            return t2n(vllm.get_llm_outpt(x, vt).logits).astype(np.float32)
    npz, js = {}, {"sentences": list(ed.locality_data), "probe": probe, "seed": 5}
    npz["pre_logits"] = probe_logits()
    r0, r1 = deepcopy(records[0]["requests"][0]), deepcopy(records[1]["requests"][0])
    lin, lout = ed.edit_in_layers[0], ed.edit_out_layers[0]
    for tag, r in (("a", r0), ("b", r1)):
        ed.edit_one_piece(deepcopy(r))
        npz[tag + "_k"] = t2n(lin.extra_weights).astype(np.float32)      # [d, n]
        npz[tag + "_b"] = t2n(lin.extra_biases).astype(np.float32)       # [n]
        npz[tag + "_v"] = t2n(lout.extra_weights).astype(np.float32)     # [n, d]
        npz[tag + "_post_logits"] = probe_logits()
    js["requests"] = [r0, r1]
    js["draws_edits"] = list(ed.rng.draws)
    ed.restore_to_original_model()
    npz["restored_logits"] = probe_logits()

    class Data(BaseVLLMEditData):
        def dataset_name(self):
            return "EVQA"
    ed.rng = Rng(9)
    ev = VLLMEditorEvaluation(ed, Data(deepcopy(records[:3]), deepcopy(records[:3])), "EVQA", "/tmp/devqa_tp_eval")
    js["results_sen1"] = ev.evaluate_sequential_edit(1, False, None)
    js["draws_eval"] = list(ed.rng.draws)
    np.savez_compressed(os.path.join(GOLD, "tiny_tp_goldens.npz"), **npz)
    json.dump(js, open(os.path.join(GOLD, "tiny_tp_goldens.json"), "w"), indent=1, default=str)
    print("tp goldens written; neuron norms", [float(np.abs(npz[k]).max()) for k in ("a_k", "a_b", "a_v", "b_k")],
          "logit change", float(np.abs(npz["a_post_logits"] - npz["pre_logits"]).max()))


def main_llava():
    from copy import deepcopy
    import datasets
    sents = sentences()
    datasets.load_dataset = lambda *a, **k: {"text": sents}
    import make_goldens_llava as ML
    import editor.vllm_editors.tp_vl.tp_vl as ref_tp
    ref_tp.load_dataset = datasets.load_dataset
    from evaluation.vllm_editor_eval import VLLMEditorEvaluation
    from dataset.vllm import BaseVLLMEditData
    from transformers import AutoTokenizer, CLIPImageProcessor
    from devqa_amd.llava_spec import TINY_LLAVA
    spec = deepcopy(TINY_LLAVA)
    tok = AutoTokenizer.from_pretrained(os.path.join(GOLD, "tiny_llava"))
    model = ML.build_model(spec, seed=3)            # the weights of tests/golden/tiny_llava
    S = spec["vision_config"]["image_size"]
    ip = CLIPImageProcessor(size={"shortest_edge": S}, crop_size={"height": S, "width": S})
    vllm = ML.make_compat(model, tok, ip)
    os.chdir(GOLD)
    rec = json.load(open(os.path.join(GOLD, "evqa8_records.json")))
    records = rec["records"]
    layer = spec["text_config"]["num_hidden_layers"] - 1
    cfg_d = {"edit_model_name": "llava-v1.5-7b", "edit_layer": layer, "num_steps": 25, "lr": 1.0e-2, "loss_a_lambda": 1.0e-4,
             "loss_m_lambda": 1.0e-4, "weight_decay": 0,
             "mlp_in_module_tmps": ["language_model.model.layers.{}.mlp.gate_proj", "language_model.model.layers.{}.mlp.up_proj"],
             "mlp_out_module_tmps": ["language_model.model.layers.{}.mlp.down_proj"]}
    yaml.safe_dump(cfg_d, open(os.path.join(GOLD, "tiny_tp_llava_cfg.yaml"), "w"))
    cfg = ref_tp.TPvlConfig.from_yaml(os.path.join(GOLD, "tiny_tp_llava_cfg.yaml"))
    cfg.mlp_in_module_tmps = ["model.language_model.layers.{}.mlp.gate_proj", "model.language_model.layers.{}.mlp.up_proj"]
    cfg.mlp_out_module_tmps = ["model.language_model.layers.{}.mlp.down_proj"]
    ed = ref_tp.TPvl(vllm, cfg, "cpu")

    class Rng:
        def __init__(self, seed):
            self.g, self.draws = np.random.default_rng(seed), []

        def choice(self, n, k):
            v = self.g.choice(n, k)
            self.draws.append(int(v[0]))
            return v
    ed.rng = Rng(5)
    probe = records[2]["generality"]["text_rephrase"][0]

    def probe_logits():
        with torch.no_grad():
            (x, vt), y, m = vllm.prompts_imgs_target_to_xym([probe["prompt"]], [probe["image"]], [probe["target"]])
            return t2n(vllm.get_llm_outpt(x, vt).logits).astype(np.float32)
    npz, js = {}, {"sentences": list(ed.locality_data), "probe": probe, "seed": 5}
    npz["pre_logits"] = probe_logits()
    r0, r1 = deepcopy(records[0]["requests"][0]), deepcopy(records[1]["requests"][0])
    (lg, lu), lout = ed.edit_in_layers, ed.edit_out_layers[0]
    for tag, r in (("a", r0), ("b", r1)):
        ed.edit_one_piece(deepcopy(r))
        npz[tag + "_kg"] = t2n(lg.extra_weights).astype(np.float32)      # [d, n]
        npz[tag + "_bg"] = t2n(lg.extra_biases).astype(np.float32)       # [n]
        npz[tag + "_ku"] = t2n(lu.extra_weights).astype(np.float32)
        npz[tag + "_bu"] = t2n(lu.extra_biases).astype(np.float32)
        npz[tag + "_v"] = t2n(lout.extra_weights).astype(np.float32)     # [n, d]
        npz[tag + "_post_logits"] = probe_logits()
    js["requests"] = [r0, r1]
    js["draws_edits"] = list(ed.rng.draws)
    ed.restore_to_original_model()
    npz["restored_logits"] = probe_logits()

    class Data(BaseVLLMEditData):
        def dataset_name(self):
            return "EVQA"
    ed.rng = Rng(9)
    ev = VLLMEditorEvaluation(ed, Data(deepcopy(records[:3]), deepcopy(records[:3])), "EVQA", "/tmp/devqa_tp_eval_llava")
    js["results_sen1"] = ev.evaluate_sequential_edit(1, False, None)
    js["draws_eval"] = list(ed.rng.draws)
    np.savez_compressed(os.path.join(GOLD, "tiny_tp_llava_goldens.npz"), **npz)
    json.dump(js, open(os.path.join(GOLD, "tiny_tp_llava_goldens.json"), "w"), indent=1, default=str)
    print("tp llava goldens written; neuron norms", [float(np.abs(npz[k]).max()) for k in ("a_kg", "a_bg", "a_ku", "a_bu", "a_v")],
          "logit change", float(np.abs(npz["a_post_logits"] - npz["pre_logits"]).max()))


if __name__ == "__main__":
    main_llava() if "--llava" in sys.argv else main()
