#!/usr/bin/env python3
"""Golden fixtures for the LLaVA-1.5 path (BASELINE config #3).  Build container only.

The reference's `LlavaForEdit` (R/editor/vllms_for_edit/llava/llava.py:10-81) cannot run against the
installed transformers 5.15 (`_merge_input_ids_with_image_features` is gone, sub-modules moved
under `.model.*`).  So the goldens come from:
  * HF `LlavaForConditionalGeneration` (third-party arithmetic, built from a tiny config, weights from
    devqa_amd.synth.param_init keyed by the OLD parameter names the reference's YAML uses), driven by
  * an adapter that restates the reference wrapper's glue on the 5.15 attribute paths
    (vision tower hidden state -2, drop CLS, projector, splice at the `<image>` token, LLM forward), and
  * the REFERENCE's own FTvl / VLLMEditorEvaluation / BaseVLLMForEdit code run on top of that adapter.
Parity for this path is therefore pinned to HF + the reference's editor/evaluator, with the wrapper glue
restated (documented in DESIGN.md).  Only data is written.
"""
import json
import os
import shutil
import sys
import types

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/DE-VQA"
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
_tb = types.ModuleType("torch.utils.tensorboard")
_tb.SummaryWriter = type("SummaryWriter", (), {"__init__": lambda s, *a, **k: None, "add_scalar": lambda s, *a, **k: None})
sys.modules["torch.utils.tensorboard"] = _tb
_st = types.ModuleType("sentence_transformers")
_st.SentenceTransformer = object
_stu = types.ModuleType("sentence_transformers.util")
_st.util = _stu
sys.modules["sentence_transformers"] = _st
sys.modules["sentence_transformers.util"] = _stu

import numpy as np  # noqa: E402
import torch  # noqa: E402
from PIL import Image  # noqa: E402

import devqa_amd  # noqa: E402,F401
from devqa_amd.synth import param_init  # noqa: E402
from devqa_amd.llava_spec import new_to_old_name, TINY_LLAVA  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)


def build_tokenizer(vocab_size=600):
    from tokenizers import Tokenizer, models, pre_tokenizers, decoders, trainers, processors
    from transformers import PreTrainedTokenizerFast
    recs = json.load(open(os.path.join(REF, "data/easy-edit-mm/vqa/vqa_eval.json")))
    corpus = []
    for d in recs[:400]:
        for k in ("src", "rephrase", "alt", "loc", "loc_ans", "m_loc_q", "m_loc_a", "m_loc", "pred"):
            corpus.append(str(d[k]))
    corpus += [" The answer is:", " The answer is:?", "\n"] * 50
    tok = Tokenizer(models.BPE(unk_token=None))
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = decoders.ByteLevel()
    trainer = trainers.BpeTrainer(vocab_size=vocab_size, special_tokens=["<unk>", "<s>", "</s>", "<pad>", "<image>"],
                                  initial_alphabet=pre_tokenizers.ByteLevel.alphabet())
    tok.train_from_iterator(corpus, trainer)
    tok.post_processor = processors.TemplateProcessing(single="<s> $A", special_tokens=[("<s>", 1)])  # LLaMA: BOS only
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, bos_token="<s>", eos_token="</s>", pad_token="<pad>",
                                   unk_token="<unk>", additional_special_tokens=["<image>"])
    return fast


def build_model(spec, seed):
    from transformers import CLIPVisionConfig, LlamaConfig, LlavaConfig, LlavaForConditionalGeneration
    vc = CLIPVisionConfig(**spec["vision_config"])
    tc = LlamaConfig(**spec["text_config"])
    cfg = LlavaConfig(vision_config=vc.to_dict(), text_config=tc.to_dict(), image_token_index=spec["image_token_index"],
                      vision_feature_layer=-2, vision_feature_select_strategy="default", projector_hidden_act="gelu")
    model = LlavaForConditionalGeneration(cfg)
    with torch.no_grad():
        for name, p in model.named_parameters():
            p.copy_(torch.from_numpy(param_init(new_to_old_name(name), p.shape, seed, "llava")))
    return model.eval().requires_grad_(False)


def make_compat(model, tok, ip):
    """The adapter the reference's editors / evaluator are driven on (also used by tools/make_goldens_tp.py --llava)."""
    from editor.vllms_for_edit.base import BaseVLLMForEdit

    class HFLlavaCompat(BaseVLLMForEdit):
        """R/editor/vllms_for_edit/llava/llava.py:25-68 on transformers-5.15 attribute paths."""

        def __init__(self):
            self.model = model
            self.tokenizer = tok
            super().__init__(model, "cpu", True)

        def get_llm_tokenizer(self):
            return self.tokenizer

        def get_llm_input_embeds(self, texts, imgs=None):
            tk = self.tokenizer(texts, return_tensors="pt", padding=True)
            ids, msk = tk["input_ids"], tk["attention_mask"]
            emb = model.model.language_model.embed_tokens(ids)
            vt = None
            if imgs is not None:
                pil = []
                for p in imgs:
                    with Image.open(p) as im:
                        pil.append(im.copy())
                pv = ip(pil, return_tensors="pt")["pixel_values"]
                hs = model.model.vision_tower(pv, output_hidden_states=True).hidden_states[-2][:, 1:]
                feats = model.model.multi_modal_projector(hs)
                assert ids.shape[0] == 1
                pos = int(torch.where(ids[0] == self.get_img_special_token_id())[0][0])
                emb = torch.cat([emb[:, :pos], feats, emb[:, pos + 1:]], 1)
                msk = torch.ones(emb.shape[:2], dtype=msk.dtype)
                vt = [pos, pos + self.get_img_token_n()]
            return {"attention_mask": msk, "inputs_embeds": emb, "position_ids": None}, vt

        def get_llm_outpt(self, llm_inpt, vt_range=None):
            h = model.model.language_model(inputs_embeds=llm_inpt["inputs_embeds"], attention_mask=llm_inpt["attention_mask"],
                                           use_cache=False).last_hidden_state
            return types.SimpleNamespace(logits=model.lm_head(h))

        def get_img_special_token_str(self):
            return "<image>"

        def get_img_special_token_id(self):
            return model.config.image_token_index

        def get_img_token_n(self):
            return (model.config.vision_config.image_size // model.config.vision_config.patch_size) ** 2

        def is_q_former_based(self):
            return False

    return HFLlavaCompat()


def main():
    from copy import deepcopy
    from editor.vllms_for_edit.base import BaseVLLMForEdit
    from editor.vllm_editors.ft_vl import ft_vl as ref_ft
    from evaluation.vllm_editor_eval import VLLMEditorEvaluation
    from dataset.vllm import BaseVLLMEditData
    from transformers import CLIPImageProcessor

    out_dir = os.path.join(GOLD, "tiny_llava")
    if os.path.isdir(out_dir):
        shutil.rmtree(out_dir)
    os.makedirs(out_dir)
    tok = build_tokenizer()
    spec = deepcopy(TINY_LLAVA)
    model = build_model(spec, seed=3)
    S = spec["vision_config"]["image_size"]
    ip = CLIPImageProcessor(size={"shortest_edge": S}, crop_size={"height": S, "width": S})
    # fixture: weights under the OLD names (what the reference's YAML addresses), tokenizer, spec
    from safetensors.torch import save_file
    save_file({new_to_old_name(n): p.detach().clone().contiguous() for n, p in model.named_parameters()},
              os.path.join(out_dir, "model.safetensors"))
    tok.save_pretrained(out_dir)
    json.dump(spec, open(os.path.join(out_dir, "devqa_llava_config.json"), "w"), indent=1)

    os.chdir(GOLD)
    vllm = make_compat(model, tok, ip)
    rec = json.load(open(os.path.join(GOLD, "evqa8_records.json")))
    records = rec["records"]
    t2n = lambda t: t.detach().cpu().numpy()  # noqa: E731
    npz, js = {}, {}
    pairs = [(records[0]["requests"][0]["prompt"], "2", records[0]["requests"][0]["image"]),
             ("nq question: what purpose did seasonal monsoon winds have on trade The answer is:?", "enabled European empire", None),
             ("Odd sized image crop check The answer is:", "blue", rec["odd_image"])]
    g1 = []
    for i, (p, t, img) in enumerate(pairs):
        with torch.no_grad():
            (x, vt), y, m = vllm.prompts_imgs_target_to_xym([p], [img], [t])
            logits = vllm.get_llm_outpt(x, vt).logits
        g1.append({"prompt": p, "target": t, "image": img, "vt_range": vt, "embeds_shape": list(x["inputs_embeds"].shape),
                   "label_ids": t2n(y).tolist(), "label_masks": t2n(m).tolist(),
                   "label_loss": float(ref_ft.label_loss(logits, y, m))})
        npz["g2_embeds_%d" % i] = t2n(x["inputs_embeds"]).astype(np.float32)
        npz["g3_logits_%d" % i] = t2n(logits).astype(np.float32)
    js["g1"] = g1
    with Image.open(rec["odd_image"]) as im:
        npz["pixel_values_odd"] = t2n(ip([im.copy()], return_tensors="pt")["pixel_values"]).astype(np.float32)

    wname_new = "model.language_model.layers.1.mlp.down_proj.weight"
    cfg = ref_ft.FTvlConfig(edit_model_name="llava-v1.5-7b", rewrite_module_tmp=wname_new, layers=[1], num_steps=25,
                            lr=1e-3, weight_decay=0, norm_constraint=False, batch_size=1)
    editor = ref_ft.FTvl(vllm, cfg, "cpu")
    g4 = []
    for i, req in enumerate([records[0]["requests"][0], records[1]["requests"][0],
                             {"image": None, "prompt": "Text only edit request The answer is:", "target_new": "green"}]):
        losses = []
        orig = ref_ft.AverageMeter.update

        def rec_update(self, val, n=1, _l=losses, _o=orig):
            _l.append(float(val))
            return _o(self, val, n)
        ref_ft.AverageMeter.update = rec_update
        try:
            deltas = editor.execute_ft([req])
        finally:
            ref_ft.AverageMeter.update = orig
        d = t2n(deltas[wname_new]).astype(np.float32)
        g4.append({"request": req, "losses": losses, "steps": len(losses), "weight": new_to_old_name(wname_new),
                   "delta_l2": float(np.sqrt((d.astype(np.float64) ** 2).sum()))})
        npz["g4_delta_%d" % i] = d
    js["g4"] = g4

    class _Data(BaseVLLMEditData):
        def dataset_name(self):
            return "EVQA"
    res_root = "/tmp/devqa_gold_eval_llava"
    shutil.rmtree(res_root, ignore_errors=True)
    data = _Data(deepcopy(records[:4]), deepcopy(records[:4]))
    VLLMEditorEvaluation(editor, data, "EVQA", res_root).evaluate_sequential_edit(1, False, None)
    dd = os.path.join(res_root, "ft_vl", "llava-v1.5-7b", "EVQA", "sequential_edit_1")
    res = json.load(open(os.path.join(dd, "results.json")))
    for split in res:
        for r in split:
            for rr in r["reliability"]:
                rr.pop("edit_time", None)
    js["g5_results_sen1"] = res
    np.savez_compressed(os.path.join(GOLD, "tiny_llava_goldens.npz"), **npz)
    json.dump(js, open(os.path.join(GOLD, "tiny_llava_goldens.json"), "w"), indent=1)
    print("llava goldens written")


if __name__ == "__main__":
    main()
