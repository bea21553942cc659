#!/usr/bin/env python3
"""Generate tests/golden/* by running the REFERENCE (sev777/DE-VQA) in this
build container.  Never runs on the GPU box: /root/reference does not travel.

What it does:
  * puts /root/reference/DE-VQA on sys.path with two in-process stubs for
    packages that are absent here and unused by the arithmetic
    (torch.utils.tensorboard, sentence_transformers);
  * builds a tiny random BLIP-2 (weights from devqa_amd.synth.param_init, so
    the product side can re-materialise them) + a locally trained byte-level
    BPE tokenizer with OPT conventions, saved to tests/golden/tiny_blip2/;
  * runs the reference's own prompts_imgs_target_to_xym, get_llm_input_embeds,
    get_llm_outpt, label_loss, FTvl.execute_ft and
    VLLMEditorEvaluation.evaluate_sequential_edit and stores inputs/outputs as
    small .npz/.json fixtures (G1..G5, G7);
  * (--realdim) repeats the logits / FT goldens on a 2-layer model with the
    true BLIP-2-OPT-2.7B per-layer dims; only slices + checksums are stored.

Only DATA (inputs and expected outputs) is written; no reference source text.
"""
import argparse
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

# ---- stubs for absent, arithmetic-irrelevant imports -----------------------
_tb = types.ModuleType("torch.utils.tensorboard")


class _SummaryWriter:
    def __init__(self, *a, **k):
        pass

    def add_scalar(self, *a, **k):
        pass


_tb.SummaryWriter = _SummaryWriter
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

torch.manual_seed(0)
torch.set_num_threads(8)


# ---------------------------------------------------------------------------
def build_tokenizer(vocab_size=600):
    from tokenizers import Tokenizer, models, pre_tokenizers, decoders, trainers, processors
    from transformers import PreTrainedTokenizerFast
    recs = json.load(open(os.path.join(REF, "data/easy-edit-mm/vqa/vqa_eval.json")))
    corpus = []
    for d in recs[:400]:
        for k in ("src", "rephrase", "alt", "loc", "loc_ans", "m_loc_q", "m_loc_a", "m_loc", "pred"):
            corpus.append(str(d[k]))
    corpus += [" The answer is:", " The answer is:?"] * 50
    tok = Tokenizer(models.BPE(unk_token=None))
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = decoders.ByteLevel()
    trainer = trainers.BpeTrainer(vocab_size=vocab_size, special_tokens=["<s>", "<pad>", "</s>", "<unk>"],
                                  initial_alphabet=pre_tokenizers.ByteLevel.alphabet())
    tok.train_from_iterator(corpus, trainer)
    tok.post_processor = processors.TemplateProcessing(single="</s> $A", special_tokens=[("</s>", 2)])
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, bos_token="</s>", eos_token="</s>",
                                   pad_token="<pad>", unk_token="<unk>")
    return fast


TINY = dict(
    vision=dict(hidden_size=48, intermediate_size=96, num_hidden_layers=2, num_attention_heads=2,
                image_size=28, patch_size=14, layer_norm_eps=1e-6, hidden_act="gelu", qkv_bias=True),
    qformer=dict(hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2,
                 cross_attention_frequency=2, encoder_hidden_size=48, layer_norm_eps=1e-12,
                 hidden_act="gelu", vocab_size=32, max_position_embeddings=64),
    text=dict(hidden_size=40, ffn_dim=80, num_hidden_layers=2, num_attention_heads=5,
              vocab_size=640, max_position_embeddings=128, word_embed_proj_dim=40),
    num_query_tokens=8,
)
REALDIM = dict(
    vision=dict(hidden_size=1408, intermediate_size=6144, num_hidden_layers=2, num_attention_heads=16,
                image_size=224, patch_size=14, layer_norm_eps=1e-6, hidden_act="gelu", qkv_bias=True),
    qformer=dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=2, num_attention_heads=12,
                 cross_attention_frequency=2, encoder_hidden_size=1408, layer_norm_eps=1e-12,
                 hidden_act="gelu", vocab_size=32, max_position_embeddings=64),
    text=dict(hidden_size=2560, ffn_dim=10240, num_hidden_layers=2, num_attention_heads=32,
              vocab_size=50272, max_position_embeddings=2048, word_embed_proj_dim=2560),
    num_query_tokens=32,
)


def build_model(spec, seed, style="unit"):
    from transformers import (Blip2Config, Blip2VisionConfig, Blip2QFormerConfig, OPTConfig,
                              Blip2ForConditionalGeneration)
    vc = Blip2VisionConfig(**spec["vision"])
    qc = Blip2QFormerConfig(**spec["qformer"])
    tc = OPTConfig(pad_token_id=1, bos_token_id=2, eos_token_id=2, do_layer_norm_before=True,
                   activation_function="relu", dropout=0.0, attention_dropout=0.0, **spec["text"])
    cfg = Blip2Config(vision_config=vc.to_dict(), qformer_config=qc.to_dict(), text_config=tc.to_dict(),
                      num_query_tokens=spec["num_query_tokens"])
    cfg.image_token_index = None
    with torch.device("meta"):
        pass
    model = Blip2ForConditionalGeneration(cfg)
    with torch.no_grad():
        for name, p in model.named_parameters():
            p.copy_(torch.from_numpy(param_init(name, p.shape, seed, style)))
    model.tie_weights()
    return model.eval()


def save_tiny(model, tok, out_dir, img_size):
    from transformers import Blip2Processor, BlipImageProcessor
    if os.path.isdir(out_dir):
        shutil.rmtree(out_dir)
    os.makedirs(out_dir)
    ip = BlipImageProcessor(size={"height": img_size, "width": img_size})
    proc = Blip2Processor(ip, tok, num_query_tokens=None)
    model.save_pretrained(out_dir, safe_serialization=True)
    proc.save_pretrained(out_dir)


def write_images(img_dir, n_samples, size, rng):
    os.makedirs(img_dir, exist_ok=True)
    paths = []
    for s in range(n_samples):
        row = []
        for tag in ("i1", "ir", "i2", "i3"):
            arr = rng.integers(0, 256, size=(size, size, 3), dtype=np.uint8)
            p = os.path.join(img_dir, "s%d_%s.png" % (s, tag))
            Image.fromarray(arr).save(p)
            row.append(os.path.relpath(p, GOLD))
        paths.append(row)
    # one non-native-size image to pin the bicubic resize path
    arr = rng.integers(0, 256, size=(size + 9, size + 5, 3), dtype=np.uint8)
    p = os.path.join(img_dir, "odd_size.png")
    Image.fromarray(arr).save(p)
    return paths, os.path.relpath(p, GOLD)


def build_records(n, img_paths):
    """EVQA-shaped probe dicts (structure of R/dataset/vllm.py:121-228,231-254) from the
    first n records of the reference's vqa_eval.json; retrieval (finds_sim) is replaced
    by a deterministic pick (t2/answer from record (i+1)%n, its image = that sample's i2)."""
    recs = json.load(open(os.path.join(REF, "data/easy-edit-mm/vqa/vqa_eval.json")))[:n]
    out = []
    raw = []
    for i, d in enumerate(recs):
        i1, ir, i2, i3 = img_paths[i]
        d2 = recs[(i + 1) % n]
        t1, t2, t3 = d["src"], d2["src"], d["m_loc"]
        sfx = " The answer is:"
        new_d = {
            "requests": [{"image": i1, "prompt": t1 + sfx, "target_new": d["alt"]}],
            "generality": {
                "text_rephrase": [{"image": i1, "prompt": d["rephrase"] + sfx, "target": d["alt"]}],
                "image_rephrase": [{"image": ir, "prompt": t1 + sfx, "target": d["alt"]}],
            },
            "locality": {
                "text_loc": [{"image": None, "prompt": d["loc"] + sfx + "?", "target": d["loc_ans"]}],
                "t3i3": [{"image": i3, "prompt": d["m_loc_q"] + sfx, "target": d["m_loc_a"]}],
                "t1i4": [{"image": None, "prompt": t1 + sfx, "target": d["alt"]}],
                "t2i4": [{"image": None, "prompt": t2 + sfx, "target": d["alt"]}],
                "t1i2": [{"image": i2, "prompt": t1 + sfx, "target": d["alt"]}],
                "t1i3": [{"image": i3, "prompt": t1 + sfx, "target": d["alt"]}],
                "t2i1": [{"image": i1, "prompt": t2 + sfx, "target": d["alt"]}],
                "t2i2": [{"image": i2, "prompt": t2 + sfx, "target": d["alt"]}],
                "t3i1": [{"image": i1, "prompt": t3 + sfx, "target": d["m_loc_a"]}],
            },
        }
        out.append(new_d)
        raw.append({k: d[k] for k in ("src", "pred", "rephrase", "alt", "image", "image_rephrase", "loc",
                                      "loc_ans", "m_loc", "m_loc_q", "m_loc_a")})
    return out, raw


def t2n(t):
    return t.detach().cpu().numpy()


# ---------------------------------------------------------------------------
def run_reference_suite(model_dir, tag, records, odd_img, full_delta, n_eval, ft_cfg_layers):
    """Run the reference on the saved model; returns dict of goldens."""
    from copy import deepcopy
    from editor.vllms_for_edit.blip2.blip2 import BLIP2OPTForEdit
    from editor.vllm_editors.ft_vl import ft_vl as ref_ft
    from evaluation.vllm_editor_eval import VLLMEditorEvaluation
    from dataset.vllm import BaseVLLMEditData

    os.chdir(GOLD)  # image paths are relative to tests/golden
    vllm = BLIP2OPTForEdit(model_dir, "cpu")
    tok = vllm.get_llm_tokenizer()
    out_npz, out_json = {}, {}

    # ---- G1: xym bookkeeping ------------------------------------------------
    pairs = [
        ("How many tennis balls are in the picture? The answer is:", "2", records[0]["requests"][0]["image"]),
        ("What sport can you use this for? The answer is:", "motocross", records[0]["locality"]["t3i3"][0]["image"]),
        ("nq question: what purpose did seasonal monsoon winds have on trade The answer is:?",
         "enabled European empire expansion into the Americas", None),
        ("Is this a trailing space prompt? ", "yes it is", records[1]["requests"][0]["image"]),
        ("Target with leading space The answer is:", " red", None),
        ("val2014/COCO_val2014_000000297147.jpg The answer is:", "motocross", records[1]["requests"][0]["image"]),
        ("Odd sized image resize check The answer is:", "blue", odd_img),
    ]
    g1 = []
    for i, (p, t, img) in enumerate(pairs):
        with torch.no_grad():
            (x, vt), y, m = vllm.prompts_imgs_target_to_xym([p], [img], [t])
            logits = vllm.get_llm_outpt(x, vt).logits
        g1.append({"prompt": p, "target": t, "image": img, "vt_range": vt,
                   "embeds_shape": list(x["inputs_embeds"].shape),
                   "label_ids": t2n(y).tolist(), "label_masks": t2n(m).tolist(),
                   "attention_mask": t2n(x["attention_mask"]).tolist()})
        out_npz["g2_embeds_%d" % i] = t2n(x["inputs_embeds"]).astype(np.float32)
        lg = t2n(logits).astype(np.float32)
        if full_delta:
            out_npz["g3_logits_%d" % i] = lg
        else:  # real-dim: last-L rows only + checksums
            L = y.shape[1]
            out_npz["g3_logits_lastL_%d" % i] = lg[:, -L:, :]
            out_npz["g3_logits_rowsum_%d" % i] = lg.astype(np.float64).sum(-1).astype(np.float64)
        # G7 label_loss KAT
        g1[-1]["label_loss"] = float(ref_ft.label_loss(logits, y, m))
        g1[-1]["label_loss_vllm"] = float(vllm.label_loss(logits, y, m))
    # batched xym (ragged prompts, text only) -- pins the pad/crop rule
    with torch.no_grad():
        bp = ["Short q? The answer is:", "A considerably longer question about something else? The answer is:"]
        bt = ["yes", "a long answer here"]
        (x, vt), y, m = vllm.prompts_imgs_target_to_xym(bp, [None, None], bt)
        logits = vllm.get_llm_outpt(x, vt).logits
        out_json["g1_batch"] = {"prompts": bp, "targets": bt, "label_ids": t2n(y).tolist(),
                                "label_masks": t2n(m).tolist(),
                                "attention_mask": t2n(x["attention_mask"]).tolist(),
                                "label_loss": float(ref_ft.label_loss(logits, y, m)),
                                "kl_self": float(vllm.logit_KL_loss(logits, logits * 0.5, m))}
    out_json["g1"] = g1
    # pixel values of the odd-size image (pins resize+normalise)
    with Image.open(odd_img) as im:
        pv = vllm.processor(im.copy(), ["x"], return_tensors="pt")["pixel_values"]
    out_npz["pixel_values_odd"] = t2n(pv).astype(np.float32)

    # ---- G4: FT_VL execute_ft ------------------------------------------------
    cfg = ref_ft.FTvlConfig(edit_model_name="blip2-opt-2.7b",
                            rewrite_module_tmp="language_model.model.decoder.layers.{}.fc2.weight",
                            layers=ft_cfg_layers, num_steps=25, lr=1e-3, weight_decay=0,
                            norm_constraint=False, batch_size=1)
    editor = ref_ft.FTvl(vllm, cfg, "cpu")
    wname = cfg.rewrite_module_tmp.format(ft_cfg_layers[0])
    g4 = []
    ft_reqs = [records[i]["requests"][0] for i in range(min(4, len(records)))]
    ft_reqs.append({"image": None, "prompt": "Text only edit request The answer is:", "target_new": "green"})
    for i, req in enumerate(ft_reqs):
        losses = []
        orig_update = ref_ft.AverageMeter.update

        def rec_update(self, val, n=1, _l=losses, _o=orig_update):
            _l.append(float(val))
            return _o(self, val, n)

        ref_ft.AverageMeter.update = rec_update
        try:
            deltas = editor.execute_ft([req])
        finally:
            ref_ft.AverageMeter.update = orig_update
        d = t2n(deltas[wname]).astype(np.float32)
        entry = {"request": req, "losses": losses, "steps": len(losses), "weight": wname,
                 "delta_sum": float(d.astype(np.float64).sum()),
                 "delta_l2": float(np.sqrt((d.astype(np.float64) ** 2).sum())),
                 "delta_absmax": float(np.abs(d).max())}
        if full_delta:
            out_npz["g4_delta_%d" % i] = d
        else:
            rs = np.random.default_rng(7 + i)
            idx = np.stack([rs.integers(0, d.shape[0], 256), rs.integers(0, d.shape[1], 256)], 1)
            out_npz["g4_delta_idx_%d" % i] = idx.astype(np.int64)
            out_npz["g4_delta_val_%d" % i] = d[idx[:, 0], idx[:, 1]]
            out_npz["g4_delta_rowsum_%d" % i] = d.astype(np.float64).sum(1)
        g4.append(entry)
        # the edit is applied; post-edit logits on the edit prompt's label rows; then restored
        editor.edit_one_piece(req)
        with torch.no_grad():
            (x, vt), y, m = vllm.prompts_imgs_target_to_xym([req["prompt"]], [req["image"]], [req["target_new"]])
            post = vllm.get_llm_outpt(x, vt).logits[:, -y.shape[1]:]
        out_npz["g4_post_logits_%d" % i] = t2n(post).astype(np.float32)
        entry["post_label_ids"] = t2n(y).tolist()
        entry["post_label_loss"] = float(ref_ft.label_loss(post, y, m))
        editor.restore_to_original_model()
    out_json["g4"] = g4
    # ---- G4b: config variants (early stop, weight decay, L-inf clamp) ----------
    g4b = []
    variants = [dict(lr=3e-2, weight_decay=0, norm_constraint=False),
                dict(lr=1e-2, weight_decay=0.1, norm_constraint=False),
                dict(lr=1e-2, weight_decay=0, norm_constraint=5e-3)]
    if not full_delta:
        variants = variants[:1]
    for vi, var in enumerate(variants):
        cfg_v = ref_ft.FTvlConfig(edit_model_name="blip2-opt-2.7b",
                                  rewrite_module_tmp="language_model.model.decoder.layers.{}.fc2.weight",
                                  layers=ft_cfg_layers, num_steps=25, batch_size=1, **var)
        ed_v = ref_ft.FTvl(vllm, cfg_v, "cpu")
        for ri, req in enumerate(ft_reqs[:2]):
            losses = []
            orig_update = ref_ft.AverageMeter.update

            def rec_update2(self, val, n=1, _l=losses, _o=orig_update):
                _l.append(float(val))
                return _o(self, val, n)

            ref_ft.AverageMeter.update = rec_update2
            try:
                deltas = ed_v.execute_ft([req])
            finally:
                ref_ft.AverageMeter.update = orig_update
            d = t2n(deltas[wname]).astype(np.float32)
            g4b.append({"cfg": var, "request": req, "losses": losses, "steps": len(losses),
                        "delta_sum": float(d.astype(np.float64).sum()),
                        "delta_l2": float(np.sqrt((d.astype(np.float64) ** 2).sum())),
                        "delta_absmax": float(np.abs(d).max())})
            if full_delta:
                out_npz["g4b_delta_%d_%d" % (vi, ri)] = d
            else:
                out_npz["g4b_delta_rowsum_%d_%d" % (vi, ri)] = d.astype(np.float64).sum(1)
    out_json["g4b"] = g4b

    # ---- G5: evaluator --------------------------------------------------------
    if n_eval:
        class _Data(BaseVLLMEditData):
            def dataset_name(self):
                return "EVQA"
        res_root = os.path.join("/tmp", "devqa_gold_eval_%s" % tag)
        shutil.rmtree(res_root, ignore_errors=True)
        for edit_n in (1, 3):
            data = _Data(deepcopy(records[:n_eval]), deepcopy(records[:n_eval]))
            ev = VLLMEditorEvaluation(editor, data, "EVQA", res_root)
            ev.evaluate_sequential_edit(edit_n, False, None)
            d = os.path.join(res_root, "ft_vl", "blip2-opt-2.7b", "EVQA", "sequential_edit_%d" % edit_n)
            res = json.load(open(os.path.join(d, "results.json")))
            mean = json.load(open(os.path.join(d, "mean_results.json")))
            for split in res:
                for r in split:
                    for rr in r["reliability"]:
                        rr.pop("edit_time", None)
            mean["total_mean"]["reliability"].pop("edit_time", None)
            for sm in mean["split_mean"]:
                sm["reliability"].pop("edit_time", None)
            out_json["g5_results_sen%d" % edit_n] = res
            out_json["g5_mean_sen%d" % edit_n] = mean
    return out_npz, out_json


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--realdim", action="store_true", help="also generate the real-head-dim goldens (slow, ~2 GB RAM)")
    ap.add_argument("--skip-tiny", action="store_true")
    args = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    rng = np.random.default_rng(20251121)
    tok = build_tokenizer()

    if not args.skip_tiny:
        tiny_dir = os.path.join(GOLD, "tiny_blip2")
        model = build_model(TINY, seed=1)
        save_tiny(model, tok, tiny_dir, TINY["vision"]["image_size"])
        img_paths, odd = write_images(os.path.join(GOLD, "images"), 8, TINY["vision"]["image_size"], rng)
        records, raw = build_records(8, img_paths)
        json.dump({"records": records, "raw": raw, "odd_image": odd}, open(os.path.join(GOLD, "evqa8_records.json"), "w"),
                  indent=1)
        npz, js = run_reference_suite(tiny_dir, "tiny", records, odd, True, 8, [1])
        np.savez_compressed(os.path.join(GOLD, "tiny_goldens.npz"), **npz)
        json.dump(js, open(os.path.join(GOLD, "tiny_goldens.json"), "w"), indent=1)
        print("tiny goldens written")

    if args.realdim:
        rd_dir = "/tmp/devqa_realdim_blip2"
        model = build_model(REALDIM, seed=2, style="opt")
        save_tiny(model, tok, rd_dir, 224)
        del model
        rng2 = np.random.default_rng(20251122)
        img_paths, odd = write_images(os.path.join(GOLD, "images224"), 2, 224, rng2)
        records, raw = build_records(2, img_paths)
        json.dump({"records": records, "raw": raw, "odd_image": odd, "spec": REALDIM, "seed": 2, "style": "opt"},
                  open(os.path.join(GOLD, "realdim_records.json"), "w"), indent=1)
        npz, js = run_reference_suite(rd_dir, "realdim", records, odd, False, 0, [1])
        # embeds are large at real dims: keep only checksums + a slice
        for k in list(npz.keys()):
            if k.startswith("g2_embeds_"):
                a = npz.pop(k)
                npz[k + "_rowsum"] = a.astype(np.float64).sum(-1)
                npz[k + "_slice"] = a[:, :, :64].copy()
        npz.pop("pixel_values_odd", None)
        np.savez_compressed(os.path.join(GOLD, "realdim_goldens.npz"), **npz)
        json.dump(js, open(os.path.join(GOLD, "realdim_goldens.json"), "w"), indent=1)
        shutil.rmtree(rd_dir, ignore_errors=True)
        print("realdim goldens written")


if __name__ == "__main__":
    main()
