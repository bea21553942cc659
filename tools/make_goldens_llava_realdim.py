#!/usr/bin/env python3
"""Real-head-dim goldens for the LLaMA-family path (BASELINE config #3).  Build container only.

A LLaVA-1.5 with the TRUE per-layer dims -- CLIP ViT-L/14-336 (d 1024, 16 heads x 64, FFN 4096, quick-GELU, 576 + 1 tokens;
3 layers, features from layer -2), projector 1024 -> 4096 -> 4096, LLaMA (d 4096, 32 heads x 128, SwiGLU 11008, RMSNorm, RoPE, V 32064;
2 layers) -- built from configs with the numpy weight recipe ("llava" style, regenerated on the product side; no weight file is
stored), run through HF `LlavaForConditionalGeneration` + the REFERENCE's own `FTvl` / `VLLMEditorEvaluation` on the compat adapter
of tools/make_goldens_llava.py (the reference's `LlavaForEdit` cannot run on transformers 5.15).

Stores slices / checksums only (tests/golden/realdim_llava_goldens.{json,npz}):
  g1  label bookkeeping, embeds slices, last-L logits rows + row sums for an image prompt, a text prompt, an odd-size image
  g4  `execute_ft` on layers.1.mlp.down_proj.weight [4096, 11008]: per-step losses, step count, delta row sums / norm / max / 256
      sampled elements, post-edit last-L logits
  g5  `evaluate_sequential_edit(1)` on one record: results.json + top-8 logits / logsumexp of the last <= 40 rows of its 21 forwards
"""
import json
import os
import shutil
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_goldens_llava as ML  # noqa: E402  (sys.path, stubs, helpers)

import numpy as np  # noqa: E402
import torch  # noqa: E402

REALDIM_LLAVA = dict(
    vision_config=dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=3, num_attention_heads=16,
                       image_size=336, patch_size=14, layer_norm_eps=1e-5, hidden_act="quick_gelu"),
    text_config=dict(hidden_size=4096, intermediate_size=11008, num_hidden_layers=2, num_attention_heads=32,
                     num_key_value_heads=32, vocab_size=32064, rms_norm_eps=1e-5, rope_theta=10000.0,
                     max_position_embeddings=4096, pad_token_id=3),
    image_token_index=4,
)
SEED, ROWS, TOPK = 5, 40, 8


def main():
    from copy import deepcopy
    from editor.vllm_editors.ft_vl import ft_vl as ref_ft
    from evaluation.vllm_editor_eval import VLLMEditorEvaluation
    from dataset.vllm import BaseVLLMEditData
    from transformers import CLIPImageProcessor
    GOLD = ML.GOLD
    tok = ML.build_tokenizer()
    spec = deepcopy(REALDIM_LLAVA)
    model = ML.build_model(spec, seed=SEED)
    S = spec["vision_config"]["image_size"]
    ip = CLIPImageProcessor(size={"shortest_edge": S}, crop_size={"height": S, "width": S})
    os.chdir(GOLD)
    vllm = ML.make_compat(model, tok, ip)
    rec = json.load(open(os.path.join(GOLD, "realdim_records.json")))
    records = rec["records"]
    t2n = lambda t: t.detach().cpu().numpy()  # noqa: E731
    npz, js = {}, {"spec": spec, "seed": SEED, "style": "llava", "rows": ROWS, "topk": TOPK}
    pairs = [(records[0]["requests"][0]["prompt"], "2", records[0]["requests"][0]["image"]),
             ("nq question: what purpose did seasonal monsoon winds have on trade The answer is:?", "enabled European empire", None),
             ("Odd sized image crop check The answer is:", "blue", rec["odd_image"])]
    g1 = []
    for i, (p, t, img) in enumerate(pairs):
        with torch.no_grad():
            (x, vt), y, m = vllm.prompts_imgs_target_to_xym([p], [img], [t])
            logits = vllm.get_llm_outpt(x, vt).logits
        L = y.shape[1]
        lg, emb = t2n(logits).astype(np.float32), t2n(x["inputs_embeds"]).astype(np.float32)
        g1.append({"prompt": p, "target": t, "image": img, "vt_range": vt, "embeds_shape": list(emb.shape), "label_ids": t2n(y).tolist(),
                   "label_masks": t2n(m).tolist(), "label_loss": float(ref_ft.label_loss(logits, y, m))})
        npz["g2_embeds_%d_slice" % i] = emb[:, :, :64].copy()
        npz["g2_embeds_%d_rowsum" % i] = emb.astype(np.float64).sum(-1)
        npz["g3_logits_lastL_%d" % i] = lg[:, -L:, :]
        npz["g3_logits_rowsum_%d" % i] = lg.astype(np.float64).sum(-1)
        print("g1", i, emb.shape, flush=True)
    js["g1"] = g1

    wname_new = "model.language_model.layers.1.mlp.down_proj.weight"
    cfg = ref_ft.FTvlConfig(edit_model_name="llava-v1.5-7b", rewrite_module_tmp=wname_new, layers=[1], num_steps=25, lr=1e-3, weight_decay=0,
                            norm_constraint=False, batch_size=1)
    editor = ref_ft.FTvl(vllm, cfg, "cpu")
    g4 = []
    for i, req in enumerate([records[0]["requests"][0], {"image": None, "prompt": "Text only edit request The answer is:", "target_new": "green"}]):
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
        rs = np.random.default_rng(11 + i)
        idx = np.stack([rs.integers(0, d.shape[0], 256), rs.integers(0, d.shape[1], 256)], 1)
        npz["g4_delta_idx_%d" % i] = idx.astype(np.int64)
        npz["g4_delta_val_%d" % i] = d[idx[:, 0], idx[:, 1]]
        npz["g4_delta_rowsum_%d" % i] = d.astype(np.float64).sum(1)
        entry = {"request": req, "losses": losses, "steps": len(losses), "weight": ML.new_to_old_name(wname_new),
                 "delta_l2": float(np.sqrt((d.astype(np.float64) ** 2).sum())), "delta_absmax": float(np.abs(d).max()),
                 "delta_nonzero_cols": int((np.abs(d).max(0) > 0).sum())}
        editor.edit_one_piece(req)
        with torch.no_grad():
            (x, vt), y, m = vllm.prompts_imgs_target_to_xym([req["prompt"]], [req["image"]], [req["target_new"]])
            post = vllm.get_llm_outpt(x, vt).logits[:, -y.shape[1]:]
        npz["g4_post_logits_%d" % i] = t2n(post).astype(np.float32)
        entry["post_label_loss"] = float(ref_ft.label_loss(post, y, m))
        editor.restore_to_original_model()
        g4.append(entry)
        print("g4", i, entry["steps"], entry["losses"][:2], entry["losses"][-1], flush=True)
    js["g4"] = g4

    # ---- evaluator on one record, with the top-8 logits of every forward outside the edit ----
    calls = {"val": [], "idx": [], "lse": [], "n": []}
    state = {"rec": True}
    orig_out, orig_edit = vllm.get_llm_outpt, editor.edit_one_piece

    def rec_out(llm_inpt, vt_range=None):
        out = orig_out(llm_inpt, vt_range)
        if state["rec"]:
            lg = out.logits[0].detach().float()
            n = min(ROWS, lg.shape[0])
            tail = lg[-n:]
            tv, ti = tail.topk(TOPK, -1)
            val, idx, lse = np.zeros((ROWS, TOPK), np.float32), np.zeros((ROWS, TOPK), np.int32), np.zeros((ROWS,), np.float32)
            val[ROWS - n:], idx[ROWS - n:], lse[ROWS - n:] = tv.numpy(), ti.numpy(), torch.logsumexp(tail, -1).numpy()
            for k, v in (("val", val), ("idx", idx), ("lse", lse), ("n", n)):
                calls[k].append(v)
        return out

    def edit_wrapped(req):
        state["rec"] = False
        try:
            return orig_edit(req)
        finally:
            state["rec"] = True
    vllm.get_llm_outpt = rec_out
    editor.edit_one_piece = edit_wrapped

    class _Data(BaseVLLMEditData):
        def dataset_name(self):
            return "EVQA"
    res_root = "/tmp/devqa_gold_eval_llava_realdim"
    shutil.rmtree(res_root, ignore_errors=True)
    data = _Data(deepcopy(records[:1]), deepcopy(records[:1]))
    VLLMEditorEvaluation(editor, data, "EVQA", res_root).evaluate_sequential_edit(1, False, None)
    dd = os.path.join(res_root, "ft_vl", "llava-v1.5-7b", "EVQA", "sequential_edit_1")
    res = json.load(open(os.path.join(dd, "results.json")))
    for split in res:
        for r in split:
            for rr in r["reliability"]:
                rr.pop("edit_time", None)
    assert len(calls["n"]) == 21
    js["g5_results_sen1"] = res
    npz["g5_top_val"], npz["g5_top_idx"] = np.stack(calls["val"]), np.stack(calls["idx"])
    npz["g5_lse"], npz["g5_n_rows"] = np.stack(calls["lse"]), np.asarray(calls["n"], np.int32)
    np.savez_compressed(os.path.join(GOLD, "realdim_llava_goldens.npz"), **npz)
    json.dump(js, open(os.path.join(GOLD, "realdim_llava_goldens.json"), "w"), indent=1)
    shutil.rmtree(res_root, ignore_errors=True)
    print("realdim llava goldens written")


if __name__ == "__main__":
    main()
