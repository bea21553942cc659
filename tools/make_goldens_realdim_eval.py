#!/usr/bin/env python3
"""Evaluator goldens at the true BLIP-2-OPT-2.7B per-layer dims (build container only; imports the REFERENCE).

Runs the reference's own `VLLMEditorEvaluation.evaluate_sequential_edit(1, False, None)` with its own `FTvl` over the two
records of tests/golden/realdim_records.json on the 2-layer real-dim model (same numpy weight recipe as
tools/make_goldens.py --realdim) and stores, data only:

  realdim_eval_goldens.json : results.json / mean_results.json of the run (edit_time removed)
  realdim_eval_goldens.npz  : for each of the 21 evaluator forwards of a cycle (9 pre-edit locality probes in dict order,
                              then reliability, 2 generality, 9 locality post-edit), over the LAST <= 40 rows of the logits:
                              the 8 largest logits + their ids and the row's logsumexp.  Probe p of cycle c is call c*21 + p.

The top-8 rows let the GPU tests (tests/test_realdim_batched_gpu.py) check logit VALUES of the batched engine against the
reference without storing [rows, 50272] tensors, and scale the bf16 argmax-agreement bar by the reference's own top-1 margin.
"""
import json
import os
import shutil
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_goldens as MG  # noqa: E402  (sets up sys.path, the two stubs, torch threads)

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROWS, TOPK = 40, 8


def main():
    from copy import deepcopy
    from editor.vllms_for_edit.blip2.blip2 import BLIP2OPTForEdit
    from editor.vllm_editors.ft_vl import ft_vl as ref_ft
    from evaluation.vllm_editor_eval import VLLMEditorEvaluation
    from dataset.vllm import BaseVLLMEditData
    rec = json.load(open(os.path.join(MG.GOLD, "realdim_records.json")))
    assert rec["spec"] == MG.REALDIM and rec["seed"] == 2 and rec["style"] == "opt"
    rd_dir = "/tmp/devqa_realdim_blip2_eval"
    tok = MG.build_tokenizer()
    model = MG.build_model(MG.REALDIM, seed=2, style="opt")
    MG.save_tiny(model, tok, rd_dir, 224)
    del model
    os.chdir(MG.GOLD)
    vllm = BLIP2OPTForEdit(rd_dir, "cpu")
    cfg = ref_ft.FTvlConfig(edit_model_name="blip2-opt-2.7b",
                            rewrite_module_tmp="language_model.model.decoder.layers.{}.fc2.weight",
                            layers=[1], num_steps=25, lr=1e-3, weight_decay=0, norm_constraint=False, batch_size=1)
    editor = ref_ft.FTvl(vllm, cfg, "cpu")

    calls = {"val": [], "idx": [], "lse": [], "n": []}
    state = {"rec": True}
    orig_out = vllm.get_llm_outpt
    orig_edit = editor.edit_one_piece

    def rec_out(llm_inpt, vt_range):
        out = orig_out(llm_inpt, vt_range)
        if state["rec"]:
            lg = out.logits[0].detach().float()
            n = min(ROWS, lg.shape[0])
            tail = lg[-n:]
            tv, ti = tail.topk(TOPK, -1)
            val = np.zeros((ROWS, TOPK), np.float32)
            idx = np.zeros((ROWS, TOPK), np.int32)
            lse = np.zeros((ROWS,), np.float32)
            val[ROWS - n:] = tv.numpy()
            idx[ROWS - n:] = ti.numpy()
            lse[ROWS - n:] = torch.logsumexp(tail, -1).numpy()
            calls["val"].append(val)
            calls["idx"].append(idx)
            calls["lse"].append(lse)
            calls["n"].append(n)
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
    res_root = "/tmp/devqa_gold_eval_realdim"
    shutil.rmtree(res_root, ignore_errors=True)
    records = rec["records"]
    data = _Data(deepcopy(records), deepcopy(records))
    ev = VLLMEditorEvaluation(editor, data, "EVQA", res_root)
    ev.evaluate_sequential_edit(1, False, None)
    d = os.path.join(res_root, "ft_vl", "blip2-opt-2.7b", "EVQA", "sequential_edit_1")
    res = json.load(open(os.path.join(d, "results.json")))
    mean = json.load(open(os.path.join(d, "mean_results.json")))
    for split in res:
        for r in split:
            for rr in r["reliability"]:
                rr.pop("edit_time", None)
    mean["total_mean"]["reliability"].pop("edit_time", None)
    for sm in mean["split_mean"]:
        sm["reliability"].pop("edit_time", None)
    assert len(calls["n"]) == 21 * len(records), len(calls["n"])
    json.dump({"g5_results_sen1": res, "g5_mean_sen1": mean, "rows": ROWS, "topk": TOPK, "calls_per_cycle": 21},
              open(os.path.join(MG.GOLD, "realdim_eval_goldens.json"), "w"), indent=1)
    np.savez_compressed(os.path.join(MG.GOLD, "realdim_eval_goldens.npz"), top_val=np.stack(calls["val"]),
                        top_idx=np.stack(calls["idx"]), lse=np.stack(calls["lse"]), n_rows=np.asarray(calls["n"], np.int32))
    shutil.rmtree(rd_dir, ignore_errors=True)
    shutil.rmtree(res_root, ignore_errors=True)
    print("realdim evaluator goldens written: %d calls" % len(calls["n"]))


if __name__ == "__main__":
    main()
