#!/usr/bin/env python3
"""Where one BLIP-2 + MEND_VL cycle spends its time at full dims (rocprofv3 --stats friendly: 1 warm-up + 3 cycles)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import bench_configs as BC  # noqa: E402


def main():
    from concurrent.futures import ThreadPoolExecutor
    from transformers import AutoTokenizer
    from devqa_amd import blip2_spec
    from devqa_amd.editor.vllms_for_edit.blip2.modeling import Blip2Native
    from devqa_amd.editor.vllms_for_edit.blip2.blip2 import BLIP2OPTForEdit
    from devqa_amd.editor.vllm_editors.mend_vl.mend_vl import MENDvl, MENDvlConfig
    model = Blip2Native(blip2_spec.BLIP2_OPT_2_7B, BC.DEV, "bf16")
    BC.fill(model, 20251121, "opt")
    tok = AutoTokenizer.from_pretrained(os.path.join(BC.GOLD, "tiny_blip2"))
    vllm = BLIP2OPTForEdit(None, BC.DEV, model=model, tokenizer=tok)
    cfg = MENDvlConfig.from_yaml(os.path.join(ROOT, "de-vqa_amd", "configs", "mend_vl", "blip2-opt-2.7b.yaml"))
    ed = MENDvl(vllm, cfg, BC.DEV, for_train=True)
    os.chdir(BC.GOLD)
    recs = BC.records(4)
    ed.set_train(True)
    ed.train_a_batch(ed.organize_batch_data([recs[0]]))     # makes the statistics finite
    ed.set_train(False)
    ed.restore_to_original_model()
    req = recs[1]["requests"][0]
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.time()
        ed.edit_one_piece(dict(req))
        torch.cuda.synchronize()
        t1 = time.time()
        ed.restore_to_original_model()
        print("edit_one_piece %.1f ms" % ((t1 - t0) * 1e3), flush=True)
    # evaluation phases of one cycle through the generic evaluator's batched-probe path
    from devqa_amd.evaluation.vllm_editor_eval import VLLMEditorEvaluation as EV
    d = recs[2]
    loc = [(e["prompt"], e["image"], e["target"]) for k in d["locality"] for e in d["locality"][k]]
    post = [(d["requests"][0]["prompt"], d["requests"][0]["image"], d["requests"][0]["target_new"])] + \
        [(e["prompt"], e["image"], e["target"]) for k in d["generality"] for e in d["generality"][k]] + loc
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    EV._argmax_many(vllm, loc)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.time()
        EV._argmax_many(vllm, loc)
        torch.cuda.synchronize()
        t1 = time.time()
        EV._argmax_many(vllm, post)
        torch.cuda.synchronize()
        t2 = time.time()
        print("pre-edit phase (9 probes) %.1f ms, post-edit phase (12 probes) %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)


if __name__ == "__main__":
    main()
