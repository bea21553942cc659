#!/usr/bin/env python3
"""cProfile of the timed evaluator run of one secondary config (tools/bench_configs.py): where the HOST spends its time while the GPU
waits (the generic evaluator is host-bound).  Usage: host_profile.py blip2_mend [n]"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)
import bench_configs as B  # noqa: E402

cfg, args = sys.argv[1], [int(a) for a in sys.argv[2:]]
orig = B.run_eval
prof = cProfile.Profile()


def run_eval(editor, n, batched, *a, **k):
    from devqa_amd.dataset.vllm import BaseVLLMEditData
    from devqa_amd.evaluation.vllm_editor_eval import VLLMEditorEvaluation
    import torch

    class D(BaseVLLMEditData):
        def dataset_name(self):
            return "EVQA"
    os.chdir(B.GOLD)
    ev = VLLMEditorEvaluation(editor, D(B.records(n), B.records(n)), "EVQA", "/tmp/devqa_bench_cfg")
    ev.evaluate_sequential_edit(1, False, None, batched=batched, save=False)
    torch.cuda.synchronize()
    ev = VLLMEditorEvaluation(editor, D(B.records(n), B.records(n)), "EVQA", "/tmp/devqa_bench_cfg")
    prof.enable()
    res = ev.evaluate_sequential_edit(1, False, None, batched=batched, save=False)
    torch.cuda.synchronize()
    prof.disable()
    return 1.0, 1.0, res


B.run_eval = run_eval
{"llava_ft": B.llava_ft, "blip2_mend": B.blip2_mend, "minigpt4_ike": B.minigpt4_ike}[cfg](*args)
st = pstats.Stats(prof)
st.sort_stats("tottime").print_stats(28)
