"""GPU: shared-prefix packing of the batched probe path (vllm_editor_eval._argmax_many): probes that start with the same image
tokens and the same (long, IKE-style) in-context text compute that prefix once.  With sharing on and off the predictions are the
same, and equal to the per-probe path; the packed batch is measurably smaller."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

CTX = ("New Fact: what color is the bus? The answer is: red\nPrompt: what color is the bus? The answer is: red\n\n"
       "New Fact: how many dogs are there? The answer is: 2\nPrompt: how many dogs are there? The answer is: 2\n\n"
       "New Fact: what is the man holding? The answer is: umbrella\nPrompt: what is the man holding? The answer is: umbrella\n\n")


@pytest.fixture(scope="module", params=["blip2", "llava", "minigpt4"])
def fam(gold_dir, request):
    import devqa_amd  # noqa: F401
    if request.param == "blip2":
        from devqa_amd.editor.vllms_for_edit.blip2.blip2 import BLIP2OPTForEdit
        vllm = BLIP2OPTForEdit(os.path.join(gold_dir, "tiny_blip2"), "cuda:0", dtype="fp32")
    elif request.param == "llava":
        from devqa_amd.editor.vllms_for_edit.llava.llava import LlavaForEdit
        vllm = LlavaForEdit(os.path.join(gold_dir, "tiny_llava"), "cuda:0", True, dtype="fp32")
    else:
        from transformers import AutoTokenizer
        from devqa_amd import minigpt4_spec as S
        from devqa_amd.editor.vllms_for_edit.minigpt4.minigpt4 import MiniGPT4ForEdit
        from devqa_amd.editor.vllms_for_edit.minigpt4.modeling import MiniGPT4Native
        model = MiniGPT4Native.from_synth(S.TINY_MINIGPT4, 31, "unit", "cuda:0", "fp32")
        vllm = MiniGPT4ForEdit(None, "cuda:0", True, model=model, tokenizer=AutoTokenizer.from_pretrained(os.path.join(gold_dir, "tiny_llava")),
                               dtype="fp32")
    rec = json.load(open(os.path.join(gold_dir, "evqa8_records.json")))["records"]
    return vllm, rec


def test_shared_prefix_equals_unshared(fam, in_gold_dir):
    from devqa_amd.evaluation.vllm_editor_eval import VLLMEditorEvaluation as E
    vllm, rec = fam
    probes = []
    for r in rec[:2]:          # two samples: two images (+ text-only probes), every prompt behind the same in-context block
        probes.append((CTX + r["requests"][0]["prompt"], r["requests"][0]["image"], r["requests"][0]["target_new"]))
        for sec in ("generality", "locality"):
            for sub in r[sec]:
                e = r[sec][sub][0]
                probes.append((CTX + e["prompt"], e["image"], e["target"]))
    eng = vllm.engine
    seen = []
    orig = eng.pack_rows
    eng.pack_rows = lambda rows, pos, desc, max_len: (seen.append((rows.shape[0], len(desc))), orig(rows, pos, desc, max_len))[1]
    try:
        shared = E._argmax_many(vllm, probes)
        os.environ["DEVQA_PROBE_PREFIX_SHARE"] = "0"
        try:
            plain = E._argmax_many(vllm, probes)
        finally:
            del os.environ["DEVQA_PROBE_PREFIX_SHARE"]
    finally:
        eng.pack_rows = orig
    (rows_shared, nseq_shared), (rows_plain, nseq_plain) = seen[0], seen[1]
    assert nseq_plain == len(probes) and nseq_shared > len(probes)          # prefix sequences were added ...
    assert rows_shared < 0.6 * rows_plain, (rows_shared, rows_plain)         # ... and the batch shrank
    single = [E._argmax_last(vllm, *p) for p in probes[:6]]
    for i, ((pa, ya, ma), (pb, yb, mb)) in enumerate(zip(shared, plain)):
        assert ya.tolist() == yb.tolist() and ma.tolist() == mb.tolist()
        assert pa.tolist() == pb.tolist(), i
    for (pa, _, _), (ps_, _, _) in zip(shared, single):
        assert pa.tolist() == ps_.tolist()
