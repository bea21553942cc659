"""GPU: BatchedMendEval (de-vqa_amd/batched_mend.py, BASELINE config #4's engine) against
  * the REFERENCE's own MENDvl through its evaluator (tiny_mend_goldens.json `results_sen1`, tools/make_goldens_mend.py),
  * the generic per-sample path of this repo (same records, batched=False): results, the per-edit low-rank factors and dW,
  * at the true OPT-2.7B layer dims (hyper-network 12800 -> rank 1920): the reference's post-edit logits of realdim_mend_goldens."""
import json
import os
from copy import deepcopy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _flat(results):
    out = []
    for split in results:
        r = split[0]
        rr = r["reliability"][0]
        out.append(("rel", None, round(rr["acc"], 4), rr["predict_after_edit"], None))
        for sec in ("generality", "locality"):
            for sub in r[sec]:
                it = r[sec][sub][0]
                out.append((sec, sub, round(it["acc"], 4), it["predict_after_edit"], it.get("predict_before_edit")))
    return out


@pytest.fixture(scope="module", params=["fp32", "bf16"])
def tiny(gold_dir, request):
    import devqa_amd  # noqa: F401
    from devqa_amd.editor.vllms_for_edit.blip2.blip2 import BLIP2OPTForEdit
    from devqa_amd.editor.vllm_editors.mend_vl.mend_vl import MENDvl, MENDvlConfig
    vllm = BLIP2OPTForEdit(os.path.join(gold_dir, "tiny_blip2"), "cuda:0", dtype=request.param)
    cfg = MENDvlConfig.from_yaml(os.path.join(gold_dir, "tiny_mend_cfg.yaml"))
    ed = MENDvl(vllm, cfg, "cuda:0", ckpt_path=os.path.join(gold_dir, "tiny_mend_ckpt.pt"))
    rec = json.load(open(os.path.join(gold_dir, "evqa8_records.json")))["records"]
    gold = json.load(open(os.path.join(gold_dir, "tiny_mend_goldens.json")))["results_sen1"]
    return vllm, ed, rec, gold, request.param


def _evaluate(ed, rec, batched, tmp, n):
    from devqa_amd.dataset.vllm import BaseVLLMEditData
    from devqa_amd.evaluation.vllm_editor_eval import VLLMEditorEvaluation

    class Data(BaseVLLMEditData):
        def dataset_name(self):
            return "EVQA"
    ed.restore_to_original_model()
    ev = VLLMEditorEvaluation(ed, Data(deepcopy(rec[:n]), deepcopy(rec[:n])), "EVQA", str(tmp))
    res = ev.evaluate_sequential_edit(1, False, None, batched=batched, save=False)
    return res, ev.last_mode


def test_batched_mend_equals_reference_and_generic(tiny, in_gold_dir, tmp_path):
    vllm, ed, rec, gold, mode = tiny
    res_b, mode_b = _evaluate(ed, rec, None, tmp_path, 8)          # auto-selected
    assert mode_b == "BatchedMendEval"
    res_g, mode_g = _evaluate(ed, rec, False, tmp_path, 8)
    assert mode_g == "generic"
    fb, fg = _flat(res_b), _flat(res_g)
    same_g = sum(a == b for a, b in zip(fb, fg))
    fr = _flat(gold)                                              # the reference evaluated the first 4 records
    same_r = sum(a[:4] == b[:4] for a, b in zip(fb[:len(fr)], fr))
    print(mode, "batched == generic %d/%d; batched == reference %d/%d" % (same_g, len(fb), same_r, len(fr)))
    assert len(fb) == 96 and len(fr) == 48
    if mode == "fp32":
        assert same_g == 96 and same_r == 48
    else:
        assert same_g >= 90 and same_r >= 36
    for sp in res_b:
        assert sp[0]["reliability"][0]["edit_time"] > 0


def test_batched_mend_factors_equal_per_sample_edits(tiny, in_gold_dir):
    """The per-cycle low-rank factors of the batched engine (candidate rows from the host, exact non-zero rule as a device mask,
    1/n and lr folded in) give the same delta weight as MENDvl.edit_one_piece for that request."""
    from devqa_amd.batched import copy_sample
    from devqa_amd.batched_mend import BatchedMendEval
    from devqa_amd import lib
    vllm, ed, rec, gold, mode = tiny
    ed.restore_to_original_model()
    be = BatchedMendEval(ed, cycles_per_batch=3)
    be.keep_debug = True
    cyc = [deepcopy(r) for r in rec[:3]]
    be.run_batch([copy_sample(c) for c in cyc], cyc)
    torch.cuda.synchronize()
    fac = be.debug["factors"]
    tol = 2e-5 if mode == "fp32" else 3e-2
    worst = 0.0
    for e in range(3):
        ed.restore_to_original_model()
        ed.edit_one_piece(deepcopy(rec[e]["requests"][0]))
        for m in ed.modules:
            want = ed.delta_weight(m["name"]).double()
            Xp, Dp, n_c = fac[m["name"]]
            got = (Xp[e].double().t() @ Dp[e].double())
            assert int(n_c[e]) == m["n"], (e, m["name"], int(n_c[e]), m["n"])
            err = float((got - want).abs().max()) / float(want.abs().max())
            worst = max(worst, err)
            assert err < tol, (e, m["name"], err)
    ed.restore_to_original_model()
    print(mode, "batched factors -> dW vs edit_one_piece: worst rel err %.2e" % worst)


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_batched_mend_realdim(gold_dir, in_gold_dir, mode):
    """True OPT-2.7B layer dims (2 decoder layers, both edited): the post-edit label-row logits of the batched engine vs the
    reference's (realdim_mend_goldens: `post_logits_lastL` of the probe after editing `request`)."""
    import devqa_amd  # noqa: F401
    from transformers import AutoTokenizer
    from devqa_amd.batched import copy_sample
    from devqa_amd.batched_mend import BatchedMendEval
    from devqa_amd.editor.vllms_for_edit.blip2.blip2 import BLIP2OPTForEdit
    from devqa_amd.editor.vllms_for_edit.blip2.modeling import Blip2Native
    from devqa_amd.editor.vllm_editors.mend_vl.mend_vl import MENDvl, MENDvlConfig
    from devqa_amd.synth import mend_aux_init
    rec = json.load(open(os.path.join(gold_dir, "realdim_records.json")))
    j = json.load(open(os.path.join(gold_dir, "realdim_mend_goldens.json")))
    z = np.load(os.path.join(gold_dir, "realdim_mend_goldens.npz"))
    cfg = {"vision_config": rec["spec"]["vision"], "qformer_config": rec["spec"]["qformer"],
           "text_config": rec["spec"]["text"], "num_query_tokens": rec["spec"]["num_query_tokens"]}
    model = Blip2Native.from_synth(cfg, rec["seed"], rec["style"], "cuda:0", mode)
    tok = AutoTokenizer.from_pretrained(os.path.join(gold_dir, "tiny_blip2"))
    vllm = BLIP2OPTForEdit(None, "cuda:0", model=model, tokenizer=tok)
    tm = {mn: {k: torch.from_numpy(mend_aux_init("%s.%s" % (mn, k), tuple(shp), j["aux_seed"])) for k, shp in d.items()}
          for mn, d in j["state_shapes"].items()}
    ed = MENDvl(vllm, MENDvlConfig.from_yaml(os.path.join(gold_dir, "realdim_mend_cfg.yaml")), "cuda:0", train_modules=tm)
    pr, req = j["probe"], j["request"]
    # a cycle whose reliability probe is the golden's probe and whose request is the golden's request, plus a second, different cycle
    # in the same batch (its deltas must not leak into the first)
    def cycle(request, probe):
        loc = {"image": None, "prompt": probe["prompt"], "target": probe["target"]}
        return {"requests": [dict(request)], "generality": {"g": [dict(probe)]}, "locality": {"l": [loc]}}
    other = deepcopy(rec["records"][1])
    cyc = [cycle(req, pr), cycle(other["requests"][0], {"image": other["requests"][0]["image"], "prompt": other["requests"][0]["prompt"],
                                                      "target": other["requests"][0]["target_new"]})]
    be = BatchedMendEval(ed, cycles_per_batch=2)
    be.keep_debug = True
    be.run_batch([copy_sample(c) for c in cyc], [deepcopy(c) for c in cyc])
    torch.cuda.synchronize()
    rows = {(k, n): (r0, L) for (k, n, r0, L) in be.debug["rows"][0]}
    r0, L = rows[("gen", "g")]
    gold_post, gold_pre = z["post_logits_lastL"][0], z["pre_logits_lastL"][0]
    Lg = gold_post.shape[0]
    post = be.debug["post_logits"][r0 + L - Lg:r0 + L].float().cpu().numpy()
    pre = be.debug["pre_logits"][r0 + L - Lg:r0 + L].float().cpu().numpy()
    e_pre = float(np.abs(pre - gold_pre).max() / np.abs(gold_pre).max())
    e_post = float(np.abs(post - gold_post).max() / np.abs(gold_post).max())
    print(mode, "real dims, batched MEND_VL: pre-edit logits rel err %.2e, post-edit %.2e" % (e_pre, e_post))
    tol = 1e-3 if mode == "fp32" else 2e-2
    assert e_pre < tol and e_post < tol
    assert float(np.abs(post - pre).max()) > 10 * tol * float(np.abs(gold_pre).max()) or mode == "bf16"    # the edit moved the logits
