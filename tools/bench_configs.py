#!/usr/bin/env python3
"""BASELINE.json configs #3-#5 at FULL model dims on one MI355X (random weights from the seeded numpy recipes, text
prompts from the committed EVQA-shaped records, the tiny fixtures' tokenizers): not the headline metric (bench.py), a
check that the other model families / editors run at scale, with their throughput.

  python tools/bench_configs.py llava_ft      # LLaVA-1.5-7B + FT_VL, batched engine
  python tools/bench_configs.py blip2_mend    # BLIP-2-OPT-2.7B + MEND_VL (layers 29-31, hyper-network 12800 -> 1920)
  python tools/bench_configs.py minigpt4_ike  # MiniGPT-4 (Vicuna-7B) + IKE_VL (k = 32 over a 15000 x 384 corpus)
"""
import json
import os
import sys
import time
import zlib
from concurrent.futures import ThreadPoolExecutor
from copy import deepcopy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import devqa_amd  # noqa: E402,F401
from devqa_amd.synth import mend_aux_init, param_init  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
DEV = "cuda:0"


def fill(model, seed, style, threads=16):
    names = list(model._shapes.keys())
    with ThreadPoolExecutor(threads) as ex:
        futs = {n: ex.submit(param_init, n, model._shapes[n], seed, style) for n in names}
        for n in names:
            arr = futs.pop(n).result()
            model.load_named_tensors(lambda _n, a=arr: torch.from_numpy(a), names=[n], refresh=False)
    model.refresh_derived(force=True)


def records(n):
    rec = json.load(open(os.path.join(GOLD, "evqa8_records.json")))["records"]
    out = []
    for i in range(n):
        r = deepcopy(rec[i % len(rec)])
        out.append(r)
    return out


class Data:
    pass


def distinct_records(n, image_size, seed=11):
    """n records with DISTINCT images (pre-processed pixel tensors resident in HBM, as bench.py passes them) and distinct prompts, so
    that the batched evaluator's image / sequence de-duplication finds nothing to share across cycles (records(n) repeats 8 records:
    fine for one batch of 8, meaningless for larger ones)."""
    base = records(n)
    g = torch.Generator(device="cpu").manual_seed(seed)
    words = ["red", "blue", "green", "seven", "north", "table", "glass", "river", "stone", "cloud", "paper", "light", "horse", "train"]
    cache = {}

    def img(tag):
        if tag not in cache:
            cache[tag] = torch.randn(3, image_size, image_size, generator=g).to(DEV)
        return cache[tag]

    def fix(item, i, k):
        if item.get("image") is not None:
            item["image"] = img((i, item["image"]))
        item["prompt"] = "%s %s %s" % (words[(i * 7 + k) % len(words)], words[(i // len(words) + 3 * k) % len(words)], item["prompt"])
    for i, r in enumerate(base):
        for k, q in enumerate(r["requests"]):
            fix(q, i, k)
        for sec in ("generality", "locality"):
            for k, name in enumerate(r[sec]):
                for q in r[sec][name]:
                    fix(q, i, 10 + k)
    return base


MFMA_PEAK_TFLOPS = 2500.0


def a_min_tflop(family, text_tokens=16):
    """SURVEY.md 8(d)'s deduplicated algorithmic work of ONE edit+eval cycle for the other model families (same rule as the
    BLIP-2 A_min = 3.81): 4 unique image encodes, the decoder over 4 image prefixes + 12 probe texts, 25 FT steps and 21 probe
    tails on the label rows.  FLOPs = 2 per MAC."""
    if family == "llava":       # CLIP-L/336 366 GFLOP/image + projector 24; Vicuna-7B: 32 x 2 x (4 d^2 + 3 d f) per token position
        d, f, layers, img_tok, vocab = 4096, 11008, 32, 576, 32064
        vision = 4 * (366.0 + 24.0) * 1e9
    elif family == "minigpt4":  # EVA ViT-g 520.7 + Q-Former 12.7 + llama_proj 0.2 GFLOP per image; 32 query tokens
        d, f, layers, img_tok, vocab = 4096, 11008, 32, 32, 32000
        vision = 4 * (520.7 + 12.7 + 0.2) * 1e9
    elif family == "blip2":
        return 3.81
    else:
        raise ValueError(family)
    pos = 4 * img_tok + 12 * text_tokens
    per_tok = layers * 2.0 * (4 * d * d + 3 * d * f)
    attn = layers * 4.0 * d * (4 * img_tok * img_tok / 2 + 12 * text_tokens * (img_tok + text_tokens / 2))
    tails = (25 * 2 + 21) * 3 * 2.0 * (d * f + d * vocab)
    return (vision + pos * per_tok + attn + tails) / 1e12


def measured(fn):
    """Run fn() with the library's slot profiler on (HIP events around every instrumented launch) -> (fn's result, wall seconds,
    {gemm_ms, gemm_tflop, gemm_launches, attention_ms, ft_ms, ln_ms, dropped})."""
    from devqa_amd import lib
    torch.cuda.synchronize()
    lib.profile(1)
    t0 = time.time()
    out = fn()
    torch.cuda.synchronize()
    dt = time.time() - t0
    lib.profile(0)
    prof = [lib.profile_read(i) for i in range(4)]
    g_ms, g_fl, g_n = (sum(p_[j] for p_ in prof) for j in range(3))
    return out, dt, {"gemm_ms": g_ms, "gemm_tflop": g_fl / 1e12, "gemm_launches": int(g_n),
                     "attention_ms": lib.profile_read(lib.PROF_ATTENTION)[0], "ft_ms": lib.profile_read(lib.PROF_FT_ADAMW)[0],
                     "ln_ms": lib.profile_read(lib.PROF_LAYERNORM)[0], "cosine_ms": lib.profile_read(lib.PROF_COSINE)[0],
                     "dropped": lib.profile_dropped()}


def config_line(name, n_cycles, dt, prof, a_min, extra=None):
    """One entry of bench.py's `configs` object.  `roofline`: the dominant kernel family of every config is the bf16 GEMM
    (gemm_bf16_pp_kernel and its smaller-tile siblings); achieved = executed 2MNK FLOPs / summed HIP-event time of the launches,
    scaled down by A_min / executed when more than the deduplicated algorithmic work is executed."""
    exec_pc = prof["gemm_tflop"] / max(n_cycles, 1)
    scale = min(1.0, a_min / exec_pc) if exec_pc > 0 else 0.0
    ach = prof["gemm_tflop"] * scale / (prof["gemm_ms"] / 1e3) if prof["gemm_ms"] > 0 else 0.0
    inst_ms = prof["gemm_ms"] + prof["attention_ms"] + prof["ft_ms"] + prof["ln_ms"] + prof["cosine_ms"]
    out = {"workload": name, "cycles": n_cycles, "cycles_per_s": round(n_cycles / dt, 3), "ms_per_cycle": round(1e3 * dt / n_cycles, 2),
           "a_min_tflop_per_cycle": round(a_min, 2), "executed_gemm_tflop_per_cycle": round(exec_pc, 2),
           "path_frac_of_mfma_peak": round(min(a_min, exec_pc if exec_pc > 0 else a_min) * n_cycles / dt / MFMA_PEAK_TFLOPS, 4),
           "roofline": {"bound": "mfma", "kernel": "gemm_bf16 family (pp 256x256 / glds / tn)", "achieved": round(ach, 1), "peak": MFMA_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "launches": prof["gemm_launches"],
                        "gemm_time_frac": round(prof["gemm_ms"] / 1e3 / dt, 3)},
           "instrumented_kernel_time_frac": round(inst_ms / 1e3 / dt, 3), "profile_dropped": prof["dropped"]}
    if extra:
        out.update(extra)
    return out


def run_eval(editor, n, batched, out_dir="/tmp/devqa_bench_cfg", distinct_image_size=None):
    from devqa_amd.dataset.vllm import BaseVLLMEditData
    from devqa_amd.evaluation.vllm_editor_eval import VLLMEditorEvaluation

    class D(BaseVLLMEditData):
        def dataset_name(self):
            return "EVQA"
    os.chdir(GOLD)
    mk = (lambda: distinct_records(n, distinct_image_size)) if distinct_image_size else (lambda: records(n))
    ev = VLLMEditorEvaluation(editor, D(mk(), mk()), "EVQA", out_dir)
    ev.evaluate_sequential_edit(1, False, None, batched=batched, save=False)   # warm-up (kernel load, caches)
    torch.cuda.synchronize()
    a, b = mk(), mk()
    torch.cuda.synchronize()
    ev = VLLMEditorEvaluation(editor, D(a, b), "EVQA", out_dir)
    res, dt, prof = measured(lambda: ev.evaluate_sequential_edit(1, False, None, batched=batched, save=False))
    run_eval.last_profile = prof
    run_eval.last_mode = getattr(ev, "last_mode", "generic")
    return len(res) / dt, dt, res


def llava_ft(n=8, per_batch=None):
    """config #3: LLaVA-1.5-7B + FT_VL on the batched engine, n DISTINCT cycles -> one `configs` entry (config_line)"""
    from transformers import AutoTokenizer
    from devqa_amd.llava_spec import LLAVA_1_5_7B
    from devqa_amd.editor.vllms_for_edit.llava.modeling import LlavaNative
    from devqa_amd.editor.vllms_for_edit.llava.llava import LlavaForEdit
    from devqa_amd.editor.vllm_editors.ft_vl.ft_vl import FTvl, FTvlConfig
    t0 = time.time()
    if per_batch:
        os.environ["DEVQA_CYCLES_PER_BATCH"] = str(per_batch)
    cfg7b = dict(LLAVA_1_5_7B, image_token_index=4)   # the stand-in tokenizer (tiny fixture) maps '<image>' to id 4
    model = LlavaNative(cfg7b, DEV, "bf16")
    fill(model, 3, "llava")
    tok = AutoTokenizer.from_pretrained(os.path.join(GOLD, "tiny_llava"))
    vllm = LlavaForEdit(None, DEV, True, model=model, tokenizer=tok)
    cfg = FTvlConfig.from_yaml(os.path.join(ROOT, "de-vqa_amd", "configs", "ft_vl", "llava-v1.5-7b.yaml"))
    ed = FTvl(vllm, cfg, DEV)
    build_s = time.time() - t0
    cps, dt, res = run_eval(ed, n, True, distinct_image_size=336)
    os.environ.pop("DEVQA_CYCLES_PER_BATCH", None)
    return config_line("BASELINE config #3: LLaVA-1.5-7B (CLIP-L/336 + Vicuna-7B dims) + FT_VL on layers.31.mlp.down_proj, batched engine, %d distinct "
                       "synthetic EVQA-shaped cycles in batches of %s, bf16" % (n, per_batch or 16), n, dt, run_eval.last_profile,
                       a_min_tflop("llava"), {"build_s": round(build_s, 1), "reliability_acc_first": res[0][0]["reliability"][0]["acc"]})


def _mend_editor():
    from transformers import AutoTokenizer
    from devqa_amd import blip2_spec
    from devqa_amd.editor.vllms_for_edit.blip2.modeling import Blip2Native
    from devqa_amd.editor.vllms_for_edit.blip2.blip2 import BLIP2OPTForEdit
    from devqa_amd.editor.vllm_editors.mend_vl.mend_vl import MENDvl, MENDvlConfig
    model = Blip2Native(blip2_spec.BLIP2_OPT_2_7B, DEV, "bf16")
    fill(model, 20251121, "opt")
    tok = AutoTokenizer.from_pretrained(os.path.join(GOLD, "tiny_blip2"))
    vllm = BLIP2OPTForEdit(None, DEV, model=model, tokenizer=tok)
    cfg = MENDvlConfig.from_yaml(os.path.join(ROOT, "de-vqa_amd", "configs", "mend_vl", "blip2-opt-2.7b.yaml"))
    D = 2560 + 10240
    shapes = {}
    for key in ("(2560, 10240)", "(10240, 2560)"):
        du, dv = eval(key)
        shapes.update({key + ".u_mean": (du,), key + ".u_std": (du,), key + ".v_mean": (dv,), key + ".v_std": (dv,)})
        for l in range(2):
            q = key + ".mlp.layers.%d." % l
            shapes.update({q + "u": (D, 1920), q + "v": (1920, D), q + "bias": (D,), q + "mode_shift.weight": (3, D),
                           q + "mode_scale.weight": (3, D)})
    with ThreadPoolExecutor(8) as ex:
        aux = dict(zip(shapes, ex.map(lambda kv: torch.from_numpy(mend_aux_init("aux_models." + kv[0], kv[1], 11)), shapes.items())))
    tm = {"aux_models": aux, "edit_lrs": {str(i): torch.tensor(1e-4) for i in range(6)}}
    return MENDvl(vllm, cfg, DEV, train_modules=tm)


def blip2_mend(n=4, batched=None):
    """config #4 on one GPU: BLIP-2-OPT-2.7B + MEND_VL (layers 29-31 fc1/fc2, hyper-network 12800 -> rank 1920), n distinct cycles"""
    t0 = time.time()
    ed = _mend_editor()
    build_s = time.time() - t0
    cps, dt, res = run_eval(ed, n, batched, distinct_image_size=224)
    # per cycle beyond the BLIP-2 A_min: backward through the 3 edited layers on the edit sequence (~2x their forward) and the
    # hyper-network (2 transforms x 2 LRLinear x 2 GEMMs of [T x 12800] . [12800 x 1920]) -- < 0.1 TFLOP, within the figure's rounding
    return config_line("BASELINE config #4 (one GPU's shard): BLIP-2-OPT-2.7B + MEND_VL (decoder layers 29-31 fc1/fc2, hyper-network 12800 -> 1920), "
                       "%d distinct synthetic EVQA-shaped cycles, bf16" % n, n, dt, run_eval.last_profile, a_min_tflop("blip2"),
                       {"build_s": round(build_s, 1), "edit_time_s": res[0][0]["reliability"][0].get("edit_time"),
                        "evaluator": getattr(run_eval, "last_mode", "generic")})


def blip2_mend_train(n=6):
    """One MEND_VL training step (train_a_batch) per sample at full BLIP-2-OPT-2.7B dims, hyper-network 12800 -> 1920."""
    from transformers import AutoTokenizer
    from devqa_amd import blip2_spec
    from devqa_amd.editor.vllms_for_edit.blip2.modeling import Blip2Native
    from devqa_amd.editor.vllms_for_edit.blip2.blip2 import BLIP2OPTForEdit
    from devqa_amd.editor.vllm_editors.mend_vl.mend_vl import MENDvl, MENDvlConfig
    t0 = time.time()
    model = Blip2Native(blip2_spec.BLIP2_OPT_2_7B, DEV, "bf16")
    fill(model, 20251121, "opt")
    tok = AutoTokenizer.from_pretrained(os.path.join(GOLD, "tiny_blip2"))
    vllm = BLIP2OPTForEdit(None, DEV, model=model, tokenizer=tok)
    cfg = MENDvlConfig.from_yaml(os.path.join(ROOT, "de-vqa_amd", "configs", "mend_vl", "blip2-opt-2.7b.yaml"))
    ed = MENDvl(vllm, cfg, DEV, for_train=True)
    ed.set_train(True)
    print("build %.1fs" % (time.time() - t0), flush=True)
    os.chdir(GOLD)
    recs = records(n + 1)
    losses = []
    loss, _ = ed.train_a_batch(ed.organize_batch_data([recs[0]]))      # warm-up
    torch.cuda.synchronize()
    t0 = time.time()
    for r in recs[1:]:
        loss, log = ed.train_a_batch(ed.organize_batch_data([r]))
        losses.append(round(loss, 4))
    torch.cuda.synchronize()
    dt = (time.time() - t0) / n
    print(json.dumps({"config": "BLIP-2-OPT-2.7B + MEND_VL train_a_batch (B = 1: 1 edit + 12 post-edit probes + 9 pre-edit)",
                      "s_per_step": round(dt, 4), "steps_per_s": round(1 / dt, 2), "losses": losses, "grad_norm_last": log["Grad-Norm"]}))
    # the training loop with ParallelDataset's producer thread on a second HIP stream vs. organising each batch in line
    from collections import OrderedDict

    class NoHit(OrderedDict):       # every image is encoded again, as in a real epoch over thousands of distinct images
        def __contains__(self, k):
            return False
    vllm._img_feat_cache = NoHit()
    for prefetch in (False, True):
        stamps = []
        ed.train_loop(recs[1:], total_epochs=3, batch_size=1, seed=1, log_fn=lambda i, d: stamps.append(time.time()), data_buffer_size=4,
                 prefetch=prefetch)
        torch.cuda.synchronize()
        k = len(stamps) // 3                     # skip the first pass
        dt = (stamps[-1] - stamps[k]) / (len(stamps) - 1 - k)
        print(json.dumps({"config": "MENDvl.train loop, prefetch=%s" % prefetch, "s_per_step": round(dt, 4), "steps_per_s": round(1 / dt, 2)}))


def _hash_encode(sentences, dim=384):
    out = np.zeros((len(sentences), dim), np.float32)
    for i, s in enumerate(sentences):
        out[i] = np.random.default_rng(zlib.crc32(s.encode())).standard_normal(dim, dtype=np.float32)
    return out


def _ike_editor(mode="bf16"):
    """MiniGPT-4 (EVA ViT-g + Q-Former + Vicuna-7B dims, synthetic weights) + IKE_VL over a synthetic 15000 x 384 corpus -> (editor, tokenizer, config)"""
    from transformers import AutoTokenizer
    from devqa_amd.minigpt4_spec import MINIGPT4_VICUNA_7B
    from devqa_amd.editor.vllms_for_edit.minigpt4.modeling import MiniGPT4Native
    from devqa_amd.editor.vllms_for_edit.minigpt4.minigpt4 import MiniGPT4ForEdit
    from devqa_amd.editor.vllm_editors.ike_vl.ike_vl import IKEvl, IKEvlConfig
    model = MiniGPT4Native(MINIGPT4_VICUNA_7B, DEV, mode)
    fill(model, 5, "llava")
    tok = AutoTokenizer.from_pretrained(os.path.join(GOLD, "tiny_llava"))
    vllm = MiniGPT4ForEdit(None, DEV, True, model=model, tokenizer=tok)
    N = 15000
    sents = ["New Fact: fact %d is %d\nPrompt: fact %d is %d\n\n" % (i, i * 7 % 13, i, i * 7 % 13) for i in range(N)]
    corpus = {"sentences": sents, "embeddings": np.random.default_rng(1).standard_normal((N, 384), dtype=np.float32)}
    cfg = IKEvlConfig.from_yaml(os.path.join(ROOT, "de-vqa_amd", "configs", "ike_vl", "minigpt-4-vicuna-7b.yaml"))
    return IKEvl(vllm, cfg, DEV, corpus, _hash_encode), tok, cfg


def minigpt4_ike(n=4):
    """config #5 on one GPU: MiniGPT-4 (EVA ViT-g + Q-Former + Vicuna-7B dims) + IKE_VL, k = 32 over a 15000 x 384 corpus"""
    t0 = time.time()
    ed, tok, cfg = _ike_editor()
    build_s = time.time() - t0
    cps, dt, res = run_eval(ed, n, None, distinct_image_size=224)
    # the in-context prefix (k = 32 retrieved demonstrations + the new fact, `demo_rows` token rows) stands BEHIND the image rows of a probe
    # ('<ImageHere>\n' + context + prompt, minigpt4.py:71-72 + ike_vl.with_context), so under causal attention its hidden states depend on
    # the image: the deduplicated minimum holds it once per DISTINCT (image | no image) context of the 12 post-edit probes = 4 images + 1
    # text-only = 5 copies (pre-edit probes carry no context)
    demo_rows = len(tok("".join(ed.retrieve("fact 1 is", "7")))["input_ids"])
    per_row = 32 * 2.0 * (4 * 4096 ** 2 + 3 * 4096 * 11008) + 32 * 4.0 * 4096 * (demo_rows / 2.0 + 32)
    a_min = a_min_tflop("minigpt4") + 5 * demo_rows * per_row / 1e12
    return config_line("BASELINE config #5 (one GPU's shard): MiniGPT-4 (Vicuna-7B dims) + IKE_VL, cosine top-k k=%d over a 15000 x 384 corpus, "
                       "%d distinct synthetic cycles, bf16, generic evaluator with shared-prefix probe packing" % (cfg.k, n), n, dt,
                       run_eval.last_profile, a_min, {"build_s": round(build_s, 1), "icl_prefix_rows": demo_rows})


if __name__ == "__main__":
    r = {"llava_ft": llava_ft, "blip2_mend": blip2_mend, "minigpt4_ike": minigpt4_ike, "blip2_mend_train": blip2_mend_train}[sys.argv[1]](*[int(a) for a in sys.argv[2:]])
    if r is not None:
        print(json.dumps(r))
