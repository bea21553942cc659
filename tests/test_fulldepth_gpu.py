"""Depth-compounded bf16 drift of the BENCHMARKED configuration, bounded inside the suite: BLIP-2-OPT-2.7B at FULL depth (39 ViT +
12 Q-Former + 32 OPT layers, synthetic weights of bench.py's headline recipe), two EVQA-shaped cycles through BatchedEditEval in the
engine's fp32 ("faithful": exact-fp32 MFMA, fp32 weights) mode and in its bf16 (benchmark) mode.

This is a SELF-comparison (the same engine in two compute modes), not a parity claim against the reference: the fp32 mode is held to
the reference by the 2-layer goldens (tests/test_realdim_batched_gpu.py, 2e-5) and, at full depth, by bench.py's `parity` block
(the oracle cycle of the cpu_baseline leg needs ~130 s of 64 host cores, too long for the suite)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

LOC = ["text_loc", "t3i3", "t1i4", "t2i4", "t1i2", "t1i3", "t2i1", "t2i2", "t3i1"]


@pytest.fixture(scope="module")
def runs():
    from concurrent.futures import ThreadPoolExecutor
    import devqa_amd  # noqa: F401
    from devqa_amd import blip2_spec
    from devqa_amd.batched import BatchedEditEval, copy_sample
    from devqa_amd.editor.vllm_editors.ft_vl.ft_vl import FTvl, FTvlConfig
    from devqa_amd.editor.vllms_for_edit.blip2.blip2 import BLIP2OPTForEdit
    from devqa_amd.editor.vllms_for_edit.blip2.modeling import Blip2Native
    from devqa_amd.synth import IdTokenizer, evqa_cycles, param_init, synth_image_u8
    cfg, dev, seed = blip2_spec.BLIP2_OPT_2_7B, "cuda:0", 20251121
    models = {m: Blip2Native(cfg, dev, m) for m in ("fp32", "bf16")}
    names = list(models["fp32"]._shapes.keys())
    with ThreadPoolExecutor(16) as ex:
        futs = {n: ex.submit(param_init, n, models["fp32"]._shapes[n], seed, "opt") for n in names}
        for n in names:
            arr = torch.from_numpy(futs.pop(n).result())
            for m in models.values():
                m.load_named_tensors(lambda _n, a=arr: a, names=[n], refresh=False)
    out = {}
    for mode, model in models.items():
        model.refresh_derived(force=True)
        vllm = BLIP2OPTForEdit(None, dev, model=model, tokenizer=IdTokenizer())
        proc = vllm.image_processor
        cyc = evqa_cycles(2, cfg["text_config"]["vocab_size"], seed + 1,
                          lambda s, tag: torch.from_numpy(proc(synth_image_u8(s, tag, 224, seed))).to(dev))
        ft = FTvlConfig(edit_model_name="blip2-opt-2.7b", rewrite_module_tmp="language_model.model.decoder.layers.{}.fc2.weight",
                        layers=[31], num_steps=25, lr=1e-3, weight_decay=0, norm_constraint=False, batch_size=1)
        be = BatchedEditEval(FTvl(vllm, ft, dev), cycles_per_batch=2)
        be.keep_debug = True
        res, _ = be.run_batch([copy_sample(c) for c in cyc], cyc)
        torch.cuda.synchronize()
        out[mode] = dict(rows=be.debug["rows"], pre=be.debug["pre_logits"].float().cpu(), post=be.debug["post_logits"].float().cpu(),
                         losses=np.array(be.last_losses), steps=np.array(be.last_steps), res=res)
        del be, vllm
    models.clear()
    torch.cuda.empty_cache()
    return out


def test_fulldepth_bf16_vs_fp32_mode(runs):
    a, b = runs["fp32"], runs["bf16"]
    assert a["rows"] == b["rows"]
    worst = {"pre": 0.0, "post": 0.0}
    n_rows = n_agree = n_dec = n_dec_ok = 0
    for plist in a["rows"]:
        for kind, name, row0, L in plist:
            for phase in (("pre", "post") if kind == "loc" else ("post",)):
                ref, got = a[phase][row0:row0 + L], b[phase][row0:row0 + L]
                scale = float(ref.abs().max())
                worst[phase] = max(worst[phase], float((got - ref).abs().max()) / scale)
                top2 = ref.topk(2, dim=1).values
                dec = (top2[:, 0] - top2[:, 1]) > 2e-2 * scale
                ok = got.argmax(1) == ref.argmax(1)
                n_rows += L
                n_agree += int(ok.sum())
                n_dec += int(dec.sum())
                n_dec_ok += int((ok & dec).sum())
    loss_err = 0.0
    for e in range(2):
        m = int(min(a["steps"][e], b["steps"][e]))
        ref = a["losses"][e, :m]
        loss_err = max(loss_err, float((np.abs(b["losses"][e, :m] - ref) / np.maximum(ref, 1.0)).max()))
    print("full depth 39/12/32, bf16 mode vs fp32 mode: label-row logits rel err pre %.3g post %.3g; per-step loss err %.3g; "
          "steps fp32 %s bf16 %s; argmax rows %d/%d (%d/%d where the fp32 margin > 2e-2 x scale)"
          % (worst["pre"], worst["post"], loss_err, a["steps"].tolist(), b["steps"].tolist(), n_agree, n_rows, n_dec_ok, n_dec))
    # Measured on MI355X (round 3): logits 1.12e-2 (pre) / 1.14e-2 (post), per-step loss 1.9e-2 of max(loss, 1); against the ORACLE the
    # same engine measured 0.99e-2 / 0.88e-2 / 1.7e-3 on another cycle (bench.py `parity`).  north_star's bf16 bar is 1e-2: at full depth the
    # engine sits AT it, not under it, and cannot do better with bf16 operands -- tools/debug/bf16_drift.py: every GEMM rounds weights and
    # activations to 8 mantissa bits (rms 1.6e-3 per GEMM output), the roundings are independent and accumulate like a random walk over
    # 39 + 12 + 32 layers: the vision tower alone gives 7.3e-3 rms on the image tokens, the decoder alone 5.4e-3 on the logits, together
    # 0.9e-2 rms; the fp32 mode of the same engine reproduces the oracle to < 5e-6.  The bars below hold the measured values with a 1.3x
    # margin (different images / prompts move them by ~15 %).
    assert worst["pre"] < 1.5e-2
    assert worst["post"] < 1.5e-2
    assert loss_err < 2.5e-2
    assert n_dec_ok == n_dec
    assert all(abs(int(x) - int(y)) <= 1 for x, y in zip(a["steps"], b["steps"]))


def test_fulldepth_results_agree(runs):
    """acc of the 2 x 12 probes: equal wherever no label row of the probe sits inside the bf16 error band (checked row by row above);
    at most one probe of the 24 may differ."""
    def flat(res):
        out = []
        for r in res:
            out.append(round(r["reliability"][0]["acc"], 4))
            out += [round(r["generality"][k][0]["acc"], 4) for k in ("text_rephrase", "image_rephrase")]
            out += [round(r["locality"][k][0]["acc"], 4) for k in LOC]
        return out
    fa, fb = flat(runs["fp32"]["res"]), flat(runs["bf16"]["res"])
    same = sum(x == y for x, y in zip(fa, fb))
    print("full depth: probes with equal acc in both modes: %d/24" % same)
    assert same >= 23
