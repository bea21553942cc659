"""GPU: MEND_VL edit path on the HIP engine against goldens produced by the reference's own MENDvl
(tools/make_goldens_mend.py): hooked x / delta, transformed factors, delta weights (single edit, sequential running
mean, text batch), post-edit / restored logits, logit_KL rows, evaluator results."""
import json
import os
from copy import deepcopy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = {"fp32": dict(x=1e-4, delta=1e-3, xt=1e-3, dt=1e-3, dw=1e-3, logits=1e-3),
       "bf16": dict(x=5e-2, delta=1e-1, xt=6e-2, dt=1e-1, dw=1e-1, logits=1e-1)}


@pytest.fixture(scope="module", params=["fp32", "bf16"])
def mend(gold_dir, request):
    import devqa_amd  # noqa: F401
    from devqa_amd.editor.vllms_for_edit.blip2.blip2 import BLIP2OPTForEdit
    from devqa_amd.editor.vllm_editors.mend_vl.mend_vl import MENDvl, MENDvlConfig
    vllm = BLIP2OPTForEdit(os.path.join(gold_dir, "tiny_blip2"), "cuda:0", dtype=request.param)
    cfg = MENDvlConfig.from_yaml(os.path.join(gold_dir, "tiny_mend_cfg.yaml"))
    ed = MENDvl(vllm, cfg, "cuda:0", ckpt_path=os.path.join(gold_dir, "tiny_mend_ckpt.pt"))
    j = json.load(open(os.path.join(gold_dir, "tiny_mend_goldens.json")))
    z = np.load(os.path.join(gold_dir, "tiny_mend_goldens.npz"))
    return vllm, ed, j, z, TOL[request.param], request.param


def _rel(a, g):
    return float(np.abs(a - g).max() / max(np.abs(g).max(), 1e-30))


def _check(ed, z, tag, tol):
    worst = {}
    for i, m in enumerate(ed.modules):
        got = dict(ed.last[m["name"]])
        got["dw"] = ed.delta_weight(m["name"])
        for key in ("x", "delta", "xt", "dt", "dw"):
            g = z["%s_%s_%d" % (tag, key, i)]
            a = got[key].float().cpu().numpy()
            assert a.shape == g.shape, (tag, key, i, a.shape, g.shape)
            if key == "x":   # right-padded rows of a ragged batch hold don't-care values (zero gradient: dropped by the
                keep = (z["%s_delta_%d" % (tag, i)] != 0).any(-1)   # nz rule, auxiliary_networks.py:118-120)
                a, g = a[keep], g[keep]
            worst[key] = max(worst.get(key, 0.0), _rel(a, g))
    print(tag, {k: "%.2e" % v for k, v in worst.items()})
    for key, v in worst.items():
        assert v < tol[key], (tag, key, v)


def test_mend_edits(mend, in_gold_dir):
    vllm, ed, j, z, tol, mode = mend
    pr = j["probe"]

    def logits():
        (x, vt), y, m = vllm.prompts_imgs_target_to_xym([pr["prompt"]], [pr["image"]], [pr["target"]])
        return vllm.get_llm_outpt(x, vt).logits.float().cpu().numpy()
    ed.restore_to_original_model()
    assert _rel(logits(), z["pre_logits"]) < tol["logits"] * 0.2
    a, b, c = j["cases"]
    ed.edit_one_piece(deepcopy(a["requests"][0]))
    _check(ed, z, "a", tol)
    e = _rel(logits(), z["a_post_logits"])
    print("a post logits", e)
    assert e < tol["logits"]
    ed.edit_one_piece(deepcopy(b["requests"][1]))
    _check(ed, z, "b", tol)
    e = _rel(logits(), z["b_post_logits"])
    print("b post logits", e)
    assert e < tol["logits"]
    ed.restore_to_original_model()
    ed.edit_batch(deepcopy(c["requests"]))
    _check(ed, z, "c", tol)
    assert _rel(logits(), z["c_post_logits"]) < tol["logits"]
    ed.restore_to_original_model()
    assert _rel(logits(), z["restored_logits"]) < tol["logits"] * 0.2


def test_logit_kl_rows(mend):
    from devqa_amd import lib
    vllm, ed, j, z, tol, mode = mend
    l1, l2, mk = torch.from_numpy(z["kl_l1"]).cuda(), torch.from_numpy(z["kl_l2"]).cuda(), torch.from_numpy(z["kl_mask"]).cuda()
    L = mk.shape[1]
    a = l1[:, -L:].reshape(-1, l1.shape[-1]).contiguous()
    b = l2[:, -L:].reshape(-1, l2.shape[-1]).contiguous()
    kl = lib.logit_kl_rows(a, b).view(mk.shape)
    tot = float((kl * mk).sum())
    assert abs(tot - j["kl_sum"]) < 1e-4 * abs(j["kl_sum"])
    assert abs(tot / float(mk.sum()) - j["kl"]) < 1e-4 * abs(j["kl"])
    assert abs(float(vllm.logit_KL_loss(l1, l2, mk)) - j["kl"]) < 1e-4 * abs(j["kl"])


def test_mend_evaluator(mend, in_gold_dir, gold_dir, tmp_path):
    from devqa_amd.dataset.vllm import BaseVLLMEditData
    from devqa_amd.evaluation.vllm_editor_eval import VLLMEditorEvaluation
    vllm, ed, j, z, tol, mode = mend
    rec = json.load(open(os.path.join(gold_dir, "evqa8_records.json")))

    class Data(BaseVLLMEditData):
        def dataset_name(self):
            return "EVQA"
    ed.restore_to_original_model()
    data = Data(deepcopy(rec["records"][:4]), deepcopy(rec["records"][:4]))
    res = VLLMEditorEvaluation(ed, data, "EVQA", str(tmp_path)).evaluate_sequential_edit(1, False, None)
    gold = j["results_sen1"]
    n = same = 0
    for rs, gs in zip(res, gold):
        r, g = rs[0], gs[0]
        pairs = [(r["reliability"][0], g["reliability"][0])]
        for sec in ("generality", "locality"):
            for sub in g[sec]:
                pairs.append((r[sec][sub][0], g[sec][sub][0]))
        for a, b in pairs:
            n += 1
            same += int(round(a["acc"], 4) == round(b["acc"], 4) and a["predict_after_edit"] == b["predict_after_edit"])
    print(mode, "evaluator == golden %d/%d" % (same, n))
    assert n == 48
    assert same == 48 if mode == "fp32" else same >= 36


# ---- true OPT-2.7B layer dims (d 2560, FFN 10240, head dim 80, V 50272; hyper-network 12800 -> rank 1920) --------
RD_TOL = {"fp32": dict(rowsum=2e-3, fac=2e-3, dw=2e-3, logits=1e-3), "bf16": dict(rowsum=6e-2, fac=2e-2, dw=2e-2, logits=2e-2)}


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_mend_realdim(gold_dir, in_gold_dir, mode):
    import devqa_amd  # noqa: F401
    from transformers import AutoTokenizer
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
    tol = RD_TOL[mode]
    pr = j["probe"]

    def logits():
        (x, vt), y, m = vllm.prompts_imgs_target_to_xym([pr["prompt"]], [pr["image"]], [pr["target"]])
        return vllm.get_llm_outpt(x, vt).logits[:, -y.shape[1]:].float().cpu().numpy()
    assert _rel(logits(), z["pre_logits_lastL"]) < tol["logits"]
    ed.edit_one_piece(deepcopy(j["request"]))
    for i, m in enumerate(ed.modules):
        got = ed.last[m["name"]]
        dw = ed.delta_weight(m["name"]).double().cpu().numpy()
        e = {"x": _rel(got["x"].double().sum(-1).cpu().numpy(), z["x_rowsum_%d" % i]) ,
             "dmax": _rel(got["delta"].abs().max(-1).values.cpu().numpy(), z["delta_absmax_%d" % i]),
             "xt": _rel(got["xt"][:, :64].cpu().numpy(), z["xt_slice_%d" % i]),
             "dt": _rel(got["dt"][:, :64].cpu().numpy(), z["dt_slice_%d" % i]),
             "dw": _rel(dw[:64, :64], z["dw_slice_%d" % i]),
             "dwn": abs(np.sqrt((dw ** 2).sum()) - z["dw_stats_%d" % i][0]) / z["dw_stats_%d" % i][0]}
        print(mode, m["name"][-12:], {k: "%.2e" % v for k, v in e.items()})
        assert got["xt"].shape[0] == z["xt_slice_%d" % i].shape[0]
        assert e["x"] < tol["rowsum"] and e["dmax"] < tol["fac"] and e["xt"] < tol["fac"] and e["dt"] < tol["fac"]
        assert e["dw"] < tol["dw"] and e["dwn"] < tol["dw"]
    e = _rel(logits(), z["post_logits_lastL"])
    print(mode, "post-edit logits rel err %.2e" % e)
    assert e < tol["logits"]
    ed.restore_to_original_model()
    assert _rel(logits(), z["pre_logits_lastL"]) < tol["logits"]


# ---- training step (train_a_batch) against the reference's MENDvl.train_a_batch (tools/make_goldens_mend.py --train) ----
TR_TOL = {"fp32": dict(loss=2e-4, grad=3e-3, state=1e-4), "bf16": dict(loss=2e-2, grad=3e-2, state=None)}
# bf16: the tiny fixture's edit moves the logits by ~10, so element-wise gradients carry bf16 noise; the bar is on each
# gradient tensor's direction (cosine >= 0.97) and on the total norm (3 %)


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_mend_training_steps(gold_dir, in_gold_dir, mode):
    import devqa_amd  # noqa: F401
    from devqa_amd.editor.vllms_for_edit.blip2.blip2 import BLIP2OPTForEdit
    from devqa_amd.editor.vllm_editors.mend_vl.mend_vl import MENDvl, MENDvlConfig
    j = json.load(open(os.path.join(gold_dir, "tiny_mend_train_goldens.json")))
    z = np.load(os.path.join(gold_dir, "tiny_mend_train_goldens.npz"))
    rec = json.load(open(os.path.join(gold_dir, "evqa8_records.json")))["records"]
    vllm = BLIP2OPTForEdit(os.path.join(gold_dir, "tiny_blip2"), "cuda:0", dtype=mode)
    cfg = MENDvlConfig.from_yaml(os.path.join(gold_dir, "tiny_mend_cfg.yaml"))
    cfg.aux_model.lr, cfg.edit_lr_lr = j["aux_lr"], j["edit_lr_lr"]
    ed = MENDvl(vllm, cfg, "cuda:0", ckpt_path=os.path.join(gold_dir, "tiny_mend_ckpt.pt"))
    ed.set_train(True)
    tol = TR_TOL[mode]
    for si, g in enumerate(j["steps"]):
        batch = ed.organize_batch_data([deepcopy(rec[g["sample"]])])
        loss, log = ed.train_a_batch(batch)
        print(mode, si, "loss %.5f (ref %.5f) grad-norm %.4f (ref %.4f)" % (loss, g["loss"], log["Grad-Norm"], g["log"]["Grad-Norm"]))
        assert abs(loss - g["loss"]) < tol["loss"] * abs(g["loss"])
        assert abs(log["Reliability loss"] - g["log"]["Reliability loss"]) < tol["loss"] * 5 * abs(g["log"]["Reliability loss"])
        for k, v in g["log"]["Locality loss"].items():
            assert abs(log["Locality loss"][k] - v) < tol["loss"] * 5 * max(abs(v), 1e-2), (k, log["Locality loss"][k], v)
        for k, v in g["log"]["Generality loss"].items():
            assert abs(log["Generality loss"][k] - v) < tol["loss"] * 5 * abs(v)
        assert abs(log["Grad-Norm"] - g["log"]["Grad-Norm"]) < tol["grad"] * g["log"]["Grad-Norm"]
        worst = 0.0
        for n, gr in ed.last_grads.items():
            if n == "edit_lrs":
                gold = np.array([z["s%d_grad_edit_lrs.%d" % (si, i)] for i in range(len(ed.modules))]).reshape(-1)
            else:
                gold = z["s%d_grad_aux_models.%s" % (si, n)]
            got = gr.cpu().numpy().reshape(gold.shape)
            e = _rel(got, gold)
            worst = max(worst, e)
            if mode == "fp32":
                assert e < tol["grad"], (si, n, e)
            elif np.abs(gold).max() > 0:
                cos = float((got * gold).sum() / (np.linalg.norm(got) * np.linalg.norm(gold) + 1e-30))
                assert cos > 0.97, (si, n, cos)
        print(mode, si, "worst gradient rel err %.2e" % worst)
        if tol["state"] is not None:
            for n in ed._trainable():
                gold = z["s%d_state_aux_models.%s" % (si, n)]
                assert np.abs(ed.aux[n].cpu().numpy() - gold).max() < 2e-5 + tol["state"] * np.abs(gold).max(), (si, n)
            for i in range(len(ed.modules)):
                assert abs(float(ed.lr_t[i]) - float(z["s%d_state_edit_lrs.%d" % (si, i)])) < 2e-6
            for key in ("(40, 80)", "(80, 40)"):
                for leaf in ("u_mean", "u_std", "v_mean", "v_std", "k"):
                    gold = z["s%d_state_aux_models.%s.%s" % (si, key, leaf)]
                    assert _rel(ed.aux["%s.%s" % (key, leaf)].cpu().numpy(), gold) < 1e-3, (si, key, leaf)
    # checkpoint round trip in the reference's layout
    ed.save_ckpt_dir = "/tmp/devqa_mend_ckpt"
    os.makedirs(ed.save_ckpt_dir, exist_ok=True)
    ed.save_ckpt(2, 1, 0.0, 0.0)               # the reference's signature (base.py:237): one file named `Best`
    path = os.path.join(ed.save_ckpt_dir, "Best")
    ck = torch.load(path, map_location="cpu", weights_only=True)
    assert set(ck) == {"i", "epoch", "loss", "ema_loss", "train_modules", "opt", "lr_scheduler"} and ck["i"] == 2
    assert set(ck["train_modules"]) == {"aux_models", "edit_lrs"} and "(40, 80).mlp.layers.0.u" in ck["train_modules"]["aux_models"]
    assert set(ck["opt"]) == {"state", "param_groups"}     # torch.optim.Adam's layout, as the reference writes it
    assert [len(g_["params"]) for g_ in ck["opt"]["param_groups"]] == [20, 4] and len(ck["opt"]["state"]) == 24
    ed2 = MENDvl(vllm, cfg, "cuda:0", ckpt_path=path)
    assert torch.equal(ed2.aux["(40, 80).mlp.layers.1.v"], ed.aux["(40, 80).mlp.layers.1.v"])
    # the reference's OWN `Best` after the same two steps (tools/make_goldens_mend.py --train): its `opt` is a torch Adam state dict
    # over [aux_models.parameters(), edit_lrs]; mapped onto the HIP moment buffers it must hold the moments this run produced
    ref = torch.load(os.path.join(gold_dir, "tiny_mend_train_best.pt"), map_location="cpu", weights_only=True)
    st = ed.get_a_new_optimizer()
    st.load_state_dict(ref["opt"])
    assert st["t"] == 2 == ed.opt["t"]
    if mode == "fp32":
        for mv in ("m", "v"):
            for k_ in st[mv]:
                a_, b_ = st[mv][k_].cpu(), ed.opt[mv][k_].cpu()
                assert float((a_ - b_).abs().max()) <= 1e-4 * float(a_.abs().max()) + 1e-12, (mv, k_)
        # `train_vllm_editor.py -lkpt <reference Best>`: train_init loads modules AND optimizer state, then training goes on
        from devqa_amd.dataset.vllm import BaseVLLMEditData

        class D(BaseVLLMEditData):
            def dataset_name(self):
                return "EVQA"
        ed3 = MENDvl(vllm, cfg, "cuda:0", for_train=True)
        recs = [deepcopy(r) for r in rec[:2]]
        ed3.train_init(D(recs, deepcopy(recs)), 1, records_dir="/tmp/devqa_mend_resume", load_ckpt_path=os.path.join(gold_dir, "tiny_mend_train_best.pt"),
                       random_seed=1)
        assert ed3.train_i == 2 and ed3.opt["t"] == 2
        assert float((ed3.opt["m"]["(40, 80).mlp.layers.0.v"].cpu() - st["m"]["(40, 80).mlp.layers.0.v"].cpu()).abs().max()) == 0.0
        gold = ref["train_modules"]["aux_models"]["(40, 80).mlp.layers.1.u"]
        assert float((ed3.aux["(40, 80).mlp.layers.1.u"].cpu() - gold).abs().max()) == 0.0
        ed3.data_generator.close()      # stop the prefetch thread first: it shares the (not thread-safe) HF tokenizer with this thread
        loss3, _ = ed3.train_a_batch(ed3.organize_batch_data([deepcopy(rec[2])]))
        assert np.isfinite(loss3) and ed3.opt["t"] == 3


def test_mend_train_from_scratch_then_edit(gold_dir, in_gold_dir, tmp_path):
    """The training loop from the reference's fresh state (u = 0, NaN normalisation buffers): finite losses, the
    statistics become finite after the first step, the Best checkpoint reloads and edits."""
    import devqa_amd  # noqa: F401
    from devqa_amd.editor.vllms_for_edit.blip2.blip2 import BLIP2OPTForEdit
    from devqa_amd.editor.vllm_editors.mend_vl.mend_vl import MENDvl, MENDvlConfig
    rec = json.load(open(os.path.join(gold_dir, "evqa8_records.json")))["records"]
    vllm = BLIP2OPTForEdit(os.path.join(gold_dir, "tiny_blip2"), "cuda:0", dtype="fp32")
    cfg = MENDvlConfig.from_yaml(os.path.join(gold_dir, "tiny_mend_cfg.yaml"))
    cfg.aux_model.lr, cfg.init_edit_lr = 1e-3, 1e-3
    ed = MENDvl(vllm, cfg, "cuda:0", for_train=True)
    with pytest.raises(RuntimeError):
        ed.edit_one_piece(deepcopy(rec[0]["requests"][0]))       # untrained: NaN buffers (auxiliary_networks.py:99-105)
    logs = []
    best = str(tmp_path / "Best")
    ema = ed.train_loop([deepcopy(r) for r in rec[:3]], total_epochs=2, batch_size=1, save_ckpt_path=best, seed=3,
                   log_fn=lambda i, d: logs.append(d["Loss"]))
    assert len(logs) == 6 and all(np.isfinite(l) for l in logs) and np.isfinite(ema)
    for key in ("(40, 80)", "(80, 40)"):
        assert bool(torch.isfinite(ed.aux[key + ".u_std"]).all()) and float(ed.aux[key + ".k"]) > 2
    assert float(ed.aux["(40, 80).mlp.layers.0.u"].abs().max()) > 0        # u left its zero init
    ed2 = MENDvl(vllm, cfg, "cuda:0", ckpt_path=best)
    ed2.edit_one_piece(deepcopy(rec[0]["requests"][0]))
    assert ed2.delta_weight(ed2.modules[0]["name"]) is not None
    ed2.restore_to_original_model()


def test_mend_train_prefetch_equals_serial(gold_dir, in_gold_dir):
    """ParallelDataset's producer thread on a second HIP stream (image encodes + embeddings of the next batches under the
    training step) must not change a single loss: same seed, same batches, same values as with the producer kept one
    batch ahead on the consumer's own thread order (prefetch=False)."""
    import devqa_amd  # noqa: F401
    from devqa_amd.editor.vllms_for_edit.blip2.blip2 import BLIP2OPTForEdit
    from devqa_amd.editor.vllm_editors.mend_vl.mend_vl import MENDvl, MENDvlConfig
    rec = json.load(open(os.path.join(gold_dir, "evqa8_records.json")))["records"]
    runs = []
    for prefetch in (True, False):
        vllm = BLIP2OPTForEdit(os.path.join(gold_dir, "tiny_blip2"), "cuda:0", dtype="fp32")
        cfg = MENDvlConfig.from_yaml(os.path.join(gold_dir, "tiny_mend_cfg.yaml"))
        cfg.aux_model.lr, cfg.init_edit_lr = 1e-3, 1e-3
        ed = MENDvl(vllm, cfg, "cuda:0", for_train=True)
        logs = []
        ed.train_loop([deepcopy(r) for r in rec[:5]], total_epochs=3, batch_size=2, seed=11, log_fn=lambda i, d: logs.append(d["Loss"]),
                 data_buffer_size=4, prefetch=prefetch)
        runs.append((logs, {k: v.clone() for k, v in ed.aux.items()}))
    (la, sa), (lb, sb) = runs
    assert len(la) == len(lb) == 9 and all(np.isfinite(la))      # ceil-less: 3 passes x (5 samples / batches of 2, tails carried over)
    np.testing.assert_allclose(la, lb, rtol=1e-5, atol=1e-6)
    for k in sa:
        if sa[k].dtype.is_floating_point:
            torch.testing.assert_close(sa[k], sb[k], rtol=1e-4, atol=1e-6, equal_nan=True)
