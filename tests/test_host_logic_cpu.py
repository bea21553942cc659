"""CPU tests of the host-side mirror of the reference interface (no GPU, no HIP compute)."""
import json
import os

import numpy as np
import pytest

import devqa_amd  # noqa: F401


def test_config_surface_matches_reference_yaml():
    from devqa_amd.editor.vllm_editors.ft_vl.ft_vl import FTvlConfig
    from devqa_amd.utils import get_editor_config_path, get_full_model_name
    assert get_full_model_name("BLIP2") == "blip2-opt-2.7b"
    cfg = FTvlConfig.from_yaml(get_editor_config_path("FT_VL", "blip2"))
    # values of R/configs/ft_vl/blip2-opt-2.7b.yaml:1-8
    assert cfg.to_dict() == {"edit_model_name": "blip2-opt-2.7b", "rewrite_module_tmp": "language_model.model.decoder.layers.{}.fc2.weight",
                             "layers": [31], "num_steps": 25, "lr": 1e-3, "weight_decay": 0, "norm_constraint": False,
                             "batch_size": 1}
    assert type(cfg.norm_constraint) is not float  # YAML `false` disables the clamp (ft_vl.py:135)


def test_param_names_match_hf_blip2(gold_dir):
    """The HF parameter-name contract: our spec enumerates exactly the tensors of the reference-saved checkpoint."""
    from safetensors import safe_open
    from devqa_amd.blip2_spec import param_shapes
    cfg = json.load(open(os.path.join(gold_dir, "tiny_blip2", "config.json")))
    shapes = param_shapes(cfg)
    with safe_open(os.path.join(gold_dir, "tiny_blip2", "model.safetensors"), framework="pt") as f:
        keys = {k: tuple(f.get_slice(k).get_shape()) for k in f.keys()}
    keys.pop("language_model.lm_head.weight", None)
    assert {k: tuple(v) for k, v in shapes.items()} == keys


def test_split_data_drops_incomplete_tail():
    from devqa_amd.evaluation.vllm_editor_eval import VLLMEditorEvaluation
    data = [{"requests": [0]} for _ in range(7)]
    s, ns = VLLMEditorEvaluation.split_data(data, 3)
    assert [len(x) for x in s] == [3, 3] and ns == [3, 3]
    s, ns = VLLMEditorEvaluation.split_data(data, 1)
    assert len(s) == 7


def test_probe_builder_and_evqa_suffixes(gold_dir):
    from devqa_amd.dataset.vllm import build_probes, SUFFIX
    rec = json.load(open(os.path.join(gold_dir, "evqa8_records.json")))
    raw, gold = rec["raw"], rec["records"]
    n = len(raw)
    retrieved = [([raw[(i + 1) % n]["src"], raw[(i + 1) % n]["alt"]], gold[i]["locality"]["t1i2"][0]["image"]) for i in range(n)]
    probes = build_probes(raw, "", retrieved)
    for p, g, d in zip(probes, gold, raw):
        assert list(p["locality"].keys()) == list(g["locality"].keys())
        assert p["locality"]["t3i1"][0]["prompt"] == d["m_loc"]            # image path used as text
        assert p["requests"][0]["prompt"] + SUFFIX == g["requests"][0]["prompt"]
        for name in g["locality"]:
            want = g["locality"][name][0]["prompt"]
            got = p["locality"][name][0]["prompt"] + SUFFIX + ("?" if name == "text_loc" else "")
            assert got == want and p["locality"][name][0]["target"] == g["locality"][name][0]["target"]


def test_pretokenised_probe_rule_matches_string_rule(gold_dir):
    """BatchedEditEval._probe_seq on id lists == the roll/mask/crop rule pinned by golden G1."""
    from devqa_amd.batched import BatchedEditEval
    j = json.load(open(os.path.join(gold_dir, "tiny_goldens.json")))
    from tokenizers import Tokenizer
    tk = Tokenizer.from_file(os.path.join(gold_dir, "tiny_blip2", "tokenizer.json"))
    be = BatchedEditEval.__new__(BatchedEditEval)
    for g in j["g1"]:
        p, t = g["prompt"], g["target"]
        if p[-1] not in " \n" and t[0] not in " \n":
            t = " " + t
        pid = tk.encode(p).ids
        full = tk.encode(p + t).ids
        if full[:len(pid)] != pid:
            continue  # BPE merge across the boundary: rule is defined on strings only
        ids, y, m = be._probe_seq(pid, full[len(pid):])
        assert ids == full and [y] == g["label_ids"] and [m] == g["label_masks"]


def test_synth_cycles_shape():
    from devqa_amd.synth import evqa_cycles
    cyc = evqa_cycles(50, seed=3)
    ks = [len(c["requests"][0]["target_new"]) for c in cyc]
    assert set(ks) <= {1, 2, 3} and ks.count(1) > 25
    for c in cyc:
        assert len(c["locality"]) == 9 and len(c["generality"]) == 2
        assert 8 <= len(c["requests"][0]["prompt"]) <= 28 and c["requests"][0]["prompt"][0] == 2
        assert c["locality"]["text_loc"][0]["image"] is None and c["locality"]["t1i4"][0]["image"] is None
        assert c["locality"]["t2i1"][0]["image"] == c["requests"][0]["image"]


def test_product_never_imports_oracle():
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bad = []
    for dp, _, fs in os.walk(os.path.join(root, "de-vqa_amd")):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle", src, re.M):
                    bad.append(f)
    assert bad == []


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from devqa_amd import lib
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(lib.DevqaError):
        lib.load()


def test_read_res_table(tmp_path):
    """R/read_res.py:10-27: column order and the 1-acc convention for the non-t3 locality probes."""
    import json
    import devqa_amd  # noqa: F401
    from devqa_amd.read_res import COLUMNS, collect
    d = tmp_path / "eval_results" / "ft_vl" / "blip2-opt-2.7b" / "EVQA" / "sequential_edit_1"
    d.mkdir(parents=True)
    loc = {k: {"acc": 0.25 + 0.05 * i} for i, k in enumerate(["text_loc", "t3i3", "t1i4", "t2i4", "t1i2", "t1i3", "t2i1", "t2i2", "t3i1"])}
    json.dump({"total_mean": {"reliability": {"acc": 1.0}, "generality": {"text_rephrase": {"acc": 1.0}, "image_rephrase": {"acc": 1.0}},
                              "locality": loc}}, open(d / "mean_results.json", "w"))
    rows = collect(str(tmp_path / "eval_results"))
    assert rows[0] == COLUMNS and len(rows) == 2
    r = dict(zip(COLUMNS, rows[1]))
    assert (r["model"], r["data"], r["method"]) == ("blip2-opt-2.7b", "EVQA", "ft_vl")
    assert abs(float(r["t1i2"]) - (1 - loc["t1i2"]["acc"])) < 1e-12 and abs(float(r["t3i1"]) - loc["t3i1"]["acc"]) < 1e-12
    assert abs(float(r["text_loc"]) - loc["text_loc"]["acc"]) < 1e-12


def test_parallel_dataset_matches_reference_id_sequences(gold_dir):
    """ParallelDataset: same id batches as the reference's class for the same seed (goldens captured from it by
    tools/make_goldens_parallel_dataset.py), over three consecutive passes; producer errors surface in the consumer."""
    import json
    import os
    import pytest
    from devqa_amd.dataset import ParallelDataset
    cases = json.load(open(os.path.join(gold_dir, "parallel_dataset_ids.json")))
    assert len(cases) == 6
    for c in cases:
        a = c["args"]
        ds = ParallelDataset(a["sample_count"], lambda ids: [int(i) for i in ids], a["batch_size"], a["shuffle"], 4, a["drop_last"],
                             a["random_seed"], True)
        if isinstance(a["batch_size"], int):
            assert len(ds) == c["len"]
        for want in c["passes"]:
            got = [[d, n] for d, n in ds]
            assert got == want, (a, got, want)
        ds.close()
    ds = ParallelDataset(4, lambda ids: (_ for _ in ()).throw(ValueError("boom")), 2, False, 2, False, 0, False)
    with pytest.raises(RuntimeError):
        next(iter(ds))
    with pytest.raises(Exception):
        ParallelDataset(4, lambda ids: ids, 0)


def test_plan_shared_prefixes():
    """Grouping rule of the batched probe path: common leading rows (by the wrappers' row identities) are shared, never into a
    member's label window, never below the minimum length, never without identities."""
    import torch
    from devqa_amd.evaluation.vllm_editor_eval import VLLMEditorEvaluation as E

    def item(i, keys, L, with_keys=True):
        return (i, torch.zeros(len(keys), 4), torch.zeros(1, L, dtype=torch.long), None, list(keys) if with_keys else None)
    img_a = [("img", "a.png", j) for j in range(8)]
    img_b = [("img", "b.png", j) for j in range(8)]
    ctx = list(range(100, 140))                     # 40 in-context tokens
    items = [item(0, img_a + ctx + [1, 2, 3], 2), item(1, img_a + ctx + [1, 9, 9, 9], 3), item(2, img_b + ctx + [1, 2, 3], 2),
             item(3, img_b + ctx + [7], 1), item(4, ctx + [5, 6], 1), item(5, img_a + ctx + [1, 2, 3], 2, with_keys=False),
             item(6, img_a[:4] + [9, 9], 1)]
    groups, alone = E._plan_shared_prefixes(items, min_share=32)
    g = {tuple(m): n for m, n in groups}
    assert g == {(0, 1): 49, (2, 3): 48}            # a: image + context + the common "1"; b: capped at item 3's label window
    assert alone == [4, 5, 6]
    # identical probes: the prefix stops in front of the label window
    same = [item(0, img_a + ctx + [1, 2, 3], 2), item(1, img_a + ctx + [1, 2, 3], 2)]
    groups, alone = E._plan_shared_prefixes(same, min_share=32)
    assert groups == [([0, 1], 49)] and alone == []


def test_shared_prefix_plan_keys_by_stored_edit_not_by_view_object():
    """probes behind DIFFERENT stored LTE prefixes must never be grouped, probes behind the SAME stored
    prefix must be -- whatever CPython does with the addresses of the temporary views `probe_prefix` returns."""
    import gc
    import torch
    from devqa_amd.evaluation.vllm_editor_eval import VLLMEditorEvaluation as E
    torch.manual_seed(0)
    pool = [torch.randn(1, 40, 8), torch.randn(1, 40, 8), torch.randn(1, 40, 8)]   # stored edits [1, P, d] as LTEvl keeps them
    for trial in range(40):
        hits = [int(h) for h in torch.randint(0, 4, (12,))]      # 3 = retrieval miss (no prefix)
        items = []
        for i, h in enumerate(hits):
            body = torch.randn(6, 8)
            keys = [("tok", i, j) for j in range(6)]              # distinct probe texts
            y = torch.zeros(1, 2, dtype=torch.long)
            if h < 3:
                pfx = pool[h][0]                                  # a NEW view object per call, freed at the next iteration
                e = torch.cat([pfx, body], 0)
                keys = E._prefix_row_keys(pfx) + keys
                del pfx
                gc.collect()
                _ = [torch.empty(1) for _ in range(3)]            # retrieval temporaries that may land on the freed address
            else:
                e = body
            items.append((i, e, y, y, keys))
        groups, alone = E._plan_shared_prefixes(items)
        grouped = {}
        for members, lcp in groups:
            assert lcp == 40
            assert len({hits[m] for m in members}) == 1, (hits, members)     # one stored edit per group
            grouped[hits[members[0]]] = grouped.get(hits[members[0]], 0) + len(members)
        for h in range(3):
            n = hits.count(h)
            assert grouped.get(h, 0) == (n if n >= 2 else 0), (hits, groups)  # every repeat retrieval DOES share
        assert sorted(alone + [m for g, _ in groups for m in g]) == list(range(12))


def test_every_editor_model_pair_has_a_config():
    """load_vllm_editor resolves configs/<editor>/<model>.yaml (R/utils/__init__.py:101-103) for every editor branch it has and
    every model name get_full_model_name maps to: each pair must exist and parse into the editor's config dataclass, so that
    `test_vllm_edit.py -en <editor> -mn <model>` never dies on FileNotFoundError (BASELINE config #5 = ike_vl + minigpt4)."""
    import devqa_amd  # noqa: F401
    from devqa_amd.utils import get_editor_config_path
    from devqa_amd.editor.vllm_editors.ft_vl.ft_vl import FTvlConfig
    from devqa_amd.editor.vllm_editors.ike_vl.ike_vl import IKEvlConfig
    from devqa_amd.editor.vllm_editors.mend_vl.mend_vl import MENDvlConfig
    from devqa_amd.editor.vllm_editors.tp_vl.tp_vl import TPvlConfig
    from devqa_amd.editor.vllm_editors.lte_vl.lte_vl import LTEvlConfig
    classes = {"ft_vl": FTvlConfig, "ike_vl": IKEvlConfig, "mend_vl": MENDvlConfig, "tp_vl": TPvlConfig, "lte_vl": LTEvlConfig}
    full = {"blip2": "blip2-opt-2.7b", "llava": "llava-v1.5-7b", "minigpt4": "minigpt-4-vicuna-7b"}
    for editor, cls in classes.items():
        for short, name in full.items():
            path = get_editor_config_path(editor, short)
            assert os.path.isfile(path), path
            cfg = cls.from_yaml(path)
            assert cfg.edit_model_name == name, (path, cfg.edit_model_name)


def test_torch_hooks_on_the_native_model_fail_loudly():
    """editors may register torch hooks on `vllm.model` sub-modules (mend_vl.py:63-85, tp_vl.py:71-111).  On the native
    model they could never fire (the forward runs in HIP kernels), so registering one raises instead of silently doing nothing."""
    import pytest
    import devqa_amd  # noqa: F401
    from devqa_amd import blip2_spec
    from devqa_amd.editor.vllms_for_edit.blip2.modeling import Blip2Native
    from devqa_amd.utils import find_module
    m = Blip2Native(blip2_spec.scaled_spec(1, 1, 1), "cpu", "fp32")
    fc1 = find_module(m, "language_model.model.decoder.layers.0.fc1")
    for reg in (fc1.register_forward_hook, fc1.register_forward_pre_hook, fc1.register_full_backward_hook, m.register_forward_hook):
        with pytest.raises(NotImplementedError, match="never fire"):
            reg(lambda *a: None)
    # parameters stay reachable by name exactly as before (get_parameter / find_module)
    assert find_module(m, "language_model.model.decoder.layers.0.fc1.weight").shape[0] == m.cfg["text_config"]["ffn_dim"]
