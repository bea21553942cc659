"""A2 / N1: the dynamic-evaluation dataset builder against the REFERENCE's own `EVQA(...)` / `VLKEB(...)` constructors.

tests/golden/dataset/goldens.json holds the probe dicts the reference's `__init_eic_evqa__` + `finds_sim` + suffix rules produced
(tools/make_goldens_dataset.py imported R/dataset/vllm.py; the retrieval corpus and the stand-in sentence encoder come from
tests/retr_common.py on both sides).  CPU: the product's EVQA / VLKEB / build_probes with a retriever whose top-k is the oracle's
float64 brute force.  GPU: the real EmbeddingRetriever -> devqa_cosine_topk, including the "first hit with a different answer, else
the last hit" rule (R/dataset/vllm.py:72-81) and a corpus smaller than `tops`."""
import json
import os

import numpy as np
import pytest

import retr_common as RC

HERE = os.path.dirname(os.path.abspath(__file__))
DS = os.path.join(HERE, "golden", "dataset")
FILES = {"EVQA": "vqa_eval_head.json", "VLKEB": "vlkeb_eval_head.json"}


def _gold():
    return json.load(open(os.path.join(DS, "goldens.json")))


class OracleRetriever:
    """Same interface as devqa_amd.dataset.vllm.EmbeddingRetriever; top-k by the oracle's float64 brute force (CPU)."""

    def __init__(self, corpus):
        self.prompts, self.save_image_path = corpus["prompts"], corpus["images"]
        self.emb = RC.encode(corpus["sentences"])

    def finds_sim_many(self, srcs, trgs, tops=5):
        from oracle import devqa_oracle as O
        idx, _ = O.cosine_topk(self.emb, RC.encode(srcs), tops)
        out = []
        for hit, trg in zip(idx, trgs):
            pick = O.finds_sim_select([int(i) for i in hit], self.prompts, trg)
            out.append((self.prompts[pick], self.save_image_path[pick]))
        return out


@pytest.mark.parametrize("name", ["EVQA", "VLKEB"])
def test_builder_equals_reference_constructor_cpu(name):
    import devqa_amd  # noqa: F401
    from devqa_amd.dataset import vllm as V
    g = _gold()
    cls = {"EVQA": V.EVQA, "VLKEB": V.VLKEB}[name]
    ds = cls(os.path.join(DS, FILES[name]), g[name]["img_root"], g["n"], OracleRetriever(g["corpus"][name]))
    assert ds.dataset_name() == name
    assert ds.data_with_img_path == g[name]["data"]                  # every probe dict, key order included
    assert ds.data_with_img == ds.data_with_img_path and ds.data_with_img is not ds.data_with_img_path
    assert json.dumps(ds.data_with_img_path) == json.dumps(g[name]["data"])
    assert ds.get_data_with_img_path() is ds.data_with_img_path and ds.data is ds.data_with_img
    # the fixture really walks all three finds_sim branches
    branches = {p["branch"] for p in g[name]["picks"]}
    assert {"rank0", "rank1", "last"} <= branches
    d0 = ds.data[0]
    assert d0["locality"]["text_loc"][0]["prompt"].endswith(" The answer is:?")              # suffix then '?' (vllm.py:246-248)
    assert d0["locality"]["t3i1"][0]["prompt"].split(" The answer is:")[0].endswith((".jpg", ".png"))   # t3 is an image PATH (:164)
    assert d0["requests"][0]["prompt"].endswith(" The answer is:") == (name == "EVQA")       # VLKEB: locality prompts only
    # data_n smaller / larger than the file
    assert len(cls(os.path.join(DS, FILES[name]), "r", 3, OracleRetriever(g["corpus"][name])).data) == 3
    assert len(cls(os.path.join(DS, FILES[name]), "r", None, OracleRetriever(g["corpus"][name])).data) == g["n"]


def test_evqa_rejects_non_vqa_file_and_missing_retriever():
    import devqa_amd  # noqa: F401
    from devqa_amd.dataset import vllm as V
    g = _gold()
    with pytest.raises(RuntimeError):
        V.EVQA(os.path.join(DS, FILES["VLKEB"]), "r", 2, OracleRetriever(g["corpus"]["EVQA"]))     # 'vqa' not in basename (:234)
    with pytest.raises(RuntimeError):
        V.EVQA(os.path.join(DS, FILES["EVQA"]), "r", 2, None)


def test_eic_probe_rules(tmp_path):
    """EIC (R/dataset/vllm.py:257-271): EVQA's builder, '?' on text_loc only, no ' The answer is:' (the reference's own constructor
    raises KeyError on `image_loc` at :265, so there is no reference output to pin this against: parity unpinned, rule read off
    the source).  Checked against the EVQA golden probes with the EVQA suffixes stripped."""
    import shutil
    import devqa_amd  # noqa: F401
    from devqa_amd.dataset import vllm as V
    g = _gold()
    path = str(tmp_path / "caption_eval_head.json")
    shutil.copy(os.path.join(DS, FILES["EVQA"]), path)
    ds = V.EIC(path, g["EVQA"]["img_root"], g["n"], OracleRetriever(g["corpus"]["EVQA"]))
    assert ds.dataset_name() == "EIC" and len(ds.data) == g["n"]
    SUF = " The answer is:"
    for d, e in zip(ds.data_with_img_path, g["EVQA"]["data"]):
        assert d["requests"][0]["prompt"] + SUF == e["requests"][0]["prompt"]
        assert d["requests"][0]["target_new"] == e["requests"][0]["target_new"] and d["requests"][0]["image"] == e["requests"][0]["image"]
        for k in ("text_rephrase", "image_rephrase"):
            assert d["generality"][k][0]["prompt"] + SUF == e["generality"][k][0]["prompt"]
        assert list(d["locality"]) == list(e["locality"])
        for k in d["locality"]:
            want = e["locality"][k][0]["prompt"]
            want = want[:-len(SUF + "?")] + "?" if k == "text_loc" else want[:-len(SUF)]
            assert d["locality"][k][0]["prompt"] == want and d["locality"][k][0]["target"] == e["locality"][k][0]["target"]
    with pytest.raises(RuntimeError):
        V.EIC(os.path.join(DS, FILES["EVQA"]), "r", 2, OracleRetriever(g["corpus"]["EVQA"]))      # 'caption' not in basename (:260)
    with pytest.raises(RuntimeError):
        V.EIC(path, "r", 2, None)


def test_corpus_file_round_trip(tmp_path):
    import pickle
    import devqa_amd  # noqa: F401
    from devqa_amd.dataset.vllm import load_corpus, save_corpus
    g = _gold()
    c = dict(g["corpus"]["EVQA"], embeddings=RC.encode(g["corpus"]["EVQA"]["sentences"]))
    save_corpus(str(tmp_path / "c.npz"), c)
    r = load_corpus(str(tmp_path / "c.npz"))
    assert r["sentences"] == c["sentences"] and r["images"] == c["images"] and r["prompts"] == c["prompts"]
    assert np.array_equal(r["embeddings"], c["embeddings"])
    # the reference's on-disk layout: a pickled dict {sentences, images, prompts, embeddings} (vllm.py:96-103, util.py:83-85)
    with open(tmp_path / "ref.pkl", "wb") as f:
        pickle.dump(c, f, protocol=pickle.HIGHEST_PROTOCOL)
    r = load_corpus(str(tmp_path / "ref.pkl"))
    assert r["prompts"] == c["prompts"] and np.array_equal(r["embeddings"], c["embeddings"])

    class Evil:
        def __reduce__(self):
            return (os.system, ("true",))
    with open(tmp_path / "evil.pkl", "wb") as f:
        pickle.dump({"sentences": [Evil()], "embeddings": []}, f)
    with pytest.raises(RuntimeError):
        load_corpus(str(tmp_path / "evil.pkl"))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["EVQA", "VLKEB"])
def test_builder_on_the_hip_retriever(name):
    """EmbeddingRetriever.finds_sim_many -> devqa_cosine_topk: same probes as the reference constructor."""
    import devqa_amd  # noqa: F401
    from devqa_amd.dataset import vllm as V
    g = _gold()
    c = g["corpus"][name]
    retr = V.EmbeddingRetriever(RC.encode, RC.encode(c["sentences"]), c["prompts"], c["images"], "cuda:0")
    cls = {"EVQA": V.EVQA, "VLKEB": V.VLKEB}[name]
    ds = cls(os.path.join(DS, FILES[name]), g[name]["img_root"], g["n"], retr)
    assert ds.data_with_img_path == g[name]["data"]
    recs = json.load(open(os.path.join(DS, FILES[name])))
    # top-5 ids of the kernel == the reference-side ids (exact, in order), per record and in one batched launch
    ids = retr.topk(RC.encode([d["src"] for d in recs]), 5)
    assert ids.tolist() == [p["top5"] for p in g[name]["picks"]]
    one = retr.finds_sim(recs[2]["src"], recs[2]["pred"])
    assert g[name]["picks"][2]["branch"] == "last" and one[1] == c["images"][g[name]["picks"][2]["top5"][-1]]
    # corpus smaller than `tops`: the kernel pads the id row with -1; the fallback must be the last VALID hit
    small = V.EmbeddingRetriever(RC.encode, RC.encode(c["sentences"][:3]), [[p[0], "same"] for p in c["prompts"][:3]], c["images"][:3],
                                 "cuda:0")
    hit = small.topk(RC.encode([recs[0]["src"]]), 5)[0]
    assert sorted(hit[:3].tolist()) == [0, 1, 2] and hit[3:].tolist() == [-1, -1]
    pick = small.finds_sim(recs[0]["src"], "same")
    assert pick[1] == c["images"][int(hit[2])]
