#!/usr/bin/env python3
"""Dataset-builder goldens (build container only; imports the REFERENCE's R/dataset/vllm.py).

Runs the reference's own `EVQA(...)` and `VLKEB(...)` constructors -- `__init_eic_evqa__` (probe recipe, vllm.py:121-228), its
`finds_sim` selection rule (:65-87) and the prompt suffix rules (:231-254, :274-297) -- on the first N records of
R/data/easy-edit-mm/vqa/vqa_eval.json and R/data/VLKEB/eval.json and stores the probe dicts it builds.

What is replaced in-process (nothing in /root/reference is modified):
  * `init_retrieval` (:89-117 loads a SentenceTransformer checkpoint and a pickle from author-local paths): sets the same five
    attributes from a deterministic corpus + encoder defined in tests/retr_common.py (so the test side re-creates them);
  * `sentence_transformers.util` (package absent): `normalize_embeddings`, `dot_score`, `semantic_search` restated from their
    published semantics (row L2 normalisation; Q.C^T; per query the top_k hits sorted by descending score, ties -> lowest id) in
    float64 -- parity of the RETRIEVAL ARITHMETIC stays "unpinned by the reference" (SURVEY 8(c)), the selection rule and the
    probe recipe around it are the reference's own code;
  * `Tensor.to(<int>)` is a no-op while the constructors run (`finds_sim` moves the query to GPU 0, :67).

Outputs (data only): tests/golden/dataset/{vqa_eval_head.json, vlkeb_eval_head.json} = the first N raw records (the reference's own
data files, truncated), tests/golden/dataset/goldens.json = {"corpus": {...}, "EVQA": [...], "VLKEB": [...]}.
"""
import json
import os
import sys
import types

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/DE-VQA"
OUT = os.path.join(ROOT, "tests", "golden", "dataset")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import retr_common as RC  # noqa: E402  (tests/retr_common.py: corpus recipe + deterministic encoder shared with the tests)

N_EVAL = 12


def _util_stub():
    u = types.ModuleType("sentence_transformers.util")

    def normalize_embeddings(e):
        return torch.nn.functional.normalize(e.double(), p=2, dim=1)

    def dot_score(a, b):
        return a.double() @ b.double().t()

    def semantic_search(query_embeddings, corpus_embeddings, score_function=dot_score, top_k=10):
        s = score_function(query_embeddings, corpus_embeddings).numpy()
        out = []
        for row in s:
            order = np.lexsort((np.arange(len(row)), -row))[:top_k]
            out.append([{"corpus_id": int(i), "score": float(row[i])} for i in order])
        return out
    u.normalize_embeddings, u.dot_score, u.semantic_search = normalize_embeddings, dot_score, semantic_search
    return u


def main():
    st = types.ModuleType("sentence_transformers")
    st.SentenceTransformer = object
    st.util = _util_stub()
    sys.modules["sentence_transformers"] = st
    sys.modules["sentence_transformers.util"] = st.util
    import dataset.vllm as RV   # the reference module

    os.makedirs(OUT, exist_ok=True)
    heads = {}
    for name, src, dst in (("EVQA", "data/easy-edit-mm/vqa/vqa_eval.json", "vqa_eval_head.json"),
                           ("VLKEB", "data/VLKEB/eval.json", "vlkeb_eval_head.json")):
        recs = json.load(open(os.path.join(REF, src)))[:N_EVAL]
        json.dump(recs, open(os.path.join(OUT, dst), "w"), indent=1)
        heads[name] = (os.path.join(OUT, dst), recs)
    corpora = {name: RC.build_corpus(heads[name][1]) for name in heads}

    class _Enc:
        def encode(self, src, show_progress_bar=False):
            return RC.encode([src])[0]

    def make_init(name):
        def init_retrieval(self, types="VLKEB"):
            c = corpora[name]
            self.sentence_model = _Enc()
            self.stored_sentences = c["sentences"]
            self.save_image_path = c["images"]
            self.prompts = c["prompts"]
            self.stored_embeddings = RV.util.normalize_embeddings(torch.tensor(RC.encode(c["sentences"])))
        return init_retrieval
    orig_to = torch.Tensor.to

    def to_noint(self, *a, **k):
        if len(a) == 1 and isinstance(a[0], int) and not k:
            return self
        return orig_to(self, *a, **k)
    out = {"n": N_EVAL, "dim": RC.DIM, "corpus": {k: {kk: v[kk] for kk in ("sentences", "images", "prompts")} for k, v in corpora.items()}}
    torch.Tensor.to = to_noint
    try:
        for name, cls, root in (("EVQA", RV.EVQA, "imgs/evqa"), ("VLKEB", RV.VLKEB, "imgs/vlkeb")):
            RV.BaseVLLMEditData.init_retrieval = make_init(name)
            ds = cls(heads[name][0], root, N_EVAL)
            assert ds.dataset_name() == name
            assert ds.data_with_img == ds.data_with_img_path      # images stay path strings (vllm.py:44-52)
            out[name] = {"img_root": root, "data": ds.data_with_img_path}
            # which selection branch each record took (for the test's coverage assertion)
            picks = []
            for d in heads[name][1]:
                q = RV.util.normalize_embeddings(torch.tensor(RC.encode([d["src"]])))
                hit = RV.util.semantic_search(q, RV.util.normalize_embeddings(torch.tensor(RC.encode(corpora[name]["sentences"]))),
                                              top_k=5)[0]
                ids = [h["corpus_id"] for h in hit]
                first_diff = next((k for k, i in enumerate(ids) if corpora[name]["prompts"][i][1] != d["pred"]), None)
                picks.append({"top5": ids, "branch": "last" if first_diff is None else "rank%d" % first_diff})
            out[name]["picks"] = picks
    finally:
        torch.Tensor.to = orig_to
    json.dump(out, open(os.path.join(OUT, "goldens.json"), "w"), indent=1)
    for name in ("EVQA", "VLKEB"):
        print(name, [p["branch"] for p in out[name]["picks"]])


if __name__ == "__main__":
    main()
