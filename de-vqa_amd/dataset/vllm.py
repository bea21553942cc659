"""VLLM edit datasets + the dynamic-evaluation probe builder (R/dataset/vllm.py).

BaseVLLMEditData keeps the reference's two views (`data_with_img`, `data_with_img_path`; images
stay PATH strings, vllm.py:44-52).  The probe recipe of `__init_eic_evqa__` (vllm.py:121-228) and
the EVQA/VLKEB prompt suffixing (vllm.py:231-297) are restated in `build_probes`/`EVQA`/`VLKEB`
with a pluggable retriever, because the reference hard-codes a SentenceTransformer checkpoint and
pickled embeddings that do not exist offline (README allows "your own retriever").
Retrieval itself (normalise -> dot -> top-k -> first hit whose answer differs) runs on the HIP
cosine top-k kernel (devqa_cosine_topk).
"""
import json
import os
from copy import deepcopy
from typing import Callable, List, Optional, Sequence

import numpy as np

from . import BaseEditData


class BaseVLLMEditData(BaseEditData):
    def __init__(self, data_with_img, data_with_img_path) -> None:
        super().__init__(data_with_img)
        self.data = data_with_img
        self.data_with_img = data_with_img
        self.data_with_img_path = data_with_img_path

    def get_data_with_img_path(self):
        return self.data_with_img_path


class EmbeddingRetriever:
    """finds_sim (vllm.py:65-87) over a stored corpus {sentences, images, prompts, embeddings}
    (the on-disk dict layout of the reference's pickles, passed in as arrays/lists).
    `encode(list[str]) -> float32 [n, D]` is the user's sentence encoder."""

    def __init__(self, encode: Callable[[Sequence[str]], np.ndarray], embeddings: np.ndarray, prompts: List,
                 images: List[str], device="cuda:0"):
        import torch
        self.encode = encode
        self.prompts = prompts
        self.save_image_path = images
        self.device = device
        self.stored = torch.as_tensor(np.asarray(embeddings, np.float32)).to(device).contiguous()
        # the corpus side of the cosine is normalised ONCE, at load (vllm.py:104,117: util.normalize_embeddings on the stored tensor)
        from .. import lib
        self.stored_inv_norm = lib.row_inv_norm(self.stored) if self.stored.is_cuda and self.stored.shape[0] else None

    def topk(self, queries: np.ndarray, tops=5):
        import torch
        from .. import lib
        q = torch.as_tensor(np.asarray(queries, np.float32)).to(self.device).contiguous()
        idx, _ = lib.cosine_topk(self.stored, q, tops, True, True, corpus_inv_norm=self.stored_inv_norm)
        return idx.cpu().numpy()

    def finds_sim_many(self, srcs: List[str], trgs: List[str], tops=5):
        """Batched finds_sim: one kernel launch for all records."""
        hits = self.topk(self.encode(srcs), tops)
        out = []
        for hit, trg in zip(hits, trgs):
            pick = None
            for cid in hit:
                if cid >= 0 and self.prompts[cid][1] != trg:  # first hit with a different stored answer
                    pick = int(cid)
                    break
            if pick is None:                                     # else the last hit (vllm.py:79-81)
                valid = [int(c) for c in hit if c >= 0]          # a corpus smaller than `tops` pads the id row with -1
                if not valid:
                    raise RuntimeError("finds_sim: empty retrieval corpus")
                pick = valid[-1]
            out.append((self.prompts[pick], self.save_image_path[pick]))
        return out

    def finds_sim(self, src, trg, tops=5):
        return self.finds_sim_many([src], [trg], tops)[0]


# ---- the stored corpus on disk (vllm.py:96-103, R/easyeditor/models/ike/util.py:83-85) ----------------------------------------
CORPUS_KEYS = ("sentences", "images", "prompts", "embeddings")


def save_corpus(path, corpus: dict):
    """{sentences [n], images [n], prompts [n][2], embeddings [n,D]} -- the dict the reference pickles -- as an .npz of plain
    arrays (unicode / float32), which loads without unpickling anything."""
    n = len(corpus["embeddings"])
    assert len(corpus["sentences"]) == len(corpus["images"]) == len(corpus["prompts"]) == n
    np.savez(path, sentences=np.asarray(corpus["sentences"], dtype=str), images=np.asarray(corpus["images"], dtype=str),
             prompts=np.asarray([[str(a), str(b)] for a, b in corpus["prompts"]], dtype=str),
             embeddings=np.asarray(corpus["embeddings"], np.float32))


class _ArrayOnlyUnpickler(__import__("pickle").Unpickler):
    """Reads the reference's `*_embeddings*.pkl` layout (a dict of lists / str / numpy arrays, pickle.HIGHEST_PROTOCOL) while
    refusing every global except the handful numpy needs to rebuild an ndarray: nothing else in the file can execute."""
    ALLOWED = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"), ("numpy", "ndarray"),
               ("numpy", "dtype"), ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
               ("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer")}

    def find_class(self, module, name):
        if (module, name) in self.ALLOWED:
            return super().find_class(module, name)
        raise RuntimeError("refusing to unpickle %s.%s from a corpus file" % (module, name))


def load_corpus(path) -> dict:
    """.npz written by save_corpus, or the reference's pickle of the same dict (array-only unpickler, see above)."""
    if str(path).endswith(".npz"):
        z = np.load(path, allow_pickle=False)
        return {"sentences": z["sentences"].tolist(), "images": z["images"].tolist(),
                "prompts": [list(p) for p in z["prompts"].tolist()], "embeddings": z["embeddings"]}
    with open(path, "rb") as f:
        d = _ArrayOnlyUnpickler(f).load()
    if not isinstance(d, dict) or not set(("sentences", "embeddings")) <= set(d):
        raise RuntimeError("%s is not a stored-corpus dict" % path)
    d["embeddings"] = np.asarray(d["embeddings"], np.float32)
    return d


def build_probes(records: List[dict], img_root_dir: str, retrieved: List) -> List[dict]:
    """records: raw JSON records (keys src, pred, rephrase, alt, image, image_rephrase, loc, loc_ans,
    m_loc, m_loc_q, m_loc_a); retrieved[i] = ([t2_prompt, t2_answer], i2_path) from finds_sim(src, pred).
    Returns the per-record probe dicts of vllm.py:130-226 (no prompt suffixes yet)."""
    out = []
    for d, sim in zip(records, retrieved):
        i1 = os.path.join(img_root_dir, d["image"])
        t1, t2, t3 = d["src"], sim[0][0], d["m_loc"]  # t3 is an image PATH used as text (quirk, vllm.py:164)
        i2, i3 = sim[1], os.path.join(img_root_dir, d["m_loc"])
        out.append({
            "requests": [{"image": i1, "prompt": d["src"], "target_new": d["alt"]}],
            "generality": {
                "text_rephrase": [{"image": i1, "prompt": d["rephrase"], "target": d["alt"]}],
                "image_rephrase": [{"image": os.path.join(img_root_dir, d["image_rephrase"]), "prompt": d["src"],
                                    "target": d["alt"]}],
            },
            "locality": {
                "text_loc": [{"image": None, "prompt": d["loc"], "target": d["loc_ans"]}],
                "t3i3": [{"image": i3, "prompt": d["m_loc_q"], "target": d["m_loc_a"]}],
                "t1i4": [{"image": None, "prompt": t1, "target": d["alt"]}],
                "t2i4": [{"image": None, "prompt": t2, "target": d["alt"]}],
                "t1i2": [{"image": i2, "prompt": t1, "target": d["alt"]}],
                "t1i3": [{"image": i3, "prompt": t1, "target": d["alt"]}],
                "t2i1": [{"image": i1, "prompt": t2, "target": d["alt"]}],
                "t2i2": [{"image": i2, "prompt": t2, "target": d["alt"]}],
                "t3i1": [{"image": i1, "prompt": t3, "target": d["m_loc_a"]}],
            },
        })
    return out


def _load_records(data_path, data_n):
    with open(data_path, "r") as f:
        data = json.load(f)
    return data[:min(len(data), data_n if data_n is not None else 99999999)]


SUFFIX = " The answer is:"


class EVQA(BaseVLLMEditData):
    """vllm.py:231-254: every prompt gets ' The answer is:'; text_loc additionally '?' AFTER it."""

    def __init__(self, data_path: str = "data/easy-edit-mm/vqa/vqa_train.json", img_root_dir: str = "data/easy-edit-mm/images",
                 data_n=None, retriever: Optional[EmbeddingRetriever] = None) -> None:
        if "vqa" not in os.path.basename(data_path):
            raise RuntimeError("not a vqa file")
        if retriever is None:
            raise RuntimeError("EVQA needs a retriever (the reference's hard-coded SentenceTransformer path does not exist here)")
        recs = _load_records(data_path, data_n)
        retrieved = retriever.finds_sim_many([d["src"] for d in recs], [d["pred"] for d in recs])
        data = build_probes(recs, img_root_dir, retrieved)
        for d in data:
            d["requests"][0]["prompt"] += SUFFIX
            d["generality"]["text_rephrase"][0]["prompt"] += SUFFIX
            d["generality"]["image_rephrase"][0]["prompt"] += SUFFIX
            for name in d["locality"]:
                d["locality"][name][0]["prompt"] += SUFFIX
            d["locality"]["text_loc"][0]["prompt"] += "?"
        super().__init__(deepcopy(data), data)

    def dataset_name(self):
        return "EVQA"


class EIC(BaseVLLMEditData):
    """vllm.py:257-271: same 11-field records and probe recipe as EVQA (`__init_eic_evqa__`); only text_loc changes ('?' appended, no
    ' The answer is:' anywhere).  The reference's constructor then indexes `d['locality']['image_loc']` (:265), a probe its builder no
    longer makes (:131 is commented out), so `EIC(...)` raises KeyError there on any non-empty file; here that suffix is applied when
    a record set carries an image_loc probe and skipped otherwise."""

    def __init__(self, data_path: str = "data/easy-edit-mm/caption/caption_train_edit.json",
                 img_root_dir: str = "data/easy-edit-mm/images", data_n=None, retriever: Optional[EmbeddingRetriever] = None):
        if "caption" not in os.path.basename(data_path):
            raise RuntimeError("not a caption file")
        if retriever is None:
            raise RuntimeError("EIC needs a retriever")
        recs = _load_records(data_path, data_n)
        retrieved = retriever.finds_sim_many([d["src"] for d in recs], [d["pred"] for d in recs])
        data = build_probes(recs, img_root_dir, retrieved)
        for d in data:
            d["locality"]["text_loc"][0]["prompt"] += "?"
            if "image_loc" in d["locality"]:
                d["locality"]["image_loc"][0]["prompt"] += SUFFIX
        super().__init__(deepcopy(data), data)

    def dataset_name(self):
        return "EIC"


class VLKEB(BaseVLLMEditData):
    """vllm.py:274-297: only locality prompts get the suffix (+ '?' on text_loc)."""

    def __init__(self, data_path: str = "data/VLKEB/train.json", img_root_dir: str = "data/VLKEB/mmkb_images", data_n=None,
                 retriever: Optional[EmbeddingRetriever] = None):
        if retriever is None:
            raise RuntimeError("VLKEB needs a retriever")
        recs = _load_records(data_path, data_n)
        retrieved = retriever.finds_sim_many([d["src"] for d in recs], [d["pred"] for d in recs])
        data = build_probes(recs, img_root_dir, retrieved)
        for d in data:
            for name in d["locality"]:
                d["locality"][name][0]["prompt"] += SUFFIX
            d["locality"]["text_loc"][0]["prompt"] += "?"
        super().__init__(deepcopy(data), data)

    def dataset_name(self):
        return "VLKEB"
