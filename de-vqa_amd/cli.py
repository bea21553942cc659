"""Shared pieces of the two entry scripts (test_vllm_edit.py / train_vllm_editor.py).

The reference's scripts take only its own flags (R/test_vllm_edit.py:7-18, R/train_vllm_editor.py:14-29) and hard-code the
author's dataset paths, image roots, SentenceTransformer checkpoint and pickled embedding files
(R/test_vllm_edit.py:45-59, R/dataset/vllm.py:89-117).  Here those become OPTIONAL flags with defaults that mirror the
reference's relative layout, so the reference's command lines parse unchanged:

  --data_path   default  <data root>/easy-edit-mm/vqa/vqa_{eval,train}.json | <data root>/VLKEB/{eval,train}.json |
                         <data root>/easy-edit-mm/caption/caption_{eval,train}_edit.json
  --img_root    default  $DEVQA_IMG_ROOT or <data root>/easy-edit-mm/images | <data root>/VLKEB/mmkb_images
  --embeddings  default  <data root>/embeddings/{vqa,vlkeb}_embeddings.npz   corpus {embeddings, prompts, images[, sentences]}
                         (the reference's pickle dict layout {sentences, images, prompts, embeddings}, stored as .npz so that
                         nothing is unpickled)
  --queries     optional .npz {sentences, embeddings}: pre-computed sentence embeddings (a lookup-table "encoder")
  --encoder     optional "package.module:callable" -- the user's sentence encoder, list[str] -> float32 [n, D]
                (all-MiniLM-L6-v2 / multi-qa-mpnet-base-dot-v1 through sentence_transformers in the reference; neither is
                available offline, and the reference's README allows "your own retriever")
  --ike_corpus  optional .npz {sentences, embeddings} for IKE_VL (default: --embeddings when it carries `sentences`)
  --tp_texts    optional text file, one locality sentence per line, for TP_VL
<data root> = $DEVQA_DATA_ROOT or ./data.
"""
import importlib
import os

import numpy as np

DATA_FILES = {("EVQA", "eval"): "easy-edit-mm/vqa/vqa_eval.json", ("EVQA", "train"): "easy-edit-mm/vqa/vqa_train.json",
              ("VLKEB", "eval"): "VLKEB/eval.json", ("VLKEB", "train"): "VLKEB/train.json",
              ("EIC", "eval"): "easy-edit-mm/caption/caption_eval_edit.json", ("EIC", "train"): "easy-edit-mm/caption/caption_train_edit.json"}
IMG_ROOTS = {"EVQA": "easy-edit-mm/images", "VLKEB": "VLKEB/mmkb_images", "EIC": "easy-edit-mm/images"}
EMB_FILES = {"EVQA": "embeddings/vqa_embeddings.npz", "VLKEB": "embeddings/vlkeb_embeddings.npz", "EIC": "embeddings/caption_embeddings.npz"}


def data_root():
    return os.environ.get("DEVQA_DATA_ROOT", "data")


def add_data_args(p):
    p.add_argument("--data_path", type=str, default=None)
    p.add_argument("--img_root", type=str, default=None)
    p.add_argument("--embeddings", type=str, default=None)
    p.add_argument("--queries", type=str, default=None)
    p.add_argument("--encoder", type=str, default=None)
    p.add_argument("--ike_corpus", type=str, default=None)
    p.add_argument("--tp_texts", type=str, default=None)
    p.add_argument("--dtype", type=str, default="bf16")
    return p


def resolve_paths(cfg, split):
    """Fill the optional path flags from the defaults above.  -> (data_path, img_root, embeddings)"""
    name = cfg.data_name.upper()
    if name not in IMG_ROOTS:
        raise BaseException("Dataset %s is not built on this path (EVQA, EIC, VLKEB)." % name)
    data_path = cfg.data_path or os.path.join(data_root(), DATA_FILES[(name, split)])
    img_root = cfg.img_root or os.environ.get("DEVQA_IMG_ROOT") or os.path.join(data_root(), IMG_ROOTS[name])
    emb = cfg.embeddings or os.path.join(data_root(), EMB_FILES[name])
    return data_path, img_root, emb


def load_encoder(cfg):
    """The sentence encoder of this run: --encoder "module:callable", else the --queries lookup table, else None."""
    if getattr(cfg, "encoder", None):
        mod, _, fn = cfg.encoder.partition(":")
        if not fn:
            raise BaseException('--encoder must be "package.module:callable"')
        return getattr(importlib.import_module(mod), fn)
    if getattr(cfg, "queries", None):
        qz = np.load(cfg.queries, allow_pickle=False)
        table = {s: e for s, e in zip(qz["sentences"].tolist(), qz["embeddings"])}

        def lookup(sentences):
            try:
                return np.stack([table[s] for s in sentences]).astype(np.float32)
            except KeyError as e:
                raise BaseException("no pre-computed embedding for %r in %s (pass --encoder for free text)" % (e.args[0], cfg.queries))
        return lookup
    return None


def build_dataset(cfg, split):
    """EVQA / VLKEB with the cosine top-k retriever on the HIP kernel (R/test_vllm_edit.py:45-59, R/train_vllm_editor.py:59-83)."""
    from .dataset.vllm import EIC, EVQA, VLKEB, EmbeddingRetriever
    data_path, img_root, emb = resolve_paths(cfg, split)
    enc = load_encoder(cfg)
    if enc is None:
        raise BaseException("the dynamic-evaluation builder needs a sentence encoder: pass --encoder module:callable or --queries "
                            "<npz of pre-computed sentence embeddings> (the reference's all-MiniLM-L6-v2 checkpoint is not here)")
    if not os.path.exists(emb):
        raise BaseException("retrieval corpus %s not found (pass --embeddings; build one with devqa_amd.dataset.vllm.save_corpus)" % emb)
    corpus = np.load(emb, allow_pickle=False)
    retriever = EmbeddingRetriever(enc, corpus["embeddings"], [tuple(p) for p in corpus["prompts"].tolist()],
                                   corpus["images"].tolist(), cfg.device)
    ds = {"EVQA": EVQA, "EIC": EIC, "VLKEB": VLKEB}[cfg.data_name.upper()]
    n = getattr(cfg, "data_sample_n", None) if hasattr(cfg, "data_sample_n") else getattr(cfg, "data_n", None)
    return ds(data_path, img_root, n, retriever)


def editor_kwargs(cfg):
    """Constructor extras of the retrieval / patch editors, taken from the optional flags (none for ft_vl / mend_vl)."""
    name = cfg.editor_name.lower()
    kw = {}
    if name in ("ike_vl", "lte_vl"):
        enc = load_encoder(cfg)
        if enc is None:
            raise BaseException("%s needs a sentence encoder: pass --encoder module:callable or --queries <npz>" % name)
        kw["encode"] = enc
    if name == "ike_vl":
        path = cfg.ike_corpus or resolve_paths(cfg, "eval")[2]
        z = np.load(path, allow_pickle=False)
        if "sentences" not in z.files:
            raise BaseException("IKE corpus %s has no `sentences` (R/easyeditor/models/ike/util.py:83-85 stores them)" % path)
        kw["corpus"] = {"sentences": z["sentences"].tolist(), "embeddings": z["embeddings"]}
    if name == "tp_vl":
        if not cfg.tp_texts:
            raise BaseException("tp_vl needs --tp_texts <file with one locality sentence per line>")
        kw["locality_data_path"] = cfg.tp_texts
    return kw
