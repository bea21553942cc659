#!/usr/bin/env python3
"""CLI of the editor-training path: same flags as R/train_vllm_editor.py:7-33 (-en -mn -dna -bs -dvc -dn -lkpt -eps
-tnp -ea -rs; -edvc / -sci / -lpi / -dbs are accepted and ignored: there is no second-device producer thread and no
TensorBoard here), plus the dataset / retriever arguments of test_vllm_edit.py.  Trains the MEND_VL hyper-network
(the trainable editor built on the HIP path) and writes the best-EMA checkpoint in the reference's `Best` layout to
records/<editor>/<model>/<train name>/checkpoints/Best.
"""
import argparse
import os
import sys
from datetime import datetime

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def get_attr():
    def parse_lkpt(value):
        return None if value.lower() == "none" else value
    p = argparse.ArgumentParser()
    p.add_argument("-en", "--editor_name", type=str, required=True)
    p.add_argument("-mn", "--edit_model_name", type=str, required=True)
    p.add_argument("-dna", "--data_name", type=str, required=True)
    p.add_argument("-bs", "--batch_size", type=int, required=True)
    p.add_argument("-dvc", "--device", type=str, required=True)
    p.add_argument("-dn", "--data_n", type=int, default=None)
    p.add_argument("-lkpt", "--load_ckpt_path", type=parse_lkpt, default=None)
    p.add_argument("-edvc", "--extra_devices", type=int, nargs="+", default=[0])
    p.add_argument("-eps", "--epochs", type=int, default=1000)
    p.add_argument("-tnp", "--train_name_prefix", type=str, default=None)
    p.add_argument("-sci", "--save_ckpt_per_i", type=int, default=1000)
    p.add_argument("-lpi", "--log_per_i", type=int, default=1)
    p.add_argument("-ea", "--ema_alpha", type=float, default=0.1)
    p.add_argument("-rs", "--random_seed", type=int, default=None)
    p.add_argument("-dbs", "--data_buffer_size", type=int, default=4)
    p.add_argument("--data_path", type=str, required=True)
    p.add_argument("--img_root", type=str, required=True)
    p.add_argument("--embeddings", type=str, required=True)
    p.add_argument("--queries", type=str, required=True)
    p.add_argument("--dtype", type=str, default="bf16")
    return p.parse_args()


if __name__ == "__main__":
    cfg = get_attr()
    import numpy as np
    import devqa_amd  # noqa: F401
    from devqa_amd.dataset.vllm import EVQA, VLKEB, EmbeddingRetriever
    from devqa_amd.utils import get_full_model_name, load_vllm_editor
    cfg.editor_name = cfg.editor_name.lower()
    if cfg.editor_name != "mend_vl":
        raise BaseException("Only mend_vl is a trainable editor on the HIP path (got %s)." % cfg.editor_name)
    model_name = get_full_model_name(cfg.edit_model_name)
    editor = load_vllm_editor(cfg.editor_name, model_name, cfg.device, None, cfg.load_ckpt_path, True, cfg.dtype)
    corpus = np.load(cfg.embeddings, allow_pickle=False)
    qz = np.load(cfg.queries, allow_pickle=False)
    qmap = {s: e for s, e in zip(qz["sentences"].tolist(), qz["embeddings"])}
    retriever = EmbeddingRetriever(lambda srcs: np.stack([qmap[s] for s in srcs]), corpus["embeddings"],
                                   [tuple(p) for p in corpus["prompts"].tolist()], corpus["images"].tolist(), cfg.device)
    ds = {"EVQA": EVQA, "VLKEB": VLKEB}[cfg.data_name.upper()]
    train_data = ds(cfg.data_path, cfg.img_root, cfg.data_n, retriever)
    name = ((cfg.train_name_prefix + "-") if cfg.train_name_prefix else "") + datetime.now().strftime("%Y.%m.%d-%H.%M.%S")
    ckpt_dir = os.path.join("records", cfg.editor_name, model_name, name, "checkpoints")
    os.makedirs(ckpt_dir, exist_ok=True)

    def log(i, d):
        if i % cfg.log_per_i == 0:
            print("iter %d  loss %.4f  ema %.4f  grad-norm %.3f" % (i, d["Loss"], d["EMA Loss"], d["Grad-Norm"]), flush=True)
    ema = editor.train(train_data, cfg.epochs, cfg.batch_size, os.path.join(ckpt_dir, "Best"), cfg.random_seed, cfg.ema_alpha, log,
                       data_buffer_size=cfg.data_buffer_size)
    print("final EMA loss %.4f; checkpoint: %s" % (ema, os.path.join(ckpt_dir, "Best")))
