#!/usr/bin/env python3
"""CLI of the edit+eval path: same flags as R/test_vllm_edit.py:7-18
(-en -mn -sen -enp -dvc -ckpt -dn -dsn), plus:
  --data_path / --img_root : dataset JSON and image root (the reference hard-codes author paths)
  --embeddings             : .npz {embeddings, prompts, images} of the retrieval corpus + --queries
                             .npz of pre-computed query embeddings (all-MiniLM-L6-v2 is unavailable
                             offline; any sentence encoder with the same interface works)
  --dtype bf16|fp32
Under `torchrun` (WORLD_SIZE>1) splits are sharded over ranks; rank 0 writes the result files.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def get_attr():
    p = argparse.ArgumentParser()
    p.add_argument("-en", "--editor_name", type=str, required=True)
    p.add_argument("-mn", "--edit_model_name", type=str, required=True)
    p.add_argument("-sen", "--sequential_edit_n", type=int, default=1)
    p.add_argument("-enp", "--eval_name_postfix", type=str, default="")
    p.add_argument("-dvc", "--device", type=str, default="cuda:0")
    p.add_argument("-ckpt", "--editor_ckpt_path", type=str, default=None)
    p.add_argument("-dn", "--data_name", type=str, default="EVQA")
    p.add_argument("-dsn", "--data_sample_n", type=int, default=None)
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
    from devqa_amd.dist import init_from_env
    from devqa_amd.evaluation.vllm_editor_eval import VLLMEditorEvaluation
    from devqa_amd.utils import get_full_model_name, load_vllm_editor
    rank, world = init_from_env()
    if world > 1:
        cfg.device = "cuda:%d" % int(os.environ.get("LOCAL_RANK", "0"))
    cfg.editor_name = cfg.editor_name.lower()
    cfg.edit_model_name = get_full_model_name(cfg.edit_model_name)
    cfg.evaluation_name = cfg.data_name.upper() + ("-%s" % cfg.eval_name_postfix if cfg.eval_name_postfix else "")
    out_dir = os.path.join("eval_results", cfg.editor_name, cfg.edit_model_name, cfg.evaluation_name,
                           "sequential_edit_%s" % cfg.sequential_edit_n)
    if os.path.exists(out_dir):  # the reference checks '<...>/single_edit', which it never writes
        print("Has evaluated: %s" % out_dir)
        sys.exit()
    editor = load_vllm_editor(cfg.editor_name, cfg.edit_model_name, cfg.device, None, cfg.editor_ckpt_path, False, cfg.dtype)
    corpus = np.load(cfg.embeddings, allow_pickle=False)
    qz = np.load(cfg.queries, allow_pickle=False)
    qmap = {s: e for s, e in zip(qz["sentences"].tolist(), qz["embeddings"])}
    retriever = EmbeddingRetriever(lambda srcs: np.stack([qmap[s] for s in srcs]), corpus["embeddings"],
                                   [tuple(p) for p in corpus["prompts"].tolist()], corpus["images"].tolist(), cfg.device)
    ds = {"EVQA": EVQA, "VLKEB": VLKEB}[cfg.data_name.upper()]
    eval_data = ds(cfg.data_path, cfg.img_root, cfg.data_sample_n, retriever)
    ev = VLLMEditorEvaluation(editor, eval_data, cfg.evaluation_name, "eval_results")
    ev.evaluate_sequential_edit(cfg.sequential_edit_n, False, None)
