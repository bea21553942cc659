#!/usr/bin/env python3
"""CLI of the editor-training path: the flags of R/train_vllm_editor.py:14-29 (-en -mn -dna -bs -dvc -dn -lkpt -edvc -eps
-tnp -sci -lpi -ea -rs -dbs) and its two calls, `editor.train_init(...)` then `editor.train(epochs)` (:85-89).
-edvc is accepted and unused: the data-preparation producer runs on a second HIP stream of the SAME GPU, not on a second
GPU with a second model copy (devqa_amd/dataset/__init__.py).  Dataset / encoder paths: optional flags, see devqa_amd/cli.py.
Checkpoints go to records/<editor>/<model>/<train name>/checkpoints/Best in the reference's layout.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def get_attr(argv=None):
    from devqa_amd.cli import add_data_args

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
    add_data_args(p)
    return p.parse_args(argv)


def main(argv=None):
    import devqa_amd  # noqa: F401
    cfg = get_attr(argv)
    from devqa_amd import cli
    from devqa_amd.editor.vllm_editors.base import VLLMBaseEditorWithTraining
    from devqa_amd.utils import load_vllm_editor
    cfg.data_name = cfg.data_name.upper()
    editor = load_vllm_editor(cfg.editor_name, cfg.edit_model_name, cfg.device, cfg.extra_devices, None, True, cfg.dtype,
                              **cli.editor_kwargs(cfg))
    if not isinstance(editor, VLLMBaseEditorWithTraining):   # the reference fails here too (FT_VL has no train_init)
        raise BaseException("%s is not a trainable editor." % cfg.editor_name)
    train_data = cli.build_dataset(cfg, "train")
    editor.train_init(train_data, cfg.batch_size, train_name_prefix=cfg.train_name_prefix, load_ckpt_path=cfg.load_ckpt_path,
                      save_ckpt_per_i=cfg.save_ckpt_per_i, log_per_i=cfg.log_per_i, ema_alpha=cfg.ema_alpha,
                      random_seed=cfg.random_seed, data_buffer_size=cfg.data_buffer_size)
    editor.train(cfg.epochs)


if __name__ == "__main__":
    main()
