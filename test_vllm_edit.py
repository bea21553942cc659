#!/usr/bin/env python3
"""CLI of the edit+eval path: the flags of R/test_vllm_edit.py:7-18 (-en -mn -sen -enp -dvc -ckpt -dn -dsn), so that
`python test_vllm_edit.py -en ft_vl -mn blip2 -sen 1 -dvc cuda:0 -dn EVQA` parses and runs as in the reference.
What the reference hard-codes (dataset / image paths, the sentence encoder checkpoint, pickled embeddings) are optional
flags with defaults here -- see devqa_amd/cli.py.  Under `torchrun` (WORLD_SIZE > 1) splits are sharded over ranks
(one process per GPU, RCCL) and rank 0 writes the result files.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def get_attr(argv=None):
    from devqa_amd.cli import add_data_args
    p = argparse.ArgumentParser()
    p.add_argument("-en", "--editor_name", type=str, help="Editor name: FT_VL, MEND_VL, IKE_VL, TP_VL, LTE_VL")
    p.add_argument("-mn", "--edit_model_name", type=str, help="Editing model name: blip2, llava, minigpt4")
    p.add_argument("-sen", "--sequential_edit_n", type=int, help="Edit number.")
    p.add_argument("-enp", "--eval_name_postfix", type=str, default="", help="Postfix name of this evaluation.")
    p.add_argument("-dvc", "--device", type=str, help="GPU for editing.")
    p.add_argument("-ckpt", "--editor_ckpt_path", type=str, default=None, help="For editors that need training.")
    p.add_argument("-dn", "--data_name", type=str, help="Evaluating dataset: EVQA, VLKEB.")
    p.add_argument("-dsn", "--data_sample_n", type=int, default=None, help="Sample number for evaluation.")
    add_data_args(p)
    return p.parse_args(argv)


def main(argv=None):
    import devqa_amd  # noqa: F401
    cfg = get_attr(argv)
    from devqa_amd import cli
    from devqa_amd.dist import init_from_env
    from devqa_amd.evaluation.vllm_editor_eval import VLLMEditorEvaluation
    from devqa_amd.utils import get_full_model_name, load_vllm_editor
    rank, world = init_from_env()
    if world > 1:
        cfg.device = "cuda:%d" % int(os.environ.get("LOCAL_RANK", "0"))
    cfg.editor_name = cfg.editor_name.lower()
    cfg.edit_model_name = get_full_model_name(cfg.edit_model_name)
    cfg.evaluation_name = cfg.data_name.upper() + ("-%s" % cfg.eval_name_postfix if cfg.eval_name_postfix != "" else "")
    out_dir = os.path.join("eval_results", cfg.editor_name, cfg.edit_model_name, cfg.evaluation_name,
                           "sequential_edit_%s" % cfg.sequential_edit_n)
    if os.path.exists(out_dir):  # the reference checks '<...>/single_edit', which it never writes
        print("Has evaluated: %s" % out_dir)
        sys.exit()
    print(cfg)
    editor = load_vllm_editor(cfg.editor_name, cfg.edit_model_name, cfg.device, None, cfg.editor_ckpt_path, False, cfg.dtype,
                              **cli.editor_kwargs(cfg))
    eval_data = cli.build_dataset(cfg, "eval")
    ev = VLLMEditorEvaluation(editor, eval_data, cfg.evaluation_name, "eval_results")
    ev.evaluate_sequential_edit(cfg.sequential_edit_n, False, None)


if __name__ == "__main__":
    main()
