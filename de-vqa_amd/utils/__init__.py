"""Factories and name maps: same functions/arguments as R/utils/__init__.py:29-175 for the
editors (ft_vl, mend_vl, ike_vl, tp_vl, lte_vl) and models (blip2-opt-2.7b, llava-v1.5-7b, minigpt-4-vicuna-7b) built on the HIP path."""
import os
from typing import List, Union

import numpy as np
import torch
from torch import nn

from .GLOBAL import ROOT_PATH, model_path_map


def find_module(module, module_path: str) -> Union[torch.Tensor, nn.Module]:  # utils/__init__.py:29-37
    for comp in module_path.split("."):
        if hasattr(module, comp):
            module = getattr(module, comp)
        elif comp.isdigit():
            module = module[int(comp)]
        else:
            raise RuntimeError(f"Couldn't find child module {comp}")
    return module


def move_to_device(data, device):  # utils/__init__.py:39-52
    if isinstance(data, (torch.Tensor, nn.Module)):
        return data.to(device)
    if isinstance(data, list):
        return [move_to_device(i, device) for i in data]
    if isinstance(data, tuple):
        return tuple(move_to_device(i, device) for i in data)
    if isinstance(data, dict):
        return {k: move_to_device(v, device) for k, v in data.items()}
    if isinstance(data, (int, float, str, bool, type(None), np.integer, np.floating)):
        return data
    raise TypeError(f"Unsupported data type: {type(data)}")


def get_full_model_name(model_name_part: str) -> str:  # utils/__init__.py:54-99 (VLLM names)
    p = model_name_part.lower()
    if "blip2" in p:
        return "blip2-opt-2.7b"
    if "llava" in p:
        return "llava-v1.5-7b"
    if "mini" in p and "4" in p and "gpt" in p:
        return "minigpt-4-vicuna-7b"
    raise RuntimeError("unknown model name %s" % model_name_part)


def get_editor_config_path(editor_name: str, edit_model_name: str):  # :101-103
    return os.path.join(ROOT_PATH, "configs", editor_name.lower(), "%s.yaml" % get_full_model_name(edit_model_name))


def get_model_path(model_name: str) -> str:
    return model_path_map[get_full_model_name(model_name)]


def load_vllm_for_edit(model_name: str, device: str, dtype="bf16"):  # :111-124
    model_name = get_full_model_name(model_name)
    model_path = get_model_path(model_name)
    print('Loading %s from "%s".' % (model_name, model_path))
    if "blip2" in model_name:
        from ..editor.vllms_for_edit.blip2.blip2 import BLIP2OPTForEdit
        return BLIP2OPTForEdit(model_path, device, dtype=dtype)
    if "llava" in model_name:
        from ..editor.vllms_for_edit.llava.llava import LlavaForEdit
        return LlavaForEdit(model_path, device, True, dtype=dtype)
    if "minigpt" in model_name:
        from ..editor.vllms_for_edit.minigpt4.minigpt4 import MiniGPT4ForEdit
        return MiniGPT4ForEdit(model_path, device, True, dtype=dtype)
    raise BaseException("Have not write `BaseVLLMForEdit` for `%s`." % model_name)


def load_vllm_editor(editor_name: str, edit_model_name: str, device, extra_devices: List[int] = [1],
                     editor_ckpt_path=None, for_train=False, dtype="bf16", **editor_kwargs):  # :126-175
    editor_name = editor_name.lower()
    config_path = get_editor_config_path(editor_name, edit_model_name)
    vllm = load_vllm_for_edit(edit_model_name, device, dtype)
    if editor_name == "ft_vl":
        from ..editor.vllm_editors.ft_vl.ft_vl import FTvl, FTvlConfig
        return FTvl(vllm, FTvlConfig.from_yaml(config_path), device)
    if editor_name == "ike_vl":  # needs corpus={sentences, embeddings} and encode=callable (see ike_vl.py)
        from ..editor.vllm_editors.ike_vl.ike_vl import IKEvl, IKEvlConfig
        return IKEvl(vllm, IKEvlConfig.from_yaml(config_path), device, **editor_kwargs)
    if editor_name == "mend_vl":  # editor_ckpt_path: a reference-format `Best` checkpoint of the trained hyper-network
        from ..editor.vllm_editors.mend_vl.mend_vl import MENDvl, MENDvlConfig
        return MENDvl(vllm, MENDvlConfig.from_yaml(config_path), device, ckpt_path=editor_ckpt_path, for_train=for_train,
                      **editor_kwargs)
    if editor_name == "tp_vl":   # needs locality_texts=[...] or locality_data_path=<text file> (the reference reads wikitext)
        from ..editor.vllm_editors.tp_vl.tp_vl import TPvl, TPvlConfig
        return TPvl(vllm, TPvlConfig.from_yaml(config_path), device, **editor_kwargs)
    if editor_name == "lte_vl":  # needs encode=callable (the reference loads sentence_transformers' multi-qa-mpnet-base-dot-v1)
        from ..editor.vllm_editors.lte_vl.lte_vl import LTEvl, LTEvlConfig
        # for_train (:159-162): a second, frozen copy of the model prepares the training batches (the reference puts it on
        # extra_devices[0]; one MI355X holds both, so it lives on the same device); a trained checkpoint is loaded below
        vllm_data_proc = load_vllm_for_edit(edit_model_name, device, dtype) if for_train else None
        ed = LTEvl(vllm, LTEvlConfig.from_yaml(config_path), device, vllm_data_proc, device if for_train else None, **editor_kwargs)
        if editor_ckpt_path is not None:      # :173-174
            ed.load_ckpt(editor_ckpt_path, True, False)
        return ed
    raise RuntimeError("No such editor %s" % editor_name)
