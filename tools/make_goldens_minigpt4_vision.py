#!/usr/bin/env python3
"""Golden vectors for MiniGPT-4's image path from the REFERENCE's own modules (build container only; needs /root/reference).

`MiniGPT4ForEdit` itself cannot be imported here (omegaconf, peft and `LLAMA_INPUTS_DOCSTRING` are absent), but the two files that
hold its vision arithmetic import on their own: `modules/eva_vit.py` (`VisionTransformer`) and `modules/Qformer.py`
(`BertLMHeadModel`).  This script loads those two files by path, builds them at TINY_MINIGPT4's dims exactly as
`modules/base_model.py:118-140` (`init_vision_encoder`: ViT + a separate `ln_vision`) and `modules/minigpt4.py:189-215`
(`init_Qformer`: BertConfig + encoder_width / add_cross_attention / cross_attention_freq 2 / query_length, then `cls`, the word and
position embeddings and every layer's text FFN set to None) do, fills every parameter from a seeded generator, and runs the body of
`modules/minigpt4.py:217-244` (`encode_img`) in fp32:

    image_embeds = ln_vision(visual_encoder(image));  query_output = Qformer.bert(query_embeds=query_tokens.expand(B, -1, -1),
    encoder_hidden_states=image_embeds, encoder_attention_mask=ones, return_dict=True);  llama_proj(query_output.last_hidden_state)

Output: tests/golden/tiny_minigpt4_vision_goldens.npz -- the state dict under MiniGPT-4's checkpoint names ("visual_encoder.*",
"ln_vision.*", "Qformer.*", "query_tokens", "llama_proj.*"), pixel_values, image_embeds, inputs_llama.  A fixture holds data only."""
import importlib.util
import os
import sys
from functools import partial

import numpy as np
import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/DE-VQA/editor/vllms_for_edit/minigpt4/modules"


def load(name, fn):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, fn))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def transformers_moved_names():
    """Qformer.py (written for transformers 4.2x) imports three helpers from `transformers.modeling_utils`; the installed transformers
    keeps two of them in `transformers.pytorch_utils` and no longer has the third (`find_pruneable_heads_and_indices`, used only by
    `prune_heads`, which this path never calls).  Re-export the library's own functions under the old location; the missing one raises
    if it is ever reached.  Nothing of the reference is replaced."""
    import transformers.modeling_utils as M
    import transformers.pytorch_utils as P
    for n in ("apply_chunking_to_forward", "prune_linear_layer"):
        if not hasattr(M, n):
            setattr(M, n, getattr(P, n))
    if not hasattr(M, "find_pruneable_heads_and_indices"):
        def find_pruneable_heads_and_indices(*a, **k):
            raise RuntimeError("prune_heads is not on the golden path")
        M.find_pruneable_heads_and_indices = find_pruneable_heads_and_indices


def main():
    import devqa_amd  # noqa: F401
    transformers_moved_names()
    from devqa_amd import minigpt4_spec as S
    cfg = S.TINY_MINIGPT4
    v, q, t = cfg["vision_config"], cfg["qformer_config"], cfg["text_config"]
    eva = load("ref_mg4_eva_vit", "eva_vit.py")
    qf = load("ref_mg4_qformer", "Qformer.py")
    torch.manual_seed(20261004)
    vit = eva.VisionTransformer(img_size=v["image_size"], patch_size=v["patch_size"], use_mean_pooling=False, embed_dim=v["hidden_size"],
                                depth=v["num_hidden_layers"], num_heads=v["num_attention_heads"],
                                mlp_ratio=v["intermediate_size"] / v["hidden_size"], qkv_bias=True, drop_path_rate=0.0,
                                norm_layer=partial(nn.LayerNorm, eps=v["layer_norm_eps"]), use_checkpoint=False).eval()
    ln_vision = nn.LayerNorm(vit.num_features).eval()              # base_model.py:131 (its LayerNorm subclass only casts to fp32 and back)
    bc = qf.BertConfig(vocab_size=32, hidden_size=q["hidden_size"], num_hidden_layers=q["num_hidden_layers"],
                       num_attention_heads=q["num_attention_heads"], intermediate_size=q["intermediate_size"],
                       layer_norm_eps=q["layer_norm_eps"], max_position_embeddings=16, hidden_dropout_prob=0.0,
                       attention_probs_dropout_prob=0.0)
    bc.encoder_width = vit.num_features                              # minigpt4.py:192-196
    bc.add_cross_attention = True
    bc.cross_attention_freq = q["cross_attention_frequency"]
    bc.query_length = cfg["num_query_tokens"]
    # the file calls the pre-4.x `self.init_weights()` in its constructors; transformers 5 reads `all_tied_weights_keys` there, which only
    # its newer `post_init()` sets.  An empty table on the loaded class (no tied weights on this path: the LM head is dropped below) lets the
    # constructors run; every parameter is overwritten from the seeded generator afterwards, so the library's initialiser has no say.
    qf.BertPreTrainedModel.all_tied_weights_keys = {}
    # `PreTrainedModel.get_head_mask` (transformers 4: `[None] * num_hidden_layers` when head_mask is None, the only case here) is gone too
    if not hasattr(qf.BertPreTrainedModel, "get_head_mask"):
        def get_head_mask(self, head_mask, num_hidden_layers, is_attention_chunked=False):
            assert head_mask is None
            return [None] * num_hidden_layers
        qf.BertPreTrainedModel.get_head_mask = get_head_mask
    Qformer = qf.BertLMHeadModel(config=bc)
    query_tokens = nn.Parameter(torch.zeros(1, cfg["num_query_tokens"], bc.hidden_size))
    Qformer.cls = None                                               # minigpt4.py:203-208
    Qformer.bert.embeddings.word_embeddings = None
    Qformer.bert.embeddings.position_embeddings = None
    for layer in Qformer.bert.encoder.layer:
        layer.output = None
        layer.intermediate = None
    Qformer = Qformer.eval()
    llama_proj = nn.Linear(bc.hidden_size, t["hidden_size"]).eval()  # minigpt4.py:70-72
    mods = {"visual_encoder": vit, "ln_vision": ln_vision, "Qformer": Qformer, "llama_proj": llama_proj}
    g = torch.Generator().manual_seed(7)
    state = {}
    with torch.no_grad():
        for pre, m in mods.items():
            for n, p in m.named_parameters():
                if n.endswith("norm1.weight") or n.endswith("norm2.weight") or n.endswith("LayerNorm.weight") or (pre == "ln_vision" and n == "weight"):
                    p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
                elif p.dim() >= 2:
                    p.copy_(torch.randn(p.shape, generator=g) * (0.6 / p.shape[-1] ** 0.5 if p.dim() == 2 else 0.05))
                else:
                    p.copy_(0.1 * torch.randn(p.shape, generator=g))
                state[pre + "." + n] = p.detach().clone()
        query_tokens.copy_(0.5 * torch.randn(query_tokens.shape, generator=g))
        state["query_tokens"] = query_tokens.detach().clone()
        pv = torch.randn(3, 3, v["image_size"], v["image_size"], generator=g)
        image_embeds = ln_vision(vit(pv))                                                                    # minigpt4.py:224
        image_atts = torch.ones(image_embeds.size()[:-1], dtype=torch.long)
        qo = Qformer.bert(query_embeds=query_tokens.expand(image_embeds.shape[0], -1, -1), encoder_hidden_states=image_embeds,
                          encoder_attention_mask=image_atts, return_dict=True)
        inputs_llama = llama_proj(qo.last_hidden_state)                                                       # minigpt4.py:236
    out = {"w/" + k: a.numpy() for k, a in state.items()}
    out.update(pixel_values=pv.numpy(), image_embeds=image_embeds.numpy(), inputs_llama=inputs_llama.numpy())
    path = os.path.join(ROOT, "tests", "golden", "tiny_minigpt4_vision_goldens.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "params", len(state), "image_embeds", tuple(image_embeds.shape), "inputs_llama", tuple(inputs_llama.shape))
    for k in sorted(state):
        print("  ", k, tuple(state[k].shape))


if __name__ == "__main__":
    main()
