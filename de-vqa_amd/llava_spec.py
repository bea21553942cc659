"""LLaVA-1.5 architecture description and parameter-name map.

The reference addresses LLaVA weights by the parameter names of the HF release it was written for
(`language_model.model.layers.31.mlp.down_proj.weight`, R/configs/ft_vl/llava-v1.5-7b.yaml:8); newer
transformers moved the sub-modules (`model.language_model.layers...`, `model.vision_tower...`, `lm_head`).
The native tree uses the OLD names (what editor configs select); `old_to_new_name` / `new_to_old_name`
translate checkpoint keys either way (SURVEY.md 7.2 "HF-version drift").
"""
from collections import OrderedDict

LLAVA_1_5_7B = dict(
    vision_config=dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16,
                       image_size=336, patch_size=14, layer_norm_eps=1e-5, hidden_act="quick_gelu"),
    text_config=dict(hidden_size=4096, intermediate_size=11008, num_hidden_layers=32, num_attention_heads=32,
                     num_key_value_heads=32, vocab_size=32064, rms_norm_eps=1e-5, rope_theta=10000.0,
                     max_position_embeddings=4096, pad_token_id=32001),
    image_token_index=32000,
)
TINY_LLAVA = dict(
    vision_config=dict(hidden_size=32, intermediate_size=64, num_hidden_layers=3, num_attention_heads=2,
                       image_size=28, patch_size=14, layer_norm_eps=1e-5, hidden_act="quick_gelu"),
    text_config=dict(hidden_size=64, intermediate_size=96, num_hidden_layers=2, num_attention_heads=4,
                     num_key_value_heads=4, vocab_size=640, rms_norm_eps=1e-5, rope_theta=10000.0,
                     max_position_embeddings=256, pad_token_id=3),
    image_token_index=4,
)


def new_to_old_name(n: str) -> str:
    if n.startswith("model.language_model."):
        return "language_model.model." + n[len("model.language_model."):]
    if n.startswith("model.vision_tower.vision_model."):
        return "vision_tower." + n[len("model.vision_tower."):]
    if n.startswith("model.vision_tower."):   # transformers >= 5: CLIPVisionModel is flattened
        return "vision_tower.vision_model." + n[len("model.vision_tower."):]
    if n.startswith("model.multi_modal_projector."):
        return "multi_modal_projector." + n[len("model.multi_modal_projector."):]
    if n == "lm_head.weight":
        return "language_model.lm_head.weight"
    return n


def old_to_new_name(n: str) -> str:
    if n.startswith("language_model.model."):
        return "model.language_model." + n[len("language_model.model."):]
    if n.startswith("vision_tower.vision_model."):
        return "model.vision_tower." + n[len("vision_tower.vision_model."):]
    if n.startswith("multi_modal_projector."):
        return "model.multi_modal_projector." + n[len("multi_modal_projector."):]
    if n == "language_model.lm_head.weight":
        return "lm_head.weight"
    return n


def param_shapes(cfg):
    """OrderedDict {old_hf_name: shape}."""
    v, t = cfg["vision_config"], cfg["text_config"]
    P = OrderedDict()
    dv, fv = v["hidden_size"], v["intermediate_size"]
    npos = (v["image_size"] // v["patch_size"]) ** 2 + 1
    p = "vision_tower.vision_model."
    P[p + "embeddings.class_embedding"] = (dv,)
    P[p + "embeddings.patch_embedding.weight"] = (dv, 3, v["patch_size"], v["patch_size"])
    P[p + "embeddings.position_embedding.weight"] = (npos, dv)
    P[p + "pre_layrnorm.weight"] = (dv,)
    P[p + "pre_layrnorm.bias"] = (dv,)
    for i in range(v["num_hidden_layers"]):
        q = p + "encoder.layers.%d." % i
        for nm in ("k_proj", "v_proj", "q_proj", "out_proj"):
            P[q + "self_attn.%s.weight" % nm] = (dv, dv)
            P[q + "self_attn.%s.bias" % nm] = (dv,)
        P[q + "layer_norm1.weight"] = (dv,)
        P[q + "layer_norm1.bias"] = (dv,)
        P[q + "mlp.fc1.weight"] = (fv, dv)
        P[q + "mlp.fc1.bias"] = (fv,)
        P[q + "mlp.fc2.weight"] = (dv, fv)
        P[q + "mlp.fc2.bias"] = (dv,)
        P[q + "layer_norm2.weight"] = (dv,)
        P[q + "layer_norm2.bias"] = (dv,)
    P[p + "post_layernorm.weight"] = (dv,)
    P[p + "post_layernorm.bias"] = (dv,)
    dt, ft = t["hidden_size"], t["intermediate_size"]
    P["multi_modal_projector.linear_1.weight"] = (dt, dv)
    P["multi_modal_projector.linear_1.bias"] = (dt,)
    P["multi_modal_projector.linear_2.weight"] = (dt, dt)
    P["multi_modal_projector.linear_2.bias"] = (dt,)
    P["language_model.model.embed_tokens.weight"] = (t["vocab_size"], dt)
    for i in range(t["num_hidden_layers"]):
        q = "language_model.model.layers.%d." % i
        for nm in ("q_proj", "k_proj", "v_proj", "o_proj"):
            P[q + "self_attn.%s.weight" % nm] = (dt, dt)
        P[q + "mlp.gate_proj.weight"] = (ft, dt)
        P[q + "mlp.up_proj.weight"] = (ft, dt)
        P[q + "mlp.down_proj.weight"] = (dt, ft)
        P[q + "input_layernorm.weight"] = (dt,)
        P[q + "post_attention_layernorm.weight"] = (dt,)
    P["language_model.model.norm.weight"] = (dt,)
    P["language_model.lm_head.weight"] = (t["vocab_size"], dt)
    return P
