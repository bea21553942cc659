"""MiniGPT-4 (Vicuna-7B) architecture description, parameter names and the aliases that let the BLIP-2 vision engine
and the LLaMA decoder engine address them.

The reference builds MiniGPT-4 from LAVIS modules (R/editor/vllms_for_edit/minigpt4/modules/minigpt4.py:10-75): EVA
ViT-g (modules/eva_vit.py:488-519: 39 blocks, d 1408, 16 heads, MLP 6144, LN eps 1e-6, q/v bias only, no final norm)
-> `ln_vision` -> Q-Former (BERT-base with cross-attention every 2nd layer, 32 queries; only the query branch is
kept, :205-212) -> `llama_proj` -> Vicuna LLaMA.  This is the network HF's Blip2VisionModel / Blip2QFormerModel were
ported from, so the arithmetic is the BLIP-2 engine's under different parameter names; the decoder is LLaVA's.
Editor configs address `llama_model.model.layers.31.mlp.down_proj.weight` (R/configs/ft_vl/minigpt-4-vicuna-7b.yaml:8).
"""
import re
from collections import OrderedDict

MINIGPT4_VICUNA_7B = dict(
    vision_config=dict(hidden_size=1408, intermediate_size=6144, num_hidden_layers=39, num_attention_heads=16,
                       image_size=224, patch_size=14, layer_norm_eps=1e-6),
    qformer_config=dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                        cross_attention_frequency=2, encoder_hidden_size=1408, layer_norm_eps=1e-12),
    text_config=dict(hidden_size=4096, intermediate_size=11008, num_hidden_layers=32, num_attention_heads=32,
                     num_key_value_heads=32, vocab_size=32000, rms_norm_eps=1e-6, rope_theta=10000.0,
                     max_position_embeddings=2048, pad_token_id=0),
    num_query_tokens=32,
)
TINY_MINIGPT4 = dict(
    vision_config=dict(hidden_size=48, intermediate_size=96, num_hidden_layers=2, num_attention_heads=2,
                       image_size=28, patch_size=14, layer_norm_eps=1e-6),
    qformer_config=dict(hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2,
                        cross_attention_frequency=2, encoder_hidden_size=48, layer_norm_eps=1e-12),
    text_config=dict(hidden_size=64, intermediate_size=96, num_hidden_layers=2, num_attention_heads=4,
                     num_key_value_heads=4, vocab_size=640, rms_norm_eps=1e-6, rope_theta=10000.0,
                     max_position_embeddings=256, pad_token_id=3),
    num_query_tokens=8,
)


def param_shapes(cfg):
    """OrderedDict {MiniGPT-4 state-dict name: shape}."""
    v, q, t = cfg["vision_config"], cfg["qformer_config"], cfg["text_config"]
    P = OrderedDict()
    dv, fv = v["hidden_size"], v["intermediate_size"]
    npos = (v["image_size"] // v["patch_size"]) ** 2 + 1
    P["visual_encoder.cls_token"] = (1, 1, dv)
    P["visual_encoder.pos_embed"] = (1, npos, dv)
    P["visual_encoder.patch_embed.proj.weight"] = (dv, 3, v["patch_size"], v["patch_size"])
    P["visual_encoder.patch_embed.proj.bias"] = (dv,)
    for i in range(v["num_hidden_layers"]):
        b = "visual_encoder.blocks.%d." % i
        P[b + "norm1.weight"] = (dv,)
        P[b + "norm1.bias"] = (dv,)
        P[b + "attn.q_bias"] = (dv,)
        P[b + "attn.v_bias"] = (dv,)
        P[b + "attn.qkv.weight"] = (3 * dv, dv)
        P[b + "attn.proj.weight"] = (dv, dv)
        P[b + "attn.proj.bias"] = (dv,)
        P[b + "norm2.weight"] = (dv,)
        P[b + "norm2.bias"] = (dv,)
        P[b + "mlp.fc1.weight"] = (fv, dv)
        P[b + "mlp.fc1.bias"] = (fv,)
        P[b + "mlp.fc2.weight"] = (dv, fv)
        P[b + "mlp.fc2.bias"] = (dv,)
    P["ln_vision.weight"] = (dv,)
    P["ln_vision.bias"] = (dv,)
    dq, fq = q["hidden_size"], q["intermediate_size"]
    P["query_tokens"] = (1, cfg["num_query_tokens"], dq)
    P["Qformer.bert.embeddings.LayerNorm.weight"] = (dq,)
    P["Qformer.bert.embeddings.LayerNorm.bias"] = (dq,)
    for i in range(q["num_hidden_layers"]):
        b = "Qformer.bert.encoder.layer.%d." % i
        blocks = [("attention", dq)] + ([("crossattention", q["encoder_hidden_size"])] if i % q["cross_attention_frequency"] == 0 else [])
        for blk, kv_in in blocks:
            P[b + blk + ".self.query.weight"] = (dq, dq)
            P[b + blk + ".self.query.bias"] = (dq,)
            P[b + blk + ".self.key.weight"] = (dq, kv_in)
            P[b + blk + ".self.key.bias"] = (dq,)
            P[b + blk + ".self.value.weight"] = (dq, kv_in)
            P[b + blk + ".self.value.bias"] = (dq,)
            P[b + blk + ".output.dense.weight"] = (dq, dq)
            P[b + blk + ".output.dense.bias"] = (dq,)
            P[b + blk + ".output.LayerNorm.weight"] = (dq,)
            P[b + blk + ".output.LayerNorm.bias"] = (dq,)
        P[b + "intermediate_query.dense.weight"] = (fq, dq)
        P[b + "intermediate_query.dense.bias"] = (fq,)
        P[b + "output_query.dense.weight"] = (dq, fq)
        P[b + "output_query.dense.bias"] = (dq,)
        P[b + "output_query.LayerNorm.weight"] = (dq,)
        P[b + "output_query.LayerNorm.bias"] = (dq,)
    dt, ft = t["hidden_size"], t["intermediate_size"]
    P["llama_proj.weight"] = (dt, dq)
    P["llama_proj.bias"] = (dt,)
    P["llama_model.model.embed_tokens.weight"] = (t["vocab_size"], dt)
    for i in range(t["num_hidden_layers"]):
        b = "llama_model.model.layers.%d." % i
        for nm in ("q_proj", "k_proj", "v_proj", "o_proj"):
            P[b + "self_attn.%s.weight" % nm] = (dt, dt)
        P[b + "mlp.gate_proj.weight"] = (ft, dt)
        P[b + "mlp.up_proj.weight"] = (ft, dt)
        P[b + "mlp.down_proj.weight"] = (dt, ft)
        P[b + "input_layernorm.weight"] = (dt,)
        P[b + "post_attention_layernorm.weight"] = (dt,)
    P["llama_model.model.norm.weight"] = (dt,)
    P["llama_model.lm_head.weight"] = (t["vocab_size"], dt)
    return P


_VIT_LAYER = re.compile(r"^vision_model\.encoder\.layers\.(\d+)\.(.*)$")
_VIT_SUB = {"layer_norm1": "norm1", "layer_norm2": "norm2", "self_attn.qkv.weight": "attn.qkv.weight",
            "self_attn.projection": "attn.proj", "mlp.fc1": "mlp.fc1", "mlp.fc2": "mlp.fc2"}
DERIVED_QKV_BIAS = "derived:vit_qkv_bias.%d"


def blip2_alias(name: str) -> str:
    """HF BLIP-2 vision / Q-Former / projection parameter name (what Blip2Engine asks for) -> MiniGPT-4 name."""
    m = _VIT_LAYER.match(name)
    if m:
        i, rest = int(m.group(1)), m.group(2)
        if rest == "self_attn.qkv.bias":          # cat(q_bias, 0, v_bias): eva_vit.py:193-197
            return DERIVED_QKV_BIAS % i
        for k, v in _VIT_SUB.items():
            if rest.startswith(k):
                return "visual_encoder.blocks.%d.%s%s" % (i, v, rest[len(k):])
        raise KeyError(name)
    fixed = {"vision_model.embeddings.class_embedding": "visual_encoder.cls_token",
             "vision_model.embeddings.position_embedding": "visual_encoder.pos_embed",
             "vision_model.embeddings.patch_embedding.weight": "visual_encoder.patch_embed.proj.weight",
             "vision_model.embeddings.patch_embedding.bias": "visual_encoder.patch_embed.proj.bias",
             "vision_model.post_layernorm.weight": "ln_vision.weight", "vision_model.post_layernorm.bias": "ln_vision.bias",
             "qformer.layernorm.weight": "Qformer.bert.embeddings.LayerNorm.weight",
             "qformer.layernorm.bias": "Qformer.bert.embeddings.LayerNorm.bias",
             "language_projection.weight": "llama_proj.weight", "language_projection.bias": "llama_proj.bias",
             "query_tokens": "query_tokens"}
    if name in fixed:
        return fixed[name]
    if name.startswith("qformer.encoder.layer."):
        return "Qformer.bert." + name[len("qformer."):].replace("attention.attention.", "attention.self.")
    raise KeyError(name)


_MG4_BLOCK = re.compile(r"^visual_encoder\.blocks\.(\d+)\.(.*)$")


def canonical_name(name: str):
    """MiniGPT-4 state-dict name -> the canonical name of the path-level weight table (include/devqa.h, DEVQA_FAMILY_MINIGPT4: the vision
    side under the BLIP-2 names, the decoder under the LLaVA names), or None for parameters the table carries in a derived form only
    (the ViT's q_bias / v_bias: the fused qkv bias).  Inverse of blip2_alias on the vision side."""
    if name.startswith("llama_model."):
        return "language_model." + name[len("llama_model."):]
    m = _MG4_BLOCK.match(name)
    if m:
        i, rest = int(m.group(1)), m.group(2)
        if rest in ("attn.q_bias", "attn.v_bias"):
            return None
        for k, v in _VIT_SUB.items():
            if rest.startswith(v):
                return "vision_model.encoder.layers.%d.%s%s" % (i, k, rest[len(v):])
        raise KeyError(name)
    fixed = {"visual_encoder.cls_token": "vision_model.embeddings.class_embedding",
             "visual_encoder.pos_embed": "vision_model.embeddings.position_embedding",
             "visual_encoder.patch_embed.proj.weight": "vision_model.embeddings.patch_embedding.weight",
             "visual_encoder.patch_embed.proj.bias": "vision_model.embeddings.patch_embedding.bias",
             "ln_vision.weight": "vision_model.post_layernorm.weight", "ln_vision.bias": "vision_model.post_layernorm.bias",
             "Qformer.bert.embeddings.LayerNorm.weight": "qformer.layernorm.weight",
             "Qformer.bert.embeddings.LayerNorm.bias": "qformer.layernorm.bias",
             "llama_proj.weight": "language_projection.weight", "llama_proj.bias": "language_projection.bias",
             "query_tokens": "query_tokens"}
    if name in fixed:
        return fixed[name]
    if name.startswith("Qformer.bert.encoder.layer."):
        return "qformer." + name[len("Qformer.bert."):].replace("attention.self.", "attention.attention.")
    raise KeyError(name)
