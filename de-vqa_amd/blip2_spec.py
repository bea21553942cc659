"""BLIP-2-OPT architecture description: dims and the HF parameter-name map.

The reference selects edit targets by parameter NAME (R/configs/ft_vl/blip2-opt-2.7b.yaml:8,
R/editor/vllm_editors/ft_vl/ft_vl.py:31-36), so the names below are part of the drop-in
contract (SURVEY.md Appendix D "Parameter names").
"""
from collections import OrderedDict

# full-size BLIP-2-OPT-2.7B
BLIP2_OPT_2_7B = dict(
    vision_config=dict(hidden_size=1408, intermediate_size=6144, num_hidden_layers=39, num_attention_heads=16,
                       image_size=224, patch_size=14, layer_norm_eps=1e-6),
    qformer_config=dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                        cross_attention_frequency=2, encoder_hidden_size=1408, layer_norm_eps=1e-12),
    text_config=dict(hidden_size=2560, ffn_dim=10240, num_hidden_layers=32, num_attention_heads=32,
                     vocab_size=50272, max_position_embeddings=2048, pad_token_id=1),
    num_query_tokens=32,
)


def scaled_spec(vision_layers, qformer_layers, text_layers, base=BLIP2_OPT_2_7B):
    """Same per-layer dims as `base` with fewer layers (the 'real-head-dim' test scale)."""
    import copy
    s = copy.deepcopy(base)
    s["vision_config"]["num_hidden_layers"] = vision_layers
    s["qformer_config"]["num_hidden_layers"] = qformer_layers
    s["text_config"]["num_hidden_layers"] = text_layers
    return s


def param_shapes(cfg):
    """OrderedDict {hf_param_name: shape} in HF `named_parameters()` order (lm_head is tied)."""
    v, q, t = cfg["vision_config"], cfg["qformer_config"], cfg["text_config"]
    P = OrderedDict()
    Q = cfg["num_query_tokens"]
    dv, fv = v["hidden_size"], v["intermediate_size"]
    npos = (v["image_size"] // v["patch_size"]) ** 2 + 1
    P["query_tokens"] = (1, Q, q["hidden_size"])
    P["vision_model.embeddings.class_embedding"] = (1, 1, dv)
    P["vision_model.embeddings.position_embedding"] = (1, npos, dv)
    P["vision_model.embeddings.patch_embedding.weight"] = (dv, 3, v["patch_size"], v["patch_size"])
    P["vision_model.embeddings.patch_embedding.bias"] = (dv,)
    for i in range(v["num_hidden_layers"]):
        p = "vision_model.encoder.layers.%d." % i
        P[p + "self_attn.qkv.weight"] = (3 * dv, dv)
        P[p + "self_attn.qkv.bias"] = (3 * dv,)
        P[p + "self_attn.projection.weight"] = (dv, dv)
        P[p + "self_attn.projection.bias"] = (dv,)
        P[p + "layer_norm1.weight"] = (dv,)
        P[p + "layer_norm1.bias"] = (dv,)
        P[p + "mlp.fc1.weight"] = (fv, dv)
        P[p + "mlp.fc1.bias"] = (fv,)
        P[p + "mlp.fc2.weight"] = (dv, fv)
        P[p + "mlp.fc2.bias"] = (dv,)
        P[p + "layer_norm2.weight"] = (dv,)
        P[p + "layer_norm2.bias"] = (dv,)
    P["vision_model.post_layernorm.weight"] = (dv,)
    P["vision_model.post_layernorm.bias"] = (dv,)
    dq, fq, de = q["hidden_size"], q["intermediate_size"], q["encoder_hidden_size"]
    P["qformer.layernorm.weight"] = (dq,)
    P["qformer.layernorm.bias"] = (dq,)
    for i in range(q["num_hidden_layers"]):
        p = "qformer.encoder.layer.%d." % i
        blocks = [("attention.", dq)]
        if i % q["cross_attention_frequency"] == 0:
            blocks.append(("crossattention.", de))
        for blk, dkv in blocks:
            P[p + blk + "attention.query.weight"] = (dq, dq)
            P[p + blk + "attention.query.bias"] = (dq,)
            P[p + blk + "attention.key.weight"] = (dq, dkv)
            P[p + blk + "attention.key.bias"] = (dq,)
            P[p + blk + "attention.value.weight"] = (dq, dkv)
            P[p + blk + "attention.value.bias"] = (dq,)
            P[p + blk + "output.dense.weight"] = (dq, dq)
            P[p + blk + "output.dense.bias"] = (dq,)
            P[p + blk + "output.LayerNorm.weight"] = (dq,)
            P[p + blk + "output.LayerNorm.bias"] = (dq,)
        P[p + "intermediate_query.dense.weight"] = (fq, dq)
        P[p + "intermediate_query.dense.bias"] = (fq,)
        P[p + "output_query.dense.weight"] = (dq, fq)
        P[p + "output_query.dense.bias"] = (dq,)
        P[p + "output_query.LayerNorm.weight"] = (dq,)
        P[p + "output_query.LayerNorm.bias"] = (dq,)
    dt, ft = t["hidden_size"], t["ffn_dim"]
    P["language_projection.weight"] = (dt, dq)
    P["language_projection.bias"] = (dt,)
    P["language_model.model.decoder.embed_tokens.weight"] = (t["vocab_size"], dt)
    P["language_model.model.decoder.embed_positions.weight"] = (t["max_position_embeddings"] + 2, dt)
    P["language_model.model.decoder.final_layer_norm.weight"] = (dt,)
    P["language_model.model.decoder.final_layer_norm.bias"] = (dt,)
    for i in range(t["num_hidden_layers"]):
        p = "language_model.model.decoder.layers.%d." % i
        for nm in ("k_proj", "v_proj", "q_proj", "out_proj"):
            P[p + "self_attn.%s.weight" % nm] = (dt, dt)
            P[p + "self_attn.%s.bias" % nm] = (dt,)
        P[p + "self_attn_layer_norm.weight"] = (dt,)
        P[p + "self_attn_layer_norm.bias"] = (dt,)
        P[p + "fc1.weight"] = (ft, dt)
        P[p + "fc1.bias"] = (ft,)
        P[p + "fc2.weight"] = (dt, ft)
        P[p + "fc2.bias"] = (dt,)
        P[p + "final_layer_norm.weight"] = (dt,)
        P[p + "final_layer_norm.bias"] = (dt,)
    return P
