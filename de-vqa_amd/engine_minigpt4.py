"""MiniGPT4Engine: EVA ViT-g + Q-Former + llama_proj (the BLIP-2 vision engine under MiniGPT-4 parameter names) in
front of the LLaMA decoder engine -- same interface as Blip2Engine / LlavaEngine, so FTvl and BatchedEditEval run
unchanged.  Replaces R/editor/vllms_for_edit/minigpt4/modules/minigpt4.py:214-241 (`encode_img`) and the
`llama_model(inputs_embeds=...)` forward of R/editor/vllms_for_edit/minigpt4/minigpt4.py:63-68.
"""
import torch

from . import lib
from .engine import Blip2Engine
from .engine_llava import LlavaEngine
from .minigpt4_spec import blip2_alias

IMG_PLACEHOLDER = -1   # token-id sentinel marking where the 32 query rows go in a pre-tokenised sequence


class _VisionPart(Blip2Engine):
    """Blip2Engine.encode_images with parameter names translated to the MiniGPT-4 tree."""

    def _w(self, name):
        return self.m.weight_for_gemm(blip2_alias(name))

    def _p(self, name):
        return self.m.get(blip2_alias(name))


class MiniGPT4Engine(LlavaEngine):
    def __init__(self, model):
        self.m = model
        cfg = model.cfg
        self.v, self.t = cfg["vision_config"], cfg["text_config"]
        self.dev = model.dev
        self.n_img = cfg["num_query_tokens"]
        self.Q = self.n_img
        self.image_token_id = IMG_PLACEHOLDER
        self.edit_layer = self.t["num_hidden_layers"] - 1
        self.adt = model.wdtype
        self.want = "bf16" if self.adt == torch.bfloat16 else "f32"
        self.lm = "llama_model."
        self.eps = self.t["rms_norm_eps"]
        self.theta = float(self.t.get("rope_theta", 10000.0))
        self._desc_cache = {}
        self.vision = _VisionPart(model)

    FAMILY = lib.FAMILY_MINIGPT4

    def _model_desc(self):
        d = super()._model_desc()
        q = self.m.cfg["qformer_config"]
        d.v_run_layers = self.v["num_hidden_layers"]
        d.q_hidden, d.q_layers, d.q_heads, d.q_ffn = q["hidden_size"], q["num_hidden_layers"], q["num_attention_heads"], q["intermediate_size"]
        d.q_cross_freq, d.q_ln_eps = q["cross_attention_frequency"], q["layer_norm_eps"]
        return d

    def image_chunks(self, n, max_chunk=128):
        return self.vision.image_chunks(n, max_chunk)

    def encode_images(self, pixels):
        """pixel_values fp32 [B,3,S,S] -> llama_proj(Q-Former(ln_vision(ViT))) fp32 [B, 32, d_llm]"""
        ctx = self.path_ctx()
        if ctx is not None:
            return ctx.vision_encode(pixels.contiguous())
        return self.vision.encode_images(pixels)
