"""LLaVA-1.5 parameter tree (CLIP ViT-L/14-336 + 2-layer projector + LLaMA/Vicuna decoder) with the
parameter names the reference's configs address (old HF layout, de-vqa_amd/llava_spec.py), backed by
device buffers laid out for the HIP kernels.  Same container mechanics as blip2/modeling.py.

Fused GEMM operands (HF parameters are row-block views): CLIP q/k/v per layer ([3d,d] + bias), LLaMA
q/k/v per layer ([3d,d]) and LLaMA gate/up per layer ([2F,d], consumed by devqa_swiglu).
"""
import json
import os
from types import SimpleNamespace

import torch

from ..blip2.modeling import Blip2Native
from ....llava_spec import new_to_old_name, old_to_new_name, param_shapes


class LlavaNative(Blip2Native):
    def __init__(self, cfg, device="cuda", dtype="bf16"):
        torch.nn.Module.__init__(self)
        assert dtype in ("bf16", "fp32")
        self.compute_dtype = dtype
        self.wdtype = torch.bfloat16 if dtype == "bf16" else torch.float32
        self.cfg = cfg
        v, t = cfg["vision_config"], cfg["text_config"]
        self.config = SimpleNamespace(is_encoder_decoder=False, image_token_index=cfg["image_token_index"],
                                      text_config=SimpleNamespace(**t), vision_config=SimpleNamespace(**v),
                                      vision_feature_layer=-2, vision_feature_select_strategy="default")
        self.dev = torch.device(device)
        self._shapes = param_shapes(cfg)
        self._fp32_masters = {}
        self._build()

    def _small_f32_names(self):
        p = "vision_tower.vision_model.embeddings."
        return (p + "class_embedding", p + "position_embedding.weight", p + "patch_embedding.weight")

    def _fused_slot(self, name):
        if ".self_attn." in name:
            kind = name.split("self_attn.")[1].split(".")[0]
            if kind in ("q_proj", "k_proj", "v_proj"):
                slot = {"q_proj": 0, "k_proj": 1, "v_proj": 2}[kind]
                if name.startswith("vision_tower."):
                    return ("clip_qkv." + name.split("encoder.layers.")[1].split(".")[0], slot, 3)
                return ("llama_qkv." + name.split("model.layers.")[1].split(".")[0], slot, 3)
        if ".mlp.gate_proj." in name or ".mlp.up_proj." in name:
            return ("llama_gu." + name.split("model.layers.")[1].split(".")[0], 0 if "gate_proj" in name else 1, 2)
        return None

    @torch.no_grad()
    def refresh_derived(self, force=False):
        self.gu_interleaved()          # (keeps the interleaved [gate | up] operands current; no-op when nothing was written)
        pw = self.get("vision_tower.vision_model.embeddings.patch_embedding.weight")
        head = self.get("language_model.lm_head.weight")
        ver = (pw._version, head._version, pw.data_ptr(), head.data_ptr())
        if not force and ver == self._derived_version:
            return
        self.patch_w_gemm.zero_()
        self.patch_w_gemm[:, :self.patch_kreal] = pw.reshape(pw.shape[0], -1).to(self.wdtype)
        if self.embed_T is None:
            self.embed_T = head.t().contiguous()
        else:
            self.embed_T.copy_(head.t())      # in place: the path-level context holds this buffer's address
        self._derived_version = ver

    def weight_table(self):
        """{canonical name: device tensor} for lib.PathContext (include/devqa.h, DEVQA_FAMILY_LLAVA): every parameter that is not a row
        block of a fused operand under its own name, the fused operands as derived.clip_qkv.<i>.{weight,bias} / derived.llama_qkv.<i>.weight /
        derived.llama_gu.<i>.weight, the patch-embedding GEMM operand and lm_head^T."""
        from collections import OrderedDict
        self.refresh_derived()
        t = OrderedDict()
        for name, p_ in self.named_parameters():
            if self._fused_slot(name) is not None:
                continue
            t[name] = p_.data
            ent = self._fp32_masters.get(name)
            if ent is not None and self.wdtype != torch.float32:
                t[name + "#shadow"] = ent[1]
        for key, w in self.fused_w.items():
            t["derived.%s.weight" % key] = w
        for key, b in self.fused_b.items():
            t["derived.%s.bias" % key] = b
        for layer, w in self.gu_interleaved().items():
            t["derived.llama_gu_il.%d.weight" % layer] = w
        t["derived.patch_w_gemm"] = self.patch_w_gemm
        t["derived.embed_T"] = self.embed_T
        return t

    @classmethod
    def from_pretrained_dir(cls, path, device="cuda", dtype="bf16"):
        """HF LLaVA directory (config.json + *.safetensors in either naming) or a devqa fixture directory
        (devqa_llava_config.json)."""
        from safetensors import safe_open
        fx = os.path.join(path, "devqa_llava_config.json")
        if os.path.exists(fx):
            cfg = json.load(open(fx))
        else:
            hf = json.load(open(os.path.join(path, "config.json")))
            cfg = {"vision_config": hf["vision_config"], "text_config": hf["text_config"],
                   "image_token_index": hf.get("image_token_index", hf.get("image_token_id", 32000))}
        model = cls(cfg, device, dtype)
        files = [f for f in sorted(os.listdir(path)) if f.endswith(".safetensors")]
        handles = [safe_open(os.path.join(path, f), framework="pt", device="cpu") for f in files]
        key2h = {}
        for h in handles:
            for k in h.keys():
                key2h[k] = h

        def get_tensor(name):
            for cand in (name, old_to_new_name(name)):
                if cand in key2h:
                    return key2h[cand].get_tensor(cand)
            raise KeyError(name)
        model.load_named_tensors(get_tensor)
        return model

    @classmethod
    def from_synth(cls, cfg, seed, style="llava", device="cuda", dtype="bf16"):
        from ....synth import param_init
        model = cls(cfg, device, dtype)
        model.load_named_tensors(lambda n: torch.from_numpy(param_init(n, model._shapes[n], seed, style)))
        return model
