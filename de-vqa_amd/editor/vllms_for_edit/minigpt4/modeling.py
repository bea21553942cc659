"""MiniGPT-4 parameter tree with the reference's state-dict names (visual_encoder.*, ln_vision.*, Qformer.bert.*,
query_tokens, llama_proj.*, llama_model.*: R/editor/vllms_for_edit/minigpt4/modules/minigpt4.py:44-75), backed by
device buffers laid out for the HIP kernels.  Same container mechanics as blip2/modeling.py; fused GEMM operands:
LLaMA q/k/v ([3d, d]) and gate/up ([2F, d]) per layer.  `get()` additionally serves the derived ViT qkv bias
cat(q_bias, 0, v_bias) (eva_vit.py:193-197) under `derived:vit_qkv_bias.<layer>`.
"""
import json
import os
from types import SimpleNamespace

import torch

from ..blip2.modeling import Blip2Native
from ....minigpt4_spec import param_shapes


class MiniGPT4Native(Blip2Native):
    def __init__(self, cfg, device="cuda", dtype="bf16"):
        torch.nn.Module.__init__(self)
        assert dtype in ("bf16", "fp32")
        self.compute_dtype = dtype
        self.wdtype = torch.bfloat16 if dtype == "bf16" else torch.float32
        self.cfg = cfg
        self.config = SimpleNamespace(is_encoder_decoder=False, num_query_tokens=cfg["num_query_tokens"],
                                      text_config=SimpleNamespace(**cfg["text_config"]),
                                      vision_config=SimpleNamespace(**cfg["vision_config"]),
                                      qformer_config=SimpleNamespace(**cfg["qformer_config"]))
        self.dev = torch.device(device)
        self._shapes = param_shapes(cfg)
        self._fp32_masters = {}
        self._build()

    def _small_f32_names(self):
        return ("query_tokens", "visual_encoder.cls_token", "visual_encoder.pos_embed", "visual_encoder.patch_embed.proj.weight")

    def _fused_slot(self, name):
        if name.startswith("llama_model.model.layers."):
            layer = name.split("model.layers.")[1].split(".")[0]
            if ".self_attn." in name:
                kind = name.split("self_attn.")[1].split(".")[0]
                if kind in ("q_proj", "k_proj", "v_proj"):
                    return ("llama_qkv." + layer, {"q_proj": 0, "k_proj": 1, "v_proj": 2}[kind], 3)
            if ".mlp.gate_proj." in name or ".mlp.up_proj." in name:
                return ("llama_gu." + layer, 0 if "gate_proj" in name else 1, 2)
        return None

    def _build_derived(self):
        super()._build_derived()
        v = self.cfg["vision_config"]
        self.vit_qkv_bias = [torch.zeros((3 * v["hidden_size"],), dtype=torch.float32, device=self.dev)
                             for _ in range(v["num_hidden_layers"])]

    def get(self, name):
        if name.startswith("derived:vit_qkv_bias."):
            return self.vit_qkv_bias[int(name.rsplit(".", 1)[1])]
        return super().get(name)

    @torch.no_grad()
    def refresh_derived(self, force=False):
        self.gu_interleaved()          # (keeps the interleaved [gate | up] operands current; no-op when nothing was written)
        pw = self.get("visual_encoder.patch_embed.proj.weight")
        head = self.get("llama_model.lm_head.weight")
        b0 = self.get("visual_encoder.blocks.0.attn.q_bias")
        ver = (pw._version, head._version, b0._version, pw.data_ptr(), head.data_ptr(), b0.data_ptr())
        if not force and ver == self._derived_version:
            return
        self.patch_w_gemm.zero_()
        self.patch_w_gemm[:, :self.patch_kreal] = pw.reshape(pw.shape[0], -1).to(self.wdtype)
        if self.embed_T is None:
            self.embed_T = head.t().contiguous()
        else:
            self.embed_T.copy_(head.t())      # in place: the path-level context holds this buffer's address
        D = self.cfg["vision_config"]["hidden_size"]
        for i, buf in enumerate(self.vit_qkv_bias):
            buf.zero_()
            buf[:D] = self.get("visual_encoder.blocks.%d.attn.q_bias" % i)
            buf[2 * D:] = self.get("visual_encoder.blocks.%d.attn.v_bias" % i)
        self._derived_version = ver

    def weight_table(self):
        """{canonical name: device tensor} for lib.PathContext (include/devqa.h, DEVQA_FAMILY_MINIGPT4): the vision side under the BLIP-2
        names, the decoder under the LLaVA names (minigpt4_spec.canonical_name), the ViT's fused qkv bias cat(q_bias, 0, v_bias)
        (eva_vit.py:193-197) as vision_model.encoder.layers.<i>.self_attn.qkv.bias, the fused LLaMA operands as derived.llama_*."""
        from collections import OrderedDict
        from ....minigpt4_spec import canonical_name
        self.refresh_derived()
        t = OrderedDict()
        for name, p_ in self.named_parameters():
            if self._fused_slot(name) is not None:
                continue
            cn = canonical_name(name)
            if cn is None:
                continue
            t[cn] = p_.data
            ent = self._fp32_masters.get(name)
            if ent is not None and self.wdtype != torch.float32:
                t[cn + "#shadow"] = ent[1]
        for i, b in enumerate(self.vit_qkv_bias):
            t["vision_model.encoder.layers.%d.self_attn.qkv.bias" % i] = b
        for key, w in self.fused_w.items():
            t["derived.%s.weight" % key] = w
        for layer, w in self.gu_interleaved().items():
            t["derived.llama_gu_il.%d.weight" % layer] = w
        t["derived.patch_w_gemm"] = self.patch_w_gemm
        t["derived.embed_T"] = self.embed_T
        return t

    def storage_fingerprint(self):
        return hash((super().storage_fingerprint(), tuple(b.data_ptr() for b in self.vit_qkv_bias)))

    @classmethod
    def from_pretrained_dir(cls, path, device="cuda", dtype="bf16"):
        """A directory with `devqa_minigpt4_config.json` (the spec dict) and *.safetensors holding the reference's
        state-dict names.  (The reference assembles the model from five upstream pickles -- eva_vit_g.pth,
        blip2_pretrained_flant5xxl.pth, prerained_minigpt4_7b.pth, the Vicuna directory, bert-base config,
        minigpt4.py:14-21; convert them once to safetensors with those key names.)"""
        from safetensors import safe_open
        cfg = json.load(open(os.path.join(path, "devqa_minigpt4_config.json")))
        model = cls(cfg, device, dtype)
        files = [f for f in sorted(os.listdir(path)) if f.endswith(".safetensors")]
        handles = [safe_open(os.path.join(path, f), framework="pt", device="cpu") for f in files]
        key2h = {k: h for h in handles for k in h.keys()}
        model.load_named_tensors(lambda n: key2h[n].get_tensor(n))
        return model

    @classmethod
    def from_synth(cls, cfg, seed, style="llava", device="cuda", dtype="bf16"):
        from ....synth import param_init
        model = cls(cfg, device, dtype)
        model.load_named_tensors(lambda n: torch.from_numpy(param_init(n, model._shapes[n], seed, style)))
        return model
