"""CPU oracle for the LLaVA-1.5 path (CLIP ViT + 2-layer projector + LLaMA decoder) -- TEST INFRASTRUCTURE.

Restates what R/editor/vllms_for_edit/llava/llava.py:25-68 computes (vision tower hidden state -2, CLS
dropped, projector, image features spliced at the `<image>` token, LLaMA forward) in plain PyTorch fp32 on a
flat {old_hf_name: tensor} dict; the wrapper's auto-prefix of '<image>\\n' is
R/editor/vllms_for_edit/base.py:50-51.  The transformer arithmetic is third-party (`transformers`: CLIP, LLaMA).
The reference's LlavaForEdit cannot run on the installed transformers; this oracle is pinned by
tests/golden/tiny_llava_goldens.* = HF LlavaForConditionalGeneration + the reference's own FTvl / evaluator run
on a compat adapter (tools/make_goldens_llava.py).  FT_VL / evaluator restatements are shared with
oracle/devqa_oracle.py.
"""
import json
import os

import numpy as np
import torch
import torch.nn.functional as F

from .devqa_oracle import CLIP_MEAN, CLIP_STD, OracleBlip2, OracleTokenizer, _lin, _ln


def _rms(x, w, eps):
    v = x.pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(v + eps))


def _rope(q, k, pos, theta):
    dh = q.shape[-1]
    inv = 1.0 / (theta ** (torch.arange(0, dh, 2, dtype=torch.float32) / dh))
    fr = pos.to(torch.float32)[:, :, None] * inv[None, None, :]           # [B,T,dh/2]
    emb = torch.cat([fr, fr], -1)[:, None]                                 # [B,1,T,dh]
    cos, sin = emb.cos(), emb.sin()

    def rot(x):
        x1, x2 = x[..., :dh // 2], x[..., dh // 2:]
        return torch.cat([-x2, x1], -1)
    return q * cos + rot(q) * sin, k * cos + rot(k) * sin


class OracleLlava(OracleBlip2):
    lm_prefix = "language_model."   # parameter-name prefix of the LLaMA decoder (MiniGPT-4 oracle: "llama_model.")

    def __init__(self, weights, cfg, tokenizer, copy=True):
        self.w = {k: (v.detach().to(torch.float32).clone() if copy else v) for k, v in weights.items()}
        self.cfg = cfg
        self.tok = tokenizer
        v, t = cfg["vision_config"], cfg["text_config"]
        self.v_layers, self.v_heads, self.v_eps = v["num_hidden_layers"], v["num_attention_heads"], v["layer_norm_eps"]
        self.patch, self.image_size = v["patch_size"], v["image_size"]
        self.t_layers, self.t_heads = t["num_hidden_layers"], t["num_attention_heads"]
        self.t_eps, self.theta = t["rms_norm_eps"], t["rope_theta"]
        self.image_token_id = cfg["image_token_index"]
        self.n_img = (self.image_size // self.patch) ** 2
        self.num_query_tokens = self.n_img

    @classmethod
    def from_pretrained_dir(cls, path):
        from safetensors.torch import load_file
        cfg = json.load(open(os.path.join(path, "devqa_llava_config.json")))
        w = load_file(os.path.join(path, "model.safetensors"))
        tok = OracleTokenizer(os.path.join(path, "tokenizer.json"), cfg["text_config"].get("pad_token_id", 3))
        return cls(w, cfg, tok)

    # HF CLIPImageProcessor: RGB, resize shortest edge -> S (bicubic), center crop SxS, 1/255, CLIP mean/std
    def preprocess_image(self, img):
        from PIL import Image
        if isinstance(img, str):
            with Image.open(img) as im:
                img = im.copy()
        img = img.convert("RGB")
        S = self.image_size
        w, h = img.size
        short, long = (w, h) if w <= h else (h, w)
        new_short, new_long = S, int(S * long / short)
        nw, nh = (new_short, new_long) if w <= h else (new_long, new_short)
        img = img.resize((nw, nh), resample=Image.BICUBIC)
        left, top = (nw - S) // 2, (nh - S) // 2
        img = img.crop((left, top, left + S, top + S))
        a = np.asarray(img).astype(np.float32) * np.float32(1.0 / 255.0)
        a = (a - np.asarray(CLIP_MEAN, np.float32)) / np.asarray(CLIP_STD, np.float32)
        return torch.from_numpy(a.transpose(2, 0, 1)[None].copy())

    def vision_features(self, pixel_values):
        """CLIP ViT hidden state -2 without CLS, projected: [B, n_img, d_llm]."""
        w = self.w
        p = "vision_tower.vision_model."
        x = F.conv2d(pixel_values, w[p + "embeddings.patch_embedding.weight"], None, stride=self.patch)
        x = x.flatten(2).transpose(1, 2)
        B = x.shape[0]
        x = torch.cat([w[p + "embeddings.class_embedding"].expand(B, 1, -1), x], 1)
        x = x + w[p + "embeddings.position_embedding.weight"][None, :x.shape[1]]
        x = _ln(x, w[p + "pre_layrnorm.weight"], w[p + "pre_layrnorm.bias"], self.v_eps)
        H = self.v_heads
        for i in range(self.v_layers - 1):   # hidden_states[-2] = output of layer L-2
            q = p + "encoder.layers.%d." % i
            h = _ln(x, w[q + "layer_norm1.weight"], w[q + "layer_norm1.bias"], self.v_eps)
            Bn, N, D = h.shape
            dh = D // H
            qq = _lin(h, w[q + "self_attn.q_proj.weight"], w[q + "self_attn.q_proj.bias"]).view(Bn, N, H, dh).transpose(1, 2)
            kk = _lin(h, w[q + "self_attn.k_proj.weight"], w[q + "self_attn.k_proj.bias"]).view(Bn, N, H, dh).transpose(1, 2)
            vv = _lin(h, w[q + "self_attn.v_proj.weight"], w[q + "self_attn.v_proj.bias"]).view(Bn, N, H, dh).transpose(1, 2)
            a = torch.softmax(torch.matmul(qq, kk.transpose(-1, -2)) * dh ** -0.5, -1)
            o = torch.matmul(a, vv).transpose(1, 2).reshape(Bn, N, D)
            x = x + _lin(o, w[q + "self_attn.out_proj.weight"], w[q + "self_attn.out_proj.bias"])
            h = _ln(x, w[q + "layer_norm2.weight"], w[q + "layer_norm2.bias"], self.v_eps)
            h = _lin(h, w[q + "mlp.fc1.weight"], w[q + "mlp.fc1.bias"])
            h = h * torch.sigmoid(1.702 * h)
            x = x + _lin(h, w[q + "mlp.fc2.weight"], w[q + "mlp.fc2.bias"])
        f = x[:, 1:]
        f = F.gelu(_lin(f, w["multi_modal_projector.linear_1.weight"], w["multi_modal_projector.linear_1.bias"]))
        return _lin(f, w["multi_modal_projector.linear_2.weight"], w["multi_modal_projector.linear_2.bias"])

    def get_llm_input_embeds(self, texts, imgs=None):
        if isinstance(imgs, list) and all(i is None for i in imgs):
            imgs = None
        if imgs is not None:  # auto_add_img_special_token (R/editor/vllms_for_edit/base.py:50-51)
            texts = ["<image>\n" + t if t.find("<image>") == -1 else t for t in texts]
        ids, msk = self._tok_batch(texts)
        emb = self.w["language_model.model.embed_tokens.weight"][ids]
        if imgs is None:
            return {"attention_mask": msk, "inputs_embeds": emb, "position_ids": None}, None
        assert ids.shape[0] == 1
        feats = self.vision_features(torch.cat([self.preprocess_image(i) for i in imgs]))
        pos = int(torch.where(ids[0] == self.image_token_id)[0][0])
        emb = torch.cat([emb[:, :pos], feats, emb[:, pos + 1:]], 1)
        msk = torch.ones(emb.shape[:2], dtype=torch.long)
        return {"attention_mask": msk, "inputs_embeds": emb, "position_ids": None}, [pos, pos + self.n_img]

    def llm_hidden(self, inputs_embeds, attention_mask):
        w, H = self.w, self.t_heads
        B, T, D = inputs_embeds.shape
        dh = D // H
        pos = torch.arange(T)[None].expand(B, T)
        neg = torch.finfo(torch.float32).min
        allow = torch.ones(T, T, dtype=torch.bool).tril()[None, None] & attention_mask.bool()[:, None, None, :]
        bias = torch.zeros(B, 1, T, T).masked_fill(~allow, neg)
        x = inputs_embeds
        for i in range(self.t_layers):
            p = self.lm_prefix + "model.layers.%d." % i
            h = _rms(x, w[p + "input_layernorm.weight"], self.t_eps)
            q = _lin(h, w[p + "self_attn.q_proj.weight"]).view(B, T, H, dh).transpose(1, 2)
            k = _lin(h, w[p + "self_attn.k_proj.weight"]).view(B, T, H, dh).transpose(1, 2)
            v = _lin(h, w[p + "self_attn.v_proj.weight"]).view(B, T, H, dh).transpose(1, 2)
            q, k = _rope(q, k, pos, self.theta)
            a = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) * dh ** -0.5 + bias, -1)
            o = torch.matmul(a, v).transpose(1, 2).reshape(B, T, D)
            x = x + _lin(o, w[p + "self_attn.o_proj.weight"])
            h = _rms(x, w[p + "post_attention_layernorm.weight"], self.t_eps)
            hook = getattr(self, "module_hook", None)   # (module name, input, output) -> output; MEND-style editors
            g1, u1 = _lin(h, w[p + "mlp.gate_proj.weight"]), _lin(h, w[p + "mlp.up_proj.weight"])
            if hook is not None:
                g1, u1 = hook(p + "mlp.gate_proj", h, g1), hook(p + "mlp.up_proj", h, u1)
            a1 = F.silu(g1) * u1
            d1 = _lin(a1, w[p + "mlp.down_proj.weight"])
            if hook is not None:
                d1 = hook(p + "mlp.down_proj", a1, d1)
            x = x + d1
        return x

    def get_llm_outpt(self, llm_inpt, vt_range=None):
        x = self.llm_hidden(llm_inpt["inputs_embeds"], llm_inpt["attention_mask"])
        x = _rms(x, self.w[self.lm_prefix + "model.norm.weight"], self.t_eps)
        return _lin(x, self.w[self.lm_prefix + "lm_head.weight"])
