"""LlavaEngine: LLaVA-1.5 (CLIP ViT + projector + LLaMA/Vicuna decoder) forward and the FT_VL tail over the
C-ABI HIP kernels -- same interface as Blip2Engine (engine.py), so FTvl and BatchedEditEval run unchanged.

Reference call sites replaced: R/editor/vllms_for_edit/llava/llava.py:25-51 (vision tower hidden state -2 without
CLS -> multi_modal_projector -> splice at the `<image>` token) and :63-68 (language model forward).
"""
import torch

from . import lib
from .engine import PackedSeqs, plan_image_chunks


class LlavaEngine:
    def __init__(self, model):
        self.m = model
        cfg = model.cfg
        self.v, self.t = cfg["vision_config"], cfg["text_config"]
        self.dev = model.dev
        self.n_img = (self.v["image_size"] // self.v["patch_size"]) ** 2
        self.Q = self.n_img
        self.image_token_id = cfg["image_token_index"]
        self.edit_layer = self.t["num_hidden_layers"] - 1
        self.adt = model.wdtype
        self.want = "bf16" if self.adt == torch.bfloat16 else "f32"
        self.lm = "language_model."      # parameter-name prefix of the LLaMA decoder (MiniGPT-4: "llama_model.")
        self.eps = self.t["rms_norm_eps"]
        self.theta = float(self.t.get("rope_theta", 10000.0))
        self._desc_cache = {}

    def _w(self, name):
        return self.m.weight_for_gemm(name)

    def _p(self, name):
        return self.m.get(name)

    def _act(self, x32):
        return lib.cast_f32_bf16(x32) if self.adt == torch.bfloat16 else x32

    def _full_desc(self, n_seq, n):
        d = self._desc_cache.get((n_seq, n))
        if d is None:
            d = torch.tensor([[i * n, n, i * n, n, 0, 0] for i in range(n_seq)], dtype=torch.int32, device=self.dev)
            self._desc_cache[(n_seq, n)] = d
        return d

    # ---- CLIP ViT (all but the last layer) + projector: [B, n_img, d_llm] fp32 -----------------------------
    def image_chunks(self, n, max_chunk=64):
        """See Blip2Engine.image_chunks: partition that minimises 256x256-tile rounds of the CLIP GEMMs."""
        v = self.v
        d, f = v["hidden_size"], v["intermediate_size"]
        tokens = (v["image_size"] // v["patch_size"]) ** 2 + 1
        return plan_image_chunks(n, tokens, [(3 * d, d), (d, d), (f, d), (d, f)], max_chunk)

    @torch.no_grad()
    def encode_images(self, pixels):
        m, v = self.m, self.v
        m.refresh_derived()
        B = pixels.shape[0]
        P, D = v["patch_size"], v["hidden_size"]
        G = v["image_size"] // P
        N = G * G + 1
        H = v["num_attention_heads"]
        dh = D // H
        eps = v["layer_norm_eps"]
        pre = "vision_tower.vision_model."
        cols = lib.im2col_patches(pixels.contiguous(), P, m.patch_kpad, self.adt)
        patches = lib.gemm(cols, m.patch_w_gemm, want="f32")
        x = lib.vit_assemble(patches, self._p(pre + "embeddings.class_embedding"),
                             self._p(pre + "embeddings.position_embedding.weight"), B, G * G, D)
        x = lib.layernorm(x, self._p(pre + "pre_layrnorm.weight"), self._p(pre + "pre_layrnorm.bias"), eps, want="f32")
        desc = self._full_desc(B, N)
        for i in range(v["num_hidden_layers"] - 1):   # hidden_states[-2]
            p = pre + "encoder.layers.%d." % i
            h = lib.layernorm(x, self._p(p + "layer_norm1.weight"), self._p(p + "layer_norm1.bias"), eps, want=self.want)
            qkv = lib.gemm(h, m.fused_w["clip_qkv.%d" % i], m.fused_b["clip_qkv.%d" % i])
            att = lib.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], desc, B, N, H, dh, dh ** -0.5, 0)
            lib.gemm(att, self._w(p + "self_attn.out_proj.weight"), self._p(p + "self_attn.out_proj.bias"), residual=x, out_f32=x)
            h = lib.layernorm(x, self._p(p + "layer_norm2.weight"), self._p(p + "layer_norm2.bias"), eps, want=self.want)
            f = lib.gemm(h, self._w(p + "mlp.fc1.weight"), self._p(p + "mlp.fc1.bias"), act=lib.ACT_QUICK_GELU)
            lib.gemm(f, self._w(p + "mlp.fc2.weight"), self._p(p + "mlp.fc2.bias"), residual=x, out_f32=x)
        # drop CLS (row 0 of every image), project
        idx = torch.arange(B * N, dtype=torch.int32, device=self.dev).view(B, N)[:, 1:].reshape(-1).contiguous()
        f = self._act(lib.gather_rows(x, idx))
        f = lib.gemm(f, self._w("multi_modal_projector.linear_1.weight"), self._p("multi_modal_projector.linear_1.bias"),
                     act=lib.ACT_GELU)
        out = lib.gemm(f, self._w("multi_modal_projector.linear_2.weight"), self._p("multi_modal_projector.linear_2.bias"),
                       want="f32")
        return out.view(B, N - 1, -1)

    # ---- decoder input rows -----------------------------------------------------------------------------------
    @torch.no_grad()
    def pack_from_embeds(self, inputs_embeds, attention_mask):
        B, T, d = inputs_embeds.shape
        lens = attention_mask.to(torch.int64).sum(1).tolist()
        x = inputs_embeds.reshape(B * T, d).to(torch.float32).contiguous().clone()
        pos = torch.arange(T, dtype=torch.int32, device=self.dev).repeat(B).contiguous()   # LLaMA: arange positions
        desc = torch.tensor([[b * T, int(lens[b]), 0, 0, b * T, int(lens[b])] for b in range(B)], dtype=torch.int32,
                            device=self.dev)
        ps = PackedSeqs(x, [b * T for b in range(B)], [int(n) for n in lens], desc, T)
        ps.pos = pos
        return ps

    @torch.no_grad()
    def pack_from_tokens(self, seqs, img_tokens, share_prefix=False):
        """seqs: (image index or None, token ids incl. ONE image placeholder id when an image is given).
        The placeholder expands to the image's n_img feature rows.  With share_prefix the rows up to and including
        the image features ([BOS][features]) are packed once per distinct (image, leading ids) and shared through
        the attention descriptor's visible-prefix range (causal attention: they do not depend on the text)."""
        n_img = self.n_img
        tok, src, pos, desc, start, length = [], [], [], [], [], []
        prefix = {}
        r = 0
        if share_prefix:
            for (img, ids) in seqs:
                if img is None:
                    continue
                p = list(ids).index(self.image_token_id)
                key = (img, tuple(ids[:p]))
                if key in prefix:
                    continue
                n = p + n_img
                prefix[key] = (r, n)
                tok += list(ids[:p]) + [0] * n_img
                src += [-1] * p + list(range(img * n_img, (img + 1) * n_img))
                pos += list(range(n))
                desc.append([r, n, 0, 0, r, n])
                r += n
        for (img, ids) in seqs:
            ids = list(ids)
            if img is None:
                n = len(ids)
                start.append(r)
                length.append(n)
                tok += ids
                src += [-1] * n
                pos += list(range(n))
                desc.append([r, n, 0, 0, r, n])
                r += n
                continue
            p = ids.index(self.image_token_id)
            if share_prefix:
                pr, pn = prefix[(img, tuple(ids[:p]))]
                rest = ids[p + 1:]
                n = len(rest)
                start.append(r)
                length.append(n)
                tok += rest
                src += [-1] * n
                pos += list(range(pn, pn + n))
                desc.append([r, n, pr, pn, r, n])
                r += n
            else:
                n = p + n_img + len(ids) - p - 1
                start.append(r)
                length.append(n)
                tok += ids[:p] + [0] * n_img + ids[p + 1:]
                src += [-1] * p + list(range(img * n_img, (img + 1) * n_img)) + [-1] * (len(ids) - p - 1)
                pos += list(range(n))
                desc.append([r, n, 0, 0, r, n])
                r += n
        dev = self.dev
        t_pos = torch.tensor(pos, dtype=torch.int32, device=dev)
        rows = None if img_tokens is None else img_tokens.reshape(-1, img_tokens.shape[-1]).contiguous()
        emb = self.embed_table()
        x = lib.embed_rows(torch.tensor(tok, dtype=torch.int32, device=dev), torch.tensor(src, dtype=torch.int32, device=dev),
                           torch.full((len(tok),), -2, dtype=torch.int32, device=dev), emb, rows, self._zero_pos(emb))
        ps = PackedSeqs(x, start, length, torch.tensor(desc, dtype=torch.int32, device=dev),
                        max(max(length), max([d_[1] for d_ in desc])))
        ps.pos = t_pos
        return ps

    def _zero_pos(self, emb):
        """embed_rows adds pos_table[pos+2]; LLaMA has no learned positions: a 1-row zero table with pos = -2."""
        z = getattr(self, "_zp", None)
        if z is None or z.dtype != emb.dtype:
            z = torch.zeros((1, emb.shape[1]), dtype=emb.dtype, device=self.dev)
            self._zp = z
        return z

    # ---- LLaMA decoder layers -----------------------------------------------------------------------------------
    @torch.no_grad()
    def decoder_layers(self, ps, upto_layer=None, stop_before_fc2=False):
        t, m = self.t, self.m
        d, H = t["hidden_size"], t["num_attention_heads"]
        dh = d // H
        x = ps.x
        n_seq = ps.desc.shape[0]
        last = t["num_hidden_layers"] - 1 if upto_layer is None else upto_layer
        for i in range(last + 1):
            p = self.lm + "model.layers.%d." % i
            h = lib.rmsnorm(x, self._p(p + "input_layernorm.weight"), self.eps, want=self.want)
            qkv = lib.gemm(h, m.fused_w["llama_qkv.%d" % i])
            lib.rope_(qkv[:, :2 * d], ps.pos, 2 * H, dh, self.theta)          # q heads then k heads
            att = lib.attention(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], ps.desc, n_seq, ps.max_len, H, dh, dh ** -0.5, 1,
                                out=torch.zeros((x.shape[0], d), dtype=self.adt, device=self.dev))
            lib.gemm(att, self._w(p + "self_attn.o_proj.weight"), residual=x, out_f32=x)
            h = lib.rmsnorm(x, self._p(p + "post_attention_layernorm.weight"), self.eps, want=self.want)
            gu = lib.gemm(h, m.fused_w["llama_gu.%d" % i])
            a = lib.swiglu(gu)
            if stop_before_fc2 and i == last:
                return x, a
            lib.gemm(a, self._w(p + "mlp.down_proj.weight"), residual=x, out_f32=x)
        return x, None

    @torch.no_grad()
    def lm_head(self, x_rows, add=None):
        h = lib.rmsnorm(x_rows, self._p(self.lm + "model.norm.weight"), self.eps, add=add, want=self.want)
        return lib.gemm(h, self._w(self.lm + "lm_head.weight"), want="f32")

    @torch.no_grad()
    def full_logits(self, ps):
        x, _ = self.decoder_layers(ps)
        return self.lm_head(x)

    # ---- FT_VL interface ---------------------------------------------------------------------------------------------
    def edit_target(self):
        return self.lm + "model.layers.%d.mlp.down_proj.weight" % self.edit_layer

    def edit_bias(self):
        return None

    def final_norm_bwd(self, x_rows, dH, add=None):
        return lib.rmsnorm_bwd_dx(x_rows, self._p(self.lm + "model.norm.weight"), dH, self.eps, add=add)

    def embed_table(self):
        return self._p(self.lm + "model.embed_tokens.weight")
