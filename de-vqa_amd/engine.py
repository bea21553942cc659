"""Blip2Engine: the BLIP-2-OPT forward / FT_VL backward schedule over the C-ABI HIP kernels.

Python here only orders kernel launches on the current HIP stream and owns no arithmetic.
Activations are packed row-major [rows, D]; the residual stream is fp32, GEMM operands bf16.

Reference call sites replaced (R/ = /root/reference/DE-VQA/):
  encode_images      <- vision_model -> qformer -> language_projection  (R/editor/vllms_for_edit/blip2/blip2.py:25-45)
  decoder_*          <- language_model(inputs_embeds, attention_mask)    (blip2.py:68-75)
  tail_logits        <- last-layer fc2 + residual + final LN + lm_head on the label rows only
  ft_edit_batch      <- FTvl.execute_ft hot loop                         (R/editor/vllm_editors/ft_vl/ft_vl.py:111-146)
"""
import os
import re
from dataclasses import dataclass, field
from typing import List, Optional

import torch

from . import lib

LN_EPS_OPT = 1e-5


def plan_image_chunks(n, tokens, gemm_shapes, max_chunk=128, n_cu=256, tile=256):
    """Partition n images into chunk sizes minimising sum over chunks and GEMMs of ceil(tiles/n_cu) * K."""
    def cost(c):
        tm = -(-c * tokens // tile)
        return sum(-(-(tm * -(-N // tile)) // n_cu) * K for N, K in gemm_shapes)
    cs = [0] + [cost(c) for c in range(1, min(n, max_chunk) + 1)]
    best = [0] + [None] * n
    pick = [0] * (n + 1)
    for i in range(1, n + 1):
        for c in range(1, min(i, max_chunk) + 1):
            v = best[i - c] + cs[c]
            if best[i] is None or v < best[i] or (v == best[i] and c > pick[i]):
                best[i], pick[i] = v, c
    out = []
    while n > 0:
        out.append(pick[n])
        n -= pick[n]
    return out


@dataclass
class PackedSeqs:
    """Sequences packed along rows.  Row r of sequence s sits at start[s] + r."""
    x: torch.Tensor                 # fp32 [R, d]  embeddings + positions (decoder input)
    start: List[int]
    length: List[int]               # real (unpadded) length per sequence
    desc: torch.Tensor              # int32 [n_seq, 6] attention descriptor (device)
    max_len: int
    dense: bool = False             # every row is a query row of some sequence (no padded rows): outputs need no zero fill


class Blip2Engine:
    def __init__(self, model):
        self.m = model
        cfg = model.cfg
        self.v, self.q, self.t = cfg["vision_config"], cfg["qformer_config"], cfg["text_config"]
        self.Q = cfg["num_query_tokens"]
        self.dev = model.dev
        self.edit_layer = self.t["num_hidden_layers"] - 1
        self._seq_desc_cache = {}
        # activation dtype follows the model's compute mode: bf16 GEMM operands, or fp32 ("faithful")
        self.adt = model.wdtype
        self.want = "bf16" if self.adt == torch.bfloat16 else "f32"

    # ---- path-level context (include/devqa.h "PATH LEVEL"): the schedules below live behind the C ABI ---------------------------
    def path_ctx(self):
        """lib.PathContext over this model's weight table, or None when the schedule must stay in Python: a subclass that
        renames parameters (MiniGPT-4's vision part), editor hooks registered on the engine (MEND_VL module deltas, TP_VL extra
        neurons), or DEVQA_PATH_ABI=0 (A/B tests of the two drivers of the same kernels)."""
        if type(self) is not Blip2Engine or not hasattr(self.m, "weight_table") or os.environ.get("DEVQA_PATH_ABI", "1") == "0":
            return None
        # the context caches raw device pointers of the whole weight table: rebuild it whenever any of those buffers has moved (an
        # editor promoted a new edit target, a parameter was reassigned out of place or reloaded, the model changed device)
        self.m.refresh_derived()
        fp = self.m.storage_fingerprint()
        if self.__dict__.get("_ctx") is None or self._ctx_fp != fp:
            v, q, t = self.v, self.q, self.t
            d = lib.ModelDesc(family=lib.FAMILY_BLIP2_OPT, compute_dtype=lib.DTYPE_BF16 if self.adt == torch.bfloat16 else lib.DTYPE_F32,
                              image_size=v["image_size"], patch_size=v["patch_size"], v_hidden=v["hidden_size"], v_layers=v["num_hidden_layers"],
                              v_heads=v["num_attention_heads"], v_ffn=v["intermediate_size"], q_hidden=q["hidden_size"],
                              q_layers=q["num_hidden_layers"], q_heads=q["num_attention_heads"], q_ffn=q["intermediate_size"],
                              q_cross_freq=q["cross_attention_frequency"], num_query_tokens=self.Q, t_hidden=t["hidden_size"],
                              t_layers=t["num_hidden_layers"], t_heads=t["num_attention_heads"], t_ffn=t["ffn_dim"], t_vocab=t["vocab_size"],
                              t_max_pos=t["max_position_embeddings"], v_ln_eps=v["layer_norm_eps"], q_ln_eps=q["layer_norm_eps"],
                              t_ln_eps=LN_EPS_OPT)
            old = self.__dict__.get("_ctx")
            if old is not None:
                torch.cuda.current_stream(self.dev).synchronize()
                old.close()
            self._ctx = lib.PathContext(self.dev.index or 0, d, self.m.weight_table())
            self._ctx_fp = fp
        self.m.refresh_shadows()
        return self._ctx

    def _hooks_active(self):
        return bool(getattr(self, "module_deltas", None)) or bool(getattr(self, "extra_neurons", None))

    def _act(self, x32):
        """fp32 rows -> GEMM operand dtype"""
        return lib.cast_f32_bf16(x32) if self.adt == torch.bfloat16 else x32

    def _ln(self, x, wname, bname, eps, add=None):
        return lib.layernorm(x, self._p(wname), self._p(bname), eps, add=add, want=self.want)

    # ------------------------------------------------------------------------------------------
    def _w(self, name):
        return self.m.weight_for_gemm(name)

    def _p(self, name):
        return self.m.get(name)

    def _full_desc(self, n_seq, q_len, kv_len, self_rows=False):
        """n_seq independent sequences: q rows [i*q_len, ...), keys [i*kv_len, ...), all visible."""
        key = (n_seq, q_len, kv_len, self_rows)
        d = self._seq_desc_cache.get(key)
        if d is None:
            # self-attention: the keys are the sequence's OWN rows (fully visible without the causal bit) -- the form the self_full
            # promise of devqa_attention describes; cross-attention: the keys are a visible range of another row set
            rows = [[i * q_len, q_len, 0, 0, i * q_len, q_len] if (q_len == kv_len and self_rows) else [i * q_len, q_len, i * kv_len, kv_len, 0, 0]
                    for i in range(n_seq)]
            d = torch.tensor(rows, dtype=torch.int32, device=self.dev)
            torch.cuda.current_stream(self.dev).synchronize()   # cached across streams (prefetch thread): publish it complete
            self._seq_desc_cache[key] = d
        return d

    def image_chunks(self, n, max_chunk=None):
        """Split n images into encode_images() calls.  The ViT GEMMs run 256x256 tiles, one workgroup per CU, so a call
        costs ceil(tiles / 256) rounds per GEMM; pick the partition of n that minimises the summed rounds x K (DP over
        chunk sizes) instead of a fixed chunk."""
        if max_chunk is None:
            max_chunk = int(os.environ.get("DEVQA_VIT_CHUNK", "512"))   # 127-image chunks: 254 cycles/s, 255: 261, one call for 508 images: 264.5
        return plan_image_chunks(n, self._vit_tokens(), self._vit_gemm_shapes(), max_chunk)

    def _vit_tokens(self):
        return (self.v["image_size"] // self.v["patch_size"]) ** 2 + 1

    def _vit_gemm_shapes(self):
        d, f = self.v["hidden_size"], self.v["intermediate_size"]
        return [(3 * d, d), (d, d), (f, d), (d, f)]

    # ------------------------------------------------------------------------------------------
    # K2-K5: images -> projected query tokens [B, Q, d_llm] (fp32)
    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def encode_images(self, pixels):
        ctx = self.path_ctx()
        if ctx is not None:
            return ctx.vision_encode(pixels.contiguous())
        img, B = self.vit_rows(pixels)
        return self.qformer_rows(img, B)

    @torch.no_grad()
    def vit_rows(self, pixels):
        """K2-K3 in Python-ordered launches: pixel_values fp32 [B,3,S,S] -> (post-LayerNorm ViT rows [B*N, D] in the operand dtype -- the keys /
        values of every Q-Former cross-attention --, B)"""
        x, B = self.vit_embed(pixels)
        return self.vit_post(self.vit_layers(x, B)), B

    @torch.no_grad()
    def vit_embed(self, pixels, save=None):
        """patch embedding + class token + positions -> (fp32 residual stream [B*N, D], B)"""
        m, v = self.m, self.v
        m.refresh_derived()
        B = pixels.shape[0]
        P, D = v["patch_size"], v["hidden_size"]
        G = v["image_size"] // P
        cols = lib.im2col_patches(pixels.contiguous(), P, m.patch_kpad, self.adt)
        patches = lib.gemm(cols, m.patch_w_gemm, self._p("vision_model.embeddings.patch_embedding.bias"), want="f32")
        x = lib.vit_assemble(patches, self._p("vision_model.embeddings.class_embedding"),
                             self._p("vision_model.embeddings.position_embedding"), B, G * G, D)
        if save is not None:
            save.update(cols=cols, B=B, n_patches=G * G)
        return x, B

    @torch.no_grad()
    def vit_embed_backward(self, save, dx, grads):
        """dx fp32 [B*N, D]: gradient w.r.t. the assembled ViT input rows -> position / class embeddings (sums over the images), the patch
        convolution as the GEMM it runs as (dW = dPatches^T . im2col rows, bias = column sums)"""
        B, Np = save["B"], save["n_patches"]
        D = dx.shape[1]
        e = "vision_model.embeddings."
        dxv = dx.view(B, Np + 1, D)
        grads[e + "position_embedding"].add_(dxv.sum(0).view_as(grads[e + "position_embedding"]))
        grads[e + "class_embedding"].add_(dxv[:, 0].sum(0).view_as(grads[e + "class_embedding"]))
        dP = dxv[:, 1:].reshape(B * Np, D).contiguous()
        lib.colsum_(dP, grads[e + "patch_embedding.bias"], True)
        gw = torch.zeros((D, self.m.patch_kpad), dtype=torch.float32, device=dx.device)
        lib.gemm(self._padk(dP.t()), self._padk(save["cols"].to(torch.float32).t()), residual=gw, out_f32=gw)
        grads[e + "patch_embedding.weight"].add_(gw[:, :self.m.patch_kreal].reshape(grads[e + "patch_embedding.weight"].shape))

    @torch.no_grad()
    def vit_layers(self, x, B, first_layer=0, save=None, end_layer=None):
        """encoder layers [first_layer, end_layer or all) in place on the fp32 residual stream x.  save: dict that receives what vit_backward
        needs per layer"""
        v = self.v
        D = v["hidden_size"]
        N = (v["image_size"] // v["patch_size"]) ** 2 + 1
        H = v["num_attention_heads"]
        dh = D // H
        eps = v["layer_norm_eps"]
        desc = self._full_desc(B, N, N, self_rows=True)
        if save is not None:
            save.update(B=B, N=N, desc=desc, first=first_layer, layers={})
        for i in range(first_layer, v["num_hidden_layers"] if end_layer is None else end_layer):
            p = "vision_model.encoder.layers.%d." % i
            rec = None
            if save is not None:
                rec = save["layers"][i] = {"x_in": x.clone()}
            h = self._ln(x, p + "layer_norm1.weight", p + "layer_norm1.bias", eps)
            qkv = lib.gemm(h, self._w(p + "self_attn.qkv.weight"), self._p(p + "self_attn.qkv.bias"))
            att = lib.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], desc, B, N, H, dh, dh ** -0.5, 0, self_full=True)
            lib.gemm(att, self._w(p + "self_attn.projection.weight"), self._p(p + "self_attn.projection.bias"),
                     residual=x, out_f32=x)
            h2 = self._ln(x, p + "layer_norm2.weight", p + "layer_norm2.bias", eps)
            if rec is None:
                f = lib.gemm(h2, self._w(p + "mlp.fc1.weight"), self._p(p + "mlp.fc1.bias"), act=lib.ACT_GELU)
            else:
                f_pre = lib.gemm(h2, self._w(p + "mlp.fc1.weight"), self._p(p + "mlp.fc1.bias"), want="f32")
                f = lib.gelu(f_pre, want=self.want)
                rec.update(h1=h, qkv=qkv, att=att, x_mid=x.clone(), h2=h2, f_pre=f_pre, f=f)
            lib.gemm(f, self._w(p + "mlp.fc2.weight"), self._p(p + "mlp.fc2.bias"), residual=x, out_f32=x)
        if save is not None:
            save["x_last"] = x.clone()
        return x

    def vit_post(self, x):
        return self._ln(x, "vision_model.post_layernorm.weight", "vision_model.post_layernorm.bias", self.v["layer_norm_eps"])

    def vit_train_params(self):
        """{name: fp32 storage} of the ViT parameters vit_backward / vit_embed_backward reach: every encoder-layer parameter, the post-LayerNorm,
        the patch convolution, the class and position embeddings"""
        return {"vision_model." + n: p_.data for n, p_ in self.m.vision_model.named_parameters()}

    @torch.no_grad()
    def vit_backward(self, save, d_img, grads):
        """d_img fp32 [B*N, D]: gradient w.r.t. the post-LayerNorm ViT rows.  Accumulates into grads[name] for the encoder layers [save.first, end)
        and the post-LayerNorm (FT_VL with a vision-tower selection); fp32 gradient rows, GEMM operands in the compute dtype."""
        v = self.v
        D = v["hidden_size"]
        H = v["num_attention_heads"]
        dhd = D // H
        eps = v["layer_norm_eps"]
        B, N, desc = save["B"], save["N"], save["desc"]
        lib.layernorm_bwd_params(save["x_last"], d_img, eps, grads["vision_model.post_layernorm.weight"], grads["vision_model.post_layernorm.bias"])
        dx = lib.layernorm_bwd_dx(save["x_last"], self._p("vision_model.post_layernorm.weight"), d_img, eps)
        for i in sorted(save["layers"], reverse=True):
            p = "vision_model.encoder.layers.%d." % i
            r = save["layers"][i]
            self.acc_linear_grads(grads, p + "mlp.fc2.weight", p + "mlp.fc2.bias", r["f"], dx)
            df = lib.gemm(self._act(dx), self._wt(p + "mlp.fc2", lambda: self._w(p + "mlp.fc2.weight")), want="f32")
            dpre = lib.gelu_bwd(r["f_pre"], df)
            self.acc_linear_grads(grads, p + "mlp.fc1.weight", p + "mlp.fc1.bias", r["h2"], dpre)
            dh2 = lib.gemm(self._act(dpre), self._wt(p + "mlp.fc1", lambda: self._w(p + "mlp.fc1.weight")), want="f32")
            lib.layernorm_bwd_params(r["x_mid"], dh2, eps, grads[p + "layer_norm2.weight"], grads[p + "layer_norm2.bias"])
            dmid = lib.layernorm_bwd_dx(r["x_mid"], self._p(p + "layer_norm2.weight"), dh2, eps)
            lib.delta_op(1, dmid, None, dx)            # + the residual branch
            self.acc_linear_grads(grads, p + "self_attn.projection.weight", p + "self_attn.projection.bias", r["att"], dmid)
            datt = lib.gemm(self._act(dmid), self._wt(p + "self_attn.projection", lambda: self._w(p + "self_attn.projection.weight")))
            qkv = r["qkv"]
            gq, gk, gv = lib.attention_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], r["att"], datt, desc, B, N, H, dhd, dhd ** -0.5, 0)
            dqkv = torch.cat([gq, gk, gv], 1)
            self.acc_linear_grads(grads, p + "self_attn.qkv.weight", p + "self_attn.qkv.bias", r["h1"], dqkv)
            dh1 = lib.gemm(dqkv, self._wt(p + "self_attn.qkv", lambda: self._w(p + "self_attn.qkv.weight")), want="f32")
            lib.layernorm_bwd_params(r["x_in"], dh1, eps, grads[p + "layer_norm1.weight"], grads[p + "layer_norm1.bias"])
            dx = lib.layernorm_bwd_dx(r["x_in"], self._p(p + "layer_norm1.weight"), dh1, eps)
            lib.delta_op(1, dx, None, dmid)
        return dx

    @torch.no_grad()
    def qformer_rows(self, img, B, save=None):
        """K4-K5: ViT rows -> language_projection(Q-Former(queries, img)) fp32 [B, Q, d_llm].  save: a dict that receives what
        qformer_backward needs (FT_VL on Q-Former parameters): per block its input rows, q / k / v, attention output, the dense output in front of
        each post-LayerNorm, the FFN's fp32 pre-activations."""
        q = self.q
        N = (self.v["image_size"] // self.v["patch_size"]) ** 2 + 1
        dq = q["hidden_size"]
        Hq = q["num_attention_heads"]
        dhq = dq // Hq
        qeps = q["layer_norm_eps"]
        Qn = self.Q
        qt = self._p("query_tokens").reshape(Qn, dq)
        h32 = lib.layernorm(qt.contiguous(), self._p("qformer.layernorm.weight"), self._p("qformer.layernorm.bias"), qeps,
                            want="f32")
        h32 = h32.repeat(B, 1)  # [B*Q, dq] (plumbing: the learned queries are shared by every image)
        self_desc = self._full_desc(B, Qn, Qn, self_rows=True)
        cross_desc = self._full_desc(B, Qn, N)
        if save is not None:
            save.update(B=B, N=N, img=img, self_desc=self_desc, cross_desc=cross_desc, layers=[])
        for i in range(q["num_hidden_layers"]):
            p = "qformer.encoder.layer.%d." % i
            rec = {} if save is not None else None
            h32 = self._bert_attention(p + "attention.", h32, None, self_desc, B, Qn, Hq, dhq, qeps, rec, "self")
            if i % q["cross_attention_frequency"] == 0:
                h32 = self._bert_attention(p + "crossattention.", h32, img, cross_desc, B, Qn, Hq, dhq, qeps, rec, "cross")
            hb = self._act(h32)
            if rec is None:
                f = lib.gemm(hb, self._w(p + "intermediate_query.dense.weight"), self._p(p + "intermediate_query.dense.bias"),
                             act=lib.ACT_GELU)
            else:       # the pre-activations are kept: GEMM to fp32, GELU as a pass of its own
                f_pre = lib.gemm(hb, self._w(p + "intermediate_query.dense.weight"), self._p(p + "intermediate_query.dense.bias"), want="f32")
                f = lib.gelu(f_pre, want=self.want)
            o = lib.gemm(f, self._w(p + "output_query.dense.weight"), self._p(p + "output_query.dense.bias"), want="f32")
            if rec is not None:
                rec["ffn"] = dict(h_in=h32, hb=hb, f_pre=f_pre, f=f, o=o)
                save["layers"].append(rec)
            h32 = lib.layernorm(o, self._p(p + "output_query.LayerNorm.weight"), self._p(p + "output_query.LayerNorm.bias"),
                                qeps, add=h32, want="f32")
        hb = self._act(h32)
        if save is not None:
            save["h_last"] = hb
        out = lib.gemm(hb, self._w("language_projection.weight"), self._p("language_projection.bias"), want="f32")
        return out.view(B, Qn, -1)

    def _bert_attention(self, p, h32, kv_bf16, desc, B, Qn, H, dh, eps, rec=None, tag=None):
        hb = self._act(h32)
        src = hb if kv_bf16 is None else kv_bf16
        qq = lib.gemm(hb, self._w(p + "attention.query.weight"), self._p(p + "attention.query.bias"))
        kk = lib.gemm(src, self._w(p + "attention.key.weight"), self._p(p + "attention.key.bias"))
        vv = lib.gemm(src, self._w(p + "attention.value.weight"), self._p(p + "attention.value.bias"))
        att = lib.attention(qq, kk, vv, desc, B, Qn, H, dh, dh ** -0.5, 0)
        o = lib.gemm(att, self._w(p + "output.dense.weight"), self._p(p + "output.dense.bias"), want="f32")
        if rec is not None:
            rec[tag] = dict(h_in=h32, hb=hb, q=qq, k=kk, v=vv, att=att, o=o)
        return lib.layernorm(o, self._p(p + "output.LayerNorm.weight"), self._p(p + "output.LayerNorm.bias"), eps,
                             add=h32, want="f32")

    # ---- FT_VL on Q-Former parameters (the substring rule with "qformer": R/editor/vllm_editors/ft_vl/ft_vl.py:31-36, R/configs/ft_vl/blip2-opt-2.7b.yaml:9) ----
    def qformer_train_params(self):
        """{name: fp32 storage} of every Q-Former parameter, the learned queries and the language projection (what autograd reaches between the
        ViT rows and the decoder's input rows)"""
        out = {"qformer." + n: p_.data for n, p_ in self.m.qformer.named_parameters()}
        for n in ("query_tokens", "language_projection.weight", "language_projection.bias"):
            out[n] = self.m.get(n).data
        return out

    @torch.no_grad()
    def qformer_backward(self, save, d_out, grads, want_d_img=False):
        """d_out fp32 [B*Q, d_llm]: gradient w.r.t. the projected query rows (qformer_rows' output).  Accumulates into grads[name] (fp32, every
        Q-Former parameter: attention / cross-attention projections, FFN, every LayerNorm incl. the one on the learned queries; grads None:
        the Q-Former is frozen, only gradient rows pass through); the language projection and the learned queries are always frozen.
        want_d_img: also return the gradient w.r.t. the ViT rows [B*N, D_vit] (through the cross-attention keys and values) for a
        vision-tower selection.  Gradient rows are kept in fp32, GEMM operands rounded to the compute dtype -- the convention of
        decoder_backward."""
        q = self.q
        B, N, img = save["B"], save["N"], save["img"]
        dq_ = q["hidden_size"]
        Hq = q["num_attention_heads"]
        dhq = dq_ // Hq
        eps = q["layer_norm_eps"]
        Qn = self.Q
        scale = dhq ** -0.5
        if grads is not None:
            self.acc_linear_grads(grads, "language_projection.weight", "language_projection.bias", save["h_last"], d_out)
        dh = lib.gemm(self._act(d_out), self._wt("language_projection", lambda: self._w("language_projection.weight")), want="f32")
        # the backward kernels read a sequence's keys from the OWN-key fields of the descriptor (the forward's cross form names them as a visible prefix)
        cross_desc = lib.h2d([[b * Qn, Qn, 0, 0, b * N, N] for b in range(B)], torch.int32, self.dev)

        d_img = None

        def acc(wname, bname, x_rows, d_rows):
            if grads is not None:
                self.acc_linear_grads(grads, wname, bname, x_rows, d_rows)

        def post_ln_bwd(pfx, o, h_in, dy):
            """backward of LayerNorm(o + h_in): parameter gradients, -> gradient w.r.t. the sum (= w.r.t. o and w.r.t. h_in)"""
            if grads is not None:
                lib.layernorm_bwd_params(o, dy, eps, grads[pfx + "LayerNorm.weight"], grads[pfx + "LayerNorm.bias"], add=h_in)
            return lib.layernorm_bwd_dx(o, self._p(pfx + "LayerNorm.weight"), dy, eps, add=h_in)

        def attention_bwd(p, r, dy, desc, kv_rows, max_len):
            """p: '...attention.' / '...crossattention.'; dy: gradient w.r.t. the block's output -> gradient w.r.t. its input rows"""
            nonlocal d_img
            dsum = post_ln_bwd(p + "output.", r["o"], r["h_in"], dy)
            acc(p + "output.dense.weight", p + "output.dense.bias", r["att"], dsum)
            datt = lib.gemm(self._act(dsum), self._wt(p + "output.dense", lambda: self._w(p + "output.dense.weight")))
            gq, gk, gv = lib.attention_bwd(r["q"], r["k"], r["v"], r["att"], datt, desc, B, max_len, Hq, dhq, scale, 0)
            acc(p + "attention.query.weight", p + "attention.query.bias", r["hb"], gq)
            acc(p + "attention.key.weight", p + "attention.key.bias", kv_rows, gk)
            acc(p + "attention.value.weight", p + "attention.value.bias", kv_rows, gv)
            d_in = lib.gemm(gq, self._wt(p + "attention.query", lambda: self._w(p + "attention.query.weight")), residual=dsum, want="f32")
            wk = self._wt(p + "attention.key", lambda: self._w(p + "attention.key.weight"))
            wv = self._wt(p + "attention.value", lambda: self._w(p + "attention.value.weight"))
            if kv_rows is r["hb"]:      # self-attention: keys and values come from the same rows
                d_in = lib.gemm(gk, wk, residual=d_in, want="f32")
                d_in = lib.gemm(gv, wv, residual=d_in, want="f32")
            elif want_d_img:            # cross-attention: keys and values are projections of the ViT rows
                d_img = lib.gemm(gk, wk, residual=d_img, want="f32")
                d_img = lib.gemm(gv, wv, residual=d_img, want="f32")
            return d_in
        for i in range(q["num_hidden_layers"] - 1, -1, -1):
            p = "qformer.encoder.layer.%d." % i
            rec = save["layers"][i]
            f = rec["ffn"]
            dsum = post_ln_bwd(p + "output_query.", f["o"], f["h_in"], dh)
            acc(p + "output_query.dense.weight", p + "output_query.dense.bias", f["f"], dsum)
            df = lib.gemm(self._act(dsum), self._wt(p + "output_query.dense", lambda: self._w(p + "output_query.dense.weight")), want="f32")
            dpre = lib.gelu_bwd(f["f_pre"], df)
            acc(p + "intermediate_query.dense.weight", p + "intermediate_query.dense.bias", f["hb"], dpre)
            dh = lib.gemm(self._act(dpre), self._wt(p + "intermediate_query.dense", lambda: self._w(p + "intermediate_query.dense.weight")),
                          residual=dsum, want="f32")
            if "cross" in rec:
                dh = attention_bwd(p + "crossattention.", rec["cross"], dh, cross_desc, img, max(Qn, N))
            dh = attention_bwd(p + "attention.", rec["self"], dh, save["self_desc"], rec["self"]["hb"], Qn)
        # LayerNorm on the learned queries: the same [Q, d] rows for every image of the batch
        if grads is not None:
            d0 = dh.view(B, Qn, dq_).sum(0).contiguous() if B > 1 else dh
            qt = self._p("query_tokens").reshape(Qn, dq_).contiguous()
            lib.layernorm_bwd_params(qt, d0, eps, grads["qformer.layernorm.weight"], grads["qformer.layernorm.bias"])
            grads["query_tokens"].add_(lib.layernorm_bwd_dx(qt, self._p("qformer.layernorm.weight"), d0, eps).view_as(grads["query_tokens"]))
        return d_img

    # ------------------------------------------------------------------------------------------
    # K6: decoder input rows
    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def pack_from_embeds(self, inputs_embeds, attention_mask, lens=None):
        """[B,T,d] embeddings (+ right-padding mask) -> packed rows with OPT positions added.
        Padded rows are kept (so logits come back as [B,T,V]) but never attended to.  `lens` (host list of sequence lengths)
        spares the device -> host read of the mask sums when the caller knows them."""
        B, T, d = inputs_embeds.shape
        am = attention_mask.to(torch.int64)
        if lens is None:
            lens = am.sum(1).tolist()
        pos = (torch.cumsum(am, 1) * am - 1).to(torch.int32).reshape(-1).contiguous()
        rows = inputs_embeds.reshape(B * T, d).to(torch.float32).contiguous()
        src = torch.arange(B * T, dtype=torch.int32, device=self.dev)
        tok = torch.zeros(B * T, dtype=torch.int32, device=self.dev)
        x = lib.embed_rows(tok, src, pos.to(self.dev), self._p("language_model.model.decoder.embed_tokens.weight"), rows,
                           self._p("language_model.model.decoder.embed_positions.weight"))
        desc = lib.h2d([[b * T, int(lens[b]), 0, 0, b * T, int(lens[b])] for b in range(B)], torch.int32, self.dev)
        return PackedSeqs(x, [b * T for b in range(B)], [int(n) for n in lens], desc, T)

    @torch.no_grad()
    def pack_rows(self, rows, pos, desc, max_len):
        """Already-embedded rows [R,d] with explicit positions and attention descriptors (dense: every row belongs to a sequence):
        the evaluator's shared-prefix packing of generic probes.  OPT's learned positions are added here."""
        R = rows.shape[0]
        src = torch.arange(R, dtype=torch.int32, device=self.dev)
        tok = torch.zeros(R, dtype=torch.int32, device=self.dev)
        x = lib.embed_rows(tok, src, lib.h2d(pos, torch.int32, self.dev), self._p("language_model.model.decoder.embed_tokens.weight"),
                           rows.to(torch.float32).contiguous(), self._p("language_model.model.decoder.embed_positions.weight"))
        return PackedSeqs(x, [d_[0] for d_ in desc], [d_[1] for d_ in desc], lib.h2d(desc, torch.int32, self.dev), max_len, True)

    @torch.no_grad()
    def pack_from_tokens(self, seqs, img_tokens, share_prefix=False):
        """seqs: list of (image_index or None, token_id_list).  Each sequence becomes
        [Q image-token rows (if any)] + token embeddings, positions 0..len-1 (blip2.py:45-52)."""
        if share_prefix:
            return self._pack_shared_prefix(seqs, img_tokens)
        tok, src, pos, start, length = [], [], [], [], []
        Qn = self.Q
        r = 0
        for (img, ids) in seqs:
            start.append(r)
            n = 0
            if img is not None:
                tok += [0] * Qn
                src += list(range(img * Qn, (img + 1) * Qn))
                pos += list(range(Qn))
                n = Qn
            tok += list(ids)
            src += [-1] * len(ids)
            pos += list(range(n, n + len(ids)))
            n += len(ids)
            length.append(n)
            r += n
        dev = self.dev
        t_tok = lib.h2d(tok, torch.int32, dev)
        t_src = lib.h2d(src, torch.int32, dev)
        t_pos = lib.h2d(pos, torch.int32, dev)
        rows = None if img_tokens is None else img_tokens.reshape(-1, img_tokens.shape[-1]).contiguous()
        x = lib.embed_rows(t_tok, t_src, t_pos, self._p("language_model.model.decoder.embed_tokens.weight"), rows,
                           self._p("language_model.model.decoder.embed_positions.weight"))
        desc = lib.h2d([[s, n, 0, 0, s, n] for s, n in zip(start, length)], torch.int32, dev)
        return PackedSeqs(x, start, length, desc, max(length), True)

    def _pack_shared_prefix(self, seqs, img_tokens):
        """Same sequences, but the Q image-token rows of every distinct image are packed ONCE: a prefix
        sequence (causal among itself) that all texts on that image attend to through the descriptor's
        visible-prefix range.  Image tokens come first and attention is causal, so their hidden states and
        K/V do not depend on the text: results equal the unshared packing (SURVEY 8(d) A_min).
        start/length of the returned PackedSeqs describe the TEXT part of each input sequence."""
        Qn = self.Q
        tok, src, pos, desc = [], [], [], []
        prefix_start = {}
        r = 0
        for img in sorted({s[0] for s in seqs if s[0] is not None}):
            prefix_start[img] = r
            tok += [0] * Qn
            src += list(range(img * Qn, (img + 1) * Qn))
            pos += list(range(Qn))
            desc.append([r, Qn, 0, 0, r, Qn])
            r += Qn
        start, length = [], []
        for (img, ids) in seqs:
            n = len(ids)
            off = 0 if img is None else Qn
            start.append(r)
            length.append(n)
            tok += list(ids)
            src += [-1] * n
            pos += list(range(off, off + n))
            desc.append([r, n, prefix_start[img], Qn, r, n] if img is not None else [r, n, 0, 0, r, n])
            r += n
        dev = self.dev
        rows = None if img_tokens is None else img_tokens.reshape(-1, img_tokens.shape[-1]).contiguous()
        x = lib.embed_rows(lib.h2d(tok, torch.int32, dev), lib.h2d(src, torch.int32, dev),
                           lib.h2d(pos, torch.int32, dev),
                           self._p("language_model.model.decoder.embed_tokens.weight"), rows,
                           self._p("language_model.model.decoder.embed_positions.weight"))
        ps = PackedSeqs(x, start, length, lib.h2d(desc, torch.int32, dev), max(max(length), Qn), True)
        ps.n_seq = len(desc)
        return ps

    # ------------------------------------------------------------------------------------------
    # K7: decoder layers.  Returns the residual stream after `n_layers` full layers; when
    # stop_before_fc2, the last processed layer stops at the fc2 input: returns (x_mid, a_bf16).
    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def decoder_layers(self, ps: PackedSeqs, upto_layer=None, stop_before_fc2=False, save=None, return_h=False, first_layer=0):
        """save: None, or {"layers": set of layer ids}; filled with save[i] = activations the backward of layer i needs
        (decoder_backward).  Low-rank module deltas registered in self.module_deltas (MEND_VL's forward_edit_hook,
        mend_vl.py:73-80) are applied to the fc1 / fc2 outputs."""
        t = self.t
        d, H = t["hidden_size"], t["num_attention_heads"]
        dh = d // H
        x = ps.x
        n_seq = ps.desc.shape[0]
        last = t["num_hidden_layers"] - 1 if upto_layer is None else upto_layer
        ctx = self.path_ctx() if (save is None and not return_h and first_layer == 0 and not self._hooks_active()) else None
        if ctx is not None:
            a = ctx.llm_layers(x, ps.desc, n_seq, ps.max_len, ps.dense, last + 1, stop_before_fc2)
            return x, a
        deltas = getattr(self, "module_deltas", None) or {}
        # the layers BELOW everything this call has to look into (saved activations, low-rank deltas, extra neurons) are plain frozen
        # layers: one path-level call instead of ~10 Python-ordered launches per layer (MEND_VL edits layers 29-31 of 32: its forward
        # was 320 launches from Python, 2.7 ms of host time per edit on a host-bound path)
        if first_layer == 0:
            touched = [last if (stop_before_fc2 or return_h) else last + 1]     # a stopping last layer is handled below
            if save is not None:
                touched.append(min(save["layers"]) if save["layers"] else last + 1)
            touched += [int(re.search(r"layers\.(\d+)\.", n).group(1)) for n in deltas]
            touched += list((getattr(self, "extra_neurons", None) or {}).keys())
            lo = min(touched)
            pctx = self.path_ctx() if lo > 0 else None
            if pctx is not None:
                pctx.llm_layers(x, ps.desc, n_seq, ps.max_len, ps.dense, lo, False)
                first_layer = lo
        for i in range(first_layer, last + 1):
            p = "language_model.model.decoder.layers.%d." % i
            rec = None
            if save is not None and i in save["layers"]:
                rec = save[i] = {"x_in": x.clone()}
            h = self._ln(x, p + "self_attn_layer_norm.weight", p + "self_attn_layer_norm.bias", LN_EPS_OPT)
            qkv = lib.gemm(h, self.m.fused_qkv_w[str(i)], self.m.fused_qkv_b[str(i)])
            att = lib.attention(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], ps.desc, n_seq, ps.max_len, H, dh, dh ** -0.5,
                                1, out=(torch.empty if ps.dense else torch.zeros)((x.shape[0], d), dtype=self.adt, device=self.dev))
            lib.gemm(att, self._w(p + "self_attn.out_proj.weight"), self._p(p + "self_attn.out_proj.bias"), residual=x,
                     out_f32=x)
            h = self._ln(x, p + "final_layer_norm.weight", p + "final_layer_norm.bias", LN_EPS_OPT)
            if rec is not None:
                rec.update(qkv=qkv, att=att, x_mid=x.clone(), h2=h)
            d1 = deltas.get(p + "fc1")
            if d1 is None:
                a = lib.gemm(h, self._w(p + "fc1.weight"), self._p(p + "fc1.bias"), act=lib.ACT_RELU)
            else:   # relu(h W1^T + b + (h xt^T) dt): one GEMM over the concatenated K = [d | n_pad]
                coef = lib.gemm(h, d1["xt"])
                a = lib.gemm(torch.cat([h, coef], 1), d1["w_cat"], self._p(p + "fc1.bias"), act=lib.ACT_RELU)
            if rec is not None:
                rec["a"] = a
            if stop_before_fc2 and i == last:
                return (x, a, h) if return_h else (x, a)
            lib.gemm(a, self._w(p + "fc2.weight"), self._p(p + "fc2.bias"), residual=x, out_f32=x)
            d2 = deltas.get(p + "fc2")
            if d2 is not None:
                lib.mend_apply_(a, d2["xt"], d2["dtT"], x)       # K17: x += (a . xt^T) . dt
            self.add_extra_neurons(i, h, x)
        return x, None

    def add_extra_neurons(self, layer, h, x):
        """x += relu(h K^T + B) V for the patch neurons appended to `layer`'s FFN (TP_VL, tp_vl.py:74-103): the hooks
        concatenate the extra pre-activations with fc1's output, the layer's ReLU acts on all of them and fc2's hooks add
        extra_act @ extra_values.  self.extra_neurons[layer] = {"K": [n_pad, d] , "B": fp32 [n_pad], "VT": [d, n_pad]}
        (operand dtype, zero-padded neurons contribute relu(0) * 0)."""
        ent = (getattr(self, "extra_neurons", None) or {}).get(layer)
        if ent is not None:
            lib.gemm(lib.gemm(h, ent["K"], ent["B"], act=lib.ACT_RELU), ent["VT"], residual=x, out_f32=x)

    MEND_MODULE_RE = r"^(.*\.layers\.)(\d+)\.(fc1|fc2)$"

    def set_module_deltas(self, deltas):
        """deltas: {module name: {"xt" [n_pad, d_in], "xtT", "dt" [n_pad, d_out] (scaled), "dtT"}} in the operand dtype;
        fc1 entries get the concatenated-K weight [W1 | dt^T] the forward uses."""
        for name, e in deltas.items():
            if name.endswith("fc1"):
                e["w_cat"] = torch.cat([self._w(name + ".weight"), e["dtT"]], 1).contiguous()
        self.module_deltas = deltas

    def _wt(self, key, getter):
        """Transposed GEMM operand W^T [in, out] (contiguous) of a frozen weight, cached: dX = dY . W runs on the TN
        kernels as gemm(dY, W^T)."""
        c = self.__dict__.setdefault("_wt_cache", {})
        if key not in c:
            c[key] = getter().t().contiguous()
        return c[key]

    @staticmethod
    def _padk(t):
        """[r, k] -> contiguous, zero-padded to k % 4 == 0 (K of the exact-fp32 GEMM)"""
        pad = (-t.shape[1]) % 4
        return t.contiguous() if pad == 0 else torch.cat([t, torch.zeros((t.shape[0], pad), dtype=t.dtype, device=t.device)], 1).contiguous()

    @torch.no_grad()
    def acc_linear_grads(self, grads, wname, bname, x_rows, d_rows):
        """grads[wname] += d_rows^T x_rows, grads[bname] += column sums of d_rows (fp32; the autograd of F.linear over the packed
        rows): one TN GEMM on the transposed operands (K = number of rows) and one deterministic column reduction."""
        x32 = x_rows.to(torch.float32)
        d32 = d_rows.to(torch.float32).contiguous()
        g = grads[wname]
        lib.gemm(self._padk(d32.t()), self._padk(x32.t()), residual=g, out_f32=g)
        if bname is not None:
            lib.colsum_(d32, grads[bname], True)

    # ---- full fine-tuning of the language model (LTE_VL training): what an editor needs beyond decoder_backward(grads=...) ----
    LM_MODULE = "language_model"

    def train_params(self):
        """{name: fp32 storage} of every trainable parameter of the language model; the q / k / v projections appear as the fused
        operand they are row blocks of ("derived.dec_qkv.<i>.weight" / ".bias": one optimizer state per storage)."""
        m, out = self.m, {}
        for n, p_ in getattr(m, self.LM_MODULE).named_parameters():
            name = self.LM_MODULE + "." + n
            if m._fused_slot(name) is None:
                out[name] = p_.data
        for layer, w in m.fused_qkv_w.items():
            out["derived.dec_qkv.%s.weight" % layer] = w
            out["derived.dec_qkv.%s.bias" % layer] = m.fused_qkv_b[layer]
        return out

    @torch.no_grad()
    def head_fwd(self, pre_ln):
        """rows before the final LayerNorm -> (normalised rows, fp32 logits)"""
        hn = self._ln(pre_ln, "language_model.model.decoder.final_layer_norm.weight", "language_model.model.decoder.final_layer_norm.bias",
                      LN_EPS_OPT)
        return hn, lib.gemm(hn, self._w("language_model.model.decoder.embed_tokens.weight"), want="f32")     # (_w: the compute-dtype operand of an fp32 master)

    @torch.no_grad()
    def head_bwd(self, pre_ln, hn, dlog, grads):
        """dlogits [R, V] fp32 -> gradient rows w.r.t. pre_ln; accumulates the tied embedding (the lm_head's weight gradient
        dlogits^T . LN(h): training inputs arrive as embeddings, so the input side contributes nothing) and the final LayerNorm."""
        self.acc_linear_grads(grads, "language_model.model.decoder.embed_tokens.weight", None, hn, dlog)
        dH = lib.gemm(dlog, self.m.embed_T, want="f32")
        lib.layernorm_bwd_params(pre_ln, dH, LN_EPS_OPT, grads["language_model.model.decoder.final_layer_norm.weight"],
                                 grads["language_model.model.decoder.final_layer_norm.bias"])
        return self.final_norm_bwd(pre_ln, dH)

    @torch.no_grad()
    def embed_bwd(self, mask, dx0, grads):
        """gradient w.r.t. the decoder input rows -> learned positions (OPT: position of a row + 2) as one-hot^T . dx on the GEMM"""
        name = "language_model.model.decoder.embed_positions.weight"
        n_pos = grads[name].shape[0]
        msk = mask.to(torch.int64)
        pos = ((torch.cumsum(msk, 1) * msk - 1).reshape(-1) + 2).clamp_(0, n_pos - 1)
        onehot = torch.zeros((dx0.shape[0], n_pos), dtype=torch.float32, device=dx0.device)
        onehot[torch.arange(dx0.shape[0], device=dx0.device), pos] = msk.reshape(-1).to(torch.float32)
        self.acc_linear_grads(grads, name, None, dx0, onehot)

    def after_param_update(self):
        self.__dict__.pop("_wt_cache", None)        # transposed operands of the backward
        self.m.refresh_derived(force=True)          # embed_T

    @torch.no_grad()
    def decoder_backward(self, ps: PackedSeqs, save, dx, capture, grads=None):
        """Backward through the saved decoder layers (highest first).  dx: fp32 [R, d] gradient w.r.t. the decoder
        output (input of the final LayerNorm).  Returns {module name: (input rows, output-gradient rows)} for the
        fc1 / fc2 modules named in `capture` -- what MEND_VL's forward/backward hooks record (mend_vl.py:62-71) --
        and the gradient w.r.t. the input of the lowest saved layer.
        grads (full fine-tuning, LTE_VL training): {parameter name: fp32 accumulator}; every decoder-layer parameter found in it
        receives its gradient -- fc1 / fc2 / out_proj weights and biases, the fused q|k|v operand under
        "derived.dec_qkv.<i>.weight" / ".bias" (the HF q/k/v parameters are row blocks of it), both LayerNorms."""
        t = self.t
        d, H = t["hidden_size"], t["num_attention_heads"]
        dh = d // H
        n_seq = ps.desc.shape[0]
        deltas = getattr(self, "module_deltas", None) or {}
        out = {}
        for i in sorted(save["layers"], reverse=True):
            p = "language_model.model.decoder.layers.%d." % i
            rec = save[i]
            # gradient activations are STORED in fp32 (they are small); only GEMM operands are rounded to the compute dtype
            dz = self._act(dx)
            if p + "fc2" in capture:
                out[p + "fc2"] = (rec["a"], dx.clone())
            if grads is not None:
                self.acc_linear_grads(grads, p + "fc2.weight", p + "fc2.bias", rec["a"], dx)
            da = lib.gemm(dz, self._wt(p + "fc2", lambda: self._w(p + "fc2.weight")), want="f32")
            d2 = deltas.get(p + "fc2")
            if d2 is not None:   # the delta branch a @ dW also carries gradient to a
                lib.gemm(lib.gemm(dz, d2["dt"]), d2["xtT"], residual=da, out_f32=da)
            dpre = lib.relu_bwd(rec["a"].to(torch.float32), da)
            if p + "fc1" in capture:
                out[p + "fc1"] = (rec["h2"], dpre)
            dpre_op = self._act(dpre)
            if grads is not None:
                self.acc_linear_grads(grads, p + "fc1.weight", p + "fc1.bias", rec["h2"], dpre)
            dh2 = lib.gemm(dpre_op, self._wt(p + "fc1", lambda: self._w(p + "fc1.weight")), want="f32")
            d1 = deltas.get(p + "fc1")
            if d1 is not None:
                lib.gemm(lib.gemm(dpre_op, d1["dt"]), d1["xtT"], residual=dh2, out_f32=dh2)
            dy = lib.layernorm_bwd_dx(rec["x_mid"], self._p(p + "final_layer_norm.weight"), dh2, LN_EPS_OPT)
            if grads is not None:
                lib.layernorm_bwd_params(rec["x_mid"], dh2, LN_EPS_OPT, grads[p + "final_layer_norm.weight"], grads[p + "final_layer_norm.bias"])
            lib.delta_op(1, dy, None, dx)      # dy += dx (residual branch)
            if grads is not None:
                self.acc_linear_grads(grads, p + "self_attn.out_proj.weight", p + "self_attn.out_proj.bias", rec["att"], dy)
            datt = lib.gemm(self._act(dy), self._wt(p + "out_proj", lambda: self._w(p + "self_attn.out_proj.weight")))
            qkv = rec["qkv"]
            dq, dk, dv = lib.attention_bwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], rec["att"], datt, ps.desc, n_seq,
                                           ps.max_len, H, dh, dh ** -0.5, 1)
            dqkv = torch.cat([dq, dk, dv], 1)
            if grads is not None:
                h1 = self._ln(rec["x_in"], p + "self_attn_layer_norm.weight", p + "self_attn_layer_norm.bias", LN_EPS_OPT)
                self.acc_linear_grads(grads, "derived.dec_qkv.%d.weight" % i, "derived.dec_qkv.%d.bias" % i, h1, dqkv)
            dh1 = lib.gemm(dqkv, self._wt(p + "qkv", lambda: self.m.fused_qkv_w[str(i)]), want="f32")
            if grads is not None:
                lib.layernorm_bwd_params(rec["x_in"], dh1, LN_EPS_OPT, grads[p + "self_attn_layer_norm.weight"],
                                         grads[p + "self_attn_layer_norm.bias"])
            dx = lib.layernorm_bwd_dx(rec["x_in"], self._p(p + "self_attn_layer_norm.weight"), dh1, LN_EPS_OPT)
            lib.delta_op(1, dx, None, dy)
        return out, dx

    # K8: final LN + tied lm_head on the given rows -> fp32 logits
    @torch.no_grad()
    def lm_head(self, x_rows, add=None):
        ctx = self.path_ctx()
        if ctx is not None:
            return ctx.llm_head(x_rows, add)
        h = self._ln(x_rows, "language_model.model.decoder.final_layer_norm.weight",
                     "language_model.model.decoder.final_layer_norm.bias", LN_EPS_OPT, add=add)
        return lib.gemm(h, self._p("language_model.model.decoder.embed_tokens.weight"), want="f32")

    # ---- what an FT_VL-style editor needs to know about the edited layer (shared with LlavaEngine) ----
    def edit_target(self):
        return "language_model.model.decoder.layers.%d.fc2.weight" % self.edit_layer

    def edit_bias(self):
        return self._p("language_model.model.decoder.layers.%d.fc2.bias" % self.edit_layer)

    def final_norm_bwd(self, x_rows, dH, add=None):
        return lib.layernorm_bwd_dx(x_rows, self._p("language_model.model.decoder.final_layer_norm.weight"), dH, LN_EPS_OPT,
                                    add=add)

    def embed_table(self):
        return self._p("language_model.model.decoder.embed_tokens.weight")

    @torch.no_grad()
    def full_logits(self, ps: PackedSeqs):
        x, _ = self.decoder_layers(ps)
        return self.lm_head(x)
