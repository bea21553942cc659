"""LlavaEngine: LLaVA-1.5 (CLIP ViT + projector + LLaMA/Vicuna decoder) forward and the FT_VL tail over the
C-ABI HIP kernels -- same interface as Blip2Engine (engine.py), so FTvl and BatchedEditEval run unchanged.

Reference call sites replaced: R/editor/vllms_for_edit/llava/llava.py:25-51 (vision tower hidden state -2 without
CLS -> multi_modal_projector -> splice at the `<image>` token) and :63-68 (language model forward).
"""
import torch

from . import lib
from .engine import Blip2Engine, PackedSeqs, plan_image_chunks


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

    # ---- path-level context (include/devqa.h "PATH LEVEL", DEVQA_FAMILY_LLAVA / _MINIGPT4): the schedules below live behind the C ABI ----
    FAMILY = lib.FAMILY_LLAVA

    def _model_desc(self):
        v, t = self.v, self.t
        return lib.ModelDesc(family=self.FAMILY, compute_dtype=lib.DTYPE_BF16 if self.adt == torch.bfloat16 else lib.DTYPE_F32,
                             image_size=v["image_size"], patch_size=v["patch_size"], v_hidden=v["hidden_size"], v_layers=v["num_hidden_layers"],
                             v_heads=v["num_attention_heads"], v_ffn=v["intermediate_size"], num_query_tokens=self.n_img,
                             t_hidden=t["hidden_size"], t_layers=t["num_hidden_layers"], t_heads=t["num_attention_heads"],
                             t_ffn=t["intermediate_size"], t_vocab=self._p(self.lm + "lm_head.weight").shape[0],
                             t_max_pos=t.get("max_position_embeddings", 0), v_ln_eps=v["layer_norm_eps"], t_rms_eps=self.eps,
                             t_rope_theta=self.theta, v_run_layers=v["num_hidden_layers"] - 1,
                             t_flags=lib.DESC_FUSE_SWIGLU if self._fuse_swiglu() else 0)

    def path_ctx(self):
        """lib.PathContext over this model's weight table (rebuilt when any of its buffers has moved: storage fingerprint), or None
        when the schedule must stay in Python: editor hooks registered on the engine are handled by decoder_layers itself, and
        DEVQA_PATH_ABI=0 keeps the Python-ordered launches (A/B tests of the two drivers of the same kernels)."""
        import os
        if not hasattr(self.m, "weight_table") or os.environ.get("DEVQA_PATH_ABI", "1") == "0":
            return None
        self.m.refresh_shadows()        # before the derived operands: the interleaved [gate | up] copy is made from the refreshed row blocks
        self.m.refresh_derived()
        fp = (self.m.storage_fingerprint(), self._fuse_swiglu())
        if self.__dict__.get("_ctx") is None or self._ctx_fp != fp:
            old = self.__dict__.get("_ctx")
            if old is not None:
                torch.cuda.current_stream(self.dev).synchronize()
                old.close()
            self._ctx = lib.PathContext(self.dev.index or 0, self._model_desc(), self.m.weight_table())
            self._ctx_fp = fp
        self.m.refresh_shadows()
        return self._ctx

    def _fuse_swiglu(self):
        """SwiGLU inside the [gate | up] GEMM's epilogue (lib.gemm_swiglu; bf16 mode, layers whose gate / up outputs nobody needs).  An editor that
        hooks gate_proj / up_proj (MEND_VL on the LLaMA FFN) turns it off for the whole model (`fuse_swiglu = False`): its pre-edit probes must
        round like its post-edit ones, which run the two-pass form with the deltas.  DEVQA_SWIGLU_FUSED=0: off."""
        import os
        return bool(self.__dict__.get("fuse_swiglu", True)) and self.adt == torch.bfloat16 and os.environ.get("DEVQA_SWIGLU_FUSED", "1") != "0"

    def _hooks_active(self):
        return bool(getattr(self, "module_deltas", None)) or bool(getattr(self, "extra_neurons", None))

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
            torch.cuda.current_stream(self.dev).synchronize()   # cached across streams (prefetch thread): publish it complete
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
        ctx = self.path_ctx()
        if ctx is not None:
            return ctx.vision_encode(pixels.contiguous())
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
    def pack_from_embeds(self, inputs_embeds, attention_mask, lens=None):
        B, T, d = inputs_embeds.shape
        if lens is None:
            lens = attention_mask.to(torch.int64).sum(1).tolist()
        x = inputs_embeds.reshape(B * T, d).to(torch.float32).contiguous().clone()
        pos = torch.arange(T, dtype=torch.int32, device=self.dev).repeat(B).contiguous()   # LLaMA: arange positions
        desc = lib.h2d([[b * T, int(lens[b]), 0, 0, b * T, int(lens[b])] for b in range(B)], torch.int32, self.dev)
        ps = PackedSeqs(x, [b * T for b in range(B)], [int(n) for n in lens], desc, T)
        ps.pos = pos
        return ps

    @torch.no_grad()
    def pack_rows(self, rows, pos, desc, max_len):
        """See Blip2Engine.pack_rows; LLaMA: the positions feed RoPE."""
        ps = PackedSeqs(rows.to(torch.float32).contiguous().clone(), [d_[0] for d_ in desc], [d_[1] for d_ in desc],
                        lib.h2d(desc, torch.int32, self.dev), max_len, True)
        ps.pos = lib.h2d(pos, torch.int32, self.dev)
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
        t_pos = lib.h2d(pos, torch.int32, dev)
        rows = None if img_tokens is None else img_tokens.reshape(-1, img_tokens.shape[-1]).contiguous()
        emb = self.embed_table()
        x = lib.embed_rows(lib.h2d(tok, torch.int32, dev), lib.h2d(src, torch.int32, dev),
                           torch.full((len(tok),), -2, dtype=torch.int32, device=dev), emb, rows, self._zero_pos(emb))
        ps = PackedSeqs(x, start, length, lib.h2d(desc, torch.int32, dev),
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
    def decoder_layers(self, ps, upto_layer=None, stop_before_fc2=False, save=None, return_h=False, first_layer=0):
        """save: None, or {"layers": set of layer ids}: filled with the activations decoder_backward needs.  Low-rank module
        deltas (set_module_deltas; MEND_VL's forward_edit_hook, mend_vl.py:73-80) are applied to gate / up / down outputs."""
        t, m = self.t, self.m
        d, H = t["hidden_size"], t["num_attention_heads"]
        dh = d // H
        x = ps.x
        n_seq = ps.desc.shape[0]
        last = t["num_hidden_layers"] - 1 if upto_layer is None else upto_layer
        deltas = getattr(self, "module_deltas", None) or {}
        ctx = self.path_ctx() if (save is None and not return_h and not self._hooks_active() and last >= first_layer) else None
        if ctx is not None:     # plain frozen layers: one path-level call (devqa_llm_layers_ex) instead of ~10 Python-ordered launches per layer
            a = ctx.llm_layers(x, ps.desc, n_seq, ps.max_len, getattr(ps, "dense", False), last + 1 - first_layer, stop_before_fc2,
                               positions=ps.pos, first_layer=first_layer)
            return x, a
        for i in range(first_layer, last + 1):
            p = self.lm + "model.layers.%d." % i
            rec = None
            if save is not None and i in save["layers"]:
                rec = save[i] = {"x_in": x.clone()}
            h = lib.rmsnorm(x, self._p(p + "input_layernorm.weight"), self.eps, want=self.want)
            qkv = lib.gemm(h, m.fused_w["llama_qkv.%d" % i])
            lib.rope_(qkv[:, :2 * d], ps.pos, 2 * H, dh, self.theta)          # q heads then k heads
            att = lib.attention(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], ps.desc, n_seq, ps.max_len, H, dh, dh ** -0.5, 1,
                                out=torch.zeros((x.shape[0], d), dtype=self.adt, device=self.dev))
            lib.gemm(att, self._w(p + "self_attn.o_proj.weight"), residual=x, out_f32=x)
            h = lib.rmsnorm(x, self._p(p + "post_attention_layernorm.weight"), self.eps, want=self.want)
            dg, du = deltas.get(p + "mlp.gate_proj"), deltas.get(p + "mlp.up_proj")
            gu = None
            w_il = m.gu_interleaved().get(i) if (rec is None and dg is None and du is None and self._fuse_swiglu()) else None
            if w_il is not None and lib.gemm_swiglu_supported(h.shape[0], w_il.shape[0], w_il.shape[1]):
                a = lib.gemm_swiglu(h, w_il)        # silu(gate) * up in the GEMM's epilogue: no [R, 2F] intermediate (same call as the path context's)
            else:
                if dg is None and du is None:
                    gu = lib.gemm(h, m.fused_w["llama_gu.%d" % i])
                else:   # [h | h xt_g^T | h xt_u^T] . [W_gu | dt_g (gate rows) | dt_u (up rows)]^T: one GEMM over the concatenated K
                    parts = [h] + [lib.gemm(h, e["xt"]) for e in (dg, du) if e is not None]
                    gu = lib.gemm(torch.cat(parts, 1), self._gu_cat[i])
                a = lib.swiglu(gu)
            if rec is not None:
                rec.update(qkv=qkv, att=att, x_mid=x.clone(), h2=h, gu=gu, a=a)
            if stop_before_fc2 and i == last:
                return (x, a, h) if return_h else (x, a)
            lib.gemm(a, self._w(p + "mlp.down_proj.weight"), residual=x, out_f32=x)
            dd = deltas.get(p + "mlp.down_proj")
            if dd is not None:
                lib.mend_apply_(a, dd["xt"], dd["dtT"], x)       # K17: x += (a . xt^T) . dt
            self.add_extra_neurons(i, h, x)
        return x, None

    def add_extra_neurons(self, layer, h, x):
        """x += (silu(h KG^T + BG) * (h KU^T + BU)) V for the patch neurons appended to `layer`'s FFN (TP_VL on the LLaMA FFN,
        tp_vl.py:74-103 with gate_proj and up_proj as in-layers: the hooks concatenate the extra columns to both projections'
        outputs, the MLP's silu(gate) * up acts on all of them and down_proj's hooks add extra_act @ extra_values).
        self.extra_neurons[layer] = {"KGU": [2 n_pad, d] (gate keys, then up keys), "BGU": fp32 [2 n_pad], "VT": [d, n_pad]}
        in the operand dtype; zero-padded neurons contribute silu(0) * 0."""
        ent = (getattr(self, "extra_neurons", None) or {}).get(layer)
        if ent is not None:
            lib.gemm(lib.swiglu(lib.gemm(h, ent["KGU"], ent["BGU"])), ent["VT"], residual=x, out_f32=x)

    # ---- MEND_VL support: module kinds, delta installation, explicit backward ---------------------------------------------
    MEND_MODULE_RE = r"^(.*\.layers\.)(\d+)\.mlp\.(gate_proj|up_proj|down_proj)$"

    def set_module_deltas(self, deltas):
        """deltas: {module name: {"xt" [n_pad, d_in], "xtT", "dt" [n_pad, d_out] (scaled), "dtT"}} in the operand dtype."""
        self.module_deltas = deltas
        self._gu_cat = {}
        F = self.t["intermediate_size"]
        for i in {int(n.split(".layers.")[1].split(".")[0]) for n in deltas if n.endswith("gate_proj") or n.endswith("up_proj")}:
            p = self.lm + "model.layers.%d." % i
            cols = [self.m.fused_w["llama_gu.%d" % i]]
            for kind, top in (("gate_proj", True), ("up_proj", False)):
                e = deltas.get(p + "mlp." + kind)
                if e is not None:
                    z = torch.zeros_like(e["dtT"])
                    cols.append(torch.cat([e["dtT"], z] if top else [z, e["dtT"]], 0))    # [2F, n_pad]
            self._gu_cat[i] = torch.cat(cols, 1).contiguous()

    def _wt(self, key, getter):
        c = self.__dict__.setdefault("_wt_cache", {})
        if key not in c:
            c[key] = getter().t().contiguous()
        return c[key]

    # ---- full fine-tuning of the language model (LTE_VL training); see Blip2Engine ----
    _padk = staticmethod(Blip2Engine._padk)
    acc_linear_grads = Blip2Engine.acc_linear_grads

    @property
    def LM_MODULE(self):
        return self.lm.rstrip(".")

    def train_params(self):
        """As Blip2Engine.train_params: fused q|k|v ("derived.llama_qkv.<i>") and gate|up ("derived.llama_gu.<i>") operands stand for
        the projections that are their row blocks.  embed_tokens is listed and never receives a gradient (training inputs arrive as
        embeddings; lm_head is not tied), exactly as in the reference, where Adam skips parameters without a gradient."""
        m, out = self.m, {}
        for n, p_ in getattr(m, self.LM_MODULE).named_parameters():
            name = self.LM_MODULE + "." + n
            if m._fused_slot(name) is None:
                out[name] = p_.data
        for key, w in m.fused_w.items():
            if key.startswith("llama_"):
                out["derived." + key] = w
        return out

    @torch.no_grad()
    def head_fwd(self, pre_ln):
        hn = lib.rmsnorm(pre_ln, self._p(self.lm + "model.norm.weight"), self.eps, want=self.want)
        return hn, lib.gemm(hn, self._w(self.lm + "lm_head.weight"), want="f32")

    @torch.no_grad()
    def head_bwd(self, pre_ln, hn, dlog, grads):
        self.acc_linear_grads(grads, self.lm + "lm_head.weight", None, hn, dlog)
        dH = lib.gemm(dlog, self.m.embed_T, want="f32")
        lib.layernorm_bwd_params(pre_ln, dH, self.eps, grads[self.lm + "model.norm.weight"], None, rms=True)
        return self.final_norm_bwd(pre_ln, dH)

    def embed_bwd(self, mask, dx0, grads):
        pass            # rotary positions: nothing learned on the input side

    def after_param_update(self):
        self.__dict__.pop("_wt_cache", None)
        self.m.refresh_derived(force=True)

    @torch.no_grad()
    def decoder_backward(self, ps, save, dx, capture, grads=None):
        """Backward through the saved LLaMA layers (highest first); see Blip2Engine.decoder_backward.  Captures (input rows,
        output-gradient rows) of the gate / up / down projections named in `capture`; with `grads` every decoder-layer parameter
        receives its gradient (fused q|k|v and gate|up operands under "derived.llama_qkv.<i>" / "derived.llama_gu.<i>")."""
        t, m = self.t, self.m
        d, H, F = t["hidden_size"], t["num_attention_heads"], t["intermediate_size"]
        dh = d // H
        n_seq = ps.desc.shape[0]
        deltas = getattr(self, "module_deltas", None) or {}
        neg_pos = (-ps.pos).contiguous()
        out = {}
        for i in sorted(save["layers"], reverse=True):
            p = self.lm + "model.layers.%d." % i
            rec = save[i]
            dz = self._act(dx)
            if p + "mlp.down_proj" in capture:
                out[p + "mlp.down_proj"] = (rec["a"], dx.clone())
            if grads is not None:
                self.acc_linear_grads(grads, p + "mlp.down_proj.weight", None, rec["a"], dx)
            da = lib.gemm(dz, self._wt(p + "down", lambda: self._w(p + "mlp.down_proj.weight")), want="f32")
            dd = deltas.get(p + "mlp.down_proj")
            if dd is not None:
                lib.gemm(lib.gemm(dz, dd["dt"]), dd["xtT"], residual=da, out_f32=da)
            dgu = lib.swiglu_bwd(rec["gu"], da)                      # fp32 [R, 2F]
            dgate, dup = dgu[:, :F].contiguous(), dgu[:, F:].contiguous()
            if p + "mlp.gate_proj" in capture:
                out[p + "mlp.gate_proj"] = (rec["h2"], dgate)
            if p + "mlp.up_proj" in capture:
                out[p + "mlp.up_proj"] = (rec["h2"], dup)
            if grads is not None:
                self.acc_linear_grads(grads, "derived.llama_gu.%d" % i, None, rec["h2"], dgu)
            dh2 = lib.gemm(self._act(dgu), self._wt(p + "gu", lambda: m.fused_w["llama_gu.%d" % i]), want="f32")
            for e, g in ((deltas.get(p + "mlp.gate_proj"), dgate), (deltas.get(p + "mlp.up_proj"), dup)):
                if e is not None:
                    lib.gemm(lib.gemm(self._act(g), e["dt"]), e["xtT"], residual=dh2, out_f32=dh2)
            dy = lib.rmsnorm_bwd_dx(rec["x_mid"], self._p(p + "post_attention_layernorm.weight"), dh2, self.eps)
            if grads is not None:
                lib.layernorm_bwd_params(rec["x_mid"], dh2, self.eps, grads[p + "post_attention_layernorm.weight"], None, rms=True)
            lib.delta_op(1, dy, None, dx)
            if grads is not None:
                self.acc_linear_grads(grads, p + "self_attn.o_proj.weight", None, rec["att"], dy)
            datt = lib.gemm(self._act(dy), self._wt(p + "o", lambda: self._w(p + "self_attn.o_proj.weight")))
            qkv = rec["qkv"]
            dq, dk, dv = lib.attention_bwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], rec["att"], datt, ps.desc, n_seq,
                                           ps.max_len, H, dh, dh ** -0.5, 1)
            dqk = torch.cat([dq, dk], 1).contiguous()
            lib.rope_(dqk, neg_pos, 2 * H, dh, self.theta)            # the rotation's transpose = rotation by -pos
            dqkv = torch.cat([dqk, dv], 1)
            if grads is not None:
                h1 = lib.rmsnorm(rec["x_in"], self._p(p + "input_layernorm.weight"), self.eps, want=self.want)
                self.acc_linear_grads(grads, "derived.llama_qkv.%d" % i, None, h1, dqkv)
            dh1 = lib.gemm(dqkv, self._wt(p + "qkv", lambda: m.fused_w["llama_qkv.%d" % i]), want="f32")
            if grads is not None:
                lib.layernorm_bwd_params(rec["x_in"], dh1, self.eps, grads[p + "input_layernorm.weight"], None, rms=True)
            dx = lib.rmsnorm_bwd_dx(rec["x_in"], self._p(p + "input_layernorm.weight"), dh1, self.eps)
            lib.delta_op(1, dx, None, dy)
        return out, dx

    @torch.no_grad()
    def lm_head(self, x_rows, add=None):
        ctx = self.path_ctx()
        if ctx is not None:
            return ctx.llm_head(x_rows.contiguous(), add)
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
