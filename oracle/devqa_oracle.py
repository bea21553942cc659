"""CPU oracle for the DE-VQA edit-then-evaluate hot path (BLIP-2 + FT_VL).

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this file; the product (de-vqa_amd/) never does.

It is a plain-PyTorch fp32 CPU restatement of what the reference executes for
this path.  `R/` = /root/reference/DE-VQA/.  The BLIP-2 / OPT forward lives in
the third-party `transformers` package (unpinned by the reference; 5.15.0 in the
build image) and is restated here from the op order the reference reaches
through R/editor/vllms_for_edit/blip2/blip2.py:25-31,35-48,63,69-74
(SURVEY.md Appendix D).

Parity pin: tests/test_oracle_golden.py checks every function here against
tests/golden/tiny_goldens.{npz,json} and realdim_goldens.{npz,json}, which were
produced by tools/make_goldens.py running the reference itself (imported from
/root/reference in the build container) on the same weights and inputs.
The reference ships no tests/golden vectors of its own for this path.
"""
import json
import math
import os
from copy import deepcopy

import numpy as np
import torch
import torch.nn.functional as F

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


# ---------------------------------------------------------------------------
# tokenizer wrapper (tokenizers lib; OPT conventions: '</s>' BOS via post-processor)
# ---------------------------------------------------------------------------
class OracleTokenizer:
    def __init__(self, tokenizer_json, pad_token_id=1):
        from tokenizers import Tokenizer
        self.tk = Tokenizer.from_file(tokenizer_json)
        self.pad_token_id = pad_token_id

    def encode(self, s):
        return self.tk.encode(s).ids

    def encode_no_special(self, s):
        return self.tk.encode(s, add_special_tokens=False).ids

    def decode(self, ids):
        return self.tk.decode([int(i) for i in ids], skip_special_tokens=False)


# ---------------------------------------------------------------------------
# model
# ---------------------------------------------------------------------------
def _ln(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def _lin(x, w, b=None):
    return F.linear(x, w, b)


class OracleBlip2:
    """BLIP-2-OPT forward on a flat {hf_param_name: tensor} dict (fp32, CPU)."""

    def __init__(self, weights, cfg, tokenizer, copy=True):
        self.w = {k: (v.detach().to(torch.float32).clone() if copy else v) for k, v in weights.items()}
        self.cfg = cfg
        self.tok = tokenizer
        v, q, t = cfg["vision_config"], cfg["qformer_config"], cfg["text_config"]
        self.v_layers, self.v_heads, self.v_eps = v["num_hidden_layers"], v["num_attention_heads"], v["layer_norm_eps"]
        self.patch, self.image_size = v["patch_size"], v["image_size"]
        self.q_layers, self.q_heads, self.q_eps = q["num_hidden_layers"], q["num_attention_heads"], q["layer_norm_eps"]
        self.q_xfreq = q["cross_attention_frequency"]
        self.t_layers, self.t_heads = t["num_hidden_layers"], t["num_attention_heads"]
        self.t_eps = 1e-5
        self.num_query_tokens = cfg["num_query_tokens"]

    # -- constructors ------------------------------------------------------
    @classmethod
    def from_pretrained_dir(cls, path):
        from safetensors.torch import load_file
        cfg = json.load(open(os.path.join(path, "config.json")))
        w = load_file(os.path.join(path, "model.safetensors"))
        tok = OracleTokenizer(os.path.join(path, "tokenizer.json"), cfg["text_config"].get("pad_token_id", 1))
        return cls(w, cfg, tok)

    # -- K1 image preprocessing (HF BlipImageProcessor as called at blip2.py:56-58)
    def preprocess_image(self, img):
        from PIL import Image
        if isinstance(img, str):
            with Image.open(img) as im:
                img = im.copy()
        img = img.convert("RGB")
        S = self.image_size
        img = img.resize((S, S), resample=Image.BICUBIC)
        a = np.asarray(img).astype(np.float32) * np.float32(1.0 / 255.0)
        a = (a - np.asarray(CLIP_MEAN, np.float32)) / np.asarray(CLIP_STD, np.float32)
        return torch.from_numpy(a.transpose(2, 0, 1)[None].copy())

    # -- K2/K3 vision tower (blip2.py:25-31) -----------------------------------
    def vision(self, pixel_values):
        w = self.w
        x = F.conv2d(pixel_values, w["vision_model.embeddings.patch_embedding.weight"],
                     w["vision_model.embeddings.patch_embedding.bias"], stride=self.patch)
        x = x.flatten(2).transpose(1, 2)
        B = x.shape[0]
        x = torch.cat([w["vision_model.embeddings.class_embedding"].expand(B, 1, -1), x], 1)
        x = x + w["vision_model.embeddings.position_embedding"][:, :x.shape[1]]
        H = self.v_heads
        for i in range(self.v_layers):
            p = "vision_model.encoder.layers.%d." % i
            h = _ln(x, w[p + "layer_norm1.weight"], w[p + "layer_norm1.bias"], self.v_eps)
            qkv = _lin(h, w[p + "self_attn.qkv.weight"], w.get(p + "self_attn.qkv.bias"))
            Bn, N, D3 = qkv.shape
            dh = D3 // 3 // H
            qkv = qkv.reshape(Bn, N, 3, H, dh).permute(2, 0, 3, 1, 4)
            q, k, v = qkv[0], qkv[1], qkv[2]
            a = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) * dh ** -0.5, -1)
            o = torch.matmul(a, v).transpose(1, 2).reshape(Bn, N, H * dh)
            x = x + _lin(o, w[p + "self_attn.projection.weight"], w[p + "self_attn.projection.bias"])
            h = _ln(x, w[p + "layer_norm2.weight"], w[p + "layer_norm2.bias"], self.v_eps)
            h = F.gelu(_lin(h, w[p + "mlp.fc1.weight"], w[p + "mlp.fc1.bias"]))
            x = x + _lin(h, w[p + "mlp.fc2.weight"], w[p + "mlp.fc2.bias"])
        return _ln(x, w["vision_model.post_layernorm.weight"], w["vision_model.post_layernorm.bias"], self.v_eps)

    # -- K4 Q-Former (blip2.py:33-43) ----------------------------------------------
    def _bert_attn(self, p, hq, hkv):
        w, H = self.w, self.q_heads
        q = _lin(hq, w[p + "attention.query.weight"], w[p + "attention.query.bias"])
        k = _lin(hkv, w[p + "attention.key.weight"], w[p + "attention.key.bias"])
        v = _lin(hkv, w[p + "attention.value.weight"], w[p + "attention.value.bias"])
        B, Nq, D = q.shape
        dh = D // H
        q = q.view(B, Nq, H, dh).transpose(1, 2)
        k = k.view(B, -1, H, dh).transpose(1, 2)
        v = v.view(B, -1, H, dh).transpose(1, 2)
        a = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) * dh ** -0.5, -1)
        o = torch.matmul(a, v).transpose(1, 2).reshape(B, Nq, D)
        o = _lin(o, w[p + "output.dense.weight"], w[p + "output.dense.bias"])
        return _ln(o + hq, w[p + "output.LayerNorm.weight"], w[p + "output.LayerNorm.bias"], self.q_eps)

    def qformer(self, image_embeds):
        w = self.w
        B = image_embeds.shape[0]
        h = w["query_tokens"].expand(B, -1, -1)
        h = _ln(h, w["qformer.layernorm.weight"], w["qformer.layernorm.bias"], self.q_eps)
        for i in range(self.q_layers):
            p = "qformer.encoder.layer.%d." % i
            a = self._bert_attn(p + "attention.", h, h)
            if i % self.q_xfreq == 0:
                a = self._bert_attn(p + "crossattention.", a, image_embeds)
            f = F.gelu(_lin(a, w[p + "intermediate_query.dense.weight"], w[p + "intermediate_query.dense.bias"]))
            f = _lin(f, w[p + "output_query.dense.weight"], w[p + "output_query.dense.bias"])
            h = _ln(f + a, w[p + "output_query.LayerNorm.weight"], w[p + "output_query.LayerNorm.bias"], self.q_eps)
        return h

    def image_tokens(self, pixel_values):
        """ViT -> Q-Former -> language projection: [B, Q, d_llm] (blip2.py:25-45)."""
        q = self.qformer(self.vision(pixel_values))
        return _lin(q, self.w["language_projection.weight"], self.w["language_projection.bias"])

    # -- tokenisation with right padding (HF tokenizer(padding=True)) ------------------
    def _tok_batch(self, texts):
        ids = [self.tok.encode(t) for t in texts]
        T = max(len(i) for i in ids)
        inp = torch.full((len(ids), T), self.tok.pad_token_id, dtype=torch.long)
        msk = torch.zeros((len(ids), T), dtype=torch.long)
        for r, i in enumerate(ids):
            inp[r, :len(i)] = torch.tensor(i)
            msk[r, :len(i)] = 1
        return inp, msk

    # -- A6 get_llm_input_embeds (blip2.py:20-66) ---------------------------------------------
    def get_llm_input_embeds(self, texts, imgs=None):
        if isinstance(imgs, list) and all(i is None for i in imgs):
            imgs = None  # wrapper rule, R/editor/vllms_for_edit/base.py:46-47
        ids, msk = self._tok_batch(texts)
        emb = self.w["language_model.model.decoder.embed_tokens.weight"][ids]
        if imgs is not None:
            img = imgs[-1] if isinstance(imgs, list) else imgs  # quirk: last image only (blip2.py:54-55)
            pv = self.preprocess_image(img)
            it = self.image_tokens(pv)  # [1,Q,d] -- broadcast against a batch of texts like HF cat would fail; B==1 in practice
            if it.shape[0] != emb.shape[0]:
                it = it.expand(emb.shape[0], -1, -1)
            emb = torch.cat([it, emb], 1)
            msk = torch.cat([torch.ones(it.shape[:2], dtype=torch.long), msk], 1)
            return {"attention_mask": msk, "inputs_embeds": emb}, [0, self.num_query_tokens]
        return {"attention_mask": msk, "inputs_embeds": emb}, None

    # -- K7/K8 decoder (blip2.py:68-75) ----------------------------------------------
    def llm_hidden_to_fc2_input(self, inputs_embeds, attention_mask, upto_layer=None):
        """Runs the decoder; returns (h_final_before_final_ln) and, for layer `upto_layer`,
        the pair (residual before the FFN add, a = relu(fc1(ln(h)))) -- the frozen prefix of FT_VL."""
        w, H = self.w, self.t_heads
        B, T, D = inputs_embeds.shape
        am = attention_mask.to(torch.long)
        pos = (torch.cumsum(am, 1) * am - 1) + 2
        x = inputs_embeds + w["language_model.model.decoder.embed_positions.weight"][pos]
        neg = torch.finfo(torch.float32).min
        causal = torch.ones(T, T, dtype=torch.bool).tril()
        allow = causal[None, None] & am.bool()[:, None, None, :]
        bias = torch.zeros(B, 1, T, T).masked_fill(~allow, neg)
        dh = D // H
        taps = None
        for i in range(self.t_layers):
            p = "language_model.model.decoder.layers.%d." % i
            h = _ln(x, w[p + "self_attn_layer_norm.weight"], w[p + "self_attn_layer_norm.bias"], self.t_eps)
            q = _lin(h, w[p + "self_attn.q_proj.weight"], w[p + "self_attn.q_proj.bias"]) * dh ** -0.5
            k = _lin(h, w[p + "self_attn.k_proj.weight"], w[p + "self_attn.k_proj.bias"])
            v = _lin(h, w[p + "self_attn.v_proj.weight"], w[p + "self_attn.v_proj.bias"])
            q = q.view(B, T, H, dh).transpose(1, 2)
            k = k.view(B, T, H, dh).transpose(1, 2)
            v = v.view(B, T, H, dh).transpose(1, 2)
            a = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) + bias, -1)
            o = torch.matmul(a, v).transpose(1, 2).reshape(B, T, D)
            x = x + _lin(o, w[p + "self_attn.out_proj.weight"], w[p + "self_attn.out_proj.bias"])
            h = _ln(x, w[p + "final_layer_norm.weight"], w[p + "final_layer_norm.bias"], self.t_eps)
            hook = getattr(self, "module_hook", None)  # (module name, input, output) -> output; MEND-style editors
            f1 = _lin(h, w[p + "fc1.weight"], w[p + "fc1.bias"])
            if hook is not None:
                f1 = hook(p + "fc1", h, f1)
            a1 = F.relu(f1)
            if upto_layer is not None and i == upto_layer:
                taps = (x, a1)
            f2 = _lin(a1, w[p + "fc2.weight"], w[p + "fc2.bias"])
            if hook is not None:
                f2 = hook(p + "fc2", a1, f2)
            x = x + f2
        return x, taps

    def get_llm_outpt(self, llm_inpt, vt_range=None):
        x, _ = self.llm_hidden_to_fc2_input(llm_inpt["inputs_embeds"], llm_inpt["attention_mask"])
        w = self.w
        x = _ln(x, w["language_model.model.decoder.final_layer_norm.weight"],
                w["language_model.model.decoder.final_layer_norm.bias"], self.t_eps)
        return _lin(x, w["language_model.model.decoder.embed_tokens.weight"])  # tied lm_head, no bias

    # -- A4 prompts_imgs_target_to_xym (R/editor/vllms_for_edit/base.py:75-109) ---------------
    def prompts_imgs_target_to_xym(self, prompts, imgs, targets):
        targets = [" " + t if p[-1] not in [" ", "\n"] and t[0] not in [" ", "\n"] else t
                   for p, t in zip(prompts, targets)]
        input_strs, label_ids, label_masks = [], [], []
        min_prompt_tok_n = 999
        for p, t in zip(prompts, targets):
            s = p + t
            input_strs.append(s)
            lt = torch.roll(torch.tensor(self.tok.encode(s), dtype=torch.long), -1, 0)
            label_ids.append(lt)
            m = torch.zeros_like(lt)
            n_p = len(self.tok.encode(p))
            min_prompt_tok_n = min(min_prompt_tok_n, n_p)
            m[n_p - 1:-1] += 1
            label_masks.append(m)
        x, vt = self.get_llm_input_embeds(input_strs, imgs)
        from torch.nn.utils.rnn import pad_sequence
        y = pad_sequence(label_ids, True, self.tok.pad_token_id)[:, min_prompt_tok_n - 1:]
        m = pad_sequence(label_masks, True, 0)[:, min_prompt_tok_n - 1:]
        return (x, vt), y, m


# -- K9 label_loss (R/editor/vllm_editors/ft_vl/ft_vl.py:191-199; base.py:111-119) -----------
def label_loss(logits, label_ids, masks, average=True):
    logits = logits[:, -label_ids.shape[1]:]
    lp = torch.log_softmax(logits, -1).gather(-1, label_ids.unsqueeze(-1)).squeeze(-1)
    loss = -(lp * masks).sum()
    return loss / masks.sum() if average else loss


# -- K18 logit_KL_loss (R/editor/vllms_for_edit/base.py:121-132) -------------------------------
def logit_KL_loss(logits1, logits2, label_masks, average=True):
    l1 = logits1[:, -label_masks.shape[1]:]
    l2 = logits2[:, -label_masks.shape[1]:]
    kl = (torch.softmax(l1, 2) * (torch.log_softmax(l1, -1) - torch.log_softmax(l2, -1))).sum(2)
    loss = (kl * label_masks).sum()
    return loss / label_masks.sum() if average else loss


# ---------------------------------------------------------------------------
# A8 FT_VL (R/editor/vllm_editors/ft_vl/ft_vl.py)
# ---------------------------------------------------------------------------
class OracleFTvl:
    """Faithful replay of FTvl: per step recompute everything, autograd backward onto the
    selected weights, torch.optim.AdamW (ft_vl.py:66-158)."""

    def __init__(self, model: OracleBlip2, layers, rewrite_module_tmp, num_steps=25, lr=1e-3, weight_decay=0,
                 norm_constraint=False, batch_size=1):
        self.m = model
        self.layers, self.tmp = layers, rewrite_module_tmp
        self.num_steps, self.lr, self.wd, self.norm_constraint = num_steps, lr, weight_decay, norm_constraint
        self.batch_size = batch_size
        # substring selection rule, ft_vl.py:31-36
        self.names = [n for n in model.w for layer in layers if rewrite_module_tmp.format(layer) in n]
        self.original_w = {n: model.w[n].clone() for n in self.names}
        self.last_losses = []

    def restore_to_original_model(self):  # ft_vl.py:44-45
        for n, v in self.original_w.items():
            self.m.w[n] = v.clone()

    def execute_ft(self, requests):
        requests = deepcopy(requests)
        for r in requests:
            if r["target_new"][0] != " ":
                r["target_new"] = " " + r["target_new"]  # ft_vl.py:73-75
        m = self.m
        weights = {n: m.w[n] for n in self.names}
        weights_copy = {k: v.detach().clone() for k, v in weights.items()}
        params = []
        for n in self.names:
            p = m.w[n].detach().clone().requires_grad_(True)
            m.w[n] = p
            params.append(p)
        opt = torch.optim.AdamW(params, lr=self.lr, weight_decay=self.wd)
        self.last_losses = []

        def chunks(a, n):
            for i in range(0, len(a), n):
                yield a[i:i + n]
        images = [r["image"] for r in requests]
        texts = [r["prompt"] for r in requests]
        targets = [r["target_new"] for r in requests]
        for _ in range(self.num_steps):
            s, c = 0.0, 0
            for imgs, txt, tgt in zip(chunks(images, self.batch_size), chunks(texts, self.batch_size),
                                      chunks(targets, self.batch_size)):
                (x, vt), y, msk = m.prompts_imgs_target_to_xym(txt, imgs, tgt)
                opt.zero_grad()
                loss = label_loss(m.get_llm_outpt(x, vt), y, msk)
                lv = loss.item()
                self.last_losses.append(lv)
                s += lv * y.shape[0]
                c += y.shape[0]
                if lv >= 1e-2:  # ft_vl.py:131-133
                    loss.backward()
                    opt.step()
                if type(self.norm_constraint) is float:  # ft_vl.py:135-141
                    with torch.no_grad():
                        for n, p in zip(self.names, params):
                            p[...] = torch.clamp(p, min=weights_copy[n] - self.norm_constraint,
                                                 max=weights_copy[n] + self.norm_constraint)
            if s / c < 1e-2:  # ft_vl.py:145-146
                break
        deltas = {n: (p.detach() - weights_copy[n]) for n, p in zip(self.names, params)}
        for n in self.names:
            m.w[n] = weights_copy[n]
        return deltas

    def edit_one_piece(self, request):  # ft_vl.py:47-61
        deltas = self.execute_ft([request])
        for n, d in deltas.items():
            self.m.w[n] = self.m.w[n] + d


# ---------------------------------------------------------------------------
# A1/A9/A10 evaluator (R/evaluation/vllm_editor_eval.py:69-247)
# ---------------------------------------------------------------------------
def split_data(data, edit_n):  # vllm_editor_eval.py:74-87 (incomplete tail dropped)
    out, ns, cur, n = [], [], [], 0
    for d in data:
        cur.append(d)
        n += len(d["requests"])
        if n >= edit_n:
            out.append(cur)
            ns.append(n)
            cur, n = [], 0
    return out, ns


def _acc_pred(model, prompt, image, target, labels_override=None):
    (x, vt), y, m = model.prompts_imgs_target_to_xym([prompt], [image], [target])
    x["query_triple"] = (prompt, image, target)   # dynamic-eval hook read by retrieval editors (vllm_editor_eval.py:140)
    logits = model.get_llm_outpt(x, vt)
    pre = torch.softmax(logits, -1).argmax(-1)[:, -y.shape[1]:]
    lab = y if labels_override is None else labels_override
    acc = float(((pre == lab) * m).sum() / m.sum())
    return acc, pre, m


def evaluate_sequential_edit(model: OracleBlip2, editor: OracleFTvl, records, edit_n):
    """Returns results[split][sample] with the reference's schema minus edit_time."""
    eval_data, ns = split_data(deepcopy(records), edit_n)
    result_data, _ = split_data(deepcopy(records), edit_n)
    tok = model.tok
    editor.restore_to_original_model()
    results = []
    with torch.no_grad():
        pass
    for split_rd, split_ed in zip(result_data, eval_data):
        split_res = []
        for rd, ed in zip(split_rd, split_ed):
            rd["reliability"] = rd.pop("requests")
            for r in rd["reliability"]:
                r["target"] = r.pop("target_new")
            for name in ed["locality"]:
                for rdl, edl in zip(rd["locality"][name], ed["locality"][name]):
                    with torch.no_grad():
                        _, pre, m = _acc_pred(model, edl["prompt"], edl["image"], edl["target"])
                    rdl["predict_before_edit"] = tok.decode(pre[m.bool()].tolist())
                    edl["before_edit_ids"] = pre
        for rd, ed in zip(split_rd, split_ed):
            for edr in ed["requests"]:
                editor.edit_one_piece(edr)
        for rd, ed in zip(split_rd, split_ed):
            with torch.no_grad():
                for rdr, edr in zip(rd["reliability"], ed["requests"]):
                    acc, pre, m = _acc_pred(model, edr["prompt"], edr["image"], edr["target_new"])
                    rdr["predict_after_edit"] = tok.decode(pre[m.bool()].tolist())
                    rdr["acc"] = acc
                for g in ed["generality"]:
                    for rdg, edg in zip(rd["generality"][g], ed["generality"][g]):
                        acc, pre, m = _acc_pred(model, edg["prompt"], edg["image"], edg["target"])
                        rdg["predict_after_edit"] = tok.decode(pre[m.bool()].tolist())
                        rdg["acc"] = acc
                for name in ed["locality"]:
                    for rdl, edl in zip(rd["locality"][name], ed["locality"][name]):
                        acc, pre, m = _acc_pred(model, edl["prompt"], edl["image"], edl["target"],
                                                edl["before_edit_ids"])
                        rdl["predict_after_edit"] = tok.decode(pre[m.bool()].tolist())
                        rdl["acc"] = acc
            split_res.append(rd)
        editor.restore_to_original_model()
        results.append(split_res)
    return results, ns


# ---------------------------------------------------------------------------
# One reference-style cycle on PRE-TOKENISED synthetic inputs (bench.py's cpu_baseline leg)
# ---------------------------------------------------------------------------
def _pretok_xym(model, prompt_ids, pixels, target_ids, img_cache=None):
    """prompts_imgs_target_to_xym (R/editor/vllms_for_edit/base.py:97-108) for token-id lists and pre-processed pixel values
    [3,S,S]: labels = roll(ids, -1), mask[len(prompt)-1 : -1] = 1, both cropped to [len(prompt)-1:]; the image is ENCODED here,
    on every call, as the reference does (blip2.py:25-52)."""
    ids = list(prompt_ids) + list(target_ids)
    n_p = len(prompt_ids)
    t = torch.tensor(ids, dtype=torch.long)
    lab = torch.roll(t, -1, 0)
    m = torch.zeros_like(lab)
    m[n_p - 1:-1] += 1
    emb = model.w["language_model.model.decoder.embed_tokens.weight"][t][None]
    if pixels is not None:
        key = pixels.data_ptr() if isinstance(pixels, torch.Tensor) else np.asarray(pixels).ctypes.data   # identity = the pixel memory
        if img_cache is not None and key in img_cache:
            it = img_cache[key][0]
        else:
            it = model.image_tokens(torch.as_tensor(pixels, dtype=torch.float32)[None])
            if img_cache is not None:
                img_cache[key] = (it, pixels)       # the pixels object stays referenced: its memory cannot be reused
        emb = torch.cat([it, emb], 1)
    x = {"inputs_embeds": emb, "attention_mask": torch.ones(emb.shape[:2], dtype=torch.long)}
    return x, lab[n_p - 1:][None], m[n_p - 1:][None]


def faithful_cycle_pretokenized(model: OracleBlip2, cycle, weight_name, num_steps=25, lr=1e-3, weight_decay=0.0, trace=False,
                                cache_images=False):
    """ONE split of `evaluate_sequential_edit` with edit_n = 1 (vllm_editor_eval.py:100-123) around ONE `FTvl.edit_one_piece`
    (ft_vl.py:47-158) exactly in the reference's call sequence -- B = 1, nothing cached or shared: 9 pre-edit locality forwards
    (6 with an image encode), <= num_steps x [image encode + decoder forward + backward onto `weight_name` + torch.optim.AdamW],
    12 post-edit forwards (9 with an image encode), restore.  `cycle`: a devqa_amd.synth.evqa_cycles sample whose images are
    pre-processed pixel arrays.  -> dict(accs=[12], steps, encodes, forwards)
    trace=True additionally returns what a full-depth parity check needs (bench.py's `parity` block): `losses` (the loss of every
    executed FT step, ft_vl.py:125-129) and `rows` = the fp32 logits of the last-L (label) rows of each of the 21 evaluator
    forwards, in call order: 9 pre-edit locality probes, then reliability, 2 generality, 9 post-edit locality probes.
    cache_images=True (parity-only runs, never the timed cpu_baseline): the image encoder is frozen and deterministic, so each
    distinct image is encoded once and its tokens reused -- the same values as re-encoding, ~3x less CPU time per cycle."""
    n_enc = n_fwd = 0
    losses, rows = [], []
    img_cache = {} if cache_images else None

    def forward(prompt, image, target):
        nonlocal n_enc, n_fwd
        x, y, m = _pretok_xym(model, prompt, image, target, img_cache)
        n_enc += image is not None
        n_fwd += 1
        return model.get_llm_outpt(x, None), y, m
    before = {}
    with torch.no_grad():
        for name, items in cycle["locality"].items():
            logits, y, m = forward(items[0]["prompt"], items[0]["image"], items[0]["target"])
            before[name] = torch.softmax(logits, -1).argmax(-1)[:, -y.shape[1]:]
            if trace:
                rows.append(logits[0, -y.shape[1]:].clone())
    req = cycle["requests"][0]
    w0 = model.w[weight_name]
    p = w0.detach().clone().requires_grad_(True)
    model.w[weight_name] = p
    opt = torch.optim.AdamW([p], lr=lr, weight_decay=weight_decay)
    steps = 0
    for _ in range(num_steps):
        opt.zero_grad()
        logits, y, m = forward(req["prompt"], req["image"], req["target_new"])
        loss = label_loss(logits, y, m)
        lv = loss.item()
        losses.append(lv)
        steps += 1
        if lv >= 1e-2:
            loss.backward()
            opt.step()
        if lv < 1e-2:
            break
    model.w[weight_name] = p.detach()
    accs = []
    with torch.no_grad():
        probes = [(req["prompt"], req["image"], req["target_new"], None)]
        probes += [(it[0]["prompt"], it[0]["image"], it[0]["target"], None) for it in cycle["generality"].values()]
        probes += [(it[0]["prompt"], it[0]["image"], it[0]["target"], before[n]) for n, it in cycle["locality"].items()]
        for prompt, image, target, ref in probes:
            logits, y, m = forward(prompt, image, target)
            pre = torch.softmax(logits, -1).argmax(-1)[:, -y.shape[1]:]
            accs.append(float(((pre == (y if ref is None else ref)) * m).sum() / m.sum()))
            if trace:
                rows.append(logits[0, -y.shape[1]:].clone())
    model.w[weight_name] = w0
    out = {"accs": accs, "steps": steps, "encodes": int(n_enc), "forwards": int(n_fwd)}
    if trace:
        out.update(losses=losses, rows=rows)
    return out


def get_mean_results(results):  # vllm_editor_eval.py:177-229
    mean = {"reliability": {}, "generality": {}, "locality": {}}

    def add(dst, item):
        for k, v in item.items():
            if isinstance(v, (int, float)):
                dst.setdefault(k, [0, 0])
                dst[k][0] += v
                dst[k][1] += 1
    for r in results:
        for rr in r["reliability"]:
            add(mean["reliability"], rr)
        for sec in ("generality", "locality"):
            for sub in r[sec]:
                mean[sec].setdefault(sub, {})
                for it in r[sec][sub]:
                    add(mean[sec][sub], it)
    mean["reliability"] = {k: v[0] / v[1] for k, v in mean["reliability"].items()}
    for sec in ("generality", "locality"):
        for sub in mean[sec]:
            mean[sec][sub] = {k: v[0] / v[1] for k, v in mean[sec][sub].items()}
    return mean


def round4(r):  # vllm_editor_eval.py:231-241
    if isinstance(r, list):
        return [round4(x) for x in r]
    if isinstance(r, dict):
        return {k: round4(v) for k, v in r.items()}
    if isinstance(r, float):
        return round(r, 4)
    return r


# ---------------------------------------------------------------------------
# K19 cosine top-k (R/dataset/vllm.py:65-70; R/easyeditor/models/ike/ike_main.py:193-202)
# sentence_transformers.util (third-party, unpinned, absent here): normalize_embeddings = row L2
# normalise; semantic_search with dot_score = per-query top-k of Q.C^T sorted by descending score.
# "parity unpinned" by the reference; pinned by float64 brute force. Tie-break: lowest corpus id.
# ---------------------------------------------------------------------------
def cosine_topk(corpus, queries, k, normalize_corpus=True, normalize_queries=True):
    c = np.asarray(corpus, np.float64)
    q = np.asarray(queries, np.float64)
    if normalize_corpus:
        c = c / np.linalg.norm(c, axis=1, keepdims=True)
    if normalize_queries:
        q = q / np.linalg.norm(q, axis=1, keepdims=True)
    s = q @ c.T
    idx = np.lexsort((np.broadcast_to(np.arange(c.shape[0]), s.shape), -s), axis=1)[:, :k]
    return idx.astype(np.int64), np.take_along_axis(s, idx, 1)


def finds_sim_select(hit_ids, prompts, trg):
    """R/dataset/vllm.py:72-81: first hit whose stored answer != trg, else the last hit."""
    for i in hit_ids:
        if prompts[i][1] != trg:
            return int(i)
    return int(hit_ids[-1])
