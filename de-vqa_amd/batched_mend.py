"""BatchedMendEval: many independent MEND_VL edit+eval cycles per GPU (BASELINE config #4), splits sharded across ranks.

One CYCLE = one split of `evaluate_sequential_edit` with edit_n == 1 (R/evaluation/vllm_editor_eval.py:100-123) around one
`MENDvl.edit_one_piece` (R/editor/vllm_editors/mend_vl/mend_vl.py:169-195): 9 pre-edit locality probes -> one forward + backward of
the edit sequence with the inputs x and output gradients delta of the edited Linear modules captured (:63-85) -> GradientTransform
(auxiliary_networks.py:110-151) -> low-rank module deltas dW = x~^T d~ * lr / n (:98-114) -> 12 post-edit probes through the
edited modules (:72-79) -> restore.

What MEND_VL shares with FT_VL's batched engine (batched.py), and how the rest is laid out:

  * ViT + Q-Former once per unique image of the batch; every decoder layer BELOW the first edited one (0 .. 28 of 32 in the shipped
    config) once per unique (image, text) sequence, image-token prefixes packed once per cycle (attention is causal: their hidden
    states do not depend on the text) -- one path-level call (devqa_llm_layers) for the whole batch;
  * the frozen rows entering the first edited layer are kept: the pre-edit pass runs the pristine edited layers on them, the
    post-edit pass runs the same layers again with each cycle's own low-rank deltas -- the big GEMMs stay batched over all cycles,
    the per-cycle part is two skinny GEMMs per edited module on that cycle's contiguous row segment;
  * the E edit sequences form a second pack (own image prefix each: the backward reaches the image rows): forward through the
    edited layers with saved activations, one head / NLL / dlogits call, one explicit backward (engine.decoder_backward), and ONE
    hyper-network call per edited module over the rows of all cycles (the transform is row-wise);
  * which rows can carry a non-zero gradient is known on the host (modules of the last layer: the label rows; below: every row up
    to the last label row), so the factors are sized without a device -> host read; the reference's exact criterion (rows with a
    non-zero input AND a non-zero output gradient, auxiliary_networks.py:118-120) is then applied on the device as a mask, which also
    yields the per-cycle row count n of the running mean.

Results equal the generic per-sample path (tests/test_mend_batched_gpu.py: fp32 exact on results.json, and vs the reference's own
MENDvl goldens).
"""
import time
from typing import Dict, List

import numpy as np
import torch

from . import lib
from .batched import BatchedEditEval, _Probe
from .engine import LN_EPS_OPT, PackedSeqs


class BatchedMendEval(BatchedEditEval):
    def __init__(self, editor, cycles_per_batch=None):
        super().__init__(editor, cycles_per_batch)
        self.lo = min(editor.layers)
        self.n_layers = self.eng.t["num_hidden_layers"]
        self.stats.update({"t_edit": 0.0})

    @staticmethod
    def supports(editor, eval_data, edit_n):
        try:
            from .editor.vllm_editors.mend_vl.mend_vl import MENDvl
            from .engine import Blip2Engine
        except Exception:
            return False
        eng = getattr(editor.vllm, "engine", None)
        if not (isinstance(editor, MENDvl) and type(eng) is Blip2Engine and editor.aux is not None and not editor.training):
            return False
        n = eng.t["num_hidden_layers"]
        if edit_n != 1 or set(editor.layers) != set(range(min(editor.layers), n)) or eng.path_ctx() is None:
            return False
        if editor.n_layers > lib.MEND_MAX_LAYERS or not getattr(editor, "_stats_finite", True):
            return False
        return all(len(split) == 1 and len(split[0]["requests"]) == 1 for split in eval_data)

    # ------------------------------------------------------------------------------------------
    def _pack(self, tok, src, pos, desc, rows, max_len):
        eng, dev = self.eng, self.eng.dev
        x = lib.embed_rows(lib.h2d(tok, torch.int32, dev), lib.h2d(src, torch.int32, dev), lib.h2d(pos, torch.int32, dev),
                           eng._p("language_model.model.decoder.embed_tokens.weight"), rows,
                           eng._p("language_model.model.decoder.embed_positions.weight"))
        return PackedSeqs(x, [d[0] for d in desc], [d[1] for d in desc], lib.h2d(desc, torch.int32, dev), max_len, True)

    @torch.no_grad()
    def _stage_a(self, rds: List[Dict], eds: List[Dict]):
        eng, vllm = self.eng, self.vllm
        dev, Qn = eng.dev, eng.Q
        t0 = time.time()
        E = len(eds)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        img_index: Dict = {}
        img_list = []

        def img_id(obj):
            if obj is None:
                return None
            key = obj if isinstance(obj, str) else ("obj", id(obj))
            if key not in img_index:
                img_index[key] = len(img_list)
                img_list.append(obj)
            return img_index[key]
        for ed in eds:      # unique images first: the vision encoder is queued before the token bookkeeping and runs under it
            for name in ed["locality"]:
                img_id(ed["locality"][name][0]["image"])
            img_id(ed["requests"][0]["image"])
            for name in ed["generality"]:
                img_id(ed["generality"][name][0]["image"])
        img_tokens = None
        if img_list:
            if all(isinstance(p, torch.Tensor) for p in img_list):
                pix = torch.stack(img_list)
            else:
                pix = torch.from_numpy(np.stack([vllm.load_pixels(p) for p in img_list])).to(dev, non_blocking=True)
            chunks, i0 = [], 0
            for c in eng.image_chunks(len(img_list)):
                chunks.append(eng.encode_images(pix[i0:i0 + c]))
                i0 += c
            img_tokens = torch.cat(chunks) if len(chunks) > 1 else chunks[0]
        t1 = time.time()
        # ---- probe pack P (per cycle contiguous: its image prefixes, then its distinct texts) and edit pack Q -------------------
        tok, src, pos, desc = [], [], [], []
        qtok, qsrc, qpos, qdesc = [], [], [], []
        probes, cyc_rows, edits = [], [], []
        r = rq = 0
        max_len, qmax = Qn, 1
        for rd, ed in zip(rds, eds):
            rd["reliability"] = rd.pop("requests")
            for rr in rd["reliability"]:
                rr["target"] = rr.pop("target_new")
            items = [("loc", name, ed["locality"][name][0], rd["locality"][name][0], "target") for name in ed["locality"]]
            items.append(("rel", None, ed["requests"][0], rd["reliability"][0], "target_new"))
            items += [("gen", name, ed["generality"][name][0], rd["generality"][name][0], "target") for name in ed["generality"]]
            recs = []
            for kind, name, item_ed, item_rd, tkey in items:
                ids, y, m = self._probe_seq(item_ed["prompt"], item_ed[tkey], item_ed["image"] is not None)
                recs.append((kind, name, item_ed, item_rd, ids, y, m, img_id(item_ed["image"])))
            r0 = r
            prefix, seq_at = {}, {}
            for img in dict.fromkeys(rc[7] for rc in recs if rc[7] is not None):
                prefix[img] = r
                tok += [0] * Qn
                src += list(range(img * Qn, (img + 1) * Qn))
                pos += list(range(Qn))
                desc.append([r, Qn, 0, 0, r, Qn])
                r += Qn
            plist = []
            for kind, name, item_ed, item_rd, ids, y, m, img in recs:
                key = (img, tuple(ids))
                if key not in seq_at:
                    n, off = len(ids), (0 if img is None else Qn)
                    seq_at[key] = r + n
                    tok += list(ids)
                    src += [-1] * n
                    pos += list(range(off, off + n))
                    desc.append([r, n, prefix[img], Qn, r, n] if img is not None else [r, n, 0, 0, r, n])
                    r += n
                    max_len = max(max_len, n)
                p = _Probe()
                p.kind, p.name, p.rd, p.ed = kind, name, item_rd, item_ed
                p.seq, p.L, p.labels, p.mask = seq_at[key], len(y), y, m        # p.seq: END row of the probe's sequence in pack P
                plist.append(p)
            probes.append(plist)
            cyc_rows.append((r0, r))
            # the edit sequence (mend_vl.py:169-175: prompts_imgs_target_to_xym of the request) with its OWN image rows
            req = ed["requests"][0]
            ids, y, m = self._probe_seq(req["prompt"], req["target_new"], req["image"] is not None)
            img = img_id(req["image"])
            off = 0
            if img is not None:
                qtok += [0] * Qn
                qsrc += list(range(img * Qn, (img + 1) * Qn))
                qpos += list(range(Qn))
                off = Qn
            qtok += list(ids)
            qsrc += [-1] * len(ids)
            qpos += list(range(off, off + len(ids)))
            n = off + len(ids)
            qdesc.append([rq, n, 0, 0, rq, n])
            L = len(y)
            lab_rows = [rq + n - L + j for j in range(L) if m[j] != 0]
            edits.append((rq, n, lab_rows, [y[j] for j in range(L) if m[j] != 0]))
            rq += n
            qmax = max(qmax, n)
        t2 = time.time()
        self.stats["t_host"] += t2 - t0
        ev[1].record()
        rows = None if img_tokens is None else img_tokens.reshape(-1, img_tokens.shape[-1]).contiguous()
        psP = self._pack(tok, src, pos, desc, rows, max_len)
        psQ = self._pack(qtok, qsrc, qpos, qdesc, rows, qmax)
        ctx = eng.path_ctx()
        if self.lo > 0:     # the frozen layers below the first edited one: one path-level call per pack
            ctx.llm_layers(psP.x, psP.desc, len(desc), psP.max_len, True, self.lo, False)
            ctx.llm_layers(psQ.x, psQ.desc, len(qdesc), psQ.max_len, True, self.lo, False)
        # label rows of every probe (cycle by cycle)
        row_idx = []
        for plist in probes:
            for p in plist:
                p.row0 = len(row_idx)
                row_idx += list(range(p.seq - p.L, p.seq))
        ev[2].record()
        return dict(rds=rds, probes=probes, E=E, psP=psP, psQ=psQ, cyc_rows=cyc_rows, edits=edits, ev=ev,
                    ridx=lib.h2d(row_idx, torch.int32, dev))

    # ------------------------------------------------------------------------------------------
    def _edited_layers(self, ps, deltas, cyc_rows):
        """Decoder layers lo .. last in place on ps.x.  deltas: None (pristine) or {module name: (xt [E, npad, d_in], dtT [E, d_out, npad])}
        in the operand dtype -- cycle e's low-rank delta applies to the rows cyc_rows[e] only (mend_vl.py:72-79: out += x @ dW)."""
        eng = self.eng
        t = eng.t
        d, H = t["hidden_size"], t["num_attention_heads"]
        dh = d // H
        x = ps.x
        n_seq = ps.desc.shape[0]
        for i in range(self.lo, self.n_layers):
            p = "language_model.model.decoder.layers.%d." % i
            h = eng._ln(x, p + "self_attn_layer_norm.weight", p + "self_attn_layer_norm.bias", LN_EPS_OPT)
            qkv = lib.gemm(h, eng.m.fused_qkv_w[str(i)], eng.m.fused_qkv_b[str(i)])
            att = lib.attention(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], ps.desc, n_seq, ps.max_len, H, dh, dh ** -0.5, 1)
            lib.gemm(att, eng._w(p + "self_attn.out_proj.weight"), eng._p(p + "self_attn.out_proj.bias"), residual=x, out_f32=x)
            h = eng._ln(x, p + "final_layer_norm.weight", p + "final_layer_norm.bias", LN_EPS_OPT)
            d1 = None if deltas is None else deltas.get(p + "fc1")
            if d1 is None:
                a = lib.gemm(h, eng._w(p + "fc1.weight"), eng._p(p + "fc1.bias"), act=lib.ACT_RELU)
            else:   # relu(h W1^T + b + (h x~^T) d~): the low-rank term enters BEFORE the activation
                xt, dtT = d1
                term = torch.empty((x.shape[0], dtT.shape[1]), dtype=torch.float32, device=x.device)
                for e, (r0, r1) in enumerate(cyc_rows):
                    lib.gemm(lib.gemm(h[r0:r1], xt[e]), dtT[e], out_f32=term[r0:r1])
                lib.gemm(h, eng._w(p + "fc1.weight"), eng._p(p + "fc1.bias"), residual=term, out_f32=term)
                a = lib.act_cast(term, lib.ACT_RELU, eng.want)
            lib.gemm(a, eng._w(p + "fc2.weight"), eng._p(p + "fc2.bias"), residual=x, out_f32=x)
            d2 = None if deltas is None else deltas.get(p + "fc2")
            if d2 is not None:
                xt, dtT = d2
                for e, (r0, r1) in enumerate(cyc_rows):
                    lib.mend_apply_(a[r0:r1], xt[e], dtT[e], x[r0:r1])
        return x

    def _stage_b_body(self, c):
        eng, ed = self.eng, self.editor
        dev = eng.dev
        E, psP, psQ, cyc_rows, edits, ridx = c["E"], c["psP"], c["psQ"], c["cyc_rows"], c["edits"], c["ridx"]
        evb = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        evb[0].record()
        # ---- pre-edit: the pristine edited layers on a copy of the frozen rows ------------------------------------------------------
        pre_ps = PackedSeqs(psP.x.clone(), psP.start, psP.length, psP.desc, psP.max_len, True)
        x_pre = self._edited_layers(pre_ps, None, cyc_rows)
        pre_logits = eng.lm_head(lib.gather_rows(x_pre, ridx))
        pre_argmax, _, _ = lib.vocab_rows(pre_logits)
        del x_pre, pre_ps
        evb[1].record()
        # ---- the E edits: forward with saved activations, NLL, explicit backward, hyper-network ----------------------------------------
        save = {"layers": set(ed.layers)}
        x_fin, _ = eng.decoder_layers(psQ, save=save, first_layer=self.lo)
        lab_rows = [r_ for e_ in edits for r_ in e_[2]]
        labels = [l_ for e_ in edits for l_ in e_[3]]
        coef = [1.0 / len(e_[2]) for e_ in edits for _ in e_[2]]          # label_loss: mean over the masked tokens of the request
        cyc_of_lab = [e for e, e_ in enumerate(edits) for _ in e_[2]]
        idx = lib.h2d(lab_rows, torch.int32, dev)
        pre_ln = lib.gather_rows(x_fin, idx)
        logits = eng.lm_head(pre_ln)
        _, nll, dlog = lib.vocab_rows(logits, lib.h2d(labels, torch.int32, dev), lib.h2d(coef, torch.float32, dev), want_argmax=False,
                                      want_nll=True, want_dlogits=True, dlogits_dtype=eng.adt)
        dH = lib.gemm_rows_longk(dlog, self.vllm.model.embed_T)
        dxr = eng.final_norm_bwd(pre_ln, dH)
        dx = torch.zeros_like(x_fin)
        dx.index_copy_(0, idx.long(), dxr)                                 # plumbing: scatter the gradient rows
        caps, _ = eng.decoder_backward(psQ, save, dx, {m["name"] for m in ed.modules})
        loss_c = torch.zeros(E, dtype=torch.float32, device=dev).index_add_(0, lib.h2d(cyc_of_lab, torch.int64, dev),
                                                                             nll * lib.h2d(coef, torch.float32, dev))
        deltas = {}
        last = self.n_layers - 1
        op = (lambda t_: lib.cast_f32_bf16(t_.contiguous())) if eng.adt == torch.bfloat16 else (lambda t_: t_.contiguous())
        for m in ed.modules:
            layer = int(m["name"].split(".layers.")[1].split(".")[0])
            # candidate rows (host-known superset of the rows with a non-zero gradient) and their slot in the per-cycle padding
            cand, slot, cyc = [], [], []
            per = [(e_[2] if layer == last else list(range(e_[0], e_[2][-1] + 1))) for e_ in edits]
            npad = (max(len(p_) for p_ in per) + 63) // 64 * 64
            for e, p_ in enumerate(per):
                cand += p_
                slot += list(range(e * npad, e * npad + len(p_)))
                cyc += [e] * len(p_)
            cidx = lib.h2d(cand, torch.int32, dev)
            xin32, d32 = caps[m["name"]]
            xg = lib.gather_rows(xin32.to(torch.float32).contiguous(), cidx)
            dg = lib.gather_rows(d32.to(torch.float32).contiguous(), cidx)
            valid = ((xg != 0).any(-1) & (dg != 0).any(-1)).to(torch.float32)          # auxiliary_networks.py:118-120
            cyc_t = lib.h2d(cyc, torch.int64, dev)
            n_c = torch.zeros(E, dtype=torch.float32, device=dev).index_add_(0, cyc_t, valid)
            xt, dt = ed.transform_rows(m, xg, dg)
            xt = xt * valid.unsqueeze(1)
            dt = dt * (valid * (m["lr"] / n_c.clamp_(min=1.0))[cyc_t]).unsqueeze(1)     # x lr, / n: the running mean over the edit's rows
            slot_t = lib.h2d(slot, torch.int64, dev)
            Xp = torch.zeros((E * npad, xt.shape[1]), dtype=torch.float32, device=dev).index_copy_(0, slot_t, xt)
            Dp = torch.zeros((E * npad, dt.shape[1]), dtype=torch.float32, device=dev).index_copy_(0, slot_t, dt)
            deltas[m["name"]] = (op(Xp).view(E, npad, -1), op(Dp.view(E, npad, -1).transpose(1, 2)).view(E, -1, npad))
            if self.keep_debug:
                self.__dict__.setdefault("debug", {}).setdefault("factors", {})[m["name"]] = (Xp.view(E, npad, -1), Dp.view(E, npad, -1), n_c)
        del caps, save, dx
        evb[2].record()
        # ---- post-edit: the edited layers with every cycle's own deltas, in place on the frozen rows --------------------------------
        x_post = self._edited_layers(psP, deltas, cyc_rows)
        post_logits = eng.lm_head(lib.gather_rows(x_post, ridx))
        post_argmax, _, _ = lib.vocab_rows(post_logits)
        if self.keep_debug:
            dbg = self.__dict__.setdefault("debug", {})
            dbg.update(pre_logits=pre_logits, post_logits=post_logits,
                       rows=[[(p.kind, p.name, p.row0, p.L) for p in plist] for plist in c["probes"]])
        evb[3].record()

        def to_host(t_):
            hbuf = torch.empty(t_.shape, dtype=t_.dtype, device="cpu", pin_memory=True)
            hbuf.copy_(t_, non_blocking=True)
            return hbuf
        host = [to_host(t_) for t_ in (pre_argmax, post_argmax, loss_c)]
        done = torch.cuda.Event()
        done.record()
        return dict(c=c, evb=evb, host=host, done=done, keep=(deltas,))

    def _stage_b_finish(self, h):
        c, evb = h["c"], h["evb"]
        rds, probes, E = c["rds"], c["probes"], c["E"]
        h["done"].synchronize()
        pre_h, post_h, loss_h = (t.numpy() for t in h["host"])
        t5 = time.time()
        eva = c["ev"]
        edit_ms = evb[1].elapsed_time(evb[2])
        self.stats["t_vision"] += eva[0].elapsed_time(eva[1]) * 1e-3
        self.stats["t_decoder"] += eva[1].elapsed_time(eva[2]) * 1e-3
        self.stats["t_edit"] += edit_ms * 1e-3
        self.stats["t_tail"] += (evb[0].elapsed_time(evb[1]) + evb[2].elapsed_time(evb[3])) * 1e-3
        out = self._fill_results(rds, probes, pre_h, post_h, edit_ms * 1e-3 / E)
        meta = [(1, float(loss_h[e])) for e in range(E)]
        self.stats["cycles"] += E
        self.stats["steps"] += E
        self.last_losses = loss_h
        self.stats["t_host"] += time.time() - t5
        return out, meta
