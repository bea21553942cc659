"""BatchedEditEval: many independent edit+eval cycles per GPU, splits sharded across ranks.

One CYCLE = one split of `evaluate_sequential_edit` with edit_n == 1
(R/evaluation/vllm_editor_eval.py:100-123): 9 pre-edit locality probes -> one FT_VL edit
(<= num_steps AdamW steps) -> 12 post-edit probes -> restore.  Cycles are independent (each
starts from pristine weights), so E of them run concurrently:

  1. host: tokenise, label/mask bookkeeping (same rules as prompts_imgs_target_to_xym)
  2. ViT + Q-Former once per UNIQUE image of the batch            (reference: 40 encodes per cycle)
  3. decoder up to the edited layer's fc2 input once per UNIQUE (image, text) sequence
     (reference: 46 full decoder forwards per cycle); only the last-L label rows are kept
  4. pre-edit tail  (fc2 rows -> final LN -> lm_head rows -> argmax) for the locality probes
  5. FT loop, device-side control flow, no host sync: per step  lm_head rows -> NLL/dlogits ->
     dH -> LN backward -> fused rank-L grad + AdamW + next fc2 rows (devqa_ft_adamw_step);
     every in-flight edit owns a private fp32 copy of the edited matrix and its moments
  6. post-edit tail with each edit's own matrix, token accuracies
  7. nothing to restore: the shared original matrix is never written.

Results are identical to the generic path up to fp reassociation (verified in
tests/test_batched_gpu.py against both the generic path and the reference goldens).

Multi-GPU: static contiguous block partition of splits over ranks, no
data-path collective; one gather of per-cycle score rows [n,16] fp32 (RCCL) plus a host-side
gather of the decoded strings for results.json.
"""
import os
import time
from copy import deepcopy
from typing import Dict, List

import numpy as np
import torch

from . import lib

LOC_ORDER = ["text_loc", "t3i3", "t1i4", "t2i4", "t1i2", "t1i3", "t2i1", "t2i2", "t3i1"]
SCORE_COLS = 16  # [sample_id, rel, text_rephrase, image_rephrase, 9 locality, edit_time, steps, final_loss]


def shard_range(n, rank, world):
    """Contiguous block partition: rank r owns [r*n/W, (r+1)*n/W)."""
    return (rank * n) // world, ((rank + 1) * n) // world


def copy_sample(d):
    """Structural copy of a sample dict that keeps leaf objects (image tensors / paths, token lists)
    shared -- the evaluator's deepcopy of result data without cloning device tensors."""
    if isinstance(d, dict):
        return {k: copy_sample(v) for k, v in d.items()}
    if isinstance(d, list) and d and isinstance(d[0], dict):
        return [copy_sample(v) for v in d]
    return d


class _Probe:
    __slots__ = ("kind", "name", "seq", "L", "labels", "mask", "row0", "rd", "ed", "before_ids")


class BatchedEditEval:
    def __init__(self, editor, cycles_per_batch=None):
        self.editor = editor
        self.vllm = editor.vllm
        self.eng = editor.vllm.engine
        if cycles_per_batch is None:
            # large batches keep every 256 x 256 GEMM round full and amortise the launches: BLIP-2 127 cycles = 508 images (bench.py's
            # default, measured; 4-5 GB of activations); the LLaMA-family models carry 576 image tokens per image through a 7B decoder
            # and have only been timed at 8-16 cycles per batch
            cycles_per_batch = int(os.environ.get("DEVQA_CYCLES_PER_BATCH", "127" if type(self.eng).__name__ == "Blip2Engine" else "16"))
        self.E = cycles_per_batch
        self.share_prefix = True  # pack each distinct image-token prefix once (exact; see engine._pack_shared_prefix)
        # parity tests only: keep the last batch's pre-/post-edit label-row logits, the probe -> row map and the compacted deltas
        self.keep_debug = False
        self.stats = {"cycles": 0, "steps": 0, "t_vision": 0.0, "t_decoder": 0.0, "t_ft": 0.0, "t_tail": 0.0,
                      "t_host": 0.0}

    @staticmethod
    def supports(editor, eval_data, edit_n):
        try:
            from .editor.vllm_editors.ft_vl.ft_vl import FTvl
        except Exception:
            return False
        if not (isinstance(editor, FTvl) and hasattr(editor.vllm, "engine") and hasattr(editor.vllm.engine, "pack_from_tokens")):
            return False
        if editor.cfg.batch_size != 1:
            return False
        try:
            editor._edit_target()
        except NotImplementedError:
            return False
        if not all(len(s["requests"]) == 1 for split in eval_data for s in split):
            return False
        if all(len(split) == 1 for split in eval_data):
            return True
        # edit_n > 1 (`-sen N`): the chained form (run -> _run_sequential) needs the device-side FT loop of the path-level context
        eng = editor.vllm.engine
        return hasattr(eng, "path_ctx") and eng.path_ctx() is not None

    # ------------------------------------------------------------------------------------------
    def run(self, result_data, eval_data, gather=True):
        """result_data / eval_data: lists of splits (each one sample).  Returns results[split][0] on
        rank 0 (None on other ranks when torch.distributed is initialised and gather=True)."""
        import torch.distributed as dist
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        rank = dist.get_rank() if world > 1 else 0
        n = len(eval_data)
        lo, hi = shard_range(n, rank, world)
        self.editor.restore_to_original_model()
        if any(len(split) != 1 for split in eval_data):       # `-sen N` with N > 1: chained edits inside a split
            return self._run_sequential(result_data, eval_data, lo, hi, rank, world, gather)
        local, meta = [], []
        batches = []
        for b0 in range(lo, hi, self.E):
            b1 = min(hi, b0 + self.E)
            batches.append(([result_data[i][0] for i in range(b0, b1)], [eval_data[i][0] for i in range(b0, b1)]))
        for out, mt in self.run_batches(batches):
            local.extend(out)
            meta.extend(mt)
        self.last_meta = meta
        self.last_scores = self.score_rows(local, meta, lo)
        if world == 1 or not gather:
            return [[r] for r in local]
        from .dist import gather_results
        allres = gather_results(local, self.last_scores, n, rank, world, self.eng.dev)
        return None if allres is None else [[r] for r in allres]

    # ------------------------------------------------------------------------------------------
    # `-sen N`, N > 1 (R/evaluation/vllm_editor_eval.py:100-123): a split's N edits are applied CUMULATIVELY (ft_vl.py:56-61: every
    # delta is added to the same matrix), then all its samples are probed on the final weights.  Only the edited matrix changes, so
    # the frozen work is the same as for N independent cycles and runs batched (stage A: vision encoder, decoder up to the edited
    # layer's fc2 input, once per unique image / sequence); the chain itself is serial by definition: edit i starts from
    # W0 + sum_{j<i} delta_j -- one device-side FT loop per edit (devqa_ft_edit, E = 1) on its active columns gathered from the RUNNING
    # matrix, its delta scattered back into it, no host synchronisation inside the chain.  Pre-edit probes use the pristine matrix,
    # post-edit probes the final one: two GEMMs over the cached fc2-input rows per chunk.
    # ------------------------------------------------------------------------------------------
    def _run_sequential(self, result_data, eval_data, lo, hi, rank, world, gather):
        local = [self._run_split_chained(result_data[i], eval_data[i]) for i in range(lo, hi)]
        self.last_meta = [mt for _, metas in local for mt in metas]
        local = [res for res, _ in local]
        if world == 1 or not gather:
            return local
        from .dist import gather_split_results
        sizes = [len(sp) for sp in eval_data]
        return gather_split_results(local, sizes, lo, rank, world, self.eng.dev)

    @torch.no_grad()
    def _run_split_chained(self, rds, eds):
        eng, cfg, dev = self.eng, self.editor.cfg, self.eng.dev
        ctx = eng.path_ctx()
        if ctx is None:
            raise RuntimeError("chained FT_VL edits on the batched engine need the path-level context (DEVQA_PATH_ABI)")
        n = len(eds)
        chunks = [self._stage_a(rds[i:i + self.E], eds[i:i + self.E]) for i in range(0, n, self.E)]
        wname = self.editor._edit_target()
        w0 = self.vllm.model.get(wname)                          # fp32 master [d, ffn], never written here
        Wc = w0.clone()                                          # the running matrix of the chain
        Din = w0.shape[1]
        clamp = float(cfg.norm_constraint) if type(cfg.norm_constraint) is float else -1.0
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        per_edit = []
        for c in chunks:
            E, kmax, a_ft, resid_ft = c["E"], c["kmax"], c["a_ft"], c["resid_ft"]
            t_lab = lib.h2d(c["labels"].reshape(-1), torch.int32, dev)
            t_mask = lib.h2d(c["mask"], torch.float32, dev)
            if c.get("act") is not None:
                idx, cnt, mx, act_ev = c["act"]
                act_ev.synchronize()
                npad = max(8, (int(mx[0]) + 7) // 8 * 8)
            else:                                                # weight decay: every column moves
                idx = torch.arange(Din, dtype=torch.int32, device=dev).repeat(E, 1).contiguous()
                cnt = torch.full((E,), Din, dtype=torch.int32, device=dev)
                npad = Din
            for e in range(E):
                ie, ce = idx[e:e + 1], cnt[e:e + 1]
                w_e = lib.gather_cols(Wc, ie, ce, npad, per_edit=False)                     # [1, d, npad] from the RUNNING matrix
                a_e = lib.gather_cols(a_ft[e:e + 1].contiguous(), ie, ce, npad, per_edit=True)
                delta, losses, n_steps, _ = ctx.ft_edit(w_e, a_e, resid_ft[e * kmax:(e + 1) * kmax], t_lab[e * kmax:(e + 1) * kmax],
                                                        t_mask[e:e + 1], cfg.num_steps, cfg.lr, cfg.weight_decay, clamp)
                lib.scatter_cols_add(delta[0], idx[e], ce, Wc)                               # Wc[:, J_e] += delta_e  (ft_vl.py:56-61)
                per_edit.append((n_steps, losses))
        ev1.record()
        w0_op = self.vllm.model.weight_for_gemm(wname)
        wc_op = Wc if eng.adt == torch.float32 else lib.cast_f32_bf16(Wc)
        outs = []
        for c in chunks:
            y_pre = lib.gemm(c["a_tail"], w0_op, c["b2"], residual=c["resid_tail"], want="f32")
            pre_argmax = self._argmax_from_y(y_pre)
            y_post = lib.gemm(c["a_tail"], wc_op, c["b2"], residual=c["resid_tail"], want="f32")
            outs.append((pre_argmax, self._argmax_from_y(y_post)))
        steps_h = torch.cat([s_ for s_, _ in per_edit]).cpu().numpy()
        loss_h = torch.cat([l_ for _, l_ in per_edit]).cpu().numpy()
        torch.cuda.current_stream(dev).synchronize()
        edit_time = ev0.elapsed_time(ev1) * 1e-3 / max(n, 1)
        results = []
        for c, (pre, post) in zip(chunks, outs):
            results += self._fill_results(c["rds"], c["probes"], pre.cpu().numpy(), post.cpu().numpy(), edit_time)
        metas = [(int(steps_h[i]), float(loss_h[i, max(int(steps_h[i]) - 1, 0)])) for i in range(n)]
        self.stats["cycles"] += n
        self.stats["steps"] += int(steps_h.sum())
        self.last_losses, self.last_steps = loss_h, steps_h
        if self.keep_debug:
            self.__dict__.setdefault("debug", {})["chain_weight"] = Wc
        return results, metas

    @staticmethod
    def score_rows(results, meta, first_id):
        """Fixed-width fp32 row per cycle -- the payload of the single RCCL gather."""
        rows = np.zeros((len(results), SCORE_COLS), np.float32)
        for i, (r, mt) in enumerate(zip(results, meta)):
            rr = r["reliability"][0]
            rows[i, 0] = first_id + i
            rows[i, 1] = rr["acc"]
            rows[i, 2] = r["generality"]["text_rephrase"][0]["acc"]
            rows[i, 3] = r["generality"]["image_rephrase"][0]["acc"]
            for j, name in enumerate(LOC_ORDER):
                if name in r["locality"]:
                    rows[i, 4 + j] = r["locality"][name][0]["acc"]
            rows[i, 13] = rr.get("edit_time", 0.0)
            rows[i, 14] = mt[0]
            rows[i, 15] = mt[1]
        return rows

    # ------------------------------------------------------------------------------------------
    def _tok(self, s, has_image=False):
        # wrappers whose text carries an image placeholder (LLaVA: '<image>\n' auto-prefix) tokenise the full string
        if hasattr(self.vllm, "batched_token_ids"):
            return self.vllm.batched_token_ids(s, has_image)
        return self.vllm.tokenizer(s)["input_ids"]

    def _probe_seq(self, prompt, target, has_image=False):
        """xym bookkeeping for one (prompt, target): -> (token ids, labels [L], mask [L]).
        Pre-tokenised inputs (lists of ids; synthetic benchmark data) follow the same rule as
        R/editor/vllms_for_edit/base.py:97-108: labels = roll(ids,-1), mask[len(prompt)-1:-1] = 1,
        both cropped to [len(prompt)-1:]."""
        if isinstance(prompt, (list, tuple)):
            ids = list(prompt) + list(target)
            n_p = len(prompt)
            lab = ids[1:] + ids[:1]
            msk = [0] * len(ids)
            for j in range(n_p - 1, len(ids) - 1):
                msk[j] = 1
            return ids, lab[n_p - 1:], msk[n_p - 1:]
        strs, y, m, _ = self.vllm.xym_token_bookkeeping([prompt], [target])
        return self._tok(strs[0], has_image), y[0].tolist(), m[0].tolist()

    def run_batch(self, rds: List[Dict], eds: List[Dict]):
        """One batch, both stages back to back on the current stream."""
        return self._stage_b(self._stage_a(rds, eds))

    def run_batches(self, batches, pipelined=True, on_iter=None):
        """batches: list of (rds, eds).  Software pipeline: the host runs ONE BATCH AHEAD of the GPU and never waits inside the loop --
        stage A of batch i+1 (host bookkeeping, vision encoder, frozen decoder prefix: the MFMA-heavy part) is queued, then stage B of
        batch i (pre-edit tails, the FT_VL loop, post-edit tails: many small launches), whose results come back through pinned
        buffers and are decoded one iteration later, when they have long arrived.  Same kernels, same results as run_batch per batch.
        Two placements of stage B (DEVQA_PIPELINE):
          serial      (default) on the main stream behind stage A of the next batch: no kernel of B shares the GPU with the GEMMs;
          concurrent  on a side stream, so B's small launches fill the CUs the GEMMs' tail rounds leave idle: +1.5 % cycles/s with the
                      column-compacted FT loop, but -24 % with dense FFN activations (the 22 GB AdamW sweeps and the GEMMs evict each
                      other from L2 / MALL), and every kernel's in-situ time then includes the other stream's share of the GPU."""
        if not pipelined or len(batches) < 2:
            return [self.run_batch(r, e) for r, e in batches]
        side = None
        if os.environ.get("DEVQA_PIPELINE", "serial") == "concurrent":
            side = self.__dict__.get("_side_stream")
            if side is None:
                side = self._side_stream = torch.cuda.Stream(device=self.eng.dev)
        outs = []
        if on_iter is not None:         # measurement hook (bench.py): iteration i queues stage A of batch i + 1 and stage B of batch i = one step's kernels
            on_iter(-1)
        ctx = self._stage_a(*batches[0])
        pending = None
        for i in range(len(batches)):
            if on_iter is not None:
                on_iter(i)
            nxt = self._stage_a(*batches[i + 1]) if i + 1 < len(batches) else None
            handle = self._stage_b(ctx, side, finish=False)
            if pending is not None:
                outs.append(self._stage_b_finish(pending))
            pending, ctx = handle, nxt
        outs.append(self._stage_b_finish(pending))
        return outs

    @torch.no_grad()
    def _stage_a(self, rds: List[Dict], eds: List[Dict]):
        eng, vllm, cfg = self.eng, self.vllm, self.editor.cfg
        dev = eng.dev
        t0 = time.time()
        E = len(eds)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]   # start of vision, end of vision, end of stage A
        ev[0].record()
        # ---- 1. host bookkeeping ----------------------------------------------------------------
        img_index: Dict[str, int] = {}
        img_list = []
        seq_index: Dict[tuple, int] = {}
        seqs = []           # (img idx or None, token ids)

        def img_id(path):
            if path is None:
                return None
            key = path if isinstance(path, str) else ("obj", id(path))
            if key not in img_index:
                img_index[key] = len(img_list)
                img_list.append(path)
            return img_index[key]

        def seq_id(img, ids):
            key = (img, tuple(ids))
            if key not in seq_index:
                seq_index[key] = len(seqs)
                seqs.append((img, ids))
            return seq_index[key]

        # pass 1: unique images in first-use order (locality, request, generality -- the order pass 2 walks), so that
        # the vision encoder can be launched before the token bookkeeping and run under it
        for ed in eds:
            for name in ed["locality"]:
                img_id(ed["locality"][name][0]["image"])
            img_id(ed["requests"][0]["image"])
            for name in ed["generality"]:
                img_id(ed["generality"][name][0]["image"])
        # pixels (host decode, as the reference) -> device
        pix = None
        if img_list:
            if all(isinstance(p, torch.Tensor) for p in img_list):
                if img_list[0].is_cuda:
                    pix = torch.stack(img_list)   # already-preprocessed pixel_values resident in HBM
                else:                             # pre-processed pixel values handed over in host memory: one pinned staging buffer, one H2D copy
                    stage = torch.empty((len(img_list),) + tuple(img_list[0].shape), dtype=torch.float32, pin_memory=True)
                    torch.stack([p.to(torch.float32) for p in img_list], out=stage)
                    pix = stage.to(dev, non_blocking=True)
            else:
                pix = np.stack([vllm.load_pixels(p) for p in img_list])
                pix = torch.from_numpy(pix).to(dev, non_blocking=True)
        t1 = time.time()
        # ---- 2. vision (asynchronous: kernels are queued here, the host goes on with pass 2) -------------------------
        img_tokens = None
        if img_list:
            chunks, i0 = [], 0
            for c in eng.image_chunks(len(img_list)):
                chunks.append(eng.encode_images(pix[i0:i0 + c]))
                i0 += c
            img_tokens = torch.cat(chunks) if len(chunks) > 1 else chunks[0]
        t1b = time.time()
        # pass 2: probes, token ids, labels
        probes: List[List[_Probe]] = []
        edits = []
        for rd, ed in zip(rds, eds):
            rd["reliability"] = rd.pop("requests")
            for r in rd["reliability"]:
                r["target"] = r.pop("target_new")
            plist = []

            def add(kind, name, item_ed, item_rd, target_key):
                ids, y, m = self._probe_seq(item_ed["prompt"], item_ed[target_key], item_ed["image"] is not None)
                p = _Probe()
                p.kind, p.name, p.rd, p.ed = kind, name, item_rd, item_ed
                p.seq = seq_id(img_id(item_ed["image"]), ids)
                p.L, p.labels, p.mask = len(y), y, m
                plist.append(p)
            for name in ed["locality"]:
                add("loc", name, ed["locality"][name][0], rd["locality"][name][0], "target")
            add("rel", None, ed["requests"][0], rd["reliability"][0], "target_new")
            for name in ed["generality"]:
                add("gen", name, ed["generality"][name][0], rd["generality"][name][0], "target")
            probes.append(plist)
            # the edit request (ft_vl.py:72-75: a leading space is forced on the target)
            req = ed["requests"][0]
            tgt = req["target_new"]
            if isinstance(tgt, str) and tgt[0] != " ":
                tgt = " " + tgt
            ids, y, m = self._probe_seq(req["prompt"], tgt, req["image"] is not None)
            rows = [j for j in range(len(y)) if m[j] != 0]
            edits.append((seq_id(img_id(req["image"]), ids), len(y), rows, [y[j] for j in rows]))
        kmax = max(len(e[2]) for e in edits)
        if kmax > lib.FT_MAX_ROWS:
            raise NotImplementedError("batched FT_VL supports <= %d target tokens per edit (got %d)" % (lib.FT_MAX_ROWS, kmax))
        t1c = time.time()
        self.stats["t_host"] += (t1 - t0) + (t1c - t1b)
        ev[1].record()
        # ---- 3. frozen decoder prefix ---------------------------------------------------------------
        ps = eng.pack_from_tokens(seqs, img_tokens, share_prefix=self.share_prefix)
        x_mid, a = eng.decoder_layers(ps, stop_before_fc2=True)
        d = x_mid.shape[1]
        b2 = eng.edit_bias()   # None for LLaMA-family decoders
        wname = self.editor._edit_target()
        w0 = self.vllm.model.get(wname)               # fp32 master [d, ffn]
        w0_op = self.vllm.model.weight_for_gemm(wname)  # GEMM operand (bf16 shadow, or the master in fp32 mode)
        # tail rows of every probe, grouped per cycle (contiguous) so each edit's rows form one GEMM
        row_idx, cyc_rows = [], []
        for plist in probes:
            r0 = len(row_idx)
            for p in plist:
                p.row0 = len(row_idx)
                end = ps.start[p.seq] + ps.length[p.seq]
                row_idx += list(range(end - p.L, end))
            cyc_rows.append((r0, len(row_idx)))
        ridx = lib.h2d(row_idx, torch.int32, dev)     # pinned + non-blocking: a pageable copy would park the host behind the queued vision encoder
        a_tail = lib.gather_rows(a, ridx)                     # operand dtype [R, ffn]
        resid_tail = lib.gather_rows(x_mid, ridx)             # fp32 [R, d]
        # FT rows
        ft_idx = []
        labels = np.zeros((E, kmax), np.int32)
        mask = np.zeros((E, kmax), np.float32)
        for e, (sq, L, rows, labs) in enumerate(edits):
            end = ps.start[sq] + ps.length[sq]
            for j in range(kmax):
                if j < len(rows):
                    ft_idx.append(end - L + rows[j])
                    labels[e, j] = labs[j]
                    mask[e, j] = 1.0
                else:
                    ft_idx.append(end - 1)  # padding row: mask 0 -> coef 0 -> no gradient
        fidx = lib.h2d(ft_idx, torch.int32, dev)
        a_ft = lib.gather_rows(a, fidx).to(torch.float32).view(E, kmax, -1).contiguous()
        a_ft = a_ft * lib.h2d(mask, torch.float32, dev).unsqueeze(-1)   # zero the padding rows (plumbing)
        resid_ft = lib.gather_rows(x_mid, fidx)                      # [E*kmax, d] (+ fc2 bias when the model has one)
        if b2 is not None:
            resid_ft = (resid_ft + b2).contiguous()
        del a, x_mid
        # active columns of every edit (column compaction of the FT loop): counted HERE, with the largest count on its way to a pinned
        # word, so that stage B can size its state without a device -> host read in the middle of the queued work
        act = None
        if cfg.weight_decay == 0:
            idx, cnt = lib.active_columns(a_ft)
            mx = torch.empty((1,), dtype=torch.int32, device="cpu", pin_memory=True)
            mx.copy_(cnt.max().reshape(1), non_blocking=True)
            act_ev = torch.cuda.Event()
            act_ev.record()
            act = (idx, cnt, mx, act_ev)
        ev[2].record()
        return dict(rds=rds, probes=probes, E=E, kmax=kmax, d=d, b2=b2, w0=w0, w0_op=w0_op, cyc_rows=cyc_rows, a_tail=a_tail,
                    resid_tail=resid_tail, a_ft=a_ft, resid_ft=resid_ft, labels=labels, mask=mask, ev=ev, act=act)

    @torch.no_grad()
    def _stage_b(self, c, stream=None, finish=True):
        """Stage B on `stream` (None: the current stream).  The tensors of stage A stay referenced by the returned handle until its
        results have been collected, so the caching allocator cannot hand them out early.  finish=False: returns the handle for
        _stage_b_finish instead of waiting for the results."""
        if stream is None:
            h = self._stage_b_body(c)
        else:
            stream.wait_event(c["ev"][2])
            with torch.cuda.stream(stream):
                h = self._stage_b_body(c)
        return self._stage_b_finish(h) if finish else h

    def _stage_b_body(self, c):
        eng, vllm, cfg = self.eng, self.vllm, self.editor.cfg
        rds, probes, E, kmax, d, b2, w0, w0_op = c["rds"], c["probes"], c["E"], c["kmax"], c["d"], c["b2"], c["w0"], c["w0_op"]
        cyc_rows, a_tail, resid_tail, a_ft, resid_ft, labels, mask = (c["cyc_rows"], c["a_tail"], c["resid_tail"], c["a_ft"],
                                                                      c["resid_ft"], c["labels"], c["mask"])
        evb = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        evb[0].record()
        # ---- 4. pre-edit tail (all probes share W0) ----------------------------------------------------
        y_pre = lib.gemm(a_tail, w0_op, b2, residual=resid_tail, want="f32")  # fc2 rows of every probe, pristine W
        pre_argmax = self._argmax_from_y(y_pre, "pre")
        # ---- 5. FT loop (on the active columns of each edit) --------------------------------------------
        evb[1].record()
        n_steps, losses, delta_c, idx, cnt, npad = self._ft_loop(w0, a_ft, resid_ft, labels, mask, E, kmax, d, cfg, c.get("act"))
        evb[2].record()
        # ---- 6. post-edit tail:  y_post = y_pre + (dW_e restricted to its active columns) . a -------------
        delta_op = delta_c if eng.adt == torch.float32 else lib.cast_f32_bf16(delta_c)  # [E, d, npad]
        y_post = y_pre  # updated in place, cycle by cycle
        for e, (r0, r1) in enumerate(cyc_rows):
            a_pc = lib.gather_cols(a_tail[r0:r1], idx[e:e + 1], cnt[e:e + 1], npad, per_edit=False)[0]
            lib.gemm(a_pc, delta_op[e], residual=y_post[r0:r1], out_f32=y_post[r0:r1])
        post_argmax = self._argmax_from_y(y_post, "post")
        if self.keep_debug:
            self.debug["delta"] = (delta_c, idx, cnt, npad)
            self.debug["rows"] = [[(p.kind, p.name, p.row0, p.L) for p in plist] for plist in probes]
        evb[3].record()

        def to_host(t):      # asynchronous device -> pinned host copy on this stream
            hbuf = torch.empty(t.shape, dtype=t.dtype, device="cpu", pin_memory=True)
            hbuf.copy_(t, non_blocking=True)
            return hbuf
        host = [to_host(t) for t in (pre_argmax, post_argmax, n_steps, losses, self._adam_t)]
        done = torch.cuda.Event()
        done.record()
        return dict(c=c, evb=evb, host=host, done=done, ft_shape=self._ft_shape, keep=(y_pre, delta_c, idx, cnt))

    def _stage_b_finish(self, h):
        """Waits for stage B's results (they have usually arrived long ago) and fills the result records (host work)."""
        vllm = self.vllm
        c, evb = h["c"], h["evb"]
        rds, probes, E = c["rds"], c["probes"], c["E"]
        h["done"].synchronize()
        pre_h, post_h, steps_h, losses_h, upd_h = (t.numpy() for t in h["host"])
        t5 = time.time()
        eva = c["ev"]
        ft_ms = evb[1].elapsed_time(evb[2])
        self.stats["t_vision"] += eva[0].elapsed_time(eva[1]) * 1e-3     # GPU time per phase (streams may overlap)
        self.stats["t_decoder"] += eva[1].elapsed_time(eva[2]) * 1e-3
        self.stats["t_ft"] += ft_ms * 1e-3
        self.stats["t_tail"] += (evb[0].elapsed_time(evb[1]) + evb[2].elapsed_time(evb[3])) * 1e-3
        # ---- 7. host: results -----------------------------------------------------------------------------
        edit_time = ft_ms * 1e-3 / E
        out = self._fill_results(rds, probes, pre_h, post_h, edit_time)
        meta = [(int(steps_h[e]), float(losses_h[e, max(int(steps_h[e]) - 1, 0)])) for e in range(E)]
        self.stats["cycles"] += E
        self.stats["steps"] += int(steps_h.sum())
        # algorithmic HBM bytes of the executed AdamW updates (include/devqa.h, devqa_ft_adamw_step_fm), fp32 on [Dout, npad]: the first update of an
        # edit reads w0 and writes w, v (3 tensors), every later one reads and writes w, v (4); an edit with ONE loss row has no v matrix either:
        # 2 tensors per update (read w0 / w, write w)
        Dout_, npad_ = h["ft_shape"]
        n_upd = int(upd_h.sum())
        self.stats["updates"] = self.stats.get("updates", 0) + n_upd
        one = np.asarray(c["mask"]).sum(1) == 1
        tensors = int(2 * upd_h[one].sum()) + int(4 * upd_h[~one].sum() - (upd_h[~one] > 0).sum())
        self.stats["ft_bytes"] = self.stats.get("ft_bytes", 0) + 4 * Dout_ * npad_ * tensors
        self.stats["one_row_edits"] = self.stats.get("one_row_edits", 0) + int(one.sum())
        self.stats["one_row_updates"] = self.stats.get("one_row_updates", 0) + int(upd_h[one].sum())
        self.last_losses = losses_h
        self.last_steps = steps_h
        self.stats["t_host"] += time.time() - t5
        return out, meta

    def _fill_results(self, rds, probes, pre_h, post_h, edit_time):
        """Result records of one batch from the host copies of the label-row argmax (R/evaluation/vllm_editor_eval.py:137-175):
        locality probes are scored against the PRE-edit argmax, the others against the gold labels."""
        tok = self.vllm.tokenizer
        out = []
        for rd, plist in zip(rds, probes):
            for p in plist:
                m = np.asarray(p.mask) != 0
                post = post_h[p.row0:p.row0 + p.L]
                if p.kind == "loc":
                    pre = pre_h[p.row0:p.row0 + p.L]
                    p.rd["predict_before_edit"] = tok.decode(torch.from_numpy(pre[m].astype(np.int64)))
                    ref = pre
                else:
                    ref = np.asarray(p.labels)
                if p.kind == "rel":
                    p.rd["edit_time"] = edit_time
                p.rd["predict_after_edit"] = tok.decode(torch.from_numpy(post[m].astype(np.int64)))
                p.rd["acc"] = float(np.float32(((post == ref) & m).sum()) / np.float32(m.sum()))
            out.append(rd)
        return out

    # ------------------------------------------------------------------------------------------
    def _argmax_from_y(self, y, tag=None):
        logits = self.eng.lm_head(y)
        am, _, _ = lib.vocab_rows(logits)
        if self.keep_debug and tag is not None:
            self.__dict__.setdefault("debug", {})[tag + "_logits"] = logits
        return am

    def _ft_loop(self, w0, a_ft, resid_ft, labels, mask, E, kmax, d, cfg, act=None):
        """Device-side FT_VL loop for E concurrent edits.  Returns (n_steps, losses, delta_c, idx, cnt, npad):
        delta_c fp32 [E, Dout, npad] is each edit's weight delta on its ACTIVE columns idx[e,:cnt[e]]
        (exactly zero elsewhere; csrc/ft_compact.hip).  With weight decay every column moves, so the
        compaction is bypassed (idx = all columns)."""
        eng = self.eng
        dev = eng.dev
        Dout, Din = w0.shape
        npad = Din
        if cfg.weight_decay == 0:
            if act is not None:        # counted at the end of stage A; its event has normally completed long ago
                idx, cnt, mx, act_ev = act
                act_ev.synchronize()
                npad = max(8, (int(mx[0]) + 7) // 8 * 8)
            else:
                idx, cnt = lib.active_columns(a_ft)
                npad = max(8, (int(cnt.max().item()) + 7) // 8 * 8)   # one small sync per batch: sizes the state
        dense = npad * 10 >= Din * 9      # (nearly) every column is active somewhere: gathering would only copy the matrix E times
        if dense:
            idx = torch.arange(Din, dtype=torch.int32, device=dev).repeat(E, 1).contiguous()
            cnt = torch.full((E,), Din, dtype=torch.int32, device=dev)
            npad = Din
        else:
            w0 = lib.gather_cols(w0, idx, cnt, npad, per_edit=False)        # [E, Dout, npad] pristine columns
            a_ft = lib.gather_cols(a_ft, idx, cnt, npad, per_edit=True)      # [E, kmax, npad]
        self.stats["npad_sum"] = self.stats.get("npad_sum", 0) + npad * E
        t_lab = lib.h2d(labels.reshape(-1), torch.int32, dev)
        t_mask = lib.h2d(mask, torch.float32, dev)
        clamp = float(cfg.norm_constraint) if type(cfg.norm_constraint) is float else -1.0
        ctx = eng.path_ctx() if hasattr(eng, "path_ctx") else None
        if ctx is not None:     # the same loop behind the C ABI (devqa_ft_edit): one call, no Python between the steps
            delta, losses, n_steps, adam_t = ctx.ft_edit(w0, a_ft, resid_ft, t_lab, t_mask, cfg.num_steps, cfg.lr, cfg.weight_decay, clamp)
            self._adam_t = adam_t
            self._ft_shape = (Dout, npad)
            return n_steps, losses, delta, idx, cnt, npad
        coef = (t_mask / t_mask.sum(1, keepdim=True)).reshape(-1).contiguous()
        active = torch.ones(E, dtype=torch.int32, device=dev)
        do_update = torch.zeros(E, dtype=torch.int32, device=dev)
        n_steps = torch.zeros(E, dtype=torch.int32, device=dev)
        adam_t = torch.zeros(E, dtype=torch.int32, device=dev)
        losses = torch.zeros((E, cfg.num_steps), dtype=torch.float32, device=dev)
        # engines without a path-level context (LLaVA / MiniGPT-4 decoders): the same loop, launched from here
        w = torch.empty((E, Dout, npad), dtype=torch.float32, device=dev) if dense else w0.clone()   # dense: the first update reads w0
        factored = os.environ.get("DEVQA_FT_FACTORED", "1") != "0"
        # EMA of dy: first moment = dstate^T (x) a_ft (ft_adamw_step_fm); DEVQA_FT_FACTORED=0: the first-moment matrix
        dstate = torch.empty((E, kmax + 1, Dout), dtype=torch.float32, device=dev) if factored else torch.empty_like(w)
        # edits with ONE loss row (label slots are filled from 0): their second moment factors too, v is not touched (devqa_ft_adamw_step_fm)
        single = lib.h2d((np.asarray(mask).sum(1) == 1).astype(np.int32), torch.int32, dev) if factored else None
        var = torch.empty_like(w)
        y = lib.rows_matvec(w0, a_ft)  # step-0 fc2 rows with the pristine matrix (active columns carry all of W.a)
        dl_dtype = eng.adt
        for it in range(cfg.num_steps):
            y2 = y.view(E * kmax, Dout)
            logits = eng.lm_head(y2, add=resid_ft)
            _, nll, dlog = lib.vocab_rows(logits, t_lab, coef, want_argmax=False, want_nll=True, want_dlogits=True,
                                          dlogits_dtype=dl_dtype)
            lib.ft_step_control(nll, t_mask, it, cfg.num_steps, 1e-2, active, do_update, n_steps, adam_t, losses)
            dH = lib.gemm_rows_longk(dlog, self.vllm.model.embed_T)
            dy = eng.final_norm_bwd(y2, dH, add=resid_ft).view(E, kmax, Dout)
            if factored:
                lib.ft_adamw_step_fm(w, dstate, var, w0, a_ft, dy, y, do_update, adam_t, cfg.lr, 0.9, 0.999, 1e-8, cfg.weight_decay, clamp, single=single)
            else:
                lib.ft_adamw_step(w, dstate, var, w0, a_ft, dy, y, do_update, adam_t, cfg.lr, 0.9, 0.999, 1e-8, cfg.weight_decay, clamp)
        self._adam_t = adam_t
        self._ft_shape = (Dout, npad)
        delta = var           # (the second-moment buffer is free now)
        if dense:   # edits that never updated (first loss already under the floor) have an unwritten w: their delta is zero
            delta.zero_()
            for e in torch.nonzero(adam_t > 0).flatten().tolist():
                lib.delta_op(0, w[e], w0, delta[e])
        else:
            lib.delta_op(0, w, w0, delta)  # delta := w - w0: the compacted delta
        return n_steps, losses, delta, idx, cnt, npad
