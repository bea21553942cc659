"""VLLMEditorEvaluation: drop-in for R/evaluation/vllm_editor_eval.py:13-247 (same constructor,
methods, result schema and JSON files).

`evaluate_sequential_edit` has two execution modes with identical results:
  * generic (any VLLMBaseEditor / BaseVLLMForEdit): the reference's call sequence through the
    plugin API (prepare -> edit -> test -> restore per split);
  * batched (FTvl on the native BLIP-2 wrapper, edit_n == 1, every sample a single request):
    devqa_amd.batched.BatchedEditEval runs many independent splits concurrently on one GPU and
    shards splits across ranks; selected automatically, `batched=False` disables.
"""
import json
import os
from collections import defaultdict
from copy import deepcopy
from datetime import datetime
from time import time
from typing import Dict, List

import numpy as np
import torch


class VLLMEditorEvaluation:
    def __init__(self, editor, eval_data, evaluation_name=None, results_dir="eval_results") -> None:
        self.editor = editor
        self.eval_data = eval_data
        editor_name, model_name = editor.name_of_editor_and_model()
        t = datetime.now().strftime("%Y.%m.%d-%H.%M.%S")
        evaluation_name = evaluation_name if evaluation_name else t
        self.result_dir = os.path.join(results_dir, editor_name, model_name, evaluation_name)
        print("Evaluation results directory: ", self.result_dir)

    # -- helpers shared by both evaluation entry points (vllm_editor_eval.py:137-175) -----------
    @staticmethod
    def _argmax_last(vllm, prompt, image, target):
        (x, vt), y, m = vllm.prompts_imgs_target_to_xym([prompt], [image], [target])
        x["query_triple"] = (prompt, image, target)                       # dynamic-eval hook (:140)
        x["query_range"] = (0, x["inputs_embeds"].shape[1] - m.shape[1] + 1)  # (:141)
        logits = vllm.get_llm_outpt(x, vt).logits
        assert len(y) == 1 and len(m) == 1
        from .. import lib
        L = y.shape[1]
        rows = logits[0, -L:].to(torch.float32).contiguous()
        pre, _, _ = lib.vocab_rows(rows)  # argmax(softmax(.)) == argmax(.)
        return pre.to(torch.long).unsqueeze(0), y, m

    @staticmethod
    def _plan_shared_prefixes(items, min_share=32):
        """items: [(i, e, y, m, keys)].  Probes whose inputs start with the same rows (same image tokens, same in-context text:
        IKE_VL puts the 32 retrieved demonstrations in front of every probe, LTE_VL the retrieved edit) form a group whose common
        prefix is computed ONCE: attention is causal, so the hidden states of a prefix do not depend on what follows it.
        -> ([(member positions, prefix length)], [positions packed on their own]).  A prefix never reaches into a member's label
        window (the last L rows), and identities come from the wrappers' `row_keys` (none -> no sharing for that probe)."""
        buckets, alone = {}, []
        for pos, (_, e, y, _m, keys) in enumerate(items):
            room = e.shape[0] - y.shape[1]
            if keys is None or len(keys) != e.shape[0] or room < min_share:
                alone.append(pos)
            else:
                buckets.setdefault(tuple(keys[:min_share]), []).append(pos)
        groups = []
        for members in buckets.values():
            if len(members) < 2:
                alone += members
                continue
            ks = [items[p_][4] for p_ in members]
            lcp = min(items[p_][1].shape[0] - items[p_][2].shape[1] for p_ in members)
            first = ks[0]
            for k_ in ks[1:]:
                n = 0
                while n < lcp and k_[n] == first[n]:
                    n += 1
                lcp = n
            if lcp >= min_share:
                groups.append((members, lcp))
            else:
                alone += members
        return groups, sorted(alone)

    @staticmethod
    def _prefix_row_keys(pfx):
        """Row identities of a retrieved prefix (LTE_VL's stored edit rows).  The key must be stable per STORED EDIT: the
        editor hands out a fresh view object of the pool entry on every call, so `id(view)` is both different for the same
        edit and -- once the temporary is freed and CPython reuses the address -- possibly equal for different edits.  The
        view's device address and extent identify the pool entry for as long as the pool holds it (the pool does not change
        inside an evaluation phase)."""
        ident = (int(pfx.data_ptr()), tuple(pfx.shape), tuple(pfx.stride()))
        return [("pfx", ident, j) for j in range(pfx.shape[0])]

    @staticmethod
    def _argmax_many(vllm, probes, max_rows=12288, prefix_fn=None, defer=False):
        """[(prompt, image, target)] -> [(pre, y, m)], same values as _argmax_last per probe, but the probes of one
        evaluation phase (the model does not change inside a phase: 9 locality probes per sample before the edit, 12
        after, vllm_editor_eval.py:98-121) go through the decoder TOGETHER: inputs are built per probe through the
        plugin API (so editor hooks on get_llm_input_embeds still apply), packed without padding rows -- common prefixes once,
        see _plan_shared_prefixes -- the decoder runs once per <= max_rows rows and logits are computed on the label rows
        only.  `prefix_fn(prompt, image, target)` (retrieval editors: LTE_VL's `probe_prefix`) may return rows to put in front
        of a probe's input; label rows are the LAST L, so nothing else moves.
        defer=True: everything is QUEUED (the argmax rows travel to pinned host memory behind an event) and a zero-argument function is
        returned that waits for the event and yields the list -- the caller prepares its next phase on the host meanwhile."""
        from .. import lib
        eng = vllm.engine
        out = [None] * len(probes)
        items = []
        waiting = []
        share = os.environ.get("DEVQA_PROBE_PREFIX_SHARE", "1") != "0"
        if hasattr(vllm, "image_features"):   # one batched encoder call for the phase's distinct images (fills the cache)
            uniq = {}
            for _, img, _ in probes:
                k = vllm.image_key(img) if hasattr(vllm, "image_key") else (img if isinstance(img, str) else None)
                if k is not None:
                    uniq.setdefault(k, img)
            if uniq:
                vllm.image_features(list(uniq.values()))
        for i, (prompt, image, target) in enumerate(probes):
            (x, vt), y, m = vllm.prompts_imgs_target_to_xym([prompt], [image], [target])
            assert len(y) == 1 and len(m) == 1
            e = x["inputs_embeds"][0]
            keys = x.get("row_keys") if share else None
            pfx = prefix_fn(prompt, image, target) if prefix_fn is not None else None
            if pfx is not None:
                e = torch.cat([pfx.to(e.dtype), e], 0)
                if keys is not None:
                    keys = VLLMEditorEvaluation._prefix_row_keys(pfx) + list(keys)
            items.append((i, e, y, m, keys))
        groups, alone = VLLMEditorEvaluation._plan_shared_prefixes(items) if share else ([], list(range(len(items))))
        units = [(members, lcp) for members, lcp in groups] + [([p_], 0) for p_ in alone]   # (member positions, shared prefix rows)
        start = 0
        while start < len(units):
            end, total = start, 0
            while end < len(units):
                members, lcp = units[end]
                need = lcp + sum(items[p_][1].shape[0] - lcp for p_ in members)
                if end > start and total + need > max_rows:
                    break
                total += need
                end += 1
            parts, pos, desc, label_rows, order, max_len, r = [], [], [], [], [], 1, 0
            for members, lcp in units[start:end]:
                pstart = r
                if lcp:
                    parts.append(items[members[0]][1][:lcp])
                    pos += list(range(lcp))
                    desc.append([r, lcp, 0, 0, r, lcp])
                    max_len = max(max_len, lcp)
                    r += lcp
                for p_ in members:
                    _, e, y, _m, _k = items[p_]
                    S, L = e.shape[0] - lcp, y.shape[1]
                    parts.append(e[lcp:])
                    pos += list(range(lcp, lcp + S))
                    desc.append([r, S, pstart, lcp, r, S] if lcp else [r, S, 0, 0, r, S])
                    label_rows += [r + S - L + j for j in range(L)]
                    order.append(p_)
                    max_len = max(max_len, S)
                    r += S
            rows = torch.cat(parts, 0)
            ps = eng.pack_rows(rows, pos, desc, max_len)
            x_fin, _ = eng.decoder_layers(ps)
            idx = lib.h2d(label_rows, torch.int32, rows.device)
            logits = eng.lm_head(lib.gather_rows(x_fin, idx))
            pre, _, _ = lib.vocab_rows(logits)
            if defer:
                pre_dev = pre.to(torch.long)
                pre_host = torch.empty(pre_dev.shape, dtype=torch.long, pin_memory=True)
                pre_host.copy_(pre_dev, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                waiting.append((ev, pre_host, pre_dev, order))
            else:
                pre = pre.to(torch.long).cpu()      # ONE device -> host transfer for the whole phase: decoding and accuracies are host work
                VLLMEditorEvaluation._scatter_pre(out, items, order, pre)
            start = end
        if not defer:
            return out

        def finish():
            for ev, pre_host, _pre_dev, order in waiting:
                ev.synchronize()
                VLLMEditorEvaluation._scatter_pre(out, items, order, pre_host)
            return out
        return finish

    @staticmethod
    def _scatter_pre(out, items, order, pre):
        r0 = 0
        for p_ in order:
            i, _, y, m, _k = items[p_]
            L = y.shape[1]
            out[i] = (pre[r0:r0 + L].unsqueeze(0), VLLMEditorEvaluation._host(y), VLLMEditorEvaluation._host(m))
            r0 += L

    @staticmethod
    def _host(t):
        """CPU copy of a label tensor: the host original the wrapper attached (no synchronisation), else a transfer"""
        h = getattr(t, "_devqa_host", None)
        return h if h is not None else (t.cpu() if t.is_cuda else t)

    @staticmethod
    def _can_batch_probes(editor):
        """Probe batching needs the native engine interface and an editor that does not read the per-probe
        `query_triple` / `query_range` keys inside get_llm_outpt -- unless it offers the same decision as `probe_prefix`
        (LTE_VL does)."""
        eng = getattr(editor.vllm, "engine", None)
        return (eng is not None and hasattr(eng, "pack_rows") and hasattr(eng, "lm_head")
                and (not getattr(editor, "reads_query_hook", False) or hasattr(editor, "probe_prefix")))

    @staticmethod
    def _acc(pre_y, label_ids, label_masks):
        if not pre_y.is_cuda:       # batched probe path: predictions are on the host already
            label_ids = label_ids.cpu() if label_ids.is_cuda else label_ids
            label_masks = label_masks.cpu() if label_masks.is_cuda else label_masks
        return float(((pre_y == label_ids) * label_masks).sum() / label_masks.sum())

    def __get_results_after_edit__(self, vllm, ed, rd, batch_probes=False, prefix_fn=None, defer=False):
        """defer (batched probes only): the probes are queued and a function is returned that completes `rd` once they have run."""
        tok = vllm.get_llm_tokenizer()
        if batch_probes:
            probes = [(e["prompt"], e["image"], e["target_new"]) for e in ed["requests"]]
            probes += [(e["prompt"], e["image"], e["target"]) for g in ed["generality"] for e in ed["generality"][g]]
            probes += [(e["prompt"], e["image"], e["target"]) for l in ed["locality"] for e in ed["locality"][l]]
            got = self._argmax_many(vllm, probes, prefix_fn=prefix_fn, defer=defer)

            def fill():
                res = iter(got() if defer else got)
                for rdr in rd["reliability"]:
                    pre, y, m = next(res)
                    rdr["predict_after_edit"] = tok.decode(pre[m.to(bool)])
                    rdr["acc"] = self._acc(pre, y, m)
                for gen_name in ed["generality"]:
                    for rdg in rd["generality"][gen_name]:
                        pre, y, m = next(res)
                        rdg["predict_after_edit"] = tok.decode(pre[m.to(bool)])
                        rdg["acc"] = self._acc(pre, y, m)
                for loc_name in ed["locality"]:
                    for rdl, edl in zip(rd["locality"][loc_name], ed["locality"][loc_name]):
                        pre, _, m = next(res)
                        rdl["predict_after_edit"] = tok.decode(pre[m.to(bool)])
                        rdl["acc"] = self._acc(pre, edl["before_edit_ids"], m)
                return rd
            return fill if defer else fill()
        for rdr, edr in zip(rd["reliability"], ed["requests"]):
            pre, y, m = self._argmax_last(vllm, edr["prompt"], edr["image"], edr["target_new"])
            rdr["predict_after_edit"] = tok.decode(pre[m.to(bool)])
            rdr["acc"] = self._acc(pre, y, m)
        for gen_name in ed["generality"]:
            for rdg, edg in zip(rd["generality"][gen_name], ed["generality"][gen_name]):
                pre, y, m = self._argmax_last(vllm, edg["prompt"], edg["image"], edg["target"])
                rdg["predict_after_edit"] = tok.decode(pre[m.to(bool)])
                rdg["acc"] = self._acc(pre, y, m)
        for loc_name in ed["locality"]:
            for rdl, edl in zip(rd["locality"][loc_name], ed["locality"][loc_name]):
                pre, _, m = self._argmax_last(vllm, edl["prompt"], edl["image"], edl["target"])
                rdl["predict_after_edit"] = tok.decode(pre[m.to(bool)])
                rdl["acc"] = self._acc(pre, edl["before_edit_ids"], m)  # agreement with PRE-edit argmax (:170-174)
        return rd

    # -- vllm_editor_eval.py:29-67 ------------------------------------------------------------------
    def evaluate_single_edit(self):
        editor = self.editor
        print("Evaluating reliability, generality and locality for %s on %s with single editing."
              % editor.name_of_editor_and_model())
        eval_data = deepcopy(self.eval_data.data_with_img)
        for ed in eval_data:
            assert len(ed["requests"]) == 1
        result_data = deepcopy(self.eval_data.data_with_img_path)
        tok = editor.vllm.get_llm_tokenizer()
        editor.restore_to_original_model()
        results = []
        for rd, ed in zip(result_data, eval_data):
            rd["reliability"] = rd.pop("requests")
            rd["reliability"][0]["target"] = rd["reliability"][0].pop("target_new")
            for loc_name in ed["locality"].keys():
                for rdl, edl in zip(rd["locality"][loc_name], ed["locality"][loc_name]):
                    pre, y, m = self._argmax_last(editor.vllm, edl["prompt"], edl["image"], edl["target"])
                    rdl["predict_before_edit"] = tok.decode(y[m.to(bool)])  # decodes the LABELS (quirk, :50)
                    edl["before_edit_ids"] = pre
            start_t = time()
            editor.edit_one_piece(ed["requests"][0])
            rd["reliability"][0]["edit_time"] = time() - start_t
            rd = self.__get_results_after_edit__(editor.vllm, ed, rd)
            results.append(rd)
            editor.restore_to_original_model()
        save_dir = os.path.join(self.result_dir, "single_edit")
        self.save_results(os.path.join(save_dir, "results.json"), results)
        mean_results = self.get_mean_results(results)
        mean_results["sample_count"] = len(results)
        self.save_results(os.path.join(save_dir, "mean_results.json"), mean_results)
        return results

    # -- vllm_editor_eval.py:69-135 -----------------------------------------------------------------
    @staticmethod
    def split_data(data, edit_n):
        """Greedy groups with >= edit_n requests; an incomplete tail group is DROPPED (:74-87)."""
        splits, ns, cur, n = [], [], [], 0
        for d in data:
            cur.append(d)
            n += len(d["requests"])
            if n >= edit_n:
                splits.append(cur)
                ns.append(n)
                cur, n = [], 0
        return splits, ns

    def evaluate_sequential_edit(self, edit_n=10, random=False, seed=None, batched=None, save=True):
        editor = self.editor
        print("Evaluating reliability, generality and locality for %s on %s with sequential editing %s."
              % (*editor.name_of_editor_and_model(), edit_n))
        eval_data = deepcopy(self.eval_data.data_with_img)
        result_data = deepcopy(self.eval_data.data_with_img_path)
        if random:
            seed = seed if seed is not None else np.random.randint(1, 999999)
            np.random.default_rng(seed).shuffle(eval_data)
            np.random.default_rng(seed).shuffle(result_data)
        eval_data, eval_data_ns = self.split_data(eval_data, edit_n)
        result_data, _ = self.split_data(result_data, edit_n)
        from ..batched import BatchedEditEval
        from ..batched_mend import BatchedMendEval
        # fully batched engines: FT_VL (batched.py) and MEND_VL (batched_mend.py); every other editor / shape of data runs the
        # reference's call sequence per split (_sequential_generic)
        engine_cls = next((cls for cls in (BatchedEditEval, BatchedMendEval) if cls.supports(editor, eval_data, edit_n)), None)
        if batched is not None:
            if not batched:
                engine_cls = None
            elif engine_cls is None:
                engine_cls = BatchedMendEval if type(editor).__name__ == "MENDvl" else BatchedEditEval
        self.last_mode = "generic" if engine_cls is None else engine_cls.__name__
        if engine_cls is not None:
            results = engine_cls(editor).run(result_data, eval_data)
        else:
            results = self._sequential_generic(editor, result_data, eval_data)
        if results is None:   # non-zero rank of a sharded run: rank 0 owns the files
            return None
        if save:
            save_dir = os.path.join(self.result_dir, "sequential_edit_%s" % edit_n)
            pre = "seed_%s_" % seed if random else ""
            self.save_results(os.path.join(save_dir, "%sresults.json" % pre), results)
            split_mean = [self.get_mean_results(sr) for sr in results]
            for mr, n in zip(split_mean, eval_data_ns):
                mr["sequential_edit_n"] = n
            total_mean = self.get_mean_results([r for sr in results for r in sr])
            total_mean["total_edit_n"] = sum(eval_data_ns)
            mean_results = {"total_mean": total_mean, "split_mean": split_mean}
            self.save_results(os.path.join(save_dir, "%smean_results.json" % pre), mean_results)
        return results

    def _sequential_generic(self, editor, result_data, eval_data):
        """The reference's call sequence per split.  Splits are independent (every split starts from restored weights,
        vllm_editor_eval.py:98,122), so under torch.distributed rank r runs the contiguous block of splits
        [r*S/W, (r+1)*S/W) and rank 0 gathers: one collective of fixed-width score rows + a host gather of the result
        dicts; other ranks return None."""
        import torch.distributed as dist
        from ..batched import BatchedEditEval, shard_range
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        rank = dist.get_rank() if world > 1 else 0
        n_splits = len(eval_data)
        lo, hi = shard_range(n_splits, rank, world)
        local = self._run_splits(editor, result_data[lo:hi], eval_data[lo:hi])
        if world == 1:
            return local
        # the gather helpers move one row / one dict per SAMPLE: flatten the splits, regroup on rank 0
        from ..dist import gather_split_results
        dev = editor.device if isinstance(editor.device, str) else "cuda:%d" % editor.device
        return gather_split_results(local, [len(sp) for sp in result_data], lo, rank, world, dev)

    def _run_splits(self, editor, result_data, eval_data):
        """prepare -> edit -> test -> restore per split (vllm_editor_eval.py:94-123).  Every split starts from the restored
        model, so the pre-edit locality probes of a GROUP of consecutive splits see the same (pristine) model: with the
        batched-probe path they go through the decoder together, ahead of the group's edits."""
        tok = editor.vllm.get_llm_tokenizer()
        bp = self._can_batch_probes(editor) and os.environ.get("DEVQA_PROBE_BATCH", "1") != "0"
        group = max(1, int(os.environ.get("DEVQA_PREEDIT_GROUP", "16"))) if bp else 1
        pfx = getattr(editor, "probe_prefix", None) if bp else None
        editor.restore_to_original_model()
        results = []
        # batched probes: the host runs ONE SPLIT AHEAD of the GPU -- a split's post-edit probes are queued, the next split's edit and
        # inputs are prepared (tokenisation, prompt assembly: tens of ms per split for the in-context editors) while they run, and only then
        # are the queued split's argmax rows read back and decoded.  Same kernels in the same stream order, same results.
        look_ahead = bp and os.environ.get("DEVQA_EVAL_LOOKAHEAD", "1") != "0"
        pending = None           # (fill functions of the queued split, its result list)

        def flush():
            nonlocal pending
            if pending is not None:
                fills, split_res = pending
                split_res.extend(f() for f in fills)
                pending = None
        for g0 in range(0, len(eval_data), group):
            grp = list(zip(result_data[g0:g0 + group], eval_data[g0:g0 + group]))
            pairs = []
            if bp and hasattr(editor.vllm, "image_features") and hasattr(editor.vllm, "image_key"):
                # every image the group's probes will show (pre- AND post-edit: the rephrase image is first seen after the edit) through
                # the vision tower in ONE call -- per split it would be a one-image call: ~160 launches of a 257-row ViT
                uniq = {}
                for _, split_ed in grp:
                    for ed in split_ed:
                        entries = list(ed["requests"]) + [e for g in ed["generality"] for e in ed["generality"][g]]
                        entries += [e for ln in ed["locality"] for e in ed["locality"][ln]]
                        for e in entries:
                            k = editor.vllm.image_key(e.get("image"))
                            if k is not None:
                                uniq.setdefault(k, e["image"])
                if 1 < len(uniq) <= 128:       # (the wrapper's cache holds 160 images)
                    editor.vllm.image_features(list(uniq.values()))
            for split_rd, split_ed in grp:
                for rd, ed in zip(split_rd, split_ed):
                    rd["reliability"] = rd.pop("requests")
                    for r in rd["reliability"]:
                        r["target"] = r.pop("target_new")
                    if bp:
                        pairs += [(rdl, edl) for ln in ed["locality"] for rdl, edl in zip(rd["locality"][ln], ed["locality"][ln])]
                        continue
                    for loc_name in ed["locality"].keys():
                        for rdl, edl in zip(rd["locality"][loc_name], ed["locality"][loc_name]):
                            pre, _, m = self._argmax_last(editor.vllm, edl["prompt"], edl["image"], edl["target"])
                            rdl["predict_before_edit"] = tok.decode(pre[m.to(bool)])
                            edl["before_edit_ids"] = pre
            if pairs:
                outs = self._argmax_many(editor.vllm, [(e["prompt"], e["image"], e["target"]) for _, e in pairs], prefix_fn=pfx)
                for (rdl, edl), (pre, _, m) in zip(pairs, outs):
                    rdl["predict_before_edit"] = tok.decode(pre[m.to(bool)])
                    edl["before_edit_ids"] = pre
            for split_rd, split_ed in grp:
                split_res = []
                for rd, ed in zip(split_rd, split_ed):
                    for rdr, edr in zip(rd["reliability"], ed["requests"]):
                        start_t = time()
                        editor.edit_one_piece(edr)
                        rdr["edit_time"] = time() - start_t
                if look_ahead:
                    fills = [self.__get_results_after_edit__(editor.vllm, ed, rd, bp, pfx, defer=True) for rd, ed in zip(split_rd, split_ed)]
                    editor.restore_to_original_model()
                    flush()                       # the PREVIOUS split: its probes ran while this one was prepared
                    pending = (fills, split_res)
                    results.append(split_res)
                    continue
                for rd, ed in zip(split_rd, split_ed):
                    rd = self.__get_results_after_edit__(editor.vllm, ed, rd, bp, pfx)
                    split_res.append(rd)
                editor.restore_to_original_model()
                results.append(split_res)
        flush()
        return results

    # -- vllm_editor_eval.py:177-229 ------------------------------------------------------------------
    def get_mean_results(self, results: List[Dict]):
        mean_res = {"reliability": {}, "generality": {}, "locality": {}}

        def acc(dst, item):
            for name, value in item.items():
                if isinstance(value, (int, float)):
                    if name not in dst:
                        dst[name] = [0, 0]
                    dst[name][0] += value
                    dst[name][1] += 1
        for r in results:
            for rr in r["reliability"]:
                acc(mean_res["reliability"], rr)
            for sec in ("generality", "locality"):
                for sub in r[sec].keys():
                    if sub not in mean_res[sec]:
                        mean_res[sec][sub] = {}
                    for sub_res in r[sec][sub]:
                        acc(mean_res[sec][sub], sub_res)
        for name, v in mean_res["reliability"].items():
            mean_res["reliability"][name] = v[0] / v[1]
        for sec in ("generality", "locality"):
            for sub in mean_res[sec].keys():
                for name, v in mean_res[sec][sub].items():
                    mean_res[sec][sub][name] = v[0] / v[1]
        return mean_res

    # -- vllm_editor_eval.py:231-247 ------------------------------------------------------------------
    def save_results(self, save_path: str, results: Dict, decimal=4):
        def set_decimal(r):
            if isinstance(r, list):
                for i in range(len(r)):
                    r[i] = set_decimal(r[i])
            elif isinstance(r, (dict, defaultdict)):
                for k in r.keys():
                    r[k] = set_decimal(r[k])
            elif isinstance(r, float):
                r = round(r, decimal)
            return r
        res = set_decimal(deepcopy(results))
        os.makedirs(os.path.dirname(save_path), exist_ok=True)
        with open(save_path, "w") as f:
            json.dump(res, f, indent=4)
        print("save to", save_path)
