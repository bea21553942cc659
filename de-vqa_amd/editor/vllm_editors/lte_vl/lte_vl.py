"""LTE_VL editor (inference path) on the HIP engine: drop-in for R/editor/vllm_editors/lte_vl/lte_vl.py:15-147 -- same
config dataclass, plugin methods, pools and hook on `get_llm_outpt`.

An edit changes no weight: `edit_one_piece` stores the request, its PREFIX -- the LLM input embeddings of
"[Updated Information]{prompt} {target_new}\\n[Query]" together with the request's image tokens (:112-121) -- and the
sentence embedding of "{prompt} {target_new}" (:143-147).  At inference `get_llm_outpt` reads the probe's
`query_triple` (set by the evaluator, R/evaluation/vllm_editor_eval.py:140), embeds the probe prompt, takes the cosine
arg-max over the stored edits (`F.normalize` + matmul + `max`, :94-110 -- here the HIP `devqa_cosine_topk` kernel with
k = 1) and, when the similarity exceeds `sim_threshold`, runs the decoder on prefix ++ probe and drops the prefix rows
from the logits (:75-92).

Two things differ from the reference in HOW, not WHAT:
  * `probe_prefix` exposes the same retrieval to the evaluator's batched probe path (vllm_editor_eval._argmax_many), so
    the 12 post-edit probes of a sample still share one decoder pass; the per-probe hook stays for every other caller;
  * the sentence encoder (`multi-qa-mpnet-base-dot-v1` through sentence_transformers in the reference, :50; neither is
    available offline) is a constructor argument `encode(list[str]) -> [n, retrieval_embed_dim]`.

Training (:152-233): the reference fine-tunes `fine_tune_modules_path` (the shipped configs: the whole
`language_model`) with Adam at lr 5e-6 on in-context edit prompts -- per step one reliability and two generality label losses on
prefix ++ probe, and for each of the nine locality probes the mean of two KL terms (plain probe, prefix ++ probe) against logits of a
frozen copy of the model.  Here that step is ONE packed decoder pass over all 21 sequences with the layer activations saved, the
row-wise loss kernels (`devqa_vocab_rows`, `devqa_kl_dlogits`), an explicit backward through every decoder layer that also
accumulates the parameter gradients (`Blip2Engine.decoder_backward(grads=...)`: transposed-operand GEMMs for the weights,
deterministic column reductions for biases and LayerNorm parameters), the tied embedding's gradient through the lm_head rows, the
position table's through a one-hot GEMM, and `devqa_adam_step` on every parameter.  Checked against two steps of the reference's own
loop body (tests/test_lte_gpu.py::test_lte_training_steps, tools/make_goldens_lte.py).  Needs the fp32 ("faithful") wrapper: at lr
5e-6 an update is below the resolution of bf16 weights, and the reference trains in fp32.  BLIP-2 (OPT) is pinned by the reference's
own steps; LLaVA / MiniGPT-4 (LLaMA: RMSNorm, RoPE, SwiGLU, untied lm_head) run the same schedule through `LlavaEngine`'s backward and are
checked against an autograd restatement over the oracle models (oracle/lte_oracle.py::train_a_batch, itself pinned on BLIP-2).
"""
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Callable, Dict, List, Sequence, Tuple, Union

import numpy as np
import torch
import yaml

from ...base import BaseConfig
from ..base import HipAdamState, VLLMBaseEditor, VLLMBaseEditorWithTraining
from .... import lib


@dataclass
class LTEvlConfig(BaseConfig):
    @dataclass
    class TrainConfig:
        lr: float
        relia_lambda: float
        gen_lambda: float
        loc_lambda: float
    edit_model_name: str
    train_config: TrainConfig
    fine_tune_modules_path: Union[str, List]
    retriever_path: str
    retrieval_embed_dim: int
    sim_threshold: float

    @classmethod
    def from_yaml(cls, fpath):  # lte_vl.py:29-34
        with open(fpath, "r") as f:
            data = yaml.safe_load(f)
        data["train_config"] = cls.TrainConfig(**data["train_config"])
        return cls(**data)

    @classmethod
    def from_json(cls, fpath):
        raise


class LTEvl(VLLMBaseEditorWithTraining):
    reads_query_hook = True     # get_llm_outpt consumes the evaluator's `query_triple`

    def __init__(self, vllm, config: LTEvlConfig, device="cuda", vllm_proc_data=None, device_proc_data=None,
                 encode: Callable[[Sequence[str]], np.ndarray] = None):
        super().__init__(vllm, config, device)
        self.cfg = config
        if vllm_proc_data is not None:      # the frozen copy that prepares the training batches (lte_vl.py:44-48)
            self.vllm_proc_data = vllm_proc_data
            self.device_proc_data = device_proc_data
        if encode is None:
            raise RuntimeError("LTEvl needs `encode`: list[str] -> [n, %d] sentence embeddings (the reference uses "
                               "SentenceTransformer(%r))" % (config.retrieval_embed_dim, config.retriever_path))
        self.encode = encode
        self.edit_sign = "[Updated Information]"
        self.query_sign = "\n[Query]"
        self.wrap_get_llm_outpt()
        self.is_train = False
        self.restore_to_original_model()

    # ------------------------------------------------------------------------------------------------------------
    def _embed(self, texts: List[str]) -> torch.Tensor:
        e = np.asarray(self.encode(texts), np.float32)
        if e.ndim != 2 or e.shape[1] != self.cfg.retrieval_embed_dim:
            raise RuntimeError("encoder returned %s, expected [n, %d]" % (e.shape, self.cfg.retrieval_embed_dim))
        return torch.from_numpy(e).to(self.device).contiguous()

    def wrap_get_llm_outpt(self):  # lte_vl.py:75-92
        def wrap(get_llm_outpt):
            def wrapped_get_llm_outpt(input_embeds, vt_range=None):
                if self.is_train or len(self.edit_requests_pool) == 0:
                    return get_llm_outpt(input_embeds, vt_range)
                assert len(input_embeds["inputs_embeds"]) == 1  # only for inference
                (prompt, image, target) = input_embeds["query_triple"]
                _, retrieved_prefixs, _ = self.retrieval([prompt])
                if retrieved_prefixs[0] is None:
                    return get_llm_outpt(input_embeds, vt_range)
                logits = self.__get_edited_output__(get_llm_outpt, retrieved_prefixs[0], input_embeds).logits
                logits = logits[:, retrieved_prefixs[0]["attention_mask"].shape[1]:]
                return SimpleNamespace(logits=logits)
            return wrapped_get_llm_outpt
        if not hasattr(self, "original_get_llm_outpt"):
            self.original_get_llm_outpt = self.vllm.get_llm_outpt
        self.vllm.get_llm_outpt = wrap(self.original_get_llm_outpt)

    def retrieval(self, texts: List[str]):  # lte_vl.py:94-110
        assert isinstance(texts, list) and len(texts) == 1
        q = self._embed(texts)
        idx, val = lib.cosine_topk(self.text_retr_pool, q, 1, True, True)      # normalise both sides, dot, arg-max
        v, i = float(val[0, 0]), int(idx[0, 0])
        if v > self.cfg.sim_threshold:
            return [self.edit_requests_pool[i]], [self.edit_prefix_pool[i]], val
        return [None], [None], val

    def probe_prefix(self, prompt, image, target):
        """Evaluator hook for the batched probe path: the rows to put in front of this probe's LLM input ([P, d] fp32),
        or None -- the decision `wrapped_get_llm_outpt` takes for the same `query_triple`."""
        if self.is_train or len(self.edit_requests_pool) == 0:
            return None
        _, prefixs, _ = self.retrieval([prompt])
        return None if prefixs[0] is None else prefixs[0]["inputs_embeds"][0]

    def __get_edit_prefix__(self, vllm, request: Dict):  # lte_vl.py:112-121
        if request["prompt"][-1] != " " and request["target_new"][0] != " ":
            t = " " + request["target_new"]
        else:
            t = request["target_new"]
        p = self.edit_sign + request["prompt"] + t + self.query_sign
        return vllm.get_llm_input_embeds([p], [request["image"]])[0]

    def __get_edited_output__(self, get_llm_outpt, prefix: Dict, original_inpt: Dict):  # lte_vl.py:123-127
        inpt = {"attention_mask": torch.cat([prefix["attention_mask"], original_inpt["attention_mask"]], 1),
                "inputs_embeds": torch.cat([prefix["inputs_embeds"], original_inpt["inputs_embeds"]], 1)}
        return get_llm_outpt(inpt, None)

    # ---- editor basic functions (lte_vl.py:132-147) ---------------------------------------------------------------
    def name_of_editor_and_model(self) -> Tuple[str, str]:
        return "lte_vl", self.cfg.edit_model_name

    def if_can_batch_edit(self):
        return False

    def restore_to_original_model(self):
        self.edit_requests_pool = []
        self.edit_prefix_pool = []
        self.text_retr_pool = torch.zeros([0, self.cfg.retrieval_embed_dim], device=self.device)

    def edit_batch(self, requests: List[Dict]):
        raise

    def edit_one_piece(self, request: Dict) -> None:
        self.edit_requests_pool.append(request)
        self.edit_prefix_pool.append(self.__get_edit_prefix__(self.vllm, request))
        t_embd = self._embed([request["prompt"] + " " + request["target_new"]])
        self.text_retr_pool = torch.cat([self.text_retr_pool, t_embd], 0).contiguous()

    # ---- training (lte_vl.py:152-233) ------------------------------------------------------------------------------------
    def set_train(self, is_train=False):  # :152-161 (requires_grad flags have no meaning here: gradients are explicit)
        if is_train:
            self._check_trainable()
        self.is_train = bool(is_train)

    def _check_trainable(self):
        eng = self.vllm.engine
        if not hasattr(eng, "train_params"):
            raise NotImplementedError("LTE_VL training: %s has no all-parameter backward" % type(eng).__name__)
        if eng.adt != torch.float32:
            raise RuntimeError("LTE_VL training needs the fp32 wrapper (dtype='fp32'): at lr %g an Adam update is below the resolution "
                               "of bf16 weights; the reference trains in fp32" % self.cfg.train_config.lr)
        if self.cfg.fine_tune_modules_path != eng.LM_MODULE:
            raise NotImplementedError("LTE_VL training fine-tunes `%s` (what the shipped configs select), not %r"
                                      % (eng.LM_MODULE, self.cfg.fine_tune_modules_path))

    def reinit_train_parameters(self):  # :163-164
        pass

    def preprocess_train_data(self, vllm_edit_data) -> List:  # :166-167
        return vllm_edit_data.data_with_img

    def organize_batch_data(self, a_batch_of_training_data: List):  # :169-187
        d = a_batch_of_training_data[0]
        pd = getattr(self, "vllm_proc_data", None) or self.vllm
        edit_prefix = self.__get_edit_prefix__(pd, d["requests"][0])
        rel_xym = pd.prompts_imgs_target_to_xym([d["requests"][0]["prompt"]], [d["requests"][0]["image"]], [d["requests"][0]["target_new"]])
        gen_xym = {k: pd.prompts_imgs_target_to_xym([v[0]["prompt"]], [v[0]["image"]], [v[0]["target"]]) for k, v in d["generality"].items()}
        loc_xym = {}
        for k, v in d["locality"].items():      # the frozen copy's get_llm_outpt carries no retrieval hook (only self.vllm's is wrapped)
            (input_embeds, vt_range), label_ids, label_masks = pd.prompts_imgs_target_to_xym([v[0]["prompt"]], [v[0]["image"]], [v[0]["target"]])
            pre_logits = pd.get_llm_outpt(input_embeds, vt_range).logits
            loc_xym[k] = ((input_embeds, vt_range), pre_logits, label_masks)
        return edit_prefix, rel_xym, gen_xym, loc_xym

    def get_modules_for_training(self) -> Dict[str, object]:  # :189-193
        from ....utils import find_module
        if isinstance(self.cfg.fine_tune_modules_path, str):
            return {"llm": find_module(self.vllm.model, self.cfg.fine_tune_modules_path)}
        return {n: find_module(self.vllm.model, n) for n in self.cfg.fine_tune_modules_path}

    def _train_params(self):
        self._check_trainable()
        return self.vllm.engine.train_params()

    def get_a_new_optimizer(self):  # :195-198: Adam(lr), torch defaults (betas 0.9 / 0.999, eps 1e-8, no weight decay)
        ps = self._train_params()
        return HipAdamState(t=0, m={n: torch.zeros_like(p_) for n, p_ in ps.items()}, v={n: torch.zeros_like(p_) for n, p_ in ps.items()})

    def other_train_init_final(self):  # :200-201
        self.restore_to_original_model()

    @torch.no_grad()
    def train_a_batch(self, a_batch_of_training_data):
        """-> (loss, log_dict) with the reference's keys (:203-233)."""
        self._check_trainable()
        if getattr(self, "opt", None) is None:
            self.opt = self.get_a_new_optimizer()
        eng, dev, tc = self.vllm.engine, self.device, self.cfg.train_config
        edit_prefix, rel_xym, gen_xym, loc_xym = a_batch_of_training_data
        pe = edit_prefix["inputs_embeds"][0].to(dev, torch.float32)
        assert int(edit_prefix["attention_mask"].sum()) == pe.shape[0]

        def seq(x, with_prefix):
            e, am = x["inputs_embeds"], x["attention_mask"]
            assert e.shape[0] == 1 and int(am.sum()) == e.shape[1], "LTE_VL trains on one record at a time (lte_vl.py:170)"
            e = e[0].to(dev, torch.float32)
            return torch.cat([pe, e], 0) if with_prefix else e
        # ---- the 21 sequences of a step: label-loss rows first, KL rows after (row layout of the loss kernels) ----
        items, groups = [], []          # items: (embeds [T, d], L);  groups: (kind, name, lambda, mask [L], item indices, labels | pre_logits)
        (x, _vt), y, m = rel_xym
        groups.append(("rel", None, tc.relia_lambda, m[0], [len(items)], y[0]))
        items.append((seq(x, True), y.shape[1]))
        for k, ((x, _vt), y, m) in gen_xym.items():
            groups.append(("gen", k, tc.gen_lambda, m[0], [len(items)], y[0]))
            items.append((seq(x, True), y.shape[1]))
        n_label_groups = len(groups)
        for k, ((x, _vt), pre_logits, m) in loc_xym.items():
            L = m.shape[1]
            groups.append(("loc", k, tc.loc_lambda, m[0], [len(items), len(items) + 1], pre_logits[0, -L:].to(dev, torch.float32)))
            items.append((seq(x, False), L))
            items.append((seq(x, True), L))
        tmax = (max(e.shape[0] for e, _ in items) + 3) // 4 * 4
        B, d = len(items), pe.shape[1]
        emb = torch.zeros((B, tmax, d), dtype=torch.float32, device=dev)
        msk = torch.zeros((B, tmax), dtype=torch.int32, device=dev)
        row0 = []                                             # first label row of every item in the packed buffer
        for b, (e, L) in enumerate(items):
            emb[b, :e.shape[0]] = e
            msk[b, :e.shape[0]] = 1
            row0.append(b * tmax + e.shape[0] - L)
        ps = eng.pack_from_embeds(emb, msk, lens=[e.shape[0] for e, _ in items])
        n_layers = eng.t["num_hidden_layers"]
        save = {"layers": set(range(n_layers))}
        x_fin, _ = eng.decoder_layers(ps, save=save)
        rows, coef, coef_log, labels, pre_rows, spans = [], [], [], [], [], []
        for gi, (kind, name, lam, mask, its, aux) in enumerate(groups):
            mk = mask.to(torch.float32).cpu()
            tot = float(mk.sum())
            a = len(rows)
            for it in its:
                L = items[it][1]
                rows += list(range(row0[it], row0[it] + L))
                w = mk / tot / len(its)                        # label_loss / logit_KL_loss average over the mask; the two KL terms are averaged
                coef_log.append(w)
                coef.append(w * lam)
                if kind == "loc":
                    pre_rows.append(aux)
                else:
                    labels.append(aux.to(torch.int32).cpu())
            spans.append((a, len(rows)))
        ridx = torch.tensor(rows, dtype=torch.int32, device=dev)
        coef_t = torch.cat(coef).to(dev).contiguous()
        pre_ln = lib.gather_rows(x_fin, ridx)
        hn, logits = eng.head_fwd(pre_ln)
        n_lab = spans[n_label_groups - 1][1]
        _, nll, dlog = lib.vocab_rows(logits[:n_lab], torch.cat(labels).to(dev).contiguous(), coef_t[:n_lab].contiguous(), want_argmax=False,
                                      want_nll=True, want_dlogits=True, dlogits_dtype=torch.float32)
        row_loss = [nll]
        if len(groups) > n_label_groups:
            kl, dkl = lib.kl_dlogits(torch.cat(pre_rows, 0).contiguous(), logits[n_lab:], coef_t[n_lab:].contiguous(), torch.float32)
            dlog = torch.cat([dlog, dkl], 0)
            row_loss.append(kl)
        per_row = (torch.cat(row_loss) * torch.cat(coef_log).to(dev)).cpu()      # unweighted, as the reference logs them
        log = {"Reliability loss": 0.0, "Generality loss": {}, "Locality loss": {}}
        loss = 0.0
        for (kind, name, lam, _m, _its, _aux), (a, b) in zip(groups, spans):
            v = float(per_row[a:b].sum())
            loss += v * lam
            if kind == "rel":
                log["Reliability loss"] = v
            else:
                log["Generality loss" if kind == "gen" else "Locality loss"][name] = v
        # ---- backward ----
        st = self.opt
        P = self._train_params()
        G = st.get("_grads")
        if G is None:
            G = st["_grads"] = {n: torch.zeros_like(p_) for n, p_ in P.items()}
        else:
            for g in G.values():
                g.zero_()
        dlog = dlog.contiguous()
        dxr = eng.head_bwd(pre_ln, hn, dlog, G)               # lm_head / tied embedding + final norm
        dx = torch.zeros_like(x_fin)
        dx.index_copy_(0, ridx.long(), dxr)
        _, dx0 = eng.decoder_backward(ps, save, dx, set(), grads=G)
        eng.embed_bwd(msk, dx0, G)                            # learned positions (OPT)
        # ---- Adam ----
        st["t"] += 1
        for n, p_ in P.items():
            lib.adam_step_(p_.reshape(-1), G[n].reshape(-1), st["m"][n].reshape(-1), st["v"][n].reshape(-1), tc.lr, st["t"], None)
        self.last_grads = G
        eng.after_param_update()
        return loss, log
