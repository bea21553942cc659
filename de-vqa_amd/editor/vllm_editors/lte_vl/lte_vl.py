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

Not built: LTE_VL *training* (:152-233) fine-tunes the whole language model with Adam on in-context edit prompts -- a full
LLM training step, outside SURVEY.md 8's hot path; `train_a_batch` & co. raise NotImplementedError.  The shipped
reference configs then evaluate with that fine-tuned LLM loaded as the editor checkpoint; here any LLM weights loaded in
the wrapper are used as they are.
"""
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Callable, Dict, List, Sequence, Tuple, Union

import numpy as np
import torch
import yaml

from ...base import BaseConfig
from ..base import VLLMBaseEditor
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


class LTEvl(VLLMBaseEditor):
    reads_query_hook = True     # get_llm_outpt consumes the evaluator's `query_triple`

    def __init__(self, vllm, config: LTEvlConfig, device="cuda", vllm_proc_data=None, device_proc_data=None,
                 encode: Callable[[Sequence[str]], np.ndarray] = None):
        super().__init__(vllm, device)
        self.cfg = config
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

    # ---- training (lte_vl.py:152-233): full-LLM fine-tuning, not on the hot path -----------------------------------
    def set_train(self, is_train=False):
        if is_train:
            raise NotImplementedError("LTE_VL training fine-tunes the whole language model; not built on the native path")
        self.is_train = False

    def train_a_batch(self, *a, **k):
        raise NotImplementedError("LTE_VL training fine-tunes the whole language model; not built on the native path")

    organize_batch_data = get_a_new_optimizer = preprocess_train_data = train_a_batch
