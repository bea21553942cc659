"""CPU oracle for IKE (in-context knowledge editing) -- TEST INFRASTRUCTURE (only tests/, smoke() and bench.py's
cpu_baseline may import it; the product never does).

Restates, in plain Python / numpy float64 / the PyTorch CPU oracle models:
  * the corpus builder          R/easyeditor/models/ike/util.py:54-86          (`encode_ike_facts_multimodal`)
  * the retrieval               R/easyeditor/models/ike/ike_main.py:171-208    (`apply_ike_to_multimodal_model`)
  * the in-context composition  R/easyeditor/evaluate/multimodal_evaluate.py:71-112 (`compute_icl_multimodal_edit_quality`,
                                `icl_multimodal_lm_eval`: every evaluated prompt x becomes
                                ''.join(icl_examples) + 'New Fact: {p} {t}\\nPrompt: {x}')
and drives it through the oracle's evaluator (devqa_oracle.evaluate_sequential_edit) on any oracle model that offers
`get_llm_input_embeds` (OracleBlip2 / OracleLlava / OracleMiniGPT4).

PARITY UNPINNED by the reference: `sentence_transformers` (encoder + `util.semantic_search`) is not installed, the
reference holds no fixture for IKE and its `editor/` stack has no IKE plugin.  What is restated from
published semantics: `util.normalize_embeddings` = row-wise L2 normalisation; `util.semantic_search(q, C, dot_score,
top_k)` = per query the top_k corpus ids by descending dot product.  Ties (equal float scores) -> lowest corpus id, a
build choice the HIP kernel shares (include/devqa.h, devqa_cosine_topk).
"""
from typing import Callable, Dict, List, Sequence

import numpy as np


def ike_sentence(new_fact: str, prompt_and_answer: str) -> str:
    return "New Fact: " + new_fact + "\nPrompt: " + prompt_and_answer + "\n\n"      # util.py:74-76, ike_main.py:198


def build_corpus(records: List[Dict], encode: Callable[[Sequence[str]], np.ndarray]) -> Dict:
    """util.py:54-86: three sentences / images / [prompt, answer] pairs per training record, then one encoder call."""
    sentences, images, prompts = [], [], []
    for d in records:
        new_fact = d["prompt"] + " " + d["target"]                                  # :61
        images.append(d["image_path"])                                              # :66-68
        images.append(d["rephrase_image_path"])
        images.append(d["locality_image_path"])
        prompts.append([d["prompt"], d["target"]])                                  # :69-71
        prompts.append([d["rephrase_prompt"], d["target"]])
        prompts.append([d["locality_prompt"], d["locality_ground_truth"]])
        sentences.append(ike_sentence(new_fact, new_fact))                          # :74-76
        sentences.append(ike_sentence(new_fact, d["rephrase_prompt"] + " " + d["target"]))
        sentences.append(ike_sentence(new_fact, d["locality_prompt"] + " " + d["locality_ground_truth"]))
    return {"sentences": sentences, "embeddings": np.asarray(encode(sentences), np.float32), "images": images,
            "prompts": prompts}


def topk_ids(corpus: np.ndarray, query: np.ndarray, k: int) -> List[int]:
    """normalize_embeddings on both sides, dot score, top-k by descending score (ike_main.py:193-202); float64."""
    c = np.asarray(corpus, np.float64)
    q = np.asarray(query, np.float64).reshape(-1)
    c = c / np.maximum(np.linalg.norm(c, axis=1, keepdims=True), 1e-12)
    q = q / max(float(np.linalg.norm(q)), 1e-12)
    s = c @ q
    order = np.lexsort((np.arange(len(s)), -s))     # descending score, ties -> lowest id
    return [int(i) for i in order[:k]]


def retrieve(corpus: Dict, encode, prompt: str, target: str, k: int) -> List[str]:
    new_fact = prompt + " " + target                                                # ike_main.py:196
    query = ike_sentence(new_fact, new_fact)                                        # :198
    ids = topk_ids(corpus["embeddings"], np.asarray(encode([query]), np.float32)[0], min(k, len(corpus["sentences"])))
    icl = [corpus["sentences"][i] for i in ids]                                     # :205
    icl.append(query)                                                               # :206
    return icl


def icl_text(icl_examples: List[str], prompt: str, target: str, x: str) -> str:
    """multimodal_evaluate.py:71,107-112: the text in front of the evaluated answer."""
    return "".join(icl_examples) + "New Fact: " + prompt + " " + target + "\nPrompt: " + x


class OracleIKEvl:
    """Editor with the interface devqa_oracle.evaluate_sequential_edit drives (restore / edit_one_piece): the edit is the
    retrieved demonstration list; afterwards every text that goes through the model's `get_llm_input_embeds` is
    composed as icl_text(...) -- the weights never change."""

    def __init__(self, model, corpus: Dict, encode, k: int = 32):
        self.model, self.corpus, self.encode, self.k = model, corpus, encode, k
        self._orig = None
        self.icl: List[str] = []
        self.fact = None

    def restore_to_original_model(self):
        if self._orig is not None:
            self.model.get_llm_input_embeds = self._orig
            self._orig = None
        self.icl, self.fact = [], None

    def edit_one_piece(self, request: Dict):
        self.icl = retrieve(self.corpus, self.encode, request["prompt"], request["target_new"], self.k)
        self.fact = (request["prompt"], request["target_new"])
        if self._orig is None:
            self._orig = self.model.get_llm_input_embeds
            inner = self._orig

            def with_context(texts, imgs=None):
                p, t = self.fact
                return inner([icl_text(self.icl, p, t, x) for x in texts], imgs)
            self.model.get_llm_input_embeds = with_context
