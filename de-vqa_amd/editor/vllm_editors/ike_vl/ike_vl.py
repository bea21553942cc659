"""IKE (in-context knowledge editing) as an editor plugin behind VLLMBaseEditor.

The reference has no IKE plugin in its `editor/` stack; what it specifies is the
retrieval arithmetic and the prompt format, in the vendored EasyEdit fork:
  * corpus of 3 sentences per training record            R/easyeditor/models/ike/util.py:54-86
  * query  "New Fact: {p} {t}\\nPrompt: {p} {t}\\n\\n", cosine top-k over the L2-normalised corpus,
    ICL list = retrieved sentences + the new fact         R/easyeditor/models/ike/ike_main.py:193-206
  * every evaluated prompt x becomes ''.join(icl_examples) + "New Fact: {p} {t}\\nPrompt: {x}"
                                                           R/easyeditor/evaluate/multimodal_evaluate.py:71-112
This plugin keeps the weights untouched: `edit_one_piece` retrieves the demonstrations with the HIP
cosine top-k kernel (devqa_cosine_topk) and installs a text transform on the wrapper's
`get_llm_input_embeds` -- the same hook mechanism the reference's retrieval editors use
(R/editor/vllm_editors/lte_vl/lte_vl.py:75-92); `restore_to_original_model` removes it.  Label rows are
the LAST L positions, so prepending context leaves the evaluator's label/mask bookkeeping intact.
The sentence encoder (all-MiniLM-L6-v2 in the reference, unavailable offline) is a constructor argument.
"""
from dataclasses import dataclass
from typing import Callable, Dict, List, Sequence, Tuple

import numpy as np
import torch

from ...base import BaseConfig
from ..base import VLLMBaseEditor
from .... import lib


@dataclass
class IKEvlConfig(BaseConfig):
    edit_model_name: str
    k: int = 32                      # IKEMultimodalHyperParams.k (ike_hparams.py:38)


def ike_sentence(new_fact: str, prompt_and_answer: str) -> str:
    return f"New Fact: {new_fact}\nPrompt: {prompt_and_answer}\n\n"


def build_ike_corpus(records: List[Dict], encode: Callable[[Sequence[str]], np.ndarray]):
    """util.py:54-86 -- per training record three sentences (the fact itself, its rephrase, its locality
    neighbour) and the parallel `prompts` / `images` lists; returns the dict layout of the reference's pickle."""
    sentences, images, prompts = [], [], []
    for d in records:
        new_fact = d["prompt"] + " " + d["target"]
        images += [d["image_path"], d["rephrase_image_path"], d["locality_image_path"]]
        prompts += [[d["prompt"], d["target"]], [d["rephrase_prompt"], d["target"]],
                    [d["locality_prompt"], d["locality_ground_truth"]]]
        sentences.append(ike_sentence(new_fact, new_fact))
        sentences.append(ike_sentence(new_fact, "%s %s" % (d["rephrase_prompt"], d["target"])))
        sentences.append(ike_sentence(new_fact, "%s %s" % (d["locality_prompt"], d["locality_ground_truth"])))
    return {"sentences": sentences, "embeddings": np.asarray(encode(sentences), np.float32), "images": images,
            "prompts": prompts}


class IKEvl(VLLMBaseEditor):
    def __init__(self, vllm, config: IKEvlConfig, device="cuda:0", corpus: Dict = None,
                 encode: Callable[[Sequence[str]], np.ndarray] = None):
        super().__init__(vllm, device)
        self.cfg = config
        if corpus is None or encode is None:
            raise RuntimeError("IKEvl needs the stored corpus {sentences, embeddings} and a sentence encoder")
        self.encode = encode
        self.stored_sentences = corpus["sentences"]
        self.stored = torch.as_tensor(np.asarray(corpus["embeddings"], np.float32)).to(self.device).contiguous()
        # ike_main.py:185-194 normalises the stored embeddings once, when the corpus is loaded: the inverse norms are cached here
        self.stored_inv_norm = lib.row_inv_norm(self.stored) if self.stored.shape[0] else None
        self._orig_embeds = None
        self.icl_examples: List[str] = []
        self.facts: List[Tuple[str, str]] = []

    def name_of_editor_and_model(self) -> Tuple[str, str]:
        return "ike_vl", self.cfg.edit_model_name

    def if_can_batch_edit(self):
        return False

    # ------------------------------------------------------------------------------------------
    def retrieve(self, prompt: str, target: str) -> List[str]:
        new_fact = prompt + " " + target
        query = ike_sentence(new_fact, new_fact)                       # ike_main.py:196-198
        k = min(self.cfg.k, self.stored.shape[0])
        # on its own stream: the ids are needed on the host now, and waiting for them must not wait for the probes the evaluator has
        # queued on the main stream (it prepares this edit while the previous split's probes run)
        side = self.__dict__.get("_retr_stream")
        if side is None and torch.device(self.device).type == "cuda":
            side = self._retr_stream = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))       # (the corpus and its norms were written on the main stream)
        q_host = torch.as_tensor(np.asarray(self.encode([query]), np.float32))
        if side is not None:
            with torch.cuda.stream(side):
                q = lib.h2d(q_host, torch.float32, self.device).contiguous()
                idx, _ = lib.cosine_topk(self.stored, q, k, True, True, corpus_inv_norm=self.stored_inv_norm)   # normalize_embeddings + semantic_search(dot)
                ids = idx[0].cpu().tolist()
        else:
            q = q_host.to(self.device).contiguous()
            idx, _ = lib.cosine_topk(self.stored, q, k, True, True, corpus_inv_norm=self.stored_inv_norm)
            ids = idx[0].tolist()
        icl = [self.stored_sentences[int(i)] for i in ids]
        icl.append(query)                                              # ike_main.py:205-206
        return icl

    def context_prefix(self) -> str:
        """Text placed in front of every evaluated prompt (multimodal_evaluate.py:71,107-112)."""
        p, t = self.facts[-1]
        return "".join(self.icl_examples) + f"New Fact: {p} {t}\nPrompt: "

    def edit_one_piece(self, request: Dict) -> None:
        self.icl_examples = self.retrieve(request["prompt"], request["target_new"])
        self.facts.append((request["prompt"], request["target_new"]))
        if self._orig_embeds is None:
            self._orig_embeds = self.vllm.get_llm_input_embeds
            inner = self._orig_embeds

            def with_context(texts, imgs=None):
                pre = self.context_prefix()
                return inner([pre + t for t in texts], imgs)
            self.vllm.get_llm_input_embeds = with_context

    def edit_batch(self, requests: List[Dict]):
        for r in requests:
            self.edit_one_piece(r)

    def restore_to_original_model(self):
        if self._orig_embeds is not None:
            self.vllm.get_llm_input_embeds = self._orig_embeds
            self._orig_embeds = None
        self.icl_examples, self.facts = [], []
