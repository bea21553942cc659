"""MiniGPT4ForEdit on the HIP path: drop-in for R/editor/vllms_for_edit/minigpt4/minigpt4.py:9-81 (same constructor
arguments and methods).  The base wrapper prefixes '<ImageHere>\\n' (R/editor/vllms_for_edit/base.py:50-51); the text
is split at '<ImageHere>', the first segment is tokenised WITH special tokens (-> [BOS]) and the rest without
(modules/minigpt4.py:88-103), the 32 projected query rows go in between, `vt_range = [1, 33]` (minigpt4.py:59).
Tokenizer and image decoding stay on the host (Blip2ImageEvalProcessor semantics, modules/blip_processors.py:37-48:
RGB, resize to SxS bicubic, /255, CLIP mean/std).
"""
from types import SimpleNamespace
from typing import List, Optional

import numpy as np
import torch
from PIL import Image

from ..base import BaseVLLMForEdit
from ..blip2.blip2 import Blip2ImagePreprocessor, load_tokenizer
from .modeling import MiniGPT4Native
from ....engine_minigpt4 import IMG_PLACEHOLDER, MiniGPT4Engine


class MiniGPT4ForEdit(BaseVLLMForEdit):
    """For MiniGPT-4 (Vicuna-7B)."""

    def __init__(self, model_path: str = None, device="cuda", auto_add_img_special_token=True, model: MiniGPT4Native = None,
                 tokenizer=None, dtype="bf16") -> None:
        if not torch.cuda.is_available():
            raise RuntimeError("MiniGPT4ForEdit (HIP path) needs a GPU; there is no CPU fallback in the product path")
        dev = "cuda:0" if device in ("cuda", "auto", 0) else (("cuda:%d" % device) if isinstance(device, int) else device)
        torch.cuda.set_device(dev)
        if model is None:
            model = MiniGPT4Native.from_pretrained_dir(model_path, dev, dtype)
        self.model = model
        self.tokenizer = tokenizer if tokenizer is not None else load_tokenizer(model_path)
        self.img_processor = Blip2ImagePreprocessor(model.cfg["vision_config"]["image_size"])
        self.engine = MiniGPT4Engine(model)
        super().__init__(self.model, dev, auto_add_img_special_token)

    def get_llm_tokenizer(self):
        return self.tokenizer

    def _lm_param_prefix(self):
        return "llama_model."

    def load_pixels(self, img):
        if isinstance(img, str):
            with Image.open(img) as im:
                img = im.copy()
        return self.img_processor(img)

    def _segments(self, text):
        segs = text.split("<ImageHere>")
        assert len(segs) == 2, "Unmatched numbers of image placeholders and images."   # modules/minigpt4.py:90
        return [self.tokenizer(s, add_special_tokens=(i == 0))["input_ids"] for i, s in enumerate(segs)]

    def batched_token_ids(self, text, has_image):
        """Token ids of one probe text as get_llm_input_embeds would see it; IMG_PLACEHOLDER marks the query rows."""
        if not has_image:
            return self.tokenizer(text)["input_ids"]
        ist = self.get_img_special_token_str()
        if self.auto_add_img_special_token and text.find(ist) == -1:
            text = ist + "\n" + text
        s0, s1 = self._segments(text)
        return list(s0) + [IMG_PLACEHOLDER] + list(s1)

    def get_llm_input_embeds(self, texts: List[str], imgs: Optional[List] = None):
        """Only one image per text."""
        from .... import lib
        emb_tab = self.engine.embed_table()

        def embed(ids):
            t = lib.h2d(ids, torch.int32, self.device)
            return lib.gather_rows(emb_tab, t).to(torch.float32)
        if imgs is not None:
            feats = self.image_features(imgs)                           # [B, 32, d]
            rows = []
            for b, text in enumerate(texts):
                s0, s1 = self._segments(text)
                rows.append(torch.cat([embed(s0), feats[b], embed(s1)], 0))
            T = max(r.shape[0] for r in rows)
            emb = torch.zeros((len(rows), T, rows[0].shape[1]), dtype=torch.float32, device=self.device)
            msk = torch.zeros((len(rows), T), dtype=torch.int32, device=self.device)
            for b, r in enumerate(rows):                                 # pad_sequence(batch_first=True): right padding
                emb[b, :r.shape[0]] = r
                msk[b, :r.shape[0]] = 1
            llm_inpt = {"inputs_embeds": emb, "attention_mask": msk}
            if len(texts) == 1 and self.image_key(imgs[0]) is not None:   # row identities (see BLIP2OPTForEdit.get_llm_input_embeds)
                s0, s1 = self._segments(texts[0])
                llm_inpt["row_keys"] = list(s0) + [("img", self.image_key(imgs[0]), j) for j in range(feats.shape[1])] + list(s1)
        else:
            tk = self.tokenizer(texts, return_tensors="pt", padding=True)
            B, T = tk["input_ids"].shape
            emb = embed(tk["input_ids"].reshape(-1)).view(B, T, -1)
            llm_inpt = {"attention_mask": lib.h2d(tk["attention_mask"], tk["attention_mask"].dtype, self.device), "inputs_embeds": emb}
            if B == 1:
                llm_inpt["row_keys"] = tk["input_ids"][0].tolist()
        if self.auto_add_img_special_token:
            vt_range = None if imgs is None else [1, self.get_img_token_n() + 1]
        else:
            raise
        return llm_inpt, vt_range

    def get_llm_outpt(self, llm_inpt, vt_range=None):
        assert "inputs_embeds" in llm_inpt.keys()
        emb, msk = llm_inpt["inputs_embeds"], llm_inpt["attention_mask"]
        ps = self.engine.pack_from_embeds(emb, msk)
        logits = self.engine.full_logits(ps).view(emb.shape[0], emb.shape[1], -1)
        return SimpleNamespace(logits=logits)

    def get_img_special_token_str(self):
        return "<ImageHere>"

    def get_img_special_token_id(self):
        raise

    def get_img_token_n(self):
        return self.model.config.num_query_tokens

    def is_q_former_based(self):
        return True
