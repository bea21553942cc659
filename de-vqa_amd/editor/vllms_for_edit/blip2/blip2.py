"""BLIP2OPTForEdit on the HIP path: drop-in for R/editor/vllms_for_edit/blip2/blip2.py:9-88.

Same constructor arguments and methods; arithmetic runs in Blip2Engine (C-ABI HIP kernels), the
tokenizer and image decoding stay on the host exactly as in the reference (HF tokenizer files in
the model directory, PIL open+copy, bicubic resize, CLIP mean/std -- blip2.py:56-58).
"""
import json
import os
from types import SimpleNamespace
from typing import List, Optional

import numpy as np
import torch
from PIL import Image

from ..base import BaseVLLMForEdit
from .modeling import Blip2Native
from ....engine import Blip2Engine

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


class Blip2ImagePreprocessor:
    """HF BlipImageProcessor defaults: RGB, bicubic resize to SxS, 1/255 rescale, CLIP normalise."""

    def __init__(self, size):
        self.size = size
        self._mean = np.asarray(CLIP_MEAN, np.float32)
        self._std = np.asarray(CLIP_STD, np.float32)

    def __call__(self, img) -> np.ndarray:
        if isinstance(img, np.ndarray):  # pre-decoded uint8 HxWx3 (synthetic benchmark inputs)
            img = Image.fromarray(img)
        img = img.convert("RGB").resize((self.size, self.size), resample=Image.BICUBIC)
        a = np.asarray(img).astype(np.float32) * np.float32(1.0 / 255.0)
        a = (a - self._mean) / self._std
        return np.ascontiguousarray(a.transpose(2, 0, 1))


def load_tokenizer(model_path):
    from transformers import AutoTokenizer
    return AutoTokenizer.from_pretrained(model_path)


class BLIP2OPTForEdit(BaseVLLMForEdit):
    """For blip2-opt-2.7b (or any BLIP-2-OPT config directory in HF layout)."""

    def __init__(self, model_path: str = None, device="cuda", model: Blip2Native = None, tokenizer=None,
                 dtype="bf16") -> None:
        if not torch.cuda.is_available():
            raise RuntimeError("BLIP2OPTForEdit (HIP path) needs a GPU; there is no CPU fallback in the product path")
        dev = "cuda:0" if device in ("cuda", "auto", 0) else (("cuda:%d" % device) if isinstance(device, int) else device)
        torch.cuda.set_device(dev)
        if model is None:
            model = Blip2Native.from_pretrained_dir(model_path, dev, dtype)
        self.model = model
        self.tokenizer = tokenizer if tokenizer is not None else load_tokenizer(model_path)
        self.image_processor = Blip2ImagePreprocessor(model.cfg["vision_config"]["image_size"])
        self.processor = SimpleNamespace(tokenizer=self.tokenizer, image_processor=self.image_processor)
        self.engine = Blip2Engine(model)
        super().__init__(self.model, dev, False)

    def get_llm_tokenizer(self):
        return self.tokenizer

    def load_pixels(self, img):
        """path / PIL / ndarray -> fp32 [3,S,S] host array (blip2.py:56-58)."""
        if isinstance(img, str):
            with Image.open(img) as im:
                img = im.copy()
        return self.image_processor(img)

    def get_llm_input_embeds(self, texts: List[str], imgs: Optional[List] = None):
        tk = self.tokenizer(texts, return_tensors="pt", padding=True)
        from .... import lib
        B, T = tk["input_ids"].shape
        ids = lib.h2d(tk["input_ids"].reshape(-1), torch.int32, self.device)
        msk = lib.h2d(tk["attention_mask"], torch.int64, self.device)
        emb_w = self.model.get("language_model.model.decoder.embed_tokens.weight")
        emb = lib.gather_rows(emb_w, ids).to(torch.float32).view(B, T, -1)
        if imgs is not None:
            if isinstance(imgs, list):
                imgs = imgs[-1]  # quirk kept: only the LAST image of the list is used (blip2.py:54-55)
            it = self.image_features([imgs])  # [1, Q, d]
            if B != 1:
                it = it.expand(B, -1, -1)
            emb = torch.cat([it, emb], dim=1)
            msk = torch.cat([torch.ones(it.shape[:2], dtype=msk.dtype, device=self.device), msk], dim=1)
        # host original of the mask (consumers that only need the sequence lengths read it without a device -> host sync)
        mh = tk["attention_mask"].to(torch.int64)
        msk._devqa_host = mh if imgs is None else torch.cat([torch.ones((B, self.model.cfg["num_query_tokens"]), dtype=torch.int64), mh], 1)
        llm_inpt = {"attention_mask": msk, "inputs_embeds": emb, "input_ids": ids.view(B, T)}   # (input_ids: FT_VL on the embedding table scatters into it)
        if B == 1:   # hashable identity of every input row (image rows: (path, j); text rows: token id) -- lets the evaluator's
            # batched probe path compute a prefix that several probes share only once (same mathematics: attention is causal)
            ik = self.image_key(imgs)
            if imgs is None or ik is not None:
                llm_inpt["row_keys"] = ([("img", ik, j) for j in range(self.get_img_token_n())] if imgs is not None else []) + \
                    tk["input_ids"][0].tolist()
        vt_range = None if imgs is None else [0, self.get_img_token_n()]
        return llm_inpt, vt_range

    def get_llm_outpt(self, llm_inpt, vt_range=None):
        emb, msk = llm_inpt["inputs_embeds"], llm_inpt["attention_mask"]
        ps = self.engine.pack_from_embeds(emb, msk)
        logits = self.engine.full_logits(ps).view(emb.shape[0], emb.shape[1], -1)
        return SimpleNamespace(logits=logits)

    def get_img_special_token_str(self):
        return None

    def get_img_special_token_id(self):
        return None

    def get_img_token_n(self):
        return self.model.config.num_query_tokens

    def is_q_former_based(self):
        return True
