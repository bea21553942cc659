"""LlavaForEdit on the HIP path: drop-in for R/editor/vllms_for_edit/llava/llava.py:10-81 (same constructor
arguments and methods).  `auto_add_img_special_token` makes the base wrapper prefix '<image>\\n'
(R/editor/vllms_for_edit/base.py:50-51); the single `<image>` token is replaced by the projected CLIP features
(576 rows at 336 px) and `vt_range` is its span (llava.py:55-58).  Tokenizer and image decoding stay on the host
(HF tokenizer files; CLIPImageProcessor semantics: RGB, shortest edge -> S bicubic, center crop, CLIP mean/std).
"""
from types import SimpleNamespace
from typing import List, Optional

import numpy as np
import torch
from PIL import Image

from ..base import BaseVLLMForEdit
from .modeling import LlavaNative
from ....engine_llava import LlavaEngine
from ..blip2.blip2 import CLIP_MEAN, CLIP_STD, load_tokenizer


class ClipImagePreprocessor:
    def __init__(self, size):
        self.size = size
        self._mean = np.asarray(CLIP_MEAN, np.float32)
        self._std = np.asarray(CLIP_STD, np.float32)

    def __call__(self, img) -> np.ndarray:
        if isinstance(img, np.ndarray):
            img = Image.fromarray(img)
        img = img.convert("RGB")
        S = self.size
        w, h = img.size
        short, long = (w, h) if w <= h else (h, w)
        new_long = int(S * long / short)
        nw, nh = (S, new_long) if w <= h else (new_long, S)
        img = img.resize((nw, nh), resample=Image.BICUBIC)
        left, top = (nw - S) // 2, (nh - S) // 2
        img = img.crop((left, top, left + S, top + S))
        a = np.asarray(img).astype(np.float32) * np.float32(1.0 / 255.0)
        a = (a - self._mean) / self._std
        return np.ascontiguousarray(a.transpose(2, 0, 1))


class LlavaForEdit(BaseVLLMForEdit):
    """For llava-v1.5-7b-hf (or a devqa fixture directory)."""

    def __init__(self, model_path: str = None, device="cuda", auto_add_img_special_token=True, model: LlavaNative = None,
                 tokenizer=None, dtype="bf16") -> None:
        if not torch.cuda.is_available():
            raise RuntimeError("LlavaForEdit (HIP path) needs a GPU; there is no CPU fallback in the product path")
        dev = "cuda:0" if device in ("cuda", "auto", 0) else (("cuda:%d" % device) if isinstance(device, int) else device)
        torch.cuda.set_device(dev)
        if model is None:
            model = LlavaNative.from_pretrained_dir(model_path, dev, dtype)
        self.model = model
        self.tokenizer = tokenizer if tokenizer is not None else load_tokenizer(model_path)
        self.image_processor = ClipImagePreprocessor(model.cfg["vision_config"]["image_size"])
        self.processor = SimpleNamespace(tokenizer=self.tokenizer, image_processor=self.image_processor)
        self.engine = LlavaEngine(model)
        super().__init__(self.model, dev, auto_add_img_special_token)

    def get_llm_tokenizer(self):
        return self.tokenizer

    def load_pixels(self, img):
        if isinstance(img, str):
            with Image.open(img) as im:
                img = im.copy()
        return self.image_processor(img)

    def batched_token_ids(self, text, has_image):
        """Token ids of one probe text as get_llm_input_embeds would see it (image placeholder included)."""
        ist = self.get_img_special_token_str()
        if has_image and self.auto_add_img_special_token and text.find(ist) == -1:
            text = ist + "\n" + text
        return self.tokenizer(text)["input_ids"]

    def get_llm_input_embeds(self, texts: List[str], imgs: Optional[List] = None):
        from .... import lib
        tk = self.tokenizer(texts, return_tensors="pt", padding=True)
        ids = lib.h2d(tk["input_ids"], tk["input_ids"].dtype, self.device)
        msk = lib.h2d(tk["attention_mask"], tk["attention_mask"].dtype, self.device)
        B, T = ids.shape
        emb = lib.gather_rows(self.engine.embed_table(), ids.reshape(-1).to(torch.int32).contiguous()).to(torch.float32).view(B, T, -1)
        vt_range = None
        if imgs is not None:
            if B != 1:
                raise BaseException("LlavaForEdit (HIP path): image inputs are supported one text at a time")
            feats = self.image_features(imgs)                            # [1, n_img, d]
            pos = int(torch.where(tk["input_ids"][0] == self.get_img_special_token_id())[0][0])   # host copy: no device sync
            emb = torch.cat([emb[:, :pos], feats, emb[:, pos + 1:]], dim=1)
            msk = torch.ones(emb.shape[:2], dtype=msk.dtype, device=self.device)
            vt_range = [pos, pos + self.get_img_token_n()]
        out = {"attention_mask": msk, "inputs_embeds": emb, "position_ids": None}
        if B == 1:   # see BLIP2OPTForEdit.get_llm_input_embeds: row identities for shared-prefix probe batching
            ids_l = tk["input_ids"][0].tolist()
            if imgs is None:
                out["row_keys"] = ids_l
            elif self.image_key(imgs[0]) is not None:
                out["row_keys"] = ids_l[:pos] + [("img", self.image_key(imgs[0]), j) for j in range(self.get_img_token_n())] + ids_l[pos + 1:]
        return out, vt_range

    def get_llm_outpt(self, llm_inpt, vt_range=None):
        assert "inputs_embeds" in llm_inpt.keys()
        emb, msk = llm_inpt["inputs_embeds"], llm_inpt["attention_mask"]
        ps = self.engine.pack_from_embeds(emb, msk)
        logits = self.engine.full_logits(ps).view(emb.shape[0], emb.shape[1], -1)
        return SimpleNamespace(logits=logits)

    def get_img_special_token_str(self):
        return "<image>"

    def get_img_special_token_id(self):
        return self.model.config.image_token_index

    def get_img_token_n(self):
        v = self.model.config.vision_config
        return (v.image_size // v.patch_size) ** 2

    def is_q_former_based(self):
        return False
