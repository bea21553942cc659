"""BaseVLLMForEdit: the VLLM wrapper plugin API of the reference
(R/editor/vllms_for_edit/base.py:22-233), host logic restated; losses run on the HIP kernels.

Same public names, argument meaning and error behaviour (BaseException on invalid input).
get_mid_module_inpt/outpt and forward_from_mid_layer (torch-forward-hook utilities in the reference, base.py:138-185) are
provided for decoder layers of the language model, the modules callers address with them; other paths raise NotImplementedError.
"""
from abc import ABC, abstractmethod
from typing import List, Optional

import torch
from torch.nn.utils.rnn import pad_sequence

from ... import lib


def set_tokenizer_pad_id(tokenizer, padding_side="right"):  # base.py:12-17
    if tokenizer.pad_token_id is None:
        tokenizer.pad_token_id = tokenizer.eos_token_id
        print("Set [pad_token] as [eos_token].")
    print('Padding side is set as "%s".' % padding_side)
    tokenizer.padding_side = padding_side


class BaseVLLMForEdit(ABC):
    def __init__(self, model, device: str, auto_add_img_special_token: bool) -> None:
        super().__init__()
        self.model = model
        self.device = device
        self.auto_add_img_special_token = auto_add_img_special_token
        set_tokenizer_pad_id(self.get_llm_tokenizer(), padding_side="right")
        self.get_llm_input_embeds = self.__get_llm_input_embeds_wrap__(self.get_llm_input_embeds)

    # base.py:37-73 -- input validation + optional image-token auto prefix
    def __get_llm_input_embeds_wrap__(self, inner):
        def wrapped(texts: List[str], imgs: Optional[List] = None):
            if not isinstance(imgs, (list, type(None))) or not isinstance(texts, list):
                raise BaseException("Not support type.")
            if isinstance(imgs, list) and all(i is None for i in imgs):
                imgs = None
            ist = self.get_img_special_token_str()
            if self.auto_add_img_special_token and imgs is not None and ist is not None:
                texts = [ist + "\n" + t if t.find(ist) == -1 else t for t in texts]
            if imgs is None:
                if ist is not None:
                    for t in texts:
                        if t.find(ist) != -1:
                            raise BaseException("`imgs` is None but found special image token in `texts`.")
            else:
                if len(texts) != len(imgs):
                    raise BaseException("Number of texts (n = %s) and images (n = %s) not matched."
                                        % (len(texts), len(imgs)))
                if ist is not None:
                    begin = texts[0].find(ist)
                    for t in texts:
                        if t.count(ist) != 1:
                            raise BaseException("One image must correspond to one text.")
                        if t[:begin] != texts[0][:begin]:
                            raise BaseException("Special image token with different prefixes is not supported")
            return inner(texts, imgs)
        return wrapped

    # base.py:75-109 -- strings -> (embeds, vt_range), label ids, label masks (int bookkeeping on host)
    def xym_token_bookkeeping(self, prompts, targets):
        """Returns (input_strs, label_ids [B,L] int64 cpu, label_masks [B,L] int64 cpu, min_prompt_tok_n)."""
        targets = [" " + t if p[-1] not in [" ", "\n"] and t[0] not in [" ", "\n"] else t
                   for p, t in zip(prompts, targets)]
        tokenizer = self.get_llm_tokenizer()
        cache = self.__dict__.setdefault("_tok_ids_cache", {})      # the evaluator tokenises every locality probe twice (before / after the edit)

        def ids_of(text):
            v = cache.get(text)
            if v is None:
                if len(cache) > 8192:
                    cache.clear()
                v = cache[text] = tuple(tokenizer(text)["input_ids"])
            return v
        input_strs, label_ids, label_masks = [], [], []
        min_prompt_tok_n = 999
        for p, t in zip(prompts, targets):
            s = p + t
            input_strs.append(s)
            ids = torch.as_tensor(ids_of(s), dtype=torch.long)
            lab = torch.roll(ids, -1, 0)
            n_prompt = len(ids_of(p))
            min_prompt_tok_n = min(min_prompt_tok_n, n_prompt)
            m = torch.zeros_like(lab)
            m[n_prompt - 1:-1] += 1
            label_ids.append(lab)
            label_masks.append(m)
        y = pad_sequence(label_ids, True, tokenizer.pad_token_id)[:, min_prompt_tok_n - 1:]
        m = pad_sequence(label_masks, True, 0)[:, min_prompt_tok_n - 1:]
        return input_strs, y, m, min_prompt_tok_n

    def prompts_imgs_target_to_xym(self, prompts: List[str], imgs: List, targets: List[str]):
        input_strs, y, m, _ = self.xym_token_bookkeeping(prompts, targets)
        input_embeds, vt_range = self.get_llm_input_embeds(input_strs, imgs)
        yd, md = lib.h2d(y, y.dtype, self.device), lib.h2d(m, m.dtype, self.device)
        yd._devqa_host, md._devqa_host = y, m        # the host originals: readers of single entries need no device -> host sync
        return (input_embeds, vt_range), yd, md

    # base.py:111-119 (K9) -- masked NLL on the HIP vocab-rows kernel (no autograd graph)
    def label_loss(self, logits, label_ids, label_masks, average=True):
        L = label_ids.shape[1]
        rows = logits[:, -L:].reshape(-1, logits.shape[-1]).to(torch.float32).contiguous()
        labels = label_ids.reshape(-1).to(torch.int32).contiguous()
        _, nll, _ = lib.vocab_rows(rows, labels, None, want_argmax=False, want_nll=True)
        msk = label_masks.reshape(-1).to(torch.float32)
        loss = torch.where(msk != 0, nll * msk, torch.zeros_like(nll)).sum()
        return loss / msk.sum() if average else loss

    def logit_KL_loss(self, logits1, logits2, label_masks, average=True):  # base.py:121-132 (K18, MEND_VL locality loss)
        from ... import lib
        L = label_masks.shape[1]
        V = logits1.shape[-1]
        a = logits1[:, -L:].to(torch.float32).reshape(-1, V).contiguous()
        b = logits2[:, -L:].to(torch.float32).reshape(-1, V).contiguous()
        kl = lib.logit_kl_rows(a, b).view(label_masks.shape)
        msk = label_masks.to(kl.device, torch.float32)
        loss = (kl * msk).sum()
        return loss / msk.sum() if average else loss

    # ---- image-feature cache (native path only) --------------------------------------------------------------
    @staticmethod
    def image_key(img):
        """Hashable identity of an image object, or None (PIL images / arrays: no identity, never cached or shared)."""
        if isinstance(img, str):
            return img
        if isinstance(img, torch.Tensor):
            return ("px", img.data_ptr(), img._version, tuple(img.shape))
        return None

    def invalidate_image_features(self):
        """Drop the cached image features and the watch list behind them: an editor wrote a parameter of the image path through a raw
        pointer (torch's version counters do not see HIP kernels) or replaced Parameter objects (promote_to_fp32)."""
        self.__dict__.get("_img_feat_cache", {}).clear()
        self.__dict__.get("_img_feat_pins", {}).clear()
        self.__dict__["_img_feat_watched"] = None
        self.__dict__["_img_feat_stamp"] = None

    def image_features(self, imgs):
        """[len(imgs), n_img, d] fp32 features of images given as paths / PIL / arrays.  An image addressed by PATH is
        encoded once and kept (160-entry LRU) for as long as no parameter outside the language model changes: the
        evaluator shows the same 4 images of a sample to 21 probes (R/evaluation/vllm_editor_eval.py:98-121) and every
        shipped editor edits language-model weights only.  The stamp below (version counters of all non-LLM
        parameters) drops the cache if an editor ever touches the vision side."""
        import numpy as np
        from collections import OrderedDict
        cache = self.__dict__.setdefault("_img_feat_cache", OrderedDict())
        watched = self.__dict__.get("_img_feat_watched")
        if watched is None:     # the module tree is fixed after construction: walk it once
            lm = self._lm_param_prefix()
            watched = self._img_feat_watched = [p for n, p in self.model.named_parameters() if not n.startswith(lm)]
        stamp = tuple(p._version for p in watched)
        if self.__dict__.get("_img_feat_stamp") != stamp:
            cache.clear()
            self.__dict__.get("_img_feat_pins", {}).clear()
            self._img_feat_stamp = stamp
        # identity of an image: its path, or -- for pre-processed pixel values already resident in HBM ([3,S,S] tensors, the form
        # synthetic benchmark data arrives in) -- the tensor's storage + version; such an entry keeps the tensor referenced, so
        # its address cannot be handed to another image while the entry lives
        keys = [self.image_key(i) for i in imgs]
        need = [j for j, k in enumerate(keys) if k is None or k not in cache]
        feats = {}
        if need:
            uniq = []
            for j in need:   # encode each distinct missing path once
                if keys[j] is None or keys[j] not in [keys[u] for u in uniq]:
                    uniq.append(j)
            if all(isinstance(imgs[j], torch.Tensor) for j in uniq):
                pix = torch.stack([imgs[j].to(self.device, torch.float32) for j in uniq])
            else:
                pix = torch.from_numpy(np.stack([self.load_pixels(imgs[j]) for j in uniq])).to(self.device)
            enc = self.engine.encode_images(pix)
            for r, j in enumerate(uniq):
                if keys[j] is not None:
                    cache[keys[j]] = enc[r]
                    if isinstance(imgs[j], torch.Tensor):
                        self.__dict__.setdefault("_img_feat_pins", {})[keys[j]] = imgs[j]
                    while len(cache) > 160:
                        old, _ = cache.popitem(last=False)
                        self.__dict__.get("_img_feat_pins", {}).pop(old, None)
                else:
                    feats[j] = enc[r]
        out = []
        for j, k in enumerate(keys):
            if k is not None:
                cache.move_to_end(k)
                out.append(cache[k])
            else:
                out.append(feats[j])
        return torch.stack(out)

    def _lm_param_prefix(self):
        return "language_model."

    def set_device(self, device):  # base.py:134-136
        self.device = device
        self.model.to(device)

    # base.py:138-185.  The reference implements these three with torch forward hooks (nethook.Trace / TraceDict) on `self.model`;
    # here the arithmetic is in HIP, so they are provided for what callers address with them -- DECODER LAYERS of the language
    # model (`llm_layer_tmp.format(i)`, e.g. "language_model.model.decoder.layers.{}"): the hidden states entering / leaving layer
    # i, and a forward that starts at layer i from given hidden states.  Any other module path raises NotImplementedError.
    def _mid_layer_index(self, module_path: str) -> int:
        import re
        from ...utils import find_module
        m = re.match(r"^(.*\.layers)\.(\d+)$", module_path)
        eng = getattr(self, "engine", None)
        if m is None or eng is None or not hasattr(eng, "decoder_layers"):
            raise NotImplementedError("native path: mid-module access is provided for decoder layers ('...layers.<i>'), got %r"
                                      % module_path)
        try:
            find_module(self.model, module_path)      # the path must exist in this model's module tree
        except Exception:
            raise NotImplementedError("native path: %r is not a module of this model" % module_path)
        n = self.engine.t["num_hidden_layers"]
        i = int(m.group(2))
        if not 0 <= i < n or not module_path.startswith(self._lm_param_prefix()):
            raise NotImplementedError("native path: %r is not a decoder layer of the language model" % module_path)
        return i

    def _pack_for_mid(self, llm_inpt):
        emb, msk = llm_inpt["inputs_embeds"], llm_inpt["attention_mask"]
        return self.engine.pack_from_embeds(emb, msk), emb.shape[0], emb.shape[1]

    @torch.no_grad()
    def get_mid_module_inpt(self, input_embeds, vt_range, mid_module_path, get_first_if_tuple=True):
        i = self._mid_layer_index(mid_module_path)
        ps, B, T = self._pack_for_mid(input_embeds)
        if i > 0:
            self.engine.decoder_layers(ps, upto_layer=i - 1)
        return ps.x.view(B, T, -1)

    @torch.no_grad()
    def get_mid_module_outpt(self, input_embeds, vt_range, mid_module_path, get_first_if_tuple=True):
        i = self._mid_layer_index(mid_module_path)
        ps, B, T = self._pack_for_mid(input_embeds)
        self.engine.decoder_layers(ps, upto_layer=i)
        return ps.x.view(B, T, -1)

    @torch.no_grad()
    def forward_from_mid_layer(self, llm_inpt, vt_range, mid_layer_inpt: torch.Tensor, llm_layer_tmp: str, mid_inpt_layer_i: int):
        """Inference from the LLM's layer `mid_inpt_layer_i` on, fed with `mid_layer_inpt` (same shape as that layer's input for
        `llm_inpt`); layers before it are skipped.  -> object with .logits [B, T, V]"""
        from types import SimpleNamespace
        i = self._mid_layer_index(llm_layer_tmp.format(mid_inpt_layer_i))
        ps, B, T = self._pack_for_mid(llm_inpt)
        if tuple(mid_layer_inpt.shape[:2]) != (B, T):
            raise BaseException("`mid_layer_inpt` must have the shape of the layer input of `llm_inpt`.")
        ps.x = mid_layer_inpt.reshape(B * T, -1).to(torch.float32).contiguous().clone()
        x, _ = self.engine.decoder_layers(ps, first_layer=i)
        return SimpleNamespace(logits=self.engine.lm_head(x).view(B, T, -1))

    # base.py:187-196 incl. the dim=1 normalisation quirk; tiny, host-side torch
    def find_closest_tokens(self, embeddings, embedding_matrix, top_k=1):
        en = embeddings / embeddings.norm(dim=1, keepdim=True)
        mn = embedding_matrix / embedding_matrix.norm(dim=1, keepdim=True)
        r = torch.topk(torch.matmul(en, mn.T), top_k, dim=-1)
        return r.indices, r.values

    @abstractmethod
    def get_llm_tokenizer(self):
        """tokenizer of the llm in the vllm"""

    @abstractmethod
    def get_llm_input_embeds(self, texts, imgs=None):
        """-> (llm_inpt dict, vt_range)"""

    @abstractmethod
    def get_llm_outpt(self, input_embeds, vt_range=None):
        """-> object with .logits"""

    @abstractmethod
    def get_img_special_token_str(self):
        pass

    @abstractmethod
    def get_img_special_token_id(self):
        pass

    @abstractmethod
    def get_img_token_n(self):
        pass

    @abstractmethod
    def is_q_former_based(self):
        pass
