"""BLIP-2-OPT parameter tree with the HF parameter names, backed by device buffers laid out for
the HIP kernels.

`.model` of the reference's wrapper is an nn.Module whose `named_parameters()` names are what
editor configs address (R/configs/ft_vl/blip2-opt-2.7b.yaml:8; selection by substring at
R/editor/vllm_editors/ft_vl/ft_vl.py:31-36) and whose parameters editors mutate in place
(ft_vl.py:56-61) and restore through `load_state_dict(strict=False)` (ft_vl.py:44-45).  This tree
keeps exactly that contract; its `forward` is not used -- arithmetic goes through
devqa_amd.engine.Blip2Engine (HIP kernels behind the C ABI).

HBM layout / dtype policy (bf16 compute mode):
  * every >=2-D weight is bf16, K-contiguous [out, in] -- the TN operand layout of the MFMA GEMM;
  * OPT q/k/v projections of a layer are three row-blocks of ONE fused [3d, d] buffer (one GEMM
    per layer); the three HF parameters are views into it, so in-place edits stay visible;
  * the ViT patch-embedding conv weight is stored as a [D, Kpad] GEMM operand (K = 3*P*P padded
    to a multiple of 32); the HF parameter is a [D,3,P,P] fp32 master kept for the name contract;
  * all 1-D parameters (biases, LayerNorm) and the small ViT/Q-Former embeddings are fp32;
  * parameters selected as EDIT TARGETS are promoted to fp32 masters (`promote_to_fp32`) with a
    bf16 shadow that GEMMs read; the shadow is refreshed when the master's version changes.
"""
import json
import os
from collections import OrderedDict
from types import SimpleNamespace

import torch
from torch import nn

from ....blip2_spec import param_shapes


class _NoTorchHooks:
    """The reference's hook-based editors register torch forward / backward hooks on sub-modules of `vllm.model`
    (R/editor/vllm_editors/mend_vl/mend_vl.py:63-85, tp_vl.py:71-111, nethook.Trace).  Here the modules are parameter containers --
    the arithmetic runs in HIP kernels that never call `forward` -- so a registered hook would silently never fire.  Registering one
    therefore fails LOUDLY and names what the native path offers instead."""

    def _no_hooks(self, kind):
        raise NotImplementedError(
            "devqa_amd: torch %s hooks never fire on the native model (modules are parameter containers; the forward runs in HIP kernels). "
            "Use the wrapper's split points instead: BaseVLLMForEdit.get_mid_module_inpt / get_mid_module_outpt / forward_from_mid_layer for "
            "decoder layers, engine.set_module_deltas (low-rank deltas on FFN projections, MEND_VL), engine.extra_neurons (TP_VL), or wrap "
            "vllm.get_llm_outpt / vllm.get_llm_input_embeds (LTE_VL, IKE_VL)." % kind)

    def register_forward_hook(self, *a, **k):
        self._no_hooks("forward")

    def register_forward_pre_hook(self, *a, **k):
        self._no_hooks("forward-pre")

    def register_full_backward_hook(self, *a, **k):
        self._no_hooks("backward")

    def register_full_backward_pre_hook(self, *a, **k):
        self._no_hooks("backward-pre")

    def register_backward_hook(self, *a, **k):
        self._no_hooks("backward")


class _Node(_NoTorchHooks, nn.Module):
    """Plain container; children/parameters are attached by name."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("devqa_amd modules are parameter containers; use Blip2Engine for arithmetic")


def _attach(root, name, param):
    parts = name.split(".")
    node = root
    for p in parts[:-1]:
        if p not in node._modules:
            node.add_module(p, _Node())
        node = node._modules[p]
    node.register_parameter(parts[-1], param)


class Blip2Native(_NoTorchHooks, nn.Module):
    def __init__(self, cfg, device="cuda", dtype="bf16"):
        super().__init__()
        assert dtype in ("bf16", "fp32")
        self.compute_dtype = dtype
        self.wdtype = torch.bfloat16 if dtype == "bf16" else torch.float32
        self.cfg = cfg
        t = cfg["text_config"]
        self.config = SimpleNamespace(is_encoder_decoder=False, num_query_tokens=cfg["num_query_tokens"],
                                      text_config=SimpleNamespace(**t), vision_config=SimpleNamespace(**cfg["vision_config"]),
                                      qformer_config=SimpleNamespace(**cfg["qformer_config"]))
        self.dev = torch.device(device)
        self._shapes = param_shapes(cfg)
        self._fp32_masters = {}   # name -> (param fp32, bf16 shadow, version seen)
        self._build()

    # -- construction -------------------------------------------------------------------------
    def _small_f32_names(self):
        return ("query_tokens", "vision_model.embeddings.class_embedding", "vision_model.embeddings.position_embedding",
                "vision_model.embeddings.patch_embedding.weight")

    def _fused_slot(self, name):
        """-> (group key, slot, n_slots) when `name` is a row block of a fused GEMM operand, else None.
        BLIP-2: the OPT q/k/v projections of a layer share one [3d, d] buffer (+ one [3d] bias)."""
        if ".self_attn." in name and "decoder.layers" in name:
            kind = name.split("self_attn.")[1].split(".")[0]
            if kind in ("q_proj", "k_proj", "v_proj"):
                layer = name.split("decoder.layers.")[1].split(".")[0]
                return ("dec_qkv." + layer, {"q_proj": 0, "k_proj": 1, "v_proj": 2}[kind], 3)
        return None

    def _build(self):
        dev = self.dev
        self.fused_w, self.fused_b = {}, {}
        small = set(self._small_f32_names())
        for name, shape in self._shapes.items():
            is_vec = len(shape) == 1
            dt = torch.float32 if (is_vec or name in small) else self.wdtype
            fs = self._fused_slot(name)
            if fs is not None:
                key, slot, nslots = fs
                rows = shape[0]
                if not is_vec:
                    buf = self.fused_w.setdefault(key, torch.zeros((nslots * rows, shape[1]), dtype=self.wdtype, device=dev))
                else:
                    buf = self.fused_b.setdefault(key, torch.zeros((nslots * rows,), dtype=torch.float32, device=dev))
                data = buf[slot * rows:(slot + 1) * rows]
            else:
                data = torch.zeros(shape, dtype=dt, device=dev)
            _attach(self, name, nn.Parameter(data, requires_grad=False))
        # legacy views used by the BLIP-2 engine: per-layer fused OPT qkv
        self.fused_qkv_w = {k.split(".", 1)[1]: v for k, v in self.fused_w.items() if k.startswith("dec_qkv.")}
        self.fused_qkv_b = {k.split(".", 1)[1]: v for k, v in self.fused_b.items() if k.startswith("dec_qkv.")}
        self._build_derived()

    def _build_derived(self):
        dev = self.dev
        v = self.cfg["vision_config"]
        self.patch_kreal = 3 * v["patch_size"] ** 2
        self.patch_kpad = (self.patch_kreal + 63) // 64 * 64   # K % 64 == 0: LDS-DMA staged GEMM path
        self.patch_w_gemm = torch.zeros((v["hidden_size"], self.patch_kpad), dtype=self.wdtype, device=dev)
        self.embed_T = None  # [d, V] transposed copy of the (tied) output embedding (dH = dlogits . E)
        self._derived_version = None

    def get(self, name):
        node = self
        for p in name.split("."):
            node = node._modules[p] if p in node._modules else node._parameters[p]
        return node

    # -- loading --------------------------------------------------------------------------------
    @torch.no_grad()
    def load_named_tensors(self, get_tensor, names=None, refresh=True):
        """get_tensor(name) -> CPU/GPU tensor (any float dtype) for each HF name."""
        for name in (names or self._shapes.keys()):
            src = get_tensor(name)
            dst = self.get(name)
            dst.copy_(src.to(dst.device).reshape(dst.shape).to(dst.dtype))
        if refresh:
            self.refresh_derived(force=True)

    @torch.no_grad()
    def refresh_derived(self, force=False):
        pw = self.get("vision_model.embeddings.patch_embedding.weight")
        emb = self.get("language_model.model.decoder.embed_tokens.weight")
        ver = (pw._version, emb._version, pw.data_ptr(), emb.data_ptr())   # in-place writes AND out-of-place replacement
        if not force and ver == self._derived_version:
            return
        self.patch_w_gemm.zero_()
        self.patch_w_gemm[:, :self.patch_kreal] = pw.reshape(pw.shape[0], -1).to(self.wdtype)
        if self.embed_T is None:
            self.embed_T = emb.t().contiguous()
        else:
            self.embed_T.copy_(emb.t())      # in place: the path-level context (lib.PathContext) holds this buffer's address
        self._derived_version = (pw._version, emb._version, pw.data_ptr(), emb.data_ptr())

    @classmethod
    def from_pretrained_dir(cls, path, device="cuda", dtype="bf16"):
        from safetensors import safe_open
        cfg = json.load(open(os.path.join(path, "config.json")))
        model = cls(cfg, device, dtype)
        files = [f for f in sorted(os.listdir(path)) if f.endswith(".safetensors")]
        if not files:
            raise FileNotFoundError("no *.safetensors under %s" % path)
        handles = [safe_open(os.path.join(path, f), framework="pt", device="cpu") for f in files]
        key2h = {}
        for h in handles:
            for k in h.keys():
                key2h[k] = h

        def get_tensor(name):
            if name in key2h:
                return key2h[name].get_tensor(name)
            if name.endswith("self_attn.qkv.bias"):  # older checkpoints: separate q_bias / v_bias
                base = name[:-len("qkv.bias")]
                qb = key2h[base + "q_bias"].get_tensor(base + "q_bias")
                vb = key2h[base + "v_bias"].get_tensor(base + "v_bias")
                return torch.cat([qb, torch.zeros_like(vb), vb])
            raise KeyError(name)
        model.load_named_tensors(get_tensor)
        return model

    @classmethod
    def from_synth(cls, cfg, seed, style="opt", device="cuda", dtype="bf16"):
        from ....synth import param_init
        model = cls(cfg, device, dtype)
        model.load_named_tensors(lambda n: torch.from_numpy(param_init(n, model._shapes[n], seed, style)))
        return model

    # -- edit targets -----------------------------------------------------------------------------
    @torch.no_grad()
    def promote_to_fp32(self, name):
        """Make parameter `name` an fp32 master (idempotent). GEMMs read its bf16 shadow."""
        if name in self._fp32_masters:
            return self._fp32_masters[name][0]
        p = self.get(name)
        if p.dtype == torch.float32:
            return p
        parts = name.split(".")
        node = self
        for q in parts[:-1]:
            node = node._modules[q]
        shadow = p.data  # keep the existing bf16 buffer as the shadow
        master = nn.Parameter(p.data.to(torch.float32), requires_grad=False)
        node._parameters[parts[-1]] = master
        self._fp32_masters[name] = [master, shadow, master._version]
        return master

    @torch.no_grad()
    def weight_for_gemm(self, name):
        """bf16 operand for parameter `name` (refreshing the shadow of an fp32 master if stale)."""
        ent = self._fp32_masters.get(name)
        if ent is None:
            return self.get(name)
        master, shadow, ver = ent
        if master._version != ver:
            from .... import lib
            lib.cast_f32_bf16(master.data, shadow)
            ent[2] = master._version
            slot = self._fused_slot(name)
            if slot is not None and slot[0].startswith("llama_gu."):
                # the shadow is a row block of a fused [gate | up] operand, written through a raw pointer: its interleaved copy is stale now
                self.__dict__.setdefault("_gu_il_ver", {}).pop(int(slot[0].split(".")[1]), None)
        return shadow

    @torch.no_grad()
    def refresh_shadows(self):
        """Bring the bf16 shadow of every fp32 master up to date (what weight_for_gemm does lazily per name)."""
        for name in self._fp32_masters:
            self.weight_for_gemm(name)

    def weight_table(self):
        """{name: device tensor} for lib.PathContext (include/devqa.h "PATH LEVEL"): every HF parameter under its own name (an
        fp32 edit target additionally as "<name>#shadow" = the bf16 operand GEMMs read), plus the derived GEMM operands."""
        self.refresh_derived()
        t = OrderedDict()
        for name, p in self.named_parameters():
            if self._fused_slot(name) is not None:
                continue          # row blocks of a fused operand: addressed through derived.dec_qkv.*
            t[name] = p.data
            ent = self._fp32_masters.get(name)
            if ent is not None and self.wdtype != torch.float32:
                t[name + "#shadow"] = ent[1]
        for layer, w in self.fused_qkv_w.items():
            t["derived.dec_qkv.%s.weight" % layer] = w
            t["derived.dec_qkv.%s.bias" % layer] = self.fused_qkv_b[layer]
        t["derived.patch_w_gemm"] = self.patch_w_gemm
        t["derived.embed_T"] = self.embed_T
        return t

    def storage_fingerprint(self):
        """Cheap identity of every buffer the path-level context (lib.PathContext) holds a raw pointer to: the device address of
        each parameter's CURRENT storage (looked up through the module tree on every call, so a Parameter object replaced by
        `promote_to_fp32`, a `p.data = ...` reassignment, a reload into new storage or a device move all show), the bf16 shadows of
        the fp32 masters and the derived GEMM operands.  ~0.15 ms for ~1100 tensors; in-place writes keep addresses and need no
        rebuild.  The engine rebuilds its context when this value changes (stale pointers meant silently wrong logits)."""
        slots = self.__dict__.get("_fp_slots")
        if slots is None:
            slots = []
            for name, _ in self.named_parameters():
                node = self
                parts = name.split(".")
                for q in parts[:-1]:
                    node = node._modules[q]
                slots.append((node._parameters, parts[-1]))
            self.__dict__["_fp_slots"] = slots
        extra = [ent[1].data_ptr() for ent in self._fp32_masters.values()]
        extra += [w.data_ptr() for w in self.fused_w.values()] + [b.data_ptr() for b in self.fused_b.values()]
        extra += [self.patch_w_gemm.data_ptr(), 0 if self.embed_T is None else self.embed_T.data_ptr()]
        extra += [w.data_ptr() for w in self.__dict__.get("_gu_il", {}).values()]
        return hash((tuple(d[k].data_ptr() for d, k in slots), tuple(extra)))

    # ---- LLaMA decoders (LLaVA, MiniGPT-4): the fused [gate | up] operand a second time with its rows interleaved in blocks of 16, the layout the
    # GEMM's fused SwiGLU epilogue wants (include/devqa.h, DEVQA_ACT_SWIGLU_IL16; bf16 compute mode only).  Refreshed IN PLACE when a gate / up
    # row block is written (the path-level context holds the buffers' addresses) ----
    def gu_interleaved(self):
        """{layer: [2F, d] bf16} or {} (fp32 mode, no LLaMA FFN, F % 16 != 0), up to date."""
        if self.wdtype != torch.bfloat16:
            return {}
        il = self.__dict__.setdefault("_gu_il", {})
        ver = self.__dict__.setdefault("_gu_il_ver", {})
        for name in self._fp32_masters:         # gate / up masters first: their bf16 row blocks are what gets interleaved
            slot = self._fused_slot(name)
            if slot is not None and slot[0].startswith("llama_gu."):
                self.weight_for_gemm(name)
        for key, w in self.fused_w.items():
            if not key.startswith("llama_gu.") or (w.shape[0] // 2) % 16 != 0:
                continue
            layer = int(key.split(".")[1])
            v = (w._version, w.data_ptr())
            if ver.get(layer) == v:
                continue
            F = w.shape[0] // 2
            src = torch.cat([w[:F].view(F // 16, 16, -1), w[F:].view(F // 16, 16, -1)], 1).reshape(2 * F, -1)
            if layer in il:
                il[layer].copy_(src)
            else:
                il[layer] = src.contiguous()
            ver[layer] = v
        return il

    def mark_dirty(self, name):
        """Call after writing an fp32 master through a raw pointer (torch's version counter
        only sees torch ops)."""
        ent = self._fp32_masters.get(name)
        if ent is not None:
            ent[2] = -1

    def is_fp32_master(self, name):
        return name in self._fp32_masters

    def _load_from_state_dict(self, *a, **k):  # default behaviour is fine (copy_ into params)
        return super()._load_from_state_dict(*a, **k)

    def to(self, *args, **kwargs):  # buffers are created on the target device; moving is a no-op
        return self
