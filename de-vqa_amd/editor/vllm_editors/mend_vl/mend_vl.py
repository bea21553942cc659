"""MEND_VL editor (edit path) on the HIP path: drop-in for R/editor/vllm_editors/mend_vl/mend_vl.py:14-225 -- same
config dataclass, same plugin methods, same arithmetic:

  edit_batch      one forward + one backward of the edit loss through the edited decoder layers; every edited Linear
                  records its input x and the gradient delta of its output (the reference's forward/backward hooks,
                  :62-71); the hyper-network (GradientTransform -> IDMLP of LRLinears, auxiliary_networks.py) maps
                  the non-zero rows to (x~, delta~); delta_W = x~^T delta~ * lr / n, a running mean over edits
                  (:97-114);
  forward         the edited modules add input @ delta_W to their output (:73-80).  delta_W has rank <= n rows, so
                  the engine keeps the FACTORS and evaluates (input @ x~^T) @ (delta~ * lr / n) -- delta_W itself
                  ([d_in, d_out] fp32 per module) is materialised only on request (`delta_weight`).
  restore         drops the factors (:160-162).

What changes is how it is computed: the forward saves the activations of the edited layers only, the backward is
explicit HIP (lm_head rows -> LayerNorm bwd -> per layer: GEMMs on cached transposed weights, ReLU bwd, LayerNorm
bwd, attention bwd), never an autograd graph over the whole network.

Trained hyper-network state comes from a reference-format `Best` checkpoint (R/editor/vllm_editors/base.py:237-252),
read with torch.load(weights_only=True).  Training runs through the reference's ABC
(`VLLMBaseEditorWithTraining.train_init` / `train`, editor/vllm_editors/base.py) with `train_a_batch` below; edited
modules must be FFN projections of decoder layers (fc1/fc2 for OPT, gate/up/down_proj for LLaMA: what the shipped
R/configs/mend_vl/*.yaml select).
"""
import os
import re
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, List

import numpy as np
import torch
import yaml

from ...base import BaseConfig
from ..base import HipAdamState, VLLMBaseEditorWithTraining
from .... import lib


@dataclass
class MENDvlConfig(BaseConfig):
    @dataclass
    class AuxModelConfig():
        n_hidden: int
        hidden_dim: int
        init: str
        norm: bool
        act: str
        rank: int
        shared: bool
        lr: float
    edit_modules: List[str]
    init_edit_lr: float
    edit_lr_lr: float
    aux_model: AuxModelConfig
    edit_model_name: str
    relia_lambda: float
    gen_lambda: float
    loc_lambda: float

    @classmethod
    def from_yaml(self, fpath):
        with open(fpath, "r") as f:
            data = yaml.safe_load(f)
        data["aux_model"] = self.AuxModelConfig(**data["aux_model"])
        return self(**data)

    @classmethod
    def from_json(self, fpath):
        raise




class _AuxState:
    """state_dict() / load_state_dict() face of the hyper-network tensors (the reference's `aux_models` ModuleDict) for
    VLLMBaseEditorWithTraining.save_ckpt / load_ckpt; the tensors themselves are plain device buffers the HIP kernels read."""

    def __init__(self, ed):
        self.ed = ed

    def state_dict(self):
        return dict(self.ed.aux)

    def load_state_dict(self, sd, strict=True):
        self.ed._load_aux(sd, strict)


class _EditLrState:
    """The reference's `edit_lrs` ParameterList: keys '0' .. 'n-1'."""

    def __init__(self, ed):
        self.ed = ed

    def state_dict(self):
        return {str(i): self.ed.lr_t[i] for i in range(len(self.ed.modules))}

    def load_state_dict(self, sd, strict=True):
        self.ed._load_lrs(sd)


class MENDvl(VLLMBaseEditorWithTraining):
    def __init__(self, vllm, config: MENDvlConfig, device="cuda:0", vllm_proc_data=None, device_proc_data=None,
                 ckpt_path=None, train_modules=None, for_train=False):
        super().__init__(vllm, config, device)
        eng = self.vllm.engine
        self.dev = eng.dev
        if config.aux_model.init != "id":
            raise NotImplementedError("native MEND_VL implements init == 'id' hyper-networks (the shipped configs)")
        # same-shape modules share a GradientTransform; edit_modules order = group by group (mend_vl.py:200-223)
        groups = OrderedDict()
        self.layers = set()
        mod_re = re.compile(eng.MEND_MODULE_RE)
        for name in config.edit_modules:
            m = mod_re.match(name)
            if m is None:
                raise NotImplementedError("native MEND_VL edits the decoder's FFN projections (%s); got %s" % (eng.MEND_MODULE_RE, name))
            self.layers.add(int(m.group(2)))
            out_dim, in_dim = self.vllm.model.get(name + ".weight").shape
            groups.setdefault((in_dim, out_dim), []).append(name)
        self.modules = []
        for shape, names in groups.items():
            for idx, name in enumerate(names):
                self.modules.append({"name": name, "shape": shape, "idx": idx, "lr": float(config.init_edit_lr),
                                     "X": [], "D": [], "n": 0})
        self.n_layers = config.aux_model.n_hidden + 1
        self.aux = None
        self.last: Dict[str, Dict] = {}
        self.training = False
        self.norm_init = {}     # GradientTransform.norm_init per shape: False after construction / load (see _transform)
        self.opt = None
        if ckpt_path is not None:
            self.load_ckpt(ckpt_path, True, False)
        elif train_modules is not None:
            self.load_train_modules(train_modules)
        elif for_train:
            self.reinit_train_parameters()
        eng.set_module_deltas({})
        if any(m["name"].endswith(("gate_proj", "up_proj")) for m in self.modules):
            # gate / up outputs receive low-rank deltas after an edit (two-pass form): the pre-edit probes must round like the post-edit ones, so
            # the model does not take the fused-SwiGLU GEMM at all (engine_llava._fuse_swiglu)
            eng.fuse_swiglu = False

    # ---- trained state ------------------------------------------------------------------------------------------
    def load_train_modules(self, tm):
        """tm = the 'train_modules' entry of a reference-layout checkpoint: {'aux_models': state_dict, 'edit_lrs': state_dict}."""
        self._load_aux(tm["aux_models"])
        self._load_lrs(tm["edit_lrs"])

    def _load_aux(self, sd, strict=True):
        new = {k: v.to(self.dev, torch.float32).contiguous() for k, v in sd.items()}
        if self.aux is not None and strict and set(new) != set(self.aux):
            raise RuntimeError("aux_models state dict keys differ: %s" % sorted(set(new) ^ set(self.aux))[:4])
        if self.aux is None or strict:
            self.aux = new
        else:
            self.aux.update(new)
        self._stats_finite = True
        for shape in {m["shape"] for m in self.modules}:
            for leaf in ("u_mean", "u_std", "v_mean", "v_std"):
                t = self.aux["%s.%s" % (str(shape), leaf)]
                if not bool(torch.isfinite(t).all()) and not self.training:
                    self._stats_finite = False

    def _load_lrs(self, sd):
        self.lr_t = torch.tensor([float(sd[str(i)]) for i in range(len(self.modules))], dtype=torch.float32, device=self.dev)
        for i, m in enumerate(self.modules):
            m["lr"] = float(sd[str(i)])

    # ---- plugin API ---------------------------------------------------------------------------------------------
    def name_of_editor_and_model(self):
        return "mend_vl", self.cfg.edit_model_name

    def if_can_batch_edit(self) -> bool:
        return True

    def restore_to_original_model(self):
        for m in self.modules:
            m["X"], m["D"], m["n"] = [], [], 0
        self.vllm.engine.set_module_deltas({})

    def edit_one_piece(self, request: Dict):
        self.edit_batch([request])

    def edit_batch(self, requests: List[Dict]):
        prompts = [r["prompt"] for r in requests]
        imgs = [r["image"] for r in requests]
        targets = [r["target_new"] for r in requests]
        (x, vt), y, msk = self.vllm.prompts_imgs_target_to_xym(prompts, imgs, targets)
        self.__edit_batch__(x, vt, y, msk)

    # ---- the edit -----------------------------------------------------------------------------------------------
    @torch.no_grad()
    def __edit_batch__(self, llm_inpt, vt_range, label_ids, label_masks):
        if self.aux is None:
            raise RuntimeError("MEND_VL needs trained hyper-network weights (ckpt_path / train_modules)")
        if not self.training and not getattr(self, "_stats_finite", True) and not any(self.norm_init.values()):
            raise RuntimeError("MEND_VL normalisation buffers are not finite: load a trained checkpoint")
        eng, dev = self.vllm.engine, self.dev
        emb, am = llm_inpt["inputs_embeds"], llm_inpt["attention_mask"]
        B, T = emb.shape[:2]
        am_h = getattr(am, "_devqa_host", None)
        ps = eng.pack_from_embeds(emb, am, lens=None if am_h is None else am_h.sum(1).tolist())
        save = {"layers": set(self.layers)}
        x_fin, _ = eng.decoder_layers(ps, save=save)
        L = label_ids.shape[1]
        lm_h = getattr(label_masks, "_devqa_host", None)      # host originals attached by prompts_imgs_target_to_xym: no sync
        li_h = getattr(label_ids, "_devqa_host", None)
        lm_h = (label_masks.cpu() if lm_h is None else lm_h).tolist()
        li_h = (label_ids.cpu() if li_h is None else li_h).tolist()
        rows, labels = [], []
        for b in range(B):
            for j in range(L):
                if int(lm_h[b][j]) != 0:
                    rows.append(b * T + (T - L) + j)
                    labels.append(int(li_h[b][j]))
        k = len(rows)
        idx = lib.h2d(rows, torch.int32, dev)
        pre_ln = lib.gather_rows(x_fin, idx)
        logits = eng.lm_head(pre_ln)
        coef = torch.full((k,), 1.0 / k, dtype=torch.float32, device=dev)   # label_loss averages over masked tokens (:344-352)
        _, nll, dlog = lib.vocab_rows(logits, lib.h2d(labels, torch.int32, dev), coef, want_argmax=False,
                                      want_nll=True, want_dlogits=True, dlogits_dtype=eng.adt)
        self._last_nll = nll                                                 # read through `last_loss` (a sync only when asked for)
        dH = lib.gemm_rows_longk(dlog, self.vllm.model.embed_T)
        dxr = eng.final_norm_bwd(pre_ln, dH)
        dx = torch.zeros_like(x_fin)
        dx.index_copy_(0, idx.long(), dxr)                                   # plumbing: scatter the k gradient rows
        caps, _ = eng.decoder_backward(ps, save, dx, {m["name"] for m in self.modules})
        self.last = {}
        # rows with a non-zero input AND a non-zero output gradient, per module (auxiliary_networks.py:118-120): the masks of all
        # modules come to the host in ONE transfer (a .nonzero() per module is a device -> host synchronisation each)
        caps32 = {m["name"]: (caps[m["name"]][0].to(torch.float32).contiguous(), caps[m["name"]][1].to(torch.float32).contiguous())
                  for m in self.modules}
        masks = torch.stack([(caps32[m["name"]][0] != 0).any(-1) & (caps32[m["name"]][1] != 0).any(-1) for m in self.modules]).cpu().numpy()
        for mi, m in enumerate(self.modules):
            xin32, d32 = caps32[m["name"]]
            nz = lib.h2d(np.nonzero(masks[mi])[0].astype(np.int32), torch.int32, dev)
            xt, dt, trace = self._transform(m, xin32, d32, nz)
            m["X"].append(xt)
            m["D"].append(dt * m["lr"])
            m["n"] += int(nz.numel())
            self.last[m["name"]] = {"x": xin32, "delta": d32, "xt": xt, "dt": dt, "trace": trace}
        self._install_deltas()

    @property
    def last_loss(self):
        """edit loss of the last edit (mean NLL over the label rows)"""
        t = getattr(self, "_last_nll", None)
        return None if t is None else float(t.mean().item())

    def _transform(self, m, xin32, d32, nz):
        """GradientTransform.forward + IDMLP (auxiliary_networks.py:112-151, 20-24, 62-83).  In training mode the running
        normalisation statistics are first updated row by row (:122-136; `norm_init` is a plain attribute the reference
        does NOT restore from a checkpoint, so the first training row after construction re-initialises them), and the
        intermediate activations are kept for the backward pass."""
        pre = "%s." % str(m["shape"])
        A = self.aux
        norm = bool(self.cfg.aux_model.norm)
        if self.training and nz.numel() > 0:
            reset = not self.norm_init.get(m["shape"], False)
            k_old = A[pre + "k"]
            k_new = lib.welford_rows(xin32, nz, reset, A[pre + "u_mean"], A[pre + "u_s"], A[pre + "u_std"], k_old)
            lib.welford_rows(d32, nz, reset, A[pre + "v_mean"], A[pre + "v_s"], A[pre + "v_std"], k_old)
            A[pre + "k"] = k_new
            self.norm_init[m["shape"]] = True
        if not self.training:       # inference: the whole transform is one path-level call (K16, devqa_mend_transform)
            layers = [{"u": A[pre + "mlp.layers.%d.u" % l], "v": A[pre + "mlp.layers.%d.v" % l], "bias": A[pre + "mlp.layers.%d.bias" % l],
                       "mode_scale": A[pre + "mlp.layers.%d.mode_scale.weight" % l][m["idx"]].contiguous(),
                       "mode_shift": A[pre + "mlp.layers.%d.mode_shift.weight" % l][m["idx"]].contiguous()} for l in range(self.n_layers)]
            D_, rank_ = xin32.shape[1] + d32.shape[1], layers[0]["v"].shape[0]
            if self.n_layers <= lib.MEND_MAX_LAYERS and D_ % 4 == 0 and rank_ % 4 == 0:
                stats = (A[pre + "u_mean"], A[pre + "u_std"], A[pre + "v_mean"], A[pre + "v_std"]) if norm else None
                xt, dt = lib.mend_transform(xin32, d32, nz, layers, stats, split_bf16=self._split_bf16())
                return xt, dt, None
        inp = lib.mend_normalize_concat(xin32, d32, nz, A[pre + "u_mean"] if norm else None, A[pre + "u_std"] if norm else None,
                                        A[pre + "v_mean"] if norm else None, A[pre + "v_std"] if norm else None, 1e-7)
        din = xin32.shape[1]
        if inp.shape[0] == 0:
            return inp[:, :din], inp[:, din:], None
        trace = []
        for l in range(self.n_layers):
            q = pre + "mlp.layers.%d." % l
            tlow = lib.gemm(inp, A[q + "v"])                  # [n, rank]   = x v^T        (exact-fp32 GEMM)
            prea = lib.gemm(tlow, A[q + "u"])                 # [n, D]      = (x v^T) u^T
            out = lib.mend_lrlinear_epilogue(prea, A[q + "bias"], A[q + "mode_scale.weight"][m["idx"]].contiguous(),
                                             A[q + "mode_shift.weight"][m["idx"]].contiguous(), inp)
            if self.training:
                trace.append({"inp": inp, "t": tlow, "pre": prea})
            inp = out
        return inp[:, :din].contiguous(), inp[:, din:].contiguous(), (trace if self.training else None)

    def transform_rows(self, m, x32, d32):
        """Inference-mode GradientTransform of EVERY given row (fp32 x [n, d_in], delta [n, d_out]) for edited module `m` -> (x~, d~):
        K16 as one path-level call.  The transform is row-wise, so rows of many edits may share a call (batched_mend.py)."""
        pre = "%s." % str(m["shape"])
        A = self.aux
        layers = [{"u": A[pre + "mlp.layers.%d.u" % l], "v": A[pre + "mlp.layers.%d.v" % l], "bias": A[pre + "mlp.layers.%d.bias" % l],
                   "mode_scale": A[pre + "mlp.layers.%d.mode_scale.weight" % l][m["idx"]].contiguous(),
                   "mode_shift": A[pre + "mlp.layers.%d.mode_shift.weight" % l][m["idx"]].contiguous()} for l in range(self.n_layers)]
        stats = (A[pre + "u_mean"], A[pre + "u_std"], A[pre + "v_mean"], A[pre + "v_std"]) if bool(self.cfg.aux_model.norm) else None
        return lib.mend_transform(x32, d32, None, layers, stats, split_bf16=self._split_bf16())

    def _split_bf16(self):
        """bf16 compute mode: the hyper-network's two GEMMs per layer run as three bf16 MFMA products of split fp32 operands (~2e-5 relative,
        3x the exact-fp32 GEMM's speed; include/devqa.h, DEVQA_MEND_SPLIT_BF16).  fp32 mode and DEVQA_MEND_SPLIT=0 keep the exact GEMM."""
        return getattr(self.vllm.model, "compute_dtype", "fp32") == "bf16" and os.environ.get("DEVQA_MEND_SPLIT", "1") != "0"

    def _install_deltas(self):
        """Factors of the running-mean delta weight for the engine: dW = X^T D / n with X, D the concatenated rows of
        all edits since the last restore (mend_vl.py:106-114), zero-padded to a multiple of 64 rows."""
        eng = self.vllm.engine
        deltas = {}
        for m in self.modules:
            if m["n"] == 0:
                continue
            X = torch.cat(m["X"])
            D = torch.cat(m["D"]) * (1.0 / m["n"])
            n = X.shape[0]
            npad = (n + 63) // 64 * 64
            Xp = torch.zeros((npad, X.shape[1]), dtype=torch.float32, device=self.dev)
            Dp = torch.zeros((npad, D.shape[1]), dtype=torch.float32, device=self.dev)
            Xp[:n], Dp[:n] = X, D
            op = (lambda t: lib.cast_f32_bf16(t.contiguous())) if eng.adt == torch.bfloat16 else (lambda t: t.contiguous())
            deltas[m["name"]] = {"xt": op(Xp), "xtT": op(Xp.t()), "dt": op(Dp), "dtT": op(Dp.t())}
        eng.set_module_deltas(deltas)

    def delta_weight(self, name):
        """fp32 [d_in, d_out] delta weight of one edited module (the reference's __delta_weight__), for inspection."""
        m = next(mm for mm in self.modules if mm["name"] == name)
        if m["n"] == 0:
            return None
        X, D = torch.cat(m["X"]), torch.cat(m["D"]) * (1.0 / m["n"])
        pad = (-X.shape[0]) % 4                                       # exact-fp32 GEMM wants K % 4 == 0: zero rows
        if pad:
            X = torch.cat([X, torch.zeros((pad, X.shape[1]), dtype=X.dtype, device=X.device)])
            D = torch.cat([D, torch.zeros((pad, D.shape[1]), dtype=D.dtype, device=D.device)])
        return lib.gemm(X.t().contiguous(), D.t().contiguous())     # X^T D: A = X^T [d_in, n], W = D^T [d_out, n]

    # ================================================================================================================
    # training (mend_vl.py:248-341; base.py:142-268).  One step = pre-edit locality logits -> edit -> reliability /
    # generality label losses + locality KL on the edited model -> gradients of the hyper-network -> clip -> Adam.
    # The reference runs 1 + 1 + 2 + 9 separate forwards and an autograd backward; here the 12 post-edit probes go
    # through the decoder as ONE packed batch (they do not interact), the backward is the explicit one of the edit
    # path, and gradients reach the hyper-network through the low-rank factors (delta_W is never materialised).
    # ================================================================================================================
    def reinit_train_parameters(self, seed=None):
        """Fresh hyper-network state as the reference constructs it (auxiliary_networks.py:31-38,45-52,99-105): u = 0,
        v ~ N(0, 1), bias = 0, mode shift 0 / scale 1, NaN normalisation buffers (filled by the first training rows),
        edit learning rates = cfg.init_edit_lr."""
        if seed is None:
            seed = getattr(self, "random_seed", None)
        g = torch.Generator().manual_seed(0 if seed is None else int(seed))
        rank = self.cfg.aux_model.rank
        tm = {"aux_models": {}, "edit_lrs": {str(i): torch.tensor(float(self.cfg.init_edit_lr)) for i in range(len(self.modules))}}
        for shape in dict.fromkeys(m["shape"] for m in self.modules):
            du, dv = shape
            D, key = du + dv, str(shape)
            n_modes = sum(1 for m in self.modules if m["shape"] == shape)
            mid = min(rank, D)
            nan = float("nan")
            tm["aux_models"].update({key + ".u_mean": torch.full((du,), nan), key + ".v_mean": torch.full((dv,), nan),
                                     key + ".u_std": torch.full((du,), nan), key + ".v_std": torch.full((dv,), nan),
                                     key + ".u_s": torch.full((du,), nan), key + ".v_s": torch.full((dv,), nan),
                                     key + ".k": torch.full((1,), nan)})
            for l in range(self.n_layers):
                q = key + ".mlp.layers.%d." % l
                tm["aux_models"].update({q + "u": torch.zeros(D, mid), q + "v": torch.randn(mid, D, generator=g),
                                         q + "bias": torch.zeros(D), q + "mode_shift.weight": torch.zeros(n_modes, D),
                                         q + "mode_scale.weight": torch.ones(n_modes, D)})
        self.load_train_modules(tm)

    def get_modules_for_training(self):
        """{'aux_models', 'edit_lrs'} -- the two entries of the reference's checkpoint (mend_vl.py:238-240)."""
        return {"aux_models": _AuxState(self), "edit_lrs": _EditLrState(self)}

    def preprocess_train_data(self, vllm_edit_data) -> List:   # mend_vl.py:245-246
        return vllm_edit_data.data

    def data_prefetch_device(self):
        return self.device if getattr(self, "prefetch", True) and str(self.device).startswith("cuda") else None

    def _trainable(self):
        return [k for k in self.aux if ".mlp.layers." in k]

    def set_train(self, if_train=False):
        self.training = bool(if_train)

    def get_a_new_optimizer(self):
        """torch.optim.Adam([{aux_models, lr = cfg.aux_model.lr}, {edit_lrs, lr = cfg.edit_lr_lr}]) (:293-296) as plain state."""
        # torch's numbering of the reference's optimizer: group 0 = aux_models.parameters() -- per GradientTransform (ModuleDict
        # order = first appearance of a weight shape in edit_modules, mend_vl.py:200-223), per LRLinear of its IDMLP: the module's own
        # parameters in registration order (u, v, bias), then its sub-modules' (mode_shift.weight, mode_scale.weight;
        # auxiliary_networks.py:31-55) -- group 1 = the edit_lrs ParameterList, one scalar per edited module (:153)
        order = []
        for shape in dict.fromkeys(m["shape"] for m in self.modules):
            for l in range(self.n_layers):
                for leaf in ("u", "v", "bias", "mode_shift.weight", "mode_scale.weight"):
                    k = "%s.mlp.layers.%d.%s" % (str(shape), l, leaf)
                    if k in self.aux:
                        order.append((k, None))
        n_aux = len(order)
        order += [("edit_lrs", i) for i in range(len(self.modules))]
        known = set(self._trainable())
        st = HipAdamState(t=0, m={}, v={}, torch_order=order if {k for k, _ in order[:n_aux]} == known else None,
                          group_sizes=[n_aux, len(self.modules)], group_lrs=[self.cfg.aux_model.lr, self.cfg.edit_lr_lr])
        for k in self._trainable():
            st["m"][k] = torch.zeros_like(self.aux[k])
            st["v"][k] = torch.zeros_like(self.aux[k])
        st["m"]["edit_lrs"] = torch.zeros_like(self.lr_t)
        st["v"]["edit_lrs"] = torch.zeros_like(self.lr_t)
        return st

    @staticmethod
    def _items(xym):
        """((llm_inpt, vt), y, m) with batch B -> per-sequence (embeds [T_b,d] fp32, labels [l], mask [l], positions [l]).
        The label window is the LAST L columns of the right-padded batch (base.py:107-108, logits[:, -L:]): column j sits at
        padded position T_pad - L + j; columns that fall into a shorter sequence's padding (mask 0 there) are dropped, so with
        B = 1 this is the sequence's own last L rows."""
        (x, _vt), y, m = xym
        emb, am = x["inputs_embeds"], x["attention_mask"]
        T_pad, L = emb.shape[1], y.shape[1]
        out = []
        for b in range(emb.shape[0]):
            T = int(am[b].sum())
            keep = [j for j in range(L) if 0 <= T_pad - L + j < T]
            if len(keep) < L and float(m[b][[j for j in range(L) if j not in keep]].sum()) != 0:
                raise RuntimeError("label mask set on a padded position")
            kj = torch.tensor(keep, dtype=torch.long, device=y.device)
            out.append((emb[b, :T].to(torch.float32), y[b][kj], m[b][kj], [T_pad - L + j for j in keep]))
        return out

    def _pack(self, items):
        eng = self.vllm.engine
        tmax = (max(it[0].shape[0] for it in items) + 3) // 4 * 4
        B, d = len(items), items[0][0].shape[1]
        emb = torch.zeros((B, tmax, d), dtype=torch.float32, device=self.dev)
        msk = torch.zeros((B, tmax), dtype=torch.int32, device=self.dev)
        rows, spans = [], []
        for b, (e, y, m, pos) in enumerate(items):
            T = e.shape[0]
            emb[b, :T] = e
            msk[b, :T] = 1
            assert len(pos) == y.shape[-1] and all(0 <= p_ < T for p_ in pos), "label rows outside the sequence"
            spans.append((len(rows), len(rows) + len(pos)))
            rows += [b * tmax + p_ for p_ in pos]
        return eng.pack_from_embeds(emb, msk), torch.tensor(rows, dtype=torch.int32, device=self.dev), spans

    @torch.no_grad()
    def train_a_batch(self, a_batch_of_training_data):
        """-> (loss, log_dict) with the reference's keys (:301-341)."""
        if self.opt is None:
            self.opt = self.get_a_new_optimizer()
        eng, dev, cfg = self.vllm.engine, self.dev, self.cfg
        edit_xym, gen_xym, loc_xym = a_batch_of_training_data
        self.restore_to_original_model()                                   # clear_module_deltas(True, True, True)
        self.training = True
        # ---- pre-edit logits of the locality probes (label rows only: the KL is taken over the last-L window) ----
        loc_names = list(loc_xym.keys())
        loc_items = [it for k in loc_names for it in self._items(loc_xym[k])]
        ps, ridx, spans = self._pack(loc_items)
        x_fin, _ = eng.decoder_layers(ps)
        pre_logits = eng.lm_head(lib.gather_rows(x_fin, ridx))              # [R_loc, V] fp32
        # ---- the edit (updates the running statistics, keeps the hyper-network activations) ------------------
        (x, vt), y, msk = edit_xym
        self.__edit_batch__(x, vt, y, msk)
        # ---- post-edit probes: reliability (the edit batch), generality, locality -- one packed pass -------------
        groups = [("rel", None, self._items(edit_xym), cfg.relia_lambda)]
        groups += [("gen", k, self._items(gen_xym[k]), cfg.gen_lambda) for k in gen_xym]
        groups += [("loc", k, self._items(loc_xym[k]), cfg.loc_lambda) for k in loc_names]
        items = [it for g in groups for it in g[2]]
        save = {"layers": set(self.layers)}
        ps, ridx, spans = self._pack(items)
        x_fin, _ = eng.decoder_layers(ps, save=save)
        pre_ln = lib.gather_rows(x_fin, ridx)
        logits = eng.lm_head(pre_ln)                                        # [R, V]
        R = logits.shape[0]
        coef = torch.zeros((R,), dtype=torch.float32, device=dev)
        labels = torch.zeros((R,), dtype=torch.int32, device=dev)
        is_kl = torch.zeros((R,), dtype=torch.bool, device=dev)
        gi, layout = 0, []
        for kind, name, its, lam in groups:
            tot = float(sum(float(it[2].sum()) for it in its))             # label_loss / logit_KL_loss average over the batch's mask
            r0 = spans[gi][0]
            for (_, y_, m_, _pos) in its:
                a, b = spans[gi]
                coef[a:b] = m_.to(dev, torch.float32) * (lam / tot)
                labels[a:b] = y_.to(dev, torch.int32)
                gi += 1
            layout.append((kind, name, r0, spans[gi - 1][1]))
        n_lab = layout[len(groups) - len(loc_names) - 1][3] if loc_names else R      # label-loss rows come first, KL rows last
        _, nll, dlog = lib.vocab_rows(logits[:n_lab], labels[:n_lab].contiguous(), coef[:n_lab].contiguous(), want_argmax=False,
                                      want_nll=True, want_dlogits=True, dlogits_dtype=eng.adt)
        row_loss = nll * coef[:n_lab]
        if loc_names:
            kl, dkl = lib.kl_dlogits(pre_logits, logits[n_lab:], coef[n_lab:].contiguous(), eng.adt)
            dlog = torch.cat([dlog, dkl], 0)
            row_loss = torch.cat([row_loss, kl * coef[n_lab:]], 0)
        row_loss_h = row_loss.cpu()
        log = {"Generality loss": {}, "Locality loss": {}}
        for kind, name, a, b in layout:
            v = float(row_loss_h[a:b].sum())
            if kind == "rel":
                log["Reliability loss"] = v
            else:
                log["Generality loss" if kind == "gen" else "Locality loss"][name] = v
        loss = float(row_loss_h.sum())
        # ---- backward: logits rows -> final norm -> edited layers; (input, output-gradient) of every edited module -----
        dH = lib.gemm_rows_longk(dlog.contiguous(), self.vllm.model.embed_T) if R <= 64 else \
            lib.gemm(dlog.contiguous(), self.vllm.model.embed_T, want="f32")
        dxr = eng.final_norm_bwd(pre_ln, dH)
        dx = torch.zeros_like(x_fin)
        dx.index_copy_(0, ridx.long(), dxr)
        caps, _ = eng.decoder_backward(ps, save, dx, {m["name"] for m in self.modules})
        # ---- gradients of the hyper-network through the low-rank factors ------------------------------------------
        G = {k: torch.zeros_like(self.aux[k]) for k in self._trainable()}
        g_lr = torch.zeros_like(self.lr_t)

        def padk(t):      # [r, k] -> zero-padded to k % 4 == 0 (exact-fp32 GEMM operand)
            pad = (-t.shape[1]) % 4
            return t.contiguous() if pad == 0 else torch.cat([t, torch.zeros((t.shape[0], pad), dtype=t.dtype, device=dev)], 1).contiguous()
        for mi, m in enumerate(self.modules):
            tr = self.last[m["name"]].get("trace")
            if not tr or m["n"] == 0:
                continue
            inp, dout = caps[m["name"]]
            inp32, dout32 = inp.to(torch.float32).contiguous(), dout.to(torch.float32).contiguous()
            xt, dt = self.last[m["name"]]["xt"], self.last[m["name"]]["dt"]
            n, s_ = xt.shape[0], m["lr"] / m["n"]
            T1 = lib.gemm(dt, dout32)                                   # [n, R'] = dt . dOut^T
            T2 = lib.gemm(xt, inp32)                                    # [n, R'] = xt . inp^T
            g_lr[mi] = (T1 * T2).sum() / m["n"]                         # d loss / d lr  (dW = xt^T dt lr / n)
            dxt = lib.gemm(padk(T1), padk(inp32.t()), alpha=s_)         # [n, d_in]  = s T1 . inp
            ddt = lib.gemm(padk(T2), padk(dout32.t()), alpha=s_)        # [n, d_out] = s T2 . dOut
            dcat = torch.cat([dxt, ddt], 1).contiguous()
            pre = "%s." % str(m["shape"])
            for l in range(self.n_layers - 1, -1, -1):
                q = pre + "mlp.layers.%d." % l
                rec = tr[l]
                A = self.aux
                dpre = lib.mend_lrlinear_bwd(rec["pre"], A[q + "bias"], A[q + "mode_scale.weight"][m["idx"]].contiguous(),
                                             A[q + "mode_shift.weight"][m["idx"]].contiguous(), dcat,
                                             G[q + "mode_scale.weight"][m["idx"]], G[q + "mode_shift.weight"][m["idx"]], G[q + "bias"])
                lib.gemm(padk(dpre.t()), padk(rec["t"].t()), residual=G[q + "u"], out_f32=G[q + "u"])          # += dpre^T t
                dtl = lib.gemm(dpre, A[q + "u"].t().contiguous())                                               # [n, rank] = dpre . u
                lib.gemm(padk(dtl.t()), padk(rec["inp"].t()), residual=G[q + "v"], out_f32=G[q + "v"])          # += dt^T inp
                dcat = lib.gemm(dtl, A[q + "v"].t().contiguous(), residual=dcat, out_f32=torch.empty_like(dcat))  # dt . v + dcat
        # ---- clip_grad_norm_(aux_models.parameters(), 100, error_if_nonfinite=True) + Adam ---------------------------
        ss = torch.zeros((1,), dtype=torch.float32, device=dev)
        for k in G:
            lib.sumsq_(G[k], ss)
        norm = float(ss.sqrt())
        if not np.isfinite(norm):
            raise RuntimeError("The total norm of the hyper-network gradients is non-finite")
        log["Grad-Norm"] = norm
        scale = torch.tensor([min(1.0, 100.0 / (norm + 1e-6))], dtype=torch.float32, device=dev)
        self.last_grads = dict(G, edit_lrs=g_lr)
        st = self.opt
        st["t"] += 1
        for k in G:
            lib.adam_step_(self.aux[k], G[k], st["m"][k], st["v"][k], cfg.aux_model.lr, st["t"], scale)
        lib.adam_step_(self.lr_t, g_lr, st["m"]["edit_lrs"], st["v"]["edit_lrs"], cfg.edit_lr_lr, st["t"], None)
        for i, m in enumerate(self.modules):
            m["lr"] = float(self.lr_t[i])
        return loss, log

    def organize_batch_data(self, a_batch_of_training_data: List):
        """mend_vl.py:264-290 -- (edit_xym, gen_xym, loc_xym) built through the wrapper's prompts_imgs_target_to_xym."""
        vllm = self.vllm
        d = a_batch_of_training_data
        edit = vllm.prompts_imgs_target_to_xym([x["requests"][0]["prompt"] for x in d], [x["requests"][0]["image"] for x in d],
                                               [x["requests"][0]["target_new"] for x in d])
        gen = {k: vllm.prompts_imgs_target_to_xym([x["generality"][k][0]["prompt"] for x in d], [x["generality"][k][0]["image"] for x in d],
                                                  [x["generality"][k][0]["target"] for x in d]) for k in d[0]["generality"]}
        loc = {k: vllm.prompts_imgs_target_to_xym([x["locality"][k][0]["prompt"] for x in d], [x["locality"][k][0]["image"] for x in d],
                                                  [x["locality"][k][0]["target"] for x in d]) for k in d[0]["locality"]}
        return edit, gen, loc

    def train_loop(self, vllm_edit_data, total_epochs=1, batch_size=1, save_ckpt_path=None, seed=None, ema_alpha=0.1, log_fn=None,
                   data_buffer_size=8, prefetch=True):
        """Convenience driver over the reference API: `train_init(...)` then `train(total_epochs)` (base.py:142-225) with the
        records in a scratch directory, `log_fn(i, log_dict)` instead of the scalar writer, and the best-EMA `Best` checkpoint copied
        to `save_ckpt_path`.  With `prefetch` the next batches are organised (image encodes, embeddings) by ParallelDataset's
        producer thread on a second HIP stream of the same GPU while this thread trains -- the reference does that on a second
        GPU with a second model copy (R/utils/__init__.py:149-156); the editor only trains the hyper-network, so the frozen
        model serves both.  Returns the final EMA loss."""
        import shutil
        import tempfile
        from ....dataset.vllm import BaseVLLMEditData
        if not isinstance(vllm_edit_data, BaseVLLMEditData):
            recs = list(vllm_edit_data)

            class _Data(BaseVLLMEditData):
                def dataset_name(self):
                    return "records"
            vllm_edit_data = _Data(recs, recs)
        self.prefetch = bool(prefetch)
        tmp = tempfile.mkdtemp(prefix="devqa_mend_train_")
        keep_state = self.aux
        try:
            self.train_init(vllm_edit_data, batch_size, records_dir=tmp, train_name="run", log_per_i=1, ema_alpha=ema_alpha,
                            random_seed=seed, data_buffer_size=data_buffer_size if prefetch else 1,
                            seed_init_train_params_if_no_ckpt_path=keep_state is None)
            if log_fn is not None:
                class _W:
                    def __init__(self):
                        self.cur, self.i = {}, None

                    def add_scalar(self, name, value, i):
                        if self.i is not None and i != self.i:
                            log_fn(self.i, self.cur)
                            self.cur = {}
                        self.i = i
                        self.cur[name] = value

                    def flush(self):
                        if self.i is not None:
                            log_fn(self.i, self.cur)
                w = self.log_writer = _W()
            self.train(total_epochs)
            if log_fn is not None:
                w.flush()
            self.data_generator.close()
            best = os.path.join(self.save_ckpt_dir, "Best")
            if save_ckpt_path is not None and os.path.exists(best):
                shutil.copyfile(best, save_ckpt_path)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        return self.ema_loss
