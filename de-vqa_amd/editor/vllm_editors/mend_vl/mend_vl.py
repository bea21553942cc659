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
read with torch.load(weights_only=True).  The training loop (train_init/train/train_a_batch) is not
built; edited modules must be fc1/fc2 of decoder layers (what R/configs/mend_vl/blip2-opt-2.7b.yaml selects).
"""
import re
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, List

import torch
import yaml

from ...base import BaseConfig
from ..base import VLLMBaseEditor
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


_MOD_RE = re.compile(r"^(.*\.layers\.)(\d+)\.(fc1|fc2)$")


class MENDvl(VLLMBaseEditor):
    def __init__(self, vllm, config: MENDvlConfig, device="cuda:0", vllm_proc_data=None, device_proc_data=None,
                 ckpt_path=None, train_modules=None):
        super().__init__(vllm, device)
        self.cfg = config
        eng = self.vllm.engine
        self.dev = eng.dev
        if config.aux_model.init != "id":
            raise NotImplementedError("native MEND_VL implements init == 'id' hyper-networks (the shipped configs)")
        # same-shape modules share a GradientTransform; edit_modules order = group by group (mend_vl.py:200-223)
        groups = OrderedDict()
        self.layers = set()
        for name in config.edit_modules:
            m = _MOD_RE.match(name)
            if m is None:
                raise NotImplementedError("native MEND_VL edits decoder fc1/fc2 modules; got %s" % name)
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
        if ckpt_path is not None:
            self.load_ckpt(ckpt_path)
        elif train_modules is not None:
            self.load_train_modules(train_modules)
        eng.module_deltas = {}

    # ---- trained state ------------------------------------------------------------------------------------------
    def load_ckpt(self, ckpt_path, restrict=True, load_opt=False):
        """Reference checkpoint layout (base.py:237-268): {'i','epoch','loss','ema_loss','train_modules': {'aux_models':
        state_dict, 'edit_lrs': state_dict}, 'opt', 'lr_scheduler'}.  weights_only=True: nothing from the file runs."""
        ck = torch.load(ckpt_path, map_location="cpu", weights_only=True)
        self.load_train_modules(ck["train_modules"])
        return ck.get("i"), ck.get("epoch"), ck.get("loss"), ck.get("ema_loss")

    def load_train_modules(self, tm):
        self.aux = {k: v.to(self.dev, torch.float32).contiguous() for k, v in tm["aux_models"].items()}
        for i, m in enumerate(self.modules):
            m["lr"] = float(tm["edit_lrs"][str(i)])
        for shape in {m["shape"] for m in self.modules}:
            for leaf in ("u_mean", "u_std", "v_mean", "v_std"):
                t = self.aux["%s.%s" % (str(shape), leaf)]
                if not bool(torch.isfinite(t).all()):
                    raise RuntimeError("MEND_VL normalisation buffers are not finite: load a trained checkpoint")

    # ---- plugin API ---------------------------------------------------------------------------------------------
    def name_of_editor_and_model(self):
        return "mend_vl", self.cfg.edit_model_name

    def if_can_batch_edit(self) -> bool:
        return True

    def restore_to_original_model(self):
        for m in self.modules:
            m["X"], m["D"], m["n"] = [], [], 0
        self.vllm.engine.module_deltas = {}

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
        eng, dev = self.vllm.engine, self.dev
        emb, am = llm_inpt["inputs_embeds"], llm_inpt["attention_mask"]
        B, T = emb.shape[:2]
        ps = eng.pack_from_embeds(emb, am)
        save = {"layers": set(self.layers)}
        x_fin, _ = eng.decoder_layers(ps, save=save)
        L = label_ids.shape[1]
        rows, labels = [], []
        for b in range(B):
            for j in range(L):
                if int(label_masks[b, j]) != 0:
                    rows.append(b * T + (T - L) + j)
                    labels.append(int(label_ids[b, j]))
        k = len(rows)
        idx = torch.tensor(rows, dtype=torch.int32, device=dev)
        pre_ln = lib.gather_rows(x_fin, idx)
        logits = eng.lm_head(pre_ln)
        coef = torch.full((k,), 1.0 / k, dtype=torch.float32, device=dev)   # label_loss averages over masked tokens (:344-352)
        _, nll, dlog = lib.vocab_rows(logits, torch.tensor(labels, dtype=torch.int32, device=dev), coef, want_argmax=False,
                                      want_nll=True, want_dlogits=True, dlogits_dtype=eng.adt)
        self.last_loss = float(nll.mean().item())
        dH = lib.gemm_rows_longk(dlog, self.vllm.model.embed_T)
        dxr = eng.final_norm_bwd(pre_ln, dH)
        dx = torch.zeros_like(x_fin)
        dx.index_copy_(0, idx.long(), dxr)                                   # plumbing: scatter the k gradient rows
        caps, _ = eng.decoder_backward(ps, save, dx, {m["name"] for m in self.modules})
        self.last = {}
        for m in self.modules:
            xin, delta = caps[m["name"]]
            xin32 = xin.to(torch.float32).contiguous()
            d32 = delta.to(torch.float32).contiguous()
            nz = ((xin32 != 0).any(-1) & (d32 != 0).any(-1)).nonzero().flatten().to(torch.int32)   # auxiliary_networks.py:118-120
            xt, dt = self._transform(m, xin32, d32, nz)
            m["X"].append(xt)
            m["D"].append(dt * m["lr"])
            m["n"] += int(nz.numel())
            self.last[m["name"]] = {"x": xin32, "delta": d32, "xt": xt, "dt": dt}
        self._install_deltas()

    def _transform(self, m, xin32, d32, nz):
        """GradientTransform.forward in eval mode + IDMLP (auxiliary_networks.py:112-151, 20-24, 62-83)."""
        pre = "%s." % str(m["shape"])
        A = self.aux
        norm = bool(self.cfg.aux_model.norm)
        inp = lib.mend_normalize_concat(xin32, d32, nz, A[pre + "u_mean"] if norm else None, A[pre + "u_std"] if norm else None,
                                        A[pre + "v_mean"] if norm else None, A[pre + "v_std"] if norm else None, 1e-7)
        if inp.shape[0] == 0:
            return inp[:, :xin32.shape[1]], inp[:, xin32.shape[1]:]
        for l in range(self.n_layers):
            q = pre + "mlp.layers.%d." % l
            tlow = lib.gemm(inp, A[q + "v"])                  # [n, rank]   = x v^T        (exact-fp32 GEMM)
            prea = lib.gemm(tlow, A[q + "u"])                 # [n, D]      = (x v^T) u^T
            inp = lib.mend_lrlinear_epilogue(prea, A[q + "bias"], A[q + "mode_scale.weight"][m["idx"]].contiguous(),
                                             A[q + "mode_shift.weight"][m["idx"]].contiguous(), inp)
        din = xin32.shape[1]
        return inp[:, :din].contiguous(), inp[:, din:].contiguous()

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
            ent = {"xt": op(Xp), "xtT": op(Xp.t()), "dt": op(Dp), "dtT": op(Dp.t())}
            if m["name"].endswith("fc1"):
                ent["w_cat"] = torch.cat([self.vllm.model.weight_for_gemm(m["name"] + ".weight"), ent["dtT"]], 1).contiguous()
            deltas[m["name"]] = ent
        eng.module_deltas = deltas

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
