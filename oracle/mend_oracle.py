"""CPU oracle for MEND_VL's edit path -- TEST INFRASTRUCTURE (only tests/, smoke() and bench.py's cpu_baseline may
import it).

Restates R/editor/vllm_editors/mend_vl/mend_vl.py:47-225 (EditLinear hooks, MENDvl.edit_batch / __edit_batch__ /
restore) and R/editor/vllm_editors/mend_vl/auxiliary_networks.py:4-151 (GradientTransform eval-mode forward, IDMLP,
LRLinear) in plain PyTorch fp32 with autograd on the OracleBlip2 decoder.  Pinned by tests/golden/tiny_mend_goldens.*
(the reference's own MENDvl run by tools/make_goldens_mend.py).
"""
from collections import OrderedDict
from copy import deepcopy

import torch

from .devqa_oracle import OracleBlip2, label_loss


def lr_linear(x, u, v, bias, scale, shift):
    """LRLinear.forward with init == 'id' (auxiliary_networks.py:62-83): the clamp is applied in every layer (the
    constructor's `relu` flag is never read) and the input is added back."""
    pre = (u @ (v @ x.T)).T + bias
    pre = pre * scale + shift
    return pre.clamp(min=0) + x


def _update_counter(x, m, s, k):   # auxiliary_networks.py:88-91
    new_m = m + (x - m) / k
    new_s = s + (x - m) * (x - new_m)
    return new_m, new_s


def gradient_transform(st, prefix, n_layers, u, v, idx, norm=True, training=False, flags=None):
    """GradientTransform.forward (auxiliary_networks.py:112-151).  training=True also runs the per-row running
    statistics update of :122-136 (`flags[prefix]` is the module's `norm_init` attribute: False after construction,
    NOT restored by load_state_dict, so the first training row re-initialises the statistics)."""
    u_ = u.reshape(-1, u.shape[-1]).to(torch.float32)
    v_ = v.reshape(-1, v.shape[-1]).to(torch.float32)
    nz = (u_ != 0).any(-1) * (v_ != 0).any(-1)
    u_, v_ = u_[nz], v_[nz]
    if training:
        with torch.no_grad():
            for r in range(u_.shape[0]):
                if not flags.get(prefix, False):
                    st[prefix + "u_mean"], st[prefix + "v_mean"] = u_[r].clone(), v_[r].clone()
                    st[prefix + "u_s"], st[prefix + "v_s"] = torch.zeros_like(u_[r]), torch.zeros_like(v_[r])
                    st[prefix + "k"] = torch.ones(1)
                    flags[prefix] = True
                else:
                    st[prefix + "k"] = st[prefix + "k"] + 1
                    st[prefix + "u_mean"], st[prefix + "u_s"] = _update_counter(u_[r], st[prefix + "u_mean"], st[prefix + "u_s"], st[prefix + "k"])
                    st[prefix + "v_mean"], st[prefix + "v_s"] = _update_counter(v_[r], st[prefix + "v_mean"], st[prefix + "v_s"], st[prefix + "k"])
            if float(st[prefix + "k"]) < 2:
                raise RuntimeError("Can't perform normalization with only %s samples so far" % st[prefix + "k"])
            st[prefix + "u_std"] = (st[prefix + "u_s"] / (st[prefix + "k"] - 1)) ** 0.5
            st[prefix + "v_std"] = (st[prefix + "v_s"] / (st[prefix + "k"] - 1)) ** 0.5
    if norm:
        ui = (u_ - st[prefix + "u_mean"]) / (st[prefix + "u_std"] + 1e-7)
        vi = (v_ - st[prefix + "v_mean"]) / (st[prefix + "v_std"] + 1e-7)
    else:
        ui, vi = u_, v_
    x = torch.cat((ui, vi), -1)
    for l in range(n_layers):
        q = prefix + "mlp.layers.%d." % l
        x = lr_linear(x, st[q + "u"], st[q + "v"], st[q + "bias"], st[q + "mode_scale.weight"][idx], st[q + "mode_shift.weight"][idx])
    return x[:, :u.shape[-1]], x[:, u.shape[-1]:]


class OracleMENDvl:
    def __init__(self, model: OracleBlip2, cfg: dict, train_modules: dict):
        self.model = model
        self.cfg = cfg
        self.aux = {k: v.to(torch.float32) for k, v in train_modules["aux_models"].items()}
        groups = OrderedDict()   # (in, out) -> [module names]: get_edit_modules groups same-shape modules (mend_vl.py:200-223)
        for name in cfg["edit_modules"]:
            out_dim, in_dim = model.w[name + ".weight"].shape
            groups.setdefault((in_dim, out_dim), []).append(name)
        self.modules = []        # reference order of self.edit_modules: group by group
        for shape, names in groups.items():
            for idx, name in enumerate(names):
                self.modules.append({"name": name, "shape": shape, "idx": idx})
        for i, m in enumerate(self.modules):
            m["lr"] = train_modules["edit_lrs"][str(i)].to(torch.float32)
            m["dw"], m["n"] = None, 0
        self.n_layers = cfg["aux_model"]["n_hidden"] + 1
        self.last = {}
        model.module_hook = self._hook
        self._taps = None

    def _hook(self, name, inp, out):
        for m in self.modules:
            if m["name"] + "." == name + "." or m["name"] == name.rstrip("."):
                if self._taps is not None:
                    z = torch.zeros_like(out, requires_grad=True)
                    self._taps[m["name"]] = (inp.detach(), z)
                    out = out + z
                if m["dw"] is not None:
                    out = out + inp @ m["dw"]      # forward_edit_hook, mend_vl.py:73-80
        return out

    def restore_to_original_model(self):
        for m in self.modules:
            m["dw"], m["n"] = None, 0

    def edit_one_piece(self, request):
        self.edit_batch([request])

    def edit_batch(self, requests):
        requests = deepcopy(requests)
        (x, vt), y, msk = self.model.prompts_imgs_target_to_xym([r["prompt"] for r in requests], [r["image"] for r in requests],
                                                                [r["target_new"] for r in requests])
        self._edit_xym(x, vt, y, msk)

    def _edit_xym(self, x, vt, y, msk, training=False):
        self._taps = {}
        with torch.enable_grad():
            logits = self.model.get_llm_outpt(x, vt)
            loss = label_loss(logits, y, msk)
            names = [m["name"] for m in self.modules]
            grads = torch.autograd.grad(loss, [self._taps[n][1] for n in names])
        taps, self._taps = self._taps, None
        self.last = {}
        for m, g in zip(self.modules, grads):
            xin = taps[m["name"]][0]
            xt, dt = gradient_transform(self.aux, "%s." % str(m["shape"]), self.n_layers, xin, g.detach(), m["idx"],
                                        self.cfg["aux_model"]["norm"], training, self.__dict__.setdefault("norm_flags", {}))
            upd = xt.T @ dt * m["lr"]                       # update_delta_weight, mend_vl.py:97-114
            if m["dw"] is None:
                m["dw"], m["n"] = upd / len(dt), len(dt)
            else:
                m["dw"] = (m["dw"] * m["n"] + upd) / (m["n"] + len(dt))
                m["n"] += len(dt)
            self.last[m["name"]] = {"x": xin.reshape(-1, xin.shape[-1]), "delta": g.reshape(-1, g.shape[-1]), "xt": xt, "dt": dt,
                                    "dw": m["dw"]}

    # ---- training (mend_vl.py:248-262, 293-341) ---------------------------------------------------------------------
    def set_train(self, aux_lr, edit_lr_lr):
        """Hyper-network weights and edit learning rates become leaves; torch.optim.Adam with the reference's two
        parameter groups (get_a_new_optimizer, :293-296)."""
        self.train_params = {}
        for k in list(self.aux):
            if ".mlp.layers." in k:
                self.aux[k] = self.aux[k].clone().requires_grad_(True)
                self.train_params["aux_models." + k] = self.aux[k]
        for i, m in enumerate(self.modules):
            m["lr"] = m["lr"].clone().requires_grad_(True)
            self.train_params["edit_lrs.%d" % i] = m["lr"]
        aux = [p for n, p in self.train_params.items() if n.startswith("aux_models.")]
        lrs = [p for n, p in self.train_params.items() if n.startswith("edit_lrs.")]
        self.opt = torch.optim.Adam([{"params": aux, "lr": aux_lr}, {"params": lrs, "lr": edit_lr_lr}])

    def train_a_batch(self, batch):
        from .devqa_oracle import logit_KL_loss
        edit_xym, gen_xym, loc_xym = batch
        self.restore_to_original_model()
        lam = self.cfg
        with torch.no_grad():
            loc_pre = {k: self.model.get_llm_outpt(x, vt) for k, ((x, vt), y, m) in loc_xym.items()}
        (x, vt), y, msk = edit_xym
        self._edit_xym(x, vt, y, msk, training=True)
        log = {}
        with torch.enable_grad():
            rel = lam["relia_lambda"] * label_loss(self.model.get_llm_outpt(x, vt), y, msk)
            loss = rel
            log["Reliability loss"] = float(rel)
            log["Generality loss"], log["Locality loss"] = {}, {}
            for k, ((gx, gvt), gy, gm) in gen_xym.items():
                g = lam["gen_lambda"] * label_loss(self.model.get_llm_outpt(gx, gvt), gy, gm)
                log["Generality loss"][k] = float(g)
                loss = loss + g
            for k, ((lx, lvt), ly, lm) in loc_xym.items():
                l = lam["loc_lambda"] * logit_KL_loss(loc_pre[k], self.model.get_llm_outpt(lx, lvt), lm)
                log["Locality loss"][k] = float(l)
                loss = loss + l
            loss.backward()
        params = [p for n, p in self.train_params.items() if n.startswith("aux_models.")]
        log["Grad-Norm"] = float(torch.nn.utils.clip_grad_norm_(params, 100.0, error_if_nonfinite=True))
        self.last_grads = {n: (None if p.grad is None else p.grad.detach().clone()) for n, p in self.train_params.items()}
        self.opt.step()
        self.opt.zero_grad()
        return float(loss), log
