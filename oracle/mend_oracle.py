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


def gradient_transform(st, prefix, n_layers, u, v, idx, norm=True):
    """GradientTransform.forward in eval mode (auxiliary_networks.py:112-151)."""
    u_ = u.reshape(-1, u.shape[-1]).to(torch.float32)
    v_ = v.reshape(-1, v.shape[-1]).to(torch.float32)
    nz = (u_ != 0).any(-1) * (v_ != 0).any(-1)
    u_, v_ = u_[nz], v_[nz]
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
                                        self.cfg["aux_model"]["norm"])
            upd = xt.T @ dt * m["lr"]                       # update_delta_weight, mend_vl.py:97-114
            if m["dw"] is None:
                m["dw"], m["n"] = upd / len(dt), len(dt)
            else:
                m["dw"] = (m["dw"] * m["n"] + upd) / (m["n"] + len(dt))
                m["n"] += len(dt)
            self.last[m["name"]] = {"x": xin.reshape(-1, xin.shape[-1]), "delta": g.reshape(-1, g.shape[-1]), "xt": xt, "dt": dt,
                                    "dw": m["dw"]}
