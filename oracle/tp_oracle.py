"""CPU oracle for TP_VL (T-Patcher) -- TEST INFRASTRUCTURE (only tests/, smoke() and bench.py's cpu_baseline may import it).

Restates R/editor/vllm_editors/tp_vl/tp_vl.py:26-196 in plain PyTorch fp32 with autograd on the OracleBlip2 decoder: every
edit appends ONE patch neuron (key k [d_in], bias b, value v [d_out]) to the FFN of `edit_layer` -- the hooks concatenate the
extra pre-activations in front of fc1's output, the layer's ReLU acts on all of them, fc2's pre-hook splits them off and
its hook adds extra_act @ extra_values (:74-103) -- trained for `num_steps` Adam steps on
loss_e (edit label loss) + lambda_a * mean(exp(-pre_new)) on the edit prompt + lambda_m * mean(exp(pre_new * (pre_new > 0)))
on one randomly drawn memory text per step (:155-192).  Observable quirk kept: the loop REUSES the names `input_embeds,
vt_range` for the memory text (:173), so from the second step on the "edit" forward of :162-163 runs on the PREVIOUS
step's memory text (its last-L rows scored against the edit labels) -- only step 0 sees the edit prompt and image.  Pinned by tests/golden/tiny_tp_goldens.* (the reference's own TPvl,
tools/make_goldens_tp.py, with its dataset loader and unseeded rng replaced by committed sentences / seeded draws).
"""
from copy import deepcopy

import torch
import torch.nn.functional as F

from .devqa_oracle import OracleBlip2, label_loss


class OracleTPvl:
    def __init__(self, model: OracleBlip2, cfg: dict, sentences, rng):
        self.model, self.cfg, self.sentences, self.rng = model, cfg, list(sentences), rng
        self.fc1 = cfg["mlp_in_module_tmps"][0].format(cfg["edit_layer"])
        self.fc2 = cfg["mlp_out_module_tmps"][0].format(cfg["edit_layer"])
        self.d_in = model.w[self.fc1 + ".weight"].shape[1]
        self.d_out = model.w[self.fc2 + ".weight"].shape[0]
        self.restore_to_original_model()
        self.new = None
        model.module_hook = self._hook

    def restore_to_original_model(self):
        self.K, self.B, self.V = torch.zeros(self.d_in, 0), torch.zeros(0), torch.zeros(0, self.d_out)

    def _hook(self, name, inp, out):
        if name == self.fc1:
            self._pre_old = inp @ self.K + self.B
            self._pre_new = None if self.new is None else inp @ self.new[0] + self.new[1]
            self.new_extra_output = self._pre_new
        elif name == self.fc2:
            out = out + F.relu(self._pre_old) @ self.V
            if self.new is not None:
                out = out + F.relu(self._pre_new) @ self.new[2]
        return out

    def edit_one_piece(self, request):
        request = deepcopy(request)
        k = torch.zeros(self.d_in, 1, requires_grad=True)
        b = torch.zeros(1, requires_grad=True)
        v = torch.zeros(1, self.d_out, requires_grad=True)
        self.new = (k, b, v)
        (x, vt), y, msk = self.model.prompts_imgs_target_to_xym([request["prompt"]], [request["image"]], [request["target_new"]])
        opt = torch.optim.Adam([k, b, v], lr=self.cfg["lr"], weight_decay=self.cfg["weight_decay"])
        self.last_losses = []
        for _ in range(self.cfg["num_steps"]):
            with torch.enable_grad():
                loss_e = label_loss(self.model.get_llm_outpt(x, vt), y, msk)
                loss_a = torch.exp(-self.new_extra_output).mean()
                idx = int(self.rng.choice(len(self.sentences), 1)[0])
                x, vt = self.model.get_llm_input_embeds([self.sentences[idx]], None)   # overwrites the edit inputs (:173)
                self.model.get_llm_outpt(x, vt)
                o = self.new_extra_output
                loss_m = torch.exp(o * (o > 0)).mean()
                loss = loss_e + loss_a * self.cfg["loss_a_lambda"] + loss_m * self.cfg["loss_m_lambda"]
                loss.backward()
            self.last_losses.append((float(loss_e), float(loss_a), float(loss_m)))
            opt.step()
            opt.zero_grad()
        self.K = torch.cat([k.detach(), self.K], 1)
        self.B = torch.cat([b.detach(), self.B], 0)
        self.V = torch.cat([v.detach(), self.V], 0)
        self.new = None
