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

LLaMA FFN (R/configs/tp_vl/llava-v1.5-7b.yaml: in = gate_proj AND up_proj, out = down_proj): the same hooks extend both
projections' outputs, the MLP's silu(gate) * up acts on the extra columns too, and loss_a / loss_m are summed over the two
in-layers (:164-177).  Pinned by tests/golden/tiny_tp_llava_goldens.* (the reference's TPvl over HF LLaVA, same generator).
"""
from copy import deepcopy

import torch
import torch.nn.functional as F

from .devqa_oracle import OracleBlip2, label_loss


class OracleTPvl:
    def __init__(self, model, cfg: dict, sentences, rng):
        self.model, self.cfg, self.sentences, self.rng = model, cfg, list(sentences), rng
        self.ins = [t.format(cfg["edit_layer"]) for t in cfg["mlp_in_module_tmps"]]
        self.out = cfg["mlp_out_module_tmps"][0].format(cfg["edit_layer"])
        self.gated = len(self.ins) == 2
        self.d_in = model.w[self.ins[0] + ".weight"].shape[1]
        self.d_out = model.w[self.out + ".weight"].shape[0]
        self.restore_to_original_model()
        self.new = None
        model.module_hook = self._hook

    def restore_to_original_model(self):
        n_in = len(self.ins)
        self.K = [torch.zeros(self.d_in, 0) for _ in range(n_in)]      # per in-layer: [d_in, n], newest neuron first
        self.B = [torch.zeros(0) for _ in range(n_in)]
        self.V = torch.zeros(0, self.d_out)

    def _hook(self, name, inp, out):
        if name in self.ins:
            j = self.ins.index(name)
            if j == 0:
                self._pre_old, self._pre_new = [None] * len(self.ins), [None] * len(self.ins)
            self._pre_old[j] = inp @ self.K[j] + self.B[j]
            self._pre_new[j] = None if self.new is None else inp @ self.new[0][j] + self.new[1][j]
            self.new_extra_outputs = self._pre_new
        elif name == self.out:
            def act(pre):
                return F.silu(pre[0]) * pre[1] if self.gated else F.relu(pre[0])
            out = out + act(self._pre_old) @ self.V
            if self.new is not None:
                out = out + act(self._pre_new) @ self.new[2]
        return out

    def edit_one_piece(self, request):
        request = deepcopy(request)
        ks = [torch.zeros(self.d_in, 1, requires_grad=True) for _ in self.ins]
        bs = [torch.zeros(1, requires_grad=True) for _ in self.ins]
        v = torch.zeros(1, self.d_out, requires_grad=True)
        self.new = (ks, bs, v)
        (x, vt), y, msk = self.model.prompts_imgs_target_to_xym([request["prompt"]], [request["image"]], [request["target_new"]])
        opt = torch.optim.Adam(ks + bs + [v], lr=self.cfg["lr"], weight_decay=self.cfg["weight_decay"])
        self.last_losses = []
        for _ in range(self.cfg["num_steps"]):
            with torch.enable_grad():
                loss_e = label_loss(self.model.get_llm_outpt(x, vt), y, msk)
                loss_a = sum(torch.exp(-o).mean() for o in self.new_extra_outputs)
                idx = int(self.rng.choice(len(self.sentences), 1)[0])
                x, vt = self.model.get_llm_input_embeds([self.sentences[idx]], None)   # overwrites the edit inputs (:173)
                self.model.get_llm_outpt(x, vt)
                loss_m = sum(torch.exp(o * (o > 0)).mean() for o in self.new_extra_outputs)
                loss = loss_e + loss_a * self.cfg["loss_a_lambda"] + loss_m * self.cfg["loss_m_lambda"]
                loss.backward()
            self.last_losses.append((float(loss_e), float(loss_a), float(loss_m)))
            opt.step()
            opt.zero_grad()
        self.K = [torch.cat([k.detach(), K], 1) for k, K in zip(ks, self.K)]
        self.B = [torch.cat([b.detach(), B], 0) for b, B in zip(bs, self.B)]
        self.V = torch.cat([v.detach(), self.V], 0)
        self.new = None
