"""TP_VL (T-Patcher) editor on the HIP path: drop-in for R/editor/vllm_editors/tp_vl/tp_vl.py:14-196 -- same config
dataclass and plugin methods.  Every edit appends ONE patch neuron (key, bias, value) to the FFN of `edit_layer` and trains
it for `num_steps` Adam steps on  loss_e + lambda_a * mean(exp(-pre_new)) + lambda_m * mean(exp(pre_new * (pre_new > 0)))
(memory loss on one randomly drawn text per step).

What changes is how a step is computed: nothing but the new neuron changes during an edit, so the decoder runs ONCE per
edit over the edit sequence and the `num_steps` memory texts (drawn up front, in the reference's draw order); each step
then works on the frozen rows of the edited layer -- `devqa_tp_neuron_fwd` (new pre-activations, patched label rows) ->
lm_head rows -> NLL / dlogits -> dH -> final LayerNorm backward -> `devqa_tp_neuron_bwd` (all three gradients and the two
auxiliary losses in one launch) -> Adam.

Observable quirk of the reference kept (tp_vl.py:162-173): its loop reuses the names `input_embeds, vt_range` for the memory
text, so from the second step on the "edit" forward runs on the PREVIOUS step's memory text, scored against the edit labels on
its last-L rows; only step 0 sees the edit prompt (and image).

The reference loads the memory texts with `datasets.load_dataset('data/wikitext/wikitext-103-raw-v1')` and filters them
(:41-44); here they are passed in (`locality_texts`, a list of strings, or a text file with one per line) and filtered by
the same rule.  Supported on this path, the LAST decoder layer's FFN in both shapes the shipped configs select:
  * OPT (R/configs/tp_vl/blip2-opt-2.7b.yaml): in = fc1, out = fc2; neuron = relu(h.k + b) * v;
  * LLaMA (R/configs/tp_vl/llava-v1.5-7b.yaml, minigpt-4-vicuna-7b.yaml): in = gate_proj AND up_proj, out = down_proj; the
    hooks extend both projections' outputs, so the neuron is silu(h.k_gate + b_gate) * (h.k_up + b_up) * v and loss_a / loss_m
    are summed over the two in-layers (tp_vl.py:164-177) -- `devqa_tp_gated_neuron_fwd/bwd`.
"""
import re
from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np
import torch

from ...base import BaseConfig
from ..base import VLLMBaseEditor
from .... import lib


@dataclass
class TPvlConfig(BaseConfig):
    edit_model_name: str
    edit_layer: int
    num_steps: int
    lr: float
    loss_a_lambda: float
    loss_m_lambda: float
    weight_decay: float
    mlp_in_module_tmps: List[str]
    mlp_out_module_tmps: List[str]


def filter_locality_texts(texts):
    """tp_vl.py:42-44: more than 20 space-separated words, not a '= heading =' line."""
    return [t for t in texts if len(t.split(" ")) > 20 and not re.search(r"^[\s\n]*=", t) and not re.search(r"=[\s\n]*$", t)]


class TPvl(VLLMBaseEditor):
    def __init__(self, vllm, config: TPvlConfig, device="cuda", verbose=False, locality_data_path="data/wikitext/wikitext-103-raw-v1",
                 locality_texts=None, rng=None):
        super().__init__(vllm, device)
        self.cfg = config
        self.verbose = verbose
        eng = self.vllm.engine
        self.dev = eng.dev
        ins = [t.format(config.edit_layer) for t in config.mlp_in_module_tmps]
        outs = [t.format(config.edit_layer) for t in config.mlp_out_module_tmps]
        if len(outs) != 1 or len(ins) not in (1, 2):
            raise NotImplementedError("native TP_VL patches one FFN: fc1 -> fc2, or gate_proj + up_proj -> down_proj")
        self.gated = len(ins) == 2
        ok = (ins[0].endswith(".mlp.gate_proj") and ins[1].endswith(".mlp.up_proj") and outs[0].endswith(".mlp.down_proj")) \
            if self.gated else (ins[0].endswith(".fc1") and outs[0].endswith(".fc2"))
        if config.edit_layer != eng.edit_layer or not ok:
            raise NotImplementedError("native TP_VL patches the FFN of the LAST decoder layer (got %s -> %s)" % (ins, outs))
        self.fc1, self.fc2 = ins[0], outs[0]
        self.d_in = self.vllm.model.get(self.fc1 + ".weight").shape[1]
        self.d_out = self.vllm.model.get(self.fc2 + ".weight").shape[0]
        if locality_texts is None:
            with open(locality_data_path) as f:     # one text per line
                locality_texts = [ln.rstrip("\n") for ln in f]
        self.locality_data = np.array(filter_locality_texts(list(locality_texts)))
        self.rng = rng if rng is not None else np.random.default_rng(None)
        self.restore_to_original_model()
        self.last_losses: List[Tuple[float, float, float]] = []

    def name_of_editor_and_model(self) -> Tuple[str, str]:
        return "tp_vl", self.cfg.edit_model_name

    def if_can_batch_edit(self):
        return False

    def restore_to_original_model(self):
        # one ROW per neuron (newest first); gated FFN: K [n, 2, d_in] (gate key, up key), B [n, 2]
        self.K = torch.zeros((0, 2, self.d_in) if self.gated else (0, self.d_in), dtype=torch.float32, device=self.dev)
        self.B = torch.zeros((0, 2) if self.gated else (0,), dtype=torch.float32, device=self.dev)
        self.V = torch.zeros((0, self.d_out), dtype=torch.float32, device=self.dev)
        self._install()

    def edit_batch(self, requests: List[Dict]):
        raise

    def _install(self):
        eng = self.vllm.engine
        n = self.K.shape[0]
        if n == 0:
            eng.extra_neurons = {}
            return
        npad = (n + 7) // 8 * 8
        op = (lambda t: lib.cast_f32_bf16(t.contiguous())) if eng.adt == torch.bfloat16 else (lambda t: t.contiguous())
        if self.gated:
            KGU = torch.zeros((2 * npad, self.d_in), dtype=torch.float32, device=self.dev)
            BGU = torch.zeros((2 * npad,), dtype=torch.float32, device=self.dev)
            VT = torch.zeros((self.d_out, npad), dtype=torch.float32, device=self.dev)
            KGU[:n], KGU[npad:npad + n] = self.K[:, 0], self.K[:, 1]
            BGU[:n], BGU[npad:npad + n] = self.B[:, 0], self.B[:, 1]
            VT[:, :n] = self.V.t()
            eng.extra_neurons = {self.cfg.edit_layer: {"KGU": op(KGU), "BGU": BGU, "VT": op(VT)}}
            return
        K = torch.zeros((npad, self.d_in), dtype=torch.float32, device=self.dev)
        B = torch.zeros((npad,), dtype=torch.float32, device=self.dev)
        VT = torch.zeros((self.d_out, npad), dtype=torch.float32, device=self.dev)
        K[:n], B[:n], VT[:, :n] = self.K, self.B, self.V.t()
        eng.extra_neurons = {self.cfg.edit_layer: {"K": op(K), "B": B, "VT": op(VT)}}

    @torch.no_grad()
    def edit_one_piece(self, request: Dict) -> None:
        """request = {'image': path|None, 'prompt': str, 'target_new': str, ...}"""
        cfg, eng, dev, vllm = self.cfg, self.vllm.engine, self.dev, self.vllm
        (x, vt), y, msk = vllm.prompts_imgs_target_to_xym([request["prompt"]], [request["image"]], [request["target_new"]])
        L = y.shape[1]
        # memory texts of all steps, drawn in the reference's order (one draw per step, :170)
        draws = [int(self.rng.choice(len(self.locality_data), 1)[0]) for _ in range(cfg.num_steps)]
        seqs = [x["inputs_embeds"][0].to(torch.float32)]
        for i in draws:
            lx, _ = vllm.get_llm_input_embeds([str(self.locality_data[i])], None)
            seqs.append(lx["inputs_embeds"][0].to(torch.float32))
        tmax = max(s.shape[0] for s in seqs)
        emb = torch.zeros((len(seqs), tmax, seqs[0].shape[1]), dtype=torch.float32, device=dev)
        am = torch.zeros((len(seqs), tmax), dtype=torch.int32, device=dev)
        for b, s in enumerate(seqs):
            if s.shape[0] < L:
                raise RuntimeError("a memory text is shorter than the edit's label window")
            emb[b, :s.shape[0]] = s
            am[b, :s.shape[0]] = 1
        ps = eng.pack_from_embeds(emb, am)
        x_mid, a, h = eng.decoder_layers(ps, stop_before_fc2=True, return_h=True)
        h32 = h.to(torch.float32)
        # layer output rows WITHOUT the new neuron, for the label windows of the 25 edit-role sequences (0 .. num_steps-1)
        rows = []
        for b in range(cfg.num_steps):
            T = seqs[b].shape[0]
            rows += [b * tmax + T - L + j for j in range(L)]
        ridx = torch.tensor(rows, dtype=torch.int32, device=dev)
        ybase = lib.gather_rows(x_mid, ridx)
        lib.gemm(lib.gather_rows(a, ridx), eng._w(self.fc2 + ".weight"), None if self.gated else eng._p(self.fc2 + ".bias"),
                 residual=ybase, out_f32=ybase)
        eng.add_extra_neurons(cfg.edit_layer, lib.gather_rows(h, ridx), ybase)          # earlier edits' neurons
        ybase = ybase.view(cfg.num_steps, L, self.d_out)
        labels = y[0].to(dev, torch.int32).contiguous()
        m = msk[0].to(dev, torch.float32)
        coef = (m / m.sum()).contiguous()
        k = torch.zeros((2, self.d_in) if self.gated else (self.d_in,), dtype=torch.float32, device=dev)
        bb = torch.zeros((2 if self.gated else 1,), dtype=torch.float32, device=dev)
        nfwd, nbwd = (lib.tp_gated_neuron_fwd, lib.tp_gated_neuron_bwd) if self.gated else (lib.tp_neuron_fwd, lib.tp_neuron_bwd)
        v = torch.zeros((self.d_out,), dtype=torch.float32, device=dev)
        mom = [torch.zeros_like(t) for t in (k, bb, v, k, bb, v)]
        losses = []
        for i in range(cfg.num_steps):
            T_e = seqs[i].shape[0]
            h_e = h32[i * tmax:i * tmax + T_e].contiguous()                # step 0: the edit sequence; then memory text i-1
            lab = torch.arange(T_e - L, T_e, dtype=torch.int32, device=dev)
            pre, yrows = nfwd(h_e, k, bb, lab, v, ybase[i].contiguous())
            logits = eng.lm_head(yrows)
            _, nll, dlog = lib.vocab_rows(logits, labels, coef, want_argmax=False, want_nll=True, want_dlogits=True,
                                          dlogits_dtype=eng.adt)
            dH = lib.gemm_rows_longk(dlog, vllm.model.embed_T)
            dy = eng.final_norm_bwd(yrows, dH)
            T_m = seqs[i + 1].shape[0]
            h_m = h32[(i + 1) * tmax:(i + 1) * tmax + T_m].contiguous()
            gk, gb, gv, la_lm = nbwd(h_e, pre, lab, dy, h_m, k, bb, v, cfg.loss_a_lambda, cfg.loss_m_lambda, cfg.weight_decay)
            losses.append((nll, coef, la_lm))
            for p_, g_, m1, m2 in ((k, gk, mom[0], mom[3]), (bb, gb, mom[1], mom[4]), (v, gv, mom[2], mom[5])):
                lib.adam_step_(p_, g_, m1, m2, cfg.lr, i + 1)
        self.last_losses = [(float((n_ * c_).sum()), float(l_[0]), float(l_[1])) for n_, c_, l_ in losses]
        if self.verbose:
            for i, (le, la, lm) in enumerate(self.last_losses):
                print(i, le + la * cfg.loss_a_lambda + lm * cfg.loss_m_lambda, "\n  loss_e: ", le, "\n  loss_a: ", la, "\n  loss_m: ", lm)
        self.K = torch.cat([k.unsqueeze(0), self.K], 0)                    # the new neuron goes in FRONT (:139-145)
        self.B = torch.cat([bb.unsqueeze(0) if self.gated else bb, self.B], 0)
        self.V = torch.cat([v.unsqueeze(0), self.V], 0)
        self._install()
