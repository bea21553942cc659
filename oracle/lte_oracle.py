"""CPU oracle for LTE_VL's inference path -- TEST INFRASTRUCTURE (only tests/, smoke() and bench.py's cpu_baseline may
import it).

Restates R/editor/vllm_editors/lte_vl/lte_vl.py:75-147 in plain PyTorch fp32 over any oracle model that offers
`get_llm_input_embeds` / `get_llm_outpt` (OracleBlip2, OracleLlava, OracleMiniGPT4): an edit stores the request, the LLM
input embeddings of "[Updated Information]{prompt} {target_new}\\n[Query]" with the request's image (:112-121) and the
sentence embedding of "{prompt} {target_new}" (:143-147); `get_llm_outpt` embeds the probe prompt carried in
`query_triple`, takes the cosine arg-max over the stored edits (:94-110) and, above `sim_threshold`, runs the decoder on
prefix ++ probe and drops the prefix rows (:75-92).

PINNED by the reference itself: tools/make_goldens_lte.py runs the reference's own `LTEvl` on the reference's BLIP-2 wrapper (the
`sentence_transformers` import, absent here and irrelevant to the arithmetic, is replaced in-process by a stub whose `encode` is the
bag-of-words function the tests pass as `encode`) and stores prefixes, retrieval pool and decisions, hook logits and results.json for
edit_n = 1 and 2 in tests/golden/tiny_lte_goldens.*; tests/test_oracle_lte.py::test_lte_oracle_matches_reference_goldens holds this
file to them (logits 1e-4, 48/48 evaluator entries).  LLaVA / MiniGPT-4 use the same class over their oracle models.  The sentence
encoder is an argument (`encode(list[str]) -> [n, d]`), as in the product.
"""
import torch
import torch.nn.functional as F


class OracleLTEvl:
    edit_sign = "[Updated Information]"
    query_sign = "\n[Query]"

    def __init__(self, model, encode, sim_threshold, retrieval_embed_dim):
        self.model, self.encode = model, encode
        self.sim_threshold, self.dim = sim_threshold, retrieval_embed_dim
        self.plain_get_llm_outpt = model.get_llm_outpt          # bound method of the class
        model.get_llm_outpt = self._get_llm_outpt               # instance attribute shadows it (the reference's wrap, :75-92)
        self.restore_to_original_model()

    def unhook(self):
        del self.model.get_llm_outpt

    def restore_to_original_model(self):  # :138-141
        self.requests, self.prefixes = [], []
        self.pool = torch.zeros(0, self.dim)

    def _embed(self, texts):
        return torch.as_tensor(self.encode(texts), dtype=torch.float32)

    def edit_prefix(self, request):  # :112-121
        p, t = request["prompt"], request["target_new"]
        if p[-1] != " " and t[0] != " ":
            t = " " + t
        return self.model.get_llm_input_embeds([self.edit_sign + p + t + self.query_sign], [request["image"]])[0]

    def edit_one_piece(self, request):  # :143-147
        self.requests.append(request)
        self.prefixes.append(self.edit_prefix(request))
        self.pool = torch.cat([self.pool, self._embed([request["prompt"] + " " + request["target_new"]])], 0)

    def retrieval(self, texts):  # :94-110
        assert len(texts) == 1
        sim = F.normalize(self._embed(texts), p=2, dim=1) @ F.normalize(self.pool, p=2, dim=1).T
        v, i = torch.max(sim, dim=1)
        if float(v[0]) > self.sim_threshold:
            return self.requests[int(i[0])], self.prefixes[int(i[0])], sim
        return None, None, sim

    def _get_llm_outpt(self, x, vt_range=None):  # :78-89
        if not self.requests:
            return self.plain_get_llm_outpt(x, vt_range)
        assert len(x["inputs_embeds"]) == 1
        prompt, _image, _target = x["query_triple"]
        _, prefix, _ = self.retrieval([prompt])
        if prefix is None:
            return self.plain_get_llm_outpt(x, vt_range)
        cat = {"attention_mask": torch.cat([prefix["attention_mask"], x["attention_mask"]], 1),
               "inputs_embeds": torch.cat([prefix["inputs_embeds"], x["inputs_embeds"]], 1)}     # :123-127
        return self.plain_get_llm_outpt(cat, None)[:, prefix["attention_mask"].shape[1]:]


# ---- training (R/editor/vllm_editors/lte_vl/lte_vl.py:152-233) -------------------------------------------------------------------
def logit_KL_loss(logits1, logits2, masks):  # lte_vl.py:254-265
    L = masks.shape[1]
    l1, l2 = logits1[:, -L:], logits2[:, -L:]
    kl = (torch.softmax(l1, 2) * (torch.log_softmax(l1, -1) - torch.log_softmax(l2, -1))).sum(2)
    return (kl * masks).sum() / masks.sum()


class OracleLTETrainer:
    """The reference's training step restated with autograd + torch.optim.Adam over an oracle model: `organize_batch_data` (:169-187)
    on a FROZEN copy of the weights (the reference's `vllm_proc_data`), `train_a_batch` (:203-233) on the trained ones.  Pinned on
    BLIP-2 by two steps of the reference's own loop (tests/test_oracle_lte.py::test_lte_oracle_training_matches_reference_goldens); the
    LLaVA / MiniGPT-4 GPU tests use it as the checker for the LLaMA-family backward."""

    def __init__(self, model, lm_prefix, lr, relia_lambda=1.0, gen_lambda=1.0, loc_lambda=1.0):
        from copy import copy
        from oracle.devqa_oracle import label_loss
        self.label_loss = label_loss
        self.model = model
        self.frozen = copy(model)                         # shares tokenizer / config, private weight dict
        self.frozen.w = {k: v.detach().clone() for k, v in model.w.items()}
        self.names = [n for n in model.w if n.startswith(lm_prefix)]
        for n in self.names:
            model.w[n] = model.w[n].detach().clone().requires_grad_(True)
        self.opt = torch.optim.Adam([model.w[n] for n in self.names], lr=lr)
        self.lam = (relia_lambda, gen_lambda, loc_lambda)

    @staticmethod
    def edit_prefix(model, request):  # :112-121
        p, t = request["prompt"], request["target_new"]
        if p[-1] != " " and t[0] != " ":
            t = " " + t
        return model.get_llm_input_embeds([OracleLTEvl.edit_sign + p + t + OracleLTEvl.query_sign], [request["image"]])[0]

    def organize_batch_data(self, d):
        f = self.frozen
        with torch.no_grad():
            prefix = self.edit_prefix(f, d["requests"][0])
            rel = f.prompts_imgs_target_to_xym([d["requests"][0]["prompt"]], [d["requests"][0]["image"]], [d["requests"][0]["target_new"]])
            gen = {k: f.prompts_imgs_target_to_xym([v[0]["prompt"]], [v[0]["image"]], [v[0]["target"]]) for k, v in d["generality"].items()}
            loc = {}
            for k, v in d["locality"].items():
                (x, vt), y, m = f.prompts_imgs_target_to_xym([v[0]["prompt"]], [v[0]["image"]], [v[0]["target"]])
                loc[k] = ((x, vt), f.get_llm_outpt(x, vt), m)
        return prefix, rel, gen, loc

    def train_a_batch(self, batch):
        prefix, rel, gen, loc = batch
        mdl = self.model

        def edited(x):
            cat = {"attention_mask": torch.cat([prefix["attention_mask"], x["attention_mask"]], 1),
                   "inputs_embeds": torch.cat([prefix["inputs_embeds"], x["inputs_embeds"]], 1)}
            return mdl.get_llm_outpt(cat, None)
        (x, vt), y, m = rel
        rel_loss = self.label_loss(edited(x), y, m)
        loss = rel_loss * self.lam[0]
        gen_losses, loc_losses = {}, {}
        for k, ((x, vt), y, m) in gen.items():
            g = self.label_loss(edited(x), y, m)
            gen_losses[k] = float(g.detach())
            loss = loss + g * self.lam[1]
        for k, ((x, vt), pre, m) in loc.items():
            l_ = (logit_KL_loss(pre, mdl.get_llm_outpt(x, vt), m) + logit_KL_loss(pre, edited(x), m)) / 2
            loc_losses[k] = float(l_.detach())
            loss = loss + l_ * self.lam[2]
        loss.backward()
        self.opt.step()
        self.opt.zero_grad()
        return float(loss.detach()), {"Reliability loss": float(rel_loss.detach()), "Generality loss": gen_losses, "Locality loss": loc_losses}
