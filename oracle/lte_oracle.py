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
