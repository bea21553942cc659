#!/usr/bin/env python3
"""Goldens for FT_VL on edit targets OTHER than the last layer's fc2 matrix, from the REFERENCE's own FTvl (build container only; the
reference does not travel).  The selection is its substring rule (R/editor/vllm_editors/ft_vl/ft_vl.py:31-36,83-88: every parameter whose
name contains rewrite_module_tmp.format(layer) for a layer of `layers`), on the tiny BLIP-2 of tests/golden/tiny_blip2:
  A  fc1.weight of decoder layers 0 and 1
  B  "self_attn.q_proj" of layer 1 (weight AND bias: a row block of the fused q|k|v operand on the HIP side)
  C  "final_layer_norm" + "fc2" of layer 0 via the template "layers.{}.f" (fc1, fc2, final_layer_norm: weights and biases)
  D  "qformer" -- the alternative the reference's own yaml carries as a comment (R/configs/ft_vl/blip2-opt-2.7b.yaml:9): every Q-Former
     parameter (self- and cross-attention, query FFN, every LayerNorm), gradients through the whole decoder and the language projection
  E  every parameter of ViT encoder layer 0 ("vision_model.encoder.layers.{}."): gradients additionally through the Q-Former's cross-attention
     keys / values, the ViT's post-LayerNorm and both encoder layers
  F  "post_layernorm" (the ViT's last LayerNorm alone: no encoder layer is trained)
  G  "vision_model": the whole tower incl. the patch convolution, the class and position embeddings
  H  "language_projection" (weight + bias)      I  "query_tokens" (the learned queries)
  J  "language_model": the whole decoder incl. the tied token embedding (head side and lookup side), the learned positions, the final norm
  K  "" -- the empty template selects EVERY parameter of the model      L  "layer_norm": LayerNorms of the ViT and of the decoder together
Stores per case and request: per-step losses, step count, the delta of every selected parameter.  DATA only."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_goldens as G  # noqa: E402  (sys.path + the two import stubs)

import numpy as np  # noqa: E402
import torch  # noqa: E402

CASES = [("A", "language_model.model.decoder.layers.{}.fc1.weight", [0, 1]),
         ("B", "language_model.model.decoder.layers.{}.self_attn.q_proj", [1]),
         ("C", "language_model.model.decoder.layers.{}.f", [0]),
         ("D", "qformer", [0]),
         ("E", "vision_model.encoder.layers.{}.", [0]),
         ("F", "post_layernorm", [0]),
         ("G", "vision_model", [0]),
         ("H", "language_projection", [0]),
         ("I", "query_tokens", [0]),
         ("J", "language_model", [0]),
         ("K", "", [0]),
         ("L", "layer_norm", [0])]


def main():
    from editor.vllms_for_edit.blip2.blip2 import BLIP2OPTForEdit
    from editor.vllm_editors.ft_vl import ft_vl as ref_ft
    os.chdir(G.GOLD)
    records = json.load(open(os.path.join(G.GOLD, "evqa8_records.json")))["records"]
    vllm = BLIP2OPTForEdit(os.path.join(G.GOLD, "tiny_blip2"), "cpu")
    out_json, out_npz = {"cases": []}, {}
    for tag, tmp, layers in CASES:
        cfg = ref_ft.FTvlConfig(edit_model_name="blip2-opt-2.7b", rewrite_module_tmp=tmp, layers=layers, num_steps=25, lr=1e-3, weight_decay=0,
                                norm_constraint=False, batch_size=1)
        ed = ref_ft.FTvl(vllm, cfg, "cpu")
        entry = {"tag": tag, "rewrite_module_tmp": tmp, "layers": layers, "names": sorted(ed.original_w.keys()), "requests": []}
        for ri in (0, 2):
            req = records[ri]["requests"][0]
            losses = []
            orig = ref_ft.AverageMeter.update

            def rec(self, val, n=1, _l=losses, _o=orig):
                _l.append(float(val))
                return _o(self, val, n)
            ref_ft.AverageMeter.update = rec
            try:
                deltas = ed.execute_ft([req])
            finally:
                ref_ft.AverageMeter.update = orig
            ed.restore_to_original_model()
            for n, d in deltas.items():
                out_npz["%s_%d_%s" % (tag, ri, n)] = d.detach().cpu().numpy().astype(np.float32)
            entry["requests"].append({"record": ri, "request": req, "losses": losses, "steps": len(losses)})
        out_json["cases"].append(entry)
        print(tag, entry["names"], [r["steps"] for r in entry["requests"]], [r["losses"][0] for r in entry["requests"]])
    np.savez_compressed(os.path.join(G.GOLD, "tiny_ft_general_goldens.npz"), **out_npz)
    json.dump(out_json, open(os.path.join(G.GOLD, "tiny_ft_general_goldens.json"), "w"), indent=1)


if __name__ == "__main__":
    torch.manual_seed(0)
    main()
