"""smoke: ONE small invocation of the hot path on cuda:0 (two edit+eval cycles on the tiny BLIP-2
fixture through the batched HIP engine), checked against the CPU oracle run on the same inputs.
The oracle is only the checker here; the product path above it never touches it."""
import json
import os
import sys
from copy import deepcopy

import numpy as np


def run():
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    gold = os.path.join(root, "tests", "golden")
    from .batched import BatchedEditEval
    from .editor.vllm_editors.ft_vl.ft_vl import FTvl, FTvlConfig
    from .editor.vllms_for_edit.blip2.blip2 import BLIP2OPTForEdit
    from oracle import devqa_oracle as O
    old = os.getcwd()
    os.chdir(gold)
    try:
        rec = json.load(open("evqa8_records.json"))["records"][:2]
        vllm = BLIP2OPTForEdit(os.path.join(gold, "tiny_blip2"), "cuda:0", dtype="fp32")
        cfg = FTvlConfig(edit_model_name="blip2-opt-2.7b",
                         rewrite_module_tmp="language_model.model.decoder.layers.{}.fc2.weight", layers=[1],
                         num_steps=25, lr=1e-3, weight_decay=0, norm_constraint=False, batch_size=1)
        ed = FTvl(vllm, cfg, "cuda:0")
        be = BatchedEditEval(ed, cycles_per_batch=2)
        res = be.run([[deepcopy(r)] for r in rec], [[deepcopy(r)] for r in rec])
        torch.cuda.synchronize()
        # ---- oracle (CPU) on the same inputs ----
        om = O.OracleBlip2.from_pretrained_dir(os.path.join(gold, "tiny_blip2"))
        oed = O.OracleFTvl(om, [1], cfg.rewrite_module_tmp)
        ores, _ = O.evaluate_sequential_edit(om, oed, rec, 1)
        n = 0
        for sr, so in zip(res, ores):
            r, o = sr[0], so[0]
            assert abs(r["reliability"][0]["acc"] - o["reliability"][0]["acc"]) < 1e-6
            for sec in ("generality", "locality"):
                for sub in o[sec]:
                    a, b = r[sec][sub][0], o[sec][sub][0]
                    assert abs(a["acc"] - b["acc"]) < 1e-6, (sec, sub, a, b)
                    assert a["predict_after_edit"] == b["predict_after_edit"]
                    n += 1
        assert np.all(be.last_steps == 25)
        print("smoke OK: 2 cycles, %d probes identical to the CPU oracle; mean final FT loss %.4f"
              % (n + 2, float(be.last_losses[:, -1].mean())))
    finally:
        os.chdir(old)
