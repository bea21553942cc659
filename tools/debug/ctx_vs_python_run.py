import os, sys, json
from copy import deepcopy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import devqa_amd
from devqa_amd.batched import BatchedEditEval
from devqa_amd.editor.vllm_editors.ft_vl.ft_vl import FTvl, FTvlConfig
from devqa_amd.editor.vllms_for_edit.blip2.blip2 import BLIP2OPTForEdit
gold = os.path.join(ROOT, "tests", "golden")
os.chdir(gold)
rec = json.load(open("evqa8_records.json"))["records"]
for mode in ("bf16",):
    vllm = BLIP2OPTForEdit(os.path.join(gold, "tiny_blip2"), "cuda:0", dtype=mode)
    cfg = FTvlConfig(edit_model_name="blip2-opt-2.7b", rewrite_module_tmp="language_model.model.decoder.layers.{}.fc2.weight",
                     layers=[1], num_steps=25, lr=1e-3, weight_decay=0, norm_constraint=False, batch_size=1)
    ed = FTvl(vllm, cfg, "cuda:0")
    def run():
        be = BatchedEditEval(ed, cycles_per_batch=3)
        be.keep_debug = True
        be.run([[deepcopy(r)] for r in rec[:3]], [[deepcopy(r)] for r in rec[:3]])
        torch.cuda.synchronize()
        return be.debug["pre_logits"].clone(), be.debug["post_logits"].clone(), be.debug["delta"][0].clone(), torch.from_numpy(be.last_losses.copy())
    res = {}
    for tag, abi in (("ctx1", "1"), ("ctx2", "1"), ("py1", "0"), ("py2", "0")):
        os.environ["DEVQA_PATH_ABI"] = abi
        res[tag] = run()
    for a, b in (("ctx1", "ctx2"), ("py1", "py2"), ("ctx1", "py1")):
        print(mode, a, b, ["%.3e" % float((u.float() - v.float()).abs().max()) for u, v in zip(res[a], res[b])], flush=True)
    d = (res["ctx1"][0] - res["py1"][0]).abs().max(1).values
    print("rows differing in pre_logits:", torch.nonzero(d > 0).flatten().tolist(), "of", d.numel())
