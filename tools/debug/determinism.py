import os, sys, json
from copy import deepcopy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import devqa_amd
from devqa_amd import lib
from devqa_amd.batched import BatchedEditEval
from devqa_amd.editor.vllm_editors.ft_vl.ft_vl import FTvl, FTvlConfig
from devqa_amd.editor.vllms_for_edit.blip2.blip2 import BLIP2OPTForEdit
gold = os.path.join(ROOT, "tests", "golden")
os.chdir(gold)
rec = json.load(open("evqa8_records.json"))["records"]
os.environ["DEVQA_PATH_ABI"] = "0"
vllm = BLIP2OPTForEdit(os.path.join(gold, "tiny_blip2"), "cuda:0", dtype="bf16")
cfg = FTvlConfig(edit_model_name="blip2-opt-2.7b", rewrite_module_tmp="language_model.model.decoder.layers.{}.fc2.weight",
                 layers=[1], num_steps=25, lr=1e-3, weight_decay=0, norm_constraint=False, batch_size=1)
ed = FTvl(vllm, cfg, "cuda:0")
eng = vllm.engine
be = BatchedEditEval(ed, cycles_per_batch=3)
# capture the packed sequences of stage A
cap = {}
orig_pack = eng.pack_from_tokens
def pack(seqs, img_tokens, share_prefix=False):
    ps = orig_pack(seqs, img_tokens, share_prefix)
    cap["ps"] = ps; cap["seqs"] = seqs; cap["img"] = img_tokens.clone()
    return ps
eng.pack_from_tokens = pack
outs = []
for k in range(3):
    c = be._stage_a([deepcopy(r) for r in rec[:3]], [deepcopy(r) for r in rec[:3]])
    torch.cuda.synchronize()
    outs.append((cap["img"], cap["ps"].x.clone(), c["a_tail"].clone(), c["resid_tail"].clone()))
for nm, i in (("img_tokens", 0), ("x_after_layers", 1), ("a_tail", 2), ("resid_tail", 3)):
    print(nm, "run0 vs run1 %.3e  run1 vs run2 %.3e" % (float((outs[0][i].float() - outs[1][i].float()).abs().max()), float((outs[1][i].float() - outs[2][i].float()).abs().max())))
ps = cap["ps"]
d = (outs[0][1] - outs[1][1]).abs().max(1).values.cpu().numpy()
desc = ps.desc.cpu().numpy()
bad = set(np.nonzero(d > 0)[0].tolist())
for s, row in enumerate(desc):
    rows = set(range(row[0], row[0] + row[1]))
    if rows & bad:
        print("seq", s, "desc", row.tolist(), "bad rows", len(rows & bad), "first bad offset", min(rows & bad) - row[0])
print("max_len", ps.max_len, "n_seq", len(desc))
# one decoder pass repeated on the SAME packed input: which op is unstable?
x0 = orig_pack(cap["seqs"], cap["img"], True)
h1 = []
for k in range(2):
    ps2 = orig_pack(cap["seqs"], cap["img"], True)
    d_, H = eng.t["hidden_size"], eng.t["num_attention_heads"]
    h = eng._ln(ps2.x, "language_model.model.decoder.layers.0.self_attn_layer_norm.weight", "language_model.model.decoder.layers.0.self_attn_layer_norm.bias", 1e-5)
    qkv = lib.gemm(h, eng.m.fused_qkv_w["0"], eng.m.fused_qkv_b["0"])
    att = lib.attention(qkv[:, :d_], qkv[:, d_:2 * d_], qkv[:, 2 * d_:], ps2.desc, ps2.desc.shape[0], ps2.max_len, H, d_ // H, (d_ // H) ** -0.5, 1)
    h1.append((ps2.x.clone(), h.clone(), qkv.clone(), att.clone()))
for nm, i in (("x", 0), ("ln", 1), ("qkv", 2), ("att", 3)):
    print("layer0", nm, "%.3e" % float((h1[0][i].float() - h1[1][i].float()).abs().max()))
