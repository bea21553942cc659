import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import devqa_amd
from devqa_amd import lib
lib.load()
torch.manual_seed(0)
H, dh = 5, 8
for (lens, name) in (([104, 30, 70, 129, 64, 65], "causal-own"),):
    starts, r = [], 0
    for n in lens:
        starts.append(r); r += n
    R = r
    for H, dh in ((5, 8), (32, 80)):
        d = H * dh
        qkv = (torch.randn(R, 3 * d, device="cuda") * 1.0).to(torch.bfloat16)
        desc = torch.tensor([[s, n, 0, 0, s, n] for s, n in zip(starts, lens)], dtype=torch.int32, device="cuda")
        q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
        # fp32 reference
        ref = torch.zeros(R, d, device="cuda")
        for s, n in zip(starts, lens):
            qf = q[s:s + n].float().view(n, H, dh).transpose(0, 1); kf = k[s:s + n].float().view(n, H, dh).transpose(0, 1); vf = v[s:s + n].float().view(n, H, dh).transpose(0, 1)
            sc = qf @ kf.transpose(-1, -2) * dh ** -0.5 + torch.full((n, n), float("-inf"), device="cuda").triu(1)
            ref[s:s + n] = (torch.softmax(sc, -1) @ vf).transpose(0, 1).reshape(n, d)
        for env in ({}, {"DEVQA_ATTENTION_DBUF": "1"}, {"DEVQA_ATTENTION_QB": "2"}):
            os.environ.update(env)
            outs = []
            for it in range(6):
                out = torch.full((R, d), 7.0, device="cuda", dtype=torch.bfloat16)
                lib.attention(q, k, v, desc, len(lens), max(lens), H, dh, dh ** -0.5, 1, out=out)
                torch.cuda.synchronize()
                outs.append(out.float())
            for kx in env: del os.environ[kx]
            dd = max(float((outs[0] - o).abs().max()) for o in outs[1:])
            err = float((outs[0] - ref).abs().max() / ref.abs().max())
            bad = torch.nonzero((outs[0] - outs[1]).abs().max(1).values > 0).flatten().tolist()
            print("H %d dh %d %-28s run-to-run max diff %.3e  rel err vs fp32 %.3e  bad rows %s" % (H, dh, env or "default", dd, err, bad[:12]), flush=True)
