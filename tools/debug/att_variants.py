#!/usr/bin/env python3
"""Which attention variant differs from which (debugging aid for tests/test_ops_gpu.py::test_attention)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import devqa_amd  # noqa: E402,F401
from devqa_amd import lib  # noqa: E402

lib.load()
g = torch.Generator().manual_seed(11)
H, dh = 32, 128
T = [598, 46, 130]
tot = sum(T)
q, k, v = (torch.randn(tot, H * dh, generator=g).to(torch.bfloat16).cuda() for _ in range(3))
st = np.cumsum([0] + T)
desc = torch.tensor([(int(st[i]), T[i], 0, 0, int(st[i]), T[i]) for i in range(3)], dtype=torch.int32, device="cuda")
outs = {}
for name, env in (("reg", {"DEVQA_ATTENTION_DMA": "0"}), ("dma4", {"DEVQA_ATTENTION_DMA": "1", "DEVQA_ATTENTION_NW": "4"}),
                  ("dma8", {"DEVQA_ATTENTION_DMA": "1", "DEVQA_ATTENTION_NW": "8"})):
    os.environ.update(env)
    rs = [lib.attention(q, k, v, desc, 3, max(T), H, dh, dh ** -0.5, 1).clone() for _ in range(4)]
    torch.cuda.synchronize()
    for k_ in env:
        del os.environ[k_]
    print(name, "deterministic:", all(torch.equal(rs[0], r) for r in rs), "nan:", bool(torch.isnan(rs[0].float()).any()))
    outs[name] = rs[0]
for a in ("dma4", "dma8"):
    dlt = (outs[a].float() - outs["reg"].float()).abs()
    rows = torch.nonzero(dlt.max(1).values > 0).flatten()
    print(a, "vs reg: max diff %.4g, rows differing %d" % (float(dlt.max()), len(rows)), rows[:20].tolist(), rows[-5:].tolist())
    cols = torch.nonzero(dlt.max(0).values > 0).flatten()
    print("   cols differing", len(cols), cols[:10].tolist())
