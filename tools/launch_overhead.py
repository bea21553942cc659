#!/usr/bin/env python3
"""Host cost of one library call from Python (ctypes + wrapper bookkeeping): a tiny GEMM launched in a loop, GPU idle."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import devqa_amd  # noqa: E402,F401
from devqa_amd import lib  # noqa: E402

lib.load()
a = torch.randn(16, 64, device="cuda").to(torch.bfloat16)
w = torch.randn(128, 64, device="cuda").to(torch.bfloat16)
o = torch.empty(16, 128, device="cuda", dtype=torch.bfloat16)
x = torch.randn(64, 256, device="cuda")
g = torch.randn(256, device="cuda")
for name, fn in (("gemm", lambda: lib.gemm(a, w, out_bf16=o)), ("layernorm", lambda: lib.layernorm(x, g, g, 1e-5, want="bf16")),
                 ("_stream", lib._stream)):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    n = 3000
    for _ in range(n):
        fn()
    t1 = time.time()
    torch.cuda.synchronize()
    print("%-10s %.1f us per call (host)" % (name, (t1 - t0) / n * 1e6))
