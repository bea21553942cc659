#!/usr/bin/env python3
"""dH = dlogits . E (K = 50272, N = 2560) through devqa_gemm_bf16_splitk: 64 rows per launch vs the default row grouping."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import devqa_amd
from devqa_amd import lib
lib.load()
def t_us(fn, n=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
torch.manual_seed(0)
K, N = 50272, 2560
w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
for M in (48, 100, 180, 250):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    ref = (a[:8].float() @ w.float().T)
    res = {}
    for g in ("64", "default"):
        if g == "default":
            os.environ.pop("DEVQA_SPLITK_ROWS", None)
        else:
            os.environ["DEVQA_SPLITK_ROWS"] = g
        out = lib.gemm_rows_longk(a, w)
        err = float((out[:8] - ref).abs().max() / ref.abs().max())
        res[g] = (t_us(lambda: lib.gemm_rows_longk(a, w)), err)
    print(M, {k: "%.1f us err %.1e" % v for k, v in res.items()}, flush=True)
