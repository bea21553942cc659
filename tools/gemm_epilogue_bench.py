#!/usr/bin/env python3
"""How much of a short-K GEMM is prologue + epilogue: the ViT shapes at K = 64 (one K-tile) vs their real K,
with the epilogues the path uses (bias, bias+GELU, bias + fp32 residual in place)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import devqa_amd  # noqa: E402,F401
from devqa_amd import lib  # noqa: E402


def t_us(fn, n=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def main():
    lib.load()
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 16448
    for name, N, K in (("qkv", 4224, 1408), ("fc1", 6144, 1408), ("proj", 1408, 1408), ("fc2", 1408, 6144)):
        for k in (64, K):
            a = torch.randn(M, k, device="cuda").to(torch.bfloat16)
            w = (torch.randn(N, k, device="cuda") / k ** 0.5).to(torch.bfloat16)
            b = torch.randn(N, device="cuda")
            ob = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            x = torch.randn(M, N, device="cuda")
            r = {}
            r["none"] = t_us(lambda: lib.gemm(a, w, out_bf16=ob))
            r["bias"] = t_us(lambda: lib.gemm(a, w, b, out_bf16=ob))
            r["gelu"] = t_us(lambda: lib.gemm(a, w, b, act=lib.ACT_GELU, out_bf16=ob))
            r["relu"] = t_us(lambda: lib.gemm(a, w, b, act=lib.ACT_RELU, out_bf16=ob))
            r["resid_f32"] = t_us(lambda: lib.gemm(a, w, b, residual=x, out_f32=x))
            print("%-5s M=%d N=%d K=%5d " % (name, M, N, k) + " ".join("%s %7.1f" % kv for kv in r.items()), flush=True)


if __name__ == "__main__":
    main()
