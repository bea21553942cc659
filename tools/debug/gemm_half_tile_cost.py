#!/usr/bin/env python3
"""Cost of a half-width column tile of the ping-pong GEMM against a full one: the same M and K with N = 128 (every tile half-width),
N = 256 (every tile full) and N = 384 (one of each per row tile), bf16 output with bias and fp32 in-place residual."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import devqa_amd  # noqa: E402,F401
from devqa_amd import lib  # noqa: E402


def t_us(fn, n=20):
    for _ in range(5):
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
    lib.gemm_set_mode(22)
    M = int(os.environ.get("GEMM_M", 130556 * 2))
    for K in (1408, 6144):
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        for N in (128, 256, 384, 1408):
            w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
            b = torch.randn(N, device="cuda")
            ob = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            x = torch.randn(M, N, device="cuda")
            t1 = t_us(lambda: lib.gemm(a, w, b, out_bf16=ob))
            t2 = t_us(lambda: lib.gemm(a, w, b, residual=x, out_f32=x))
            print("K=%d N=%4d  bf16 out %8.1f us %5.0f TF   fp32 residual %8.1f us %5.0f TF" % (K, N, t1, 2.0 * M * N * K / t1 / 1e6, t2, 2.0 * M * N * K / t2 / 1e6), flush=True)
    lib.gemm_set_mode(0)


if __name__ == "__main__":
    main()
