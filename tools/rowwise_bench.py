#!/usr/bin/env python3
"""HBM-bound row kernels against the roofline: LayerNorm (fp32 in, bf16 out) at the ViT / OPT shapes of one bench step."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import devqa_amd  # noqa: E402,F401
from devqa_amd import lib  # noqa: E402


def t_us(fn, n=30):
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
    for M, D in ((20560, 1408), (6400, 2560), (2560, 768)):
        xs = [torch.randn(M, D, device="cuda") for _ in range(6)]      # rotate buffers: 6 x 116 MB > the 256-MiB Infinity Cache
        g, b = torch.randn(D, device="cuda"), torch.randn(D, device="cuda")
        i = [0]

        def f():
            i[0] = (i[0] + 1) % len(xs)
            lib.layernorm(xs[i[0]], g, b, 1e-5, want="bf16")
        us = t_us(f)
        by = M * D * 6
        print("layernorm M=%6d D=%5d  %7.1f us  %6.2f TB/s (read fp32 + write bf16 = %.0f MB)" % (M, D, us, by / us / 1e6, by / 1e6), flush=True)


if __name__ == "__main__":
    main()
