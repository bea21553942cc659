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


def llama_rows():
    """RoPE / SwiGLU / RMSNorm at one LLaVA-7B bench step (20.8k rows)."""
    R, d, F, H = 20800, 4096, 11008, 32
    qkv = torch.randn(R, 3 * d, device="cuda").to(torch.bfloat16)
    pos = torch.arange(R, dtype=torch.int32, device="cuda") % 600
    us = t_us(lambda: lib.rope_(qkv[:, :2 * d], pos, 2 * H, 128, 10000.0))
    print("rope    R=%d 2x%d heads x 128   %7.1f us  %5.2f TB/s" % (R, H, us, R * 2 * d * 2 * 2 / us / 1e6), flush=True)
    gu = torch.randn(R, 2 * F, device="cuda").to(torch.bfloat16)
    us = t_us(lambda: lib.swiglu(gu))
    print("swiglu  R=%d F=%d          %7.1f us  %5.2f TB/s" % (R, F, us, R * 3 * F * 2 / us / 1e6), flush=True)
    x = torch.randn(R, d, device="cuda")
    w = torch.randn(d, device="cuda")
    us = t_us(lambda: lib.rmsnorm(x, w, 1e-5, want="bf16"))
    print("rmsnorm R=%d D=%d            %7.1f us  %5.2f TB/s" % (R, d, us, R * d * 6 / us / 1e6), flush=True)


if __name__ == "__main__":
    llama_rows() if len(sys.argv) > 1 and sys.argv[1] == "llama" else main()
