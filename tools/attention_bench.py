#!/usr/bin/env python3
"""The MFMA attention kernel on the bench's two shapes: ViT-g (127 images x 16 heads x 257 tokens, dh 88, all-visible) and the OPT
decoder pack (60 cycles: 300 sequences of ~48 tokens x 32 heads, dh 80, causal), HIP-event averages + error vs fp32 torch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import devqa_amd  # noqa: E402,F401
from devqa_amd import lib  # noqa: E402


def t_us(fn, n=20, warm=600):
    for _ in range(warm):      # clocks ramp over ~100 ms of continuous load: the first launches after an idle period run 10-15 % slower
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
    torch.manual_seed(0)
    only = sys.argv[1] if len(sys.argv) > 1 else None      # "vit" / "opt": one shape (PMC passes)
    for name, n_seq, T, H, dh, causal in (("vit", 127, 257, 16, 88, 0), ("opt", 300, 48, 32, 80, 1)):
        if only and name != only:
            continue
        M = n_seq * T
        qkv = (torch.randn(M, 3 * H * dh, device="cuda") * 1.5).to(torch.bfloat16)
        q, k, v = qkv[:, :H * dh], qkv[:, H * dh:2 * H * dh], qkv[:, 2 * H * dh:]
        desc = torch.tensor([[i * T, T, 0, 0, i * T, T] for i in range(n_seq)], dtype=torch.int32, device="cuda")
        out = torch.zeros(M, H * dh, device="cuda", dtype=torch.bfloat16)
        fn = lambda: lib.attention(q, k, v, desc, n_seq, T, H, dh, dh ** -0.5, causal, out=out, self_full=not causal)
        fn()
        # reference on the first 3 sequences
        n = 3 * T
        qf = q[:n].float().view(3, T, H, dh).transpose(1, 2)
        kf = k[:n].float().view(3, T, H, dh).transpose(1, 2)
        vf = v[:n].float().view(3, T, H, dh).transpose(1, 2)
        sc = qf @ kf.transpose(-1, -2) * dh ** -0.5
        if causal:
            sc = sc + torch.full((T, T), float("-inf"), device="cuda").triu(1)
        ref = (torch.softmax(sc, -1) @ vf).transpose(1, 2).reshape(n, H * dh)
        err = float((out[:n].float() - ref).abs().max() / ref.abs().max())
        us = t_us(fn)
        flops = 4.0 * n_seq * H * T * T * dh * (0.5 if causal else 1.0)
        print("%-4s %d seq x %d tok x %d heads dh %d causal %d: %7.1f us  %6.1f TFLOP/s  rel err %.2e" % (name, n_seq, T, H, dh, causal, us, flops / us / 1e6, err), flush=True)
        base = out.clone()
        for var in ("DEVQA_ATTENTION_DBUF=1", "DEVQA_ATTENTION_QB=2", "DEVQA_ATTENTION_RESIDENT=1"):   # opt-in instantiations
            kx, vx = var.split("=")
            os.environ[kx] = vx
            try:
                out.zero_()
                fn()
                same = bool(torch.equal(out, base))
                print("     %-28s %7.1f us   bit-identical to the default: %s" % (var, t_us(fn), same), flush=True)
            finally:
                del os.environ[kx]


if __name__ == "__main__":
    main()
