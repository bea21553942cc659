#!/usr/bin/env python3
"""ViT attention at the bench's launch size (508 images x 16 heads x 257 tokens, dh 88, fused qkv rows) and CLIP-L (577 tokens, dh 64):
the ring kernel (9- / 8-wave tiles, key fold on / off) against the two-image LDS-DMA kernel, with the HBM floor beside each."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import devqa_amd  # noqa: E402,F401
from devqa_amd import lib  # noqa: E402


def t_us(fn, n=40):
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
    torch.manual_seed(0)
    for name, n_seq, T, H, dh in (("ViT-g", 508, 257, 16, 88), ("ViT-g", 127, 257, 16, 88), ("ViT-g 256", 508, 256, 16, 88), ("CLIP-L", 128, 577, 16, 64)):
        M = n_seq * T
        qkv = (torch.randn(M, 3 * H * dh, device="cuda") * 1.5).to(torch.bfloat16)
        q, k, v = qkv[:, :H * dh], qkv[:, H * dh:2 * H * dh], qkv[:, 2 * H * dh:]
        desc = torch.tensor([[i * T, T, 0, 0, i * T, T] for i in range(n_seq)], dtype=torch.int32, device="cuda")
        out = torch.zeros(M, H * dh, device="cuda", dtype=torch.bfloat16)
        fn = lambda: lib.attention(q, k, v, desc, n_seq, T, H, dh, dh ** -0.5, 0, out=out, self_full=True)
        gb = 4 * M * H * dh * 2 / 1e9
        print("%s: %d x %d heads x %d tokens, dh %d; q, k, v, out once = %.2f GB (%.0f us at 6 TB/s)" % (name, n_seq, H, T, dh, gb, gb / 6e3 * 1e6), flush=True)
        variants = (("ring, 9 waves", {"DEVQA_ATTENTION_NW": "9"}), ("ring, 8 waves", {"DEVQA_ATTENTION_NW": "8"}),
                    ("ring, 9 waves, no fold", {"DEVQA_ATTENTION_NW": "9", "DEVQA_ATTENTION_FOLD": "0"}),
                    ("ring, 8 waves, no fold", {"DEVQA_ATTENTION_NW": "8", "DEVQA_ATTENTION_FOLD": "0"}),
                    ("ring, 9 waves, 3 images", {"DEVQA_ATTENTION_NW": "9", "DEVQA_ATTENTION_NBUF": "3"}),
                    ("ring, 8 waves, 3 images", {"DEVQA_ATTENTION_NW": "8", "DEVQA_ATTENTION_NBUF": "3"}),
                    ("two-image DMA, 8 waves", {"DEVQA_ATTENTION_RING": "0"}), ("two-image DMA, 4 waves", {"DEVQA_ATTENTION_RING": "0", "DEVQA_ATTENTION_NW": "4"}),
                    ("register-staged", {"DEVQA_ATTENTION_DMA": "0"}))
        best, outs = {}, {}
        for rep in range(3):          # interleaved repeats, best of three: the clock the chip holds drifts over a run
            for label, env in variants:
                os.environ.update(env)
                if rep == 0:
                    out.zero_()
                    fn()
                    outs[label] = out.float().clone()
                us = t_us(fn, n=15)
                for k_ in env:
                    del os.environ[k_]
                best[label] = min(best.get(label, 1e30), us)
        ref = outs[variants[0][0]]
        for label, _ in variants:
            us = best[label]
            print("  %-26s %8.1f us  %6.2f TB/s  %6.1f TFLOP/s  max |diff to first| %.3g" % (
                label, us, gb / us * 1e3, 4.0 * n_seq * H * T * T * dh / us / 1e6, (outs[label] - ref).abs().max().item()), flush=True)


if __name__ == "__main__":
    main()
