#!/usr/bin/env python3
"""What the ragged ends of ViT-g's 257 tokens cost the MFMA attention kernel: the same launch at T = 256 (4 full tiles / chunks),
257 and 320, plus per-T scaling (fixed cost per workgroup)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
    torch.manual_seed(0)
    H, dh, n_seq = 16, 88, 127
    for T in (64, 128, 192, 256, 257, 272, 320):
        M = n_seq * T
        qkv = (torch.randn(M, 3 * H * dh, device="cuda") * 1.5).to(torch.bfloat16)
        q, k, v = qkv[:, :H * dh], qkv[:, H * dh:2 * H * dh], qkv[:, 2 * H * dh:]
        desc = torch.tensor([[i * T, T, 0, 0, i * T, T] for i in range(n_seq)], dtype=torch.int32, device="cuda")
        out = torch.zeros(M, H * dh, device="cuda", dtype=torch.bfloat16)
        fn = lambda: lib.attention(q, k, v, desc, n_seq, T, H, dh, dh ** -0.5, 0, out=out, self_full=True)
        res, outs = [], []
        for env in ({"DEVQA_ATTENTION_DMA": "0"}, {}, {"DEVQA_ATTENTION_NW": "8"}):
            os.environ.update(env)
            out.zero_()
            fn()
            outs.append(out.clone())
            res.append(t_us(fn))
            for k_ in env:
                del os.environ[k_]
        ex = []
        if os.environ.get("ATT_EXP"):
            os.environ["DEVQA_ATTENTION_DMA"] = "0"
            for e in ("1", "2", "3", "4"):      # timing-only variants: no loads in the loop / no exp / no barriers / no LDS refill
                os.environ["DEVQA_ATTENTION_EXP"] = e
                ex.append(t_us(fn))
            del os.environ["DEVQA_ATTENTION_EXP"], os.environ["DEVQA_ATTENTION_DMA"]
        print("T %3d: register-staged %7.1f us, LDS-DMA %7.1f us (identical %s), LDS-DMA 8 waves %7.1f us (identical %s); best %6.1f TFLOP/s algorithmic" % (
            T, res[0], res[1], bool(torch.equal(outs[0], outs[1])), res[2], bool(torch.equal(outs[0], outs[2])),
            4.0 * n_seq * H * T * T * dh / min(res) / 1e6), flush=True)
        if ex:
            print("       timing-only: no loads %.1f  no exp %.1f  no barriers %.1f  no refill %.1f" % tuple(ex), flush=True)


if __name__ == "__main__":
    main()
