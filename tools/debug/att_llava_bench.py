#!/usr/bin/env python3
"""The LLaVA decoder pack of one 16-cycle batch as the batched engine launches it (shared-prefix packing: 64 prefix sequences of 577 rows,
causal; 192 probe texts of ~20 rows behind a 577-row visible prefix; 32 heads x dh 128): one launch over all descriptors vs the prefixes
and the texts as two launches (the texts then take the short-sequence kernel)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import devqa_amd  # noqa: E402,F401
from devqa_amd import lib  # noqa: E402


def t_us(fn, n=10, warm=30):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def main(cycles=16):
    lib.load()
    H, dh, P = 32, 128, 577
    rng = np.random.default_rng(0)
    desc, r = [], 0
    pre = []
    for c in range(cycles * 4):
        pre.append(r)
        desc.append([r, P, 0, 0, r, P])
        r += P
    n_pre = len(desc)
    for c in range(cycles):
        for p in range(12):
            n = int(rng.integers(14, 26))
            img = [0, 0, 0, 0, 1, 2, 2, 3, 3, None, None, None][p]
            desc.append([r, n, pre[c * 4 + img], P, r, n] if img is not None else [r, n, 0, 0, r, n])
            r += n
    R = r
    qkv = (torch.randn(R, 3 * H * dh, device="cuda") * 0.5).to(torch.bfloat16)
    q, k, v = qkv[:, :H * dh], qkv[:, H * dh:2 * H * dh], qkv[:, 2 * H * dh:]
    d_all = torch.tensor(desc, dtype=torch.int32, device="cuda")
    d_pre, d_txt = d_all[:n_pre].contiguous(), d_all[n_pre:].contiguous()
    out1 = torch.zeros(R, H * dh, device="cuda", dtype=torch.bfloat16)
    out2 = torch.zeros_like(out1)
    one = lambda: lib.attention(q, k, v, d_all, len(desc), P, H, dh, dh ** -0.5, 1, out=out1)

    def two():
        lib.attention(q, k, v, d_pre, n_pre, P, H, dh, dh ** -0.5, 1, out=out2)
        lib.attention(q, k, v, d_txt, len(desc) - n_pre, 32, H, dh, dh ** -0.5, 1, out=out2)
    one(); two()
    torch.cuda.synchronize()
    print("rows %d, %d prefix + %d text sequences; outputs identical: %s" % (R, n_pre, len(desc) - n_pre, bool(torch.equal(out1, out2))))
    flops = 4.0 * H * dh * (n_pre * P * P / 2 + sum(d[1] * (d[3] + d[1] / 2) for d in desc[n_pre:]))
    for name, fn in (("one launch", one), ("prefixes | texts", two), ("prefixes only", lambda: lib.attention(q, k, v, d_pre, n_pre, P, H, dh, dh ** -0.5, 1, out=out2)),
                     ("texts only (max_q 32)", lambda: lib.attention(q, k, v, d_txt, len(desc) - n_pre, 32, H, dh, dh ** -0.5, 1, out=out2)),
                     ("texts only (max_q 577)", lambda: lib.attention(q, k, v, d_txt, len(desc) - n_pre, P, H, dh, dh ** -0.5, 1, out=out2))):
        us = t_us(fn)
        print("%-26s %8.1f us   (%.0f TFLOP/s on the whole pack's %.1f GFLOP)" % (name, us, flops / us / 1e6, flops / 1e9))
    ref = out1.clone()
    for var in ({"DEVQA_ATTENTION_NW": "4"}, {"DEVQA_ATTENTION_DMA": "0"}, {"DEVQA_ATTENTION_RING": "2"}, {"DEVQA_ATTENTION_RING": "2", "DEVQA_ATTENTION_NBUF": "3"}):
        os.environ.update(var)
        out1.zero_()
        one()
        torch.cuda.synchronize()
        same = bool(torch.equal(out1, ref))
        print(var, "prefixes only %8.1f us, one launch %8.1f us (identical to the default: %s, max diff %.3g)" % (
            t_us(lambda: lib.attention(q, k, v, d_pre, n_pre, P, H, dh, dh ** -0.5, 1, out=out2)), t_us(one), same,
            (out1.float() - ref.float()).abs().max().item()))
        for k_ in var:
            del os.environ[k_]


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:2]])
