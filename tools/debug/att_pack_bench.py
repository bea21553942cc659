#!/usr/bin/env python3
"""Attention on a decoder probe pack shaped like a 127-cycle batch of the bench (per cycle: 4 image prefixes of 32 rows, 13 text
sequences of ~17 rows, 11 of them behind one of the prefixes), OPT heads (32 x 80): the default (single-image two-wave LDS-DMA
kernel for causal packs of short sequences) against DEVQA_ATTENTION_SHORT=0 (register-staged 64-query tiles), in one process."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import devqa_amd  # noqa: E402,F401
from devqa_amd import lib  # noqa: E402


def t_us(fn, n=20, warm=300):
    for _ in range(warm):
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
    rng = np.random.default_rng(0)
    cycles = int(os.environ.get("PACK_CYCLES", 127))
    H, dh = (32, 80) if os.environ.get("PACK_MODEL", "opt") == "opt" else (32, 128)
    desc, pos = [], 0
    for _ in range(cycles):
        pre = []
        for _ in range(4):
            desc.append((pos, 32, 0, 0, pos, 32))
            pre.append(pos)
            pos += 32
        for i in range(13):
            n = int(np.clip(round(rng.normal(17, 3)), 6, 32))
            if i < 11:
                desc.append((pos, n, pre[i % 4], 32, pos, n))
            else:
                desc.append((pos, n, 0, 0, pos, n))
            pos += n
    M = pos
    qkv = (torch.randn(M, 3 * H * dh, device="cuda") * 1.2).to(torch.bfloat16)
    q, k, v = qkv[:, :H * dh], qkv[:, H * dh:2 * H * dh], qkv[:, 2 * H * dh:]
    d = torch.tensor(desc, dtype=torch.int32, device="cuda")
    mx = max(x[1] for x in desc)
    out = torch.zeros(M, H * dh, device="cuda", dtype=torch.bfloat16)
    fn = lambda: lib.attention(q, k, v, d, len(desc), mx, H, dh, dh ** -0.5, 1, out=out)
    res = {}
    gb = (3 * M * H * dh * 2 + M * H * dh * 2) / 1e9
    for rep in range(2):
        for label, env in (("pack kernel (wave per item)", {}), ("tiled, register-staged", {"DEVQA_ATTENTION_PACK": "0", "DEVQA_ATTENTION_SHORT": "0"}),
                           ("tiled, single-image DMA", {"DEVQA_ATTENTION_PACK": "0", "DEVQA_ATTENTION_SHORT": "1"})):
            os.environ.update(env)
            out.zero_()
            fn()
            res.setdefault(label, out.clone())
            us = t_us(fn, warm=50)
            for k_ in env:
                del os.environ[k_]
            print("%-28s %d sequences, %d rows, max q %d: %7.1f us  (q, k, v, out once: %.2f GB = %.2f TB/s)" % (label, len(desc), M, mx, us, gb, gb / us * 1e3), flush=True)
    vals = list(res.values())
    print("max |pack - tiled|:", (vals[0].float() - vals[1].float()).abs().max().item(), " tiled forms bit-identical:", bool(torch.equal(vals[1], vals[2])))


if __name__ == "__main__":
    main()
