#!/usr/bin/env python3
"""Microbenchmark of devqa_gemm_bf16 on the shapes the BLIP-2 path launches (random operands,
interleaved rounds in one process, HIP-event timing).  Usage: python tools/gemm_bench.py [modes...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import devqa_amd  # noqa: E402,F401
from devqa_amd import lib  # noqa: E402

SHAPES = [("vit_qkv", 16448, 4224, 1408), ("vit_proj", 16448, 1408, 1408), ("vit_fc1", 16448, 6144, 1408),
          ("vit_fc2", 16448, 1408, 6144), ("opt_qkv", 4300, 7680, 2560), ("opt_out", 4300, 2560, 2560),
          ("opt_fc1", 4300, 10240, 2560), ("opt_fc2", 4300, 2560, 10240), ("sq_8k", 8192, 8192, 8192),
          ("lm_head_rows", 48, 50272, 2560), ("tail_fc2", 700, 2560, 10240)]


def main():
    modes = [int(m) for m in sys.argv[1:]] or [1, 0, 2]
    lib.load()
    torch.manual_seed(0)
    for name, M, N, K in SHAPES:
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        res = {}
        for rnd in range(3):
            for mode in modes:
                lib.gemm_set_mode(mode)
                for _ in range(2):
                    lib.gemm(a, w, out_bf16=out)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                n = 10
                for _ in range(n):
                    lib.gemm(a, w, out_bf16=out)
                e.record()
                torch.cuda.synchronize()
                res.setdefault(mode, []).append(s.elapsed_time(e) / n)
        line = "%-13s M=%6d N=%6d K=%6d" % (name, M, N, K)
        for mode in modes:
            t = min(res[mode])
            line += " | mode%d %8.1f us %7.1f TF/s" % (mode, t * 1e3, 2.0 * M * N * K / t / 1e9)
        print(line, flush=True)
    lib.gemm_set_mode(0)


if __name__ == "__main__":
    main()
