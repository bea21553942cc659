#!/usr/bin/env python3
"""The ping-pong GEMM on the ViT / OPT shapes of the bench workload with the epilogues the path uses, under two or more
DEVQA gemm modes (default: 0 = production dispatch, 26 = fp32 LDS transposition for every output kind); also checks that the modes
produce identical outputs.  The first mode timed on a new shape pays first-touch effects: list modes twice (e.g. `26 0 26 0`)."""
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
    modes = [int(x) for x in sys.argv[1:]] or [0, 26]
    M = int(os.environ.get("GEMM_M", 32639))        # 32639 = one 127-image ViT chunk; 130556 = the 508 images of a 127-cycle batch
    tot = {m: 0.0 for m in modes}
    for name, N, K, kind, m_rows in (("qkv", 4224, 1408, "bias", M), ("fc1", 6144, 1408, "gelu", M), ("proj", 1408, 1408, "resid", M),
                                     ("fc2", 1408, 6144, "resid", M), ("opt_fc1", 10240, 2560, "relu", 20400),
                                     ("opt_qkv", 7680, 2560, "bias", 20400), ("opt_fc2", 2560, 10240, "resid", 20400)):
        a = torch.randn(m_rows, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
        b = torch.randn(N, device="cuda")
        x0 = torch.randn(m_rows, N, device="cuda")
        outs, line = {}, "%-8s M=%d N=%d K=%d %-5s" % (name, m_rows, N, K, kind)
        for mode in modes:
            lib.gemm_set_mode(mode)
            ob = torch.empty(m_rows, N, device="cuda", dtype=torch.bfloat16)
            x = x0.clone()
            if kind == "bias":
                fn = lambda: lib.gemm(a, w, b, out_bf16=ob)
            elif kind == "gelu":
                fn = lambda: lib.gemm(a, w, b, act=lib.ACT_GELU, out_bf16=ob)
            elif kind == "relu":
                fn = lambda: lib.gemm(a, w, b, act=lib.ACT_RELU, out_bf16=ob)
            else:
                fn = lambda: lib.gemm(a, w, b, residual=x0, out_f32=x)
            fn()
            outs[mode] = (x if kind == "resid" else ob).float().clone()
            us = t_us(fn)
            tot[mode] += us
            line += "  mode %d: %7.1f us %5.0f TF" % (mode, us, 2.0 * m_rows * N * K / us / 1e6)
        same = all(torch.equal(outs[modes[0]], outs[m]) for m in modes[1:] )
        print(line + ("  identical" if same else "  DIFFERENT"), flush=True)
    lib.gemm_set_mode(0)
    print("sum us:", {m: round(v, 1) for m, v in tot.items()})


if __name__ == "__main__":
    main()
