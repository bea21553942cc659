#!/usr/bin/env python3
"""FT AdamW sweep on column-compacted states (60 edits x [2560, npad] fp32 x (w, m, v)): GB/s of the variant DEVQA_FT_GROUPED selects
(0: wave per row, 8 / 16: lanes per row; unset: the launcher's choice) + a checksum to compare variants across processes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import devqa_amd  # noqa: E402,F401
from devqa_amd import lib  # noqa: E402

lib.load()
torch.manual_seed(0)
E, Dout, L = 60, 2560, 4
for npad in (208, 272, 288, 320, 512, 1024):
    w0 = torch.randn(Dout, npad, device="cuda") * 0.02
    w0e = w0.unsqueeze(0).expand(E, Dout, npad).contiguous()
    w, m, v = torch.empty_like(w0e), torch.empty_like(w0e), torch.empty_like(w0e)
    a = torch.relu(torch.randn(E, L, npad, device="cuda"))
    dy = torch.randn(E, L, Dout, device="cuda") * 1e-3
    y = torch.empty(E, L, Dout, device="cuda")
    upd = torch.ones(E, dtype=torch.int32, device="cuda")
    t1 = torch.ones(E, dtype=torch.int32, device="cuda")
    t2 = torch.full((E,), 2, dtype=torch.int32, device="cuda")

    def run(tt):
        lib.ft_adamw_step(w, m, v, w0e, a, dy, y, upd, tt, 1e-3, 0.9, 0.999, 1e-8, 0.0, -1.0)
    run(t1)
    run(t2)
    torch.cuda.synchronize()
    chk = float(w.double().sum()), float(y.double().abs().sum()), float(v.double().sum())
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        run(t2)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    print("npad %4d: %7.1f us  %6.0f GB/s   checksum w %.9g |y| %.9g v %.9g" % (npad, us, 24.0 * E * Dout * npad / us / 1e3, *chk), flush=True)
