#!/usr/bin/env python3
"""Where does the 257th token's cost come from?  Same buffer layout (257 rows per image), descriptors with 256 / 257 queries x 256 / 257 keys."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import devqa_amd  # noqa: E402,F401
from devqa_amd import lib  # noqa: E402
from att_vit_bench import t_us  # noqa: E402


def main():
    lib.load()
    torch.manual_seed(0)
    n_seq, H, dh, S = 508, 16, 88, 257
    M = n_seq * S
    qkv = (torch.randn(M, 3 * H * dh, device="cuda") * 1.5).to(torch.bfloat16)
    q, k, v = qkv[:, :H * dh], qkv[:, H * dh:2 * H * dh], qkv[:, 2 * H * dh:]
    out = torch.zeros(M, H * dh, device="cuda", dtype=torch.bfloat16)
    for tq, tk in ((256, 256), (257, 256), (256, 257), (257, 257), (272, 272), (256, 320)):
        rows = max(tq, tk)
        if rows > S:
            continue
        desc = torch.tensor([[i * S, tq, 0, 0, i * S, tk] for i in range(n_seq)], dtype=torch.int32, device="cuda")
        for label, env in (("ring 9", {}), ("ring 8", {"DEVQA_ATTENTION_NW": "8"}), ("ring 9 no fold", {"DEVQA_ATTENTION_FOLD": "0"}),
                           ("ring 8 no fold", {"DEVQA_ATTENTION_NW": "8", "DEVQA_ATTENTION_FOLD": "0"}), ("2-image 8", {"DEVQA_ATTENTION_RING": "0"})):
            os.environ.update(env)
            fn = lambda: lib.attention(q, k, v, desc, n_seq, max(tq, 224), H, dh, dh ** -0.5, 0, out=out)
            us = t_us(fn)
            for k_ in env:
                del os.environ[k_]
            print("queries %d keys %d  %-16s %8.1f us" % (tq, tk, label, us), flush=True)


if __name__ == "__main__":
    main()
