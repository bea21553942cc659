#!/usr/bin/env python3
"""Cosine top-k (IKE / dynamic-eval retrieval, BASELINE config #5 kernel): time per query batch and
algorithmic HBM GB/s = N*D*4 bytes / time.  Corpus f32 [15000,384] ~ N(0,1), queries [Q,384]."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import devqa_amd  # noqa: E402,F401
from devqa_amd import lib  # noqa: E402


def main():
    lib.load()
    rng = np.random.default_rng(20251121)
    N, D = 15000, 384
    corpus = torch.from_numpy(rng.standard_normal((N, D)).astype(np.float32)).cuda()
    out = []
    for Q, k in ((1000, 32), (1000, 5), (1, 5), (1, 32), (4, 32), (64, 32)):
        q = torch.from_numpy(rng.standard_normal((Q, D)).astype(np.float32)).cuda()
        for _ in range(3):
            lib.cosine_topk(corpus, q, k)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        s.record()
        for _ in range(n):
            lib.cosine_topk(corpus, q, k)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / n
        # GPU time of the call alone (HIP events around the launches, the library's slot profiler), raw and with the corpus' inverse
        # norms cached as the product call sites do -- the back-to-back figure above includes the host's per-call work
        gpu_us = {}
        inv = lib.row_inv_norm(corpus)
        for name, kw in (("gpu_us", {}), ("gpu_us_cached_norms", {"corpus_inv_norm": inv})):
            torch.cuda.synchronize()
            lib.profile(1)
            for _ in range(n):
                lib.cosine_topk(corpus, q, k, True, True, **kw)
            lib.profile(0)
            pms, _, pn = lib.profile_read(lib.PROF_COSINE)
            gpu_us[name] = round(1e3 * pms / max(pn, 1), 1)
        out.append({"N": N, "Q": Q, "D": D, "k": k, **gpu_us, "ms_per_batch": round(ms, 4), "queries_per_s": round(Q / ms * 1e3),
                    "algorithmic_GBps": round(N * D * 4 / ms / 1e6, 1), "frac_of_8TBps": round(N * D * 4 / ms / 1e6 / 8000, 4),
                    "gflops": round(2.0 * N * D * Q / ms / 1e6, 1)})
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
