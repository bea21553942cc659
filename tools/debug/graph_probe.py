#!/usr/bin/env python3
"""What would a hipGraph of stage A return?    Captures `devqa_vision_encode` (full-depth BLIP-2 vision tower + Q-Former,
the largest launch sequence of a step: ~600 kernels) for one batch of images in a HIP graph and times replay against the eager launch
sequence on the same stream, same buffers, no tracer: HIP events around N back-to-back calls, interleaved, best of 3.
Usage: graph_probe.py [n_images=127] [reps=4]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, t_host * 1e3 / reps


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 127
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    dev = torch.device("cuda:0")
    model, cfg, _ = bench.build_full_model(dev, 0, threads=16)
    from devqa_amd.engine import Blip2Engine
    eng = Blip2Engine(model)
    ctx = eng.path_ctx()
    assert ctx is not None
    size = cfg["vision_config"]["image_size"]
    pixels = torch.randn((n, 3, size, size), device=dev, dtype=torch.float32)
    ref = ctx.vision_encode(pixels)          # warm: one-time attribute / env reads happen outside the capture
    ref = ctx.vision_encode(pixels).clone()
    torch.cuda.synchronize()

    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ctx.vision_encode(pixels)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        out_g = ctx.vision_encode(pixels)
    g.replay()
    torch.cuda.synchronize()
    print("graph output == eager output bit for bit:", bool(torch.equal(out_g, ref)), flush=True)

    best = {"eager": (1e9, 0), "graph": (1e9, 0)}
    for _ in range(3):
        for name, fn in (("eager", lambda: ctx.vision_encode(pixels)), ("graph", g.replay)):
            ms, host = timed(fn, reps)
            if ms < best[name][0]:
                best[name] = (ms, host)
            print("  %-5s %9.3f ms GPU per call   %8.3f ms host per call" % (name, ms, host), flush=True)
    e, gr = best["eager"][0], best["graph"][0]
    print("vision_encode(%d images): eager %.3f ms, graph replay %.3f ms: %+.2f %%" % (n, e, gr, 100.0 * (gr - e) / e))


if __name__ == "__main__":
    main()
