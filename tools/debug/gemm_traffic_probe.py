#!/usr/bin/env python3
"""L2-miss read bytes of `gemm_bf16_pp_kernel` per shape (what `roofline.traffic` is made of).  Three roles:
  run   (under `rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d DIR -- python3 tools/debug/gemm_traffic_probe.py run`):
        launches each shape of SHAPES five times, in order, and nothing else on the ping-pong kernel.
  time  (`python tools/debug/gemm_traffic_probe.py time`, no profiler): the same shapes timed with HIP events.
  parse (`python tools/debug/gemm_traffic_probe.py parse DIR`): groups the kernel's dispatches in launch order, prints per shape the
        counter (2 x FETCH_SIZE KiB, the guide's gfx950 correction) beside the algorithmic bytes, the no-reuse bytes (every 256x256 tile reads
        its two panels) and the floor of an L2 that serves one XCD's 32 co-resident tiles perfectly.
The first shape is the calibration: ONE column tile, so every A panel is read by exactly one workgroup and W stays in L2 -- its counter must equal
M*K*2 if the 2x correction applies to the kernel's LDS-DMA loads (16 B per lane, 128 B per row and K-tile)."""
import csv
import glob
import os
import sys

SHAPES = (("calib_1col", 65536, 256, 1408), ("vit_qkv", 130556, 4224, 1408), ("vit_fc1", 130556, 6144, 1408), ("vit_proj", 130556, 1408, 1408),
          ("vit_fc2", 130556, 1408, 6144), ("opt_fc1", 20400, 10240, 2560), ("opt_qkv", 20400, 7680, 2560), ("opt_fc2", 20400, 2560, 10240),
          ("square_8192", 8192, 8192, 8192), ("fc2_5col", 130556, 1280, 6144), ("fc2_6col", 130556, 1536, 6144), ("proj_5col", 130556, 1280, 1408),
          ("proj_6col", 130556, 1536, 1408))
REPS = 5
if os.environ.get("GTP_ONLY"):
    SHAPES = tuple(sh for sh in SHAPES if sh[0] in os.environ["GTP_ONLY"].split(","))


def run(timing=False):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import torch
    import devqa_amd  # noqa: F401
    from devqa_amd import lib
    lib.load()
    for name, M, N, K in SHAPES:
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
        ob = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        torch.cuda.synchronize()
        if timing:          # no profiler attached: HIP events around 20 launches after 3 warm-up launches, best of 3
            best = 1e9
            for _ in range(3):
                for _ in range(3):
                    lib.gemm(a, w, None, out_bf16=ob)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    lib.gemm(a, w, None, out_bf16=ob)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
            tiles = -(-M // 256) * (N / 256.0)
            print("%-12s %26s %9.1f us  %6.0f TFLOP/s  %7.3f us per full-tile equivalent x 256 CUs  checksum %.6f" % (
                name, "%d x %d x %d" % (M, N, K), best, 2.0 * M * N * K / best / 1e6, best / tiles * 256, ob.double().abs().sum().item()), flush=True)
        else:
            for _ in range(REPS):
                lib.gemm(a, w, None, out_bf16=ob)
        torch.cuda.synchronize()
        del a, w, ob


def parse(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == "FETCH_SIZE" and "gemm_bf16_pp_kernel" in r["Kernel_Name"]:
                    rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    rows.sort()
    assert len(rows) == REPS * len(SHAPES), (len(rows), "dispatches of the ping-pong kernel; expected", REPS * len(SHAPES))
    print("%-12s %26s %10s %10s %10s %10s %8s" % ("shape", "M x N x K", "counted MB", "algor. MB", "no-reuse", "XCD floor", "x algor."))
    for i, (name, M, N, K) in enumerate(SHAPES):
        vals = [v for _, v in rows[i * REPS:(i + 1) * REPS]][1:]            # first launch of a shape: cold L2 / MALL
        counted = 2.0 * 1024 * sum(vals) / len(vals)
        alg = 2.0 * K * (M + N)
        tm, tn = -(-M // 256), N / 256.0
        panel = 256 * K * 2.0
        no_reuse = tm * tn * 2 * panel
        # 32 co-resident tiles of an XCD arranged r x c (r*c = 32, c <= column tiles): r + c panel reads per 32 tiles
        best = min((r + c) / (r * c) for r in range(1, 33) for c in range(1, 33) if r * c == 32 and c <= max(1, int(tn + 0.5)))
        print("%-12s %26s %10.0f %10.0f %10.0f %10.0f %8.2f" % (name, "%d x %d x %d" % (M, N, K), counted / 1e6, alg / 1e6, no_reuse / 1e6,
                                                              tm * tn * best * panel / 1e6, counted / alg))


if __name__ == "__main__":
    if sys.argv[1] in ("run", "time"):
        run(sys.argv[1] == "time")
    else:
        parse(sys.argv[2])
