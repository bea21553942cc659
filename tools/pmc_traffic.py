#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in SEPARATE runs,
as MI355X_MICROARCH.md 'HBM' / 'rocprofv3 PMC slots' prescribes: the two counters do not fit one pass).

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
  python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_traffic.json

Units/corrections (same guide): both counters are in KiB; on gfx950 FETCH_SIZE tallies 128-B requests of wide
(16 B/lane) streaming reads at 64 B, so read bytes = 2 x FETCH_SIZE; WRITE_SIZE is exact for 16-B stores.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def collect(d, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                a = acc[row["Kernel_Name"]]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    return acc


def short(name):
    m = re.match(r"(?:void )?([A-Za-z0-9_:]+(?:<[^(]*>)?)", name)
    n = (m.group(1) if m else name)[:90]
    if n.startswith("gemm_bf16_pp_kernel"):   # activation / stream-K template variants of one kernel
        return "gemm_bf16_pp_kernel"
    return n


def main():
    fd, wd, out = sys.argv[1:4]
    fe0, wr0 = collect(fd, "FETCH_SIZE"), collect(wd, "WRITE_SIZE")
    fe, wr = defaultdict(lambda: [0.0, 0]), defaultdict(lambda: [0.0, 0])
    for src, dst in ((fe0, fe), (wr0, wr)):
        for k, (v, n) in src.items():
            dst[short(k)][0] += v
            dst[short(k)][1] += n
    rows = []
    for k in fe:
        f_kib, n = fe[k]
        w_kib, nw = wr.get(k, [0.0, 0])
        rd = 2.0 * f_kib * 1024 / max(n, 1)
        wt = w_kib * 1024 / max(nw, 1)
        rows.append({"kernel": short(k), "launches": n, "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wt),
                     "hbm_bytes_per_launch": round(rd + wt), "total_gb": round((rd + wt) * n / 1e9, 3)})
    rows.sort(key=lambda r: -r["total_gb"])
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 2 --warmup 1",
               "correction": "read = 2 x FETCH_SIZE KiB (gfx950 128-B requests tallied at 64 B); write = WRITE_SIZE KiB",
               "kernels": rows[:24]}, open(out, "w"), indent=1)
    for r in rows[:14]:
        print("%-70s n=%5d rd %10.2f MB wr %9.2f MB" % (r["kernel"][:70], r["launches"], r["read_bytes_per_launch"] / 1e6, r["write_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
