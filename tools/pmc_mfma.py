#!/usr/bin/env python3
"""MFMA utilisation per kernel from ONE rocprofv3 PMC pass:

  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace --output-format csv \
            -d gpurun_out/pmc_mfma -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --ffn sparse --no-hbm-micro
  python tools/pmc_mfma.py gpurun_out/pmc_mfma profiles/rNN_pmc_mfma.json

mfma_bf16_tflops = MOPS x 512 / summed kernel duration (the MfmaFlopsBF16 expression of `rocprofv3 -L`); mfma_busy_frac =
SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) -- GRBM_GUI_ACTIVE is reported summed over the 8 XCDs; clock_ghz =
GRBM_GUI_ACTIVE / 8 / summed duration (the clock the chip held inside these dispatches)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.match(r"(?:void )?([A-Za-z0-9_:]+(?:<[^(]*>)?)", name)
    n = (m.group(1) if m else name)[:80]
    return "gemm_bf16_pp_kernel" if n.startswith("gemm_bf16_pp_kernel") else n


def main():
    d, out = sys.argv[1], sys.argv[2]
    dur = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    acc = defaultdict(lambda: defaultdict(float))
    seen = defaultdict(set)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                k = short(r["Kernel_Name"])
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                seen[k].add(r["Dispatch_Id"])
    rows = []
    for k, c in acc.items():
        ns = sum(dur.get(i, 0) for i in seen[k])
        if ns <= 0:
            continue
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        rows.append({"kernel": k, "launches": len(seen[k]), "total_ms": round(ns / 1e6, 2),
                     "SQ_INSTS_VALU_MFMA_MOPS_BF16": c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0),
                     "mfma_bf16_tflops": round(c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) * 512 / ns / 1e3, 1),
                     "SQ_VALU_MFMA_BUSY_CYCLES": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), "GRBM_GUI_ACTIVE": gui,
                     "mfma_busy_frac": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8 * 1024), 3) if gui else None,
                     "clock_ghz": round(gui / 8 / ns, 3) if gui else None})
    rows.sort(key=lambda r: -r["total_ms"])
    json.dump({"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace -- "
                         "python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --ffn sparse --no-hbm-micro",
               "notes": __doc__.split("\n\n")[-1].replace("\n", " "), "kernels": rows[:12]}, open(out, "w"), indent=1)
    for r in rows[:8]:
        print("%-60s n=%5d %8.1f ms  %7.1f TF  busy %s  clock %s GHz" % (r["kernel"][:60], r["launches"], r["total_ms"], r["mfma_bf16_tflops"],
                                                                      r["mfma_busy_frac"], r["clock_ghz"]))


if __name__ == "__main__":
    main()
