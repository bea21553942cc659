"""Result-table reader: what R/read_res.py:1-33 prints from the `mean_results.json` files the evaluator writes
(SURVEY 8(f) N2).  Same column order and the same convention: for the non-t3 image/text locality probes the table
holds 1 - acc (an edited answer there is a locality FAILURE), for t3i1 / t3i3 / text_loc it holds acc.

    python -m devqa_amd.read_res eval_results/
"""
import json
import os
import sys

COLUMNS = ["model", "data", "method", "t1i2", "t2i1", "t2i2", "t1i4", "t2i4", "t1i3", "t3i1", "t3i3", "text_loc"]


def collect(root):
    """-> list of rows (first row = header) for every mean_results.json under `root` whose total_mean holds a
    9-probe locality block (read_res.py:12-26); model / data / method come from the directory names the evaluator
    uses: <root>/<method>/<model>/<data>/sequential_edit_<n>/mean_results.json (vllm_editor_eval.py:33-34)."""
    rows = [list(COLUMNS)]
    for d, _, files in sorted(os.walk(root)):
        if "mean_results.json" not in files:
            continue
        f = os.path.join(d, "mean_results.json")
        data = json.load(open(f))
        parts = f.split("/")
        for _, blk in data["total_mean"].items():
            if isinstance(blk, dict) and len(blk) == 9:
                row = [parts[-4], parts[-3], parts[-5]]
                for k in COLUMNS[3:]:
                    v = blk[k]["acc"]
                    row.append(str(1 - v) if ("t3" not in k and k != "text_loc") else str(v))
                rows.append(row)
    return rows


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    for r in collect(argv[0] if argv else "eval_results"):
        print("\t".join(r))


if __name__ == "__main__":
    main()
