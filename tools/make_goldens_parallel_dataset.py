#!/usr/bin/env python3
"""Golden id sequences for ParallelDataset: runs the REFERENCE's class (dataset/__init__.py:13-114) in this build container
with an identity `get_data_by_ids_func` and stores, per configuration, the id batches of three consecutive `iter()` passes.
Stores data only (tests/golden/parallel_dataset_ids.json)."""
import json
import os
import sys

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference/DE-VQA")
from dataset import ParallelDataset  # noqa: E402

CASES = [dict(sample_count=10, batch_size=3, shuffle=True, drop_last=False, random_seed=7),
         dict(sample_count=10, batch_size=4, shuffle=False, drop_last=False, random_seed=1),
         dict(sample_count=9, batch_size=[2, 4], shuffle=True, drop_last=False, random_seed=3),
         dict(sample_count=7, batch_size=3, shuffle=True, drop_last=True, random_seed=11),
         dict(sample_count=5, batch_size=8, shuffle=True, drop_last=False, random_seed=2),
         dict(sample_count=6, batch_size=1, shuffle=True, drop_last=False, random_seed=5)]


def main():
    out = []
    for c in CASES:
        ds = ParallelDataset(c["sample_count"], lambda ids: [int(i) for i in ids], c["batch_size"], c["shuffle"], 4, c["drop_last"],
                             c["random_seed"], True)
        passes = []
        for _ in range(3):
            passes.append([[d, n] for d, n in ds])
        out.append({"args": c, "passes": passes, "len": len(ds)})
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "parallel_dataset_ids.json"), "w"))
    print("written", [len(p) for o in out for p in o["passes"]])


if __name__ == "__main__":
    main()
