#!/usr/bin/env python3
"""Where does the host wait for the GPU?  Runs one secondary config of tools/bench_configs.py with torch's sync debug mode on and counts
the warning sites (file:line of the first frame inside this repo).  Usage: sync_points.py blip2_mend [n]"""
import collections
import os
import sys
import traceback
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

counts = collections.Counter()


def hook(message, category, filename, lineno, file=None, line=None):
    if "synchroniz" not in str(message):
        return
    for fr in reversed(traceback.extract_stack()[:-1]):
        if ROOT in fr.filename and "sync_points" not in fr.filename and "/tools/bench_configs" not in fr.filename:
            counts["%s:%d %s" % (os.path.relpath(fr.filename, ROOT), fr.lineno, (fr.line or "").strip()[:90])] += 1
            break


warnings.showwarning = hook
warnings.simplefilter("always")
import bench_configs as B  # noqa: E402

cfg, args = sys.argv[1], [int(a) for a in sys.argv[2:]]
orig = B.run_eval


def run_eval(*a, **k):
    torch.cuda.set_sync_debug_mode("warn")
    try:
        return orig(*a, **k)
    finally:
        torch.cuda.set_sync_debug_mode("default")


B.run_eval = run_eval
{"llava_ft": B.llava_ft, "blip2_mend": B.blip2_mend, "minigpt4_ike": B.minigpt4_ike}[cfg](*args)
print("host synchronisation sites (count over warm-up + timed run):")
for k, v in counts.most_common(40):
    print("%6d  %s" % (v, k))
