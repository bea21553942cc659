#!/usr/bin/env python3
"""Latency of ONE encode_images call (ViT-g + Q-Former, full BLIP-2 dims) at small batch -- the generic per-sample evaluator's regime
(MEND_VL / IKE_VL / TP_VL / LTE_VL edit one sample at a time), where the GEMMs are weight streams."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import bench_configs as BC  # noqa: E402


def main():
    from devqa_amd import blip2_spec
    from devqa_amd.editor.vllms_for_edit.blip2.modeling import Blip2Native
    from devqa_amd.engine import Blip2Engine
    model = Blip2Native(blip2_spec.BLIP2_OPT_2_7B, BC.DEV, "bf16")
    BC.fill(model, 20251121, "opt")
    eng = Blip2Engine(model)
    for B in (1, 2, 4, 16):
        pix = torch.randn(B, 3, 224, 224, device="cuda")
        eng.encode_images(pix)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(5):
            eng.encode_images(pix)
        torch.cuda.synchronize()
        print("B=%2d encode_images %.2f ms" % (B, (time.time() - t0) / 5 * 1e3), flush=True)


if __name__ == "__main__":
    main()
