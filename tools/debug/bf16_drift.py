#!/usr/bin/env python3
"""Where does the bf16-mode error of the full-depth BLIP-2 forward come from?  Builds the SAME synthetic model (bench recipe) in
fp32 mode and bf16 mode and compares, for one image + text probe: the image tokens (ViT-g + Q-Former + projection), the decoder's
residual stream after selected layers, and the label-row logits -- also for mixed pipelines (fp32 vision -> bf16 decoder and the
reverse), which separates the two towers' contributions.  Diagnostic only."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import devqa_amd  # noqa: E402,F401
from devqa_amd import blip2_spec  # noqa: E402
from devqa_amd.editor.vllms_for_edit.blip2.blip2 import BLIP2OPTForEdit  # noqa: E402
from devqa_amd.editor.vllms_for_edit.blip2.modeling import Blip2Native  # noqa: E402
from devqa_amd.synth import IdTokenizer, param_init, synth_image_u8  # noqa: E402


def rel(a, b):
    return float((a.float() - b.float()).abs().max()) / float(b.float().abs().max())


def rms_rel(a, b):
    return float((a.float() - b.float()).norm()) / float(b.float().norm())


def main(style="opt", n_img=4):
    cfg, dev, seed = blip2_spec.BLIP2_OPT_2_7B, "cuda:0", 20251121
    models = {m: Blip2Native(cfg, dev, m) for m in ("fp32", "bf16")}
    names = list(models["fp32"]._shapes.keys())
    with ThreadPoolExecutor(16) as ex:
        futs = {n: ex.submit(param_init, n, models["fp32"]._shapes[n], seed, style) for n in names}
        for n in names:
            arr = torch.from_numpy(futs.pop(n).result())
            for m in models.values():
                m.load_named_tensors(lambda _n, a=arr: a, names=[n], refresh=False)
    v = {}
    for mode, model in models.items():
        model.refresh_derived(force=True)
        v[mode] = BLIP2OPTForEdit(None, dev, model=model, tokenizer=IdTokenizer())
    proc = v["fp32"].image_processor
    pix = torch.from_numpy(np.stack([proc(synth_image_u8(s, "i1", 224, seed)) for s in range(n_img)])).to(dev)
    it = {m: v[m].engine.encode_images(pix) for m in v}
    print("image tokens [%d, 32, 2560]: bf16 vs fp32  max-rel %.3g  rms-rel %.3g" % (n_img, rel(it["bf16"], it["fp32"]), rms_rel(it["bf16"], it["fp32"])))
    rng = np.random.default_rng(3)
    seqs = [(i, [2] + [int(t) for t in rng.integers(4, 50272, 15)]) for i in range(n_img)]

    def run(dec, tok_src, upto=None):
        eng = v[dec].engine
        ps = eng.pack_from_tokens(seqs, it[tok_src].to(torch.float32))
        if upto is not None:
            os.environ["DEVQA_PATH_ABI"] = "0"
            try:
                x, _ = eng.decoder_layers(ps, upto_layer=upto)
            finally:
                del os.environ["DEVQA_PATH_ABI"]
            return x.clone()
        rows = torch.tensor([ps.start[i] + ps.length[i] - 1 - j for i in range(n_img) for j in range(3)], dtype=torch.int32, device=dev)
        x, _ = eng.decoder_layers(ps)
        from devqa_amd import lib
        return eng.lm_head(lib.gather_rows(x, rows)).clone()
    ref = run("fp32", "fp32")
    for dec, src in (("bf16", "bf16"), ("bf16", "fp32"), ("fp32", "bf16")):
        lg = run(dec, src)
        print("label-row logits: decoder %s on image tokens from the %s tower:  max-rel %.3g  rms-rel %.3g" % (dec, src, rel(lg, ref), rms_rel(lg, ref)))
    for upto in (0, 3, 7, 15, 23, 31):
        a, b = run("bf16", "fp32", upto), run("fp32", "fp32", upto)
        print("decoder residual stream after layer %2d (same fp32 image tokens): max-rel %.3g rms-rel %.3g" % (upto, rel(a, b), rms_rel(a, b)))


if __name__ == "__main__":
    main(*sys.argv[1:2])
