"""Shared by tools/make_goldens_dataset.py (reference side, build container) and the dataset-builder tests: a deterministic
sentence encoder (stand-in for all-MiniLM-L6-v2, unavailable offline) and the recipe of a small stored corpus in the reference's
dict layout {sentences, images, prompts, embeddings} (R/dataset/vllm.py:96-103) that drives `finds_sim` (vllm.py:65-87) through
all of its branches: first hit differs from `pred` (rank 0), an earlier hit carries the same answer (rank >= 1), all five hits
carry the same answer (-> the LAST hit)."""
import re
import zlib

import numpy as np

DIM = 96


def encode(sentences, dim=DIM):
    """Bag of hashed words (+-1 on 6 coordinates per word) plus a little per-sentence noise that breaks exact ties."""
    out = np.zeros((len(sentences), dim), np.float32)
    for r, s in enumerate(sentences):
        for w in re.findall(r"[a-z0-9]+", s.lower()):
            rng = np.random.default_rng(zlib.crc32(w.encode()))
            out[r, rng.integers(0, dim, 6)] += rng.choice([-1.0, 1.0], 6).astype(np.float32)
        out[r] += 0.01 * np.random.default_rng(zlib.crc32(s.encode()) ^ 0x5bd1e995).standard_normal(dim).astype(np.float32)
    return out


def _sent(new_fact, pa):
    return "New Fact: %s\nPrompt: %s\n\n" % (new_fact, pa)


def build_corpus(eval_records):
    """Three entries per record in the layout of R/easyeditor/models/ike/util.py:54-86 (the fact, its rephrase, its locality
    neighbour), then decoys that sit closer to some records' `src` than those and carry the record's `pred` as answer."""
    sentences, images, prompts = [], [], []
    for d in eval_records:
        nf = d["src"] + " " + d["alt"]
        images += [d["image"], d["image_rephrase"], d["m_loc"]]
        prompts += [[d["src"], d["alt"]], [d["rephrase"], d["alt"]], [d["loc"], d["loc_ans"]]]
        sentences += [_sent(nf, nf), _sent(nf, d["rephrase"] + " " + d["alt"]), _sent(nf, d["loc"] + " " + d["loc_ans"])]
    n = len(eval_records)
    for i, d in enumerate(eval_records):
        if i % 4 == 1:      # one decoy with the same answer as `pred`: finds_sim must skip it and take the next hit
            sentences.append(d["src"])
            images.append("decoy/%d_a.jpg" % i)
            prompts.append(["decoy question %d?" % i, d["pred"]])
        if i % 4 == 2:      # five decoys, all with `pred` as answer: no hit differs -> the last of the five
            for k in range(5):
                sentences.append(d["src"] + ["", "!", " .", " ...", " ?!"][k])      # same word bag: cosine ~ 1 with the query
                images.append("decoy/%d_%d.jpg" % (i, k))
                prompts.append(["decoy question %d/%d?" % (i, k), d["pred"]])
        if i % 4 == 3 and i + 1 < n:   # two decoys, second one differs
            sentences += [d["src"], d["src"] + " indeed"]
            images += ["decoy/%d_a.jpg" % i, "decoy/%d_b.jpg" % i]
            prompts += [["decoy question %da?" % i, d["pred"]], ["decoy question %db?" % i, d["pred"] + " not"]]
    return {"sentences": sentences, "images": images, "prompts": prompts}
