"""Shared by the LTE_VL tests: a deterministic bag-of-words sentence encoder (stand-in for the reference's
sentence_transformers model, which is not available offline) and the tiny evaluation records."""
import re
import zlib

import numpy as np

DIM = 768


def bow_encode(sentences, dim=DIM):
    """Each lower-cased word adds a fixed pseudo-random +-1 pattern on 8 hashed coordinates: sentences sharing words have a
    high cosine, unrelated ones a cosine near 0 -- both sides of LTE_VL's sim_threshold get exercised."""
    out = np.zeros((len(sentences), dim), np.float32)
    for r, s in enumerate(sentences):
        for w in re.findall(r"[a-z0-9]+", s.lower()):
            h = zlib.crc32(w.encode())
            rng = np.random.default_rng(h)
            idx = rng.integers(0, dim, 8)
            out[r, idx] += rng.choice([-1.0, 1.0], 8).astype(np.float32)
        if not out[r].any():
            out[r, 0] = 1.0
    return out
