"""Platform-stable synthetic weights and EVQA-shaped inputs.

Every parameter is drawn from ``numpy.random.default_rng([seed, crc32(name)])``
so that the reference-side golden generator (tools/make_goldens.py, runs only
in the build container) and the product/oracle (run anywhere) materialise the
same model without any weight file travelling.
"""
import zlib
import numpy as np


def param_init(name: str, shape, seed: int = 20251121, style: str = "unit") -> np.ndarray:
    """fp32 array for parameter `name`.

    style "unit" (tiny fixtures): 2-D+ weights ~ N(0, 0.8/sqrt(fan_in)), LayerNorm
    weight = 1 + 0.05 N, biases/LN-bias = 0.02 N, token/class/query embeddings 0.5 N,
    positions 0.1 N.
    style "opt" (real-dim models, bench): same except token embeddings 0.05 N,
    positions 0.02 N, and decoder fc1 ~ N(0, 0.15/sqrt(fan_in)) with bias -0.3, which
    makes the ReLU FFN activations sparse (~2% active) like a trained OPT so that the
    FT_VL loop (lr 1e-3, 25 steps) converges over ~20-25 steps with logits of O(1-20)
    instead of saturating in one step."""
    rng = np.random.default_rng([seed, zlib.crc32(name.encode())])
    shape = tuple(int(s) for s in shape)
    z = rng.standard_normal(shape, dtype=np.float32)
    low = name.lower()
    is_ln = ("layer_norm" in low or "layernorm" in low)
    if is_ln and name.endswith("weight"):
        return (1.0 + 0.05 * z).astype(np.float32)
    if style == "opt":
        if "decoder.layers" in low and name.endswith("fc1.bias"):
            return (-0.3 + 0.02 * z).astype(np.float32)
        if "decoder.layers" in low and name.endswith("fc1.weight"):
            return (0.15 / np.sqrt(shape[1]) * z).astype(np.float32)
        if "embed_positions" in low:
            return (0.02 * z).astype(np.float32)
        if "embed_tokens" in low:
            return (0.05 * z).astype(np.float32)
    if name.endswith("bias"):
        return (0.02 * z).astype(np.float32)
    if "embed_positions" in low or "position_embedding" in low:
        return (0.1 * z).astype(np.float32)
    if "embed_tokens" in low or "class_embedding" in low or "query_tokens" in low:
        return (0.5 * z).astype(np.float32)
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        return (0.8 / np.sqrt(fan_in) * z).astype(np.float32)
    return (0.02 * z).astype(np.float32)
