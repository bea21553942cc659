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
    instead of saturating in one step.
    style "survey": SURVEY.md 8(d)'s recipe as written -- every weight N(0, 0.02), LayerNorm weight 1, all biases 0.  The decoder
    FFN is then DENSE (half of the ReLU units fire on any row), which disables the FT loop's column compaction: bench.py's
    `--ffn dense` leg."""
    rng = np.random.default_rng([seed, zlib.crc32(name.encode())])
    shape = tuple(int(s) for s in shape)
    z = rng.standard_normal(shape, dtype=np.float32)
    low = name.lower()
    is_ln = ("layer_norm" in low or "layernorm" in low)
    if style == "survey":   # SURVEY.md 8(d) verbatim: N(0, 0.02) per parameter name, LayerNorm weight = 1, every bias = 0
        if is_ln:
            return np.ones(shape, np.float32) if name.endswith("weight") else np.zeros(shape, np.float32)
        if name.endswith("bias"):
            return np.zeros(shape, np.float32)
        return (0.02 * z).astype(np.float32)
    if is_ln and name.endswith("weight"):
        return (1.0 + 0.05 * z).astype(np.float32)
    if style == "llava":  # CLIP ViT + LLaMA naming (de-vqa_amd/llava_spec.py)
        if "norm" in low and name.endswith("weight") and len(shape) == 1:   # RMSNorm / pre_layrnorm / norm
            return (1.0 + 0.05 * z).astype(np.float32)
        if "class_embedding" in low:
            return (0.5 * z).astype(np.float32)
        if "position_embedding" in low:
            return (0.02 * z).astype(np.float32)
        if "embed_tokens" in low:
            return (0.05 * z).astype(np.float32)
        if name.endswith("mlp.up_proj.weight") and shape[0] > 1024:
            # full-size models only: small SwiGLU activations so FT_VL (lr 1e-3) needs ~20 steps, as with "opt"
            return (0.02 / np.sqrt(shape[1]) * z).astype(np.float32)
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


# ---------------------------------------------------------------------------------------------
# EVQA-shaped synthetic cycles (SURVEY.md 8(d) "Synthetic inputs")
# ---------------------------------------------------------------------------------------------
class IdTokenizer:
    """Stand-in tokenizer for pre-tokenised synthetic data (there is no 50272-entry vocabulary
    offline): prompts/targets are lists of token ids; decode prints the ids."""
    pad_token_id = 1
    eos_token_id = 2
    padding_side = "right"

    def decode(self, ids):
        return " ".join(str(int(i)) for i in ids)


def evqa_cycles(n, vocab=50272, seed=20251121, image_of=None):
    """n synthetic samples with the probe structure of R/dataset/vllm.py:121-254.
    Prompts/targets are token-id lists (prompt incl. the BOS id 2 and the 4-token ' The answer is:'
    suffix; text_loc one more for '?').  `image_of(sample, tag)` supplies the image object for tag in
    {i1, ir, i2, i3} (default: the string key 's<k>_<tag>').  Returns a list of dicts in the
    `BaseVLLMEditData` layout (requests / generality / locality)."""
    rng = np.random.default_rng(seed)
    SUFFIX = [int(t) for t in rng.integers(4, vocab, 4)]  # one fixed 4-token suffix, like ' The answer is:'
    QMARK = int(rng.integers(4, vocab))

    def prompt(extra=0):
        ln = int(np.clip(np.round(rng.normal(14, 3)), 8, 28))
        body = [int(t) for t in rng.integers(4, vocab, max(ln - 5, 1))]
        return [2] + body + SUFFIX + ([QMARK] if extra else [])

    def target(k):
        return [int(t) for t in rng.integers(4, vocab, k)]
    out = []
    for s in range(n):
        img = (lambda tag: image_of(s, tag)) if image_of else (lambda tag: "s%d_%s" % (s, tag))
        P1, P2, P4, P5, P6 = prompt(), prompt(), prompt(), prompt(1), prompt()
        P3 = [2] + [int(t) for t in rng.integers(4, vocab, 12)] + SUFFIX  # an image path used as text
        A = target(int(rng.choice([1, 2, 3], p=[0.7, 0.2, 0.1])))
        B = target(int(rng.choice([1, 2, 3], p=[0.7, 0.2, 0.1])))
        C = target(int(np.clip(rng.geometric(0.3), 1, 16)))
        i1, ir, i2, i3 = img("i1"), img("ir"), img("i2"), img("i3")
        out.append({
            "requests": [{"image": i1, "prompt": P1, "target_new": A}],
            "generality": {"text_rephrase": [{"image": i1, "prompt": P4, "target": A}],
                           "image_rephrase": [{"image": ir, "prompt": P1, "target": A}]},
            "locality": {"text_loc": [{"image": None, "prompt": P5, "target": C}],
                         "t3i3": [{"image": i3, "prompt": P6, "target": B}],
                         "t1i4": [{"image": None, "prompt": P1, "target": A}],
                         "t2i4": [{"image": None, "prompt": P2, "target": A}],
                         "t1i2": [{"image": i2, "prompt": P1, "target": A}],
                         "t1i3": [{"image": i3, "prompt": P1, "target": A}],
                         "t2i1": [{"image": i1, "prompt": P2, "target": A}],
                         "t2i2": [{"image": i2, "prompt": P2, "target": A}],
                         "t3i1": [{"image": i1, "prompt": P3, "target": B}]},
        })
    return out


def synth_image_u8(sample, tag, size=224, seed=20251121):
    """uint8 [size,size,3] i.i.d. U{0..255} image for (sample, tag)."""
    rng = np.random.default_rng([seed, sample, zlib.crc32(tag.encode())])
    return rng.integers(0, 256, size=(size, size, 3), dtype=np.uint8)


def mend_aux_init(name: str, shape, seed: int = 7) -> np.ndarray:
    """Deterministic, finite MEND hyper-network state for fixtures/benchmarks (a trained `Best` checkpoint is the real
    source): low-rank factors of O(1/sqrt(fan_in)), mode scale ~1, normalisation buffers with positive std.  Names are
    the state_dict keys of the reference's `aux_models` / `edit_lrs` (mend_vl.py:254-256)."""
    rng = np.random.default_rng([seed, zlib.crc32(name.encode())])
    shape = tuple(int(s) for s in shape)
    z = rng.standard_normal(shape, dtype=np.float32) if len(shape) else np.float32(rng.standard_normal())
    leaf = name.split(".")[-1]
    if name.startswith("edit_lrs"):
        return np.asarray(2e-2 * (1.0 + 0.1 * abs(float(z))), np.float32).reshape(shape)
    if leaf in ("u_std", "v_std"):
        base = 0.8 if leaf == "u_std" else 0.3
        return (base * (1.0 + 0.2 * np.abs(z))).astype(np.float32)
    if leaf in ("u_mean", "v_mean"):
        base = 0.05 if leaf == "u_mean" else 1e-2
        return (base * z).astype(np.float32)
    if leaf in ("u_s", "v_s"):
        return np.abs(z).astype(np.float32)
    if leaf == "k":
        return np.full(shape, 100.0, np.float32)
    if name.endswith("mode_scale.weight"):
        return (1.0 + 0.1 * z).astype(np.float32)
    if name.endswith("mode_shift.weight") or leaf == "bias":
        return (0.05 * z).astype(np.float32)
    if leaf == "u":   # [outf, rank]
        return (0.5 / np.sqrt(shape[1]) * z).astype(np.float32)
    if leaf == "v":   # [rank, inf]
        return (1.0 / np.sqrt(shape[1]) * z).astype(np.float32)
    return (0.02 * z).astype(np.float32)
